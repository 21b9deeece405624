// Error plumbing, version string and device-attribute queries of libnmx_hip.
// Device utilities replace csrc/cuda_utils_kernels.cu of the reference.
#include <stdarg.h>
#include <string.h>

#include "nmx_common.h"

static thread_local char g_err[512] = "";

void nmx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* nmx_last_error(void) { return g_err; }

extern "C" const char* nmx_version(void) { return "nmx 0.1 (gfx950, wave64, hand-written HIP)"; }

extern "C" int nmx_get_max_shared_memory_per_block_device_attribute(int device, int* value) {
  NMX_CHECK(value != nullptr, NMX_ERR_INVALID_ARG, "value pointer is null");
  // csrc/cuda_utils_kernels.cu:18-29: cudaDevAttrMaxSharedMemoryPerBlockOptin; on ROCm the opt-in and plain limits coincide
  NMX_HIP(hipDeviceGetAttribute(value, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
  return NMX_OK;
}

extern "C" int nmx_get_device_attribute(int attribute, int device, int* value) {
  NMX_CHECK(value != nullptr, NMX_ERR_INVALID_ARG, "value pointer is null");
  NMX_HIP(hipDeviceGetAttribute(value, static_cast<hipDeviceAttribute_t>(attribute), device));
  return NMX_OK;
}
