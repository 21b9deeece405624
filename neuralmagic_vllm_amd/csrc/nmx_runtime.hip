// Error plumbing, version string and device-attribute queries of libnmx_hip.
// Device utilities replace csrc/cuda_utils_kernels.cu of the reference.
#include <stdarg.h>
#include <stdlib.h>
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

// ---- tuning registry: environment read once at load, nmx_tuning_set() for sweeps / tests ----
namespace {
const char* const kTuneNames[NMX_TUNE_COUNT] = {
    "NMX_GEMM_CFG", "NMX_GEMM_LEAN", "NMX_GEMM_LARGE", "NMX_GEMM_LARGE_NGRP", "NMX_GEMM_WIDE", "NMX_ATTN_NW",
    "NMX_PREFILL_GQ", "NMX_MM_NO_LDS", "NMX_MM_NT", "NMX_AWQ_NO_RING", "NMX_GPTQ_NO_RING", "NMX_GPTQ_NT", "NMX_ATTN_FP8W", "NMX_MM_TILE",
    "NMX_GEMM_XCD_SPLIT", "NMX_GEMM_DMA", "NMX_SLAB_F32", "NMX_ATTN_PART", "NMX_GEMM_NORM_ROWS", "NMX_GEMM_ATTN"};
struct TuneTable {
  char value[NMX_TUNE_COUNT][64];
  bool set[NMX_TUNE_COUNT];
  TuneTable() {
    for (int i = 0; i < NMX_TUNE_COUNT; ++i) {
      const char* e = getenv(kTuneNames[i]);
      set[i] = e != nullptr;
      value[i][0] = 0;
      if (e != nullptr) { strncpy(value[i], e, sizeof(value[i]) - 1); value[i][sizeof(value[i]) - 1] = 0; }
    }
  }
};
TuneTable g_tune;  // constructed when the library is loaded
}  // namespace

const char* nmx_tune(int id) { return (id >= 0 && id < NMX_TUNE_COUNT && g_tune.set[id]) ? g_tune.value[id] : nullptr; }

extern "C" int nmx_tuning_set(const char* name, const char* value) {
  NMX_CHECK(name != nullptr, NMX_ERR_INVALID_ARG, "tuning name is null");
  for (int i = 0; i < NMX_TUNE_COUNT; ++i) {
    if (strcmp(name, kTuneNames[i]) != 0) continue;
    g_tune.set[i] = value != nullptr;
    if (value != nullptr) { strncpy(g_tune.value[i], value, sizeof(g_tune.value[i]) - 1); g_tune.value[i][sizeof(g_tune.value[i]) - 1] = 0; }
    return NMX_OK;
  }
  nmx_set_error("unknown tuning variable %s", name);
  return NMX_ERR_INVALID_ARG;
}

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
