// Probe for the gfx950 sparse MFMA (v_smfmac_f32_16x16x32_f16 / 16x16x64_f16) operand layouts and index semantics.
// Built by tools/probes/run_smfmac_probe.py on the GPU box; not part of the product library.
#include <hip/hip_runtime.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int ABID> __global__ void k32(const h4* a, const h8* b, f4* c, const int* idx) {
  int l = threadIdx.x;
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_smfmac_f32_16x16x32_f16(a[l], b[l], acc, idx[l], 0, ABID);
  c[l] = acc;
}
template <int ABID> __global__ void k64(const h8* a, const h16* b, f4* c, const int* idx) {
  int l = threadIdx.x;
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a[l], b[l], acc, idx[l], 0, ABID);
  c[l] = acc;
}
extern "C" int probe32(const void* a, const void* b, void* c, const void* idx, int abid) {
  if (abid == 0) k32<0><<<1, 64>>>((const h4*)a, (const h8*)b, (f4*)c, (const int*)idx);
  else if (abid == 1) k32<1><<<1, 64>>>((const h4*)a, (const h8*)b, (f4*)c, (const int*)idx);
  else if (abid == 2) k32<2><<<1, 64>>>((const h4*)a, (const h8*)b, (f4*)c, (const int*)idx);
  else k32<3><<<1, 64>>>((const h4*)a, (const h8*)b, (f4*)c, (const int*)idx);
  return (int)hipDeviceSynchronize();
}
extern "C" int probe64(const void* a, const void* b, void* c, const void* idx, int abid) {
  if (abid == 0) k64<0><<<1, 64>>>((const h8*)a, (const h16*)b, (f4*)c, (const int*)idx);
  else k64<1><<<1, 64>>>((const h8*)a, (const h16*)b, (f4*)c, (const int*)idx);
  return (int)hipDeviceSynchronize();
}
