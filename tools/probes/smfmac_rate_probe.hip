// Issue rate of the sparse / dense fp16 MFMA forms on gfx950: one wave per SIMD, 16 independent accumulators, 4096 instructions
// each, 1 / 2 / 4 waves per SIMD; prints cycles per instruction and SIMD (s_memtime, shader clock). Build: hipcc --offload-arch=gfx950 -O3 -o exp/smfmac_rate_probe tools/probes/smfmac_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(1024) void rate(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f16x8 a8, b8;
  f16x4 a4;
  f16x16 b16;
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.001f + i); b8[i] = (_Float16)(i * 0.5f); }
  for (int i = 0; i < 4; ++i) a4[i] = a8[i];
  for (int i = 0; i < 16; ++i) b16[i] = (_Float16)(i * 0.25f);
  const int idx = threadIdx.x * 0x01010101;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
      else if constexpr (KIND == 1) acc[i] = __builtin_amdgcn_smfmac_f32_16x16x32_f16(a4, b8, acc[i], idx, 0, 0);
      else if constexpr (KIND == 2) acc[i] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a8, b16, acc[i], idx, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, a4, acc[i], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 1024 * 8);
  const int iters = 256;
  for (int threads : {256, 512, 1024})   // 1, 2, 4 waves per SIMD
  for (int wgs : {1, 256}) {
    rate<KIND><<<wgs, threads>>>(out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    rate<KIND><<<wgs, threads>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per_simd = iters * 16.0 * (threads / 256);
    printf("%-28s %d waves/SIMD, wgs %3d: %.2f us per launch, %.2f ns per instruction and SIMD (wall), %.1f shader cycles per instruction and SIMD\n", name,
           threads / 256, wgs, ms * 1e3, ms * 1e6 / per_simd, (double)h / per_simd);
  }
}
int main() {
  run<0>("v_mfma_f32_16x16x32_f16");
  run<3>("v_mfma_f32_16x16x16_f16");
  run<1>("v_smfmac_f32_16x16x32_f16");
  run<2>("v_smfmac_f32_16x16x64_f16");
  return 0;
}
