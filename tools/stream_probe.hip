// Micro-probe (not part of the product): how fast can 256 CUs stream a [K/16][N*2] int32 matrix
//  (a) linearly, 16 B / lane, grid-stride;
//  (b) in the Marlin access pattern of marlin_gemm_kernel (wave = 64-col group x K slice, 2 x 8-B loads per 32-k step)
//      with a register ring of PF k-steps, no compute (xor-reduce to keep the loads alive).
// build: hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void linear_read(const u32x4* __restrict__ src, size_t n16, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) acc ^= src[i];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

template <int PF, int BYTES>
__global__ __launch_bounds__(256) void marlin_pattern(const unsigned* __restrict__ b, int K, int N, int splits, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15, c8 = li & 7, hi = li >> 3;
  const int n0 = blockIdx.x * 64;
  const size_t row_words = (size_t)N * 2;
  const int total_steps = K / 32;
  const int workers = splits * 4;
  const int per = (total_steps + workers - 1) / workers;
  const int worker = blockIdx.y * 4 + wave;
  const int s0 = min(worker * per, total_steps), s1 = min(s0 + per, total_steps);
  const unsigned* bw;
  if (BYTES == 8) bw = b + (size_t)(n0 / 64) * 128 + (4 * c8 + g) * 4 + 2 * hi;
  else bw = b + (size_t)(n0 / 64) * 128 + (lane & 31) * 4;  // 16-B: lanes 0-31 row 0, 32-63 row 1
  unsigned acc = 0;
  if (BYTES == 8) {
    u32x2 ring[PF][2];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      int ks = min(s0 + i, total_steps - 1);
      ring[i][0] = *(const u32x2*)(bw + (size_t)(2 * ks) * row_words);
      ring[i][1] = *(const u32x2*)(bw + (size_t)(2 * ks + 1) * row_words);
    }
    for (int s = s0; s < s1; s += PF) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        acc ^= ring[i][0][0] ^ ring[i][0][1] ^ ring[i][1][0] ^ ring[i][1][1];
        int ks = min(s + i + PF, total_steps - 1);
        ring[i][0] = *(const u32x2*)(bw + (size_t)(2 * ks) * row_words);
        ring[i][1] = *(const u32x2*)(bw + (size_t)(2 * ks + 1) * row_words);
      }
    }
  } else {
    u32x4 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      int ks = min(s0 + i, total_steps - 1);
      ring[i] = *(const u32x4*)(bw + (size_t)(2 * ks + (lane >> 5)) * row_words);
    }
    for (int s = s0; s < s1; s += PF) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        acc ^= ring[i][0] ^ ring[i][1] ^ ring[i][2] ^ ring[i][3];
        int ks = min(s + i + PF, total_steps - 1);
        ring[i] = *(const u32x4*)(bw + (size_t)(2 * ks + (lane >> 5)) * row_words);
      }
    }
  }
  if (acc == 0x12345678u) out[0] = 1;
}

int main() {
  const int shapes[3][2] = {{4096, 28672}, {4096, 6144}, {14336, 4096}};
  unsigned* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (auto& sh : shapes) {
    const int K = sh[0], N = sh[1];
    const size_t bytes = (size_t)K * N / 2;
    // 8 distinct copies so that the 256 MiB infinity cache does not help
    const int NC = 8;
    unsigned* buf[NC];
    for (int c = 0; c < NC; ++c) { hipMalloc(&buf[c], bytes); hipMemset(buf[c], 0x5a + c, bytes); }
    auto timeit = [&](const char* name, auto launch) {
      for (int c = 0; c < NC; ++c) launch(buf[c]);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int reps = 5;
      for (int r = 0; r < reps; ++r)
        for (int c = 0; c < NC; ++c) launch(buf[c]);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double us = ms * 1e3 / (reps * NC);
      printf("K=%5d N=%5d %-34s %8.2f us  %7.1f GB/s\n", K, N, name, us, bytes / us / 1e3);
    };
    timeit("linear 16B grid=2048x256", [&](unsigned* p) { linear_read<<<2048, 256>>>((const u32x4*)p, bytes / 16, out); });
    timeit("linear 16B grid=8192x256", [&](unsigned* p) { linear_read<<<8192, 256>>>((const u32x4*)p, bytes / 16, out); });
    for (int splits : {1, 2, 4, 8}) {
      char nm[64];
      snprintf(nm, 64, "marlin 8B  PF=8 splits=%d", splits);
      timeit(nm, [&](unsigned* p) { marlin_pattern<8, 8><<<dim3(N / 64, splits), 256>>>(p, K, N, splits, out); });
      snprintf(nm, 64, "marlin 8B  PF=16 splits=%d", splits);
      timeit(nm, [&](unsigned* p) { marlin_pattern<16, 8><<<dim3(N / 64, splits), 256>>>(p, K, N, splits, out); });
      snprintf(nm, 64, "marlin 16B PF=8 splits=%d", splits);
      timeit(nm, [&](unsigned* p) { marlin_pattern<8, 16><<<dim3(N / 64, splits), 256>>>(p, K, N, splits, out); });
      snprintf(nm, 64, "marlin 16B PF=16 splits=%d", splits);
      timeit(nm, [&](unsigned* p) { marlin_pattern<16, 16><<<dim3(N / 64, splits), 256>>>(p, K, N, splits, out); });
    }
    for (int c = 0; c < NC; ++c) hipFree(buf[c]);
  }
  return 0;
}
