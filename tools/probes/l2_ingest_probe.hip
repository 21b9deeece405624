// Micro-probe (not part of the product): per-CU ingest rate of 16-byte loads when every CU re-reads the SAME
// L2-resident buffer (the activation tile of a GEMM), alone and mixed 1:1 with a private once-read HBM stream (the
// weights). Answers: is the ~12 B/clk/CU the int4 GEMMs see a property of L2 hits or of the kernels?
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/l2_ingest_probe.hip -o /tmp/l2_ingest_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// each wave walks the shared buffer `reps` times; PF loads in flight per wave; MIX = also stream a private region
// HOT hot loads and COLD cold loads per ring slot
template <int PF, int HOT, int COLD>
__global__ __launch_bounds__(256) void ingest2(const u32x4* __restrict__ hot, int hot_n16, int iters,
                                               const u32x4* __restrict__ cold, size_t cold_per_wave16, unsigned* out) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  u32x4 acc = {0, 0, 0, 0};
  const int steps = hot_n16 / 64;
  int pos = (wave_global * 37) % steps;
  const u32x4* cp = cold + (size_t)wave_global * cold_per_wave16 + lane;
  u32x4 hring[PF][HOT > 0 ? HOT : 1], cring[PF][COLD > 0 ? COLD : 1];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
#pragma unroll
    for (int c = 0; c < COLD; ++c) { cring[i][c] = __builtin_nontemporal_load(cp); cp += 64; }
#pragma unroll
    for (int h = 0; h < HOT; ++h) { hring[i][h] = hot[(size_t)pos * 64 + lane]; pos = pos + 1 == steps ? 0 : pos + 1; }
  }
  for (int s = 0; s < iters; s += PF) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
#pragma unroll
      for (int c = 0; c < COLD; ++c) { acc ^= cring[i][c]; cring[i][c] = __builtin_nontemporal_load(cp); cp += 64; }
#pragma unroll
      for (int h = 0; h < HOT; ++h) { acc ^= hring[i][h]; hring[i][h] = hot[(size_t)pos * 64 + lane]; pos = pos + 1 == steps ? 0 : pos + 1; }
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

// wave-specialised: waves with (wave % 4) < HW read only the hot buffer, the others only their private cold stream
template <int PF, int HW>
__global__ __launch_bounds__(256) void ingest3(const u32x4* __restrict__ hot, int hot_n16, int iters_hot, int iters_cold,
                                               const u32x4* __restrict__ cold, size_t cold_per_wave16, unsigned* out) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wave_global = blockIdx.x * (blockDim.x >> 6) + wave;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 ring[PF];
  if (wave < HW) {
    const int steps = hot_n16 / 64;
    int pos = (wave_global * 37) % steps;
#pragma unroll
    for (int i = 0; i < PF; ++i) { ring[i] = hot[(size_t)pos * 64 + lane]; pos = pos + 1 == steps ? 0 : pos + 1; }
    for (int s = 0; s < iters_hot; s += PF) {
#pragma unroll
      for (int i = 0; i < PF; ++i) { acc ^= ring[i]; ring[i] = hot[(size_t)pos * 64 + lane]; pos = pos + 1 == steps ? 0 : pos + 1; }
    }
  } else {
    const u32x4* cp = cold + (size_t)wave_global * cold_per_wave16 + lane;
#pragma unroll
    for (int i = 0; i < PF; ++i) { ring[i] = __builtin_nontemporal_load(cp); cp += 64; }
    for (int s = 0; s < iters_cold; s += PF) {
#pragma unroll
      for (int i = 0; i < PF; ++i) { acc ^= ring[i]; ring[i] = __builtin_nontemporal_load(cp); cp += 64; }
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

template <int PF, int HW>
void run3(const u32x4* hot, size_t hot_bytes, const u32x4* cold, size_t cold_bytes, int wgs, int hot_per_cold, unsigned* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters_cold = 1024;                                       // 1 MiB per cold wave
  const int iters_hot = iters_cold * hot_per_cold * (4 - HW) / HW;   // total hot bytes = hot_per_cold x total cold bytes
  const size_t per_wave16 = (size_t)(iters_cold + PF) * 64 + 64;
  for (int it = 0; it < 2; ++it) ingest3<PF, HW><<<wgs, 256>>>(hot, (int)(hot_bytes / 16), iters_hot, iters_cold, cold, per_wave16, out);
  hipEventRecord(e0);
  ingest3<PF, HW><<<wgs, 256>>>(hot, (int)(hot_bytes / 16), iters_hot, iters_cold, cold, per_wave16, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double hb = (double)wgs * HW * iters_hot * 1024.0, cb = (double)wgs * (4 - HW) * iters_cold * 1024.0;
  printf("specialised %d hot waves of 4, hot bytes = %d x cold, wgs %4d PF %2d : %8.1f us  hot %6.2f TB/s  cold(HBM) %6.2f TB/s (if both ran the whole time)\n",
         HW, hot_per_cold, wgs, PF, ms * 1e3, hb / ms / 1e9, cb / ms / 1e9);
}

template <int PF, int HOT, int COLD>
void run2(const u32x4* hot, size_t hot_bytes, const u32x4* cold, size_t cold_bytes, int wgs, unsigned* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int waves = wgs * 4;
  const int iters = 2048 / (HOT + COLD) / PF * PF;  // ring slots per wave: ~2 MiB per wave in total
  const size_t per_wave16 = (size_t)(iters + PF) * COLD * 64 + 64;
  if (COLD && per_wave16 * 16 * waves > cold_bytes) { printf("cold buffer too small\n"); return; }
  for (int it = 0; it < 2; ++it) ingest2<PF, HOT, COLD><<<wgs, 256>>>(hot, (int)(hot_bytes / 16), iters, cold, per_wave16, out);
  hipEventRecord(e0);
  ingest2<PF, HOT, COLD><<<wgs, 256>>>(hot, (int)(hot_bytes / 16), iters, cold, per_wave16, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double hb = (double)waves * iters * HOT * 1024.0, cb = (double)waves * iters * COLD * 1024.0;
  printf("hot:cold %d:%d wgs %4d PF %2d : %8.1f us  hot %6.2f TB/s  cold(HBM) %6.2f TB/s  total %6.1f B/clk/CU\n", HOT, COLD, wgs, PF,
         ms * 1e3, hb / ms / 1e9, cb / ms / 1e9, (hb + cb) / (ms * 1e-3) / 2.1e9 / 256.0);
}

template <int PF, int MIX>
__global__ __launch_bounds__(256) void ingest(const u32x4* __restrict__ hot, int hot_n16, int reps,
                                              const u32x4* __restrict__ cold, size_t cold_per_wave16, unsigned* out) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  u32x4 acc = {0, 0, 0, 0};
  const int steps = hot_n16 / 64;  // wave-instructions per pass
  // waves start at different offsets so that they do not all hit one channel at once
  int pos = (wave_global * 37) % steps;
  const u32x4* cp = cold + (size_t)wave_global * cold_per_wave16 + lane;
  const int total = steps * reps;
  u32x4 ring[PF], cring[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    ring[i] = hot[(size_t)pos * 64 + lane];
    pos = pos + 1 == steps ? 0 : pos + 1;
    if (MIX) { cring[i] = __builtin_nontemporal_load(cp); cp += 64; }
  }
  for (int s = 0; s < total; s += PF) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      acc ^= ring[i];
      ring[i] = hot[(size_t)pos * 64 + lane];
      pos = pos + 1 == steps ? 0 : pos + 1;
      if (MIX) { acc ^= cring[i]; cring[i] = __builtin_nontemporal_load(cp); cp += 64; }
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

template <int PF, int MIX>
void run(const char* name, const u32x4* hot, size_t hot_bytes, const u32x4* cold, size_t cold_bytes, int wgs, unsigned* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int hot_n16 = (int)(hot_bytes / 16);
  const int steps = hot_n16 / 64;
  const int waves = wgs * 4;
  // each wave: reps passes; choose reps so that a wave moves ~8 MiB
  int reps = (int)((8u << 20) / hot_bytes);
  if (reps < 1) reps = 1;
  const size_t per_wave16 = (size_t)steps * reps * 64 + 64 * PF;
  if (MIX && per_wave16 * 16 * waves > cold_bytes) { printf("%s: cold buffer too small\n", name); return; }
  for (int it = 0; it < 2; ++it) ingest<PF, MIX><<<wgs, 256>>>(hot, hot_n16, reps, cold, per_wave16, out);
  hipEventRecord(e0);
  ingest<PF, MIX><<<wgs, 256>>>(hot, hot_n16, reps, cold, per_wave16, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)waves * steps * reps * 1024.0 * (MIX ? 2 : 1);
  const double clk = 2.1e9;  // nominal; the ratio is what matters
  printf("%-34s hot %5zu KiB wgs %4d PF %2d : %8.1f us  %7.2f TB/s  %6.1f B/clk/CU (at 2.1 GHz)\n", name, hot_bytes >> 10, wgs, PF,
         ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e-3) / clk / 256.0);
}

int main() {
  unsigned* out;
  hipMalloc(&out, 4);
  u32x4 *hot, *cold;
  const size_t cold_bytes = (size_t)20 << 30;
  hipMalloc(&hot, 8 << 20);
  hipMalloc(&cold, cold_bytes);
  hipMemset(hot, 1, 8 << 20);
  hipMemset(cold, 2, cold_bytes);
  for (size_t hb : {(size_t)512 << 10, (size_t)2 << 20}) {
    for (int wgs : {256, 512, 1024}) {
      run<4, 0>("shared L2-hot only", hot, hb, cold, cold_bytes, wgs, out);
      run<8, 0>("shared L2-hot only", hot, hb, cold, cold_bytes, wgs, out);
      run<16, 0>("shared L2-hot only", hot, hb, cold, cold_bytes, wgs, out);
    }
    for (int wgs : {256, 512}) {
      run<4, 1>("hot + private HBM stream 1:1", hot, hb, cold, cold_bytes, wgs, out);
      run<8, 1>("hot + private HBM stream 1:1", hot, hb, cold, cold_bytes, wgs, out);
    }
  }
  for (int wgs : {256, 512}) {
    run2<4, 0, 1>(hot, 512 << 10, cold, cold_bytes, wgs, out);
    run2<8, 0, 1>(hot, 512 << 10, cold, cold_bytes, wgs, out);
    run2<16, 0, 1>(hot, 512 << 10, cold, cold_bytes, wgs, out);
    run2<4, 1, 1>(hot, 512 << 10, cold, cold_bytes, wgs, out);
    run2<4, 2, 1>(hot, 512 << 10, cold, cold_bytes, wgs, out);
    run2<4, 4, 1>(hot, 512 << 10, cold, cold_bytes, wgs, out);
    run2<8, 4, 1>(hot, 512 << 10, cold, cold_bytes, wgs, out);
    run2<4, 8, 1>(hot, 512 << 10, cold, cold_bytes, wgs, out);
  }
  for (int wgs : {256, 512}) {
    run3<8, 1>(hot, 512 << 10, cold, cold_bytes, wgs, 1, out);
    run3<8, 1>(hot, 512 << 10, cold, cold_bytes, wgs, 2, out);
    run3<8, 1>(hot, 512 << 10, cold, cold_bytes, wgs, 4, out);
    run3<16, 1>(hot, 512 << 10, cold, cold_bytes, wgs, 4, out);
    run3<8, 2>(hot, 512 << 10, cold, cold_bytes, wgs, 4, out);
  }
  return 0;
}
