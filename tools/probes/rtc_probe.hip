// Calibrates s_memrealtime / s_memtime against HIP events: one wave spins for TICKS real-time ticks.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
  unsigned long long r0, c0, r1, c1;
  asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(c0)::"memory");
  do {
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1), "=s"(c1)::"memory");
  } while (r1 - r0 < ticks);
  if (threadIdx.x == 0) { out[0] = r1 - r0; out[1] = c1 - c0; }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (unsigned long long t : {10000ull, 100000ull, 1000000ull}) {
    spin<<<1, 64>>>(t, d); hipDeviceSynchronize();
    hipEventRecord(e0); spin<<<1, 64>>>(t, d); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("ticks %llu: event %.1f us -> realtime %.2f MHz; memtime %llu -> %.3f GHz\n", h[0], ms * 1e3, h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e6));
  }
  return 0;
}
