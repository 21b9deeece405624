// Does s_waitcnt vmcnt(N) retire buffer loads strictly in issue order when an older load misses to HBM and a younger
// one hits L2/L1 (different buffer descriptors)?  Prints how often the older load's data was NOT there after vmcnt(1).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
__global__ void probe(const uint32_t* cold, const uint32_t* hot, uint32_t* bad, int iters, uint32_t cold_bytes) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const i32x4 rc = make_rsrc(cold, cold_bytes), rh = make_rsrc(hot, 4096);
  uint32_t nbad = 0;
  for (int it = 0; it < iters; ++it) {
    u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    // cold: a different 1 KiB chunk far apart each time (value at every dword = its own dword index + 1)
    const uint32_t off = (uint32_t)(((uint64_t)(wave * 7919u + it * 104729u) * 4096u) % (cold_bytes - 4096u)) & ~15u;
    int voff = lane * 16, soff = __builtin_amdgcn_readfirstlane((int)off);
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(a) : "v"(voff), "s"(rc), "s"(soff) : "memory");
    int zero = 0;
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(b) : "v"(voff), "s"(rh), "s"(zero) : "memory");
    asm volatile("s_waitcnt vmcnt(1)" : "+v"(a), "+v"(b)::"memory");
    const uint32_t expect = (off + lane * 16) / 4 + 1;
    if (a[0] != expect) nbad++;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b)::"memory");
    if (a[0] != expect) nbad += 1000000;  // would mean wrong addressing, not ordering
  }
  if (nbad) atomicAdd(bad, nbad);
}
int main() {
  const uint32_t cold_bytes = 1u << 30;
  uint32_t *cold, *hot, *bad;
  hipMalloc(&cold, cold_bytes); hipMalloc(&hot, 4096); hipMalloc(&bad, 4);
  std::vector<uint32_t> h(cold_bytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i + 1;
  hipMemcpy(cold, h.data(), cold_bytes, hipMemcpyHostToDevice);
  hipMemset(hot, 0, 4096); hipMemset(bad, 0, 4);
  probe<<<1024, 256>>>(cold, hot, bad, 200, cold_bytes);
  hipDeviceSynchronize();
  uint32_t nb = 0; hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
  printf("older load missing after vmcnt(1): %u of %u (values >= 1000000 mean an addressing error)\n", nb, 1024u * 4 * 200);
  return 0;
}
