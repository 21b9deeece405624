// One-shot all-reduce over the xGMI mesh for small decode messages (<= 256 KiB at 6-8 GPUs): replaces the reference's
// CUDA-IPC custom all-reduce (csrc/custom_all_reduce.cuh:179-255 one-stage kernel, thresholds :442-450, csrc/custom_all_reduce.cu
// bindings, `_C_custom_ar`). MI355X-native design:
//   * an MI355X node is a full xGMI mesh (7 links per GPU): every rank reads the message of all peers directly over its own
//     link to each of them - 1/7 of the traffic per link, no ring, no intermediate copy - and sums in a fixed rank order
//     (bitwise identical results on every rank, fp32 accumulation);
//   * the two barriers are EPOCH flags (a per-block counter that only grows, no reset stores) written with system-scope
//     atomics into every peer's signal block and polled with system-scope loads; peer payload loads carry sc0 sc1 so that
//     they are served coherently; every spin is bounded and reports through an error word instead of hanging the GPU;
//   * buffers are exchanged as IPC handles by the host side (neuralmagic_vllm_amd/distributed/custom_all_reduce.py).
// Larger messages stay on RCCL (nmx_custom_ar_should says which).  NOT yet measured on a multi-GPU node: the Python side
// keeps it behind NMX_CUSTOM_AR=1.
#include <string.h>

#include <map>

#include "nmx_common.h"

namespace {

constexpr int kMaxRanks = 8;
constexpr int kMaxBlocks = 64;

struct alignas(128) Signal {
  uint32_t start[kMaxBlocks][kMaxRanks];  // start[b][r]: rank r has entered call number `epoch` (its payload is readable)
  uint32_t end[kMaxBlocks][kMaxRanks];    // end[b][r]: rank r has finished reading everyone's payload
  uint32_t epoch[kMaxBlocks];             // private to the owning rank: calls issued so far, per block
  uint32_t error;                         // != 0: a bounded spin gave up (peer missing / not launched)
};

struct PeerPtrs { void* p[kMaxRanks]; };

__device__ __forceinline__ void sys_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ uint32_t sys_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }

// 16 bytes from a peer's buffer, system-coherent (never served from this GPU's non-coherent caches)
__device__ __forceinline__ u32x4 peer_load16(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}

template <int NG>
__device__ __forceinline__ bool mesh_barrier(uint32_t (Signal::*flags)[kMaxBlocks][kMaxRanks], const PeerPtrs& sig, Signal* self, int rank,
                                             uint32_t epoch, uint32_t spin_limit) {
  bool ok = true;
  if (threadIdx.x < NG) {
    Signal* peer = reinterpret_cast<Signal*>(sig.p[threadIdx.x]);
    sys_store(&(peer->*flags)[blockIdx.x][rank], epoch);                    // one peer store per link
    uint32_t spins = 0;
    while ((int32_t)(sys_load(&(self->*flags)[blockIdx.x][threadIdx.x]) - epoch) < 0) {
      if (++spins > spin_limit) { ok = false; self->error = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;
}

template <typename T> struct Acc8 {
  float f[8];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = 0.f;
  }
};

template <typename T, int NG>
__global__ __launch_bounds__(512) void xgmi_all_reduce_1stage(PeerPtrs data, PeerPtrs sig, int rank, void* out, int64_t n16,
                                                             uint32_t spin_limit) {
  Signal* self = reinterpret_cast<Signal*>(sig.p[rank]);
  __shared__ uint32_t s_epoch;
  if (threadIdx.x == 0) {
    s_epoch = self->epoch[blockIdx.x] + 1;
    self->epoch[blockIdx.x] = s_epoch;
  }
  __syncthreads();
  const uint32_t epoch = s_epoch;
  mesh_barrier<NG>(&Signal::start, sig, self, rank, epoch, spin_limit);
  constexpr int EPV = 16 / sizeof(T);  // elements per 16-byte packet
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
    float acc[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
#pragma unroll
    for (int r = 0; r < NG; ++r) {  // fixed order: every rank computes the same bits
      union { u32x4 u; T e[EPV]; } v;
      v.u = peer_load16(reinterpret_cast<const char*>(data.p[r]) + i * 16);
#pragma unroll
      for (int e = 0; e < EPV; ++e) acc[e] += Scalar<T>::to_f32(v.e[e]);
    }
    union { u32x4 u; T e[EPV]; } o;
#pragma unroll
    for (int e = 0; e < EPV; ++e) o.e[e] = Scalar<T>::from_f32(acc[e]);
    reinterpret_cast<u32x4*>(out)[i] = o.u;
  }
  __syncthreads();
  // nobody may overwrite its payload (the next kernel on its stream) before every peer has finished reading it
  mesh_barrier<NG>(&Signal::end, sig, self, rank, epoch, spin_limit);
}

struct CustomAr {
  int rank, world;
  PeerPtrs signals;
  std::map<const void*, PeerPtrs> buffers;  // own registered pointer -> the same buffer of every rank
};

template <typename T>
int launch_ar(CustomAr* fa, const PeerPtrs& data, void* out, int64_t n16, hipStream_t stream) {
  const int threads = 512;
  const int blocks = (int)std::min<int64_t>(36, std::max<int64_t>(1, (n16 + threads - 1) / threads));
  const uint32_t spin_limit = 1u << 24;  // ~ seconds: a missing peer ends in an error word, not a hung GPU
  switch (fa->world) {
    case 2: xgmi_all_reduce_1stage<T, 2><<<blocks, threads, 0, stream>>>(data, fa->signals, fa->rank, out, n16, spin_limit); break;
    case 4: xgmi_all_reduce_1stage<T, 4><<<blocks, threads, 0, stream>>>(data, fa->signals, fa->rank, out, n16, spin_limit); break;
    case 6: xgmi_all_reduce_1stage<T, 6><<<blocks, threads, 0, stream>>>(data, fa->signals, fa->rank, out, n16, spin_limit); break;
    case 8: xgmi_all_reduce_1stage<T, 8><<<blocks, threads, 0, stream>>>(data, fa->signals, fa->rank, out, n16, spin_limit); break;
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "custom allreduce only supports num gpus in (2,4,6,8), got %d", fa->world);
  }
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

}  // namespace

extern "C" int64_t nmx_custom_ar_meta_size(void) { return (int64_t)sizeof(Signal); }

// custom_all_reduce.cu:init_custom_ar - signal_ptrs[r] = rank r's (zero-filled) signal block as mapped in THIS process
extern "C" int nmx_custom_ar_init(void* const* signal_ptrs, int rank, int world_size, void** fa_out) {
  NMX_CHECK(world_size == 2 || world_size == 4 || world_size == 6 || world_size == 8, NMX_ERR_UNSUPPORTED,
            "custom allreduce only supports num gpus in (2,4,6,8), got %d", world_size);
  NMX_CHECK(rank >= 0 && rank < world_size && fa_out != nullptr && signal_ptrs != nullptr, NMX_ERR_INVALID_ARG, "invalid rank passed in");
  CustomAr* fa = new CustomAr();
  fa->rank = rank;
  fa->world = world_size;
  for (int r = 0; r < kMaxRanks; ++r) fa->signals.p[r] = r < world_size ? signal_ptrs[r] : nullptr;
  *fa_out = fa;
  return NMX_OK;
}

// custom_all_reduce.cu:register_buffer - peer_ptrs[r] = rank r's copy of the buffer (own pointer at [rank])
extern "C" int nmx_custom_ar_register_buffer(void* fa_, void* const* peer_ptrs) {
  CustomAr* fa = reinterpret_cast<CustomAr*>(fa_);
  NMX_CHECK(fa != nullptr && peer_ptrs != nullptr, NMX_ERR_INVALID_ARG, "null custom all-reduce handle");
  PeerPtrs pp;
  for (int r = 0; r < kMaxRanks; ++r) pp.p[r] = r < fa->world ? peer_ptrs[r] : nullptr;
  fa->buffers[peer_ptrs[fa->rank]] = pp;
  return NMX_OK;
}

// custom_all_reduce.cu:should_custom_ar restricted to the sizes the one-stage kernel serves (custom_all_reduce.cuh:442-450):
// world 2: up to max_size; fully connected: < 512 KiB at <= 4 ranks, < 256 KiB at 6 / 8 ranks; 16-byte multiples only
extern "C" int nmx_custom_ar_should(int64_t bytes, int64_t max_size, int world_size, int full_xgmi) {
  if (bytes <= 0 || bytes % 16 != 0 || bytes > max_size) return 0;
  if (world_size == 2) return 1;
  if (!full_xgmi) return 0;
  if (world_size <= 4) return bytes < 512 * 1024;
  if (world_size <= 8) return bytes < 256 * 1024;
  return 0;
}

extern "C" int nmx_custom_ar_all_reduce(void* fa_, const void* inp, void* out, int64_t numel, int dtype, nmx_stream_t stream) {
  CustomAr* fa = reinterpret_cast<CustomAr*>(fa_);
  NMX_CHECK(fa != nullptr, NMX_ERR_INVALID_ARG, "null custom all-reduce handle");
  auto it = fa->buffers.find(inp);
  NMX_CHECK(it != fa->buffers.end(), NMX_ERR_INVALID_ARG, "buffer address %p is not registered!", inp);
  const int64_t bytes = numel * nmx_dtype_size(dtype);
  NMX_CHECK(bytes % 16 == 0 && ((uintptr_t)out % 16 == 0), NMX_ERR_INVALID_ARG, "custom allreduce currently requires input length to be multiple of 16 bytes");
  if (bytes == 0) return NMX_OK;
  switch (dtype) {
    case NMX_F32: return launch_ar<float>(fa, it->second, out, bytes / 16, (hipStream_t)stream);
    case NMX_F16: return launch_ar<f16>(fa, it->second, out, bytes / 16, (hipStream_t)stream);
    case NMX_BF16: return launch_ar<bf16>(fa, it->second, out, bytes / 16, (hipStream_t)stream);
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "custom allreduce only supports float32, float16 and bfloat16");
  }
}

extern "C" int nmx_custom_ar_dispose(void* fa_) {
  delete reinterpret_cast<CustomAr*>(fa_);
  return NMX_OK;
}

// IPC plumbing for the host side (hipIpcMemHandle_t is 64 bytes)
extern "C" int nmx_ipc_get_mem_handle(const void* ptr, void* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  NMX_HIP(hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), const_cast<void*>(ptr)));
  return NMX_OK;
}
extern "C" int nmx_ipc_open_mem_handle(const void* handle64, void** ptr) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  NMX_HIP(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
  return NMX_OK;
}
extern "C" int nmx_ipc_close_mem_handle(void* ptr) {
  NMX_HIP(hipIpcCloseMemHandle(ptr));
  return NMX_OK;
}
