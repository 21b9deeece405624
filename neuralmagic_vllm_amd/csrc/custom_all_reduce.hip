// All-reduce over the xGMI mesh for decode-sized messages: replaces the reference's CUDA-IPC custom all-reduce
// (csrc/custom_all_reduce.cuh: one-stage kernel :179-203, two-stage kernel :204-255, dispatch thresholds :442-450;
// csrc/custom_all_reduce.cu bindings, `_C_custom_ar`). MI355X-native design:
//   * an MI355X node is a full xGMI mesh (7 links per GPU). ONE-STAGE (small messages): every rank reads the message of all
//     peers directly over its own link to each of them - 1/7 of the traffic per link, no ring, no intermediate copy - and
//     sums in a fixed rank order (bitwise identical results on every rank, fp32 accumulation). TWO-STAGE (larger messages,
//     round 3): reduce-scatter - rank r sums slice r of every peer's message (same fixed order) into its own peer-visible
//     scratch - then all-gather - every rank copies the N reduced slices out of the peers' scratch: 2 (N - 1) / N of the
//     message per rank over the wire instead of (N - 1), again 1/7 of it per link;
//   * the barriers are EPOCH flags (a per-block counter that only grows, no reset stores) written with system-scope atomics
//     into every peer's signal block and polled with system-scope loads; peer payload loads carry sc0 sc1 (served from the
//     owner's memory, never from this GPU's caches), scratch stores are write-through (sc0 sc1) and drained before the flag;
//     every spin is bounded: a barrier that times out sets the error word, the launch then writes NOTHING further (no sum
//     of whatever payload happens to be there) and skips its remaining barriers; the host reads the word
//     (nmx_custom_ar_check) after a sync in eager mode / after a graph replay;
//   * buffers are exchanged as IPC handles by the host side (neuralmagic_vllm_amd/distributed/custom_all_reduce.py); the
//     signal + scratch block comes from nmx_custom_ar_alloc_meta (uncached, fine-grained device memory: flags polled here
//     are written by peers over xGMI).
// Larger messages stay on RCCL (nmx_custom_ar_should says which).  NOT yet measured on a multi-GPU node: the Python side
// keeps it behind NMX_CUSTOM_AR=1. On one GPU nmx_custom_ar_loopback runs all N "ranks" as slices of ONE grid (co-resident
// by construction), which exercises the barriers, both schedules and the cross-XCD visibility of flags and scratch.
#include <string.h>

#include <map>

#include "nmx_common.h"

namespace {

constexpr int kMaxRanks = 8;
constexpr int kMaxBlocks = 64;

struct alignas(128) Signal {
  uint32_t start[kMaxBlocks][kMaxRanks];  // start[b][r]: rank r has entered call number `epoch` (its payload is readable)
  uint32_t mid[kMaxBlocks][kMaxRanks];    // two-stage: rank r's reduced slice is in its scratch
  uint32_t end[kMaxBlocks][kMaxRanks];    // one-stage: rank r has finished reading everyone's payload
  uint32_t epoch[kMaxBlocks];             // private to the owning rank: calls issued so far, per block
  uint32_t error;                         // != 0: a bounded spin gave up (peer missing / not launched)
};
// the two-stage scratch of a rank starts right behind its Signal (same IPC allocation, like the reference's meta buffer)
__device__ __host__ __forceinline__ char* scratch_of(void* sig) { return reinterpret_cast<char*>(sig) + sizeof(Signal); }

struct PeerPtrs { void* p[kMaxRanks]; };

__device__ __forceinline__ void sys_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ uint32_t sys_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }

// 16 bytes from a peer's buffer, system-coherent (never served from this GPU's non-coherent caches)
__device__ __forceinline__ u32x4 peer_load16(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
// 16 bytes into this rank's peer-visible scratch, write-through to memory
__device__ __forceinline__ void scratch_store16(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// Every rank tells every peer "I am at `epoch`" and waits until every peer has said so. Returns false (to ALL threads of the
// block) when a spin ran out; the caller then stops touching payloads.
template <int NG>
__device__ __forceinline__ bool mesh_barrier(uint32_t (Signal::*flags)[kMaxBlocks][kMaxRanks], const PeerPtrs& sig, Signal* self, int rank,
                                             int block, uint32_t epoch, uint32_t spin_limit, uint32_t* s_ok) {
  if (threadIdx.x == 0) *s_ok = 1;
  __syncthreads();
  if (threadIdx.x < NG) {
    Signal* peer = reinterpret_cast<Signal*>(sig.p[threadIdx.x]);
    sys_store(&(peer->*flags)[block][rank], epoch);                    // one peer store per link
    uint32_t spins = 0;
    while ((int32_t)(sys_load(&(self->*flags)[block][threadIdx.x]) - epoch) < 0) {
      if (++spins > spin_limit) {
        __hip_atomic_store(&self->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        *s_ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return *s_ok != 0;
}

// sum of packet i over the NG payloads, fixed order r = 0, 1, ...: every rank computes the same bits
template <typename T, int NG>
__device__ __forceinline__ u32x4 reduce_packet(const PeerPtrs& data, int64_t i) {
  constexpr int EPV = 16 / sizeof(T);  // elements per 16-byte packet
  float acc[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
#pragma unroll
  for (int r = 0; r < NG; ++r) {
    union { u32x4 u; T e[EPV]; } v;
    v.u = peer_load16(reinterpret_cast<const char*>(data.p[r]) + i * 16);
#pragma unroll
    for (int e = 0; e < EPV; ++e) acc[e] += Scalar<T>::to_f32(v.e[e]);
  }
  union { u32x4 u; T e[EPV]; } o;
#pragma unroll
  for (int e = 0; e < EPV; ++e) o.e[e] = Scalar<T>::from_f32(acc[e]);
  return o.u;
}

// One rank's share of an all-reduce call: block `block` of `nblocks`, as rank `rank`. Called by the real kernel (one rank per
// GPU) and by the loop-back kernel (ranks = slices of one grid).
template <typename T, int NG, bool TWO_STAGE>
__device__ __forceinline__ void all_reduce_body(const PeerPtrs& data, const PeerPtrs& sig, int rank, void* out, int64_t n16,
                                                uint32_t spin_limit, int block, int nblocks) {
  Signal* self = reinterpret_cast<Signal*>(sig.p[rank]);
  __shared__ uint32_t s_epoch, s_ok;
  if (threadIdx.x == 0) {
    s_epoch = self->epoch[block] + 1;
    self->epoch[block] = s_epoch;
  }
  __syncthreads();
  const uint32_t epoch = s_epoch;
  // a timed-out start barrier: some peer's payload may not be there - write nothing, skip the remaining barriers (their
  // flags of this epoch stay unset; the epochs only grow, so a later healthy call is not confused by them)
  if (!mesh_barrier<NG>(&Signal::start, sig, self, rank, block, epoch, spin_limit, &s_ok)) return;
  const int64_t tid = (int64_t)block * blockDim.x + threadIdx.x, stride = (int64_t)nblocks * blockDim.x;
  if constexpr (!TWO_STAGE) {
    for (int64_t i = tid; i < n16; i += stride) reinterpret_cast<u32x4*>(out)[i] = reduce_packet<T, NG>(data, i);
    __syncthreads();
    // nobody may overwrite its payload (the next kernel on its stream) before every peer has finished reading it
    mesh_barrier<NG>(&Signal::end, sig, self, rank, block, epoch, spin_limit, &s_ok);
  } else {
    // slice r = packets [r * part, (r + 1) * part), the last rank also takes the remainder (custom_all_reduce.cuh:214-221)
    const int64_t part = n16 / NG;
    const int64_t lo = rank * part, hi = rank == NG - 1 ? n16 : lo + part;
    char* my_tmp = scratch_of(self);
    for (int64_t i = lo + tid; i < hi; i += stride) scratch_store16(my_tmp + (i - lo) * 16, reduce_packet<T, NG>(data, i));
    // publish: every storing wave drains its write-through stores, the workgroup meets, then the (release) flag stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // after this barrier every peer has finished READING the payloads too (its stage 1 is done): no end barrier needed,
    // and the scratch is protected by the next call's start barrier (a peer enters it only after its own gather)
    if (!mesh_barrier<NG>(&Signal::mid, sig, self, rank, block, epoch, spin_limit, &s_ok)) return;
#pragma unroll
    for (int r = 0; r < NG; ++r) {
      const int64_t rlo = r * part, rhi = r == NG - 1 ? n16 : rlo + part;
      const char* tmp = scratch_of(sig.p[r]);
      for (int64_t i = rlo + tid; i < rhi; i += stride) reinterpret_cast<u32x4*>(out)[i] = peer_load16(tmp + (i - rlo) * 16);
    }
  }
}

template <typename T, int NG, bool TWO_STAGE>
__global__ __launch_bounds__(512) void xgmi_all_reduce_kernel(PeerPtrs data, PeerPtrs sig, int rank, void* out, int64_t n16,
                                                             uint32_t spin_limit) {
  all_reduce_body<T, NG, TWO_STAGE>(data, sig, rank, out, n16, spin_limit, blockIdx.x, gridDim.x);
}

// ONE grid, blockIdx.y = rank: the loop-back arrangement of the one-GPU test. All NG * gridDim.x blocks are resident together
// (the host checks the bound), so the barriers complete without relying on concurrent launches.
template <typename T, int NG, bool TWO_STAGE>
__global__ __launch_bounds__(512) void xgmi_all_reduce_loopback_kernel(PeerPtrs data, PeerPtrs sig, PeerPtrs outs, int64_t n16,
                                                                      uint32_t spin_limit) {
  all_reduce_body<T, NG, TWO_STAGE>(data, sig, blockIdx.y, outs.p[blockIdx.y], n16, spin_limit, blockIdx.x, gridDim.x);
}

struct CustomAr {
  int rank, world;
  PeerPtrs signals;
  std::map<const void*, PeerPtrs> buffers;  // own registered pointer -> the same buffer of every rank
  uint32_t spin_limit = 1u << 24;           // ~ seconds: a missing peer ends in an error word, not a hung GPU
  int64_t scratch_bytes = 0;                // bytes behind every rank's Signal (two-stage needs ceil(bytes / world) + 16 * world)
};

constexpr int kThreads = 512;
inline int ar_blocks(int64_t n16) { return (int)std::min<int64_t>(36, std::max<int64_t>(1, (n16 + kThreads - 1) / kThreads)); }

// custom_all_reduce.cuh:442-450: two ranks always one-stage; full mesh: one-stage below 512 KiB (<= 4 ranks) / 256 KiB
// (<= 8 ranks), two-stage above
inline bool use_two_stage(int world, int64_t bytes) {
  if (world == 2) return false;
  if (world <= 4) return bytes >= 512 * 1024;
  return bytes >= 256 * 1024;
}

template <typename T, int NG>
int launch_ar_ng(CustomAr* fa, const PeerPtrs& data, void* out, int64_t n16, bool two_stage, hipStream_t stream) {
  const int blocks = ar_blocks(n16);
  if (two_stage) xgmi_all_reduce_kernel<T, NG, true><<<blocks, kThreads, 0, stream>>>(data, fa->signals, fa->rank, out, n16, fa->spin_limit);
  else xgmi_all_reduce_kernel<T, NG, false><<<blocks, kThreads, 0, stream>>>(data, fa->signals, fa->rank, out, n16, fa->spin_limit);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

template <typename T>
int launch_ar(CustomAr* fa, const PeerPtrs& data, void* out, int64_t n16, bool two_stage, hipStream_t stream) {
  switch (fa->world) {
    case 2: return launch_ar_ng<T, 2>(fa, data, out, n16, two_stage, stream);
    case 4: return launch_ar_ng<T, 4>(fa, data, out, n16, two_stage, stream);
    case 6: return launch_ar_ng<T, 6>(fa, data, out, n16, two_stage, stream);
    case 8: return launch_ar_ng<T, 8>(fa, data, out, n16, two_stage, stream);
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "custom allreduce only supports num gpus in (2,4,6,8), got %d", fa->world);
  }
}

template <typename T, int NG>
int launch_loopback_ng(const PeerPtrs& data, const PeerPtrs& sig, const PeerPtrs& outs, int64_t n16, bool two_stage, uint32_t spin_limit,
                       hipStream_t stream) {
  dim3 grid(ar_blocks(n16), NG);
  if (two_stage) xgmi_all_reduce_loopback_kernel<T, NG, true><<<grid, kThreads, 0, stream>>>(data, sig, outs, n16, spin_limit);
  else xgmi_all_reduce_loopback_kernel<T, NG, false><<<grid, kThreads, 0, stream>>>(data, sig, outs, n16, spin_limit);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

template <typename T>
int launch_loopback(int world, const PeerPtrs& data, const PeerPtrs& sig, const PeerPtrs& outs, int64_t n16, bool two_stage,
                    uint32_t spin_limit, hipStream_t stream) {
  switch (world) {
    case 2: return launch_loopback_ng<T, 2>(data, sig, outs, n16, two_stage, spin_limit, stream);
    case 4: return launch_loopback_ng<T, 4>(data, sig, outs, n16, two_stage, spin_limit, stream);
    case 6: return launch_loopback_ng<T, 6>(data, sig, outs, n16, two_stage, spin_limit, stream);
    case 8: return launch_loopback_ng<T, 8>(data, sig, outs, n16, two_stage, spin_limit, stream);
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "custom allreduce only supports num gpus in (2,4,6,8), got %d", world);
  }
}

}  // namespace

// bytes of the signal block alone; the meta allocation of a rank is this + its two-stage scratch
extern "C" int64_t nmx_custom_ar_meta_size(void) { return (int64_t)sizeof(Signal); }

// The signal + scratch block: zero-filled, UNCACHED fine-grained device memory (its flags are polled by this GPU while peers
// write them over xGMI, and its scratch is read by peers inside a running kernel - an ordinary cached allocation promises
// visibility at kernel boundaries only). Released with nmx_custom_ar_free_meta.
extern "C" int nmx_custom_ar_alloc_meta(int64_t bytes, void** ptr) {
  NMX_CHECK(ptr != nullptr && bytes >= (int64_t)sizeof(Signal), NMX_ERR_INVALID_ARG, "custom_ar_alloc_meta: at least meta_size bytes");
  NMX_HIP(hipExtMallocWithFlags(ptr, (size_t)bytes, hipDeviceMallocUncached));
  NMX_HIP(hipMemset(*ptr, 0, (size_t)bytes));
  NMX_HIP(hipDeviceSynchronize());
  return NMX_OK;
}
extern "C" int nmx_custom_ar_free_meta(void* ptr) {
  if (ptr != nullptr) NMX_HIP(hipFree(ptr));
  return NMX_OK;
}

// custom_all_reduce.cu:init_custom_ar - signal_ptrs[r] = rank r's (zero-filled) signal block as mapped in THIS process;
// scratch_bytes = bytes behind each Signal usable by the two-stage schedule (0: one-stage only)
extern "C" int nmx_custom_ar_init(void* const* signal_ptrs, int rank, int world_size, int64_t scratch_bytes, void** fa_out) {
  NMX_CHECK(world_size == 2 || world_size == 4 || world_size == 6 || world_size == 8, NMX_ERR_UNSUPPORTED,
            "custom allreduce only supports num gpus in (2,4,6,8), got %d", world_size);
  NMX_CHECK(rank >= 0 && rank < world_size && fa_out != nullptr && signal_ptrs != nullptr, NMX_ERR_INVALID_ARG, "invalid rank passed in");
  CustomAr* fa = new CustomAr();
  fa->rank = rank;
  fa->world = world_size;
  fa->scratch_bytes = scratch_bytes > 0 ? scratch_bytes : 0;
  for (int r = 0; r < kMaxRanks; ++r) fa->signals.p[r] = r < world_size ? signal_ptrs[r] : nullptr;
  *fa_out = fa;
  return NMX_OK;
}

// bound of every barrier spin (iterations of a ~64-cycle sleep + one remote-visible load); tests use a small one
extern "C" int nmx_custom_ar_set_spin_limit(void* fa_, uint32_t spin_limit) {
  CustomAr* fa = reinterpret_cast<CustomAr*>(fa_);
  NMX_CHECK(fa != nullptr && spin_limit > 0, NMX_ERR_INVALID_ARG, "null custom all-reduce handle / zero spin limit");
  fa->spin_limit = spin_limit;
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

// custom_all_reduce.cu:should_custom_ar (custom_all_reduce.py:116-131 on the host side): 16-byte multiples up to max_size;
// two ranks always, more ranks only on the full mesh. Which schedule runs: nmx_custom_ar_stages.
extern "C" int nmx_custom_ar_should(int64_t bytes, int64_t max_size, int world_size, int full_xgmi) {
  if (bytes <= 0 || bytes % 16 != 0 || bytes > max_size) return 0;
  if (world_size == 2) return 1;
  if (!full_xgmi) return 0;
  return world_size <= 8 ? 1 : 0;
}

// 1 = one-stage, 2 = two-stage (custom_all_reduce.cuh:442-450)
extern "C" int nmx_custom_ar_stages(int64_t bytes, int world_size) { return use_two_stage(world_size, bytes) ? 2 : 1; }

// scratch a rank needs behind its Signal for a two-stage call on `bytes`: its slice (+ the remainder the last rank takes)
extern "C" int64_t nmx_custom_ar_scratch_bytes(int64_t bytes, int world_size) {
  if (world_size < 2) return 0;
  const int64_t n16 = (bytes + 15) / 16;
  return (n16 / world_size + n16 % world_size) * 16;
}

extern "C" int nmx_custom_ar_all_reduce(void* fa_, const void* inp, void* out, int64_t numel, int dtype, nmx_stream_t stream) {
  CustomAr* fa = reinterpret_cast<CustomAr*>(fa_);
  NMX_CHECK(fa != nullptr, NMX_ERR_INVALID_ARG, "null custom all-reduce handle");
  auto it = fa->buffers.find(inp);
  NMX_CHECK(it != fa->buffers.end(), NMX_ERR_INVALID_ARG, "buffer address %p is not registered!", inp);
  const int64_t bytes = numel * nmx_dtype_size(dtype);
  NMX_CHECK(bytes % 16 == 0 && ((uintptr_t)out % 16 == 0), NMX_ERR_INVALID_ARG, "custom allreduce currently requires input length to be multiple of 16 bytes");
  if (bytes == 0) return NMX_OK;
  const bool two = use_two_stage(fa->world, bytes);
  NMX_CHECK(!two || nmx_custom_ar_scratch_bytes(bytes, fa->world) <= fa->scratch_bytes, NMX_ERR_INVALID_ARG,
            "custom allreduce: two-stage schedule needs %lld scratch bytes behind the signal block, %lld registered",
            (long long)nmx_custom_ar_scratch_bytes(bytes, fa->world), (long long)fa->scratch_bytes);
  switch (dtype) {
    case NMX_F32: return launch_ar<float>(fa, it->second, out, bytes / 16, two, (hipStream_t)stream);
    case NMX_F16: return launch_ar<f16>(fa, it->second, out, bytes / 16, two, (hipStream_t)stream);
    case NMX_BF16: return launch_ar<bf16>(fa, it->second, out, bytes / 16, two, (hipStream_t)stream);
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "custom allreduce only supports float32, float16 and bfloat16");
  }
}

// The error word of this rank's signal block, read AFTER the stream has been synchronised (eager mode: after the call's
// sync; graphs: after a replay). Non-zero: a barrier of some call since the last check timed out and that call's output is
// not a sum. clear != 0 resets the word.
extern "C" int nmx_custom_ar_check(void* fa_, int clear, int* error_out) {
  CustomAr* fa = reinterpret_cast<CustomAr*>(fa_);
  NMX_CHECK(fa != nullptr && error_out != nullptr, NMX_ERR_INVALID_ARG, "null custom all-reduce handle");
  Signal* self = reinterpret_cast<Signal*>(fa->signals.p[fa->rank]);
  uint32_t e = 0;
  NMX_HIP(hipMemcpy(&e, &self->error, sizeof(e), hipMemcpyDeviceToHost));
  if (e != 0 && clear) NMX_HIP(hipMemset(&self->error, 0, sizeof(e)));
  *error_out = (int)e;
  return NMX_OK;
}

// One-GPU loop-back (tests): all `world` ranks of one call as ONE launch, rank = blockIdx.y. signal_ptrs / data_ptrs / out_ptrs
// [world]; stages 1 / 2 forces the schedule, 0 = the dispatch rule.
extern "C" int nmx_custom_ar_loopback(void* const* signal_ptrs, void* const* data_ptrs, void* const* out_ptrs, int world_size,
                                      int64_t numel, int dtype, int stages, uint32_t spin_limit, nmx_stream_t stream) {
  NMX_CHECK(world_size == 2 || world_size == 4 || world_size == 6 || world_size == 8, NMX_ERR_UNSUPPORTED,
            "custom allreduce only supports num gpus in (2,4,6,8), got %d", world_size);
  NMX_CHECK(signal_ptrs && data_ptrs && out_ptrs && spin_limit > 0, NMX_ERR_INVALID_ARG, "custom_ar_loopback: null argument");
  const int64_t bytes = numel * nmx_dtype_size(dtype);
  NMX_CHECK(bytes > 0 && bytes % 16 == 0, NMX_ERR_INVALID_ARG, "custom allreduce currently requires input length to be multiple of 16 bytes");
  PeerPtrs sig{}, data{}, outs{};
  for (int r = 0; r < world_size; ++r) { sig.p[r] = signal_ptrs[r]; data.p[r] = data_ptrs[r]; outs.p[r] = out_ptrs[r]; }
  const bool two = stages == 0 ? use_two_stage(world_size, bytes) : stages == 2;
  // residency: world * blocks workgroups of 512 threads, 4 per CU
  int cus = 0;
  NMX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  NMX_CHECK(world_size * ar_blocks(bytes / 16) <= 2 * cus, NMX_ERR_UNSUPPORTED, "custom_ar_loopback: grid would not be co-resident");
  switch (dtype) {
    case NMX_F32: return launch_loopback<float>(world_size, data, sig, outs, bytes / 16, two, spin_limit, (hipStream_t)stream);
    case NMX_F16: return launch_loopback<f16>(world_size, data, sig, outs, bytes / 16, two, spin_limit, (hipStream_t)stream);
    case NMX_BF16: return launch_loopback<bf16>(world_size, data, sig, outs, bytes / 16, two, spin_limit, (hipStream_t)stream);
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
