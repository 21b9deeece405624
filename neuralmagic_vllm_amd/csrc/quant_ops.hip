// Activation quantisation (fp8 per-tensor static / dynamic, int8 per-tensor static / per-token dynamic) and the
// W8A8 scaled GEMM (fp8 x fp8 and int8 x int8 on MFMA) for gfx950.
// Replaces csrc/quantization/fp8/common.cu, csrc/quantization/compressed_tensors/int8_quant_kernels.cu and the
// semantics of csrc/quantization/cutlass_w8a8/scaled_mm_entry.cu (CUTLASS itself is not ported: the contraction is
// a hand-written MFMA kernel; fp8 is OCP e4m3fn, the gfx950-native format).
#include <stdlib.h>

#include "nmx_common.h"

typedef int i32x8 __attribute__((ext_vector_type(8)));

// timing ablations of scaled_mm_tile_kernel (results are WRONG when set; never defined in the product build):
// bit 0 skip global loads, 1 skip MFMAs, 2 skip LDS writes, 3 skip LDS fragment reads, 4 skip barriers
#ifndef NMX_TABLATE
#define NMX_TABLATE 0
#endif

namespace {

// ---- fp8 per-tensor quant (fp8/common.cu:24-125) ----------------------------------------------------------------
// dynamic: scale = max|x| / 448 over the whole tensor. The reference reduces with atomicMax into a zero-initialised
// scale; here pass 1 leaves one maximum per workgroup in a small scratch array and pass 2 lets every workgroup take the
// maximum of those (<= 64 L2-resident floats) before it quantises - no atomics on one address (256 of them took the
// first pass to 7.5 us), no need for the caller to zero the scale (a fill launch), same value bit for bit.
constexpr int kAbsmaxBlocks = 64;

template <typename T>
__device__ __forceinline__ float absmax8(const T* p) {  // 8 consecutive elements (16 B for 2-byte types)
  float m = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(Scalar<T>::to_f32(p[e])));
  return m;
}

template <typename T>
__global__ __launch_bounds__(1024) void fp8_absmax_kernel(float* __restrict__ partial, const T* __restrict__ x, int64_t n,
                                                          int64_t n8) {
  __shared__ float smem[16];
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    struct alignas(sizeof(T) * 8) V { T e[8]; };
    const V v = reinterpret_cast<const V*>(x)[i];
    m = fmaxf(m, absmax8<T>(v.e));
  }
  for (int64_t i = n8 * 8 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    m = fmaxf(m, fabsf(Scalar<T>::to_f32(x[i])));
  m = wave_reduce_max(m);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) smem[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) t = fmaxf(t, smem[w]);
    partial[blockIdx.x] = t;
  }
}

// DYNAMIC: scale = max(partial[0 .. nparts)) / 448, written to *scale by workgroup 0
template <typename T, bool DYNAMIC>
__global__ __launch_bounds__(1024) void fp8_quant_kernel(uint8_t* __restrict__ out, const T* __restrict__ x, float* __restrict__ scale,
                                                         const float* __restrict__ partial, int nparts, int64_t n,
                                                         int64_t n8) {
  float sc;
  if constexpr (DYNAMIC) {
    // <= 64 partials from fp8_absmax_kernel, one per token from a producer kernel, or one per (kv head, sequence) from the
    // attention kernels (2048 at batch 256): the whole workgroup reads them once (L2-resident), waves combine through LDS
    __shared__ float wmax[16];
    float m = 0.f;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) m = fmaxf(m, partial[i]);
    m = wave_reduce_max(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    m = 0.f;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) m = fmaxf(m, wmax[w]);
    sc = m / 448.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) *scale = sc;
  } else {
    sc = *scale;
  }
  const float inv = 1.0f / sc;  // the reference multiplies by the inverted scale (common.cu:91)
  auto q = [&](T xv) -> uint32_t {
    float v = Scalar<T>::to_f32(xv) * inv;
    asm volatile("" : "+v"(v));  // keep the product a separate fp32 rounding step (no fusion into the conversion)
    v = fmaxf(-448.0f, fminf(v, 448.0f));
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false) & 0xffu;
  };
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    struct alignas(sizeof(T) * 8) V { T e[8]; };
    const V v = reinterpret_cast<const V*>(x)[i];
    u32x2 o;
    o[0] = q(v.e[0]) | (q(v.e[1]) << 8) | (q(v.e[2]) << 16) | (q(v.e[3]) << 24);
    o[1] = q(v.e[4]) | (q(v.e[5]) << 8) | (q(v.e[6]) << 16) | (q(v.e[7]) << 24);
    reinterpret_cast<u32x2*>(out)[i] = o;
  }
  for (int64_t i = n8 * 8 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (uint8_t)q(x[i]);
}

// ---- int8 quant (compressed_tensors/int8_quant_kernels.cu:8-71) -------------------------------------------------
__device__ __forceinline__ int8_t f32_to_i8_rn_sat(float x) {
  float r = rintf(x);
  r = fminf(fmaxf(r, -128.f), 127.f);
  return (int8_t)r;
}

template <typename T, bool DYNAMIC>
__global__ void int8_quant_kernel(int8_t* __restrict__ out, const T* __restrict__ x, float* __restrict__ scales, int hidden) {
  __shared__ float smem[17];
  const int64_t row = (int64_t)blockIdx.x * hidden;
  if constexpr (DYNAMIC) {
    float m = 0.f;
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) m = fmaxf(m, fabsf(Scalar<T>::to_f32(x[row + i])));
    m = wave_reduce_max(m);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) smem[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) t = fmaxf(t, smem[w]);
      smem[16] = t;
      scales[blockIdx.x] = t / 127.0f;
    }
    __syncthreads();
    const float ts = 127.0f / smem[16];
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) out[row + i] = f32_to_i8_rn_sat(Scalar<T>::to_f32(x[row + i]) * ts);
  } else {
    const float s = scales[0];
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) out[row + i] = f32_to_i8_rn_sat(Scalar<T>::to_f32(x[row + i]) / s);
  }
}

// ---- W8A8 scaled GEMM -------------------------------------------------------------------------------------------
// out[M,N] = cast(a_scale (.) (A . B) (.) b_scale) (+ bias). A [M,K] row-major, B column-major = Bt [N,K] row-major:
// both operands are K-contiguous, i.e. already in MFMA fragment order. Lane (g, i) of a 16-row tile loads 16
// consecutive k-bytes (k0 + 16 g ..) of its row and feeds two MFMA 16x16x32 (fp8) / one 16x16x64 (int8) k-steps.
// Workgroup: 4 waves = 4 contiguous K-slices of one 64-column x (16 MT)-row tile, reduced through LDS. Every wave
// keeps PF 64-byte k-steps of both operands in flight (a register ring; without it each step paid a full memory round
// trip and the kernel was latency-bound: 43 us for 25 MB at M = 64). K is also split across gridDim.y workgroups when
// the tile count alone leaves CUs idle (decode: N / 64 = 64..96 tiles): raw fp32 / int32 partial slabs in a scratch
// buffer, summed in a fixed order by scaled_mm_reduce_kernel, which then applies the scale / bias epilogue.
// grid (ceil(N / 64), k_splits, ceil(M / (16 MT)))
struct MmParams {
  const uint8_t* a;
  const uint8_t* bt;
  void* out;
  const float* a_scales;
  const float* b_scales;
  const void* bias;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int a_per_row, b_per_col;
  int k_splits;
  void* partial;  // [k_splits][M][N] fp32 (fp8) / int32 (int8)
  int defer_reduce;  // leave the raw slabs for the consumer op (no reduce launch)
};

// four consecutive output columns n .. n + 3 of row m: scales, bias, ONE 8-byte store (n % 4 == 0, ldc % 16 == 0)
template <typename out_t, typename acc_t>
__device__ __forceinline__ void mm_epilogue4(const MmParams& p, acc_t acc, int m, int n) {
  const float sa = p.a_scales[p.a_per_row ? m : 0];
  union { out_t h[4]; u32x2 u; } o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float sb = p.b_scales[p.b_per_col ? n + r : 0];
    float v = sa * (sb * (float)acc[r]);  // tests/kernels/test_cutlass.py:35-47
    // keep the fp32 product a rounding step of its own: hipcc otherwise fuses the last multiply with the conversion
    // (v_fma_mixlo_f16: ONE rounding of the exact product), which differs from the reference's float epilogue +
    // NumericConverter on exact ties - and from the consumer ops that apply the same epilogue to deferred slabs
    asm volatile("" : "+v"(v));
    o.h[r] = Scalar<out_t>::from_f32(v);
  }
  if (p.bias != nullptr) {
    union { out_t h[4]; u32x2 u; } b;
    b.u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const out_t*>(p.bias) + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) o.h[r] = Scalar<out_t>::from_f32(Scalar<out_t>::to_f32(o.h[r]) + Scalar<out_t>::to_f32(b.h[r]));
  }
  *reinterpret_cast<u32x2*>(reinterpret_cast<out_t*>(p.out) + (int64_t)m * p.ldc + n) = o.u;
}

template <typename out_t, bool FP8, int MT, bool TAIL>
__global__ __launch_bounds__(256, 2) void scaled_mm_kernel(const MmParams p) {
  constexpr int NT = 4;  // 16-column tiles per wave (64 columns)
  constexpr int PF = 4;  // k-steps in flight per wave
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.z * 16 * MT;
  const int K = p.K;
  using acc_t = typename std::conditional<FP8, f32x4, i32x4>::type;
  acc_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[mt][t] = acc_t{0, 0, 0, 0};

  // this workgroup's range of 64-byte k-steps; its 4 waves interleave inside it
  const int steps = (K + 63) / 64;
  const int per = (steps + p.k_splits - 1) / p.k_splits;
  const int sb = min((int)blockIdx.y * per, steps), se = min(sb + per, steps);

  // Buffer loads: per-lane byte offset of the row (rows past the matrix get an offset beyond the descriptor and read as
  // zeros), wave-uniform k offset in an SGPR. No value is touched between the load and its MFMA, so nothing forces an
  // early wait and the ring really stays in flight (a select on the loaded value right after the load did exactly that).
  const __amdgpu_buffer_rsrc_t rs_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.bt), 0, (int)((int64_t)p.N * p.ldb), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.a), 0, (int)((int64_t)p.M * p.lda), 0x00020000);
  int b_voff[NT], a_voff[MT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = n0 + 16 * t + li;
    b_voff[t] = n < p.N ? (int)(n * p.ldb + 16 * g) : (int)0xfffffff0u;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + li;
    a_voff[mt] = m < p.M ? (int)(m * p.lda + 16 * g) : (int)0xfffffff0u;
  }
  struct Step { u32x4 bf[NT]; u32x4 af[MT]; };
  auto load = [&](int s, Step& r) {
    const int soff = min(s, steps - 1) * 64;  // past the range: the last step again (never computed on)
#pragma unroll
    for (int t = 0; t < NT; ++t) r.bf[t] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff[t], soff, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) r.af[mt] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff[mt], soff, 0);
    if constexpr (TAIL) {
      // K % 64 != 0: the lanes of the last step that start at or beyond K read the next row; drop them
      const bool kok = s * 64 + 16 * g < K;
#pragma unroll
      for (int t = 0; t < NT; ++t) r.bf[t] = kok ? r.bf[t] : u32x4{0, 0, 0, 0};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) r.af[mt] = kok ? r.af[mt] : u32x4{0, 0, 0, 0};
    } else {
      __builtin_amdgcn_sched_barrier(0);  // same queue order in the prologue and in the loop
    }
  };
  auto compute = [&](const Step& r) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        // weights as the MFMA "A" operand (rows = columns of the output), activations as "B": D[n][m]
        if constexpr (FP8) {
          const long b0 = (long)(((uint64_t)r.bf[t][1] << 32) | r.bf[t][0]), b1 = (long)(((uint64_t)r.bf[t][3] << 32) | r.bf[t][2]);
          const long a0 = (long)(((uint64_t)r.af[mt][1] << 32) | r.af[mt][0]), a1 = (long)(((uint64_t)r.af[mt][3] << 32) | r.af[mt][2]);
          acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b0, a0, acc[mt][t], 0, 0, 0);
          acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b1, a1, acc[mt][t], 0, 0, 0);
        } else {
          acc[mt][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, r.bf[t]), __builtin_bit_cast(i32x4, r.af[mt]),
                                                            acc[mt][t], 0, 0, 0);
        }
      }
  };
  {
    // each wave a CONTIGUOUS quarter of the workgroup's steps: consecutive 64-byte pieces of a row (the two halves of a
    // 128-byte line) are then requested back to back by the same wave
    const int pw = (se - sb + 3) / 4;
    const int ws = min(sb + wave * pw, se), we = min(ws + pw, se);
    // Every workgroup starts its sweep at a different k (rotated by its column-tile index, wrapping inside the wave's
    // slice): rows of B are K bytes apart, so workgroups that all read the same k offset of their rows at the same time
    // concentrate on the few memory channels those addresses share.
    const int len = we - ws;
    const int rot = len > 0 ? (int)((blockIdx.x * 5u) % (unsigned)len) : 0;
    auto step_of = [&](int j) {  // j-th step of this wave's sweep; past the end: a valid step (loaded, never used)
      int t = min(j, len - 1) + rot;
      t = t >= len ? t - len : t;
      return ws + t;
    };
    Step ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) load(step_of(i), ring[i]);
    // whole rounds of PF steps; a step past the slice is skipped (MFMAs only: the load pattern stays the same)
    for (int j = 0; j < len; j += PF) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        if (j + i < len) compute(ring[i]);
        load(step_of(j + i + PF), ring[i]);
      }
    }
  }

  // reduce the 4 K-slices through LDS
  extern __shared__ __attribute__((aligned(16))) char smem[];
  acc_t* red = reinterpret_cast<acc_t*>(smem);  // [3][MT][NT][64]
  if (wave > 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < NT; ++t) red[(((wave - 1) * MT + mt) * NT + t) * 64 + lane] = acc[mt][t];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[mt][t] += red[((w * MT + mt) * NT + t) * 64 + lane];

  // D layout: col = lane & 15 = activation row m, rows 4 g + r = output columns n0 + 16 t + 4 g + r
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + 16 * t + 4 * g;
      if (n >= p.N) continue;  // N % 16 == 0: the 4 columns of a lane are inside or outside together
      if (p.k_splits > 1) {
        *reinterpret_cast<acc_t*>(reinterpret_cast<char*>(p.partial) + (((int64_t)blockIdx.y * p.M + m) * p.N + n) * 4) = acc[mt][t];
      } else {
        mm_epilogue4<out_t>(p, acc[mt][t], m, n);
      }
    }
  }
}

// ---- the same contraction with both operands staged through LDS in full 128-byte lines (K % 128 == 0) ----------------
// The fragment-shaped loads of scaled_mm_kernel (16 rows x 64 B per wave instruction) keep the CU's texture-address
// path busy ~3.5x longer per instruction than line-shaped ones (measured: ~140 vs ~38 cycles; the guide's M = 256
// projection study reports TA_BUSY 2x for the same shape). Here a wave instruction reads 8 rows x one whole 128-byte
// line; the 16-byte chunks go to a wave-private LDS image [row][128 B] with chunk c stored at c ^ ((row >> 1) & 7)
// (ds_write_b128 and the ds_read_b128 of 16 consecutive rows are then both bank-conflict free) and the MFMA fragments
// (row i, bytes 64 q + 16 g) are read back from there. No workgroup barrier in the loop: every wave owns a K slice, its
// image and its registers; LDS operations of one wave complete in order. The next stage's global loads are issued
// before the current stage's MFMAs (two register sets).
// grid (ceil(N / 64), k_splits, ceil(M / (16 MT))), 4 waves, LDS 4 x (64 + 16 MT) x 128 B.
// WNT: non-temporal hint on the B (weight) loads - set when the launch has one row block, i.e. reads every weight byte once
template <typename out_t, bool FP8, int MT, bool WNT>
__global__ __launch_bounds__(256, 2) void scaled_mm_lds_kernel(const MmParams p) {
  constexpr int NT = 4;
  constexpr int BROWS = 64, AROWS = 16 * MT;
  constexpr int BI = BROWS / 8, AI = AROWS / 8;  // load instructions per stage (8 rows x 128 B each)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int lr = lane >> 3, lc = lane & 7;  // load role: row within the 8-row piece, 16-byte chunk of the line
  const int n0 = blockIdx.x * 64, m0 = blockIdx.z * 16 * MT;
  using acc_t = typename std::conditional<FP8, f32x4, i32x4>::type;
  acc_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[mt][t] = acc_t{0, 0, 0, 0};

  const int stages = p.K / 128;
  const int per = (stages + p.k_splits - 1) / p.k_splits;
  const int sb = min((int)blockIdx.y * per, stages), se = min(sb + per, stages);
  const int pw = (se - sb + 3) / 4;
  const int ws = min(sb + wave * pw, se), we = min(ws + pw, se);
  const int len = we - ws;

  const __amdgpu_buffer_rsrc_t rs_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.bt), 0, (int)((int64_t)p.N * p.ldb), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.a), 0, (int)((int64_t)p.M * p.lda), 0x00020000);
  int b_voff[BI], a_voff[AI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int n = n0 + 8 * j + lr;
    b_voff[j] = n < p.N ? (int)(n * p.ldb + 16 * lc) : (int)0xfffffff0u;  // rows past the matrix read as zeros
  }
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int m = m0 + 8 * j + lr;
    a_voff[j] = m < p.M ? (int)(m * p.lda + 16 * lc) : (int)0xfffffff0u;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* img = smem + wave * ((BROWS + AROWS) * 128);
  // row r, chunk c -> byte offset inside the image
  auto slot = [](int r, int c) { return r * 128 + 16 * (c ^ ((r >> 1) & 7)); };

  struct Stage { u32x4 b[BI]; u32x4 a[AI]; };
  auto load = [&](int s, Stage& r) {
    const int soff = min(s, stages - 1) * 128;  // past the slice: a valid stage again (never used)
#pragma unroll
    for (int j = 0; j < BI; ++j) r.b[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff[j], soff, WNT ? 2 : 0);
#pragma unroll
    for (int j = 0; j < AI; ++j) r.a[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff[j], soff, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto compute = [&](const Stage& r) {
#pragma unroll
    for (int j = 0; j < BI; ++j) *reinterpret_cast<u32x4*>(img + slot(8 * j + lr, lc)) = r.b[j];
#pragma unroll
    for (int j = 0; j < AI; ++j) *reinterpret_cast<u32x4*>(img + slot(BROWS + 8 * j + lr, lc)) = r.a[j];
#pragma unroll
    for (int q = 0; q < 2; ++q) {  // the two 64-byte halves of the line
      u32x4 bf[NT], af[MT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bf[t] = *reinterpret_cast<const u32x4*>(img + slot(16 * t + li, 4 * q + g));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const u32x4*>(img + slot(BROWS + 16 * mt + li, 4 * q + g));
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if constexpr (FP8) {
            const long b0 = (long)(((uint64_t)bf[t][1] << 32) | bf[t][0]), b1 = (long)(((uint64_t)bf[t][3] << 32) | bf[t][2]);
            const long a0 = (long)(((uint64_t)af[mt][1] << 32) | af[mt][0]), a1 = (long)(((uint64_t)af[mt][3] << 32) | af[mt][2]);
            acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b0, a0, acc[mt][t], 0, 0, 0);
            acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b1, a1, acc[mt][t], 0, 0, 0);
          } else {
            acc[mt][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, bf[t]), __builtin_bit_cast(i32x4, af[mt]),
                                                              acc[mt][t], 0, 0, 0);
          }
        }
    }
  };
  if (len > 0) {
    Stage r0, r1;
    load(ws, r0);
    load(ws + 1, r1);
    // pairs of stages, branch-free load pattern; a stage past the slice is loaded (clamped) but not computed
    for (int j = 0; j < len; j += 2) {
      compute(r0);
      load(ws + j + 2, r0);
      if (j + 1 < len) compute(r1);
      load(ws + j + 3, r1);
    }
  }

  // reduce the 4 K-slices through LDS (the images are dead: every wave is past its last fragment read at the barrier)
  __syncthreads();
  acc_t* red = reinterpret_cast<acc_t*>(smem);  // [3][MT][NT][64]
  if (wave > 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < NT; ++t) red[(((wave - 1) * MT + mt) * NT + t) * 64 + lane] = acc[mt][t];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[mt][t] += red[((w * MT + mt) * NT + t) * 64 + lane];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + 16 * t + 4 * g;
      if (n >= p.N) continue;
      if (p.k_splits > 1) {
        *reinterpret_cast<acc_t*>(reinterpret_cast<char*>(p.partial) + (((int64_t)blockIdx.y * p.M + m) * p.N + n) * 4) = acc[mt][t];
      } else {
        mm_epilogue4<out_t>(p, acc[mt][t], m, n);
      }
    }
  }
}

// ---- M > 64: workgroup-tiled kernel ------------------------------------------------------------------------------------
// The kernels above give every wave its own K slice of a 64-column tile and let it fetch both operands itself: right for
// decode (few rows, the weight stream is the whole cost), but at M = 256 a wave issues one 1-KiB load + one LDS write +
// one LDS read per two MFMAs and the weights are fetched once per 64-row block (gate_up fp8 at M = 256: 107 us =
// 0.11 of the fp8 MFMA peak). Here a workgroup of WM x WN waves owns a (64 WM) x (64 WN) output tile over the whole K
// range of its split: per 128-byte stage the operand tiles are loaded ONCE per workgroup in whole 128-byte lines (8 rows
// per wave instruction), parked in LDS as [row][128 B] with the chunk swizzle of scaled_mm_lds_kernel, and every wave
// reads the fragments of its 64 x 64 sub-tile: 16 ds_read_b128 per 64 (fp8) / 32 (int8) MFMAs. LDS double-buffered, the
// next stage's global loads in flight during the MFMAs, one workgroup barrier per stage.
// grid (ceil(N / (64 WN)), k_splits, ceil(M / (64 WM))), block 64 WM WN; LDS NBUF x 64 (WM + WN) x 128 B.
template <typename out_t, bool FP8, int WM, int WN, int NBUF, bool PIPE = false>
__global__ __launch_bounds__(64 * WM * WN) void scaled_mm_tile_kernel(const MmParams p) {
  static_assert(!PIPE || (FP8 && NBUF == 3), "the fragment-pipelined form is the fp8 three-buffer DMA loop");
  constexpr int NWAVE = WM * WN;
  constexpr int AROWS = 64 * WM, BROWS = 64 * WN;
  constexpr int PIECES = (AROWS + BROWS) / 8;     // 8-row x 128-byte pieces per stage
  constexpr int PW = PIECES / NWAVE;              // per wave
  static_assert(PIECES % NWAVE == 0, "pieces must divide over the waves");
  constexpr int IMG = (AROWS + BROWS) * 128;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int lr = lane >> 3, lc = lane & 7;
  const int wn = wave % WN, wm = wave / WN;
  const int n0 = blockIdx.x * BROWS, m0 = blockIdx.z * AROWS;
  using acc_t = typename std::conditional<FP8, f32x4, i32x4>::type;
  acc_t acc[4][4];  // [mt][t]: rows m0 + 64 wm + 16 mt + li, columns n0 + 64 wn + 16 t + 4 g + r
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mt][t] = acc_t{0, 0, 0, 0};

  const int stages = p.K / 128;
  const int per = (stages + p.k_splits - 1) / p.k_splits;
  const int sb = min((int)blockIdx.y * per, stages), se = min(sb + per, stages);

  const __amdgpu_buffer_rsrc_t rs_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.bt), 0, (int)((int64_t)p.N * p.ldb), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.a), 0, (int)((int64_t)p.M * p.lda), 0x00020000);
  // piece q of the stage: q < AROWS / 8 -> A rows 8 q .. 8 q + 7, else B rows; image row index = the same order
  int voff[PW], lds_off[PW];
  bool is_a[PW];
  // chunk swizzle inside a 128-byte row: the writes (8 lanes = one row's 8 chunks) are conflict-free under any per-row
  // XOR; the fragment reads differ: int8 reads chunk 4 q + g of 16 consecutive rows (XOR with (r >> 1) & 7), fp8 reads
  // chunks 2 g and 2 g + 1 (XOR with 2 ((r >> 1) & 3) + ((r >> 3) & 1): found by exhaustive search over the ds_read_b128
  // lane groups; with the int8 swizzle the fp8 reads were 2-way conflicted, SQ_LDS_BANK_CONFLICT = 2.9 cycles per read)
  auto slot = [](int r, int c) {
    const int f = FP8 ? (2 * ((r >> 1) & 3) + ((r >> 3) & 1)) : ((r >> 1) & 7);
    return r * 128 + 16 * (c ^ f);
  };
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int q = wave * PW + j;
    const int r = 8 * q + lr;  // image row
    is_a[j] = q < AROWS / 8;
    if (is_a[j]) {
      const int m = m0 + r;
      voff[j] = m < p.M ? (int)(m * p.lda + 16 * lc) : (int)0xfffffff0u;  // rows past the matrix read as zeros
    } else {
      const int n = n0 + r - AROWS;
      voff[j] = n < p.N ? (int)(n * p.ldb + 16 * lc) : (int)0xfffffff0u;
    }
    lds_off[j] = slot(r, lc);
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // Two register sets: the loads of stage s + 2 are issued before the MFMAs of stage s and written to LDS at the end of
  // stage s + 1 - a whole stage of MFMAs (~1k cycles) is shorter than a loaded HBM / L2 round trip.
  // Every workgroup reads byte columns [128 s, 128 s + 128) of rows that are K (a power of two) bytes apart: walking s in
  // the same order everywhere sends all of them to the same few HBM channels / L2 lines at the same time. Each column tile
  // therefore starts its walk at its own stage and wraps (the row blocks of a column tile share the order so that they
  // share the weight lines in L2); the sum is the same set of products in a rotated order.
  const int nst = se - sb;
  const int rot = nst > 1 ? (int)((blockIdx.x * 11u) % (unsigned)nst) : 0;
  auto load = [&](int s, u32x4 (&regs)[PW]) {
    int r = min(s - sb, nst - 1) + rot;  // past the split: a valid stage again (never computed)
    r = r >= nst ? r - nst : r;
    const int soff = (sb + r) * 128;
    if constexpr ((NMX_TABLATE & 1) != 0) return;
#pragma unroll
    for (int j = 0; j < PW; ++j)
      regs[j] = is_a[j] ? __builtin_amdgcn_raw_buffer_load_b128(rs_a, voff[j], soff, 0)
                        : __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff[j], soff, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto store = [&](char* img, const u32x4 (&regs)[PW]) {
    if constexpr ((NMX_TABLATE & 4) != 0) return;
#pragma unroll
    for (int j = 0; j < PW; ++j) *reinterpret_cast<u32x4*>(img + lds_off[j]) = regs[j];
  };
  // fp8: ONE v_mfma_scale_f32_16x16x128_f8f6f4 per (row tile, column tile) and stage - the block-scaled gfx950 form with
  // unit scales (E8M0 127) runs e4m3 at twice the per-clock rate of v_mfma_f32_16x16x32_fp8_fp8 (which has the fp16
  // rate: rocprofv3 counted 16 MFMA-busy cycles per instruction). Lane (g, i) supplies bytes 32 g .. 32 g + 31 of row
  // i of the stage for both operands; the hardware's k order inside the instruction is irrelevant as long as the two
  // operands use the same one.
  auto compute = [&](const char* img) {
    const char* ia = img + (64 * wm) * 128;
    const char* ib = img + (AROWS + 64 * wn) * 128;
    if constexpr (FP8) {
      i32x8 bf[4], af[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(ib + slot(16 * t + li, 2 * g));
        const u32x4 hi = *reinterpret_cast<const u32x4*>(ib + slot(16 * t + li, 2 * g + 1));
        bf[t] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(ia + slot(16 * mt + li, 2 * g));
        const u32x4 hi = *reinterpret_cast<const u32x4*>(ia + slot(16 * mt + li, 2 * g + 1));
        af[mt] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[mt][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[t], af[mt], acc[mt][t], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {  // the two 64-byte halves of the line
        u32x4 bf[4], af[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) bf[t] = *reinterpret_cast<const u32x4*>(ib + slot(16 * t + li, 4 * q + g));
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const u32x4*>(ia + slot(16 * mt + li, 4 * q + g));
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            acc[mt][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, bf[t]), __builtin_bit_cast(i32x4, af[mt]),
                                                              acc[mt][t], 0, 0, 0);
      }
    }
  };

  // ---- fp8, three LDS buffers: LDS-DMA staging -------------------------------------------------------------------------------
  // Timing ablations of the register-staged loop (gate_up, M = 256, 45.9 us): without the LDS writes 33.4, without the
  // global loads 37.0, without the MFMAs 41.2, MFMAs + barriers alone 27.0 - the ds_write_b128 pass (64 B/clk write path,
  // 48 KiB per stage and CU) was the largest removable term. `buffer_load_dwordx4 ... lds` deposits a wave instruction's
  // 8 rows x 128 B straight into the image (lane l -> base + 16 l, so the chunk swizzle moves to the SOURCE address: lane
  // (row lr, slot lc) fetches chunk lc ^ f(row)); no staging registers, no LDS store instructions. The DMA of stage
  // j + 2 is issued before the MFMAs of stage j; a counted vmcnt lets it stay in flight across the raw s_barrier that
  // publishes stage j + 1 (a __syncthreads() would drain it: its fence waits vmcnt(0) while an LDS-DMA is pending).
  int gvoff[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int q = wave * PW + j;
    const int r = 8 * q + lr;  // image row of this lane's piece
    const int f = 2 * ((r >> 1) & 3) + ((r >> 3) & 1);
    const int c = lc ^ f;
    if (q < AROWS / 8) gvoff[j] = (int)(min(m0 + r, p.M - 1) * p.lda + 16 * c);   // rows past the matrix: a valid row again,
    else gvoff[j] = (int)(min(n0 + r - AROWS, p.N - 1) * p.ldb + 16 * c);         // its outputs are never stored
    // timing experiment (wrong products): the weight tile of a stage as ONE contiguous BROWS x 128-byte block
    if constexpr ((NMX_TABLATE & 8) != 0)
      if (q >= AROWS / 8) gvoff[j] = (int)(n0 * p.ldb + (r - AROWS) * 128 + 16 * c);
  }
  auto dma = [&](int s, char* img) {
    int r = min(s - sb, nst - 1) + rot;
    r = r >= nst ? r - nst : r;
    const int soff = (sb + r) * 128;
    if constexpr ((NMX_TABLATE & 1) != 0) return;
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the kernel's stub; it has no LDS address space to cast to
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      auto* dst = (__attribute__((address_space(3))) void*)(img + (wave * PW + j) * 1024);
      if (is_a[j]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, dst, 16, gvoff[j], soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, dst, 16, gvoff[j], soff, 0, 0);
    }
#endif
  };

  if (sb < se) {
    u32x4 ra[PW], rb[PW];
    // (with two buffers - one stage of lookahead, vmcnt(0) at every barrier - the DMA form measured 10-20 % SLOWER than the
    // two-register-set pipeline below, which keeps two stages of loads in flight: three buffers only)
    if constexpr (PIPE) {
      // Round 3: fragment-pipelined form. In the loop below (round 2) a stage is [16 fragment reads -> wait -> 16 MFMAs ->
      // barrier] and the eight waves of the workgroup run it in step, so the LDS array and the matrix pipes take turns
      // (rocprofv3: 2,830 cycles per stage against 1,024 of MFMA issue for the two waves of a SIMD). Here the fragments of
      // stage s + 1 are read into a second register set BETWEEN the MFMAs of stage s (one ds_read_b128 per MFMA), and the
      // stage's DMA instructions are spread over the same MFMAs, so every wave is in its MFMA stream all the time. Stage s
      // lives in registers while it is multiplied, hence the three buffers hold stages s + 1 (being read), s + 2 (landing,
      // awaited at the end of s) and s + 3 (issued during s into the buffer stage s left at the previous barrier).
      u32x4 fr[2][16];  // [set][2 o + half]: o = 0..3 weight column tiles, 4..7 activation row tiles
      const int iao = (64 * wm) * 128, ibo = (AROWS + 64 * wn) * 128;
      auto rd = [&](const char* img, int o, int h) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(img + (o < 4 ? ibo : iao) + slot(16 * (o & 3) + li, 2 * g + h));
      };
      auto stage_soff = [&](int s) {
        int r = min(s - sb, nst - 1) + rot;
        r = r >= nst ? r - nst : r;
        return (sb + r) * 128;
      };
      auto dma_one = [&](int j, int soff, char* img) {
        if constexpr ((NMX_TABLATE & 1) != 0) return;            // timing experiments: no in-loop DMA at all,
        if constexpr ((NMX_TABLATE & 2) != 0) if (!is_a[j]) return;  // activations only,
        if constexpr ((NMX_TABLATE & 16) != 0) if (is_a[j]) return;  // weights only
#if defined(__HIP_DEVICE_COMPILE__)
        auto* dst = (__attribute__((address_space(3))) void*)(img + (wave * PW + j) * 1024);
        if (is_a[j]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, dst, 16, gvoff[j], soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, dst, 16, gvoff[j], (NMX_TABLATE & 8) != 0 ? soff * BROWS : soff, 0, 0);
#endif
      };
      auto cat8 = [](const u32x4& lo, const u32x4& hi) {
        return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      };
      // stage s (fragments in set X) is multiplied while stage s + 1 is read from `inext` into set X ^ 1 and stage s + 3 is
      // fetched into `idma` (the buffer stage s was read from)
      auto body = [&](int s, const char* inext, char* idma, auto setc) {
        constexpr int X = decltype(setc)::value, Y = X ^ 1;
        const int soff = stage_soff(s + 3);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int t = i >> 2, mt = i & 3;
          acc[mt][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat8(fr[X][2 * t], fr[X][2 * t + 1]),
                                                                        cat8(fr[X][8 + 2 * mt], fr[X][9 + 2 * mt]), acc[mt][t], 0, 0,
                                                                        0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          // next stage's fragments in the order its MFMAs need them: weight tile 0, the four row tiles, weight tiles 1..3
          constexpr int order[16] = {0, 1, 8, 9, 10, 11, 12, 13, 14, 15, 2, 3, 4, 5, 6, 7};
          fr[Y][order[i]] = rd(inext, order[i] >> 1, order[i] & 1);
#pragma unroll
          for (int j = 0; j < PW; ++j)
            if (j * 16 / PW == i) dma_one(j, soff, idma);
          __builtin_amdgcn_sched_barrier(0);
        }
        // own reads of stage s + 1 done (its buffer takes stage s + 4 next), stage s + 2 landed (all but the newest DMAs)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PW) : "memory");
      };
      dma(sb, smem);
      dma(sb + 1, smem + IMG);
      dma(sb + 2, smem + 2 * IMG);
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * PW) : "memory");  // stage 0 landed everywhere
#pragma unroll
      for (int i = 0; i < 16; ++i) fr[0][i] = rd(smem, i >> 1, i & 1);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PW) : "memory");  // stage 1 landed, buffer 0 released
      int cur = 0;  // buffer stage s was read from
      for (int s = sb; s < se; s += 2) {
        const int n1 = cur + 1 == NBUF ? 0 : cur + 1, n2 = n1 + 1 == NBUF ? 0 : n1 + 1;
        body(s, smem + n1 * IMG, smem + cur * IMG, std::integral_constant<int, 0>{});
        if (s + 1 >= se) break;
        body(s + 1, smem + n2 * IMG, smem + n1 * IMG, std::integral_constant<int, 1>{});
        cur = n2;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped DMAs past the split: nothing may land after the exit
    } else if constexpr (FP8 && NBUF == 3) {
      constexpr int LA = NBUF - 1;  // stages of DMA lookahead
      dma(sb, smem);
      if constexpr (LA == 2) dma(sb + 1, smem + IMG);
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PW * (LA - 1)) : "memory");  // stage 0 landed everywhere
      int cur = 0;
      for (int s = sb; s < se; ++s) {
        int nb = cur + LA;
        nb = nb >= NBUF ? nb - NBUF : nb;
        dma(s + LA, smem + nb * IMG);  // the buffer stage s - 1 released at the last barrier
        compute(smem + cur * IMG);
        // own fragment reads done (the buffer is overwritten next stage), stage s + 1 landed (all but the newest DMAs)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PW * (LA - 1)) : "memory");
        cur = cur + 1 == NBUF ? 0 : cur + 1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped DMAs past the split: nothing may land after the exit
    } else {
    load(sb, ra);
    load(sb + 1, rb);
    store(smem, ra);
    load(sb + 2, ra);
    __syncthreads();
    // stage pairs, branch-free load pattern (stages past the split are loaded clamped and never computed): at the top of
    // an iteration the LDS holds stage s, rb stage s + 1 (in flight), ra stage s + 2 (in flight).
    // NBUF = 3: stage s + 1 goes to a THIRD buffer before the MFMAs of stage s, so the LDS writes (about as long as the
    // fragment reads: the write path moves 64 B/clk) run under the MFMAs instead of after them; NBUF = 2 (the 4-wave
    // tile, two workgroups per CU) writes after the MFMAs into the buffer the previous stage has released.
    int cur = 0;
    for (int s = sb; s < se; s += 2) {
      const int n1 = cur + 1 == NBUF ? 0 : cur + 1, n2 = n1 + 1 == NBUF ? 0 : n1 + 1;
      if constexpr (NBUF == 3) store(smem + n1 * IMG, rb);
      if constexpr (NBUF == 3) load(s + 3, rb);
      compute(smem + cur * IMG);
      if constexpr (NBUF == 2) store(smem + n1 * IMG, rb);
      if constexpr (NBUF == 2) load(s + 3, rb);
      __syncthreads();
      if constexpr (NBUF == 3) store(smem + n2 * IMG, ra);
      if constexpr (NBUF == 3) load(s + 4, ra);
      if (s + 1 < se) compute(smem + n1 * IMG);
      if constexpr (NBUF == 2) store(smem + n2 * IMG, ra);
      if constexpr (NBUF == 2) load(s + 4, ra);
      __syncthreads();
      cur = n2;
    }
    }
  }

#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + 64 * wm + 16 * mt + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = n0 + 64 * wn + 16 * t + 4 * g;
      if (n >= p.N) continue;
      if (p.k_splits > 1) {
        *reinterpret_cast<acc_t*>(reinterpret_cast<char*>(p.partial) + (((int64_t)blockIdx.y * p.M + m) * p.N + n) * 4) = acc[mt][t];
      } else {
        mm_epilogue4<out_t>(p, acc[mt][t], m, n);
      }
    }
  }
}

// tile shape and K splits of scaled_mm_tile_kernel: 128 x 256 tiles when they alone give >= 128 workgroups, else
// 128 x 128 tiles with K splits until >= 192 workgroups exist (>= 8 stages per split)
struct TileCfg { int wn, splits, pipe, wm; };
inline TileCfg mm_tile_cfg(int M, int N, int K, bool fp8) {
  TileCfg c{4, 1, 1, 2};
  const int rows = ceil_div(M, 128), stages = K / 128;
  // NMX_MM_TILE="wn[,splits[,form]]" (sweeps): wn 2 / 4 = the 128- / 256-column tile, splits 0 = the rule's, form 0 = the
  // round-2 loop (fp8, wn 4: DMA, fragments read per stage; wn 2: register-staged), 1 = the fragment-pipelined DMA loop
  int force = 0, fsplits = 0;
  if (const char* e = nmx_tune(NMX_TUNE_MM_TILE)) {
    int f = 1, fwm = 2;
    const int got = sscanf(e, "%d,%d,%d,%d", &force, &fsplits, &f, &fwm);
    if (got >= 3) c.pipe = f;
    if (got >= 4 && fp8 && fwm == 4 && force == 2) {  // forced tall tile (sweeps): 256 rows x 128 columns, fragment-pipelined
      c.wn = 2, c.wm = 4, c.pipe = 1;
      c.splits = fsplits > 0 ? fsplits : 1;
      return c;
    }
  }
  // fp8, 128 < M (round 3, late): TALL tiles - 256 rows x 128 columns, same eight waves and the same fragment-pipelined loop -
  // fetch every weight line once per 256 rows. Where they alone fill the chip they beat the 128 x 256 tile (gate_up M = 256:
  // 43.7 vs 45.4 us, M = 512: 75.1 vs 81.5; same-process A/B, gpurun_out/mm_tall1.txt -> profiles/r03_fp8_tile_experiments.txt),
  // and with two K splits the mid-size matrices from 512 rows (qkv M = 512: 22.3 vs 28.8 us); o / down stay as they were.
  if (fp8 && force == 0 && c.pipe != 0 && M > 128) {
    const int tall = ceil_div(M, 256) * ceil_div(N, 128);
    if (tall >= 192) return TileCfg{2, 1, 1, 4};
    if (M > 256 && tall * 2 >= 160 && tall * 2 <= 256 && stages / 2 >= 8 && K < 8192) return TileCfg{2, 2, 1, 4};
  }
  // long K with few tiles (down: 14336 x 4096): the 256-column DMA tile with K splits - int8 from M = 65, fp8 above M = 256
  // (fp8, round 3: the fragment-pipelined 128-column tile is ahead up to there - M = 128: 24.8 vs 29.2 us, M = 256: 31.9 vs
  // 35.2, M = 512: 50.2 vs 44.5; profiles/r03_fp8_tile_experiments.txt)
  const int tiles4 = rows * ceil_div(N, 256);
  const bool long_k = K >= 8192 && tiles4 >= 16 && !(fp8 && c.pipe != 0 && rows <= 2);
  if (force != 2 && (force == 4 || long_k || tiles4 >= 128)) {
    while ((force == 4 || long_k) && tiles4 * c.splits < 192 && c.splits < 16 && stages / (c.splits * 2) >= 4) c.splits *= 2;
    if (fsplits > 0) c.splits = fsplits;
    return c;
  }
  c.wn = 2;
  const int tiles = rows * ceil_div(N, 128);
  while (tiles * c.splits < 192 && c.splits < 16 && stages / (c.splits * 2) >= 8) c.splits *= 2;
  if (fsplits > 0) c.splits = fsplits;
  return c;
}
inline bool mm_use_tile(int M, int N, int K) {
  if (const char* e = nmx_tune(NMX_TUNE_MM_TILE)) return atoi(e) != 0 && K % 128 == 0;
  return M > 64 && K % 128 == 0;
}

// out = epilogue(sum_s partial[s]); 4 columns per thread; summation order s = 0, 1, ... (deterministic)
template <typename out_t, bool FP8>
__global__ void scaled_mm_reduce_kernel(const MmParams p) {
  using acc_t = typename std::conditional<FP8, f32x4, i32x4>::type;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t mn4 = (int64_t)p.M * p.N / 4;
  if (i >= mn4) return;
  const acc_t* part = reinterpret_cast<const acc_t*>(p.partial);
  acc_t acc = part[i];
  for (int s = 1; s < p.k_splits; ++s) acc += part[(int64_t)s * mn4 + i];
  const int m = (int)((i * 4) / p.N), n = (int)((i * 4) % p.N);
  mm_epilogue4<out_t>(p, acc, m, n);
}

// K splits: fill the chip when the output tiles alone do not, keeping >= 4 k-steps per wave
inline int mm_splits(int M, int N, int K) {
  const int mt = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
  const int tiles = ceil_div(N, 64) * ceil_div(M, 16 * mt), steps = ceil_div(K, 64);
  int sp = 1;
  while (tiles * sp < 192 && sp < 16 && steps / (4 * sp * 2) >= 4) sp *= 2;
  return sp;
}

template <typename out_t, bool FP8>
int launch_mm_tile(MmParams& p, void* scratch, int64_t scratch_bytes, hipStream_t stream) {
  const TileCfg c = mm_tile_cfg(p.M, p.N, p.K, FP8);
  p.k_splits = c.splits;
  const int64_t per = (int64_t)p.M * p.N * 4;
  if (p.k_splits > 1) {  // never allocate here (graph capture): degrade to what fits
    const int fit = scratch == nullptr ? 1 : (int)std::min<int64_t>(p.k_splits, scratch_bytes / per);
    p.k_splits = std::max(1, fit);
  }
  p.partial = scratch;
  dim3 grid(ceil_div(p.N, 64 * c.wn), p.k_splits, ceil_div(p.M, 64 * c.wm));
  const bool pipe = FP8 && c.pipe != 0;
  const int smem = ((c.wn == 4 || pipe) ? 3 : 2) * 64 * (c.wm + c.wn) * 128;
  auto go = [&](auto kern, int threads) {
    NMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    kern<<<grid, threads, smem, stream>>>(p);
    return NMX_OK;
  };
  if constexpr (FP8) {
    if (pipe) {
      const int rc = c.wm == 4 ? go(scaled_mm_tile_kernel<out_t, true, 4, 2, 3, true>, 512)
                   : c.wn == 4 ? go(scaled_mm_tile_kernel<out_t, true, 2, 4, 3, true>, 512)
                               : go(scaled_mm_tile_kernel<out_t, true, 2, 2, 3, true>, 256);
      if (rc != NMX_OK) return rc;
    }
  }
  if (!pipe) {
    const int rc = c.wn == 4 ? go(scaled_mm_tile_kernel<out_t, FP8, 2, 4, 3>, 512) : go(scaled_mm_tile_kernel<out_t, FP8, 2, 2, 2>, 256);
    if (rc != NMX_OK) return rc;
  }
  NMX_LAUNCH_CHECK();
  if (p.k_splits > 1 && !p.defer_reduce) {
    const int64_t mn4 = (int64_t)p.M * p.N / 4;
    scaled_mm_reduce_kernel<out_t, FP8><<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>(p);
    NMX_LAUNCH_CHECK();
  }
  return NMX_OK;
}

template <typename out_t, bool FP8>
int launch_mm(MmParams& p, void* scratch, int64_t scratch_bytes, hipStream_t stream) {
  if (mm_use_tile(p.M, p.N, p.K)) return launch_mm_tile<out_t, FP8>(p, scratch, scratch_bytes, stream);
  const int mt = p.M <= 16 ? 1 : (p.M <= 32 ? 2 : 4);
  p.k_splits = mm_splits(p.M, p.N, p.K);
  const int64_t per = (int64_t)p.M * p.N * 4;
  if (p.k_splits > 1) {  // never allocate here (graph capture): degrade to what fits
    const int fit = scratch == nullptr ? 1 : (int)std::min<int64_t>(p.k_splits, scratch_bytes / per);
    p.k_splits = std::max(1, fit);
  }
  p.partial = scratch;
  dim3 grid(ceil_div(p.N, 64), p.k_splits, ceil_div(p.M, 16 * mt));
  const size_t smem = (size_t)3 * mt * 4 * 64 * 16;
  if (p.K % 128 == 0 && nmx_tune(NMX_TUNE_MM_NO_LDS) == nullptr) {
    const size_t img = (size_t)4 * (64 + 16 * mt) * 128;
    const size_t smem_l = std::max(img, smem);
    // measured on the fp8 decode step (batch 64): 5.62 ms with the hint, 5.56 without - off unless NMX_MM_NT is set
    const bool nt = grid.z == 1 && nmx_tune(NMX_TUNE_MM_NT) != nullptr;
    switch (mt) {
      case 1: scaled_mm_lds_kernel<out_t, FP8, 1, false><<<grid, 256, smem_l, stream>>>(p); break;
      case 2: scaled_mm_lds_kernel<out_t, FP8, 2, false><<<grid, 256, smem_l, stream>>>(p); break;
      default:
        if (nt) scaled_mm_lds_kernel<out_t, FP8, 4, true><<<grid, 256, smem_l, stream>>>(p);
        else scaled_mm_lds_kernel<out_t, FP8, 4, false><<<grid, 256, smem_l, stream>>>(p);
        break;
    }
  } else if (p.K % 64 == 0) {
    switch (mt) {
      case 1: scaled_mm_kernel<out_t, FP8, 1, false><<<grid, 256, smem, stream>>>(p); break;
      case 2: scaled_mm_kernel<out_t, FP8, 2, false><<<grid, 256, smem, stream>>>(p); break;
      default: scaled_mm_kernel<out_t, FP8, 4, false><<<grid, 256, smem, stream>>>(p); break;
    }
  } else {
    switch (mt) {
      case 1: scaled_mm_kernel<out_t, FP8, 1, true><<<grid, 256, smem, stream>>>(p); break;
      case 2: scaled_mm_kernel<out_t, FP8, 2, true><<<grid, 256, smem, stream>>>(p); break;
      default: scaled_mm_kernel<out_t, FP8, 4, true><<<grid, 256, smem, stream>>>(p); break;
    }
  }
  NMX_LAUNCH_CHECK();
  if (p.k_splits > 1 && !p.defer_reduce) {
    const int64_t mn4 = (int64_t)p.M * p.N / 4;
    scaled_mm_reduce_kernel<out_t, FP8><<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>(p);
    NMX_LAUNCH_CHECK();
  }
  return NMX_OK;
}

}  // namespace

extern "C" int nmx_scaled_fp8_quant(void* out, const void* input, float* scale, float* scratch, int64_t scratch_bytes,
                                    int64_t numel, int dtype, int dynamic, nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (numel == 0) return NMX_OK;
  const int esz = dtype == NMX_F32 ? 4 : 2;
  // 8 elements per thread and access when the pointers allow it (n8 = number of such vectors), else element-wise
  const bool vec = (uintptr_t)input % (8 * esz) == 0 && (uintptr_t)out % 8 == 0;
  const int64_t n8 = vec ? numel / 8 : 0;
  const int threads = 1024;
  // one 16-byte piece per thread and iteration; at most kAbsmaxBlocks workgroups (one partial maximum per lane of a wave)
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div64(vec ? numel / 8 : numel, threads), kAbsmaxBlocks));
  if (dynamic)
    NMX_CHECK(scratch != nullptr && scratch_bytes >= (int64_t)kAbsmaxBlocks * 4, NMX_ERR_INVALID_ARG,
              "scaled_fp8_quant: dynamic scaling needs %d bytes of scratch", kAbsmaxBlocks * 4);
  const int qblocks = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div64(vec ? numel / 8 : numel, threads), 1024));
#define NMX_FP8Q(T)                                                                                              \
  do {                                                                                                           \
    if (dynamic) {                                                                                               \
      fp8_absmax_kernel<T><<<blocks, threads, 0, stream>>>(scratch, (const T*)input, numel, n8);                  \
      fp8_quant_kernel<T, true><<<qblocks, threads, 0, stream>>>((uint8_t*)out, (const T*)input, scale, scratch, \
                                                                 blocks, numel, n8);                             \
    } else {                                                                                                     \
      fp8_quant_kernel<T, false><<<qblocks, threads, 0, stream>>>((uint8_t*)out, (const T*)input, scale, nullptr, \
                                                                  0, numel, n8);                                 \
    }                                                                                                            \
  } while (0)
  switch (dtype) {
    case NMX_F32: NMX_FP8Q(float); break;
    case NMX_F16: NMX_FP8Q(f16); break;
    case NMX_BF16: NMX_FP8Q(bf16); break;
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "scaled_fp8_quant: unsupported dtype %d", dtype);
  }
#undef NMX_FP8Q
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

// Dynamic per-tensor fp8 quantisation from partial maxima that the PRODUCER of `input` left behind
// (nmx_rms_norm_absmax / nmx_fused_add_rms_norm_absmax / nmx_act_and_mul_absmax: one |max| per token): one launch
// instead of absmax + quantise, the same scale and codes bit for bit (the maximum does not depend on how it is grouped).
extern "C" int nmx_scaled_fp8_quant_partials(void* out, const void* input, float* scale, const float* partials, int nparts,
                                             int64_t numel, int dtype, nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (numel == 0) return NMX_OK;
  NMX_CHECK(partials != nullptr && nparts > 0, NMX_ERR_INVALID_ARG, "scaled_fp8_quant_partials: needs the producer's maxima");
  const int esz = dtype == NMX_F32 ? 4 : 2;
  const bool vec = (uintptr_t)input % (8 * esz) == 0 && (uintptr_t)out % 8 == 0;
  const int64_t n8 = vec ? numel / 8 : 0;
  const int threads = 1024;
  const int qblocks = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div64(vec ? numel / 8 : numel, threads), 1024));
  switch (dtype) {
    case NMX_F32: fp8_quant_kernel<float, true><<<qblocks, threads, 0, stream>>>((uint8_t*)out, (const float*)input, scale, partials, nparts, numel, n8); break;
    case NMX_F16: fp8_quant_kernel<f16, true><<<qblocks, threads, 0, stream>>>((uint8_t*)out, (const f16*)input, scale, partials, nparts, numel, n8); break;
    case NMX_BF16: fp8_quant_kernel<bf16, true><<<qblocks, threads, 0, stream>>>((uint8_t*)out, (const bf16*)input, scale, partials, nparts, numel, n8); break;
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "scaled_fp8_quant_partials: unsupported dtype %d", dtype);
  }
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int nmx_scaled_int8_quant(void* out, const void* input, float* scales, int num_tokens, int hidden_size,
                                     int dtype, int dynamic, nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (num_tokens == 0) return NMX_OK;
  const int threads = std::min(1024, std::max(64, ((hidden_size + 63) / 64) * 64));
#define NMX_I8Q(T)                                                                                                    \
  do {                                                                                                                \
    if (dynamic) int8_quant_kernel<T, true><<<num_tokens, threads, 0, stream>>>((int8_t*)out, (const T*)input, scales, hidden_size); \
    else int8_quant_kernel<T, false><<<num_tokens, threads, 0, stream>>>((int8_t*)out, (const T*)input, scales, hidden_size);        \
  } while (0)
  switch (dtype) {
    case NMX_F32: NMX_I8Q(float); break;
    case NMX_F16: NMX_I8Q(f16); break;
    case NMX_BF16: NMX_I8Q(bf16); break;
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "scaled_int8_quant: unsupported dtype %d", dtype);
  }
#undef NMX_I8Q
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int64_t nmx_scaled_mm_scratch_bytes(int m, int n, int k) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  const int sp = mm_use_tile(m, n, k) ? std::max(mm_tile_cfg(m, n, k, true).splits, mm_tile_cfg(m, n, k, false).splits) : mm_splits(m, n, k);
  return sp > 1 ? (int64_t)sp * m * n * 4 : 0;
}

static int scaled_mm_common(void* out, const void* a, const void* b, const float* a_scales, int a_scales_numel,
                            const float* b_scales, int b_scales_numel, const void* bias, void* scratch, int64_t scratch_bytes,
                            int m, int n, int k, int64_t lda, int64_t ldb, int64_t ldc, int is_fp8, int out_dtype, bool defer,
                            int* splits_out, hipStream_t stream) {
  if (splits_out != nullptr) *splits_out = 1;
  // checks mirror cutlass_w8a8/scaled_mm_entry.cu:59-76
  NMX_CHECK(a_scales_numel == 1 || a_scales_numel == m, NMX_ERR_INVALID_ARG, "a_scales.numel() must be 1 or a.size(0)");
  NMX_CHECK(b_scales_numel == 1 || b_scales_numel == n, NMX_ERR_INVALID_ARG, "b_scales.numel() must be 1 or b.size(1)");
  NMX_CHECK(k % 16 == 0 && n % 16 == 0, NMX_ERR_INVALID_ARG, "scaled_mm: K and N must be multiples of 16");
  NMX_CHECK(ldb % 16 == 0 && ldc % 16 == 0 && lda % 16 == 0, NMX_ERR_INVALID_ARG, "scaled_mm: 16-byte row alignment required");
  NMX_CHECK(out_dtype == NMX_F16 || out_dtype == NMX_BF16, NMX_ERR_UNSUPPORTED, "scaled_mm: output must be float16 or bfloat16");
  NMX_CHECK(((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0), NMX_ERR_INVALID_ARG, "scaled_mm: operands must be 16-byte aligned");
  NMX_CHECK(((uintptr_t)out % 8 == 0) && ((uintptr_t)bias % 8 == 0), NMX_ERR_INVALID_ARG, "scaled_mm: out and bias must be 8-byte aligned");
  if (m == 0 || n == 0) return NMX_OK;
  NMX_CHECK((int64_t)n * ldb < (1ll << 31) && (int64_t)m * lda < (1ll << 31), NMX_ERR_UNSUPPORTED,
            "scaled_mm: operands of 2 GiB or more are not supported");
  MmParams p{(const uint8_t*)a, (const uint8_t*)b, out, a_scales, b_scales, bias, m, n, k, lda, ldb, ldc,
             a_scales_numel > 1 ? 1 : 0, b_scales_numel > 1 ? 1 : 0, 1, nullptr, defer ? 1 : 0};
  int rc;
  if (out_dtype == NMX_F16)
    rc = is_fp8 ? launch_mm<f16, true>(p, scratch, scratch_bytes, stream) : launch_mm<f16, false>(p, scratch, scratch_bytes, stream);
  else
    rc = is_fp8 ? launch_mm<bf16, true>(p, scratch, scratch_bytes, stream) : launch_mm<bf16, false>(p, scratch, scratch_bytes, stream);
  if (rc == NMX_OK && splits_out != nullptr) *splits_out = p.k_splits;
  return rc;
}

extern "C" int nmx_scaled_mm(void* out, const void* a, const void* b, const float* a_scales, int a_scales_numel,
                             const float* b_scales, int b_scales_numel, const void* bias, void* scratch,
                             int64_t scratch_bytes, int m, int n, int k, int64_t lda, int64_t ldb, int64_t ldc, int is_fp8,
                             int out_dtype, nmx_stream_t stream) {
  return scaled_mm_common(out, a, b, a_scales, a_scales_numel, b_scales, b_scales_numel, bias, scratch, scratch_bytes, m, n, k, lda,
                          ldb, ldc, is_fp8, out_dtype, false, nullptr, (hipStream_t)stream);
}

// fp8 x fp8 with per-tensor scales and no bias, the K-split reduce AND the scale epilogue left to the consumer op
// (nmx_fused_add_rms_norm_splitk / nmx_silu_and_mul_splitk / nmx_rope_reshape_and_cache with their scale pointers): the raw
// fp32 slabs [*splits_out, m, n] stay in `scratch`; out is written (complete) only when *splits_out == 1.
extern "C" int nmx_scaled_mm_deferred(void* out, const void* a, const void* b, const float* a_scale, const float* b_scale,
                                      void* scratch, int64_t scratch_bytes, int m, int n, int k, int64_t lda, int64_t ldb,
                                      int64_t ldc, int out_dtype, int* splits_out, nmx_stream_t stream) {
  NMX_CHECK(splits_out != nullptr, NMX_ERR_INVALID_ARG, "scaled_mm_deferred: splits_out must be non-null");
  return scaled_mm_common(out, a, b, a_scale, 1, b_scale, 1, nullptr, scratch, scratch_bytes, m, n, k, lda, ldb, ldc, 1, out_dtype,
                          true, splits_out, (hipStream_t)stream);
}

// out [m, n] = epilogue(sum_s partial[s]) for the fp32 slabs a deferred fp8 nmx_scaled_mm_deferred left behind, per-tensor
// scales: scaled_mm_reduce_kernel itself (slabs summed s = 0, 1, ..., then sa * (sb * sum) with the fp32 product a rounding
// step of its own), i.e. the bits the plain op would have produced. For consumers without a fused form
// (DeferredGemm.materialize).
extern "C" int nmx_splitk_reduce_scaled(void* out, const float* partial, int splits, const float* a_scale, const float* b_scale,
                                        int m, int n, int64_t ldc, int out_dtype, nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NMX_CHECK(out_dtype == NMX_F16 || out_dtype == NMX_BF16, NMX_ERR_UNSUPPORTED, "splitk_reduce_scaled: fp16 / bf16 output only");
  NMX_CHECK(splits >= 1 && n % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)out % 8 == 0) && ((uintptr_t)partial % 16 == 0), NMX_ERR_INVALID_ARG,
            "splitk_reduce_scaled: n %% 4 == 0, ldc %% 4 == 0, out 8-byte and partial 16-byte aligned");
  NMX_CHECK(a_scale != nullptr && b_scale != nullptr, NMX_ERR_INVALID_ARG, "splitk_reduce_scaled: scales must be non-null");
  if ((int64_t)m * n == 0) return NMX_OK;
  MmParams p{};
  p.out = out; p.a_scales = a_scale; p.b_scales = b_scale; p.bias = nullptr;
  p.M = m; p.N = n; p.K = 0; p.ldc = ldc; p.a_per_row = 0; p.b_per_col = 0; p.k_splits = splits;
  p.partial = const_cast<float*>(partial);
  const int64_t mn4 = (int64_t)m * n / 4;
  if (out_dtype == NMX_F16) scaled_mm_reduce_kernel<f16, true><<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>(p);
  else scaled_mm_reduce_kernel<bf16, true><<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>(p);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int nmx_scaled_mm_supports_fp8(int capability) {
  (void)capability;
  return 1;  // gfx950 has native OCP fp8 MFMA (cutlass_scaled_mm_supports_fp8, scaled_mm_entry.cu:25-45)
}
