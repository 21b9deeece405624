// Marlin-format W4A16 / W8A16 / fp8-W8A16 GEMMs and the GPTQ->Marlin repack for gfx950.
//
// Replaces csrc/quantization/gptq_marlin/{gptq_marlin.cu, gptq_marlin_repack.cu},
// csrc/quantization/marlin/dense/marlin_cuda_kernel.cu and csrc/quantization/fp8/fp8_marlin.cu of the reference.
//
// The op contract hands over weights in the *Marlin layout* (built for NVIDIA mma.m16n8k16 fragments). It turns out
// to be directly consumable by MFMA 16x16x32 with zero re-layout:
//   * one lane's 16-byte load of a Marlin row (k-tile kt, 64-column group, chunk i = 4 c + m) holds, for the 8
//     columns {c, c+8} + 16 j (j = 0..3), the 4 k-rows {2m, 2m+1, 2m+8, 2m+9} of the 16-row k-tile;
//   * taking k-tiles 2 ks and 2 ks + 1 gives the lane 8 k-values for each of 8 columns = eight MFMA operand
//     fragments, provided the activation operand uses the same k order inside each 32-k step
//     (slot (g, jj) <-> k = 16 (jj >> 2) + 2 g + {0, 1, 8, 9}[jj & 3]);
//   * a wave's 64 lanes (c = lane & 7, m = lane >> 4, column-group = (lane >> 3) & 1) cover 1 KiB of contiguous
//     HBM per load instruction: fully coalesced, every fetched bit is used exactly once.
// So the weight stream goes HBM -> VGPR -> dequant (2 VALU per packed pair) -> MFMA with no LDS round trip and no
// repack pass. Activations (tiny, L2-resident) are staged per wave into LDS in fragment order.
//
// This file holds the "skinny" kernel (M <= 64 rows per pass, HBM-bound regime of decode). Rows are processed in
// blocks of 16 * MT; K is split over the 4 waves of a workgroup (LDS tree reduce) and over gridDim.y workgroups
// (fp32 partial slabs + a small reduce kernel).
//
// Algorithmic bytes per call: K*N*bits/8 (weights) + groups*N*2 (scales) + 2*M*K + 2*M*N.
#include <type_traits>

#include "nmx_common.h"

namespace {

constexpr int kSubSteps = 4;  // 32-k steps per activation staging sub-chunk (128 k)

enum WeightKind { W_INT4 = 0, W_INT8 = 1, W_FP8 = 2 };

template <typename scalar_t>
__device__ __forceinline__ f32x4 mfma_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (__is_same(scalar_t, f16)) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
}

__device__ __forceinline__ uint32_t h2_bits(f16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ f16x2 bits_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  union { bf16 h[2]; uint32_t u; } r;
  r.h[0] = (bf16)lo;
  r.h[1] = (bf16)hi;
  return r.u;
}

// ---- dequantisation of one packed dword into two operand dwords --------------------------------------------
// int4, fp16: q holds (after the optional >> 8 for the "+8 column" half) nibbles n0 n1 . . n4 n5 . .
//   (n0, n4) = k-rows (2m, 2m+1), (n1, n5) = k-rows (2m+8, 2m+9).
// Exact integer -> fp16 conversion with the 0x6400 exponent trick (same constants the reference relies on,
// gptq_marlin.cu:162-181): (q & 0x000f000f) | 0x64006400 = 1024 + v ; (q & 0x00f000f0) | 0x64006400 = 1024 + 16 v.
template <typename scalar_t, int KIND>
struct Dequant;

template <>
struct Dequant<f16, W_INT4> {
  // s2 = (s, s) packed fp16 scale for this column, or 1.0
  static __device__ __forceinline__ void run(uint32_t q, uint32_t s2, bool scaled, uint32_t& w01, uint32_t& w23) {
    const f16x2 a = bits_h2((q & 0x000f000fu) | 0x64006400u) - bits_h2(0x64086408u);                       // v - 8
    const f16x2 b = bits_h2((q & 0x00f000f0u) | 0x64006400u) * bits_h2(0x2c002c00u) + bits_h2(0xd480d480u);  // /16 - 72
    if (scaled) {
      w01 = h2_bits(a * bits_h2(s2));
      w23 = h2_bits(b * bits_h2(s2));
    } else {
      w01 = h2_bits(a);
      w23 = h2_bits(b);
    }
  }
};

template <>
struct Dequant<f16, W_INT8> {
  // bytes b0 b1 b2 b3 = v0 v2 v1 v3 (k-rows 2m, 2m+8, 2m+1, 2m+9); zero point 128
  static __device__ __forceinline__ void run(uint32_t q, uint32_t s2, bool scaled, uint32_t& w01, uint32_t& w23) {
    const f16x2 a = bits_h2((q & 0x00ff00ffu) | 0x64006400u) - bits_h2(0x64806480u);         // 1024 + b - 1152
    const f16x2 b = bits_h2(((q >> 8) & 0x00ff00ffu) | 0x64006400u) - bits_h2(0x64806480u);
    if (scaled) {
      w01 = h2_bits(a * bits_h2(s2));
      w23 = h2_bits(b * bits_h2(s2));
    } else {
      w01 = h2_bits(a);
      w23 = h2_bits(b);
    }
  }
};

template <>
struct Dequant<f16, W_FP8> {
  // e4m3fn byte -> fp16: move sign, shift exponent/mantissa into place, fix the bias with * 2^8
  // (same construction as fp8/fp8_marlin.cu:132-196)
  static __device__ __forceinline__ uint32_t cvt(uint32_t t) {  // t = 0x00XX00YY
    const uint32_t r = ((t << 8) & 0x80008000u) | ((t << 7) & 0x3f803f80u);
    return h2_bits(bits_h2(r) * bits_h2(0x5c005c00u));  // * 256
  }
  static __device__ __forceinline__ void run(uint32_t q, uint32_t s2, bool scaled, uint32_t& w01, uint32_t& w23) {
    w01 = cvt(q & 0x00ff00ffu);
    w23 = cvt((q >> 8) & 0x00ff00ffu);
    if (scaled) {
      w01 = h2_bits(bits_h2(w01) * bits_h2(s2));
      w23 = h2_bits(bits_h2(w23) * bits_h2(s2));
    }
  }
};

// bf16 has no packed arithmetic on gfx950: go through fp32 (exact integer, one rounding at the end).
// s2 carries the fp32 scale bits for bf16.
template <>
struct Dequant<bf16, W_INT4> {
  static __device__ __forceinline__ void run(uint32_t q, uint32_t sbits, bool scaled, uint32_t& w01, uint32_t& w23) {
    const float s = scaled ? __builtin_bit_cast(float, sbits) : 1.0f;
    const float v0 = (float)(int)(q & 0xf) - 8.f, v1 = (float)(int)((q >> 16) & 0xf) - 8.f;
    const float v2 = (float)(int)((q >> 4) & 0xf) - 8.f, v3 = (float)(int)((q >> 20) & 0xf) - 8.f;
    w01 = pack_bf16(v0 * s, v1 * s);
    w23 = pack_bf16(v2 * s, v3 * s);
  }
};
template <>
struct Dequant<bf16, W_INT8> {
  static __device__ __forceinline__ void run(uint32_t q, uint32_t sbits, bool scaled, uint32_t& w01, uint32_t& w23) {
    const float s = scaled ? __builtin_bit_cast(float, sbits) : 1.0f;
    const float v0 = (float)(int)(q & 0xff) - 128.f, v1 = (float)(int)((q >> 16) & 0xff) - 128.f;
    const float v2 = (float)(int)((q >> 8) & 0xff) - 128.f, v3 = (float)(int)((q >> 24) & 0xff) - 128.f;
    w01 = pack_bf16(v0 * s, v1 * s);
    w23 = pack_bf16(v2 * s, v3 * s);
  }
};
template <>
struct Dequant<bf16, W_FP8> {
  static __device__ __forceinline__ void run(uint32_t q, uint32_t sbits, bool scaled, uint32_t& w01, uint32_t& w23) {
    const float s = scaled ? __builtin_bit_cast(float, sbits) : 1.0f;
    const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(q, false);  // bytes 0,1 = v0, v2
    const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8(q, true);   // bytes 2,3 = v1, v3
    w01 = pack_bf16(lo[0] * s, hi[0] * s);
    w23 = pack_bf16(lo[1] * s, hi[1] * s);
  }
};

struct GemmParams {
  const void* a;           // [M, K]
  const int32_t* b;        // Marlin-packed weight
  const void* scales;      // [num_groups, N] Marlin-permuted
  const int32_t* g_idx;    // [K] or null
  const int32_t* perm;     // [K] or null
  void* c;                 // [M, N] scalar_t
  float* partial;          // [k_splits, M, N] fp32 (k_splits > 1)
  int M, N, K;
  int num_groups, group_size;  // group_size = K for channel-wise
  int k_splits;
  int slow_act_order;      // act-order with partial K: per-row scale lookup
};

// ---- the skinny GEMM kernel -----------------------------------------------------------------------------------
// grid (ceil(N / COLS), k_splits, ceil(M / (16 MT))), block 256 (4 waves)
template <typename scalar_t, int KIND, int MT>
__global__ __launch_bounds__(256) void marlin_skinny_kernel(const GemmParams p) {
  constexpr bool I4 = (KIND == W_INT4);
  constexpr int NTILE = I4 ? 8 : 4;       // MFMA column tiles per wave
  constexpr int COLS = I4 ? 128 : 64;     // output columns per workgroup
  constexpr int ROWS = 16 * MT;
  constexpr int ROW_WORDS_PER_64 = I4 ? 128 : 256;  // int32 per (k-tile, 64-column group)

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4;
  const int li = lane & 15;
  const int c8 = li & 7;
  const int half = li >> 3;

  const int n0 = blockIdx.x * COLS;
  const int m0 = blockIdx.z * ROWS;
  const int N = p.N, K = p.K, M = p.M;

  // per-wave LDS: activation staging [kSubSteps][4 g][ROWS][8 halves] = 2 KiB * MT ... reused for the reduction
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_a = smem + (size_t)wave * (kSubSteps * 4 * ROWS * 16);

  // ---- this wave's range of sub-chunks (128 k each) ----
  const int total_steps = (K + 31) / 32;
  const int total_sub = (total_steps + kSubSteps - 1) / kSubSteps;
  const int nworkers = p.k_splits * 4;
  const int sub_per = (total_sub + nworkers - 1) / nworkers;
  const int worker = blockIdx.y * 4 + wave;
  const int sub_begin = min(worker * sub_per, total_sub);
  const int sub_end = min(sub_begin + sub_per, total_sub);

  // ---- weight addressing ----
  // 4-bit: lane loads 16 B at word offset (64-col group ng + half) * 128 + (4 c8 + g) * 4 of k-tile row kt
  // 8-bit: lane loads 16 B at word offset ng * 256 + (4 c8 + g) * 8 + 4 half
  const int ktiles = K / 16;
  const int64_t row_words = (int64_t)N * 16 / (I4 ? 8 : 4);
  int64_t lane_word;
  bool col_ok;  // this lane's 64-column group exists (N % 128 == 64 tail for 4-bit)
  if constexpr (I4) {
    const int ng = n0 / 64 + half;
    col_ok = (ng * 64) < N;
    lane_word = (int64_t)(col_ok ? ng : n0 / 64) * ROW_WORDS_PER_64 + (4 * c8 + g) * 4;
  } else {
    col_ok = true;
    lane_word = (int64_t)(n0 / 64) * ROW_WORDS_PER_64 + (4 * c8 + g) * 8 + 4 * half;
  }
  const int32_t* bw = p.b + lane_word;

  // ---- scale addressing: lane (., li) needs the scales of its NTILE columns, contiguous in the Marlin-permuted
  // row for grouped scales (position 8 c8 + b, b = hi + 2 j [+ 4 half for 8-bit]) ----
  const bool grouped = p.num_groups > 1;
  const scalar_t* sc = reinterpret_cast<const scalar_t*>(p.scales);
  int64_t scale_off;  // element offset inside a scale row
  if constexpr (I4) scale_off = (int64_t)(n0 / 64 + (col_ok ? half : 0)) * 64 + 8 * c8;
  else scale_off = (int64_t)(n0 / 64) * 64 + 8 * c8 + 4 * half;

  uint32_t s2[NTILE];  // per-column scale operand (f16: packed pair, bf16: fp32 bits)
#pragma unroll
  for (int t = 0; t < NTILE; ++t) s2[t] = 0;
  int cur_group = -1;

  auto load_group_scales = [&](int grp) {
    union { u32x4 v; scalar_t e[8]; } raw;
    if constexpr (I4) {
      raw.v = *reinterpret_cast<const u32x4*>(sc + (int64_t)grp * N + scale_off);
    } else {
      const u32x2 r2 = *reinterpret_cast<const u32x2*>(sc + (int64_t)grp * N + scale_off);
      raw.v = u32x4{r2[0], r2[1], 0, 0};
    }
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      // tile t = (j, hi): b = hi + 2 j  -> element index b
      if constexpr (__is_same(scalar_t, f16)) {
        union { f16 h[2]; uint32_t u; } pk;
        pk.h[0] = raw.e[t];
        pk.h[1] = raw.e[t];
        s2[t] = pk.u;
      } else {
        s2[t] = __builtin_bit_cast(uint32_t, (float)raw.e[t]);
      }
    }
  };

  f32x4 acc[MT][NTILE];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NTILE; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const scalar_t* A = reinterpret_cast<const scalar_t*>(p.a);

  for (int sub = sub_begin; sub < sub_end; ++sub) {
    const int kbase = sub * (kSubSteps * 32);

    // ---- stage activations [ROWS x 128 k] into this wave's LDS region, in MFMA fragment order ----
    // piece = (row, 16-B chunk cc16 of the 128-k slab); 16 chunks per row.
    // chunk (ks = cc16 / 4, cc = cc16 % 4), dword e2 -> fragment (ks, g = e2, row) dword cc
#pragma unroll
    for (int it = 0; it < ROWS * 16 / 64; ++it) {
      const int piece = it * 64 + lane;
      const int row = piece >> 4;
      const int cc16 = piece & 15;
      const int k = kbase + cc16 * 8;
      u32x4 v = {0, 0, 0, 0};
      const int m = m0 + row;
      if (m < M && k < K) {
        if (p.perm == nullptr) {
          v = *reinterpret_cast<const u32x4*>(A + (int64_t)m * K + k);
        } else {
          // act-order: A'[m][k] = A[m][perm[k]] (gptq_marlin.cu:345-394)
          union { scalar_t h[8]; u32x4 u; } gth;
#pragma unroll
          for (int e = 0; e < 8; ++e) gth.h[e] = A[(int64_t)m * K + p.perm[k + e]];
          v = gth.u;
        }
      }
      const int ks = cc16 >> 2, cc = cc16 & 3;
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        *reinterpret_cast<uint32_t*>(lds_a + (((ks * 4 + e2) * ROWS + row) * 16) + 4 * cc) = v[e2];
      }
    }
    // same wave writes and reads: LDS ops of one wave complete in order; make the writes visible to the reads
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();

#pragma unroll
    for (int ksl = 0; ksl < kSubSteps; ++ksl) {
      const int kstep = sub * kSubSteps + ksl;
      if (kstep >= total_steps) break;
      const int kt0 = 2 * kstep;
      const int kt1 = min(kt0 + 1, ktiles - 1);  // K % 32 == 16 tail: activations are zero-filled there
      const u32x4 q0 = *reinterpret_cast<const u32x4*>(bw + (int64_t)kt0 * row_words);
      const u32x4 q1 = *reinterpret_cast<const u32x4*>(bw + (int64_t)kt1 * row_words);

      if (grouped && !p.slow_act_order) {
        const int grp = (kstep * 32) / p.group_size;
        if (grp != cur_group) {
          cur_group = grp;
          load_group_scales(grp);
        }
      }

      u32x4 af[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        af[mt] = *reinterpret_cast<const u32x4*>(lds_a + (((ksl * 4 + g) * ROWS + mt * 16 + li) * 16));

#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        uint32_t w0, w1;
        if constexpr (I4) {
          const int j = t >> 1, hi = t & 1;
          w0 = hi ? (q0[j] >> 8) : q0[j];
          w1 = hi ? (q1[j] >> 8) : q1[j];
        } else {
          w0 = q0[t];  // t = 2 jl + hi : words [jl lo, jl hi] in order
          w1 = q1[t];
        }
        u32x4 wf;
        uint32_t d0, d1, d2, d3;
        if (!p.slow_act_order) {
          Dequant<scalar_t, KIND>::run(w0, s2[t], grouped, d0, d1);
          Dequant<scalar_t, KIND>::run(w1, s2[t], grouped, d2, d3);
          wf = u32x4{d0, d1, d2, d3};
        } else {
          // act-order on a K-shard (is_k_full == false): every k-row carries its own group id
          // (gptq_marlin.cu:965-980 with g_idx). Dequantise unscaled, then scale element-wise.
          Dequant<scalar_t, KIND>::run(w0, 0, false, d0, d1);
          Dequant<scalar_t, KIND>::run(w1, 0, false, d2, d3);
          union { u32x4 u; scalar_t h[8]; } wv;
          wv.u = u32x4{d0, d1, d2, d3};
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const int koff[4] = {0, 1, 8, 9};
            int k = kstep * 32 + 16 * (jj >> 2) + 2 * g + koff[jj & 3];
            k = min(k, K - 1);
            const int grp = p.g_idx[k];
            const float sv = Scalar<scalar_t>::to_f32(sc[(int64_t)grp * N + scale_off + t]);
            wv.h[jj] = Scalar<scalar_t>::from_f32(Scalar<scalar_t>::to_f32(wv.h[jj]) * sv);
          }
          wf = wv.u;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][t] = mfma_16x16x32<scalar_t>(wf, af[mt], acc[mt][t]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- channel-wise scales are applied to the fp32 accumulators (rows of D = column slots 4 g + r) ----
  if (!grouped && !p.slow_act_order) {
    // scale_perm_single (marlin_perms.py:44-47): within a 32-column chunk position 8 (c/2) + (c%2) + 2 b' holds
    // column c + 8 b'. Fetch this lane's column scales in the (., li) layout, then move them to the D-row layout.
    float srow[NTILE];
    {
      const int64_t base64 = I4 ? (int64_t)(n0 / 64 + (col_ok ? half : 0)) * 64 : (int64_t)(n0 / 64) * 64;
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        int b;  // column = c8 + 8 b within the 64 group
        if constexpr (I4) b = (t & 1) + 2 * (t >> 1);
        else b = (t & 1) + 2 * (t >> 1) + 4 * half;
        const int chunk32 = b >> 2, bp = b & 3;
        const int pos = 32 * chunk32 + 8 * (c8 >> 1) + (c8 & 1) + 2 * bp;
        srow[t] = Scalar<scalar_t>::to_f32(sc[base64 + pos]);
      }
    }
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sv = __shfl(srow[t], 4 * g + r, 64);  // scale of column slot 4 g + r (any lane group holds it)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][t][r] *= sv;
      }
    }
  }

  // ---- reduce the 4 waves (tree through LDS), wave 0 writes ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);  // [2][MT][NTILE][64 lanes][4]
  constexpr int ACC_FLOATS = MT * NTILE * 64 * 4;
  if (wave >= 2) {
    float* dst = red + (wave - 2) * ACC_FLOATS;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < NTILE; ++t) *reinterpret_cast<f32x4*>(dst + ((mt * NTILE + t) * 64 + lane) * 4) = acc[mt][t];
  }
  __syncthreads();
  if (wave < 2) {
    const float* src = red + wave * ACC_FLOATS;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < NTILE; ++t) acc[mt][t] += *reinterpret_cast<const f32x4*>(src + ((mt * NTILE + t) * 64 + lane) * 4);
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < NTILE; ++t) *reinterpret_cast<f32x4*>(red + ((mt * NTILE + t) * 64 + lane) * 4) = acc[mt][t];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NTILE; ++t) acc[mt][t] += *reinterpret_cast<const f32x4*>(red + ((mt * NTILE + t) * 64 + lane) * 4);

  // lane (g, li): D rows = column slots 4 g + r (4 consecutive output columns), D col = activation row li
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    int n;
    if constexpr (I4) n = n0 + 64 * (g >> 1) + 4 * (g & 1) + 8 * (t & 1) + 16 * (t >> 1);
    else n = n0 + 32 * (g >> 1) + 4 * (g & 1) + 8 * (t & 1) + 16 * (t >> 1);
    if (n >= N) continue;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + mt * 16 + li;
      if (m >= M) continue;
      if (p.k_splits == 1) {
        union { scalar_t h[4]; u32x2 u; } r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.h[j] = Scalar<scalar_t>::from_f32(acc[mt][t][j]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<scalar_t*>(p.c) + (int64_t)m * N + n) = r.u;
      } else {
        *reinterpret_cast<f32x4*>(p.partial + ((int64_t)blockIdx.y * M + m) * N + n) = acc[mt][t];
      }
    }
  }
}

// out[m][n] = cast(sum_s partial[s][m][n]); 4 columns per thread
template <typename scalar_t>
__global__ void splitk_reduce_kernel(scalar_t* __restrict__ c, const float* __restrict__ partial, int64_t mn4, int splits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mn4) return;
  f32x4 acc = *reinterpret_cast<const f32x4*>(partial + i * 4);
  for (int s = 1; s < splits; ++s) acc += *reinterpret_cast<const f32x4*>(partial + ((int64_t)s * mn4 + i) * 4);
  union { scalar_t h[4]; u32x2 u; } r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r.h[j] = Scalar<scalar_t>::from_f32(acc[j]);
  *reinterpret_cast<u32x2*>(c + i * 4) = r.u;
}

// ---- GPTQ -> Marlin repack (gptq_marlin_repack.cu:32-260; element map SURVEY.md appendix A.2) ----------------
// one thread per output int32
template <int BITS>
__global__ void marlin_repack_kernel(const uint32_t* __restrict__ in, const int32_t* __restrict__ perm,
                                     uint32_t* __restrict__ out, int size_k, int size_n) {
  constexpr int PF = 32 / BITS;
  constexpr int WORDS64 = 1024 / PF;  // words per (k-tile, 64-column group)
  const int64_t row_words = (int64_t)size_n * 16 / PF;
  const int64_t total = (int64_t)(size_k / 16) * row_words;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int kt = idx / row_words;
  const int w = idx % row_words;
  const int ng = w / WORDS64;
  const int wi = w % WORDS64;
  int i, j, part;
  if constexpr (BITS == 4) { i = wi >> 2; j = wi & 3; part = 0; }
  else { i = wi >> 3; j = (wi >> 1) & 3; part = wi & 1; }
  const int col = i >> 2;
  const int row0 = 2 * (i & 3);
  const int rows[4] = {row0, row0 + 1, row0 + 8, row0 + 9};
  auto fetch = [&](int e) -> uint32_t {  // element e of v[0..7]
    int k = kt * 16 + rows[e & 3];
    const int n = ng * 64 + 16 * j + col + 8 * (e >> 2);
    if (perm != nullptr) k = perm[k];
    const uint32_t word = in[(int64_t)(k / PF) * size_n + n];
    return (word >> (BITS * (k % PF))) & ((1u << BITS) - 1);
  };
  uint32_t r = 0;
  if constexpr (BITS == 4) {
    const int il[8] = {0, 2, 4, 6, 1, 3, 5, 7};
#pragma unroll
    for (int pz = 0; pz < 8; ++pz) r |= fetch(il[pz]) << (4 * pz);
  } else {
    const int il[4] = {0, 2, 1, 3};
#pragma unroll
    for (int pz = 0; pz < 4; ++pz) r |= fetch(4 * part + il[pz]) << (8 * pz);
  }
  out[idx] = r;
}

int pick_mt(int M) { return M <= 16 ? 1 : (M <= 32 ? 2 : 4); }

// number of K splits over workgroups: enough workgroups to fill 256 CUs ~2x, at least 1 sub-chunk per wave
int pick_k_splits(int M, int N, int K, int cols, int mt) {
  const int n_tiles = ceil_div(N, cols);
  const int m_blocks = ceil_div(M, 16 * mt);
  const int total_sub = ceil_div(ceil_div(K, 32), kSubSteps);
  const int target_wgs = 512;
  int splits = ceil_div(target_wgs, n_tiles * m_blocks);
  const int max_splits = std::max(1, total_sub / 4);  // >= 1 sub-chunk per wave
  splits = std::max(1, std::min(splits, max_splits));
  return std::min(splits, 64);
}

template <typename scalar_t, int KIND>
int launch_skinny(GemmParams& p, void* scratch, int64_t scratch_bytes, hipStream_t stream) {
  constexpr int COLS = (KIND == W_INT4) ? 128 : 64;
  // rows are processed in passes of <= 64 (weights re-streamed per pass; large M has its own kernel)
  const int mt = pick_mt(p.M);
  p.k_splits = pick_k_splits(p.M, p.N, p.K, COLS, mt);
  if (p.k_splits > 1) {
    const int64_t need = (int64_t)p.k_splits * p.M * p.N * sizeof(float);
    if (scratch == nullptr || scratch_bytes < need) {
      // fall back to fewer splits that fit (never allocate here: graph capture)
      int fit = scratch == nullptr ? 1 : (int)(scratch_bytes / ((int64_t)p.M * p.N * sizeof(float)));
      p.k_splits = std::max(1, std::min(p.k_splits, fit));
    }
  }
  p.partial = reinterpret_cast<float*>(scratch);
  dim3 grid(ceil_div(p.N, COLS), p.k_splits, ceil_div(p.M, 16 * mt));
  constexpr int NTILE = (KIND == W_INT4) ? 8 : 4;
  auto smem_for = [&](int MT) {
    const size_t stage = (size_t)4 * kSubSteps * 4 * (16 * MT) * 16;
    const size_t red = (size_t)2 * MT * NTILE * 64 * 4 * sizeof(float);
    return std::max(stage, red);
  };
  switch (mt) {
    case 1: marlin_skinny_kernel<scalar_t, KIND, 1><<<grid, 256, smem_for(1), stream>>>(p); break;
    case 2: marlin_skinny_kernel<scalar_t, KIND, 2><<<grid, 256, smem_for(2), stream>>>(p); break;
    default: {
      auto kern = marlin_skinny_kernel<scalar_t, KIND, 4>;
      const size_t sm = smem_for(4);
      if (sm > 64 * 1024) NMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
      kern<<<grid, 256, sm, stream>>>(p);
    }
  }
  NMX_LAUNCH_CHECK();
  if (p.k_splits > 1) {
    const int64_t mn4 = (int64_t)p.M * p.N / 4;
    splitk_reduce_kernel<scalar_t><<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>(
        reinterpret_cast<scalar_t*>(p.c), p.partial, mn4, p.k_splits);
    NMX_LAUNCH_CHECK();
  }
  return NMX_OK;
}

int marlin_common(const void* a, const int32_t* b_q_weight, const void* b_scales, const int32_t* g_idx,
                  const int32_t* perm, void* c, int64_t workspace_numel, void* scratch, int64_t scratch_bytes,
                  int size_m, int size_n, int size_k, int kind, int num_groups, int is_k_full, int dtype,
                  hipStream_t stream) {
  // checks mirror gptq_marlin.cu:1741-1843
  NMX_CHECK(size_k % 16 == 0, NMX_ERR_INVALID_ARG, "size_k = %d is not divisible by tile_size = 16", size_k);
  NMX_CHECK(size_n % 64 == 0, NMX_ERR_INVALID_ARG, "size_n = %d is not divisible by min_thread_n = 64", size_n);
  NMX_CHECK(workspace_numel >= (int64_t)(size_n / 64) * 16, NMX_ERR_INVALID_ARG,
            "workspace.numel = %lld is below min_workspace_size = %d", (long long)workspace_numel, (size_n / 64) * 16);
  NMX_CHECK(dtype == NMX_F16 || dtype == NMX_BF16, NMX_ERR_UNSUPPORTED, "gpt_marlin_gemm only supports bfloat16 and float16");
  NMX_CHECK((g_idx == nullptr) == (perm == nullptr), NMX_ERR_INVALID_ARG, "g_idx and perm must both be given or both be empty");
  NMX_CHECK(num_groups >= 1, NMX_ERR_INVALID_ARG, "b_scales must have at least one row");
  const bool has_act_order = g_idx != nullptr;
  GemmParams p;
  p.a = a; p.b = b_q_weight; p.scales = b_scales; p.g_idx = g_idx; p.perm = perm; p.c = c; p.partial = nullptr;
  p.M = size_m; p.N = size_n; p.K = size_k; p.num_groups = num_groups; p.k_splits = 1; p.slow_act_order = 0;
  if (has_act_order) {
    if (is_k_full) {
      NMX_CHECK(num_groups > 1, NMX_ERR_INVALID_ARG, "For act_order, num_groups must be > 1");
      NMX_CHECK(size_k % num_groups == 0, NMX_ERR_INVALID_ARG, "size_k = %d, is not divisible by num_groups = %d", size_k, num_groups);
      p.group_size = size_k / num_groups;  // sorted g_idx + full K: groups are contiguous runs of group_size rows
    } else {
      p.group_size = 0;
      p.slow_act_order = 1;
    }
  } else if (num_groups > 1) {
    NMX_CHECK(size_k % num_groups == 0, NMX_ERR_INVALID_ARG, "size_k = %d, is not divisible by b_scales.size(0) = %d", size_k, num_groups);
    p.group_size = size_k / num_groups;
  } else {
    p.group_size = size_k;
  }
  if (num_groups > 1 && !p.slow_act_order)
    NMX_CHECK(p.group_size % 32 == 0, NMX_ERR_UNSUPPORTED, "group_size = %d must be a multiple of 32", p.group_size);
  NMX_CHECK(((uintptr_t)a % 16 == 0) && ((uintptr_t)b_q_weight % 16 == 0) && ((uintptr_t)b_scales % 16 == 0) &&
                ((uintptr_t)c % 8 == 0) && size_k % 8 == 0,
            NMX_ERR_INVALID_ARG, "marlin gemm: operands must be 16-byte aligned");
  if (size_m == 0 || size_n == 0) return NMX_OK;

#define NMX_DISPATCH_KIND(T)                                                                   \
  switch (kind) {                                                                              \
    case W_INT4: return launch_skinny<T, W_INT4>(p, scratch, scratch_bytes, stream);           \
    case W_INT8: return launch_skinny<T, W_INT8>(p, scratch, scratch_bytes, stream);           \
    default: return launch_skinny<T, W_FP8>(p, scratch, scratch_bytes, stream);                \
  }
  if (dtype == NMX_F16) { NMX_DISPATCH_KIND(f16) }
  else { NMX_DISPATCH_KIND(bf16) }
#undef NMX_DISPATCH_KIND
}

}  // namespace

extern "C" int64_t nmx_marlin_gemm_scratch_bytes(int size_m, int size_n, int size_k) {
  if (size_m <= 0 || size_n <= 0 || size_k <= 0) return 0;
  const int mt = pick_mt(size_m);
  const int splits = std::max(pick_k_splits(size_m, size_n, size_k, 128, mt), pick_k_splits(size_m, size_n, size_k, 64, mt));
  return splits > 1 ? (int64_t)splits * size_m * size_n * sizeof(float) : 0;
}

extern "C" int nmx_gptq_marlin_repack(const int32_t* b_q_weight, const int32_t* perm, int32_t* out, int size_k,
                                      int size_n, int num_bits, nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  // checks mirror gptq_marlin_repack.cu:281-300
  NMX_CHECK(num_bits == 4 || num_bits == 8, NMX_ERR_INVALID_ARG, "num_bits must be 4 or 8. Got = %d", num_bits);
  NMX_CHECK(size_k % 16 == 0, NMX_ERR_INVALID_ARG, "size_k = %d is not divisible by tile_k_size = 16", size_k);
  NMX_CHECK(size_n % 64 == 0, NMX_ERR_INVALID_ARG, "size_n = %d is not divisible by tile_n_size = 64", size_n);
  const int pf = 32 / num_bits;
  const int64_t total = (int64_t)(size_k / 16) * ((int64_t)size_n * 16 / pf);
  if (total == 0) return NMX_OK;
  const unsigned blocks = (unsigned)ceil_div64(total, 256);
  if (num_bits == 4)
    marlin_repack_kernel<4><<<blocks, 256, 0, stream>>>((const uint32_t*)b_q_weight, perm, (uint32_t*)out, size_k, size_n);
  else
    marlin_repack_kernel<8><<<blocks, 256, 0, stream>>>((const uint32_t*)b_q_weight, perm, (uint32_t*)out, size_k, size_n);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int nmx_gptq_marlin_gemm(const void* a, const int32_t* b_q_weight, const void* b_scales,
                                    const int32_t* g_idx, const int32_t* perm, void* c, int64_t workspace_numel,
                                    void* scratch, int64_t scratch_bytes, int size_m, int size_n, int size_k,
                                    int num_bits, int num_groups, int is_k_full, int dtype, nmx_stream_t stream) {
  NMX_CHECK(num_bits == 4 || num_bits == 8, NMX_ERR_INVALID_ARG, "num_bits must be 4 or 8. Got = %d", num_bits);
  return marlin_common(a, b_q_weight, b_scales, g_idx, perm, c, workspace_numel, scratch, scratch_bytes, size_m,
                       size_n, size_k, num_bits == 4 ? W_INT4 : W_INT8, num_groups, is_k_full, dtype,
                       (hipStream_t)stream);
}

extern "C" int nmx_marlin_gemm(const void* a, const int32_t* b_q_weight, const void* b_scales, void* c,
                               int64_t workspace_numel, void* scratch, int64_t scratch_bytes, int size_m,
                               int size_n, int size_k, int num_groups, nmx_stream_t stream) {
  // marlin_cuda_kernel.cu:1045-1136: groupsize must be -1 or 128
  if (num_groups > 1) {
    NMX_CHECK(size_k % num_groups == 0 && size_k / num_groups == 128, NMX_ERR_INVALID_ARG,
              "Unexpected groupsize = %d", num_groups ? size_k / num_groups : -1);
  }
  return marlin_common(a, b_q_weight, b_scales, nullptr, nullptr, c, workspace_numel, scratch, scratch_bytes, size_m,
                       size_n, size_k, W_INT4, num_groups, 1, NMX_F16, (hipStream_t)stream);
}

extern "C" int nmx_fp8_marlin_gemm(const void* a, const int32_t* b_q_weight, const void* b_scales, void* c,
                                   int64_t workspace_numel, void* scratch, int64_t scratch_bytes, int num_bits,
                                   int size_m, int size_n, int size_k, int dtype, nmx_stream_t stream) {
  NMX_CHECK(num_bits == 8, NMX_ERR_INVALID_ARG, "num_bits must be 8 for fp8 marlin. Got = %d", num_bits);
  return marlin_common(a, b_q_weight, b_scales, nullptr, nullptr, c, workspace_numel, scratch, scratch_bytes, size_m,
                       size_n, size_k, W_FP8, 1, 1, dtype, (hipStream_t)stream);
}
