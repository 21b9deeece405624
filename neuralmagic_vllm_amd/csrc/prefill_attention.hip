// Prefill attention over a paged context plus the new tokens of the same sequence (prefix caching / chunked prefill).
// Replaces vllm/attention/ops/prefix_prefill.py (Triton `context_attention_fwd`, :674-812) of the reference.
//
// Query token i of sequence b (position ctx_len + i) attends to the ctx_len cached tokens of b - read through the
// block table from the paged KV cache, K [NB, Hkv, D/8, BS, 8], V [NB, Hkv, D, BS] - and causally to the new tokens
// j <= i, read from the k / v tensors of this step. Same orientation as the decode kernel: S^T = K . Q^T and
// O^T = V^T . P^T on v_mfma_f32_16x16x32, so the query row sits on lane & 15, the softmax statistics are per-lane
// scalars and P is re-shaped with two permlane swaps.
//   * cached K, cached V and new K are MFMA A-operand fragments as stored (16-byte loads, no LDS);
//   * new V is [token][d] in memory but the product contracts over tokens: each wave stages its 32-token tile
//     row-major in LDS (ds_write_b128) and reads it back with the gfx950 transposing LDS read
//     (ds_read_b64_tr_b16), which hands every lane 4 tokens of its own d column per instruction.
// One wave = 16 consecutive query tokens of one or two heads; a workgroup = 4 such waves (64 tokens).
// Two kernels: prefill_attention_shared_kernel (head sizes that are a multiple of 64, >= 64 new tokens) stages every
// K / V tile once per workgroup in LDS for its four waves; prefill_attention_kernel (everything else: short chunks,
// odd head sizes) lets every wave load its own fragments and has no workgroup barrier.
// The softmax is VALU work beside 16 MFMAs per 32 keys, so it is kept to ~5 instructions per logit: a mask-free path
// for tiles every row sees completely, exp2 with the scale folded into one fma, row reductions with
// v_permlane{16,32}_swap, and a lazy running maximum (softmax_pv).
// grid (num_heads [/ 2], batch, ceil(max_input_len / 64)), block 256.
//
// Compute-bound: 4 * (ctx + (i + 1)) * D flop per (query token, head); algorithmic bytes = q + out + the KV it reads.
#include <float.h>
#include <stdlib.h>

#include <type_traits>

#include "nmx_common.h"

namespace {

struct PrefillParams {
  void* out;
  const void* q;
  const void* k;
  const void* v;
  const void* k_cache;
  const void* v_cache;
  const int32_t* b_loc;
  const int32_t* b_start_loc;
  const int32_t* b_seq_len;
  const int32_t* b_ctx_len;
  const float* alibi_slopes;
  int64_t q_st, q_sh, k_st, k_sh, v_st, v_sh, o_st, o_sh, kc_sb, kc_sh, vc_sb, vc_sh, bloc_stride;
  int num_heads, num_kv_heads, block_size, bs_shift, sliding_window;
  float scale;
};

template <typename scalar_t>
__device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (__is_same(scalar_t, f16))
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));

// two floats -> one packed register, round-to-nearest-even (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32: one instruction)
template <typename scalar_t>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  const f32x2v v = {lo, hi};
  if constexpr (__is_same(scalar_t, f16)) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2v));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
}

// max of three without the quieting self-max hipcc puts in front of every fmaxf operand (the operands are MFMA
// results and running maxima: never signalling NaNs)
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// P from the MFMA C/D layout (lane (g, q): tokens 4g..4g+3 of two 16-token sub-tiles) to the B-operand layout
// (lane (g, q): tokens 8g..8g+7 of the 32-token tile)
__device__ __forceinline__ u32x4 p_to_operand(u32x2 a, u32x2 b) {
  u32x4 r;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    auto s1 = __builtin_amdgcn_permlane32_swap(a[d], b[d], false, false);
    auto s2 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
    r[d] = s2[0];
    r[2 + d] = s2[1];
  }
  return r;
}

typedef short s16x4 __attribute__((ext_vector_type(4)));

// reductions over the four 16-lane rows of a wave (the lanes that hold the same query row), in registers:
// v_permlane32_swap / v_permlane16_swap of a value with itself leave lane l with l ^ 32 resp. l ^ 16's copy
// in the second result (an LDS ds_bpermute round trip per step was on the critical path of every tile)
__device__ __forceinline__ float rows_max(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = max2(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return max2(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

constexpr float NEG_BIG = -1e30f;  // "minus infinity" of the running maxima and of masked logits (finite: no inf - inf)

struct RowCtx {
  int qpos, g, W;
  bool row_ok, has_alibi;
  float scale;
};

// One 32-key tile of one head: logits s = K.Q^T (raw), key positions kpos0 + 16 u + 4 g + r, V fragments vf.
// Online softmax with a LAZY running maximum: the accumulators are rescaled only when some row's maximum grows by more
// than 5 (e^5 = 148: P stays far inside fp16 / bf16 range; the final O / l is unchanged because l carries the same stale
// maximum) - in steady state that removes the NT x 4 accumulator multiplies of every tile.
// `fast` (wave-uniform): no alibi, no window, every key of the tile visible to every row - no masks at all, the scale
// folded into one fma per logit: exp2(s * scale * log2e - m * log2e).
template <typename scalar_t, int NT>
__device__ __forceinline__ void softmax_pv(f32x4 (&s)[2], float& m_run, float& l_part, f32x4 (&o)[NT], const u32x4 (&vf)[NT],
                                           const RowCtx& rc, float slope_h, int kpos0, int limit, bool causal, bool fast) {
  constexpr float LOG2E = 1.4426950408889634f;
  // The accumulators o are touched in exactly one place each (the rescale and the MFMAs below), outside the
  // fast / general split: with o written on several control-flow paths hipcc shuffled all NT x 4 of them through
  // copies around every tile.
  float m_tile;
  if (fast) {
    const float mx = max2(max3(max3(s[0][0], s[0][1], s[0][2]), s[0][3], s[1][0]), max3(s[1][1], s[1][2], s[1][3]));
    m_tile = rows_max(mx) * rc.scale;
  } else {
    // a masked logit becomes NEG_BIG: exp2 of it is exactly 0 against any finite running maximum (a row with no
    // visible key yet - only rows past the sequence - gets finite garbage that is never stored)
    m_tile = NEG_BIG;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kpos = kpos0 + 16 * u + 4 * rc.g + r;
        float val = s[u][r] * rc.scale;
        if (rc.W > 0 && rc.qpos - kpos >= rc.W) val = -10000.f;  // prefix_prefill.py:88-104, :201-204
        if (rc.has_alibi) val += slope_h * (float)(kpos - rc.qpos);  // :552-557
        const bool masked = kpos >= limit || (causal && kpos > rc.qpos) || !rc.row_ok;
        val = masked ? NEG_BIG : val;
        s[u][r] = val;
        m_tile = fmaxf(m_tile, val);
      }
    }
    m_tile = rows_max(m_tile);
  }
  if (__any(m_tile - m_run > 5.f)) {
    const float m_new = fmaxf(m_run, m_tile);
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    l_part *= alpha;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] *= alpha;
  }
  float e[2][4];
  // fast: s is the raw logit, exp2(s * scale * log2e - m * log2e); general: s already scaled / biased, c = log2e
  const float c = (fast ? rc.scale : 1.f) * LOG2E, mc = -m_run * LOG2E;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) e[u][r] = __builtin_amdgcn_exp2f(fmaf(s[u][r], c, mc));
  u32x2 pk[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    l_part += (e[u][0] + e[u][1]) + (e[u][2] + e[u][3]);
    pk[u][0] = pack2<scalar_t>(e[u][0], e[u][1]);
    pk[u][1] = pack2<scalar_t>(e[u][2], e[u][3]);
  }
  const u32x4 pb = p_to_operand(pk[0], pk[1]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) o[nt] = mfma<scalar_t>(vf[nt], pb, o[nt]);
}


// GQ = query heads of one kv head processed by a wave (1 or 2): the K / V fragments of a tile are loaded once and
// feed GQ x as many MFMAs - the kernel is bound by the wave-instruction rate of its operand loads, not by the MFMAs.
template <typename scalar_t, int D, int GQ>
__global__ __launch_bounds__(256) void prefill_attention_kernel(const PrefillParams p) {
  constexpr int KS = (D + 31) / 32;
  constexpr int NT = D / 16;
  constexpr int CHUNKS = D / 8;
  constexpr int VROW = D * 2 + 16;  // bytes per staged V row (+16: rows start on different banks)
  // grid (head groups, batch, row blocks), heaviest row blocks first (see prefill_attention_shared_kernel)
  const int b = blockIdx.y, head0 = blockIdx.x * GQ;  // heads head0 .. head0 + GQ - 1 share one kv head
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int ctx = p.b_ctx_len[b];
  const int q_len = p.b_seq_len[b] - ctx;
  const int start = p.b_start_loc[b];
  const int r0 = ((gridDim.z - 1 - blockIdx.z) * 4 + wave) * 16;
  if (r0 >= q_len) return;  // whole wave: EXEC stays full for the transposing reads of the active waves
  const int kvh = head0 / (p.num_heads / p.num_kv_heads);
  const int row = r0 + li;
  const bool row_ok = row < q_len;
  const int qpos = ctx + row;
  float slope[GQ];
#pragma unroll
  for (int h = 0; h < GQ; ++h) slope[h] = p.alibi_slopes != nullptr ? p.alibi_slopes[head0 + h] : 0.f;
  const bool has_alibi = p.alibi_slopes != nullptr;
  const int W = p.sliding_window;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* vs = smem + (size_t)wave * 32 * VROW;  // this wave's [32 tokens][D] tile

  const scalar_t* Q = reinterpret_cast<const scalar_t*>(p.q);
  u32x4 qf[GQ][KS];
#pragma unroll
  for (int h = 0; h < GQ; ++h) {
    const scalar_t* qp = Q + (int64_t)(start + min(row, q_len - 1)) * p.q_st + (int64_t)(head0 + h) * p.q_sh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int chunk = 4 * ks + g;
      u32x4 val = {0, 0, 0, 0};
      if (chunk < CHUNKS) val = *reinterpret_cast<const u32x4*>(qp + chunk * 8);
      qf[h][ks] = val;
    }
  }

  float m_runs[GQ], l_parts[GQ];
  f32x4 os[GQ][NT];
#pragma unroll
  for (int h = 0; h < GQ; ++h) {
    m_runs[h] = NEG_BIG;
    l_parts[h] = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) os[h][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const RowCtx rc{qpos, g, W, row_ok, has_alibi, p.scale};
  // one 32-token tile of head h (softmax_pv above); the fast path needs every row valid and every key visible
  const bool plain = !has_alibi && W == 0 && r0 + 16 <= q_len;
  auto tile_pv = [&](auto hc, f32x4 (&s)[2], int kpos0, int limit, bool causal, const u32x4 (&vf)[NT], bool fast) {
    constexpr int h = decltype(hc)::value;
    softmax_pv<scalar_t, NT>(s, m_runs[h], l_parts[h], os[h], vf, rc, slope[h], kpos0, limit, causal, fast);
  };

  // Both phases request the next tile's K fragments (and, for the new tokens, the next V tile) before the MFMAs and
  // the softmax of the current one: a wave's tile is otherwise two dependent memory round trips (block table -> K,
  // or V -> LDS -> transposed read) followed by ~16 MFMAs, and the kernel ran at the latency, not the MFMA, rate.

  // ---- phase 1: the cached context (no causal mask: every context token precedes every query token) ----
  {
    const int32_t* bt = p.b_loc + (int64_t)b * p.bloc_stride;
    const scalar_t* kc = reinterpret_cast<const scalar_t*>(p.k_cache) + (int64_t)kvh * p.kc_sh;
    const scalar_t* vc = reinterpret_cast<const scalar_t*>(p.v_cache) + (int64_t)kvh * p.vc_sh;
    const int BS = p.block_size, last = ctx - 1;
    auto load_k = [&](int t0, u32x4 (&kf)[2][KS], int64_t& vphys) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int tok = min(t0 + 16 * u + li, last);
        const scalar_t* kb = kc + (int64_t)bt[tok >> p.bs_shift] * p.kc_sb;
        const int off = tok & (BS - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int chunk = 4 * ks + g;
          u32x4 val = {0, 0, 0, 0};
          if (chunk < CHUNKS) val = *reinterpret_cast<const u32x4*>(kb + ((int64_t)chunk * BS + off) * 8);
          kf[u][ks] = val;
        }
      }
      vphys = bt[min(t0 + 8 * g, last & ~7) >> p.bs_shift];
    };
    u32x4 kf[2][KS];
    int64_t vphys = 0;
    if (ctx > 0) load_k(0, kf, vphys);
    for (int t0 = 0; t0 < ctx; t0 += 32) {
      const bool more = t0 + 32 < ctx;
      u32x4 vf[NT];
      {
        const int tokv = t0 + 8 * g;
        const int tokc = min(tokv, last & ~7);
        const scalar_t* vb = vc + vphys * p.vc_sb + (tokc & (BS - 1));
        const int nvalid = max(0, min(8, ctx - tokv));  // slots past the context may hold anything (NaNs included)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          u32x4 val = *reinterpret_cast<const u32x4*>(vb + (int64_t)(16 * nt + li) * BS);
          if (t0 + 32 > ctx) {
#pragma unroll
            for (int dw = 0; dw < 4; ++dw)
              val[dw] &= (nvalid >= 2 * dw + 2) ? 0xffffffffu : ((nvalid == 2 * dw + 1) ? 0x0000ffffu : 0u);
          }
          vf[nt] = val;
        }
      }
      f32x4 s[GQ][2];
#pragma unroll
      for (int h = 0; h < GQ; ++h)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          s[h][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) s[h][u] = mfma<scalar_t>(kf[u][ks], qf[h][ks], s[h][u]);
        }
      if (more) load_k(t0 + 32, kf, vphys);  // into the registers the MFMAs above have just consumed
      const bool fast = plain && t0 + 32 <= ctx;
      tile_pv(std::integral_constant<int, 0>{}, s[0], t0, ctx, false, vf, fast);
      if constexpr (GQ > 1) tile_pv(std::integral_constant<int, GQ - 1>{}, s[GQ - 1], t0, ctx, false, vf, fast);
    }
  }

  // ---- phase 2: the new tokens, causal ----
  {
    const scalar_t* Kn = reinterpret_cast<const scalar_t*>(p.k) + (int64_t)kvh * p.k_sh;
    const scalar_t* Vn = reinterpret_cast<const scalar_t*>(p.v) + (int64_t)kvh * p.v_sh;
    const int n_end = min(q_len, r0 + 16);
    constexpr int VP = (32 * CHUNKS) / 64;  // 16-byte pieces of a V tile per lane
    auto load_kv = [&](int t0, u32x4 (&kf)[2][KS], u32x4 (&vr)[VP]) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int tok = min(t0 + 16 * u + li, q_len - 1);
        const scalar_t* kb = Kn + (int64_t)(start + tok) * p.k_st;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int chunk = 4 * ks + g;
          u32x4 val = {0, 0, 0, 0};
          if (chunk < CHUNKS) val = *reinterpret_cast<const u32x4*>(kb + chunk * 8);
          kf[u][ks] = val;
        }
      }
#pragma unroll
      for (int it = 0; it < VP; ++it) {
        const int piece = it * 64 + lane;
        const int tok = piece / CHUNKS, ch = piece % CHUNKS;
        vr[it] = *reinterpret_cast<const u32x4*>(Vn + (int64_t)(start + min(t0 + tok, q_len - 1)) * p.v_st + ch * 8);
      }
    };
    u32x4 kf[2][KS];
    u32x4 vr[VP];
    if (n_end > 0) load_kv(0, kf, vr);
    for (int t0 = 0; t0 < n_end; t0 += 32) {
      const bool more = t0 + 32 < n_end;
      // stage V[t0 .. t0+31][0 .. D) row-major
#pragma unroll
      for (int it = 0; it < VP; ++it) {
        const int piece = it * 64 + lane;
        // tokens past the sequence as zeros (blanked here, not at the load: a select on the loaded value would wait
        // for it one tile early)
        *reinterpret_cast<u32x4*>(vs + (piece / CHUNKS) * VROW + (piece % CHUNKS) * 16) =
            (t0 + piece / CHUNKS < q_len) ? vr[it] : u32x4{0, 0, 0, 0};
      }
      f32x4 s[GQ][2];
#pragma unroll
      for (int h = 0; h < GQ; ++h)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          s[h][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) s[h][u] = mfma<scalar_t>(kf[u][ks], qf[h][ks], s[h][u]);
        }
      if (more) load_kv(t0 + 32, kf, vr);  // into the registers the LDS writes and MFMAs above have just consumed
      __builtin_amdgcn_wave_barrier();  // the tile above is this wave's own; LDS operations of a wave stay in order
      // V^T fragments: lane i of a 16-lane group supplies the address of row (i >> 2), columns 4 (i & 3) .. + 3 of a
      // 4 x 16 block and receives column i of its 4 rows; two blocks = tokens 8g .. 8g + 7 of d column 16 nt + i
      u32x4 vf[NT];
      {
        const char* base = vs + (8 * g + (li >> 2)) * VROW + (li & 3) * 8;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(base + nt * 32));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(base + 4 * VROW + nt * 32));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          vf[nt] = u32x4{l2[0], l2[1], h2[0], h2[1]};
        }
      }
      const bool fast = plain && t0 + 31 <= r0;
      tile_pv(std::integral_constant<int, 0>{}, s[0], ctx + t0, ctx + q_len, true, vf, fast);
      if constexpr (GQ > 1) tile_pv(std::integral_constant<int, GQ - 1>{}, s[GQ - 1], ctx + t0, ctx + q_len, true, vf, fast);
      __builtin_amdgcn_wave_barrier();
    }
  }

  // ---- out[row][d] = O / l ; lane (g, q) holds O^T[16 nt + 4 g + r][q] ----
#pragma unroll
  for (int h = 0; h < GQ; ++h) {
    const float l_part = rows_sum(l_parts[h]);
    if (!row_ok) continue;
    const float inv = 1.f / l_part;
    scalar_t* op = reinterpret_cast<scalar_t*>(p.out) + (int64_t)(start + row) * p.o_st + (int64_t)(head0 + h) * p.o_sh;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      union { scalar_t e[4]; u32x2 u; } r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r.e[j] = Scalar<scalar_t>::from_f32(os[h][nt][j] * inv);
      *reinterpret_cast<u32x2*>(op + 16 * nt + 4 * g) = r.u;
    }
  }
}


template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Workgroup-shared variant (D a multiple of 64): the four waves of a workgroup work on 64 consecutive query tokens of
// the same kv head, so every 32-token K / V tile is fetched from HBM/L2 ONCE per workgroup - each thread brings
// D / 64 16-byte pieces of K and of V - and written to LDS in MFMA-fragment order (K, cached V) or row-major (new V, read
// back through ds_read_b64_tr_b16); the waves then read their operands with lane-linear ds_read_b128. The tile images
// are double-buffered: the global loads of tile t + 1 are issued before the MFMAs of tile t and written to the other
// buffer after them; one workgroup barrier per tile. Per 32-token tile a wave now issues 2 x D/64 global loads instead
// of 2 x D/8 (the per-wave kernel above is bound by exactly that instruction stream), and GQ can grow to 4 because the
// prefetch registers are gone. Context tiles and new-token tiles form one tile sequence.
template <typename scalar_t, int D, int GQ>
__global__ __launch_bounds__(256) void prefill_attention_shared_kernel(const PrefillParams p) {
  static_assert(D % 64 == 0, "tile pieces must divide evenly over 256 threads");
  constexpr int KS = D / 32;
  constexpr int NT = D / 16;
  constexpr int CHUNKS = D / 8;
  constexpr int PP = CHUNKS / 8;             // 16-byte pieces of K (and of V) per thread per tile
  constexpr int KROW = D * 2 + 32;           // bytes per staged new-K row: ds_read_b128 fragment reads conflict-free
  constexpr int VROW = D * 2 + 16;           // bytes per staged new-V row (+16: rows start on different banks)
  constexpr int KIMG = 32 * KROW;            // K tile image, [key][KROW]
  constexpr int BUF = KIMG + 32 * VROW;      // one buffer: K image, then the V image
  // grid (head groups, batch, row blocks): workgroups are dispatched in linear-id order, so the row blocks - whose causal
  // work grows with their index - are the slowest dimension and run heaviest first (longest-processing-time order: the
  // tail of the launch is made of the 2-tile blocks, not of a 32-tile block that started last); consecutive ids are
  // the heads / sequences of one row block, which spreads every weight class evenly over the 8 XCDs.
  const int b = blockIdx.y, head0 = blockIdx.x * GQ;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int ctx = p.b_ctx_len[b];
  const int q_len = p.b_seq_len[b] - ctx;
  const int start = p.b_start_loc[b];
  const int xb = gridDim.z - 1 - blockIdx.z;
  const int wg_r0 = xb * 64;
  if (wg_r0 >= q_len) return;  // whole workgroup
  const int r0 = wg_r0 + wave * 16;
  const bool wave_on = r0 < q_len;  // idle waves still load tiles and meet the barriers
  const int kvh = head0 / (p.num_heads / p.num_kv_heads);
  const int row = r0 + li;
  const bool row_ok = row < q_len;
  const int qpos = ctx + row;
  float slope[GQ];
#pragma unroll
  for (int h = 0; h < GQ; ++h) slope[h] = p.alibi_slopes != nullptr ? p.alibi_slopes[head0 + h] : 0.f;
  const bool has_alibi = p.alibi_slopes != nullptr;
  const int W = p.sliding_window;
  const RowCtx rc{qpos, g, W, row_ok, has_alibi, p.scale};
  const bool plain = !has_alibi && W == 0 && r0 + 16 <= q_len;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const scalar_t* Q = reinterpret_cast<const scalar_t*>(p.q);
  u32x4 qf[GQ][KS];
#pragma unroll
  for (int h = 0; h < GQ; ++h) {
    const scalar_t* qp = Q + (int64_t)(start + min(row, q_len - 1)) * p.q_st + (int64_t)(head0 + h) * p.q_sh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[h][ks] = *reinterpret_cast<const u32x4*>(qp + (4 * ks + g) * 8);
  }

  float m_runs[GQ], l_parts[GQ];
  f32x4 os[GQ][NT];
#pragma unroll
  for (int h = 0; h < GQ; ++h) {
    m_runs[h] = NEG_BIG;
    l_parts[h] = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) os[h][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int32_t* bt = p.b_loc + (int64_t)b * p.bloc_stride;
  const scalar_t* kc = reinterpret_cast<const scalar_t*>(p.k_cache) + (int64_t)kvh * p.kc_sh;
  const scalar_t* vc = reinterpret_cast<const scalar_t*>(p.v_cache) + (int64_t)kvh * p.vc_sh;
  const int BS = p.block_size, last = ctx - 1;
  const int n1 = (ctx + 31) >> 5;                           // context tiles
  const int n2 = (min(q_len, wg_r0 + 64) + 31) >> 5;        // new-token tiles any wave of this workgroup needs
  const int n_end = min(q_len, r0 + 16);                    // this wave's causal horizon among the new tokens

  // New tokens: buffer loads over exactly this sequence's rows of this kv head - a token past the sequence is out of
  // range and reads as zeros, so no clamp and no select; the offset of a piece advances by a constant per tile.
  // (The host routes here only when the byte extents fit 32 bits.)
  const int k_row_bytes = (int)p.k_st * 2, v_row_bytes = (int)p.v_st * 2;
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<scalar_t*>(reinterpret_cast<const scalar_t*>(p.k) + (int64_t)kvh * p.k_sh + (int64_t)start * p.k_st), 0,
      (q_len - 1) * k_row_bytes + D * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<scalar_t*>(reinterpret_cast<const scalar_t*>(p.v) + (int64_t)kvh * p.v_sh + (int64_t)start * p.v_st), 0,
      (q_len - 1) * v_row_bytes + D * 2, 0x00020000);
  int nk_off[PP], nv_off[PP];    // byte offsets of this thread's pieces in tile 0 of the new tokens
  int nk_lds[PP], nv_lds[PP];    // where they go in a buffer (row-major images)
  int ck_lds[PP], cv_lds[PP];    // cached tiles: fragment-order images
#pragma unroll
  for (int it = 0; it < PP; ++it) {
    const int piece = it * 256 + tid;
    const int tok = piece / CHUNKS, ch = piece % CHUNKS;
    nk_off[it] = tok * k_row_bytes + ch * 16;
    nv_off[it] = tok * v_row_bytes + ch * 16;
    nk_lds[it] = tok * KROW + ch * 16;
    nv_lds[it] = KIMG + tok * VROW + ch * 16;
    // cached K piece (key = piece % 32, chunk = piece / 32): neighbouring lanes read neighbouring block offsets; same
    // row-major image as the new tokens' K
    const int key = piece & 31, chunk = piece >> 5;
    ck_lds[it] = key * KROW + chunk * 16;
    // cached V piece (token group gg = piece % 4, d = piece / 4); image: fragment nt is 1 KiB, row i = d % 16 holds
    // its four groups in slots 4 i + (gg ^ ((i >> 2) & 2)): eight neighbouring lanes write 128 contiguous bytes and
    // every 16-lane group of the fragment read lands on 16 different slots
    const int gg = piece & 3, d = piece >> 2;
    cv_lds[it] = KIMG + (d >> 4) * 1024 + ((d & 15) * 4 + (gg ^ (((d & 15) >> 2) & 2))) * 16;
  }

  // Nothing in load_tile may consume a loaded value (a select on it would stall the wave for the full memory latency
  // at the top of every tile): the block-table entries arrive one tile ahead (load_bt) and the slots past the context
  // are blanked in store_tile, after the MFMAs of the current tile.
  auto load_bt = [&](int t, int (&kphys)[PP], int (&vphys)[PP]) {
    if (t < n1) {
      const int t0 = t * 32;
#pragma unroll
      for (int it = 0; it < PP; ++it) {
        const int piece = it * 256 + tid;
        kphys[it] = bt[min(t0 + (piece & 31), last) >> p.bs_shift];
        vphys[it] = bt[min(t0 + 8 * (piece & 3), last & ~7) >> p.bs_shift];
      }
    }
  };
  auto load_tile = [&](int t, u32x4 (&kr)[PP], u32x4 (&vr)[PP], const int (&kphys)[PP], const int (&vphys)[PP]) {
    if (t < n1) {
      const int t0 = t * 32;
#pragma unroll
      for (int it = 0; it < PP; ++it) {
        const int piece = it * 256 + tid;
        const int key = piece & 31, chunk = piece >> 5;
        const int tok = min(t0 + key, last);
        const scalar_t* kb = kc + (int64_t)kphys[it] * p.kc_sb;
        kr[it] = *reinterpret_cast<const u32x4*>(kb + ((int64_t)chunk * BS + (tok & (BS - 1))) * 8);
        const int gg = piece & 3, d = piece >> 2;
        const int tokc = min(t0 + 8 * gg, last & ~7);
        const scalar_t* vb = vc + (int64_t)vphys[it] * p.vc_sb + (tokc & (BS - 1));
        vr[it] = *reinterpret_cast<const u32x4*>(vb + (int64_t)d * BS);
      }
    } else {
      const int adv = t - n1;
#pragma unroll
      for (int it = 0; it < PP; ++it) {
        kr[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_k, nk_off[it] + adv * 32 * k_row_bytes, 0, 0));
        vr[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, nv_off[it] + adv * 32 * v_row_bytes, 0, 0));
      }
    }
  };
  auto store_tile = [&](int t, char* buf, const u32x4 (&kr)[PP], const u32x4 (&vr)[PP]) {
#pragma unroll
    for (int it = 0; it < PP; ++it) {
      if (t < n1) {
        u32x4 val = vr[it];
        const int t0 = t * 32;
        if (t0 + 32 > ctx) {  // slots past the context may hold anything (NaNs included)
          const int nvalid = max(0, min(8, ctx - (t0 + 8 * (tid & 3))));
#pragma unroll
          for (int dw = 0; dw < 4; ++dw)
            val[dw] &= (nvalid >= 2 * dw + 2) ? 0xffffffffu : ((nvalid == 2 * dw + 1) ? 0x0000ffffu : 0u);
        }
        *reinterpret_cast<u32x4*>(buf + cv_lds[it]) = val;
        *reinterpret_cast<u32x4*>(buf + ck_lds[it]) = kr[it];
      } else {
        *reinterpret_cast<u32x4*>(buf + nv_lds[it]) = vr[it];
        *reinterpret_cast<u32x4*>(buf + nk_lds[it]) = kr[it];
      }
    }
  };

  // one tile of this wave out of the image in `buf`; a single code path for cached and new tiles (they differ in the
  // V fragment reads and in the mask parameters only), so the accumulators live in one place of the loop body
  auto compute = [&](bool is_ctx, const char* buf, int t0) {
    f32x4 s[GQ][2];
    const char* kb = buf + li * KROW + g * 16;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      u32x4 kf[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const u32x4*>(kb + u * 16 * KROW + ks * 64);
      __builtin_amdgcn_sched_barrier(0);  // all fragment reads in flight before the first MFMA waits on one
#pragma unroll
      for (int h = 0; h < GQ; ++h) {
        s[h][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s[h][u] = mfma<scalar_t>(kf[ks], qf[h][ks], s[h][u]);
      }
    }
    u32x4 vf[NT];
    const char* vbuf = buf + KIMG;
    if (is_ctx) {
      const char* base = vbuf + (li * 4 + (g ^ ((li >> 2) & 2))) * 16;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) vf[nt] = *reinterpret_cast<const u32x4*>(base + nt * 1024);
    } else {
      // V^T fragments: lane i of a 16-lane group supplies the address of row (i >> 2), columns 4 (i & 3) .. + 3 of a
      // 4 x 16 block and receives column i of its 4 rows; two blocks = tokens 8g .. 8g + 7 of d column 16 nt + i
      const char* base = vbuf + (8 * g + (li >> 2)) * VROW + (li & 3) * 8;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(base + nt * 32));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(base + 4 * VROW + nt * 32));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        vf[nt] = u32x4{l2[0], l2[1], h2[0], h2[1]};
      }
    }
    const int kpos0 = is_ctx ? t0 : ctx + t0;
    const int limit = is_ctx ? ctx : ctx + q_len;
    const bool fast = plain && (is_ctx ? t0 + 32 <= ctx : t0 + 31 <= r0);
    static_for<0, GQ>([&](auto hc) {
      constexpr int h = decltype(hc)::value;
      softmax_pv<scalar_t, NT>(s[h], m_runs[h], l_parts[h], os[h], vf, rc, slope[h], kpos0, limit, !is_ctx, fast);
    });
  };

  const int nt_all = n1 + n2;
  u32x4 kr[PP], vr[PP];
  int kphys[PP] = {}, vphys[PP] = {};
  load_bt(0, kphys, vphys);
  load_tile(0, kr, vr, kphys, vphys);
  load_bt(1, kphys, vphys);
  store_tile(0, smem, kr, vr);
  __syncthreads();
  for (int t = 0; t < nt_all; ++t) {
    char* buf = smem + (t & 1) * BUF;
    const bool more = t + 1 < nt_all;
    if (more) {
      load_tile(t + 1, kr, vr, kphys, vphys);
      load_bt(t + 2, kphys, vphys);
    }
    const bool is_ctx = t < n1;
    const int t0 = is_ctx ? t * 32 : (t - n1) * 32;
    if (wave_on && (is_ctx || t0 < n_end)) compute(is_ctx, buf, t0);
    if (more) store_tile(t + 1, smem + ((t + 1) & 1) * BUF, kr, vr);
    __syncthreads();
  }

  if (!wave_on) return;
#pragma unroll
  for (int h = 0; h < GQ; ++h) {
    const float l_part = rows_sum(l_parts[h]);
    if (!row_ok) continue;
    const float inv = 1.f / l_part;
    scalar_t* op = reinterpret_cast<scalar_t*>(p.out) + (int64_t)(start + row) * p.o_st + (int64_t)(head0 + h) * p.o_sh;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      union { scalar_t e[4]; u32x2 u; } r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r.e[j] = Scalar<scalar_t>::from_f32(os[h][nt][j] * inv);
      *reinterpret_cast<u32x2*>(op + 16 * nt + 4 * g) = r.u;
    }
  }
}

template <typename scalar_t, int D, int GQ>
int launch_shared(const PrefillParams& p, int batch, int max_input_len, hipStream_t stream) {
  constexpr int smem = 2 * (32 * (D * 2 + 32) + 32 * (D * 2 + 16));
  auto kern = prefill_attention_shared_kernel<scalar_t, D, GQ>;
  if (smem > 64 * 1024)
    NMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  dim3 grid(p.num_heads / GQ, batch, ceil_div(max_input_len, 64));
  kern<<<grid, 256, smem, stream>>>(p);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

template <typename scalar_t, int D>
int launch(const PrefillParams& p, int batch, int max_input_len, hipStream_t stream) {
  // NMX_PREFILL_GQ: 1 / 2 = the per-wave kernel with that many heads per wave (tests / sweeps); 11 / 12 / 14 = the
  // workgroup-shared kernel with 1 / 2 / 4 heads per wave
  int force = 0;
  if (const char* e = nmx_tune(NMX_TUNE_PREFILL_GQ)) force = atoi(e);
  // the shared kernel addresses the new K / V rows of a sequence with 32-bit byte offsets
  const bool fits32 = (int64_t)max_input_len * 2 * (p.k_st > p.v_st ? p.k_st : p.v_st) < (int64_t)1 << 31;
  if constexpr (D % 64 == 0) {
    const int group = p.num_heads / p.num_kv_heads;
    int sgq = 0;
    if (force >= 11 && fits32) sgq = force - 10;
    else if (force == 0 && max_input_len >= 64 && fits32) sgq = (D <= 128 && group % 2 == 0) ? 2 : 1;
    if (sgq == 4 && (D > 128 || group % 4 != 0)) sgq = 2;
    if (sgq == 2 && (D > 128 || group % 2 != 0)) sgq = 1;
    if (sgq == 4) { if constexpr (D <= 128) return launch_shared<scalar_t, D, 4>(p, batch, max_input_len, stream); }
    if (sgq == 2) { if constexpr (D <= 128) return launch_shared<scalar_t, D, 2>(p, batch, max_input_len, stream); }
    if (sgq == 1) return launch_shared<scalar_t, D, 1>(p, batch, max_input_len, stream);
  }

  const size_t smem = (size_t)4 * 32 * (D * 2 + 16);
  // two query heads per wave whenever the GQA group allows it: the K / V fragments of a tile feed twice the MFMAs
  // (Llama-3-8B heads, 64 sequences x 16 new tokens over 1024 cached: 81 us against 175 us with one head per wave)
  const bool gq2_ok = D <= 128 && (p.num_heads / p.num_kv_heads) % 2 == 0;
  bool gq2 = gq2_ok;
  if (force == 1 || force == 2) gq2 = gq2_ok && force == 2;  // tests / sweeps: force either shape
  dim3 grid(gq2 ? p.num_heads / 2 : p.num_heads, batch, ceil_div(max_input_len, 64));
  if (gq2) {
    if constexpr (D <= 128) {
      auto kern = prefill_attention_kernel<scalar_t, D, 2>;
      kern<<<grid, 256, smem, stream>>>(p);
    }
  } else {
    auto kern = prefill_attention_kernel<scalar_t, D, 1>;
    if (smem > 64 * 1024)
      NMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<grid, 256, smem, stream>>>(p);
  }
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

template <typename scalar_t>
int dispatch_head(const PrefillParams& p, int head_size, int batch, int max_input_len, hipStream_t stream) {
  switch (head_size) {
    case 64: return launch<scalar_t, 64>(p, batch, max_input_len, stream);
    case 80: return launch<scalar_t, 80>(p, batch, max_input_len, stream);
    case 96: return launch<scalar_t, 96>(p, batch, max_input_len, stream);
    case 112: return launch<scalar_t, 112>(p, batch, max_input_len, stream);
    case 128: return launch<scalar_t, 128>(p, batch, max_input_len, stream);
    case 192: return launch<scalar_t, 192>(p, batch, max_input_len, stream);
    case 256: return launch<scalar_t, 256>(p, batch, max_input_len, stream);
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "context_attention_fwd: unsupported head size %d", head_size);
  }
}

}  // namespace

extern "C" int nmx_context_attention_fwd(void* out, const void* q, const void* k, const void* v, const void* k_cache,
                                         const void* v_cache, const int32_t* b_loc, const int32_t* b_start_loc,
                                         const int32_t* b_seq_len, const int32_t* b_ctx_len, const float* alibi_slopes,
                                         int batch, int num_heads, int num_kv_heads, int head_size, int block_size, int x,
                                         int64_t q_st, int64_t q_sh, int64_t k_st, int64_t k_sh, int64_t v_st,
                                         int64_t v_sh, int64_t o_st, int64_t o_sh, int64_t kc_sb, int64_t kc_sh,
                                         int64_t vc_sb, int64_t vc_sh, int64_t bloc_stride, int max_input_len,
                                         int sliding_window, float sm_scale, int dtype, nmx_stream_t stream) {
  NMX_CHECK(dtype == NMX_F16 || dtype == NMX_BF16, NMX_ERR_UNSUPPORTED, "context_attention_fwd: float16 / bfloat16 only");
  NMX_CHECK(x == 8, NMX_ERR_UNSUPPORTED, "context_attention_fwd: the key cache must use x = 8 (16-bit cache), got %d", x);
  NMX_CHECK(block_size == 8 || block_size == 16 || block_size == 32, NMX_ERR_UNSUPPORTED, "Unsupported block size: %d",
            block_size);
  NMX_CHECK(num_kv_heads > 0 && num_heads % num_kv_heads == 0, NMX_ERR_INVALID_ARG,
            "num_heads (%d) must be a multiple of num_kv_heads (%d)", num_heads, num_kv_heads);
  NMX_CHECK(q_st % 8 == 0 && q_sh % 8 == 0 && k_st % 8 == 0 && k_sh % 8 == 0 && v_st % 8 == 0 && v_sh % 8 == 0 &&
                o_st % 4 == 0 && o_sh % 4 == 0 && kc_sb % 8 == 0 && kc_sh % 8 == 0 && vc_sb % 8 == 0 && vc_sh % 8 == 0 &&
                (uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 && (uintptr_t)out % 8 == 0 &&
                (uintptr_t)k_cache % 16 == 0 && (uintptr_t)v_cache % 16 == 0,
            NMX_ERR_INVALID_ARG, "context_attention_fwd: tensors must be 16-byte aligned with strides in multiples of 8");
  if (batch == 0 || max_input_len <= 0) return NMX_OK;
  NMX_CHECK(batch <= 65535, NMX_ERR_UNSUPPORTED, "context_attention_fwd: at most 65535 sequences per call, got %d", batch);
  PrefillParams p;
  p.out = out; p.q = q; p.k = k; p.v = v; p.k_cache = k_cache; p.v_cache = v_cache;
  p.b_loc = b_loc; p.b_start_loc = b_start_loc; p.b_seq_len = b_seq_len; p.b_ctx_len = b_ctx_len;
  p.alibi_slopes = alibi_slopes;
  p.q_st = q_st; p.q_sh = q_sh; p.k_st = k_st; p.k_sh = k_sh; p.v_st = v_st; p.v_sh = v_sh; p.o_st = o_st; p.o_sh = o_sh;
  p.kc_sb = kc_sb; p.kc_sh = kc_sh; p.vc_sb = vc_sb; p.vc_sh = vc_sh; p.bloc_stride = bloc_stride;
  p.num_heads = num_heads; p.num_kv_heads = num_kv_heads; p.block_size = block_size;
  p.bs_shift = block_size == 8 ? 3 : (block_size == 16 ? 4 : 5);
  p.sliding_window = sliding_window > 0 ? sliding_window : 0;
  p.scale = sm_scale;
  if (dtype == NMX_F16) return dispatch_head<f16>(p, head_size, batch, max_input_len, (hipStream_t)stream);
  return dispatch_head<bf16>(p, head_size, batch, max_input_len, (hipStream_t)stream);
}
