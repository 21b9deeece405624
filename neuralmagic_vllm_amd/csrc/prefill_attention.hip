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
// One wave = 16 consecutive query tokens of one head; a workgroup = 4 such waves (64 tokens). No workgroup barrier.
// grid (ceil(max_input_len / 64), num_heads, batch), block 256.
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

template <typename scalar_t>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  union { scalar_t h[2]; uint32_t u; } r;
  r.h[0] = Scalar<scalar_t>::from_f32(lo);
  r.h[1] = Scalar<scalar_t>::from_f32(hi);
  return r.u;
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

// GQ = query heads of one kv head processed by a wave (1 or 2): the K / V fragments of a tile are loaded once and
// feed GQ x as many MFMAs - the kernel is bound by the wave-instruction rate of its operand loads, not by the MFMAs.
template <typename scalar_t, int D, int GQ>
__global__ __launch_bounds__(256) void prefill_attention_kernel(const PrefillParams p) {
  constexpr int KS = (D + 31) / 32;
  constexpr int NT = D / 16;
  constexpr int CHUNKS = D / 8;
  constexpr int VROW = D * 2 + 16;  // bytes per staged V row (+16: rows start on different banks)
  const int b = blockIdx.z, head0 = blockIdx.y * GQ;  // heads head0 .. head0 + GQ - 1 share one kv head
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int ctx = p.b_ctx_len[b];
  const int q_len = p.b_seq_len[b] - ctx;
  const int start = p.b_start_loc[b];
  const int r0 = (blockIdx.x * 4 + wave) * 16;
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
    m_runs[h] = -FLT_MAX;
    l_parts[h] = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) os[h][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // one 32-token tile of head h: logits s (already K.Q^T), key positions kpos0 + 16u + 4g + r, V fragments vf
  auto softmax_pv = [&](auto hc, f32x4 (&s)[2], int kpos0, int limit, bool causal, const u32x4 (&vf)[NT]) {
    constexpr int h = decltype(hc)::value;
    float& m_run = m_runs[h];
    float& l_part = l_parts[h];
    f32x4 (&o)[NT] = os[h];
    const float slope_h = slope[h];
    bool msk[2][4];
    float m_tile = -FLT_MAX;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kpos = kpos0 + 16 * u + 4 * g + r;
        float val = s[u][r] * p.scale;
        if (W > 0 && qpos - kpos >= W) val = -10000.f;  // prefix_prefill.py:88-104, :201-204
        if (has_alibi) val += slope_h * (float)(kpos - qpos);  // :552-557
        const bool masked = kpos >= limit || (causal && kpos > qpos) || !row_ok;
        msk[u][r] = masked;
        s[u][r] = val;
        m_tile = masked ? m_tile : fmaxf(m_tile, val);
      }
    }
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 16, 64));
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
    const float m_new = fmaxf(m_run, m_tile);
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    u32x2 pk[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        e[r] = msk[u][r] ? 0.f : __expf(s[u][r] - m_new);
        psum += e[r];
      }
      pk[u][0] = pack2<scalar_t>(e[0], e[1]);
      pk[u][1] = pack2<scalar_t>(e[2], e[3]);
    }
    l_part = l_part * alpha + psum;
    if (__any(alpha != 1.0f)) {  // the running maximum settles after a few tiles: skip the NT x 4 multiplies then
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[nt] *= alpha;
    }
    const u32x4 pb = p_to_operand(pk[0], pk[1]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = mfma<scalar_t>(vf[nt], pb, o[nt]);
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
      softmax_pv(std::integral_constant<int, 0>{}, s[0], t0, ctx, false, vf);
      if constexpr (GQ > 1) softmax_pv(std::integral_constant<int, GQ - 1>{}, s[GQ - 1], t0, ctx, false, vf);
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
        const u32x4 val = *reinterpret_cast<const u32x4*>(Vn + (int64_t)(start + min(t0 + tok, q_len - 1)) * p.v_st + ch * 8);
        vr[it] = (t0 + tok < q_len) ? val : u32x4{0, 0, 0, 0};  // tokens past the sequence as zeros
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
        *reinterpret_cast<u32x4*>(vs + (piece / CHUNKS) * VROW + (piece % CHUNKS) * 16) = vr[it];
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
      softmax_pv(std::integral_constant<int, 0>{}, s[0], ctx + t0, ctx + q_len, true, vf);
      if constexpr (GQ > 1) softmax_pv(std::integral_constant<int, GQ - 1>{}, s[GQ - 1], ctx + t0, ctx + q_len, true, vf);
      __builtin_amdgcn_wave_barrier();
    }
  }

  // ---- out[row][d] = O / l ; lane (g, q) holds O^T[16 nt + 4 g + r][q] ----
#pragma unroll
  for (int h = 0; h < GQ; ++h) {
    float l_part = l_parts[h];
    l_part += __shfl_xor(l_part, 16, 64);
    l_part += __shfl_xor(l_part, 32, 64);
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

template <typename scalar_t, int D>
int launch(const PrefillParams& p, int batch, int max_input_len, hipStream_t stream) {
  const size_t smem = (size_t)4 * 32 * (D * 2 + 16);
  // two query heads per wave when the GQA group allows it (and the wider heads' accumulators still fit)
  // (measured, Llama-3-8B heads: +17-20 % on 1-4 K-token prefills, equal at 512 new + 512 cached tokens; with 16 new
  // tokens per sequence only one wave of a workgroup is active and the second head's registers just cost occupancy)
  const bool gq2_ok = D <= 128 && (p.num_heads / p.num_kv_heads) % 2 == 0;
  bool gq2 = gq2_ok && max_input_len >= 256;
  if (const char* e = nmx_tune(NMX_TUNE_PREFILL_GQ)) gq2 = gq2_ok && atoi(e) == 2;  // tests / sweeps: force either shape
  dim3 grid(ceil_div(max_input_len, 64), gq2 ? p.num_heads / 2 : p.num_heads, batch);
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
