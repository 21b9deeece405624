// Paged-attention decode (v1 / v2) for gfx950 — MFMA, wave64, GQA-aware.
//
// Replaces csrc/attention/attention_kernels.cu of the reference (paged_attention_v1 :805-826,
// paged_attention_v2 :966-990, reduce kernel :567-669). Same op semantics; a different algorithm:
//
//  * One workgroup per (kv_head, sequence[, 512-token partition]) handles ALL query heads that share the kv head
//    (the reference launches one block per query head, so a GQA-4 model fetches each KV byte 4 times).
//  * The KV-cache layout the reference prescribes maps 1:1 onto MFMA 16x16x32 operand fragments:
//      K cache [.., D/8, BS, 8] (fp16): the 16 B at chunk c, token t ARE the A-fragment of lane (g = c % 4, i = t)
//      for S^T = K . Q^T, so K goes HBM -> VGPR -> MFMA with one coalesced dwordx4 per lane and no LDS;
//      V cache [.., D, BS]: the 16 B at row d, tokens 8g..8g+7 ARE the A-fragment of lane (g, i = d % 16) for
//      O^T = V^T . P^T.
//    Both products are computed transposed so that the query row sits on lane & 15 for the softmax statistics, the
//    P fragment and the O accumulator alike: running max / sum / rescale are lane-local.
//  * P (4 tokens per lane per 16-token sub-tile, C/D layout) is re-shaped into the B-operand layout (8 consecutive
//    tokens per lane) with two cross-lane swaps (v_permlane32_swap + v_permlane16_swap), no LDS round trip.
//  * Online softmax over 32-token tiles, fp32 statistics; probabilities are cast to scalar_t before P.V like the
//    reference (:398-400); the final normaliser is 1 / (sum + 1e-6) (:342, :652).
//
// HBM-bound: algorithmic bytes per (sequence, kv head) = 2 * seq_len * D * sizeof(cache element).
#include <float.h>
#include <stdlib.h>

#include "nmx_common.h"

// KV bytes are read exactly once per launch (one workgroup per kv head): they are requested with the non-temporal hint
// so that they do not displace each other in L2 on their way through - measured 187 -> 170 us per launch at batch 256
// (5.75 -> 6.32 TB/s), 49.7 -> 44.4 us at batch 64. NMX_KV_NT=0 builds the plain loads (A/B).
#ifndef NMX_KV_NT
#define NMX_KV_NT 1
#endif
#if NMX_KV_NT
#define NMX_KV_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define NMX_KV_LOAD(ptr) (*(ptr))
#endif

namespace {

constexpr int kPartitionSize = 512;  // the op contract's partition (attention_kernels.cu:847); nmx_paged_attention_v2_ps takes its own
constexpr int kTile = 32;            // tokens per wave iteration

struct AttnParams {
  void* out;          // v1: [S, H, D]; v2: tmp_out [S, H, P, D]
  void* final_out;    // v2 with the in-kernel reduce (counters != null): [S, H, D]
  float* exp_sums;    // v2: [S, H, P]
  float* max_logits;  // v2: [S, H, P]
  const void* q;
  const void* k_cache;
  const void* v_cache;
  const int32_t* block_tables;
  const int32_t* seq_lens;
  const float* alibi_slopes;
  int64_t q_stride, kv_block_stride, kv_head_stride;
  float scale, kv_scale;
  int num_heads, num_kv_heads, q_per_kv, q_tiles;
  int max_blocks_per_seq, block_size, bs_shift;
  int partitioned, max_num_partitions;
  int* counters;      // v2, in-kernel reduce: [num_seqs][kv heads x q tiles] arrival counts, all zero before AND after a launch (or null)
  int part_size;      // tokens per partition (v2: 512 by contract; nmx_paged_attention_v2_ps: 64 .. 512, a multiple of 64)
  int sparse, tp_rank, bs_local_blocks, bs_vert_stride, bs_block_size, bs_head_sliding_step;
  // optional (nmx_paged_attention_v1/v2_absmax): max |out| of what a workgroup (v1: one per kv head x q tile x sequence) / the
  // v2 reduce (one per head x sequence) wrote, so that a dynamic fp8 quantisation of the attention output needs no absmax pass
  float* absmax;
};

template <typename scalar_t>
__device__ __forceinline__ f32x4 mfma_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (sizeof(scalar_t) == 2 && __is_same(scalar_t, f16)) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
}

typedef float pa_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 pa_f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pa_bf16x2 __attribute__((ext_vector_type(2)));

// two floats -> one packed register, RNE: ONE v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32 (the scalar form through a union
// compiled to v_cvt + v_cvt_sdwa + v_or: three instructions per pair)
template <typename scalar_t>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  const pa_f32x2 v = {lo, hi};
  if constexpr (__is_same(scalar_t, f16)) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pa_f16x2));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pa_bf16x2));
}

// 8 fp8 bytes (two dwords) -> 8 scalar_t (four dwords): scalar_t(float(fp8) * scale), RNE
// (reference: csrc/quantization/fp8/nvidia/quant_utils.cuh:293-345)
template <typename scalar_t, int KV>
__device__ __forceinline__ u32x4 cvt8_fp8(u32x2 w, float scale) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    f32x2 lo, hi;
    if constexpr (KV == NMX_KV_FP8_E4M3) {
      lo = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], false);
      hi = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], true);
    } else {
      lo = __builtin_amdgcn_cvt_pk_f32_bf8(w[i], false);
      hi = __builtin_amdgcn_cvt_pk_f32_bf8(w[i], true);
    }
    r[2 * i] = pack2<scalar_t>(lo[0] * scale, lo[1] * scale);
    r[2 * i + 1] = pack2<scalar_t>(hi[0] * scale, hi[1] * scale);
  }
  return r;
}

// 8 fp8 bytes -> 8 scalar_t, no scale: every e4m3 / e5m2 value is exactly representable in fp16 and bf16
template <typename scalar_t, int KV>
__device__ __forceinline__ u32x4 cvt8_fp8_exact(u32x2 w) {
  // gfx950 converts two fp8 bytes straight into a packed fp16 / bf16 pair (v_cvt_scalef32_pk_{f16,bf16}_{fp8,bf8}, unit
  // scale): one instruction per pair instead of fp8 -> f32 -> pack (the fp8-KV decode kernel was VALU-busy 68 %)
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if constexpr (__is_same(scalar_t, f16)) {
      if constexpr (KV == NMX_KV_FP8_E4M3) {
        r[2 * i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w[i], 1.0f, false));
        r[2 * i + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w[i], 1.0f, true));
      } else {
        r[2 * i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(w[i], 1.0f, false));
        r[2 * i + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(w[i], 1.0f, true));
      }
    } else {
      if constexpr (KV == NMX_KV_FP8_E4M3) {
        r[2 * i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[i], 1.0f, false));
        r[2 * i + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[i], 1.0f, true));
      } else {
        r[2 * i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(w[i], 1.0f, false));
        r[2 * i + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(w[i], 1.0f, true));
      }
    }
  }
  return r;
}

// Re-shape P from the MFMA C/D layout into the B-operand layout.
// In : a = sub-tile 0, b = sub-tile 1; lane (g, q) holds tokens 4g..4g+3 of each (two packed dwords).
// Out: lane (g, q) holds tokens 8(g&1)..8(g&1)+7 of sub-tile g>>1 (four packed dwords, token order).
__device__ __forceinline__ u32x4 p_to_operand(u32x2 a, u32x2 b) {
  u32x4 r;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    // lanes 32-63 of a <-> lanes 0-31 of b
    auto s1 = __builtin_amdgcn_permlane32_swap(a[d], b[d], false, false);
    // rows 1,3 of the first <-> rows 0,2 of the second
    auto s2 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
    r[d] = s2[0];
    r[2 + d] = s2[1];
  }
  return r;
}


// (v2_reduce_head - the reduce of one (sequence, head) by one wave - lives in nmx_common.h: the reduce kernel, the in-kernel
// reduce below and marlin_decode_kernel's attention-reduce prologue share it)

// ---- v2 without a reduce launch (round 3, late): every partition workgroup of a (sequence, kv head, q tile) publishes its
// partial results, then takes a ticket on that group's counter; the LAST one to arrive reduces the group's heads (one wave per
// head, v2_reduce_head) and puts the counter back to zero. Nobody waits for anybody: a workgroup either finds it is last or
// leaves. Visibility across CUs / XCDs (MI355X_MICROARCH.md, inter-workgroup visibility, the form without fences): EVERY
// partial store is a write-through agent-scope store, drained (vmcnt(0)) before the workgroup's barrier and the ticket (an
// agent-scope atomic); the last arriver reads EVERY partial with agent-scope loads. Called by all threads of the workgroup at the end of a partitioned launch; smem_f: NW * np floats, free to use. ----
template <typename scalar_t, int NW>
__device__ __forceinline__ void v2_last_arriver_reduce(const AttnParams& p, int seq, int kvh, int qt, int seq_len, int D,
                                                       float* smem_f) {
  __shared__ int s_last;
  // The partial results went out as write-through (agent-scope) stores: drained here, they are in memory before the ticket.
  // (An agent-scope release FENCE instead writes the whole L2's dirty lines back: measured 14.2 vs 9.4 us per attention call at
  // batch 1 and 33.8 vs 10.8 at batch 4 - slower than the reduce launch it was meant to save.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // (also: everyone is done with the LDS images smem_f overlays)
  const int np = (seq_len + p.part_size - 1) / p.part_size;
  int* ctr = p.counters + (int64_t)seq * gridDim.x + blockIdx.x;
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (old == np - 1) ? 1 : 0;
    if (old == np - 1) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  }
  __syncthreads();
  if (s_last == 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows = min(16, p.q_per_kv - qt * 16);  // heads of this q tile
  for (int r = wave; r < rows; r += NW) {
    const int head = kvh * p.q_per_kv + qt * 16 + r;
    const int64_t pb = ((int64_t)seq * p.num_heads + head) * p.max_num_partitions;
    v2_reduce_head<scalar_t, true>(reinterpret_cast<scalar_t*>(p.final_out) + ((int64_t)seq * p.num_heads + head) * D, p.exp_sums + pb,
                             p.max_logits + pb, reinterpret_cast<const scalar_t*>(p.out) + pb * D, np, D,
                             p.absmax != nullptr ? p.absmax + (int64_t)seq * p.num_heads + head : nullptr, smem_f + wave * np, lane);
  }
}

template <typename scalar_t, int KV, int D, int NW>
__global__ __launch_bounds__(NW * 64, (D <= 128) ? 2 : 1) void paged_attention_kernel(const AttnParams p) {
  constexpr int KS = (D + 31) / 32;  // k-steps of the QK^T product
  constexpr int NT = D / 16;         // 16-wide d tiles of the output
  constexpr int CHUNKS = D / 8;      // 8-element chunks per head vector
  constexpr bool FP8 = (KV != NMX_KV_AUTO);
  using cache_t = typename std::conditional<FP8, uint8_t, scalar_t>::type;

  const int kvh = blockIdx.x / p.q_tiles;
  const int qt = blockIdx.x % p.q_tiles;
  const int seq = blockIdx.y;
  const int part = blockIdx.z;
  const int seq_len = p.seq_lens[seq];

  int tok_begin = 0, tok_end = seq_len;
  if (p.partitioned) {
    tok_begin = part * p.part_size;
    if (tok_begin >= seq_len) return;  // attention_kernels.cu:116-119
    tok_end = min(seq_len, tok_begin + p.part_size);
  }

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4;
  const int li = lane & 15;

  const int q_row = qt * 16 + li;
  const bool q_valid = q_row < p.q_per_kv;
  const int head = kvh * p.q_per_kv + (q_valid ? q_row : 0);

  // ---- Q fragments (B operand of S^T = K . Q^T): lane (g, q) holds Q[q][32 ks + 8 g .. +7] ----
  u32x4 qf[KS];
  {
    const scalar_t* qp = reinterpret_cast<const scalar_t*>(p.q) + (int64_t)seq * p.q_stride + (int64_t)head * D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int chunk = 4 * ks + g;
      u32x4 v = {0, 0, 0, 0};
      if (q_valid && chunk < CHUNKS) v = *reinterpret_cast<const u32x4*>(qp + chunk * 8);
      qf[ks] = v;
    }
  }

  const float slope = (p.alibi_slopes != nullptr && q_valid) ? p.alibi_slopes[head] : 0.f;
  const bool has_alibi = p.alibi_slopes != nullptr;
  // block-sparse (attention_kernels.cu:209-256)
  int bs_block_offset = 0, q_bs_block_id = 0;
  if (p.sparse) {
    q_bs_block_id = (seq_len - 1) / p.bs_block_size;
    if (p.bs_head_sliding_step >= 0)
      bs_block_offset = (p.tp_rank * p.num_heads + head) * p.bs_head_sliding_step + 1;
    else
      bs_block_offset = (p.tp_rank * p.num_kv_heads + kvh) * (-p.bs_head_sliding_step) + 1;
  }

  const int32_t* bt = p.block_tables + (int64_t)seq * p.max_blocks_per_seq;
  const cache_t* kc = reinterpret_cast<const cache_t*>(p.k_cache) + (int64_t)kvh * p.kv_head_stride;
  const cache_t* vc = reinterpret_cast<const cache_t*>(p.v_cache) + (int64_t)kvh * p.kv_head_stride;
  const int BS = p.block_size;
  const int bs_mask = BS - 1;
  const int last_tok = seq_len - 1;

  float m_run = -FLT_MAX;
  float l_part = 0.f;
  f32x4 o[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // K fragments of one 32-token tile: lane (g, i) <- token t0 + 16u + i, chunk 4 ks + g; also looks up the physical
  // block of this lane's V tokens (t0 + 8 g .. + 7)
  auto load_k = [&](int t0, u32x4 (&kf)[2][KS], int64_t& vphys) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int tok = min(t0 + 16 * u + li, last_tok);
      const int64_t phys = bt[tok >> p.bs_shift];
      const int off = tok & bs_mask;
      const cache_t* kb = kc + phys * p.kv_block_stride;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int chunk = 4 * ks + g;
        u32x4 v = {0, 0, 0, 0};
        if (chunk < CHUNKS) {
          if constexpr (!FP8) {
            v = NMX_KV_LOAD(reinterpret_cast<const u32x4*>(kb + ((int64_t)chunk * BS + off) * 8));
          } else {
            // x = 16: 16-element chunks; this lane's 8 elements are the (chunk & 1) half of chunk >> 1
            const u32x2 w = NMX_KV_LOAD(reinterpret_cast<const u32x2*>(kb + ((int64_t)(chunk >> 1) * BS + off) * 16 + 8 * (chunk & 1)));
            v = cvt8_fp8<scalar_t, KV>(w, p.kv_scale);
          }
        }
        kf[u][ks] = v;
      }
    }
    vphys = bt[min(t0 + 8 * g, last_tok & ~7) >> p.bs_shift];
  };
  // V fragments: lane (g, i) <- row d = 16 nt + i, tokens t0 + 8 g .. + 7
  auto load_v = [&](int t0, int64_t vphys, u32x4 (&vf)[NT]) {
    const int tokc = min(t0 + 8 * g, last_tok & ~7);
    const cache_t* vb = vc + vphys * p.kv_block_stride + (tokc & bs_mask);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int d = 16 * nt + li;
      if constexpr (!FP8) {
        vf[nt] = NMX_KV_LOAD(reinterpret_cast<const u32x4*>(vb + (int64_t)d * BS));
      } else {
        const u32x2 w = NMX_KV_LOAD(reinterpret_cast<const u32x2*>(vb + (int64_t)d * BS));
        vf[nt] = cvt8_fp8<scalar_t, KV>(w, p.kv_scale);
      }
    }
  };

  // block-table lookup -> K load -> S = K.Q^T is a chain of two dependent memory round trips per tile; the next
  // tile's block ids and K fragments are requested before this tile's MFMAs so that it overlaps the compute (a wave
  // has only 4-8 tiles at decode contexts). V depends on nothing but the block id and is issued at the tile's start.
  constexpr bool PREFETCH = (D <= 128);  // 32 more registers; the wider heads stay at two waves per SIMD without it
  const int n_tiles = (tok_end - tok_begin + kTile - 1) / kTile;
  u32x4 kf[2][KS], kf_n[PREFETCH ? 2 : 1][PREFETCH ? KS : 1];
  int64_t vphys = 0, vphys_n = 0;
  if (PREFETCH && wave < n_tiles) load_k(tok_begin + wave * kTile, kf, vphys);
  for (int tile = wave; tile < n_tiles; tile += NW) {
    const int t0 = tok_begin + tile * kTile;
    const bool more = tile + NW < n_tiles;
    if constexpr (!PREFETCH) load_k(t0, kf, vphys);
    u32x4 vf[NT];
    load_v(t0, vphys, vf);
    if constexpr (PREFETCH) {
      if (more) load_k(t0 + NW * kTile, kf_n, vphys_n);
    }
    if (t0 + kTile > seq_len) {
      // zero V for tokens past the end of the sequence: they may hold NaNs (attention_kernels.cu:420-430)
      const int nvalid = max(0, min(8, seq_len - (t0 + 8 * g)));
#pragma unroll
      for (int dw = 0; dw < 4; ++dw) {
        const uint32_t keep = (nvalid >= 2 * dw + 2) ? 0xffffffffu : ((nvalid == 2 * dw + 1) ? 0x0000ffffu : 0u);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) vf[nt][dw] &= keep;
      }
    }

    // ---- S^T = K . Q^T : lane (g, q) gets tokens t0 + 16u + 4g + r ----
    f32x4 s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[u] = mfma_16x16x32<scalar_t>(kf[u][ks], qf[ks], s[u]);
    }

    // ---- logits, mask, online softmax (the scheme of paged_attention_fp8w_kernel: mask-free path for tiles inside the
    //      sequence, permlane reduction, lazy running maximum) ----
    constexpr float LOG2E = 1.4426950408889634f;
    const bool fast = !p.sparse && !has_alibi && t0 + kTile <= seq_len;
    bool msk[2][4] = {};
    float m_tile;
    if (fast) {
      float mx = s[0][0];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[u][r]);
      m_tile = mx * p.scale;
    } else {
      m_tile = -FLT_MAX;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tok = t0 + 16 * u + 4 * g + r;
          float v = s[u][r] * p.scale;
          v += (slope != 0.f) ? slope * (float)(tok - seq_len + 1) : 0.f;
          bool masked = tok >= seq_len;
          if (p.sparse) {
            const int kb_id = ((tok >> p.bs_shift) << p.bs_shift) / p.bs_block_size;
            const bool is_remote = ((kb_id + bs_block_offset) % p.bs_vert_stride) == 0;
            const bool is_local = kb_id > q_bs_block_id - p.bs_local_blocks;
            masked = masked || !(is_remote || is_local);
          }
          msk[u][r] = masked;
          s[u][r] = v;
          m_tile = masked ? m_tile : fmaxf(m_tile, v);
        }
      }
    }
    {
      auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(m_tile), __float_as_uint(m_tile), false, false);
      m_tile = fmaxf(__uint_as_float(x[0]), __uint_as_float(x[1]));
      auto y = __builtin_amdgcn_permlane16_swap(__float_as_uint(m_tile), __float_as_uint(m_tile), false, false);
      m_tile = fmaxf(__uint_as_float(y[0]), __uint_as_float(y[1]));
    }
    if (__any(m_tile - m_run > 5.f)) {
      const float m_new = fmaxf(m_run, m_tile);
      const float alpha = __expf(m_run - m_new);
      m_run = m_new;
      l_part *= alpha;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[nt] *= alpha;
    }
    const float c = (fast ? p.scale : 1.f) * LOG2E, mc = -m_run * LOG2E;
    u32x2 pk[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = __builtin_amdgcn_exp2f(fmaf(s[u][r], c, mc));
        e[r] = msk[u][r] ? 0.f : x;
      }
      l_part += (e[0] + e[1]) + (e[2] + e[3]);
      pk[u][0] = pack2<scalar_t>(e[0], e[1]);
      pk[u][1] = pack2<scalar_t>(e[2], e[3]);
    }

    // ---- O^T += V^T . P^T ----
    const u32x4 pb = p_to_operand(pk[0], pk[1]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = mfma_16x16x32<scalar_t>(vf[nt], pb, o[nt]);
    if constexpr (PREFETCH) {
      if (more) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) kf[u][ks] = kf_n[u][ks];
        vphys = vphys_n;
      }
    }
  }

  // ---- combine the NW waves through LDS ----
  l_part += __shfl_xor(l_part, 16, 64);
  l_part += __shfl_xor(l_part, 32, 64);

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_m = reinterpret_cast<float*>(smem);          // [NW][16]
  float* lds_l = lds_m + NW * 16;                          // [NW][16]
  float* lds_o = lds_l + NW * 16;                          // [NW][16][D]
  if (g == 0) {
    lds_m[wave * 16 + li] = m_run;
    lds_l[wave * 16 + li] = l_part;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    // lane (g, q) holds O[q][16 nt + 4 g + r]
    *reinterpret_cast<f32x4*>(lds_o + ((int64_t)wave * 16 + li) * D + 16 * nt + 4 * g) = o[nt];
  }
  __syncthreads();

  const int t = threadIdx.x;
  const int cq = t & 15;            // query row within the tile
  const int cq_row = qt * 16 + cq;
  const bool active = cq_row < p.q_per_kv;  // (idle rows of the tile stay for the absmax barrier)
  const int chead = kvh * p.q_per_kv + min(cq_row, p.q_per_kv - 1);
  float M = -FLT_MAX;
#pragma unroll
  for (int w = 0; w < NW; ++w) M = fmaxf(M, lds_m[w * 16 + cq]);
  float f[NW];
  float L = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    f[w] = __expf(lds_m[w * 16 + cq] - M);
    L += lds_l[w * 16 + cq] * f[w];
  }
  const float inv = __fdividef(1.f, L + 1e-6f);
  scalar_t* outp;
  if (p.partitioned) {
    const int64_t pidx = ((int64_t)seq * p.num_heads + chead) * p.max_num_partitions + part;
    outp = reinterpret_cast<scalar_t*>(p.out) + pidx * D;
    if ((t >> 4) == 0 && active) {
      if (p.counters != nullptr) {  // in-kernel reduce: write-through (agent-scope) stores, see v2_last_arriver_reduce
        __hip_atomic_store(p.max_logits + pidx, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.exp_sums + pidx, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        p.max_logits[pidx] = M;
        p.exp_sums[pidx] = L;
      }
    }
  } else {
    outp = reinterpret_cast<scalar_t*>(p.out) + ((int64_t)seq * p.num_heads + chead) * D;
  }
  // thread handles d = 4 * (t >> 4) + {0..3}, stepping by 4 * (NW * 4)
  float amax = 0.f;
  if (active) {
    for (int d0 = 4 * (t >> 4); d0 < D; d0 += 16 * NW) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds_o + ((int64_t)w * 16 + cq) * D + d0);
        acc += v * f[w];
      }
      acc *= inv;
      union { scalar_t h[4]; u32x2 u; } r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r.h[j] = Scalar<scalar_t>::from_f32(acc[j]);
        amax = fmaxf(amax, fabsf(Scalar<scalar_t>::to_f32(r.h[j])));  // of the ROUNDED output: what a later absmax pass would see
      }
      if (p.partitioned && p.counters != nullptr)
        __hip_atomic_store(reinterpret_cast<uint64_t*>(outp + d0), __builtin_bit_cast(uint64_t, r.u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        *reinterpret_cast<u32x2*>(outp + d0) = r.u;
    }
  }
  if (p.absmax != nullptr && !p.partitioned) {  // (uniform) one maximum per workgroup
    amax = wave_reduce_max(amax);
    __syncthreads();  // everyone is done with lds_m
    if (lane == 0) lds_m[wave] = amax;
    __syncthreads();
    if (t == 0) {
      float m = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) m = fmaxf(m, lds_m[w]);
      p.absmax[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = m;
    }
  }
  if (p.partitioned && p.counters != nullptr) v2_last_arriver_reduce<scalar_t, NW>(p, seq, kvh, qt, seq_len, D, reinterpret_cast<float*>(smem));
}

// ---- fp8 KV cache, block_size >= 16: 64-token tiles with 16-byte loads ---------------------------------------------
// paged_attention_kernel reads an fp8 cache with 8 bytes per lane (one lane's 8-element operand fragment), i.e. 512 B per
// wave instruction; a CU retires about one vector-memory wave instruction per ~38 cycles whatever its width, so the fp8
// cache streamed at 3.6 TB/s against 6.4 TB/s for fp16 (1 KiB per instruction). Here every load is 16 bytes per lane:
//  * K: the cache's x = 16 chunk (16 elements of one token) feeds TWO k-steps of S^T = K . Q^T - the k order of a
//    contraction is free, so k-step 2 j + h takes elements 16 (4 j + g) + 8 h .. + 7 and the Q fragments are gathered
//    to match;
//  * V: a lane's 16 bytes of row d are 16 consecutive tokens = the operands of two P.V k-steps over a 64-token tile;
//    the tokens are assigned to the rows of the four 16-token S sub-tiles (tau below) so that the existing two-swap
//    re-shape of P (p_to_operand) delivers exactly the tokens each V half holds.
// Conversion and scaling of every element is scalar_t(float(fp8) * kv_scale) as in the reference
// (quant_utils.cuh:293-345): results equal paged_attention_kernel's up to the summation order of the tile.
template <typename scalar_t, int KV, int D, int NW>
__global__ __launch_bounds__(NW * 64, 2) void paged_attention_fp8w_kernel(const AttnParams p) {
  static_assert(KV != NMX_KV_AUTO && D <= 128 && D % 16 == 0, "fp8 cache, head size <= 128");
  constexpr int C16 = D / 16;        // 16-element chunks per head vector
  constexpr int KP = (C16 + 3) / 4;  // pairs of k-steps of the QK^T product
  constexpr int NT = D / 16;         // 16-wide d tiles of the output
  constexpr int TILE = 64;

  const int kvh = blockIdx.x / p.q_tiles;
  const int qt = blockIdx.x % p.q_tiles;
  const int seq = blockIdx.y;
  const int part = blockIdx.z;
  const int seq_len = p.seq_lens[seq];
  int tok_begin = 0, tok_end = seq_len;
  if (p.partitioned) {
    tok_begin = part * p.part_size;
    if (tok_begin >= seq_len) return;
    tok_end = min(seq_len, tok_begin + p.part_size);
  }
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4;
  const int li = lane & 15;
  const int q_row = qt * 16 + li;
  const bool q_valid = q_row < p.q_per_kv;
  const int head = kvh * p.q_per_kv + (q_valid ? q_row : 0);

  // Q fragments: k-step 2 j + h <- Q[q][16 (4 j + g) + 8 h .. + 7]
  u32x4 qf[KP][2];
  {
    const scalar_t* qp = reinterpret_cast<const scalar_t*>(p.q) + (int64_t)seq * p.q_stride + (int64_t)head * D;
#pragma unroll
    for (int j = 0; j < KP; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c16 = 4 * j + g;
        u32x4 v = {0, 0, 0, 0};
        if (q_valid && c16 < C16) v = *reinterpret_cast<const u32x4*>(qp + c16 * 16 + 8 * h);
        // kv_scale is applied ONCE to the query (and once to the output) instead of to every cache element: the cache
        // values then convert exactly, and K . (s q) = (s K) . q up to one scalar_t rounding of s q in place of one per
        // s K element (the reference's order, quant_utils.cuh:293-345) - same error size, half the conversion VALU
        union { u32x4 u; scalar_t e[8]; } qs;
        qs.u = v;
#pragma unroll
        for (int e = 0; e < 8; ++e) qs.e[e] = Scalar<scalar_t>::from_f32(Scalar<scalar_t>::to_f32(qs.e[e]) * p.kv_scale);
        qf[j][h] = qs.u;
      }
  }
  const float slope = (p.alibi_slopes != nullptr && q_valid) ? p.alibi_slopes[head] : 0.f;
  const bool has_alibi = p.alibi_slopes != nullptr;
  int bs_block_offset = 0, q_bs_block_id = 0;
  if (p.sparse) {
    q_bs_block_id = (seq_len - 1) / p.bs_block_size;
    if (p.bs_head_sliding_step >= 0) bs_block_offset = (p.tp_rank * p.num_heads + head) * p.bs_head_sliding_step + 1;
    else bs_block_offset = (p.tp_rank * p.num_kv_heads + kvh) * (-p.bs_head_sliding_step) + 1;
  }
  const int32_t* bt = p.block_tables + (int64_t)seq * p.max_blocks_per_seq;
  const uint8_t* kc = reinterpret_cast<const uint8_t*>(p.k_cache) + (int64_t)kvh * p.kv_head_stride;
  const uint8_t* vc = reinterpret_cast<const uint8_t*>(p.v_cache) + (int64_t)kvh * p.kv_head_stride;
  const int BS = p.block_size;
  const int bs_mask = BS - 1;
  const int last_tok = seq_len - 1;

  // row i of S sub-tile u is token t0 + tau(u, i): sub-tiles 0, 1 hold the first 8 tokens of the four 16-token groups,
  // sub-tiles 2, 3 the second 8 - after p_to_operand(sub 0, sub 1) lane group g holds tokens 16 g .. 16 g + 7, after
  // p_to_operand(sub 2, sub 3) tokens 16 g + 8 .. 16 g + 15: the two halves of its 16-byte V loads
  auto tau = [](int u, int i) { return 16 * (2 * (u & 1) + (i >> 3)) + 8 * (u >> 1) + (i & 7); };

  float m_run = -FLT_MAX;
  float l_part = 0.f;
  f32x4 o[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load_k = [&](int t0, u32x4 (&kr)[4][KP], int64_t& vphys) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tok = min(t0 + tau(u, li), last_tok);
      const int64_t phys = bt[tok >> p.bs_shift];
      const uint8_t* kb = kc + phys * p.kv_block_stride + (tok & bs_mask) * 16;
#pragma unroll
      for (int j = 0; j < KP; ++j) {
        const int c16 = 4 * j + g;
        u32x4 v = {0, 0, 0, 0};
        if (c16 < C16) v = NMX_KV_LOAD(reinterpret_cast<const u32x4*>(kb + (int64_t)c16 * BS * 16));
        kr[u][j] = v;
      }
    }
    vphys = bt[min(t0 + 16 * g, last_tok & ~15) >> p.bs_shift];
  };
  auto load_v = [&](int t0, int64_t vphys, u32x4 (&vr)[NT]) {
    const int tokc = min(t0 + 16 * g, last_tok & ~15);
    const uint8_t* vb = vc + vphys * p.kv_block_stride + (tokc & bs_mask);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) vr[nt] = NMX_KV_LOAD(reinterpret_cast<const u32x4*>(vb + (int64_t)(16 * nt + li) * BS));
  };

  const int n_tiles = (tok_end - tok_begin + TILE - 1) / TILE;
  u32x4 kr[4][KP], kr_n[4][KP];
  int64_t vphys = 0, vphys_n = 0;
  if (wave < n_tiles) load_k(tok_begin + wave * TILE, kr, vphys);
  for (int tile = wave; tile < n_tiles; tile += NW) {
    const int t0 = tok_begin + tile * TILE;
    const bool more = tile + NW < n_tiles;
    u32x4 vr[NT];
    load_v(t0, vphys, vr);
    if (more) load_k(t0 + NW * TILE, kr_n, vphys_n);

    f32x4 s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < KP; ++j) {
        s[u] = mfma_16x16x32<scalar_t>(cvt8_fp8_exact<scalar_t, KV>(u32x2{kr[u][j][0], kr[u][j][1]}), qf[j][0], s[u]);
        s[u] = mfma_16x16x32<scalar_t>(cvt8_fp8_exact<scalar_t, KV>(u32x2{kr[u][j][2], kr[u][j][3]}), qf[j][1], s[u]);
      }
    }

    // Softmax of the tile, ~3 VALU per logit on the common path (this kernel was VALU-busy 68 %): tiles that lie wholly
    // inside the sequence, without alibi / block-sparse masks, need no per-element masks and fold the scale into one fma
    // per logit (exp2(s * scale * log2e - m * log2e)); the row maximum is reduced with permlane swaps instead of two LDS
    // round trips; the running maximum is LAZY - accumulators are rescaled only when a row's maximum grows by more than 5
    // (P stays < e^5, well inside fp16 / bf16 range; the waves' (m, l, O) triples are merged with their own maxima below).
    constexpr float LOG2E = 1.4426950408889634f;
    const bool fast = !p.sparse && !has_alibi && t0 + TILE <= seq_len;
    float e[4][4];
    float m_tile;
    bool msk[4][4] = {};
    if (fast) {
      float mx = s[0][0];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[u][r]);
      m_tile = mx * p.scale;
    } else {
      m_tile = -FLT_MAX;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tok = t0 + tau(u, 4 * g + r);
          float v = s[u][r] * p.scale;
          v += (slope != 0.f) ? slope * (float)(tok - seq_len + 1) : 0.f;
          bool masked = tok >= seq_len;
          if (p.sparse) {
            const int kb_id = ((tok >> p.bs_shift) << p.bs_shift) / p.bs_block_size;
            const bool is_remote = ((kb_id + bs_block_offset) % p.bs_vert_stride) == 0;
            const bool is_local = kb_id > q_bs_block_id - p.bs_local_blocks;
            masked = masked || !(is_remote || is_local);
          }
          msk[u][r] = masked;
          s[u][r] = v;
          m_tile = masked ? m_tile : fmaxf(m_tile, v);
        }
      }
    }
    {
      auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(m_tile), __float_as_uint(m_tile), false, false);
      m_tile = fmaxf(__uint_as_float(x[0]), __uint_as_float(x[1]));
      auto y = __builtin_amdgcn_permlane16_swap(__float_as_uint(m_tile), __float_as_uint(m_tile), false, false);
      m_tile = fmaxf(__uint_as_float(y[0]), __uint_as_float(y[1]));
    }
    if (__any(m_tile - m_run > 5.f)) {
      const float m_new = fmaxf(m_run, m_tile);
      const float alpha = __expf(m_run - m_new);
      m_run = m_new;
      l_part *= alpha;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[nt] *= alpha;
    }
    const float c = (fast ? p.scale : 1.f) * LOG2E, mc = -m_run * LOG2E;
    u32x2 pk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = __builtin_amdgcn_exp2f(fmaf(s[u][r], c, mc));
        e[u][r] = msk[u][r] ? 0.f : x;
      }
      l_part += (e[u][0] + e[u][1]) + (e[u][2] + e[u][3]);
      pk[u][0] = pack2<scalar_t>(e[u][0], e[u][1]);
      pk[u][1] = pack2<scalar_t>(e[u][2], e[u][3]);
    }

    const u32x4 pb_a = p_to_operand(pk[0], pk[1]);
    const u32x4 pb_b = p_to_operand(pk[2], pk[3]);
    const bool tail = t0 + TILE > seq_len;
    const int nva = max(0, min(8, seq_len - (t0 + 16 * g))), nvb = max(0, min(8, seq_len - (t0 + 16 * g + 8)));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      u32x4 va = cvt8_fp8_exact<scalar_t, KV>(u32x2{vr[nt][0], vr[nt][1]});
      u32x4 vb = cvt8_fp8_exact<scalar_t, KV>(u32x2{vr[nt][2], vr[nt][3]});
      if (tail) {  // zero V past the end of the sequence: those bytes may decode to NaNs (attention_kernels.cu:420-430)
#pragma unroll
        for (int dw = 0; dw < 4; ++dw) {
          va[dw] &= (nva >= 2 * dw + 2) ? 0xffffffffu : ((nva == 2 * dw + 1) ? 0x0000ffffu : 0u);
          vb[dw] &= (nvb >= 2 * dw + 2) ? 0xffffffffu : ((nvb == 2 * dw + 1) ? 0x0000ffffu : 0u);
        }
      }
      o[nt] = mfma_16x16x32<scalar_t>(va, pb_a, o[nt]);
      o[nt] = mfma_16x16x32<scalar_t>(vb, pb_b, o[nt]);
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < KP; ++j) kr[u][j] = kr_n[u][j];
      vphys = vphys_n;
    }
  }

  // ---- combine the NW waves through LDS ----
  l_part += __shfl_xor(l_part, 16, 64);
  l_part += __shfl_xor(l_part, 32, 64);

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_m = reinterpret_cast<float*>(smem);          // [NW][16]
  float* lds_l = lds_m + NW * 16;                          // [NW][16]
  float* lds_o = lds_l + NW * 16;                          // [NW][16][D]
  if (g == 0) {
    lds_m[wave * 16 + li] = m_run;
    lds_l[wave * 16 + li] = l_part;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    // lane (g, q) holds O[q][16 nt + 4 g + r]
    *reinterpret_cast<f32x4*>(lds_o + ((int64_t)wave * 16 + li) * D + 16 * nt + 4 * g) = o[nt];
  }
  __syncthreads();

  const int t = threadIdx.x;
  const int cq = t & 15;            // query row within the tile
  const int cq_row = qt * 16 + cq;
  const bool active = cq_row < p.q_per_kv;  // (idle rows of the tile stay for the absmax barrier)
  const int chead = kvh * p.q_per_kv + min(cq_row, p.q_per_kv - 1);
  float M = -FLT_MAX;
#pragma unroll
  for (int w = 0; w < NW; ++w) M = fmaxf(M, lds_m[w * 16 + cq]);
  float f[NW];
  float L = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    f[w] = __expf(lds_m[w * 16 + cq] - M);
    L += lds_l[w * 16 + cq] * f[w];
  }
  const float inv = __fdividef(1.f, L + 1e-6f) * p.kv_scale;  // V was accumulated unscaled
  scalar_t* outp;
  if (p.partitioned) {
    const int64_t pidx = ((int64_t)seq * p.num_heads + chead) * p.max_num_partitions + part;
    outp = reinterpret_cast<scalar_t*>(p.out) + pidx * D;
    if ((t >> 4) == 0 && active) {
      if (p.counters != nullptr) {  // in-kernel reduce: write-through (agent-scope) stores, see v2_last_arriver_reduce
        __hip_atomic_store(p.max_logits + pidx, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.exp_sums + pidx, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        p.max_logits[pidx] = M;
        p.exp_sums[pidx] = L;
      }
    }
  } else {
    outp = reinterpret_cast<scalar_t*>(p.out) + ((int64_t)seq * p.num_heads + chead) * D;
  }
  // thread handles d = 4 * (t >> 4) + {0..3}, stepping by 4 * (NW * 4)
  float amax = 0.f;
  if (active) {
    for (int d0 = 4 * (t >> 4); d0 < D; d0 += 16 * NW) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds_o + ((int64_t)w * 16 + cq) * D + d0);
        acc += v * f[w];
      }
      acc *= inv;
      union { scalar_t h[4]; u32x2 u; } r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r.h[j] = Scalar<scalar_t>::from_f32(acc[j]);
        amax = fmaxf(amax, fabsf(Scalar<scalar_t>::to_f32(r.h[j])));  // of the ROUNDED output: what a later absmax pass would see
      }
      if (p.partitioned && p.counters != nullptr)
        __hip_atomic_store(reinterpret_cast<uint64_t*>(outp + d0), __builtin_bit_cast(uint64_t, r.u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        *reinterpret_cast<u32x2*>(outp + d0) = r.u;
    }
  }
  if (p.absmax != nullptr && !p.partitioned) {  // (uniform) one maximum per workgroup
    amax = wave_reduce_max(amax);
    __syncthreads();  // everyone is done with lds_m
    if (lane == 0) lds_m[wave] = amax;
    __syncthreads();
    if (t == 0) {
      float m = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) m = fmaxf(m, lds_m[w]);
      p.absmax[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = m;
    }
  }
  if (p.partitioned && p.counters != nullptr) v2_last_arriver_reduce<scalar_t, NW>(p, seq, kvh, qt, seq_len, D, reinterpret_cast<float*>(smem));
}


// v2 reduce: grid (num_heads, num_seqs), one wave. Follows attention_kernels.cu:567-669.
template <typename scalar_t>
__global__ void paged_attention_v2_reduce_kernel(scalar_t* __restrict__ out, const float* __restrict__ exp_sums,
                                                 const float* __restrict__ max_logits,
                                                 const scalar_t* __restrict__ tmp_out,
                                                 const int32_t* __restrict__ seq_lens, int max_num_partitions,
                                                 int head_size, float* __restrict__ absmax, int part_size) {
  const int head = blockIdx.x, num_heads = gridDim.x, seq = blockIdx.y;
  const int seq_len = seq_lens[seq];
  const int np = (seq_len + part_size - 1) / part_size;
  const int64_t pb = ((int64_t)seq * num_heads + head) * max_num_partitions;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2_reduce_head<scalar_t>(out + ((int64_t)seq * num_heads + head) * head_size, exp_sums + pb, max_logits + pb, tmp_out + pb * head_size,
                           np, head_size, absmax != nullptr ? absmax + (int64_t)seq * num_heads + head : nullptr,
                           reinterpret_cast<float*>(smem), (int)threadIdx.x);
}

// ---- float32 queries / float32 or fp8 cache (attention_kernels.cu:742-803 instantiates `float` too) -------------
// A plain VALU kernel - fp32 has no fast MFMA shape and the dtype only appears in tests and debugging runs, so it is
// written for clarity: one workgroup per (query head, sequence[, partition]); a wave takes 64 tokens at a time, one
// token per lane for q.k (a lane's loads walk the 16-byte K chunks of its token, coalesced across the lanes of a
// block), online softmax over the wave's tiles, then one head dimension per lane for P.V (V rows are contiguous
// over the tokens of a block). The four waves are merged through LDS like the MFMA kernel's.
template <int KV>
__global__ __launch_bounds__(256) void paged_attention_f32_kernel(const AttnParams p, int D) {
  constexpr bool FP8 = (KV != NMX_KV_AUTO);
  using cache_t = typename std::conditional<FP8, uint8_t, float>::type;
  constexpr int X = FP8 ? 16 : 4;  // elements per 16-byte K chunk
  constexpr int NW = 4;
  const int head = blockIdx.x, seq = blockIdx.y, part = blockIdx.z;
  const int seq_len = p.seq_lens[seq];
  int tok_begin = 0, tok_end = seq_len;
  if (p.partitioned) {
    tok_begin = part * p.part_size;
    if (tok_begin >= seq_len) return;
    tok_end = min(seq_len, tok_begin + p.part_size);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kvh = head / p.q_per_kv;
  const int BS = p.block_size;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* q_s = reinterpret_cast<float*>(smem);             // [D]
  float* p_s = q_s + ((D + 3) & ~3);                       // [NW][64]
  float* m_s = p_s + NW * 64;                              // [NW]
  float* l_s = m_s + NW;                                   // [NW]
  float* o_s = l_s + NW;                                   // [NW][D]
  const float* qp = reinterpret_cast<const float*>(p.q) + (int64_t)seq * p.q_stride + (int64_t)head * D;
  for (int d = threadIdx.x; d < D; d += blockDim.x) q_s[d] = qp[d];
  __syncthreads();

  const float slope = p.alibi_slopes != nullptr ? p.alibi_slopes[head] : 0.f;
  int bs_block_offset = 0, q_bs_block_id = 0;
  if (p.sparse) {
    q_bs_block_id = (seq_len - 1) / p.bs_block_size;
    if (p.bs_head_sliding_step >= 0) bs_block_offset = (p.tp_rank * p.num_heads + head) * p.bs_head_sliding_step + 1;
    else bs_block_offset = (p.tp_rank * p.num_kv_heads + kvh) * (-p.bs_head_sliding_step) + 1;
  }
  const int32_t* bt = p.block_tables + (int64_t)seq * p.max_blocks_per_seq;
  const cache_t* kc = reinterpret_cast<const cache_t*>(p.k_cache) + (int64_t)kvh * p.kv_head_stride;
  const cache_t* vc = reinterpret_cast<const cache_t*>(p.v_cache) + (int64_t)kvh * p.kv_head_stride;
  auto ld = [&](const cache_t* ptr) -> float {
    if constexpr (FP8) return fp8_to_f32<KV>(*ptr) * p.kv_scale;
    else return *ptr;
  };

  float m_run = -FLT_MAX, l_run = 0.f;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};  // d = lane + 64 j
  float* pw = p_s + wave * 64;
  for (int t0 = tok_begin + wave * 64; t0 < tok_end; t0 += NW * 64) {
    const int tok = t0 + lane;
    const bool valid = tok < tok_end;
    const int blk = tok >> p.bs_shift, off = tok & (BS - 1);
    const int64_t phys = valid ? (int64_t)bt[blk] : 0;
    float dot = 0.f;
    if (valid) {
      const cache_t* kb = kc + phys * p.kv_block_stride + (int64_t)off * X;
      for (int c = 0; c < D / X; ++c) {
#pragma unroll
        for (int e = 0; e < X; ++e) dot += q_s[c * X + e] * ld(kb + (int64_t)c * BS * X + e);
      }
    }
    float logit = p.scale * dot + ((slope != 0.f) ? slope * (float)(tok - seq_len + 1) : 0.f);
    bool masked = !valid;
    if (p.sparse) {
      const int kb_id = (blk * BS) / p.bs_block_size;
      const bool is_remote = ((kb_id + bs_block_offset) % p.bs_vert_stride) == 0;
      const bool is_local = kb_id > q_bs_block_id - p.bs_local_blocks;
      masked = masked || !(is_remote || is_local);
    }
    const float m_new = fmaxf(m_run, wave_reduce_max(masked ? -FLT_MAX : logit));
    const float pv = (masked || m_new == -FLT_MAX) ? 0.f : __expf(logit - m_new);
    const float alpha = (m_run == -FLT_MAX) ? 0.f : __expf(m_run - m_new);
    l_run = l_run * alpha + wave_reduce_sum(pv);
    m_run = m_new;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] *= alpha;
    pw[lane] = pv;
    __builtin_amdgcn_wave_barrier();
    const int ntok = min(64, tok_end - t0);
    for (int b0 = 0; b0 < ntok; b0 += BS) {
      const int64_t pb = (int64_t)bt[(t0 + b0) >> p.bs_shift];
      const int nb = min(BS, ntok - b0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d = lane + 64 * j;
        if (d < D) {
          const cache_t* vr = vc + pb * p.kv_block_stride + (int64_t)d * BS;
          float a = acc[j];
          for (int o = 0; o < nb; ++o) a += pw[b0 + o] * ld(vr + o);
          acc[j] = a;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    m_s[wave] = m_run;
    l_s[wave] = l_run;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (lane + 64 * j < D) o_s[wave * D + lane + 64 * j] = acc[j];
  __syncthreads();
  float M = -FLT_MAX;
  for (int w = 0; w < NW; ++w) M = fmaxf(M, m_s[w]);
  float f[NW], L = 0.f;
  for (int w = 0; w < NW; ++w) {
    f[w] = (m_s[w] == -FLT_MAX) ? 0.f : __expf(m_s[w] - M);
    L += l_s[w] * f[w];
  }
  const float inv = __fdividef(1.f, L + 1e-6f);
  float* outp;
  if (p.partitioned) {
    const int64_t pidx = ((int64_t)seq * p.num_heads + head) * p.max_num_partitions + part;
    outp = reinterpret_cast<float*>(p.out) + pidx * D;
    if (threadIdx.x == 0) {
      p.max_logits[pidx] = M;
      p.exp_sums[pidx] = L;
    }
  } else {
    outp = reinterpret_cast<float*>(p.out) + ((int64_t)seq * p.num_heads + head) * D;
  }
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float a = 0.f;
    for (int w = 0; w < NW; ++w) a += o_s[w * D + d] * f[w];
    outp[d] = a * inv;
  }
}

int launch_attn_f32(const AttnParams& p, int kv_dtype, int head_size, int num_seqs, int num_partitions, hipStream_t stream) {
  switch (head_size) {
    case 64: case 80: case 96: case 112: case 128: case 192: case 256: break;
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "Unsupported head size: %d", head_size);
  }
  const size_t smem = (size_t)(((head_size + 3) & ~3) + 4 * 64 + 8 + 4 * head_size) * sizeof(float);
  dim3 grid(p.num_heads, num_seqs, num_partitions);
  switch (kv_dtype) {
    case NMX_KV_AUTO: paged_attention_f32_kernel<NMX_KV_AUTO><<<grid, 256, smem, stream>>>(p, head_size); break;
    case NMX_KV_FP8_E4M3: paged_attention_f32_kernel<NMX_KV_FP8_E4M3><<<grid, 256, smem, stream>>>(p, head_size); break;
    case NMX_KV_FP8_E5M2: paged_attention_f32_kernel<NMX_KV_FP8_E5M2><<<grid, 256, smem, stream>>>(p, head_size); break;
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "Unsupported data type of kv cache: %d", kv_dtype);
  }
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

template <typename scalar_t, int KV, int D, int NW>
int launch_attn_nw(const AttnParams& p, int num_seqs, int num_partitions, hipStream_t stream) {
  const size_t smem = (size_t)NW * 16 * (2 + D) * sizeof(float);
  dim3 grid(p.num_kv_heads * p.q_tiles, num_seqs, num_partitions);
  auto kern = paged_attention_kernel<scalar_t, KV, D, NW>;
  if (smem > 64 * 1024) {
    NMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  }
  kern<<<grid, dim3(NW * 64), smem, stream>>>(p);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

template <typename scalar_t, int KV, int D, int NW>
int launch_attn_fp8w(const AttnParams& p, int num_seqs, int num_partitions, hipStream_t stream) {
  const size_t smem = (size_t)NW * 16 * (2 + D) * sizeof(float);
  dim3 grid(p.num_kv_heads * p.q_tiles, num_seqs, num_partitions);
  paged_attention_fp8w_kernel<scalar_t, KV, D, NW><<<grid, dim3(NW * 64), smem, stream>>>(p);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

template <typename scalar_t, int KV, int D>
int launch_attn(const AttnParams& p, int num_seqs, int num_partitions, hipStream_t stream) {
  if constexpr (KV != NMX_KV_AUTO && D <= 128) {
    // fp8 cache in blocks of >= 16 tokens: 64-token tiles, 16-byte loads (NMX_ATTN_FP8W=0: the 8-byte path, for A/B)
    const char* e = nmx_tune(NMX_TUNE_ATTN_FP8W);
    if (p.block_size >= 16 && !(e != nullptr && atoi(e) == 0)) {
      const long wgs = (long)p.num_kv_heads * p.q_tiles * num_seqs * num_partitions;
      int nw = (wgs <= 128 && p.part_size > 256) ? 8 : 4;
      if (const char* n = nmx_tune(NMX_TUNE_ATTN_NW)) nw = atoi(n) == 8 ? 8 : 4;
      if (nw == 8) return launch_attn_fp8w<scalar_t, KV, D, 8>(p, num_seqs, num_partitions, stream);
      return launch_attn_fp8w<scalar_t, KV, D, 4>(p, num_seqs, num_partitions, stream);
    }
  }
  // Few workgroups (small batch: batch 1 x 8 kv heads x 2 partitions = 16): a CU streams its partition at its own
  // per-CU rate whatever the rest of the chip does, so give every workgroup 8 waves (2 tiles of 32 tokens each at a
  // 512-token partition, all K / V requests of the partition in flight at once) instead of 4.
  if constexpr (D <= 128) {
    const long wgs = (long)p.num_kv_heads * p.q_tiles * num_seqs * num_partitions;
    int nw = (wgs <= 128 && p.part_size > 4 * kTile) ? 8 : 4;  // (a partition of <= 4 tiles: one tile per wave)
    if (const char* e = nmx_tune(NMX_TUNE_ATTN_NW)) nw = atoi(e) == 8 ? 8 : 4;  // sweeps / tests
    if (nw == 8) return launch_attn_nw<scalar_t, KV, D, 8>(p, num_seqs, num_partitions, stream);
  }
  return launch_attn_nw<scalar_t, KV, D, 4>(p, num_seqs, num_partitions, stream);
}

template <typename scalar_t, int KV>
int dispatch_head(const AttnParams& p, int head_size, int num_seqs, int num_partitions, hipStream_t stream) {
  switch (head_size) {
    case 64: return launch_attn<scalar_t, KV, 64>(p, num_seqs, num_partitions, stream);
    case 80: return launch_attn<scalar_t, KV, 80>(p, num_seqs, num_partitions, stream);
    case 96: return launch_attn<scalar_t, KV, 96>(p, num_seqs, num_partitions, stream);
    case 112: return launch_attn<scalar_t, KV, 112>(p, num_seqs, num_partitions, stream);
    case 128: return launch_attn<scalar_t, KV, 128>(p, num_seqs, num_partitions, stream);
    case 192: return launch_attn<scalar_t, KV, 192>(p, num_seqs, num_partitions, stream);
    case 256: return launch_attn<scalar_t, KV, 256>(p, num_seqs, num_partitions, stream);
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "Unsupported head size: %d", head_size);
  }
}

template <typename scalar_t>
int dispatch_kv(const AttnParams& p, int kv_dtype, int head_size, int num_seqs, int num_partitions, hipStream_t stream) {
  switch (kv_dtype) {
    case NMX_KV_AUTO: return dispatch_head<scalar_t, NMX_KV_AUTO>(p, head_size, num_seqs, num_partitions, stream);
    case NMX_KV_FP8_E4M3: return dispatch_head<scalar_t, NMX_KV_FP8_E4M3>(p, head_size, num_seqs, num_partitions, stream);
    case NMX_KV_FP8_E5M2: return dispatch_head<scalar_t, NMX_KV_FP8_E5M2>(p, head_size, num_seqs, num_partitions, stream);
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "Unsupported data type of kv cache: %d", kv_dtype);
  }
}

int run_attention(bool partitioned, void* out, float* absmax, float* exp_sums, float* max_logits, void* tmp_out, const void* query,
                  const void* key_cache, const void* value_cache, int num_seqs, int num_heads, int num_kv_heads,
                  int head_size, int block_size, int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                  float scale, const int32_t* block_tables, int max_num_blocks_per_seq, const int32_t* seq_lens,
                  int max_seq_len, const float* alibi_slopes, int dtype, int kv_dtype, float kv_scale, int tp_rank,
                  int bs_local_blocks, int bs_vert_stride, int bs_block_size, int bs_head_sliding_step,
                  hipStream_t stream, int part_size = kPartitionSize, int* counters = nullptr, bool skip_reduce = false) {
  NMX_CHECK(part_size >= 64 && part_size <= kPartitionSize && part_size % 64 == 0, NMX_ERR_INVALID_ARG,
            "paged_attention: partition size %d (64 .. 512, a multiple of 64)", part_size);
  NMX_CHECK(block_size == 8 || block_size == 16 || block_size == 32, NMX_ERR_UNSUPPORTED,
            "Unsupported block size: %d", block_size);
  NMX_CHECK(num_kv_heads > 0 && num_heads % num_kv_heads == 0, NMX_ERR_INVALID_ARG,
            "num_heads (%d) must be a multiple of num_kv_heads (%d)", num_heads, num_kv_heads);
  NMX_CHECK(dtype == NMX_F16 || dtype == NMX_BF16 || dtype == NMX_F32, NMX_ERR_UNSUPPORTED,
            "paged_attention: unsupported query dtype code %d", dtype);
  NMX_CHECK(((uintptr_t)query % 16 == 0) && (dtype == NMX_F32 || q_stride % 8 == 0), NMX_ERR_INVALID_ARG,
            "paged_attention: query must be 16-byte aligned with a row stride that is a multiple of 8 elements");
  NMX_CHECK(((uintptr_t)key_cache % 16 == 0) && ((uintptr_t)value_cache % 16 == 0) && kv_block_stride % 16 == 0 &&
                kv_head_stride % 16 == 0,
            NMX_ERR_INVALID_ARG, "paged_attention: KV cache must be 16-byte aligned");
  if (num_seqs == 0) return NMX_OK;

  AttnParams p;
  NMX_CHECK(absmax == nullptr || dtype != NMX_F32, NMX_ERR_UNSUPPORTED, "paged_attention absmax: float16 / bfloat16 queries only");
  p.absmax = absmax;
  p.out = partitioned ? tmp_out : out;
  p.exp_sums = exp_sums;
  p.max_logits = max_logits;
  p.q = query;
  p.k_cache = key_cache;
  p.v_cache = value_cache;
  p.block_tables = block_tables;
  p.seq_lens = seq_lens;
  p.alibi_slopes = alibi_slopes;
  p.q_stride = q_stride;
  p.kv_block_stride = kv_block_stride;
  p.kv_head_stride = kv_head_stride;
  p.scale = scale;
  p.kv_scale = kv_scale;
  p.num_heads = num_heads;
  p.num_kv_heads = num_kv_heads;
  p.q_per_kv = num_heads / num_kv_heads;
  p.q_tiles = (p.q_per_kv + 15) / 16;
  p.max_blocks_per_seq = max_num_blocks_per_seq;
  p.block_size = block_size;
  p.bs_shift = block_size == 8 ? 3 : (block_size == 16 ? 4 : 5);
  p.partitioned = partitioned ? 1 : 0;
  const int num_partitions = partitioned ? (max_seq_len + part_size - 1) / part_size : 1;
  p.max_num_partitions = num_partitions;
  p.part_size = part_size;
  // in-kernel reduce by the last-arriving partition: the MFMA kernels only (fp16 / bf16 queries), a sane partition count
  p.counters = (partitioned && dtype != NMX_F32 && num_partitions <= 512) ? counters : nullptr;
  p.final_out = out;
  p.sparse = bs_vert_stride > 1 ? 1 : 0;  // attention_kernels.cu:822
  p.tp_rank = tp_rank;
  p.bs_local_blocks = bs_local_blocks;
  p.bs_vert_stride = bs_vert_stride;
  p.bs_block_size = bs_block_size;
  p.bs_head_sliding_step = bs_head_sliding_step;
  if (p.sparse) NMX_CHECK(bs_block_size > 0, NMX_ERR_INVALID_ARG, "blocksparse_block_size must be > 0");
  if (partitioned && num_partitions == 0) return NMX_OK;

  int rc;
  if (dtype == NMX_F32) rc = launch_attn_f32(p, kv_dtype, head_size, num_seqs, num_partitions, stream);
  else if (dtype == NMX_F16) rc = dispatch_kv<f16>(p, kv_dtype, head_size, num_seqs, num_partitions, stream);
  else rc = dispatch_kv<bf16>(p, kv_dtype, head_size, num_seqs, num_partitions, stream);
  if (rc != NMX_OK || !partitioned || p.counters != nullptr || skip_reduce) return rc;

  dim3 rgrid(num_heads, num_seqs);
  const size_t rsmem = (size_t)num_partitions * sizeof(float);
  if (dtype == NMX_F32)
    paged_attention_v2_reduce_kernel<float><<<rgrid, 64, rsmem, stream>>>((float*)out, exp_sums, max_logits,
                                                                          (const float*)tmp_out, seq_lens, num_partitions,
                                                                          head_size, nullptr, part_size);
  else if (dtype == NMX_F16)
    paged_attention_v2_reduce_kernel<f16><<<rgrid, 64, rsmem, stream>>>((f16*)out, exp_sums, max_logits, (const f16*)tmp_out,
                                                                        seq_lens, num_partitions, head_size, absmax, part_size);
  else
    paged_attention_v2_reduce_kernel<bf16><<<rgrid, 64, rsmem, stream>>>((bf16*)out, exp_sums, max_logits,
                                                                         (const bf16*)tmp_out, seq_lens, num_partitions,
                                                                         head_size, absmax, part_size);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

}  // namespace

extern "C" int nmx_paged_attention_v1(void* out, const void* query, const void* key_cache, const void* value_cache,
                                      int num_seqs, int num_heads, int num_kv_heads, int head_size, int block_size,
                                      int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride, float scale,
                                      const int32_t* block_tables, int max_num_blocks_per_seq,
                                      const int32_t* seq_lens, int max_seq_len, const float* alibi_slopes, int dtype,
                                      int kv_dtype, float kv_scale, int tp_rank, int bs_local_blocks,
                                      int bs_vert_stride, int bs_block_size, int bs_head_sliding_step,
                                      nmx_stream_t stream) {
  return run_attention(false, out, nullptr, nullptr, nullptr, nullptr, query, key_cache, value_cache, num_seqs, num_heads,
                       num_kv_heads, head_size, block_size, q_stride, kv_block_stride, kv_head_stride, scale,
                       block_tables, max_num_blocks_per_seq, seq_lens, max_seq_len, alibi_slopes, dtype, kv_dtype,
                       kv_scale, tp_rank, bs_local_blocks, bs_vert_stride, bs_block_size, bs_head_sliding_step,
                       (hipStream_t)stream);
}

extern "C" int nmx_paged_attention_v2(void* out, float* exp_sums, float* max_logits, void* tmp_out,
                                      const void* query, const void* key_cache, const void* value_cache,
                                      int num_seqs, int num_heads, int num_kv_heads, int head_size, int block_size,
                                      int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride, float scale,
                                      const int32_t* block_tables, int max_num_blocks_per_seq,
                                      const int32_t* seq_lens, int max_seq_len, const float* alibi_slopes, int dtype,
                                      int kv_dtype, float kv_scale, int tp_rank, int bs_local_blocks,
                                      int bs_vert_stride, int bs_block_size, int bs_head_sliding_step,
                                      nmx_stream_t stream) {
  return run_attention(true, out, nullptr, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs, num_heads,
                       num_kv_heads, head_size, block_size, q_stride, kv_block_stride, kv_head_stride, scale,
                       block_tables, max_num_blocks_per_seq, seq_lens, max_seq_len, alibi_slopes, dtype, kv_dtype,
                       kv_scale, tp_rank, bs_local_blocks, bs_vert_stride, bs_block_size, bs_head_sliding_step,
                       (hipStream_t)stream);
}

// ---- the same ops with the output's absolute maxima as a by-product (fp8 W8A8 models: o_proj's dynamic per-tensor activation
// quantisation then needs no absmax pass over the attention output - nmx_scaled_fp8_quant_partials takes these). absmax:
// float32 [nmx_paged_attention_absmax_numel(...)], every entry written (v1: one per kv head x q tile x sequence, v2: one per
// head x sequence); max over all entries = max |out| exactly (taken on the rounded outputs). fp16 / bf16 queries.
extern "C" int nmx_paged_attention_absmax_numel(int num_seqs, int num_heads, int num_kv_heads, int partitioned) {
  if (num_seqs <= 0 || num_heads <= 0 || num_kv_heads <= 0) return 0;
  if (partitioned) return num_seqs * num_heads;
  const int q_per_kv = num_heads / num_kv_heads;
  return num_seqs * num_kv_heads * ((q_per_kv + 15) / 16);
}

extern "C" int nmx_paged_attention_v1_absmax(void* out, float* absmax, const void* query, const void* key_cache,
                                             const void* value_cache, int num_seqs, int num_heads, int num_kv_heads, int head_size,
                                             int block_size, int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                                             float scale, const int32_t* block_tables, int max_num_blocks_per_seq,
                                             const int32_t* seq_lens, int max_seq_len, const float* alibi_slopes, int dtype,
                                             int kv_dtype, float kv_scale, int tp_rank, int bs_local_blocks, int bs_vert_stride,
                                             int bs_block_size, int bs_head_sliding_step, nmx_stream_t stream) {
  NMX_CHECK(absmax != nullptr, NMX_ERR_INVALID_ARG, "paged_attention_v1_absmax: absmax must be non-null");
  return run_attention(false, out, absmax, nullptr, nullptr, nullptr, query, key_cache, value_cache, num_seqs, num_heads,
                       num_kv_heads, head_size, block_size, q_stride, kv_block_stride, kv_head_stride, scale,
                       block_tables, max_num_blocks_per_seq, seq_lens, max_seq_len, alibi_slopes, dtype, kv_dtype,
                       kv_scale, tp_rank, bs_local_blocks, bs_vert_stride, bs_block_size, bs_head_sliding_step,
                       (hipStream_t)stream);
}

extern "C" int nmx_paged_attention_v2_absmax(void* out, float* absmax, float* exp_sums, float* max_logits, void* tmp_out,
                                             const void* query, const void* key_cache, const void* value_cache, int num_seqs,
                                             int num_heads, int num_kv_heads, int head_size, int block_size, int64_t q_stride,
                                             int64_t kv_block_stride, int64_t kv_head_stride, float scale,
                                             const int32_t* block_tables, int max_num_blocks_per_seq, const int32_t* seq_lens,
                                             int max_seq_len, const float* alibi_slopes, int dtype, int kv_dtype, float kv_scale,
                                             int tp_rank, int bs_local_blocks, int bs_vert_stride, int bs_block_size,
                                             int bs_head_sliding_step, nmx_stream_t stream) {
  NMX_CHECK(absmax != nullptr, NMX_ERR_INVALID_ARG, "paged_attention_v2_absmax: absmax must be non-null");
  return run_attention(true, out, absmax, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs, num_heads,
                       num_kv_heads, head_size, block_size, q_stride, kv_block_stride, kv_head_stride, scale,
                       block_tables, max_num_blocks_per_seq, seq_lens, max_seq_len, alibi_slopes, dtype, kv_dtype,
                       kv_scale, tp_rank, bs_local_blocks, bs_vert_stride, bs_block_size, bs_head_sliding_step,
                       (hipStream_t)stream);
}

// ---- v2 with the caller's choice of partition size (round 3, late; no reference counterpart). At small batch the 512-token
// partitions of the op contract leave most CUs idle (batch 1 x 8 kv heads x 2 partitions = 16 workgroups, each streaming its
// 256 KiB of K / V at ONE CU's rate): the decode mirror of PagedAttention.forward_decode asks
// nmx_paged_attention_partition_size() and, when it answers less than 512, allocates its own exp_sums / max_logits / tmp_out
// for ceil(max_seq_len / partition_size) partitions and calls this entry. Same arithmetic per partition and the same reduce
// arithmetic; only the split of the softmax sum differs (as between v1 and v2). absmax may be null.
// counters (may be null): int32 [nmx_paged_attention_counters_numel(...)], ALL ZERO before the call and all zero again after
// it, used by no other launch in flight - with it the last-arriving partition workgroup of every (sequence, kv head) reduces
// the partitions itself (v2_last_arriver_reduce) and no reduce kernel is launched: one dependent launch less (fp16 / bf16
// queries; float32 queries ignore it). The output bits are those of the reduce kernel.
extern "C" int nmx_paged_attention_partition_size(int num_seqs, int num_heads, int num_kv_heads, int max_seq_len) {
  if (num_seqs <= 0 || num_heads <= 0 || num_kv_heads <= 0) return kPartitionSize;
  const int q_tiles = (num_heads / num_kv_heads + 15) / 16;
  int ps = kPartitionSize;
  if (const char* e = nmx_tune(NMX_TUNE_ATTN_PART)) {  // sweeps / tests
    const int v = atoi(e);
    return (v >= 64 && v <= kPartitionSize && v % 64 == 0) ? v : kPartitionSize;
  }
  // halve while the finer split still fits one round of the 256 CUs, a partition keeps >= 128 tokens and a sequence has at most
  // 8 partitions (the reduce walks them one by one: at 4,096 / 8,192 tokens of context the contract's 512 is already the
  // best or within 4 % of it - tools/attn_part_sweep.py, profiles/r03_attn_part_sweep.txt)
  while (ps > 128) {
    const int np = (max_seq_len + ps / 2 - 1) / (ps / 2);
    if (np > 8 || (long)num_seqs * num_kv_heads * q_tiles * np > 256) break;
    ps /= 2;
  }
  return ps;
}

extern "C" int nmx_paged_attention_v2_ps(void* out, float* absmax, float* exp_sums, float* max_logits, void* tmp_out,
                                         const void* query, const void* key_cache, const void* value_cache, int num_seqs,
                                         int num_heads, int num_kv_heads, int head_size, int block_size, int64_t q_stride,
                                         int64_t kv_block_stride, int64_t kv_head_stride, float scale,
                                         const int32_t* block_tables, int max_num_blocks_per_seq, const int32_t* seq_lens,
                                         int max_seq_len, const float* alibi_slopes, int dtype, int kv_dtype, float kv_scale,
                                         int tp_rank, int bs_local_blocks, int bs_vert_stride, int bs_block_size,
                                         int bs_head_sliding_step, int partition_size, int* counters, nmx_stream_t stream) {
  NMX_CHECK(counters == nullptr || (uintptr_t)counters % 4 == 0, NMX_ERR_INVALID_ARG, "paged_attention_v2_ps: counters misaligned");
  return run_attention(true, out, absmax, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs, num_heads,
                       num_kv_heads, head_size, block_size, q_stride, kv_block_stride, kv_head_stride, scale,
                       block_tables, max_num_blocks_per_seq, seq_lens, max_seq_len, alibi_slopes, dtype, kv_dtype,
                       kv_scale, tp_rank, bs_local_blocks, bs_vert_stride, bs_block_size, bs_head_sliding_step,
                       (hipStream_t)stream, partition_size, counters);
}

extern "C" int64_t nmx_paged_attention_counters_numel(int num_seqs, int num_heads, int num_kv_heads) {
  if (num_seqs <= 0 || num_heads <= 0 || num_kv_heads <= 0) return 0;
  return (int64_t)num_seqs * num_kv_heads * ((num_heads / num_kv_heads + 15) / 16);
}

// ---- v2 in two halves (round 3, late): the partition launch alone, and the reduce as an op of its own. A caller that feeds the
// attention output straight into a GPTQ-Marlin o_proj hands the partition results to nmx_gptq_marlin_gemm_attn instead (the GEMM
// reduces them in its prologue: one launch less); nmx_paged_attention_v2_reduce is the fallback for every other consumer.
extern "C" int nmx_paged_attention_v2_partials(float* exp_sums, float* max_logits, void* tmp_out, const void* query,
                                               const void* key_cache, const void* value_cache, int num_seqs, int num_heads,
                                               int num_kv_heads, int head_size, int block_size, int64_t q_stride,
                                               int64_t kv_block_stride, int64_t kv_head_stride, float scale,
                                               const int32_t* block_tables, int max_num_blocks_per_seq, const int32_t* seq_lens,
                                               int max_seq_len, const float* alibi_slopes, int dtype, int kv_dtype, float kv_scale,
                                               int tp_rank, int bs_local_blocks, int bs_vert_stride, int bs_block_size,
                                               int bs_head_sliding_step, int partition_size, nmx_stream_t stream) {
  return run_attention(true, tmp_out /* unused */, nullptr, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs,
                       num_heads, num_kv_heads, head_size, block_size, q_stride, kv_block_stride, kv_head_stride, scale, block_tables,
                       max_num_blocks_per_seq, seq_lens, max_seq_len, alibi_slopes, dtype, kv_dtype, kv_scale, tp_rank,
                       bs_local_blocks, bs_vert_stride, bs_block_size, bs_head_sliding_step, (hipStream_t)stream, partition_size,
                       nullptr, true);
}

extern "C" int nmx_paged_attention_v2_reduce(void* out, const float* exp_sums, const float* max_logits, const void* tmp_out,
                                             const int32_t* seq_lens, int num_seqs, int num_heads, int head_size,
                                             int max_num_partitions, int partition_size, int dtype, nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (num_seqs == 0) return NMX_OK;
  NMX_CHECK(partition_size >= 64 && partition_size % 64 == 0 && max_num_partitions >= 1, NMX_ERR_INVALID_ARG,
            "paged_attention_v2_reduce: partition size %d / partitions %d", partition_size, max_num_partitions);
  dim3 rgrid(num_heads, num_seqs);
  const size_t rsmem = (size_t)max_num_partitions * sizeof(float);
  if (dtype == NMX_F32)
    paged_attention_v2_reduce_kernel<float><<<rgrid, 64, rsmem, stream>>>((float*)out, exp_sums, max_logits, (const float*)tmp_out, seq_lens,
                                                                          max_num_partitions, head_size, nullptr, partition_size);
  else if (dtype == NMX_F16)
    paged_attention_v2_reduce_kernel<f16><<<rgrid, 64, rsmem, stream>>>((f16*)out, exp_sums, max_logits, (const f16*)tmp_out, seq_lens,
                                                                        max_num_partitions, head_size, nullptr, partition_size);
  else if (dtype == NMX_BF16)
    paged_attention_v2_reduce_kernel<bf16><<<rgrid, 64, rsmem, stream>>>((bf16*)out, exp_sums, max_logits, (const bf16*)tmp_out, seq_lens,
                                                                         max_num_partitions, head_size, nullptr, partition_size);
  else
    NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "paged_attention_v2_reduce: unsupported dtype code %d", dtype);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}
