// TEST INFRASTRUCTURE ONLY — CPU oracle for the paged-attention + KV-cache ops.
//
// A plain C++ restatement of the reference algorithms. It is the *checker* for the HIP kernels;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it. It is never
// linked into, imported by or called from the product library.
//
// Pinning: validated in the build container against the reference's own CPU backend
// (csrc/cpu/attention.cpp, csrc/cpu/cache.cpp compiled into oracle/_ref by oracle/build_ref.py) and
// against golden vectors generated from it (tests/golden/, tests/golden/gen_golden.py).
//
// Each function cites the reference file:line it follows (paths relative to the reference root).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "numfmt.h"

using namespace orc;

namespace {

// Reads element `idx` (in elements) of a KV cache buffer and returns it as the value the kernel
// would see after conversion to scalar_t.
// auto: plain load. fp8: scalar_t(float(fp8) * kv_scale)  (csrc/quantization/fp8/nvidia/quant_utils.cuh:293-299)
inline float kv_load(const void* cache, int64_t idx, int dt, int kv, float kv_scale) {
  if (kv == KV_AUTO) return ld(cache, idx, dt);
  float f = fp8_to_float(((const uint8_t*)cache)[idx], kv);
  // fp8 -> half is exact, then (half_to_float * scale) rounded to scalar_t
  return rnd(f * kv_scale, dt);
}

struct AttnArgs {
  void* out;  // v1: [S, H, D]; v2: tmp_out [S, H, P, D]
  float* exp_sums;    // v2 only [S, H, P]
  float* max_logits;  // v2 only [S, H, P]
  const void* q;
  const void* k_cache;
  const void* v_cache;
  int num_seqs, num_heads, num_kv_heads, head_size, block_size;
  int64_t q_stride, kv_block_stride, kv_head_stride;
  float scale;
  const int32_t* block_tables;
  int max_blocks_per_seq;
  const int32_t* seq_lens;
  const float* alibi_slopes;
  int dt, kv;
  float kv_scale;
  int tp_rank, bs_local_blocks, bs_vert_stride, bs_block_size, bs_head_sliding_step;
  int partition_size;  // 0 = no partitioning (v1)
  int max_num_partitions;
};

// One (seq, head, partition) work item of paged_attention_kernel
// (csrc/attention/attention_kernels.cu:90-496).
void attn_one(const AttnArgs& a, int seq_idx, int head_idx, int partition_idx) {
  const int D = a.head_size, BS = a.block_size;
  const bool part = a.partition_size > 0;
  const int seq_len = a.seq_lens[seq_idx];
  if (part && partition_idx * a.partition_size >= seq_len) return;  // :116-119
  const int num_seq_blocks = (seq_len + BS - 1) / BS;
  const int blocks_per_part = part ? a.partition_size / BS : num_seq_blocks;
  const int start_block = part ? partition_idx * blocks_per_part : 0;
  const int end_block = std::min(start_block + blocks_per_part, num_seq_blocks);
  const int nblocks = end_block - start_block;
  const int start_tok = start_block * BS;
  const int end_tok = std::min(start_tok + nblocks * BS, seq_len);
  const int num_tokens = end_tok - start_tok;

  const int q_per_kv = a.num_heads / a.num_kv_heads;
  const int kv_head = head_idx / q_per_kv;
  const float slope = a.alibi_slopes ? a.alibi_slopes[head_idx] : 0.f;
  const int x = 16 / (a.kv == KV_AUTO ? dt_size(a.dt) : 1);  // :200
  const bool sparse = a.bs_vert_stride > 1;                   // :822

  int bs_block_offset = 0, q_bs_block_id = 0;
  if (sparse) {  // :212-225
    q_bs_block_id = (seq_len - 1) / a.bs_block_size;
    if (a.bs_head_sliding_step >= 0)
      bs_block_offset = (a.tp_rank * a.num_heads + head_idx) * a.bs_head_sliding_step + 1;
    else
      bs_block_offset = (a.tp_rank * a.num_kv_heads + kv_head) * (-a.bs_head_sliding_step) + 1;
  }
  auto attended = [&](int block_idx) {
    if (!sparse) return true;
    const int kb = block_idx * BS / a.bs_block_size;
    const bool is_remote = ((kb + bs_block_offset) % a.bs_vert_stride == 0);
    const bool is_local = (kb > q_bs_block_id - a.bs_local_blocks);
    return is_remote || is_local;
  };

  std::vector<float> qv(D);
  for (int d = 0; d < D; ++d) qv[d] = ld(a.q, seq_idx * a.q_stride + (int64_t)head_idx * D + d, a.dt);

  const int32_t* bt = a.block_tables + (int64_t)seq_idx * a.max_blocks_per_seq;
  std::vector<float> logits((size_t)nblocks * BS, 0.f);
  float qk_max = -FLT_MAX;
  for (int b = start_block; b < end_block; ++b) {
    if (!attended(b)) {  // :240-254
      for (int o = 0; o < BS; ++o) logits[(size_t)(b - start_block) * BS + o] = -FLT_MAX;
      continue;
    }
    const int64_t phys = bt[b];
    for (int o = 0; o < BS; ++o) {
      const int tok = b * BS + o;
      const int64_t base = phys * a.kv_block_stride + (int64_t)kv_head * a.kv_head_stride + (int64_t)o * x;
      float acc = 0.f;
      for (int d = 0; d < D; ++d) {
        const int64_t idx = base + (int64_t)(d / x) * BS * x + (d % x);  // :273-282
        acc += qv[d] * kv_load(a.k_cache, idx, a.dt, a.kv, a.kv_scale);
      }
      float qk = a.scale * acc;
      qk += (slope != 0) ? slope * (tok - seq_len + 1) : 0;  // :297
      const bool mask = tok >= seq_len;
      logits[tok - start_tok] = mask ? 0.f : qk;  // :303
      if (!mask) qk_max = std::max(qk_max, qk);
    }
  }
  // softmax (:332-345)
  float exp_sum = 0.f;
  for (int i = 0; i < num_tokens; ++i) {
    float v = std::exp(logits[i] - qk_max);
    logits[i] = v;
    exp_sum += v;
  }
  const float inv_sum = 1.f / (exp_sum + 1e-6f);
  for (int i = 0; i < num_tokens; ++i) logits[i] *= inv_sum;

  if (part) {  // :349-357
    const int64_t o = (int64_t)seq_idx * a.num_heads * a.max_num_partitions +
                      (int64_t)head_idx * a.max_num_partitions + partition_idx;
    a.max_logits[o] = qk_max;
    a.exp_sums[o] = exp_sum;
  }

  // P.V with probabilities cast to scalar_t (:398-400), fp32 accumulation
  std::vector<float> acc(D, 0.f);
  for (int b = start_block; b < end_block; ++b) {
    if (!attended(b)) continue;  // :387-393
    const int64_t phys = bt[b];
    const int64_t base = phys * a.kv_block_stride + (int64_t)kv_head * a.kv_head_stride;
    for (int o = 0; o < BS; ++o) {
      const int tok = b * BS + o;
      if (tok >= seq_len) continue;  // zeroed v (:420-430)
      const float p = rnd(logits[tok - start_tok], a.dt);
      for (int d = 0; d < D; ++d) {
        acc[d] += p * kv_load(a.v_cache, base + (int64_t)d * BS + o, a.dt, a.kv, a.kv_scale);
      }
    }
  }
  const int P = part ? a.max_num_partitions : 1;
  const int64_t obase = ((int64_t)seq_idx * a.num_heads + head_idx) * P * D + (int64_t)(part ? partition_idx : 0) * D;
  for (int d = 0; d < D; ++d) st(a.out, obase + d, a.dt, acc[d]);
}

}  // namespace

extern "C" {

// csrc/attention/attention_kernels.cu:805-826 (paged_attention_v1)
void orc_paged_attention_v1(void* out, const void* q, const void* k_cache, const void* v_cache, int num_seqs,
                            int num_heads, int num_kv_heads, int head_size, int block_size, int64_t q_stride,
                            int64_t kv_block_stride, int64_t kv_head_stride, float scale,
                            const int32_t* block_tables, int max_blocks_per_seq, const int32_t* seq_lens,
                            const float* alibi_slopes, int dt, int kv, float kv_scale, int tp_rank,
                            int bs_local_blocks, int bs_vert_stride, int bs_block_size,
                            int bs_head_sliding_step) {
  AttnArgs a{out, nullptr, nullptr, q, k_cache, v_cache, num_seqs, num_heads, num_kv_heads, head_size,
             block_size, q_stride, kv_block_stride, kv_head_stride, scale, block_tables, max_blocks_per_seq,
             seq_lens, alibi_slopes, dt, kv, kv_scale, tp_rank, bs_local_blocks, bs_vert_stride, bs_block_size,
             bs_head_sliding_step, 0, 1};
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int s = 0; s < num_seqs; ++s)
    for (int h = 0; h < num_heads; ++h) attn_one(a, s, h, 0);
}

// csrc/attention/attention_kernels.cu:966-990 (paged_attention_v2) + reduce kernel :567-669
void orc_paged_attention_v2(void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* q,
                            const void* k_cache, const void* v_cache, int num_seqs, int num_heads,
                            int num_kv_heads, int head_size, int block_size, int64_t q_stride,
                            int64_t kv_block_stride, int64_t kv_head_stride, float scale,
                            const int32_t* block_tables, int max_blocks_per_seq, const int32_t* seq_lens,
                            int max_seq_len, const float* alibi_slopes, int dt, int kv, float kv_scale,
                            int tp_rank, int bs_local_blocks, int bs_vert_stride, int bs_block_size,
                            int bs_head_sliding_step) {
  const int PART = 512;
  const int P = (max_seq_len + PART - 1) / PART;  // :885
  AttnArgs a{tmp_out, exp_sums, max_logits, q, k_cache, v_cache, num_seqs, num_heads, num_kv_heads, head_size,
             block_size, q_stride, kv_block_stride, kv_head_stride, scale, block_tables, max_blocks_per_seq,
             seq_lens, alibi_slopes, dt, kv, kv_scale, tp_rank, bs_local_blocks, bs_vert_stride, bs_block_size,
             bs_head_sliding_step, PART, P};
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int s = 0; s < num_seqs; ++s)
    for (int h = 0; h < num_heads; ++h)
      for (int p = 0; p < P; ++p) attn_one(a, s, h, p);

  const int D = head_size;
#pragma omp parallel for collapse(2)
  for (int s = 0; s < num_seqs; ++s)
    for (int h = 0; h < num_heads; ++h) {
      const int seq_len = seq_lens[s];
      const int np = (seq_len + PART - 1) / PART;
      const int64_t pb = ((int64_t)s * num_heads + h) * P;
      const int64_t ob = ((int64_t)s * num_heads + h) * D;
      if (np == 1) {  // :582-594 plain copy
        for (int d = 0; d < D; ++d) st(out, ob + d, dt, ld(tmp_out, pb * D + d, dt));
        continue;
      }
      float m = -FLT_MAX;
      for (int p = 0; p < np; ++p) m = std::max(m, max_logits[pb + p]);
      std::vector<float> resc(np);
      float gsum = 0.f;
      for (int p = 0; p < np; ++p) {  // :644-649
        resc[p] = exp_sums[pb + p] * std::exp(max_logits[pb + p] - m);
        gsum += resc[p];
      }
      const float inv = 1.f / (gsum + 1e-6f);  // :652
      for (int d = 0; d < D; ++d) {
        float acc = 0.f;
        for (int p = 0; p < np; ++p) acc += ld(tmp_out, (pb + p) * D + d, dt) * resc[p] * inv;  // :663-666
        st(out, ob + d, dt, acc);
      }
    }
}

// csrc/cache_kernels.cu:153-204 (reshape_and_cache_kernel)
void orc_reshape_and_cache(const void* key, const void* value, void* key_cache, void* value_cache,
                           const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
                           int block_size, int x, int64_t key_stride, int64_t value_stride, int dt, int kv,
                           float kv_scale) {
  const int n = num_heads * head_size;
  for (int64_t t = 0; t < num_tokens; ++t) {
    const int64_t slot = slot_mapping[t];
    if (slot < 0) continue;
    const int64_t b = slot / block_size, o = slot % block_size;
    for (int i = 0; i < n; ++i) {
      const int h = i / head_size, d = i % head_size;
      const int64_t tk = b * num_heads * (head_size / x) * block_size * x +
                         (int64_t)h * (head_size / x) * block_size * x + (int64_t)(d / x) * block_size * x +
                         o * x + (d % x);
      const int64_t tv = b * num_heads * head_size * block_size + (int64_t)h * head_size * block_size +
                         (int64_t)d * block_size + o;
      if (kv == KV_AUTO) {
        const int es = dt_size(dt);
        std::memcpy((char*)key_cache + tk * es, (const char*)key + (t * key_stride + i) * es, es);
        std::memcpy((char*)value_cache + tv * es, (const char*)value + (t * value_stride + i) * es, es);
      } else {  // fp8(val / kv_scale), saturating RNE (quant_utils.cuh:456-486)
        ((uint8_t*)key_cache)[tk] = float_to_fp8(ld(key, t * key_stride + i, dt) / kv_scale, kv);
        ((uint8_t*)value_cache)[tv] = float_to_fp8(ld(value, t * value_stride + i, dt) / kv_scale, kv);
      }
    }
  }
}

// csrc/cache_kernels.cu:207-237 (reshape_and_cache_flash_kernel)
void orc_reshape_and_cache_flash(const void* key, const void* value, void* k_cache, void* v_cache,
                                 const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
                                 int block_size, int64_t block_stride, int64_t key_stride, int64_t value_stride,
                                 int elem_size) {
  const int n = num_heads * head_size;
  for (int64_t t = 0; t < num_tokens; ++t) {
    const int64_t slot = slot_mapping[t];
    if (slot < 0) continue;
    const int64_t b = slot / block_size, o = slot % block_size;
    for (int i = 0; i < n; ++i) {
      const int64_t tgt = b * block_stride + o * n + i;
      std::memcpy((char*)k_cache + tgt * elem_size, (const char*)key + (t * key_stride + i) * elem_size, elem_size);
      std::memcpy((char*)v_cache + tgt * elem_size, (const char*)value + (t * value_stride + i) * elem_size, elem_size);
    }
  }
}

// csrc/cache_kernels.cu:69-94 (copy_blocks_kernel): in order of pairs, per layer
void orc_copy_blocks(void** key_caches, void** value_caches, int num_layers, const int64_t* block_mapping,
                     int num_pairs, int64_t block_bytes) {
  for (int l = 0; l < num_layers; ++l)
    for (int p = 0; p < num_pairs; ++p) {
      const int64_t src = block_mapping[2 * p], dst = block_mapping[2 * p + 1];
      std::memmove((char*)key_caches[l] + dst * block_bytes, (char*)key_caches[l] + src * block_bytes, block_bytes);
      std::memmove((char*)value_caches[l] + dst * block_bytes, (char*)value_caches[l] + src * block_bytes, block_bytes);
    }
}

// csrc/cache_kernels.cu:24-63 (swap_blocks)
void orc_swap_blocks(const void* src, void* dst, const int64_t* block_mapping, int num_pairs, int64_t block_bytes) {
  for (int p = 0; p < num_pairs; ++p)
    std::memcpy((char*)dst + block_mapping[2 * p + 1] * block_bytes,
                (const char*)src + block_mapping[2 * p] * block_bytes, block_bytes);
}

// csrc/cache_kernels.cu:318-389 (convert_fp8): to_fp8 != 0: dst(u8) = fp8(src / scale); else dst = scalar(fp8 * scale)
// kv_cache_dtype "auto" in the reference falls back to e4m3 on CUDA (kAuto conversion is e4m3, quant_utils.cuh).
void orc_convert_fp8(void* dst, const void* src, int64_t numel, float scale, int dt, int kv, int to_fp8) {
  if (to_fp8) {
    for (int64_t i = 0; i < numel; ++i) ((uint8_t*)dst)[i] = float_to_fp8(ld(src, i, dt) / scale, kv);
  } else {
    for (int64_t i = 0; i < numel; ++i) st(dst, i, dt, fp8_to_float(((const uint8_t*)src)[i], kv) * scale);
  }
}

}  // extern "C"

// ---- prefill attention over (paged context + new tokens): vllm/attention/ops/prefix_prefill.py ----------------
// context_attention_fwd (:674-812; kernels _fwd_kernel :12-247 and _fwd_kernel_alibi :437-672) read as arithmetic:
// query token i of sequence b sits at position ctx_len + i; it attends to every cached context token (K cache
// [NB, Hkv, D/x, BS, x], V cache [NB, Hkv, D, BS], located through b_loc) and to the new tokens j <= i (k, v tensors).
// Logit = sm_scale * q.k; sliding window W > 0: a key at distance >= W gets the logit -10000 (:88-104, :201-204 -
// NOT -inf, so it still takes part in max / sum exactly as in the reference); alibi: + slope * (key_pos - query_pos)
// (:552-557, :623-628). Plain softmax in fp32; probabilities are rounded to the value dtype before P.V (:141, :228).
extern "C" void orc_context_attention_fwd(void* out, const void* q, const void* k, const void* v, const void* k_cache,
                                          const void* v_cache, const int32_t* b_loc, const int32_t* b_start_loc,
                                          const int32_t* b_seq_len, const int32_t* b_ctx_len, const float* alibi_slopes,
                                          int batch, int num_heads, int num_kv_heads, int D, int block_size, int x,
                                          int64_t q_st, int64_t q_sh, int64_t k_st, int64_t k_sh, int64_t v_st,
                                          int64_t v_sh, int64_t o_st, int64_t o_sh, int64_t kc_sb, int64_t kc_sh,
                                          int64_t vc_sb, int64_t vc_sh, int64_t bloc_stride, int sliding_window,
                                          float sm_scale, int dt) {
  const int qpk = num_heads / num_kv_heads;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int b = 0; b < batch; ++b) {
    for (int h = 0; h < num_heads; ++h) {
      const int ctx = b_ctx_len[b], q_len = b_seq_len[b] - ctx, start = b_start_loc[b], kvh = h / qpk;
      const float slope = alibi_slopes ? alibi_slopes[h] : 0.f;
      std::vector<float> logit((size_t)ctx + q_len), acc(D);
      for (int i = 0; i < q_len; ++i) {
        const int qpos = ctx + i, nk = ctx + i + 1;
        float mx = -INFINITY;
        for (int j = 0; j < nk; ++j) {
          float dot = 0.f;
          if (j < ctx) {
            const int64_t base = (int64_t)b_loc[b * bloc_stride + j / block_size] * kc_sb + (int64_t)kvh * kc_sh;
            const int off = j % block_size;
            for (int d = 0; d < D; ++d)
              dot += ld(q, (int64_t)(start + i) * q_st + (int64_t)h * q_sh + d, dt) *
                     ld(k_cache, base + (int64_t)(d / x) * block_size * x + (int64_t)off * x + d % x, dt);
          } else {
            const int64_t base = (int64_t)(start + j - ctx) * k_st + (int64_t)kvh * k_sh;
            for (int d = 0; d < D; ++d)
              dot += ld(q, (int64_t)(start + i) * q_st + (int64_t)h * q_sh + d, dt) * ld(k, base + d, dt);
          }
          float l = dot * sm_scale;
          if (sliding_window > 0 && qpos - j >= sliding_window) l = -10000.f;
          if (alibi_slopes) l += slope * (float)(j - qpos);
          logit[j] = l;
          mx = std::max(mx, l);
        }
        float sum = 0.f;
        for (int j = 0; j < nk; ++j) {
          logit[j] = std::exp(logit[j] - mx);
          sum += logit[j];
        }
        std::fill(acc.begin(), acc.end(), 0.f);
        for (int j = 0; j < nk; ++j) {
          const float pj = rnd(logit[j] / sum, dt);
          if (pj == 0.f) continue;
          if (j < ctx) {
            const int64_t base = (int64_t)b_loc[b * bloc_stride + j / block_size] * vc_sb + (int64_t)kvh * vc_sh;
            const int off = j % block_size;
            for (int d = 0; d < D; ++d) acc[d] += pj * ld(v_cache, base + (int64_t)d * block_size + off, dt);
          } else {
            const int64_t base = (int64_t)(start + j - ctx) * v_st + (int64_t)kvh * v_sh;
            for (int d = 0; d < D; ++d) acc[d] += pj * ld(v, base + d, dt);
          }
        }
        for (int d = 0; d < D; ++d) st(out, (int64_t)(start + i) * o_st + (int64_t)h * o_sh + d, dt, acc[d]);
      }
    }
  }
}
