// Entry points of the dense Marlin-format GEMMs (gptq_marlin_gemm, marlin_gemm, fp8_marlin_gemm) and the GPTQ->Marlin
// repack; the kernel itself lives in marlin_kernel.h (shared with the 2:4-sparse variant in marlin24_gemm.hip).
#include "marlin_kernel.h"

namespace {

int marlin_common(const void* a, const int32_t* b_q_weight, const void* b_scales, const int32_t* g_idx,
                  const int32_t* perm, void* c, int64_t workspace_numel, void* scratch, int64_t scratch_bytes,
                  int size_m, int size_n, int size_k, int kind, int num_groups, int is_k_full, int dtype,
                  hipStream_t stream, int defer_reduce = 0, int* splits_out = nullptr, void* act_out = nullptr,
                  int* act_done = nullptr, const GemmParams* norm = nullptr) {
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
  p.a = a; p.b = b_q_weight; p.meta = nullptr; p.zeros = nullptr; p.scales = b_scales; p.g_idx = g_idx; p.perm = perm; p.c = c; p.partial = nullptr;
  p.M = size_m; p.N = size_n; p.K = size_k; p.num_groups = num_groups; p.k_splits = 1; p.slow_act_order = 0;
  p.defer_reduce = defer_reduce;
  p.act_out = act_out;
  if (norm != nullptr && norm->attn_tmp != nullptr) {
    p.attn_exp_sums = norm->attn_exp_sums; p.attn_max_logits = norm->attn_max_logits; p.attn_tmp = norm->attn_tmp;
    p.attn_seq_lens = norm->attn_seq_lens; p.attn_part_size = norm->attn_part_size; p.attn_max_parts = norm->attn_max_parts;
    p.attn_heads = norm->attn_heads;
  } else if (norm != nullptr) {
    p.norm_partial = norm->norm_partial; p.norm_splits = norm->norm_splits; p.norm_res_in = norm->norm_res_in;
    p.norm_res_out = norm->norm_res_out; p.norm_weight = norm->norm_weight; p.norm_eps = norm->norm_eps;
  }
  if (splits_out != nullptr) *splits_out = 1;
  if (act_done != nullptr) *act_done = 0;
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
  NMX_CHECK(norm == nullptr || norm->attn_tmp != nullptr ||
                (kind == W_INT4 && !has_act_order &&
                 decode_norm_supported(size_m, size_n, size_k, num_groups, act_out != nullptr)),
            NMX_ERR_UNSUPPORTED, "norm-fused gptq_marlin_gemm: int4 without act-order, shapes of nmx_gptq_marlin_gemm_norm_supported");
  NMX_CHECK(norm == nullptr || norm->attn_tmp == nullptr ||
                (kind == W_INT4 && !has_act_order && act_out == nullptr &&
                 decode_attn_supported(size_m, size_n, size_k, num_groups, norm->attn_heads, 128, norm->attn_max_parts)),
            NMX_ERR_UNSUPPORTED, "attention-reduce gptq_marlin_gemm: int4 without act-order, shapes of nmx_gptq_marlin_gemm_attn_supported");
  NMX_CHECK(((uintptr_t)a % 16 == 0) && ((uintptr_t)b_q_weight % 16 == 0) && ((uintptr_t)b_scales % 16 == 0) &&
                ((uintptr_t)c % 8 == 0) && size_k % 8 == 0,
            NMX_ERR_INVALID_ARG, "marlin gemm: operands must be 16-byte aligned");
  if (size_m == 0 || size_n == 0) return NMX_OK;

  int rc;
#define NMX_DISPATCH_KIND(T)                                                                   \
  switch (kind) {                                                                              \
    case W_INT4: rc = launch_skinny<T, W_INT4>(p, scratch, scratch_bytes, stream); break;      \
    case W_INT8: rc = launch_skinny<T, W_INT8>(p, scratch, scratch_bytes, stream); break;      \
    default: rc = launch_skinny<T, W_FP8>(p, scratch, scratch_bytes, stream); break;           \
  }
  if (dtype == NMX_F16) { NMX_DISPATCH_KIND(f16) }
  else { NMX_DISPATCH_KIND(bf16) }
#undef NMX_DISPATCH_KIND
  if (splits_out != nullptr) *splits_out = p.k_splits;  // > 1 only when the reduce was deferred to the consumer
  if (act_done != nullptr) *act_done = p.act_done;
  return rc;
}

}  // namespace

// fp32 split-K partial slabs the dispatch can use for (m, n, k): the splits of the path launch_skinny() takes for a plain
// layout, and of the row-block kernel that act-order / odd group sizes fall back to (the layout is not known here)
extern "C" int64_t nmx_marlin_gemm_scratch_bytes(int size_m, int size_n, int size_k) {
  if (size_m <= 0 || size_n <= 0 || size_k <= 0) return 0;
  int splits = pick_cfg(size_m, size_n, size_k).splits;
  const DecodeCfg dc = pick_decode_cfg(size_m, size_n, size_k);
  if (dc.nw != 0) splits = std::max(splits, dc.splits);
  NmxWideCfg wc;  // channel-wise and 64-multiple groups pick the same tile shape
  if (nmx_wide_pick(size_m, size_n, size_k, 1, size_k, &wc)) splits = std::max(splits, wc.splits);
  else if (size_m > 128 && ceil_div(size_n, 256) * ceil_div(size_m, 256) >= 192) splits = std::max(splits, large_splits(size_m, size_n, size_k));
  if (nmx_wide_pick(size_m, size_n, size_k, 1, size_k, &wc, 0, true)) splits = std::max(splits, wc.splits);  // gptq_marlin_24_gemm shares this sizing
  int ds = 1;  // marlin_dma_kernel (fp16 int4; the dtype is not known here)
  if (nmx_dma_pick(size_m, size_n, size_k, 1, size_k, 0, 0, &ds)) splits = std::max(splits, ds);
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

// gptq_marlin_gemm with the split-K reduction DEFERRED to the consumer: when the dispatch splits K across workgroups the
// fp32 slabs [splits, size_m, size_n] stay at the start of `scratch`, *splits_out says how many, c is left untouched, and
// nmx_fused_add_rms_norm_splitk / nmx_silu_and_mul_splitk / nmx_rope_reshape_and_cache sum them while loading their rows.
// *splits_out == 1: c holds the result as usual.
extern "C" int nmx_gptq_marlin_gemm_deferred(const void* a, const int32_t* b_q_weight, const void* b_scales,
                                             const int32_t* g_idx, const int32_t* perm, void* c, int64_t workspace_numel,
                                             void* scratch, int64_t scratch_bytes, int size_m, int size_n, int size_k,
                                             int num_bits, int num_groups, int is_k_full, int dtype, int* splits_out,
                                             nmx_stream_t stream) {
  NMX_CHECK(num_bits == 4 || num_bits == 8, NMX_ERR_INVALID_ARG, "num_bits must be 4 or 8. Got = %d", num_bits);
  NMX_CHECK(splits_out != nullptr, NMX_ERR_INVALID_ARG, "splits_out is null");
  return marlin_common(a, b_q_weight, b_scales, g_idx, perm, c, workspace_numel, scratch, scratch_bytes, size_m,
                       size_n, size_k, num_bits == 4 ? W_INT4 : W_INT8, num_groups, is_k_full, dtype,
                       (hipStream_t)stream, 1, splits_out);
}

// gate_up projection + silu_and_mul as ONE op: act_out [size_m, size_n / 2] = silu(c[:, :size_n/2]) * c[:, size_n/2:] with
// c = gptq_marlin_gemm(...). Where the dispatch takes the wide-tile kernel without a K split the activation runs in the
// GEMM's epilogue (one launch, c untouched); everywhere else the deferred GEMM is followed by the consumer launch.
extern "C" int nmx_gptq_marlin_gemm_silu_and_mul(const void* a, const int32_t* b_q_weight, const void* b_scales,
                                                 const int32_t* g_idx, const int32_t* perm, void* c, void* act_out,
                                                 int64_t workspace_numel, void* scratch, int64_t scratch_bytes, int size_m,
                                                 int size_n, int size_k, int num_bits, int num_groups, int is_k_full,
                                                 int dtype, nmx_stream_t stream) {
  NMX_CHECK(num_bits == 4 || num_bits == 8, NMX_ERR_INVALID_ARG, "num_bits must be 4 or 8. Got = %d", num_bits);
  NMX_CHECK(act_out != nullptr && size_n % 2 == 0 && ((uintptr_t)act_out % 16 == 0) && (size_n / 2) % 8 == 0, NMX_ERR_INVALID_ARG,
            "gptq_marlin_gemm_silu_and_mul: act_out [size_m, size_n / 2] must be 16-byte aligned, size_n / 2 a multiple of 8");
  int splits = 1, done = 0;
  const int rc = marlin_common(a, b_q_weight, b_scales, g_idx, perm, c, workspace_numel, scratch, scratch_bytes, size_m,
                               size_n, size_k, num_bits == 4 ? W_INT4 : W_INT8, num_groups, is_k_full, dtype,
                               (hipStream_t)stream, 1, &splits, act_out, &done);
  if (rc != NMX_OK || done || size_m == 0 || size_n == 0) return rc;
  if (splits > 1) return nmx_silu_and_mul_splitk(act_out, reinterpret_cast<const float*>(scratch), splits, size_m, size_n / 2, dtype, stream);
  return nmx_act_and_mul(act_out, c, size_m, size_n / 2, NMX_ACT_SILU, dtype, stream);
}

// out [m, n] = scalar_t(sum_s partial[s]) in the order s = 0, 1, ...: the reduce launch of the split-K GEMMs as an op of its
// own, for consumers of a deferred GEMM that have no fused form (bit-identical to the plain GEMM's output)
extern "C" int nmx_splitk_reduce(void* out, const float* partial, int splits, int size_m, int size_n, int dtype,
                                 nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NMX_CHECK(dtype == NMX_F16 || dtype == NMX_BF16, NMX_ERR_UNSUPPORTED, "splitk_reduce: fp16 / bf16 only");
  NMX_CHECK(NMX_SPLITK_COUNT(splits) >= 1 && size_n % 4 == 0 && ((uintptr_t)out % 8 == 0) && ((uintptr_t)partial % 16 == 0), NMX_ERR_INVALID_ARG,
            "splitk_reduce: size_n %% 4 == 0, out 8-byte and partial 16-byte aligned");
  const int64_t mn4 = (int64_t)size_m * size_n / 4;
  if (mn4 == 0) return NMX_OK;
  if (dtype == NMX_F16)
    splitk_reduce_kernel<f16><<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>(reinterpret_cast<f16*>(out), partial, mn4, splits);
  else
    splitk_reduce_kernel<bf16><<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>(reinterpret_cast<bf16*>(out), partial, mn4, splits);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
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

// ---- fused_add_rms_norm + gptq_marlin_gemm as ONE launch at batch <= 4 (round 3, late; no reference counterpart: the reference
// runs fused_add_rms_norm and the GEMM one after the other, models/llama.py:205-230). The GEMM's A operand is
//   x = round(sum_s norm_partial[s]) + residual_in (rounded);  residual_out = x;  A = round(round(x * rsqrt(mean x^2 + eps)) * weight)
// i.e. nmx_fused_add_rms_norm_splitk's arithmetic thread for thread, computed in the prologue of every workgroup of
// marlin_decode_kernel while its first weight loads are in flight - one dependent launch less per norm. residual_out must not
// alias residual_in. act_out == NULL: the GEMM as nmx_gptq_marlin_gemm_deferred leaves it (c or K-split slabs + *splits_out);
// act_out != NULL: silu_and_mul of the result as nmx_gptq_marlin_gemm_silu_and_mul writes it. Results are bit-identical to the
// unfused sequence. Ask nmx_gptq_marlin_gemm_norm_supported() first: other shapes return NMX_ERR_UNSUPPORTED.
extern "C" int nmx_gptq_marlin_gemm_norm_supported(int size_m, int size_n, int size_k, int num_groups, int num_bits, int dtype,
                                                   int with_act) {
  return num_bits == 4 && (dtype == NMX_F16 || dtype == NMX_BF16) && decode_norm_supported(size_m, size_n, size_k, num_groups, with_act != 0) ? 1 : 0;
}

extern "C" int nmx_gptq_marlin_gemm_norm(const float* norm_partial, int norm_splits, const void* residual_in, void* residual_out,
                                         const void* norm_weight, float epsilon, const int32_t* b_q_weight, const void* b_scales,
                                         void* c, void* act_out, int64_t workspace_numel, void* scratch, int64_t scratch_bytes,
                                         int size_m, int size_n, int size_k, int num_bits, int num_groups, int dtype, int* splits_out,
                                         nmx_stream_t stream) {
  NMX_CHECK(num_bits == 4, NMX_ERR_UNSUPPORTED, "norm-fused gptq_marlin_gemm: num_bits must be 4. Got = %d", num_bits);
  NMX_CHECK(norm_partial != nullptr && NMX_SPLITK_COUNT(norm_splits) >= 2 && residual_in != nullptr && residual_out != nullptr &&
                residual_in != residual_out && norm_weight != nullptr && splits_out != nullptr,
            NMX_ERR_INVALID_ARG, "norm-fused gptq_marlin_gemm: >= 2 producer slabs, distinct residual buffers, weight, splits_out");
  NMX_CHECK((((uintptr_t)norm_partial | (uintptr_t)residual_in | (uintptr_t)residual_out | (uintptr_t)norm_weight) % 16) == 0,
            NMX_ERR_INVALID_ARG, "norm-fused gptq_marlin_gemm: operands must be 16-byte aligned");
  GemmParams n;
  n.norm_partial = norm_partial; n.norm_splits = norm_splits; n.norm_res_in = residual_in; n.norm_res_out = residual_out;
  n.norm_weight = norm_weight; n.norm_eps = epsilon;
  int done = 0;
  // (a: any valid 16-byte aligned pointer - the kernel never reads it)
  const int rc = marlin_common(residual_in, b_q_weight, b_scales, nullptr, nullptr, c, workspace_numel, scratch, scratch_bytes, size_m,
                               size_n, size_k, W_INT4, num_groups, 1, dtype, (hipStream_t)stream, 1, splits_out, act_out, &done, &n);
  if (rc == NMX_OK && act_out != nullptr && !done && size_m > 0 && size_n > 0)
    NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "norm-fused gate_up: the dispatch did not take the fused-activation shape");
  return rc;
}

// ---- paged_attention_v2's reduce + o_proj as ONE launch at batch <= 16 (round 3, late; no reference counterpart: the reference
// runs paged_attention_v2_reduce_kernel inside the attention op and o_proj afterwards, csrc/attention/attention_kernels.cu:567-669,
// vllm/model_executor/models/llama.py:171-172). The GEMM's A operand [size_m = sequences, size_k = heads x 128] is the v2 reduce of
// the partition results nmx_paged_attention_v2_partials left (exp_sums / max_logits [seqs, heads, max_parts], tmp_out [seqs, heads,
// max_parts, 128]); every wave of marlin_decode_kernel reduces the heads of its own K slice in its prologue with the reduce
// kernel's device function (same bits as the two ops), no reduce launch. c / scratch / *splits_out as nmx_gptq_marlin_gemm_deferred.
extern "C" int nmx_gptq_marlin_gemm_attn_supported(int size_m, int size_n, int size_k, int num_groups, int num_bits, int dtype,
                                                   int num_heads, int head_size, int max_num_partitions) {
  return num_bits == 4 && (dtype == NMX_F16 || dtype == NMX_BF16) &&
                 decode_attn_supported(size_m, size_n, size_k, num_groups, num_heads, head_size, max_num_partitions)
             ? 1 : 0;
}

extern "C" int nmx_gptq_marlin_gemm_attn(const float* exp_sums, const float* max_logits, const void* tmp_out, const int32_t* seq_lens,
                                         int partition_size, int max_num_partitions, int num_heads, int head_size,
                                         const int32_t* b_q_weight, const void* b_scales, void* c, int64_t workspace_numel,
                                         void* scratch, int64_t scratch_bytes, int size_m, int size_n, int size_k, int num_bits,
                                         int num_groups, int dtype, int* splits_out, nmx_stream_t stream) {
  NMX_CHECK(num_bits == 4 && head_size == 128, NMX_ERR_UNSUPPORTED, "attention-reduce gptq_marlin_gemm: 4 bits, head size 128");
  NMX_CHECK(exp_sums != nullptr && max_logits != nullptr && tmp_out != nullptr && seq_lens != nullptr && splits_out != nullptr &&
                partition_size >= 64 && partition_size % 64 == 0 && max_num_partitions >= 1,
            NMX_ERR_INVALID_ARG, "attention-reduce gptq_marlin_gemm: partition results, seq_lens, splits_out; partition size a multiple of 64");
  NMX_CHECK(((uintptr_t)tmp_out % 16) == 0, NMX_ERR_INVALID_ARG, "attention-reduce gptq_marlin_gemm: tmp_out must be 16-byte aligned");
  GemmParams n;
  n.attn_exp_sums = exp_sums; n.attn_max_logits = max_logits; n.attn_tmp = tmp_out; n.attn_seq_lens = seq_lens;
  n.attn_part_size = partition_size; n.attn_max_parts = max_num_partitions; n.attn_heads = num_heads;
  return marlin_common(tmp_out, b_q_weight, b_scales, nullptr, nullptr, c, workspace_numel, scratch, scratch_bytes, size_m, size_n,
                       size_k, W_INT4, num_groups, 1, dtype, (hipStream_t)stream, 1, splits_out, nullptr, nullptr, &n);
}
