/*
 * nmx.h — C ABI of libnmx_hip.so: the MI355X (gfx950) implementation of nm-vllm's quantized-linear +
 * paged-attention + KV-cache hot path.
 *
 * This is the drop-in boundary. Every entry point replaces one op of the reference's native extension
 * (`vllm._C`; schemas in csrc/torch_bindings.cpp:18-295, C++ signatures in csrc/ops.h and csrc/cache.h,
 * Python callers in vllm/_custom_ops.py). No torch types cross this ABI: plain device pointers, sizes,
 * strides (in elements unless stated), scalar parameters and a HIP stream. All functions only ENQUEUE work on
 * `stream` (no host synchronisation, no allocation) unless stated, so they are hipGraph-capturable.
 *
 * Error model: return 0 on success, a negative NMX_ERR_* code otherwise; `nmx_last_error()` returns a
 * thread-local message. The Python binding raises RuntimeError with that message, mirroring the reference where
 * every argument violation is a TORCH_CHECK -> RuntimeError (e.g. csrc/quantization/gptq_marlin/gptq_marlin.cu:1741-1843).
 *
 * Paths in comments are relative to the reference root (neuralmagic/nm-vllm @ 2025-01-17).
 */
#ifndef NMX_H_
#define NMX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nmx_stream_t; /* hipStream_t */

/* scalar dtypes of activations / outputs / unquantised KV cache */
enum { NMX_F32 = 0, NMX_F16 = 1, NMX_BF16 = 2 };
/* KV-cache storage: "auto" (= scalar dtype), "fp8"/"fp8_e4m3" (OCP e4m3fn), "fp8_e5m2" */
enum { NMX_KV_AUTO = 0, NMX_KV_FP8_E4M3 = 1, NMX_KV_FP8_E5M2 = 2 };
/* memcpy direction for nmx_swap_blocks */
enum { NMX_COPY_D2D = 0, NMX_COPY_D2H = 1, NMX_COPY_H2D = 2 };

enum {
  NMX_OK = 0,
  NMX_ERR_INVALID_ARG = -1, /* shape / size / alignment check failed (reference: TORCH_CHECK) */
  NMX_ERR_UNSUPPORTED = -2, /* unsupported head size / block size / dtype (reference: TORCH_CHECK(false, "Unsupported ...")) */
  NMX_ERR_HIP = -3          /* a HIP runtime call failed */
};

const char* nmx_last_error(void);
/* library version / build info string, e.g. "nmx 0.1 gfx950" */
const char* nmx_version(void);
/* Tuning overrides (kernel sweeps and tests only; results never depend on them). The variables of DESIGN.md §7a
 * (NMX_GEMM_CFG, NMX_GEMM_WIDE, NMX_ATTN_NW, ...) are read from the environment ONCE, when the library is loaded;
 * this call changes one afterwards (value NULL = unset). Not thread-safe against concurrent launches. No reference
 * counterpart (the reference has no such overrides). */
int nmx_tuning_set(const char* name, const char* value);

/* ------------------------------------------------------------------------------------------------------------
 * Paged attention (decode). Replaces paged_attention_v1 / paged_attention_v2
 * (csrc/attention/attention_kernels.cu:805-826, :966-990; schema csrc/torch_bindings.cpp:23-46).
 *   out          [num_seqs, num_heads, head_size]           scalar dtype
 *   query        [num_seqs, num_heads, head_size], row stride q_stride (elements)
 *   key_cache    [num_blocks, num_kv_heads, head_size/x, block_size, x],  x = 16 / sizeof(cache element)
 *   value_cache  [num_blocks, num_kv_heads, head_size, block_size]
 *   block_tables [num_seqs, max_num_blocks_per_seq] int32, seq_lens [num_seqs] int32
 *   alibi_slopes [num_heads] fp32 or NULL
 * head_size in {64,80,96,112,128,192,256}, block_size in {8,16,32}; anything else -> NMX_ERR_UNSUPPORTED.
 * Block-sparse parameters are active iff bs_vert_stride > 1 (attention_kernels.cu:822).
 * v2 partitions the sequence in 512-token partitions; tmp_out [S,H,P,D] scalar, exp_sums/max_logits [S,H,P] fp32,
 * P = ceil(max_seq_len / 512) (attention_kernels.cu:885).
 * ---------------------------------------------------------------------------------------------------------- */
int nmx_paged_attention_v1(void* out, const void* query, const void* key_cache, const void* value_cache,
                           int num_seqs, int num_heads, int num_kv_heads, int head_size, int block_size,
                           int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride, float scale,
                           const int32_t* block_tables, int max_num_blocks_per_seq, const int32_t* seq_lens,
                           int max_seq_len, const float* alibi_slopes, int dtype, int kv_dtype, float kv_scale,
                           int tp_rank, int bs_local_blocks, int bs_vert_stride, int bs_block_size,
                           int bs_head_sliding_step, nmx_stream_t stream);

int nmx_paged_attention_v2(void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query,
                           const void* key_cache, const void* value_cache, int num_seqs, int num_heads,
                           int num_kv_heads, int head_size, int block_size, int64_t q_stride,
                           int64_t kv_block_stride, int64_t kv_head_stride, float scale,
                           const int32_t* block_tables, int max_num_blocks_per_seq, const int32_t* seq_lens,
                           int max_seq_len, const float* alibi_slopes, int dtype, int kv_dtype, float kv_scale,
                           int tp_rank, int bs_local_blocks, int bs_vert_stride, int bs_block_size,
                           int bs_head_sliding_step, nmx_stream_t stream);

/* paged_attention_v1 / v2 with the output's absolute maxima as a by-product (no reference counterpart: the fp8 W8A8 path of
 * vllm/model_executor/layers/quantization/fp8.py:231-259 quantises o_proj's input with ops.scaled_fp8_quant(x, None), whose
 * first pass is this maximum). absmax: float32 [nmx_paged_attention_absmax_numel(...)], every entry written; the maximum over
 * all entries equals max |out| exactly. fp16 / bf16 queries. Consumer: nmx_scaled_fp8_quant_partials. */
int nmx_paged_attention_absmax_numel(int num_seqs, int num_heads, int num_kv_heads, int partitioned);
int nmx_paged_attention_v1_absmax(void* out, float* absmax, const void* query, const void* key_cache, const void* value_cache,
                                  int num_seqs, int num_heads, int num_kv_heads, int head_size, int block_size,
                                  int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride, float scale,
                                  const int32_t* block_tables, int max_num_blocks_per_seq, const int32_t* seq_lens,
                                  int max_seq_len, const float* alibi_slopes, int dtype, int kv_dtype, float kv_scale,
                                  int tp_rank, int bs_local_blocks, int bs_vert_stride, int bs_block_size,
                                  int bs_head_sliding_step, nmx_stream_t stream);
int nmx_paged_attention_v2_absmax(void* out, float* absmax, float* exp_sums, float* max_logits, void* tmp_out,
                                  const void* query, const void* key_cache, const void* value_cache, int num_seqs,
                                  int num_heads, int num_kv_heads, int head_size, int block_size, int64_t q_stride,
                                  int64_t kv_block_stride, int64_t kv_head_stride, float scale, const int32_t* block_tables,
                                  int max_num_blocks_per_seq, const int32_t* seq_lens, int max_seq_len,
                                  const float* alibi_slopes, int dtype, int kv_dtype, float kv_scale, int tp_rank,
                                  int bs_local_blocks, int bs_vert_stride, int bs_block_size, int bs_head_sliding_step,
                                  nmx_stream_t stream);

/* paged_attention_v2 with the caller's choice of partition size (no reference counterpart: the reference fixes 512 tokens,
 * csrc/attention/attention_kernels.cu:847, vllm/attention/ops/paged_attn.py:17). At small batch 512-token partitions leave
 * most CUs idle; nmx_paged_attention_partition_size() answers the size this library recommends for a launch (512, or 256 / 128
 * while the finer split still fits one round of workgroups), and nmx_paged_attention_v2_ps() runs the same kernels with it.
 * exp_sums / max_logits: float32 [num_seqs, num_heads, P], tmp_out: [num_seqs, num_heads, P, head_size] in the query dtype,
 * P = ceil(max_seq_len / partition_size); partition_size 64 .. 512, a multiple of 64; absmax as above, or NULL. */
int nmx_paged_attention_partition_size(int num_seqs, int num_heads, int num_kv_heads, int max_seq_len);
int nmx_paged_attention_v2_ps(void* out, float* absmax, float* exp_sums, float* max_logits, void* tmp_out, const void* query,
                              const void* key_cache, const void* value_cache, int num_seqs, int num_heads, int num_kv_heads,
                              int head_size, int block_size, int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                              float scale, const int32_t* block_tables, int max_num_blocks_per_seq, const int32_t* seq_lens,
                              int max_seq_len, const float* alibi_slopes, int dtype, int kv_dtype, float kv_scale, int tp_rank,
                              int bs_local_blocks, int bs_vert_stride, int bs_block_size, int bs_head_sliding_step,
                              int partition_size, int* counters, nmx_stream_t stream);
/* counters (may be NULL): int32 [nmx_paged_attention_counters_numel()], all zero before the call (and all zero again after it),
 * not shared with another launch in flight: the last partition workgroup to finish a (sequence, kv head) then reduces the
 * partitions itself and no reduce kernel is launched (fp16 / bf16 queries; same output bits). */
int64_t nmx_paged_attention_counters_numel(int num_seqs, int num_heads, int num_kv_heads);

/* paged_attention_v2 in two halves (no reference counterpart): the partition launch alone - exp_sums / max_logits / tmp_out as
 * for nmx_paged_attention_v2_ps, nothing reduced - and the reduce (csrc/attention/attention_kernels.cu:567-669) as an op of its own.
 * nmx_gptq_marlin_gemm_attn (below) takes the partition results directly. */
int nmx_paged_attention_v2_partials(float* exp_sums, float* max_logits, void* tmp_out, const void* query, const void* key_cache,
                                    const void* value_cache, int num_seqs, int num_heads, int num_kv_heads, int head_size,
                                    int block_size, int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride, float scale,
                                    const int32_t* block_tables, int max_num_blocks_per_seq, const int32_t* seq_lens, int max_seq_len,
                                    const float* alibi_slopes, int dtype, int kv_dtype, float kv_scale, int tp_rank,
                                    int bs_local_blocks, int bs_vert_stride, int bs_block_size, int bs_head_sliding_step,
                                    int partition_size, nmx_stream_t stream);
int nmx_paged_attention_v2_reduce(void* out, const float* exp_sums, const float* max_logits, const void* tmp_out,
                                  const int32_t* seq_lens, int num_seqs, int num_heads, int head_size, int max_num_partitions,
                                  int partition_size, int dtype, nmx_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * KV-cache ops. Replace csrc/cache_kernels.cu (schema csrc/torch_bindings.cpp:207-244, csrc/cache.h:8-32).
 * ---------------------------------------------------------------------------------------------------------- */
/* context_attention_fwd (vllm/attention/ops/prefix_prefill.py:674-812, Triton in the reference): prefill attention of
 * the new tokens of every sequence over its paged context (k_cache [NB, Hkv, D/x, BS, x] with x = 8, v_cache
 * [NB, Hkv, D, BS], block table b_loc [batch, bloc_stride]) plus, causally, the new tokens themselves (k, v
 * [tokens, Hkv, D]). q / out [tokens, H, D]; b_start_loc = first token of each sequence in q, b_seq_len = context +
 * new tokens, b_ctx_len = context tokens (all int32). Strides in elements (token, head). sliding_window <= 0 = off;
 * alibi_slopes [H] fp32 or NULL. sm_scale = 1 / sqrt(D) in the reference. float16 / bfloat16. */
int nmx_context_attention_fwd(void* out, const void* q, const void* k, const void* v, const void* k_cache,
                              const void* v_cache, const int32_t* b_loc, const int32_t* b_start_loc,
                              const int32_t* b_seq_len, const int32_t* b_ctx_len, const float* alibi_slopes, int batch,
                              int num_heads, int num_kv_heads, int head_size, int block_size, int x, int64_t q_stride_t,
                              int64_t q_stride_h, int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t,
                              int64_t v_stride_h, int64_t o_stride_t, int64_t o_stride_h, int64_t kc_stride_b,
                              int64_t kc_stride_h, int64_t vc_stride_b, int64_t vc_stride_h, int64_t bloc_stride,
                              int max_input_len, int sliding_window, float sm_scale, int dtype, nmx_stream_t stream);

/* reshape_and_cache (cache_kernels.cu:253-278). key/value [num_tokens, num_heads, head_size] with row strides in
 * elements; slot_mapping [num_tokens] int64, negative = padding token (skipped). fp8: stores fp8(val / kv_scale). */
int nmx_reshape_and_cache(const void* key, const void* value, void* key_cache, void* value_cache,
                          const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size, int block_size,
                          int x, int64_t key_stride, int64_t value_stride, int dtype, int kv_dtype, float kv_scale,
                          nmx_stream_t stream);

/* reshape_and_cache_flash (cache_kernels.cu:280-314): caches are [num_blocks, block_size, num_heads, head_size]. */
int nmx_reshape_and_cache_flash(const void* key, const void* value, void* k_cache, void* v_cache,
                                const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
                                int block_size, int64_t block_stride, int64_t key_stride, int64_t value_stride,
                                int elem_size, nmx_stream_t stream);

/* copy_blocks (cache_kernels.cu:101-148). key_cache_ptrs / value_cache_ptrs: DEVICE arrays of num_layers base
 * pointers; block_mapping: DEVICE [num_pairs, 2] int64 (src, dst); block_bytes = bytes of one block of one layer. */
int nmx_copy_blocks(void* const* key_cache_ptrs, void* const* value_cache_ptrs, const int64_t* block_mapping,
                    int num_layers, int num_pairs, int64_t block_bytes, nmx_stream_t stream);

/* swap_blocks (cache_kernels.cu:24-63). block_mapping: HOST [num_pairs, 2] int64. One async copy per pair. */
int nmx_swap_blocks(const void* src, void* dst, const int64_t* block_mapping_host, int num_pairs,
                    int64_t block_bytes, int copy_kind, nmx_stream_t stream);

/* convert_fp8 (cache_kernels.cu:339-389). to_fp8 != 0: dst u8 = fp8(src / scale), src has `dtype`;
 * else dst (`dtype`) = float(fp8 src) * scale. */
int nmx_convert_fp8(void* dst, const void* src, int64_t numel, float scale, int dtype, int kv_dtype, int to_fp8,
                    nmx_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Marlin-format W4A16 / W8A16 GEMMs.
 * ---------------------------------------------------------------------------------------------------------- */
/* gptq_marlin_repack (csrc/quantization/gptq_marlin/gptq_marlin_repack.cu:276-348).
 * b_q_weight [size_k / pack, size_n] int32 (GPTQ) -> out [size_k / 16, size_n * 16 / pack] int32 (Marlin).
 * perm [size_k] int32 row gather (act-order) or NULL. Bit-exact with the reference. */
int nmx_gptq_marlin_repack(const int32_t* b_q_weight, const int32_t* perm, int32_t* out, int size_k, int size_n,
                           int num_bits, nmx_stream_t stream);

/* Bytes of device scratch the Marlin-family GEMMs need for (size_m, size_n, size_k): split-K partial sums and the
 * act-order column-permuted copy of A. The caller owns the scratch (allocated once, outside graph capture). */
int64_t nmx_marlin_gemm_scratch_bytes(int size_m, int size_n, int size_k);

/* gptq_marlin_gemm (csrc/quantization/gptq_marlin/gptq_marlin.cu:1735-1868).
 * c [size_m, size_n] = a [size_m, size_k] * dequant(b_q_weight), W[k,n] = (q - 2^(bits-1)) * s[g(k), n].
 * b_scales [num_groups, size_n] (Marlin-permuted, scalar dtype). g_idx/perm: [size_k] int32 or NULL (both).
 * workspace_numel is only validated (>= size_n / 64 * 16, gptq_marlin.cu:1833-1843); the lock words are not used. */
int nmx_gptq_marlin_gemm(const void* a, const int32_t* b_q_weight, const void* b_scales, const int32_t* g_idx,
                         const int32_t* perm, void* c, int64_t workspace_numel, void* scratch,
                         int64_t scratch_bytes, int size_m, int size_n, int size_k, int num_bits, int num_groups,
                         int is_k_full, int dtype, nmx_stream_t stream);

/* gptq_marlin_gemm with the split-K reduction deferred to the op that consumes the output (no reference counterpart:
 * the reference runs gptq_marlin_gemm, then the next op, as two passes over the activation). When the dispatch splits K
 * across workgroups, the fp32 slabs [*splits_out, size_m, size_n] are left at the start of `scratch`, c is untouched and
 * one of the *_splitk consumers below sums them while it loads its rows; *splits_out == 1 means c holds the result.
 * Results of GEMM + consumer are bit-identical to nmx_gptq_marlin_gemm followed by the plain op. */
int nmx_gptq_marlin_gemm_deferred(const void* a, const int32_t* b_q_weight, const void* b_scales, const int32_t* g_idx,
                                  const int32_t* perm, void* c, int64_t workspace_numel, void* scratch,
                                  int64_t scratch_bytes, int size_m, int size_n, int size_k, int num_bits,
                                  int num_groups, int is_k_full, int dtype, int* splits_out, nmx_stream_t stream);
/* `splits` of the split-K consumers below and of nmx_splitk_reduce: bits 0..7 = number of slabs; NMX_SPLITK_F16 set = the slabs
 * hold fp16 instead of fp32 partial sums. The *_deferred GEMM entries report the value to pass on in *splits_out (the M > 64
 * Marlin kernels write fp16 slabs for fp16 outputs since round 3: half the slab traffic; the reference's own global reduce hands
 * fp16 partials from block to block, gptq_marlin.cu). */
#define NMX_SPLITK_F16 0x100

/* out [size_m, size_n] = scalar_t(sum_s partial[s]), s = 0, 1, ... : the split-K reduce launch as an op, for consumers of a
 * deferred GEMM without a fused form (same bits as the plain GEMM's output). */
int nmx_splitk_reduce(void* out, const float* partial, int splits, int size_m, int size_n, int dtype, nmx_stream_t stream);
/* gate_up projection + silu_and_mul as one op (LlamaMLP.forward, vllm/model_executor/models/llama.py:79-83: gate_up_proj
 * then SiluAndMul): act_out [size_m, size_n / 2] = silu(c[:, :size_n/2]) * c[:, size_n/2:], c = gptq_marlin_gemm(...), both
 * roundings of the two-op sequence kept (bit-identical). One launch where the dispatch takes the wide-tile kernel without a
 * K split (c is then NOT written); otherwise the deferred GEMM + the consumer launch. c [size_m, size_n] and scratch as for
 * nmx_gptq_marlin_gemm. */
int nmx_gptq_marlin_gemm_silu_and_mul(const void* a, const int32_t* b_q_weight, const void* b_scales, const int32_t* g_idx,
                                      const int32_t* perm, void* c, void* act_out, int64_t workspace_numel, void* scratch,
                                      int64_t scratch_bytes, int size_m, int size_n, int size_k, int num_bits,
                                      int num_groups, int is_k_full, int dtype, nmx_stream_t stream);

/* fused_add_rms_norm + gptq_marlin_gemm as ONE launch at batch <= 4 (no reference counterpart: the reference runs
 * ops.fused_add_rms_norm (csrc/layernorm_kernels.cu:258-291) and the GEMM one after the other, vllm/model_executor/models/
 * llama.py:205-230). A = fused_add_rms_norm of the producer GEMM's deferred K-split slabs:
 *   x = round(sum_s norm_partial[s]) + residual_in;  residual_out = x (must not alias residual_in);
 *   A = round(round(x * rsqrt(mean(x^2) + epsilon)) * norm_weight)
 * computed in the prologue of every workgroup, then gptq_marlin_gemm on A: act_out == NULL leaves c or K-split slabs and
 * *splits_out as nmx_gptq_marlin_gemm_deferred does, act_out != NULL writes silu_and_mul of the result as
 * nmx_gptq_marlin_gemm_silu_and_mul does. norm_splits: slab count (>= 2) | NMX_SPLITK_F16. Bit-identical to the unfused
 * sequence. nmx_gptq_marlin_gemm_norm_supported() says which shapes are served (fp16 / bf16, 4 bits, no act-order, one row by default,
 * the decode kernel's shapes); others return NMX_ERR_UNSUPPORTED. */
int nmx_gptq_marlin_gemm_norm_supported(int size_m, int size_n, int size_k, int num_groups, int num_bits, int dtype,
                                        int with_act);
int nmx_gptq_marlin_gemm_norm(const float* norm_partial, int norm_splits, const void* residual_in, void* residual_out,
                              const void* norm_weight, float epsilon, const int32_t* b_q_weight, const void* b_scales, void* c,
                              void* act_out, int64_t workspace_numel, void* scratch, int64_t scratch_bytes, int size_m,
                              int size_n, int size_k, int num_bits, int num_groups, int dtype, int* splits_out,
                              nmx_stream_t stream);
/* paged_attention_v2's reduce + o_proj as ONE launch at batch <= 16 (no reference counterpart: attention_kernels.cu:567-669 runs
 * inside the attention op, o_proj afterwards - vllm/model_executor/models/llama.py:171-172). The GEMM's A operand
 * [size_m = sequences, size_k = num_heads x 128] is the v2 reduce of what nmx_paged_attention_v2_partials left; every wave reduces
 * the heads of its own K slice in its prologue (the reduce kernel's arithmetic: same bits). c / scratch / *splits_out as
 * nmx_gptq_marlin_gemm_deferred. nmx_gptq_marlin_gemm_attn_supported() says which shapes are served (4 bits, no act-order, head size
 * 128, size_m <= 16, the decode kernel's shapes); others return NMX_ERR_UNSUPPORTED - run nmx_paged_attention_v2_reduce + the GEMM. */
int nmx_gptq_marlin_gemm_attn_supported(int size_m, int size_n, int size_k, int num_groups, int num_bits, int dtype, int num_heads,
                                        int head_size, int max_num_partitions);
int nmx_gptq_marlin_gemm_attn(const float* exp_sums, const float* max_logits, const void* tmp_out, const int32_t* seq_lens,
                              int partition_size, int max_num_partitions, int num_heads, int head_size, const int32_t* b_q_weight,
                              const void* b_scales, void* c, int64_t workspace_numel, void* scratch, int64_t scratch_bytes, int size_m,
                              int size_n, int size_k, int num_bits, int num_groups, int dtype, int* splits_out, nmx_stream_t stream);
/* fused_add_rms_norm (csrc/layernorm_kernels.cu:258-291) on x = round(sum_s partial[s]): residual += x,
 * input_out = rms_norm(residual) * weight. partial [splits, num_tokens, hidden] fp32, splits >= 2. */
int nmx_fused_add_rms_norm_splitk(void* input_out, const float* partial, int splits, void* residual, const void* weight,
                                  float epsilon, int num_tokens, int hidden_size, int dtype, nmx_stream_t stream);
/* silu_and_mul (csrc/activation_kernels.cu:12-30) on x = round(sum_s partial[s]); partial [splits, num_tokens, 2 d]. */
int nmx_silu_and_mul_splitk(void* out, const float* partial, int splits, int num_tokens, int d, int dtype,
                            nmx_stream_t stream);
/* fp8 x fp8 scaled_mm (per-tensor scales, no bias) with the K-split reduce AND the scale epilogue left to the consumer:
 * raw fp32 slabs [*splits_out, m, n] stay in `scratch`; out is complete only when *splits_out == 1. The `_scaled` consumers
 * below apply sa[0] * (sb[0] * sum) - the epilogue of scaled_mm_entry.cu / test_cutlass.py:35-47 - before rounding, and can
 * leave the per-token |max| of their output for nmx_scaled_fp8_quant_partials (absmax may be null). Bit-identical to
 * nmx_scaled_mm + the plain op. */
int nmx_scaled_mm_deferred(void* out, const void* a, const void* b, const float* a_scale, const float* b_scale, void* scratch,
                           int64_t scratch_bytes, int m, int n, int k, int64_t lda, int64_t ldb, int64_t ldc, int out_dtype,
                           int* splits_out, nmx_stream_t stream);
/* out [m, n] (row stride ldc) = out_t(sa * (sb * sum_s partial[s])), s = 0, 1, ...: the reduce + scale epilogue of a K-split
 * nmx_scaled_mm as an op of its own, for consumers of nmx_scaled_mm_deferred without a fused form (same bits as nmx_scaled_mm;
 * reference epilogue order: tests/kernels/test_cutlass.py:35-47, csrc/quantization/cutlass_w8a8/scaled_mm_entry.cu:47-100). */
int nmx_splitk_reduce_scaled(void* out, const float* partial, int splits, const float* a_scale, const float* b_scale, int m, int n,
                             int64_t ldc, int out_dtype, nmx_stream_t stream);
int nmx_fused_add_rms_norm_splitk_scaled(void* input_out, const float* partial, int splits, const float* sa, const float* sb,
                                         void* residual, const void* weight, float epsilon, int num_tokens, int hidden_size,
                                         int dtype, float* absmax, nmx_stream_t stream);
int nmx_silu_and_mul_splitk_scaled(void* out, const float* partial, int splits, const float* sa, const float* sb, int num_tokens,
                                   int d, int dtype, float* absmax, nmx_stream_t stream);
int nmx_rope_reshape_and_cache_scaled(const int64_t* positions, void* qkv, const float* partial, int splits, const float* sa,
                                      const float* sb, const void* cos_sin_cache, void* key_cache, void* value_cache,
                                      const int64_t* slot_mapping, int num_tokens, int num_heads, int num_kv_heads,
                                      int head_size, int block_size, int dtype, int kv_dtype, float kv_scale,
                                      nmx_stream_t stream);
/* rotary_embedding (NeoX, rot_dim == head_size; csrc/pos_encoding_kernels.cu:10-96) on the q and k heads of a fused
 * qkv row [q heads | k heads | v heads] followed by reshape_and_cache (csrc/cache_kernels.cu:153-278) of the rotated k
 * and the v heads, in ONE launch. splits == 1: qkv [num_tokens, (H + 2 KVH) * D] is rotated in place; splits >= 2: the
 * row is round(sum_s partial[s]) and written to qkv. cos_sin_cache [max_pos, D]; caches / slot_mapping / kv_dtype /
 * kv_scale as nmx_reshape_and_cache. */
int nmx_rope_reshape_and_cache(const int64_t* positions, void* qkv, const float* partial, int splits,
                               const void* cos_sin_cache, void* key_cache, void* value_cache,
                               const int64_t* slot_mapping, int num_tokens, int num_heads, int num_kv_heads,
                               int head_size, int block_size, int dtype, int kv_dtype, float kv_scale,
                               nmx_stream_t stream);

/* AWQ checkpoints on the Marlin-format kernels (no reference counterpart in this snapshot; the reference repacks GPTQ ->
 * Marlin at load the same way, gptq_marlin.py:330-420). One-time repack of qweight [size_k, size_n / 8] / qzeros
 * [num_groups, size_n / 8] (AWQ nibble order) / scales [num_groups, size_n] fp16 into out_q [size_k / 16, size_n * 2],
 * out_scales and out_zeros [num_groups, size_n] fp16 (zeros stored as -(1024 + z)), then
 * c = a * ((q - z) * s) with the arithmetic of awq/dequantize.cuh:17-98. Supported when the group size is a multiple of
 * 128 and size_n of 64 (nmx_awq_marlin_supported); otherwise use nmx_awq_gemm. fp16 only (as AWQ in the reference). */
int nmx_awq_marlin_supported(int size_n, int size_k, int num_groups);
int nmx_awq_marlin_repack(const int32_t* qweight, const int32_t* qzeros, const void* scales, int32_t* out_q,
                          void* out_scales, void* out_zeros, int size_k, int size_n, int num_groups, nmx_stream_t stream);
int nmx_awq_marlin_gemm(const void* a, const int32_t* q, const void* scales, const void* zeros, void* c, void* scratch,
                        int64_t scratch_bytes, int size_m, int size_n, int size_k, int num_groups, nmx_stream_t stream);
/* the same with the split-K reduce left to the consumer op (see nmx_gptq_marlin_gemm_deferred) */
int nmx_awq_marlin_gemm_deferred(const void* a, const int32_t* q, const void* scales, const void* zeros, void* c,
                                 void* scratch, int64_t scratch_bytes, int size_m, int size_n, int size_k, int num_groups,
                                 int* splits_out, nmx_stream_t stream);

/* marlin_gemm (csrc/quantization/marlin/dense/marlin_cuda_kernel.cu:1045-1136): int4, fp16, groups of 128 or
 * channel-wise, weights Marlin-packed in the checkpoint. */
int nmx_marlin_gemm(const void* a, const int32_t* b_q_weight, const void* b_scales, void* c,
                    int64_t workspace_numel, void* scratch, int64_t scratch_bytes, int size_m, int size_n,
                    int size_k, int num_groups, nmx_stream_t stream);

/* gptq_marlin_24_gemm (csrc/quantization/marlin/sparse/marlin_24_cuda_kernel.cu:1017-1125): C = A * W24, W 2:4-sparse
 * along K. b_q_weight [size_k/2/16, size_n*16/pack_factor] int32 kept values (Marlin-24 permutation,
 * marlin_24_perms.py:16-50); b_meta [size_k/32, size_n*2] int16 2-bit positions in the CUTLASS reordered layout
 * (format_24.py:21-50,171-177); b_scales [num_groups, size_n] with num_groups == 1 or size_k/128; fp16 only
 * (gptq_marlin_24.py:145-147); size_n % 128 == 0; workspace_numel >= (size_n/128)*64. Runs on the hardware sparse
 * MFMA (v_smfmac_f32_16x16x32_f16): the compressed operand is never expanded. */
int nmx_gptq_marlin_24_gemm(const void* a, const int32_t* b_q_weight, const void* b_meta, const void* b_scales,
                            void* c, int64_t workspace_numel, void* scratch, int64_t scratch_bytes, int num_bits,
                            int size_m, int size_n, int size_k, int num_groups, int dtype, nmx_stream_t stream);
/* the same with the split-K reduce left to the consumer op (see nmx_gptq_marlin_gemm_deferred) */
int nmx_gptq_marlin_24_gemm_deferred(const void* a, const int32_t* b_q_weight, const void* b_meta, const void* b_scales,
                            void* c, int64_t workspace_numel, void* scratch, int64_t scratch_bytes, int num_bits,
                            int size_m, int size_n, int size_k, int num_groups, int dtype, int* splits_out,
                                     nmx_stream_t stream);

/* fp8_marlin_gemm (csrc/quantization/fp8/fp8_marlin.cu:1212-1308): W8A16, weight bytes are e4m3fn, channel-wise
 * scales [1, size_n] (Marlin single-permuted). */
int nmx_fp8_marlin_gemm(const void* a, const int32_t* b_q_weight, const void* b_scales, void* c,
                        int64_t workspace_numel, void* scratch, int64_t scratch_bytes, int num_bits, int size_m,
                        int size_n, int size_k, int dtype, nmx_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Zero-point int4 formats consumed as stored in the checkpoint: AWQ and GPTQ (exllama). fp16 only.
 * ---------------------------------------------------------------------------------------------------------- */
/* scratch (bytes) for the split-K partial sums of nmx_awq_gemm / nmx_gptq_gemm */
int64_t nmx_zp_gemm_scratch_bytes(int m, int n);
/* awq_gemm (csrc/quantization/awq/gemm_kernels.cu:492-549): out [m, oc] = in_feats [m, k] . W,
 * W[k, 8c+j] = (nib(kernel[k,c], o_j) - nib(zeros[k/G,c], o_j)) * scaling_factors[k/G, 8c+j], o = [0,4,1,5,2,6,3,7].
 * kernel [k, oc/8], zeros [k/G, oc/8] int32, scaling_factors [k/G, oc] fp16. Argument order of the C++ op
 * (csrc/ops.h:66-68): (in, kernel, scaling_factors, zeros, split_k) — split_k is a reference tuning knob, unused. */
int nmx_awq_gemm(const void* in_feats, const int32_t* kernel, const void* scaling_factors, const int32_t* zeros,
                 void* out, void* scratch, int64_t scratch_bytes, int m, int k, int oc, int group_size,
                 nmx_stream_t stream);
/* awq_dequantize (awq/gemm_kernels.cu:436-484): out [in_c, 8 * qout_c] fp16 */
int nmx_awq_dequantize(const int32_t* kernel, const void* scaling_factors, const int32_t* zeros, void* out, int in_c,
                       int qout_c, int group_size, nmx_stream_t stream);
/* gptq_gemm (csrc/quantization/gptq/q_gemm.cu:1823-1848), 4-bit: c [m, n] = a [m, k] . W,
 * W[k, n] = (q - (z + 1)) * scales[g(k), n]. qweight [k/8, n], qzeros [groups, n/8], scales [groups, n] fp16.
 * use_exllama != 0: qweight went through nmx_gptq_shuffle and g_idx (if not NULL) is the row permutation applied
 * there (vllm/model_executor/layers/quantization/gptq.py:212-222); else g_idx is the per-row group index or NULL. */
int nmx_gptq_gemm(const void* a, const int32_t* qweight, const int32_t* qzeros, const void* scales,
                  const int32_t* g_idx, void* c, void* scratch, int64_t scratch_bytes, int m, int n, int k, int groups,
                  int use_exllama, int bit, nmx_stream_t stream);
/* gptq_shuffle (q_gemm.cu:1850-1858), 4-bit, in place. q_perm [size_k] or NULL; tmp: same size as q_weight, needed
 * only with q_perm. */
int nmx_gptq_shuffle(int32_t* q_weight, int32_t* tmp, const int32_t* q_perm, int size_k, int size_n, int bit,
                     nmx_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Activation quantisation and the W8A8 scaled GEMM.
 * ---------------------------------------------------------------------------------------------------------- */
/* static_scaled_fp8_quant / dynamic_scaled_fp8_quant (csrc/quantization/fp8/common.cu:129-165).
 * out u8 = e4m3fn(clamp(x * (1 / scale), +-448)). dynamic != 0: *scale is first set to max|x| / 448 over the whole
 * tensor (whatever it held before: no zero-initialisation needed); that path needs >= 256 bytes of device scratch
 * (per-workgroup maxima), ignored otherwise. */
int nmx_scaled_fp8_quant(void* out, const void* input, float* scale, float* scratch, int64_t scratch_bytes,
                         int64_t numel, int dtype, int dynamic, nmx_stream_t stream);
/* dynamic_scaled_fp8_quant from per-workgroup maxima the producer of `input` left behind (round 2): partials[0 .. nparts)
 * = max|input| over disjoint pieces covering the tensor (nmx_rms_norm_absmax & co. write one per token). One launch;
 * *scale and the codes are bit-identical to nmx_scaled_fp8_quant(dynamic). Optional fusion of Fp8LinearMethod.apply's
 * `ops.scaled_fp8_quant(x, None)` (vllm/model_executor/layers/quantization/fp8.py:340-359) with the op before it. */
int nmx_scaled_fp8_quant_partials(void* out, const void* input, float* scale, const float* partials, int nparts,
                                  int64_t numel, int dtype, nmx_stream_t stream);
/* static_scaled_int8_quant / dynamic_scaled_int8_quant (csrc/quantization/compressed_tensors/int8_quant_kernels.cu:75-115).
 * static: int8(rn(x / scales[0])); dynamic: per token scales[t] = max|x_t| / 127, int8(rn(x * 127 / max|x_t|)). */
int nmx_scaled_int8_quant(void* out, const void* input, float* scales, int num_tokens, int hidden_size, int dtype,
                          int dynamic, nmx_stream_t stream);
/* cutlass_scaled_mm (csrc/quantization/cutlass_w8a8/scaled_mm_entry.cu:47-100):
 * out [m, n] = cast(a_scale (.) (A . B) (.) b_scale) (+ bias). a [m, k] row-major (row stride lda bytes), b given
 * column-major: pointer to B^T [n, k] rows (row stride ldb bytes); fp8 e4m3fn (is_fp8 != 0) or int8. Scales fp32,
 * numel 1 or m / n. ldc in elements. bias [n] in the output dtype or NULL. */
int nmx_scaled_mm(void* out, const void* a, const void* b, const float* a_scales, int a_scales_numel,
                  const float* b_scales, int b_scales_numel, const void* bias, void* scratch, int64_t scratch_bytes,
                  int m, int n, int k, int64_t lda, int64_t ldb, int64_t ldc, int is_fp8, int out_dtype,
                  nmx_stream_t stream);
/* Bytes of split-K scratch (fp32 / int32 partial slabs) nmx_scaled_mm can use for this shape; 0 when it does not split.
 * The caller owns the buffer (allocated outside graph capture); with less the kernel uses fewer splits. */
int64_t nmx_scaled_mm_scratch_bytes(int m, int n, int k);
/* cutlass_scaled_mm_supports_fp8 (scaled_mm_entry.cu:25-45): always 1 on gfx950 (native OCP fp8 MFMA). */
int nmx_scaled_mm_supports_fp8(int capability);

/* ------------------------------------------------------------------------------------------------------------
 * Element-wise neighbours of the GEMMs on the decode path.
 * ---------------------------------------------------------------------------------------------------------- */
/* rms_norm (csrc/layernorm_kernels.cu:293-312): out = scalar(x * rsqrt(mean(x^2) + eps)) * weight, rows contiguous. */
int nmx_rms_norm(void* out, const void* input, const void* weight, float epsilon, int num_tokens, int hidden_size,
                 int dtype, nmx_stream_t stream);
/* fused_add_rms_norm (csrc/layernorm_kernels.cu:327-352): residual += input (in place), input = rms_norm(residual). */
int nmx_fused_add_rms_norm(void* input, void* residual, const void* weight, float epsilon, int num_tokens,
                           int hidden_size, int dtype, nmx_stream_t stream);
/* rotary_embedding / batched_rotary_embedding (csrc/pos_encoding_kernels.cu:124-203). In place on query / key
 * ([num_tokens, heads * head_size], row strides in elements). cos_sin_cache [max_pos, rot_dim];
 * cos_sin_cache_offsets [num_tokens] int64 or NULL (plain rotary_embedding). */
int nmx_rotary_embedding(const int64_t* positions, void* query, void* key, const void* cos_sin_cache,
                         const int64_t* cos_sin_cache_offsets, int rot_dim, int64_t query_stride, int64_t key_stride,
                         int num_tokens, int num_heads, int num_kv_heads, int head_size, int is_neox, int dtype,
                         nmx_stream_t stream);
/* activation codes for nmx_act_and_mul / nmx_activation (csrc/activation_kernels.cu) */
enum { NMX_ACT_SILU = 0, NMX_ACT_GELU = 1, NMX_ACT_GELU_TANH = 2, NMX_ACT_GELU_NEW = 3, NMX_ACT_GELU_FAST = 4,
       NMX_ACT_GELU_QUICK = 5 };
/* silu_and_mul / gelu_and_mul / gelu_tanh_and_mul: out [T, d] = ACT(in[T, :d]) * in[T, d:] */
int nmx_act_and_mul(void* out, const void* input, int num_tokens, int d, int act, int dtype, nmx_stream_t stream);
/* rms_norm / fused_add_rms_norm / act_and_mul that ALSO leave absmax[t] = max |out[t, :]| (of the rounded outputs) for
 * nmx_scaled_fp8_quant_partials: the layer norm / activation in front of an fp8 linear layer
 * (csrc/layernorm_kernels.cu:22-46,258-291, csrc/activation_kernels.cu:12-24; outputs identical to the plain entries). */
int nmx_rms_norm_absmax(void* out, const void* input, const void* weight, float epsilon, int num_tokens, int hidden_size,
                        int dtype, float* absmax, nmx_stream_t stream);
int nmx_fused_add_rms_norm_absmax(void* input, void* residual, const void* weight, float epsilon, int num_tokens,
                                  int hidden_size, int dtype, float* absmax, nmx_stream_t stream);
int nmx_act_and_mul_absmax(void* out, const void* input, int num_tokens, int d, int act, int dtype, float* absmax,
                           nmx_stream_t stream);
/* gelu_new / gelu_fast / gelu_quick: out [T, d] = ACT(in [T, d]) */
int nmx_activation(void* out, const void* input, int num_tokens, int d, int act, int dtype, nmx_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Device utilities (csrc/cuda_utils_kernels.cu).
 * ---------------------------------------------------------------------------------------------------------- */
int nmx_get_max_shared_memory_per_block_device_attribute(int device, int* value);
int nmx_get_device_attribute(int attribute, int device, int* value);

/* ------------------------------------------------------------------------------------------------------------
 * Mixture-of-experts routing. topk_softmax replaces `_moe_C.topk_softmax` (csrc/moe/topk_softmax_kernels.cu:476-507):
 * softmax over gating_output [num_tokens, num_experts] fp32, the topk largest (ties: lowest expert id) into
 * topk_weights / topk_ids [num_tokens, topk]; token_expert_indices[t][j] = j * num_tokens + t.
 * moe_align_block_size replaces csrc/moe_align_block_size_kernels.cu:108-125: the numel flat (token, k) pairs sorted by
 * expert, each expert's run padded to a multiple of block_size with the value numel; expert_ids[b] = expert of block b;
 * *num_tokens_post_pad = padded length. sorted_token_ids must hold max_sorted >= numel + num_experts * (block_size - 1).
 * ---------------------------------------------------------------------------------------------------------- */
int nmx_topk_softmax(float* topk_weights, int32_t* topk_ids, int32_t* token_expert_indices, const float* gating_output,
                     int num_tokens, int num_experts, int topk, nmx_stream_t stream);
int nmx_moe_align_block_size(const int32_t* topk_ids, int num_experts, int block_size, int numel, int32_t* sorted_token_ids,
                             int max_sorted, int32_t* expert_ids, int32_t* num_tokens_post_pad, nmx_stream_t stream);
/* Grouped fp8 GEMM over the expert-sorted (token, k) pairs - the fused_moe Triton kernel of the reference
 * (vllm/model_executor/layers/fused_moe/fused_moe.py:20-222, invoke_fused_moe_kernel :225-292) with use_fp8:
 * out[id, :n] = cast(dot(a[id / a_row_div, :k], w[e, :n, :k]) * [topk_weights[id]] * a_scale[0] * w_scale[e]) for every
 * id of every block of `block_rows` entries of sorted_token_ids (entries >= num_valid are padding), e = expert_ids[block].
 * a [a_rows, k] fp8-e4m3, w [num_experts, n, k] fp8-e4m3, out [num_valid, n] fp16 / bf16; topk_weights may be null
 * (mul_routed_weight = False). max_blocks = entries of expert_ids; blocks past num_tokens_post_padded[0] do nothing.
 * Nothing is read on the host: capturable. */
int nmx_moe_scaled_mm(void* out, const void* a, const void* w, const float* a_scale, const float* w_scale,
                      const float* topk_weights, const int32_t* sorted_token_ids, const int32_t* expert_ids,
                      const int32_t* num_tokens_post_padded, int num_valid, int a_rows, int a_row_div, int n, int k,
                      int num_experts, int block_rows, int max_blocks, int out_dtype, nmx_stream_t stream);

/* Unquantised form of nmx_moe_scaled_mm (round 3): a [rows, k] and w [E, n, k] in `dtype` (fp16 / bf16), k a multiple of 64;
 * out[id, :] = cast((a[id / a_row_div] . w[expert_of(id)]) * [topk_weights[id]]) - the reference's Triton fused_moe_kernel with
 * use_fp8 = False (vllm/model_executor/layers/fused_moe/fused_moe.py:20-292). Nothing is read on the host: capturable. */
int nmx_moe_mm(void* out, const void* a, const void* w, const float* topk_weights, const int32_t* sorted_token_ids,
               const int32_t* expert_ids, const int32_t* num_tokens_post_padded, int num_valid, int a_rows, int a_row_div, int n, int k,
               int num_experts, int block_rows, int max_blocks, int dtype, nmx_stream_t stream);

/*
 * All-reduce over the xGMI mesh for decode-sized messages. Replaces the `_C_custom_ar` ops of the reference
 * (csrc/custom_all_reduce.cu: meta_size, init_custom_ar, register_buffer, should_custom_ar, all_reduce_reg, dispose; kernels
 * csrc/custom_all_reduce.cuh:179-255, schedule choice :442-450). The host side exchanges IPC handles itself (nmx_ipc_*:
 * hipIpcGetMemHandle / hipIpcOpenMemHandle, 64-byte handles) and hands over every rank's pointers as mapped in the calling
 * process. Two schedules like the reference: one-stage (every rank reads every peer) for small messages, two-stage
 * (reduce-scatter into a peer-visible scratch behind each rank's signal block, then all-gather) above the reference's
 * thresholds; nmx_custom_ar_stages() says which, nmx_custom_ar_should() whether this communicator takes the message at all.
 * A rank's meta block = nmx_custom_ar_meta_size() bytes of flags + its two-stage scratch (nmx_custom_ar_scratch_bytes(max
 * message)); allocate it with nmx_custom_ar_alloc_meta (uncached fine-grained memory, zero-filled). A barrier that times out
 * (spin bound: nmx_custom_ar_set_spin_limit) sets an error word and the call writes nothing further; read it with
 * nmx_custom_ar_check after synchronising. nmx_custom_ar_loopback runs all ranks of one call as one launch on ONE GPU (tests).
 */
int64_t nmx_custom_ar_meta_size(void);
int nmx_custom_ar_alloc_meta(int64_t bytes, void** ptr);
int nmx_custom_ar_free_meta(void* ptr);
int nmx_custom_ar_init(void* const* signal_ptrs, int rank, int world_size, int64_t scratch_bytes, void** fa_out);
int nmx_custom_ar_set_spin_limit(void* fa, uint32_t spin_limit);
int nmx_custom_ar_register_buffer(void* fa, void* const* peer_ptrs);
int nmx_custom_ar_should(int64_t bytes, int64_t max_size, int world_size, int full_xgmi);
int nmx_custom_ar_stages(int64_t bytes, int world_size);
int64_t nmx_custom_ar_scratch_bytes(int64_t bytes, int world_size);
int nmx_custom_ar_all_reduce(void* fa, const void* inp, void* out, int64_t numel, int dtype, nmx_stream_t stream);
int nmx_custom_ar_check(void* fa, int clear, int* error_out);
int nmx_custom_ar_loopback(void* const* signal_ptrs, void* const* data_ptrs, void* const* out_ptrs, int world_size, int64_t numel,
                           int dtype, int stages, uint32_t spin_limit, nmx_stream_t stream);
int nmx_custom_ar_dispose(void* fa);
int nmx_ipc_get_mem_handle(const void* ptr, void* handle64);
int nmx_ipc_open_mem_handle(const void* handle64, void** ptr);
int nmx_ipc_close_mem_handle(void* ptr);

#ifdef __cplusplus
}
#endif
#endif /* NMX_H_ */
