// gptq_marlin_24_gemm: W4A16 / W8A16 GEMM on 2:4-sparse weights in the Marlin-24 layout, on the gfx950 sparse MFMA.
// Replaces csrc/quantization/marlin/sparse/marlin_24_cuda_kernel.cu (entry :1017-1125) of the reference; the kernel
// is the SP = true instantiation of marlin_kernel.h.
//
// Algorithmic bytes per call: K*N*bits/16 (kept values) + K*N/8 (2-bit positions) + groups*N*2 + 2*M*K + 2*M*N.
#include "marlin_kernel.h"

static int marlin24_common(const void* a, const int32_t* b_q_weight, const void* b_meta, const void* b_scales, void* c,
                           int64_t workspace_numel, void* scratch, int64_t scratch_bytes, int num_bits, int size_m, int size_n,
                           int size_k, int num_groups, int dtype, bool defer, int* splits_out, hipStream_t stream) {
  if (splits_out != nullptr) *splits_out = 1;
  // checks mirror marlin_24_cuda_kernel.cu:1024-1116
  NMX_CHECK(num_bits == 4 || num_bits == 8, NMX_ERR_INVALID_ARG, "num_bits must be 4 or 8. Got = %d", num_bits);
  NMX_CHECK(dtype == NMX_F16, NMX_ERR_UNSUPPORTED, "gptq_marlin_24_gemm only supports float16 activations");
  NMX_CHECK(size_k % 16 == 0, NMX_ERR_INVALID_ARG, "size_k = %d is not divisible by tile_size = 16", size_k);
  NMX_CHECK(size_k % 32 == 0, NMX_ERR_INVALID_ARG, "size_k = %d: the metadata tensor covers whole 32-row k-tiles", size_k);
  NMX_CHECK(size_n % 128 == 0, NMX_ERR_INVALID_ARG, "size_n = %d, is not divisible by min_thread_n = 128", size_n);
  NMX_CHECK(workspace_numel >= (int64_t)(size_n / 128) * 64, NMX_ERR_INVALID_ARG,
            "workspace.numel = %lld is below min_workspace_size = %d", (long long)workspace_numel, (size_n / 128) * 64);
  NMX_CHECK(num_groups >= 1, NMX_ERR_INVALID_ARG, "b_scales must have at least one row");
  int group_size = size_k;
  if (num_groups > 1) {
    NMX_CHECK(size_k % num_groups == 0, NMX_ERR_INVALID_ARG, "size_k = %d, is not divisible by b_scales.size(0) = %d",
              size_k, num_groups);
    group_size = size_k / num_groups;
    NMX_CHECK(group_size == 128, NMX_ERR_INVALID_ARG, "Unexpected groupsize = %d", group_size / 2);
  }
  NMX_CHECK(((uintptr_t)a % 16 == 0) && ((uintptr_t)b_q_weight % 16 == 0) && ((uintptr_t)b_meta % 16 == 0) &&
                ((uintptr_t)b_scales % 16 == 0) && ((uintptr_t)c % 4 == 0),
            NMX_ERR_INVALID_ARG, "gptq_marlin_24_gemm: operands must be 16-byte aligned");
  if (size_m == 0 || size_n == 0) return NMX_OK;
  GemmParams p;
  p.a = a; p.b = b_q_weight; p.meta = b_meta; p.zeros = nullptr; p.scales = b_scales; p.g_idx = nullptr; p.perm = nullptr; p.c = c;
  p.partial = nullptr; p.M = size_m; p.N = size_n; p.K = size_k; p.num_groups = num_groups; p.group_size = group_size;
  p.k_splits = 1; p.slow_act_order = 0; p.defer_reduce = defer ? 1 : 0;
  const int rc = num_bits == 4 ? launch_skinny<f16, W_INT4, true>(p, scratch, scratch_bytes, stream)
                               : launch_skinny<f16, W_INT8, true>(p, scratch, scratch_bytes, stream);
  if (rc == NMX_OK && splits_out != nullptr) *splits_out = p.k_splits;
  return rc;
}

extern "C" int nmx_gptq_marlin_24_gemm(const void* a, const int32_t* b_q_weight, const void* b_meta,
                                       const void* b_scales, void* c, int64_t workspace_numel, void* scratch,
                                       int64_t scratch_bytes, int num_bits, int size_m, int size_n, int size_k,
                                       int num_groups, int dtype, nmx_stream_t stream) {
  return marlin24_common(a, b_q_weight, b_meta, b_scales, c, workspace_numel, scratch, scratch_bytes, num_bits, size_m, size_n,
                         size_k, num_groups, dtype, false, nullptr, (hipStream_t)stream);
}

// The same GEMM with the split-K reduce left to the consumer op (see nmx_gptq_marlin_gemm_deferred).
extern "C" int nmx_gptq_marlin_24_gemm_deferred(const void* a, const int32_t* b_q_weight, const void* b_meta,
                                                const void* b_scales, void* c, int64_t workspace_numel, void* scratch,
                                                int64_t scratch_bytes, int num_bits, int size_m, int size_n, int size_k,
                                                int num_groups, int dtype, int* splits_out, nmx_stream_t stream) {
  NMX_CHECK(splits_out != nullptr, NMX_ERR_INVALID_ARG, "gptq_marlin_24_gemm_deferred: splits_out must be non-null");
  return marlin24_common(a, b_q_weight, b_meta, b_scales, c, workspace_numel, scratch, scratch_bytes, num_bits, size_m, size_n,
                         size_k, num_groups, dtype, true, splits_out, (hipStream_t)stream);
}
