// AWQ weights on the Marlin-format kernels: a one-time repack of the AWQ checkpoint layout (qweight [K, N/8] packed along
// N in nibble order [0,4,1,5,2,6,3,7], qzeros [K/g, N/8], scales [K/g, N]; awq.py:104-152) into the Marlin tile layout
// plus permuted scales and fp16 zero points, and the GEMM on it (marlin_gemm_kernel<.., ZP = true>).
//
// Why: awq_gemm (zp_gemm.hip) has to transpose 8 x 8 nibble blocks on EVERY call, because an AWQ word holds 8 columns of
// one k while an MFMA operand lane needs 8 k of one column - measured 0.35-0.9 TB/s on the Llama-3-70B / TP8 shapes. The
// bit permutation is done once at weight-load time instead (AWQLinearMethod.process_weights_after_loading, the same
// place the reference repacks GPTQ -> Marlin, gptq_marlin.py:330-420), and decode then runs at the Marlin kernels' rate.
// w = (q - z) * s exactly as awq/dequantize.cuh:17-98: (1024 + q) - (1024 + z) is exact in fp16, then one rounding by s.
// The op awq_gemm itself (checkpoint layout in, vllm._custom_ops surface) is unchanged.
#include "marlin_kernel.h"
#include "marlin_wide_api.h"

namespace {

__constant__ int kAwqNibble[8] = {0, 4, 1, 5, 2, 6, 3, 7};  // column j of a group of 8 sits in nibble kAwqNibble[j]

// one thread per output int32 of the Marlin tensor (element map: SURVEY.md appendix A.2, marlin_repack_kernel)
__global__ void awq_marlin_repack_kernel(const uint32_t* __restrict__ qweight, uint32_t* __restrict__ out, int size_k, int size_n) {
  const int64_t row_words = (int64_t)size_n * 2;
  const int64_t total = (int64_t)(size_k / 16) * row_words;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int kt = idx / row_words;
  const int w = idx % row_words;
  const int ng = w / 128, wi = w % 128;
  const int i = wi >> 2, j = wi & 3;
  const int col = i >> 2, row0 = 2 * (i & 3);
  const int rows[4] = {row0, row0 + 1, row0 + 8, row0 + 9};
  const int il[8] = {0, 2, 4, 6, 1, 3, 5, 7};
  uint32_t r = 0;
#pragma unroll
  for (int pz = 0; pz < 8; ++pz) {
    const int e = il[pz];
    const int k = kt * 16 + rows[e & 3];
    const int n = ng * 64 + 16 * j + col + 8 * (e >> 2);
    const uint32_t word = qweight[(int64_t)k * (size_n / 8) + n / 8];
    r |= ((word >> (4 * kAwqNibble[n & 7])) & 0xfu) << (4 * pz);
  }
  out[idx] = r;
}

// scales -> scale_perm order (within each run of 64 columns out[8 a + b] = in[a + 8 b]); zeros -> fp16 -(1024 + z), same order
__global__ void awq_marlin_scales_kernel(const f16* __restrict__ scales, const uint32_t* __restrict__ qzeros, f16* __restrict__ out_s,
                                         f16* __restrict__ out_z, int groups, int size_n) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)groups * size_n) return;
  const int grp = idx / size_n, pos = idx % size_n;
  const int base = pos & ~63, o = pos & 63;
  const int n = base + (o >> 3) + 8 * (o & 7);
  out_s[idx] = scales[(int64_t)grp * size_n + n];
  const uint32_t word = qzeros[(int64_t)grp * (size_n / 8) + n / 8];
  const int z = (word >> (4 * kAwqNibble[n & 7])) & 0xf;
  out_z[idx] = (f16)(-(float)(1024 + z));
}

}  // namespace

extern "C" int nmx_awq_marlin_repack(const int32_t* qweight, const int32_t* qzeros, const void* scales, int32_t* out_q,
                                     void* out_scales, void* out_zeros, int size_k, int size_n, int num_groups,
                                     nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NMX_CHECK(size_k % 16 == 0 && size_n % 64 == 0, NMX_ERR_INVALID_ARG, "awq_marlin_repack: size_k %% 16 == 0 and size_n %% 64 == 0");
  NMX_CHECK(num_groups >= 1 && size_k % num_groups == 0, NMX_ERR_INVALID_ARG, "awq_marlin_repack: size_k must be a multiple of the group count");
  const int64_t total = (int64_t)(size_k / 16) * size_n * 2;
  if (total == 0) return NMX_OK;
  awq_marlin_repack_kernel<<<(unsigned)ceil_div64(total, 256), 256, 0, stream>>>((const uint32_t*)qweight, (uint32_t*)out_q, size_k, size_n);
  NMX_LAUNCH_CHECK();
  const int64_t ns = (int64_t)num_groups * size_n;
  awq_marlin_scales_kernel<<<(unsigned)ceil_div64(ns, 256), 256, 0, stream>>>((const f16*)scales, (const uint32_t*)qzeros, (f16*)out_scales,
                                                                             (f16*)out_zeros, num_groups, size_n);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int nmx_awq_marlin_supported(int size_n, int size_k, int num_groups) {
  if (num_groups < 2 || size_k % num_groups != 0) return 0;
  const int g = size_k / num_groups;
  return (g % 128 == 0 && size_k % 128 == 0 && size_n % 64 == 0 && (int64_t)size_k * size_n < (1ll << 31)) ? 1 : 0;
}

static int awq_marlin_common(const void* a, const int32_t* q, const void* scales, const void* zeros, void* c, void* scratch,
                             int64_t scratch_bytes, int size_m, int size_n, int size_k, int num_groups, bool defer,
                             int* splits_out, hipStream_t stream) {
  if (splits_out != nullptr) *splits_out = 1;
  NMX_CHECK(nmx_awq_marlin_supported(size_n, size_k, num_groups), NMX_ERR_UNSUPPORTED,
            "awq_marlin_gemm: group size must be a multiple of 128, size_n of 64 (use awq_gemm)");
  NMX_CHECK(((uintptr_t)a % 16 == 0) && ((uintptr_t)q % 16 == 0) && ((uintptr_t)scales % 16 == 0) && ((uintptr_t)zeros % 16 == 0) &&
                ((uintptr_t)c % 8 == 0), NMX_ERR_INVALID_ARG, "awq_marlin_gemm: operands must be 16-byte aligned");
  if (size_m == 0) return NMX_OK;
  NMX_CHECK((int64_t)size_m * size_k * 2 < (1ll << 31), NMX_ERR_UNSUPPORTED, "awq_marlin_gemm: activation tensor too large for 32-bit offsets");
  GemmParams p;
  p.a = a; p.b = q; p.meta = nullptr; p.scales = scales; p.zeros = zeros; p.g_idx = nullptr; p.perm = nullptr; p.c = c;
  p.M = size_m; p.N = size_n; p.K = size_k; p.num_groups = num_groups; p.group_size = size_k / num_groups;
  p.slow_act_order = 0; p.defer_reduce = defer ? 1 : 0;
  // M > 64: the wide 128 x 256 tiles (marlin_wide_kernel<ZP>, round 3) where the dense dispatch takes them (70B / TP = 8 gate_up at
  // M = 256: 40.3 vs 42.9 us)
  NmxWideCfg wc;
  if (nmx_wide_pick(size_m, size_n, size_k, num_groups, p.group_size, &wc, W_INT4) && wc.wm == 1 && size_m > 64 &&
      wc.mt == 8 && wc.wn == 4) {
    NmxWideCall call;
    call.a = a; call.b = q; call.scales = scales; call.c = c; call.scratch = scratch; call.scratch_bytes = scratch_bytes;
    call.M = size_m; call.N = size_n; call.K = size_k; call.num_groups = num_groups; call.group_size = p.group_size;
    call.kind = W_INT4; call.is_bf16 = 0; call.defer_reduce = defer ? 1 : 0;
    call.zeros = zeros;
    const int rc = nmx_wide_launch(call, wc, stream);
    if (rc == NMX_OK && splits_out != nullptr) *splits_out = call.splits_done;
    return rc;
  }
  GemmCfg cfg = pick_cfg(size_m, size_n, size_k);
  // the 64-row x 128-column tiles need more registers than two waves per SIMD leave with the zero points on top (hipcc
  // spills, and a spill of a register an in-flight load is writing is not safe): 256-column tiles instead
  if (cfg.mt == 4 && cfg.ng == 2) cfg.ng = 4;
  p.k_splits = cfg.splits;
  if (p.k_splits > 1) {  // never allocate here (graph capture): degrade to the splits that fit
    const int fit = scratch == nullptr ? 1 : (int)(scratch_bytes / ((int64_t)size_m * size_n * sizeof(float)));
    p.k_splits = std::max(1, std::min(p.k_splits, fit));
  }
  p.partial = reinterpret_cast<float*>(scratch);
  const int rc = launch_mode<f16, W_INT4, 1, false, true>(p, cfg, stream);
  if (rc != NMX_OK) return rc;
  if (splits_out != nullptr) *splits_out = p.k_splits;
  if (p.k_splits > 1 && !defer) {
    const int64_t mn4 = (int64_t)size_m * size_n / 4;
    splitk_reduce_kernel<f16><<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>(reinterpret_cast<f16*>(c), p.partial, mn4, p.k_splits);
    NMX_LAUNCH_CHECK();
  }
  return NMX_OK;
}

extern "C" int nmx_awq_marlin_gemm(const void* a, const int32_t* q, const void* scales, const void* zeros, void* c,
                                   void* scratch, int64_t scratch_bytes, int size_m, int size_n, int size_k,
                                   int num_groups, nmx_stream_t stream) {
  return awq_marlin_common(a, q, scales, zeros, c, scratch, scratch_bytes, size_m, size_n, size_k, num_groups, false, nullptr,
                           (hipStream_t)stream);
}

// The same GEMM with the split-K reduce left to the consumer (nmx_fused_add_rms_norm_splitk / nmx_silu_and_mul_splitk /
// nmx_rope_reshape_and_cache): *splits_out slabs of fp32 [M, N] are in `scratch`, c is written only when *splits_out == 1.
extern "C" int nmx_awq_marlin_gemm_deferred(const void* a, const int32_t* q, const void* scales, const void* zeros, void* c,
                                            void* scratch, int64_t scratch_bytes, int size_m, int size_n, int size_k,
                                            int num_groups, int* splits_out, nmx_stream_t stream) {
  NMX_CHECK(splits_out != nullptr, NMX_ERR_INVALID_ARG, "awq_marlin_gemm_deferred: splits_out must be non-null");
  return awq_marlin_common(a, q, scales, zeros, c, scratch, scratch_bytes, size_m, size_n, size_k, num_groups, true, splits_out,
                           (hipStream_t)stream);
}
