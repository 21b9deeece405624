// Mixture-of-experts routing ops for gfx950: topk_softmax (replaces csrc/moe/topk_softmax_kernels.cu, `_moe_C.topk_softmax`)
// and moe_align_block_size (replaces csrc/moe_align_block_size_kernels.cu). Index / small-reduction work, latency-bound.
// The expert GEMMs themselves run on the fp8 / int8 scaled_mm kernels (quant_ops.hip) per expert - see
// neuralmagic_vllm_amd/layers/fused_moe.py.
#include <float.h>

#include "nmx_common.h"

namespace {

// One wave per token: softmax over the experts in fp32, then k rounds of arg-max (ties -> the lowest expert id, as the
// reference's block-reduce does). token_expert_indices[t][j] = j * num_tokens + t (topk_softmax_kernels.cu:148-166).
__global__ void topk_softmax_kernel(float* __restrict__ topk_weights, int32_t* __restrict__ topk_ids,
                                    int32_t* __restrict__ token_expert_indices, const float* __restrict__ gating, int num_tokens,
                                    int num_experts, int topk) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (tok >= num_tokens) return;
  const float* g = gating + (int64_t)tok * num_experts;
  constexpr int PER = 8;  // experts per lane: up to 512 experts
  float v[PER];
  float mx = -FLT_MAX;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = lane + 64 * i;
    v[i] = e < num_experts ? g[e] : -FLT_MAX;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_reduce_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = lane + 64 * i;
    v[i] = e < num_experts ? expf(v[i] - mx) : 0.f;
    sum += v[i];
  }
  sum = wave_reduce_sum(sum);
  const float inv = 1.f / sum;
  for (int j = 0; j < topk; ++j) {
    float best = -1.f;
    int best_e = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = lane + 64 * i;
      if (e < num_experts && (v[i] > best || (v[i] == best && e < best_e))) { best = v[i]; best_e = e; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const float ob = __shfl_xor(best, m, 64);
      const int oe = __shfl_xor(best_e, m, 64);
      if (ob > best || (ob == best && oe < best_e)) { best = ob; best_e = oe; }
    }
    if (lane == 0) {
      topk_weights[(int64_t)tok * topk + j] = best * inv;
      topk_ids[(int64_t)tok * topk + j] = best_e;
      token_expert_indices[(int64_t)tok * topk + j] = j * num_tokens + tok;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i)
      if (lane + 64 * i == best_e) v[i] = -1.f;  // taken
  }
}

// One workgroup: count the (token, expert) pairs per expert, pad every expert's run to a multiple of block_size, then
// scatter the flat pair ids into their expert's run (moe_align_block_size_kernels.cu:21-120). Padding slots keep `numel`.
__global__ void moe_align_block_size_kernel(const int32_t* __restrict__ topk_ids, int32_t* __restrict__ sorted_token_ids,
                                            int32_t* __restrict__ expert_ids, int32_t* __restrict__ total_tokens_post_pad,
                                            int num_experts, int block_size, int numel, int max_sorted) {
  extern __shared__ int32_t sh[];
  int32_t* cnt = sh;                       // [num_experts]
  int32_t* start = sh + num_experts;       // [num_experts + 1]
  for (int e = threadIdx.x; e < num_experts; e += blockDim.x) cnt[e] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < numel; i += blockDim.x) atomicAdd(&cnt[topk_ids[i]], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    start[0] = 0;
    for (int e = 0; e < num_experts; ++e) start[e + 1] = start[e] + (cnt[e] + block_size - 1) / block_size * block_size;
    *total_tokens_post_pad = start[num_experts];
  }
  __syncthreads();
  const int total = start[num_experts];
  for (int i = threadIdx.x; i < max_sorted; i += blockDim.x) sorted_token_ids[i] = numel;
  for (int e = threadIdx.x; e < num_experts; e += blockDim.x)
    for (int b = start[e]; b < start[e + 1]; b += block_size) expert_ids[b / block_size] = e;
  __syncthreads();
  // stable placement (pair ids ascending inside an expert's run): thread e walks the list for its experts
  for (int e = threadIdx.x; e < num_experts; e += blockDim.x) {
    int pos = start[e];
    for (int i = 0; i < numel; ++i)
      if (topk_ids[i] == e) sorted_token_ids[pos++] = i;
  }
  (void)total;
}

}  // namespace

extern "C" int nmx_topk_softmax(float* topk_weights, int32_t* topk_ids, int32_t* token_expert_indices, const float* gating_output,
                                int num_tokens, int num_experts, int topk, nmx_stream_t stream) {
  if (num_tokens == 0) return NMX_OK;
  NMX_CHECK(num_experts >= 1 && num_experts <= 512, NMX_ERR_UNSUPPORTED, "topk_softmax: 1..512 experts, got %d", num_experts);
  NMX_CHECK(topk >= 1 && topk <= num_experts, NMX_ERR_INVALID_ARG, "topk = %d must be in [1, num_experts = %d]", topk, num_experts);
  const int waves = 4;
  topk_softmax_kernel<<<ceil_div(num_tokens, waves), waves * 64, 0, (hipStream_t)stream>>>(topk_weights, topk_ids, token_expert_indices,
                                                                                            gating_output, num_tokens, num_experts, topk);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int nmx_moe_align_block_size(const int32_t* topk_ids, int num_experts, int block_size, int numel, int32_t* sorted_token_ids,
                                        int max_sorted, int32_t* expert_ids, int32_t* num_tokens_post_pad, nmx_stream_t stream) {
  NMX_CHECK(num_experts >= 1 && num_experts <= 4096 && block_size >= 1, NMX_ERR_INVALID_ARG, "moe_align_block_size: bad expert count / block size");
  NMX_CHECK(max_sorted >= numel + num_experts * (block_size - 1), NMX_ERR_INVALID_ARG,
            "sorted_token_ids holds %d entries, %d needed", max_sorted, numel + num_experts * (block_size - 1));
  const size_t smem = (size_t)(2 * num_experts + 1) * sizeof(int32_t);
  moe_align_block_size_kernel<<<1, 1024, smem, (hipStream_t)stream>>>(topk_ids, sorted_token_ids, expert_ids, num_tokens_post_pad,
                                                                      num_experts, block_size, numel, max_sorted);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}
