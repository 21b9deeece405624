// Mixture-of-experts routing ops for gfx950: topk_softmax (replaces csrc/moe/topk_softmax_kernels.cu, `_moe_C.topk_softmax`)
// and moe_align_block_size (replaces csrc/moe_align_block_size_kernels.cu). Index / small-reduction work, latency-bound.
// The expert GEMMs themselves run on the fp8 / int8 scaled_mm kernels (quant_ops.hip) per expert - see
// neuralmagic_vllm_amd/layers/fused_moe.py.
#include <float.h>

#include <algorithm>

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

// ---- grouped fp8 GEMM over the expert-sorted rows (fused_moe.py:20-222, the reference's Triton fused_moe_kernel) --------
// out[id, :] = cast((sum_k A[id / a_row_div, k] * W[e, :, k]) * [topk_weights[id]] * a_scale * w_scale[e]) for every sorted
// pair id of block z, e = expert_ids[z]. One workgroup = one block of 16 MT sorted rows x a 64-column tile of the block's
// expert; 4 waves = 4 contiguous K slices summed through LDS (the structure of scaled_mm_lds_kernel: whole-line operand
// loads into a wave-private swizzled LDS image, no barrier in the loop, two register sets). Rows are GATHERED through
// sorted_token_ids (padding slots carry ids >= num_valid: loaded as zeros, never stored) and the result is scattered to
// row id - nothing is read back on the host, so the whole MoE layer is graph-capturable. Blocks past
// num_tokens_post_padded[0] exit at once.
struct MoeMmParams {
  const uint8_t* a;        // [rows, K] fp8
  const uint8_t* w;        // [E, N, K] fp8 (each expert's [N, K] row-major = the column-major [K, N] of scaled_mm)
  void* out;               // [num_valid, N]
  const float* a_scale;    // [1]
  const float* w_scale;    // [E]
  const float* topk_weights;  // [num_valid] or null
  const int32_t* sorted_token_ids;
  const int32_t* expert_ids;
  const int32_t* num_tokens_post_padded;
  int N, K, num_valid, a_row_div, a_rows;
};

// HALF = true (round 3): the unquantised form of the same kernel - a and w hold out_t (fp16 / bf16) elements, a 128-byte stage is
// 64 k, the MFMA is v_mfma_f32_16x16x32_{f16,bf16} on the 16-byte chunks as loaded (natural k order on both operands), and the
// epilogue is the routing weight alone (fused_moe.py:186-195 with use_fp8 = False). p.K stays the row length in ELEMENTS.
template <typename out_t, int MT, bool HALF = false>
__global__ __launch_bounds__(256, 2) void moe_scaled_mm_kernel(const MoeMmParams p) {
  constexpr int ES = HALF ? 2 : 1;  // bytes per element of a and w
  constexpr int NT = 4;
  constexpr int BROWS = 64, AROWS = 16 * MT;
  constexpr int BI = BROWS / 8, AI = AROWS / 8;
  const int blk = blockIdx.y;
  if (blk * AROWS >= p.num_tokens_post_padded[0]) return;
  const int expert = p.expert_ids[blk];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int lr = lane >> 3, lc = lane & 7;
  const int n0 = blockIdx.x * 64;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int stages = p.K * ES / 128;
  const int pw = (stages + 3) / 4;
  const int ws = min(wave * pw, stages), we = min(ws + pw, stages);
  const int len = we - ws;

  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.w + (int64_t)expert * p.N * p.K * ES), 0, (int)((int64_t)p.N * p.K * ES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.a), 0, (int)((int64_t)p.a_rows * p.K * ES), 0x00020000);
  int b_voff[BI], a_voff[AI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int n = n0 + 8 * j + lr;
    b_voff[j] = n < p.N ? (int)(n * p.K * ES + 16 * lc) : (int)0xfffffff0u;  // rows past the matrix read as zeros
  }
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int id = p.sorted_token_ids[blk * AROWS + 8 * j + lr];
    a_voff[j] = id < p.num_valid ? (int)((id / p.a_row_div) * p.K * ES + 16 * lc) : (int)0xfffffff0u;  // padding slot
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* img = smem + wave * ((BROWS + AROWS) * 128);
  auto slot = [](int r, int c) { return r * 128 + 16 * (c ^ ((r >> 1) & 7)); };

  struct Stage { u32x4 b[BI]; u32x4 a[AI]; };
  auto load = [&](int s, Stage& r) {
    const int soff = min(s, stages - 1) * 128;
#pragma unroll
    for (int j = 0; j < BI; ++j) r.b[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff[j], soff, 0);
#pragma unroll
    for (int j = 0; j < AI; ++j) r.a[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff[j], soff, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto compute = [&](const Stage& r) {
#pragma unroll
    for (int j = 0; j < BI; ++j) *reinterpret_cast<u32x4*>(img + slot(8 * j + lr, lc)) = r.b[j];
#pragma unroll
    for (int j = 0; j < AI; ++j) *reinterpret_cast<u32x4*>(img + slot(BROWS + 8 * j + lr, lc)) = r.a[j];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      u32x4 bf[NT], af[MT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bf[t] = *reinterpret_cast<const u32x4*>(img + slot(16 * t + li, 4 * q + g));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const u32x4*>(img + slot(BROWS + 16 * mt + li, 4 * q + g));
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if constexpr (HALF) {
            if constexpr (__is_same(out_t, f16))
              acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bf[t]), __builtin_bit_cast(f16x8, af[mt]), acc[mt][t], 0, 0, 0);
            else
              acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[t]), __builtin_bit_cast(bf16x8, af[mt]), acc[mt][t], 0, 0, 0);
            continue;
          }
          const long b0 = (long)(((uint64_t)bf[t][1] << 32) | bf[t][0]), b1 = (long)(((uint64_t)bf[t][3] << 32) | bf[t][2]);
          const long a0 = (long)(((uint64_t)af[mt][1] << 32) | af[mt][0]), a1 = (long)(((uint64_t)af[mt][3] << 32) | af[mt][2]);
          acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b0, a0, acc[mt][t], 0, 0, 0);
          acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b1, a1, acc[mt][t], 0, 0, 0);
        }
    }
  };
  if (len > 0) {
    Stage r0, r1;
    load(ws, r0);
    load(ws + 1, r1);
    for (int j = 0; j < len; j += 2) {
      compute(r0);
      load(ws + j + 2, r0);
      if (j + 1 < len) compute(r1);
      load(ws + j + 3, r1);
    }
  }
  __syncthreads();
  f32x4* red = reinterpret_cast<f32x4*>(smem);  // [3][MT][NT][64]
  if (wave > 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < NT; ++t) red[(((wave - 1) * MT + mt) * NT + t) * 64 + lane] = acc[mt][t];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[mt][t] += red[((w * MT + mt) * NT + t) * 64 + lane];
  const float sa = HALF ? 1.0f : p.a_scale[0], sb = HALF ? 1.0f : p.w_scale[expert];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int id = p.sorted_token_ids[blk * AROWS + 16 * mt + li];
    if (id >= p.num_valid) continue;
    const float rw = p.topk_weights != nullptr ? p.topk_weights[id] : 1.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + 16 * t + 4 * g;
      if (n >= p.N) continue;
      union { out_t h[4]; u32x2 u; } o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[mt][t][r];
        if (p.topk_weights != nullptr) v = v * rw;  // fused_moe.py:186-190: routed weight first, then the fp8 scales
        o.h[r] = Scalar<out_t>::from_f32(HALF ? v : v * sa * sb);  // :192-195
      }
      *reinterpret_cast<u32x2*>(reinterpret_cast<out_t*>(p.out) + (int64_t)id * p.N + n) = o.u;
    }
  }
}

template <typename out_t, bool HALF = false>
int launch_moe_mm(const MoeMmParams& p, int block_rows, int max_blocks, hipStream_t stream) {
  dim3 grid(ceil_div(p.N, 64), max_blocks, 1);
#define NMX_MOE(MT_)                                                                                   \
  {                                                                                                    \
    const size_t smem = std::max((size_t)4 * (64 + 16 * MT_) * 128, (size_t)3 * MT_ * 4 * 64 * 16);     \
    moe_scaled_mm_kernel<out_t, MT_, HALF><<<grid, 256, smem, stream>>>(p);                            \
  }
  if (block_rows == 16) NMX_MOE(1) else if (block_rows == 32) NMX_MOE(2) else NMX_MOE(4)
#undef NMX_MOE
  NMX_LAUNCH_CHECK();
  return NMX_OK;
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

extern "C" int nmx_moe_scaled_mm(void* out, const void* a, const void* w, const float* a_scale, const float* w_scale,
                                 const float* topk_weights, const int32_t* sorted_token_ids, const int32_t* expert_ids,
                                 const int32_t* num_tokens_post_padded, int num_valid, int a_rows, int a_row_div, int n, int k,
                                 int num_experts, int block_rows, int max_blocks, int out_dtype, nmx_stream_t stream) {
  NMX_CHECK(out_dtype == NMX_F16 || out_dtype == NMX_BF16, NMX_ERR_UNSUPPORTED, "moe_scaled_mm: float16 / bfloat16 output only");
  NMX_CHECK(block_rows == 16 || block_rows == 32 || block_rows == 64, NMX_ERR_INVALID_ARG, "moe_scaled_mm: block size 16, 32 or 64, got %d",
            block_rows);
  NMX_CHECK(k % 128 == 0 && n % 4 == 0, NMX_ERR_UNSUPPORTED, "moe_scaled_mm: K must be a multiple of 128 and N of 4 (K = %d, N = %d)", k, n);
  NMX_CHECK(a_row_div >= 1 && num_experts >= 1 && (int64_t)n * k < (1ll << 31) && (int64_t)a_rows * k < (1ll << 31), NMX_ERR_INVALID_ARG,
            "moe_scaled_mm: bad shape");
  NMX_CHECK(((uintptr_t)a | (uintptr_t)w) % 16 == 0 && (uintptr_t)out % 8 == 0, NMX_ERR_INVALID_ARG, "moe_scaled_mm: operands must be 16-byte aligned");
  if (num_valid == 0 || max_blocks == 0) return NMX_OK;
  MoeMmParams p;
  p.a = (const uint8_t*)a; p.w = (const uint8_t*)w; p.out = out; p.a_scale = a_scale; p.w_scale = w_scale; p.topk_weights = topk_weights;
  p.sorted_token_ids = sorted_token_ids; p.expert_ids = expert_ids; p.num_tokens_post_padded = num_tokens_post_padded;
  p.N = n; p.K = k; p.num_valid = num_valid; p.a_row_div = a_row_div; p.a_rows = a_rows;
  if (out_dtype == NMX_F16) return launch_moe_mm<f16>(p, block_rows, max_blocks, (hipStream_t)stream);
  return launch_moe_mm<bf16>(p, block_rows, max_blocks, (hipStream_t)stream);
}

// The unquantised grouped GEMM of fused_moe (fused_moe.py:20-222 with use_fp8 = False; round 3): a [rows, K] and w [E, N, K] in the
// output dtype (fp16 / bf16), out[id, :] = cast((sum_k a[id / a_row_div, k] * w[e, :, k]) * [topk_weights[id]]). Same blocks,
// gather / scatter and graph-capturability as nmx_moe_scaled_mm.
extern "C" int nmx_moe_mm(void* out, const void* a, const void* w, const float* topk_weights, const int32_t* sorted_token_ids,
                          const int32_t* expert_ids, const int32_t* num_tokens_post_padded, int num_valid, int a_rows, int a_row_div,
                          int n, int k, int num_experts, int block_rows, int max_blocks, int dtype, nmx_stream_t stream) {
  NMX_CHECK(dtype == NMX_F16 || dtype == NMX_BF16, NMX_ERR_UNSUPPORTED, "moe_mm: float16 / bfloat16 only");
  NMX_CHECK(block_rows == 16 || block_rows == 32 || block_rows == 64, NMX_ERR_INVALID_ARG, "moe_mm: block size 16, 32 or 64, got %d", block_rows);
  NMX_CHECK(k % 64 == 0 && n % 4 == 0, NMX_ERR_UNSUPPORTED, "moe_mm: K must be a multiple of 64 and N of 4 (K = %d, N = %d)", k, n);
  NMX_CHECK(a_row_div >= 1 && num_experts >= 1 && (int64_t)n * k * 2 < (1ll << 31) && (int64_t)a_rows * k * 2 < (1ll << 31), NMX_ERR_INVALID_ARG,
            "moe_mm: bad shape");
  NMX_CHECK(((uintptr_t)a | (uintptr_t)w) % 16 == 0 && (uintptr_t)out % 8 == 0, NMX_ERR_INVALID_ARG, "moe_mm: operands must be 16-byte aligned");
  if (num_valid == 0 || max_blocks == 0) return NMX_OK;
  MoeMmParams p;
  p.a = (const uint8_t*)a; p.w = (const uint8_t*)w; p.out = out; p.a_scale = nullptr; p.w_scale = nullptr; p.topk_weights = topk_weights;
  p.sorted_token_ids = sorted_token_ids; p.expert_ids = expert_ids; p.num_tokens_post_padded = num_tokens_post_padded;
  p.N = n; p.K = k; p.num_valid = num_valid; p.a_row_div = a_row_div; p.a_rows = a_rows;
  if (dtype == NMX_F16) return launch_moe_mm<f16, true>(p, block_rows, max_blocks, (hipStream_t)stream);
  return launch_moe_mm<bf16, true>(p, block_rows, max_blocks, (hipStream_t)stream);
}
