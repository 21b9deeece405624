// Activation quantisation (fp8 per-tensor static / dynamic, int8 per-tensor static / per-token dynamic) and the
// W8A8 scaled GEMM (fp8 x fp8 and int8 x int8 on MFMA) for gfx950.
// Replaces csrc/quantization/fp8/common.cu, csrc/quantization/compressed_tensors/int8_quant_kernels.cu and the
// semantics of csrc/quantization/cutlass_w8a8/scaled_mm_entry.cu (CUTLASS itself is not ported: the contraction is
// a hand-written MFMA kernel; fp8 is OCP e4m3fn, the gfx950-native format).
#include "nmx_common.h"

namespace {

// ---- fp8 per-tensor quant (fp8/common.cu:24-125) ----------------------------------------------------------------
// dynamic: scale = max|x| / 448, reduced with an integer atomicMax on the (non-negative) float bits; the caller
// provides scale initialised to <= 0 (vllm/_custom_ops.py:316 uses zeros).
template <typename T>
__global__ void fp8_absmax_kernel(float* __restrict__ scale, const T* __restrict__ x, int64_t n) {
  __shared__ float smem[17];
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(Scalar<T>::to_f32(x[i])));
  m = wave_reduce_max(m);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) smem[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) t = fmaxf(t, smem[w]);
    atomicMax(reinterpret_cast<int*>(scale), __float_as_int(t / 448.0f));  // non-negative floats order like ints
  }
}

template <typename T>
__global__ void fp8_quant_kernel(uint8_t* __restrict__ out, const T* __restrict__ x, const float* __restrict__ scale,
                                 int64_t n) {
  const float inv = 1.0f / (*scale);  // the reference multiplies by the inverted scale (common.cu:91)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float v = Scalar<T>::to_f32(x[i]) * inv;
    asm volatile("" : "+v"(v));  // keep the product a separate fp32 rounding step (no fusion into the conversion)
    v = fmaxf(-448.0f, fminf(v, 448.0f));
    out[i] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false) & 0xff);
  }
}

// ---- int8 quant (compressed_tensors/int8_quant_kernels.cu:8-71) -------------------------------------------------
__device__ __forceinline__ int8_t f32_to_i8_rn_sat(float x) {
  float r = rintf(x);
  r = fminf(fmaxf(r, -128.f), 127.f);
  return (int8_t)r;
}

template <typename T, bool DYNAMIC>
__global__ void int8_quant_kernel(int8_t* __restrict__ out, const T* __restrict__ x, float* __restrict__ scales, int hidden) {
  __shared__ float smem[17];
  const int64_t row = (int64_t)blockIdx.x * hidden;
  if constexpr (DYNAMIC) {
    float m = 0.f;
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) m = fmaxf(m, fabsf(Scalar<T>::to_f32(x[row + i])));
    m = wave_reduce_max(m);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) smem[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) t = fmaxf(t, smem[w]);
      smem[16] = t;
      scales[blockIdx.x] = t / 127.0f;
    }
    __syncthreads();
    const float ts = 127.0f / smem[16];
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) out[row + i] = f32_to_i8_rn_sat(Scalar<T>::to_f32(x[row + i]) * ts);
  } else {
    const float s = scales[0];
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) out[row + i] = f32_to_i8_rn_sat(Scalar<T>::to_f32(x[row + i]) / s);
  }
}

// ---- W8A8 scaled GEMM -------------------------------------------------------------------------------------------
// out[M,N] = cast(a_scale (.) (A . B) (.) b_scale) (+ bias). A [M,K] row-major, B column-major = Bt [N,K] row-major:
// both operands are K-contiguous, i.e. already in MFMA fragment order. Lane (g, i) of a 16-row tile loads 16
// consecutive k-bytes (k0 + 16 g ..) of its row and feeds two MFMA 16x16x32 (fp8) / one 16x16x64 (int8) k-steps.
// Workgroup: 4 waves = 4 K-slices of one 64-column x (16 MT)-row tile, reduced through LDS.
// grid (ceil(N / 64), k_splits (1 for now), ceil(M / (16 MT)))
struct MmParams {
  const uint8_t* a;
  const uint8_t* bt;
  void* out;
  const float* a_scales;
  const float* b_scales;
  const void* bias;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int a_per_row, b_per_col;
};

template <typename out_t, bool FP8, int MT>
__global__ __launch_bounds__(256) void scaled_mm_kernel(const MmParams p) {
  constexpr int NT = 4;  // 16-column tiles per wave (64 columns)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.z * 16 * MT;
  const int K = p.K;
  using acc_t = typename std::conditional<FP8, f32x4, i32x4>::type;
  acc_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[mt][t] = acc_t{0, 0, 0, 0};

  // K is split in 64-byte steps over the 4 waves (interleaved)
  const int steps = (K + 63) / 64;
  for (int s = wave; s < steps; s += 4) {
    const int k = s * 64 + 16 * g;
    const bool kok = k < K;  // K % 16 == 0 is required
    u32x4 bf[NT], af[MT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + 16 * t + li;
      bf[t] = (kok && n < p.N) ? *reinterpret_cast<const u32x4*>(p.bt + (int64_t)n * p.ldb + k) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + 16 * mt + li;
      af[mt] = (kok && m < p.M) ? *reinterpret_cast<const u32x4*>(p.a + (int64_t)m * p.lda + k) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        // weights as the MFMA "A" operand (rows = columns of the output), activations as "B": D[n][m]
        if constexpr (FP8) {
          const long b0 = (long)(((uint64_t)bf[t][1] << 32) | bf[t][0]), b1 = (long)(((uint64_t)bf[t][3] << 32) | bf[t][2]);
          const long a0 = (long)(((uint64_t)af[mt][1] << 32) | af[mt][0]), a1 = (long)(((uint64_t)af[mt][3] << 32) | af[mt][2]);
          acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b0, a0, acc[mt][t], 0, 0, 0);
          acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b1, a1, acc[mt][t], 0, 0, 0);
        } else {
          acc[mt][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, bf[t]), __builtin_bit_cast(i32x4, af[mt]),
                                                            acc[mt][t], 0, 0, 0);
        }
      }
  }

  // reduce the 4 K-slices through LDS
  extern __shared__ __attribute__((aligned(16))) char smem[];
  acc_t* red = reinterpret_cast<acc_t*>(smem);  // [3][MT][NT][64]
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

  // D layout: col = lane & 15 = activation row m, rows 4 g + r = output columns n0 + 16 t + 4 g + r
  out_t* out = reinterpret_cast<out_t*>(p.out);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + li;
    if (m >= p.M) continue;
    const float sa = p.a_scales[p.a_per_row ? m : 0];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * t + 4 * g + r;
        if (n >= p.N) continue;
        const float sb = p.b_scales[p.b_per_col ? n : 0];
        float v = sa * (sb * (float)acc[mt][t][r]);  // tests/kernels/test_cutlass.py:35-47
        out_t o = Scalar<out_t>::from_f32(v);
        if (p.bias != nullptr) o = Scalar<out_t>::from_f32(Scalar<out_t>::to_f32(o) + Scalar<out_t>::to_f32(reinterpret_cast<const out_t*>(p.bias)[n]));
        out[(int64_t)m * p.ldc + n] = o;
      }
    }
  }
}

template <typename out_t, bool FP8>
int launch_mm(const MmParams& p, hipStream_t stream) {
  const int mt = p.M <= 16 ? 1 : (p.M <= 32 ? 2 : 4);
  dim3 grid(ceil_div(p.N, 64), 1, ceil_div(p.M, 16 * mt));
  const size_t smem = (size_t)3 * mt * 4 * 64 * 16;
  switch (mt) {
    case 1: scaled_mm_kernel<out_t, FP8, 1><<<grid, 256, smem, stream>>>(p); break;
    case 2: scaled_mm_kernel<out_t, FP8, 2><<<grid, 256, smem, stream>>>(p); break;
    default: scaled_mm_kernel<out_t, FP8, 4><<<grid, 256, smem, stream>>>(p); break;
  }
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

}  // namespace

extern "C" int nmx_scaled_fp8_quant(void* out, const void* input, float* scale, int64_t numel, int dtype, int dynamic,
                                    nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (numel == 0) return NMX_OK;
  const int threads = 1024;
  const int blocks = (int)std::min<int64_t>(ceil_div64(numel, threads), 1024);
#define NMX_FP8Q(T)                                                                                        \
  do {                                                                                                     \
    if (dynamic) fp8_absmax_kernel<T><<<blocks, threads, 0, stream>>>(scale, (const T*)input, numel);      \
    fp8_quant_kernel<T><<<blocks, threads, 0, stream>>>((uint8_t*)out, (const T*)input, scale, numel);     \
  } while (0)
  switch (dtype) {
    case NMX_F32: NMX_FP8Q(float); break;
    case NMX_F16: NMX_FP8Q(f16); break;
    case NMX_BF16: NMX_FP8Q(bf16); break;
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "scaled_fp8_quant: unsupported dtype %d", dtype);
  }
#undef NMX_FP8Q
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int nmx_scaled_int8_quant(void* out, const void* input, float* scales, int num_tokens, int hidden_size,
                                     int dtype, int dynamic, nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (num_tokens == 0) return NMX_OK;
  const int threads = std::min(1024, std::max(64, ((hidden_size + 63) / 64) * 64));
#define NMX_I8Q(T)                                                                                                    \
  do {                                                                                                                \
    if (dynamic) int8_quant_kernel<T, true><<<num_tokens, threads, 0, stream>>>((int8_t*)out, (const T*)input, scales, hidden_size); \
    else int8_quant_kernel<T, false><<<num_tokens, threads, 0, stream>>>((int8_t*)out, (const T*)input, scales, hidden_size);        \
  } while (0)
  switch (dtype) {
    case NMX_F32: NMX_I8Q(float); break;
    case NMX_F16: NMX_I8Q(f16); break;
    case NMX_BF16: NMX_I8Q(bf16); break;
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "scaled_int8_quant: unsupported dtype %d", dtype);
  }
#undef NMX_I8Q
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int nmx_scaled_mm(void* out, const void* a, const void* b, const float* a_scales, int a_scales_numel,
                             const float* b_scales, int b_scales_numel, const void* bias, int m, int n, int k,
                             int64_t lda, int64_t ldb, int64_t ldc, int is_fp8, int out_dtype, nmx_stream_t stream) {
  // checks mirror cutlass_w8a8/scaled_mm_entry.cu:59-76
  NMX_CHECK(a_scales_numel == 1 || a_scales_numel == m, NMX_ERR_INVALID_ARG, "a_scales.numel() must be 1 or a.size(0)");
  NMX_CHECK(b_scales_numel == 1 || b_scales_numel == n, NMX_ERR_INVALID_ARG, "b_scales.numel() must be 1 or b.size(1)");
  NMX_CHECK(k % 16 == 0 && n % 16 == 0, NMX_ERR_INVALID_ARG, "scaled_mm: K and N must be multiples of 16");
  NMX_CHECK(ldb % 16 == 0 && ldc % 16 == 0 && lda % 16 == 0, NMX_ERR_INVALID_ARG, "scaled_mm: 16-byte row alignment required");
  NMX_CHECK(out_dtype == NMX_F16 || out_dtype == NMX_BF16, NMX_ERR_UNSUPPORTED, "scaled_mm: output must be float16 or bfloat16");
  NMX_CHECK(((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0), NMX_ERR_INVALID_ARG, "scaled_mm: operands must be 16-byte aligned");
  if (m == 0 || n == 0) return NMX_OK;
  MmParams p{(const uint8_t*)a, (const uint8_t*)b, out, a_scales, b_scales, bias, m, n, k, lda, ldb, ldc,
             a_scales_numel > 1 ? 1 : 0, b_scales_numel > 1 ? 1 : 0};
  if (out_dtype == NMX_F16) return is_fp8 ? launch_mm<f16, true>(p, (hipStream_t)stream) : launch_mm<f16, false>(p, (hipStream_t)stream);
  return is_fp8 ? launch_mm<bf16, true>(p, (hipStream_t)stream) : launch_mm<bf16, false>(p, (hipStream_t)stream);
}

extern "C" int nmx_scaled_mm_supports_fp8(int capability) {
  (void)capability;
  return 1;  // gfx950 has native OCP fp8 MFMA (cutlass_scaled_mm_supports_fp8, scaled_mm_entry.cu:25-45)
}
