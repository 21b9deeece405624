// Element-wise neighbours of the hot GEMMs on the decode path: RMSNorm (+ fused residual add), rotary embedding,
// gated activations. Replaces csrc/layernorm_kernels.cu, csrc/pos_encoding_kernels.cu, csrc/activation_kernels.cu of
// the reference. One workgroup per token; 16-byte vector loads wherever the row allows it.
// Rounding follows the reference's scalar_t arithmetic step by step (every scalar_t operation rounds to scalar_t),
// so results are bit-comparable with the CPU oracle.
#include "nmx_common.h"

namespace {

// max over the workgroup, written by thread 0 (dynamic fp8 quantisation: the producer of an activation leaves one |max|
// per workgroup for nmx_scaled_fp8_quant_partials instead of a separate absmax launch re-reading the tensor)
__device__ __forceinline__ void block_max_store(float v, float* smem, float* dst) {
  v = wave_reduce_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();  // smem may still be read by block_sum's callers
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t = fmaxf(t, smem[w]);
    *dst = t;
  }
}

// rms_norm (layernorm_kernels.cu:22-46): out = scalar_t(x * rsqrt(mean(x^2) + eps)) * weight
// fused_add_rms_norm (:258-291): z = input + residual (scalar_t); residual = z; input = scalar_t(z * s) * weight
template <typename T, bool FUSED_ADD>
__global__ void rms_norm_kernel(T* __restrict__ out, T* __restrict__ input, T* __restrict__ residual,
                                const T* __restrict__ weight, float eps, int hidden, float* __restrict__ absmax) {
  __shared__ float smem[17];
  const int64_t row = (int64_t)blockIdx.x * hidden;
  float var = 0.f;
  for (int i = threadIdx.x; i < hidden; i += blockDim.x) {
    T x = input[row + i];
    if constexpr (FUSED_ADD) {
      x = Scalar<T>::from_f32(Scalar<T>::to_f32(x) + Scalar<T>::to_f32(residual[row + i]));
      residual[row + i] = x;
    }
    const float f = Scalar<T>::to_f32(x);
    var += f * f;
  }
  var = block_sum(var, smem);
  const float s = rsqrtf(var / (float)hidden + eps);
  float amax = 0.f;
  for (int i = threadIdx.x; i < hidden; i += blockDim.x) {
    const T x = FUSED_ADD ? residual[row + i] : input[row + i];
    const T n = Scalar<T>::from_f32(Scalar<T>::to_f32(x) * s);
    const T o = rnd_mul<T>(n, weight[i]);
    out[row + i] = o;
    amax = fmaxf(amax, fabsf(Scalar<T>::to_f32(o)));
  }
  if (absmax != nullptr) block_max_store(amax, smem, absmax + blockIdx.x);
}

// 16-B vectorised fp16 / bf16 variant (hidden % 8 == 0, 16-B aligned rows), one pass over registers
template <typename T, bool FUSED_ADD, int VPT>  // VPT = 16-B vectors per thread
__global__ void rms_norm_vec_kernel(T* __restrict__ out, T* __restrict__ input, T* __restrict__ residual,
                                    const T* __restrict__ weight, float eps, int hidden, float* __restrict__ absmax) {
  __shared__ float smem[17];
  const int64_t row = (int64_t)blockIdx.x * hidden;
  const int nvec = hidden / 8;
  union V { u32x4 u; T e[8]; };
  V x[VPT], w[VPT];
  float var = 0.f;
  // the weight vectors are requested together with the row: behind the block reduction (a barrier the compiler does not
  // move loads across) their L2 round trip would be exposed, and this kernel is all latency (4 KB per workgroup)
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int v = threadIdx.x + k * blockDim.x;
    if (v < nvec) w[k].u = *reinterpret_cast<const u32x4*>(weight + v * 8);
  }
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int v = threadIdx.x + k * blockDim.x;
    if (v < nvec) {
      x[k].u = *reinterpret_cast<const u32x4*>(input + row + v * 8);
      if constexpr (FUSED_ADD) {
        V r;
        r.u = *reinterpret_cast<const u32x4*>(residual + row + v * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[k].e[j] = Scalar<T>::from_f32(Scalar<T>::to_f32(x[k].e[j]) + Scalar<T>::to_f32(r.e[j]));
        *reinterpret_cast<u32x4*>(residual + row + v * 8) = x[k].u;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = Scalar<T>::to_f32(x[k].e[j]);
        var += f * f;
      }
    }
  }
  var = block_sum(var, smem);
  const float s = rsqrtf(var / (float)hidden + eps);
  float amax = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int v = threadIdx.x + k * blockDim.x;
    if (v < nvec) {
      V o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o.e[j] = rnd_mul<T>(Scalar<T>::from_f32(Scalar<T>::to_f32(x[k].e[j]) * s), w[k].e[j]);
        amax = fmaxf(amax, fabsf(Scalar<T>::to_f32(o.e[j])));
      }
      *reinterpret_cast<u32x4*>(out + row + v * 8) = o.u;
    }
  }
  if (absmax != nullptr) block_max_store(amax, smem, absmax + blockIdx.x);
}

// rotary embedding (pos_encoding_kernels.cu:10-96): in place on query / key, NeoX or GPT-J pairing.
// arr[x] = x * cos - y * sin ; arr[y] = y * cos + x * sin, every operation rounded to scalar_t.
template <typename T, bool NEOX>
__global__ void rotary_kernel(const int64_t* __restrict__ positions, T* __restrict__ query, T* __restrict__ key,
                              const T* __restrict__ cos_sin_cache, const int64_t* __restrict__ offsets, int rot_dim,
                              int64_t q_stride, int64_t k_stride, int num_heads, int num_kv_heads, int head_size) {
  const int64_t tok = blockIdx.x;
  int64_t pos = positions[tok];
  if (offsets != nullptr) pos += offsets[tok];
  const T* cache = cos_sin_cache + pos * rot_dim;
  const int embed = rot_dim / 2;
  const int nq = num_heads * embed, nk = num_kv_heads * embed;
  for (int i = threadIdx.x; i < nq + nk; i += blockDim.x) {
    const bool is_q = i < nq;
    const int ii = is_q ? i : i - nq;
    const int head = ii / embed, ro = ii % embed;
    T* arr = (is_q ? query + tok * q_stride : key + tok * k_stride) + (int64_t)head * head_size;
    const int xi = NEOX ? ro : 2 * ro;
    const int yi = NEOX ? embed + ro : 2 * ro + 1;
    const T c = cache[ro], s = cache[embed + ro];
    const T x = arr[xi], y = arr[yi];
    arr[xi] = Scalar<T>::from_f32(Scalar<T>::to_f32(rnd_mul<T>(x, c)) - Scalar<T>::to_f32(rnd_mul<T>(y, s)));
    arr[yi] = Scalar<T>::from_f32(Scalar<T>::to_f32(rnd_mul<T>(y, c)) + Scalar<T>::to_f32(rnd_mul<T>(x, s)));
  }
}

// NeoX pairing, 2-byte types, everything 16-byte aligned: one thread = 8 consecutive rotary indices of one head (x run,
// y run, cos run, sin run = four 16-byte accesses). The x / y loads do not depend on the position, so they are issued
// before the positions -> cos/sin chain instead of behind it. Same arithmetic, element by element, as rotary_kernel.
template <typename T>
__global__ void rotary_neox_vec_kernel(const int64_t* __restrict__ positions, T* __restrict__ query, T* __restrict__ key,
                                       const T* __restrict__ cos_sin_cache, const int64_t* __restrict__ offsets, int rot_dim,
                                       int64_t q_stride, int64_t k_stride, int num_heads, int num_kv_heads, int head_size) {
  const int64_t tok = blockIdx.x;
  const int embed = rot_dim / 2, vper = embed / 8;  // 16-byte vectors per head half
  const int nvec = (num_heads + num_kv_heads) * vper;
  union V { u32x4 u; T e[8]; };
  for (int i = threadIdx.x; i < nvec; i += blockDim.x) {
    const int head = i / vper, ro = (i % vper) * 8;
    T* arr = (head < num_heads ? query + tok * q_stride + (int64_t)head * head_size
                               : key + tok * k_stride + (int64_t)(head - num_heads) * head_size);
    V x, y, c, sn;
    x.u = *reinterpret_cast<const u32x4*>(arr + ro);
    y.u = *reinterpret_cast<const u32x4*>(arr + embed + ro);
    int64_t pos = positions[tok];
    if (offsets != nullptr) pos += offsets[tok];
    const T* cache = cos_sin_cache + pos * rot_dim;
    c.u = *reinterpret_cast<const u32x4*>(cache + ro);
    sn.u = *reinterpret_cast<const u32x4*>(cache + embed + ro);
    V ox, oy;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ox.e[j] = Scalar<T>::from_f32(Scalar<T>::to_f32(rnd_mul<T>(x.e[j], c.e[j])) - Scalar<T>::to_f32(rnd_mul<T>(y.e[j], sn.e[j])));
      oy.e[j] = Scalar<T>::from_f32(Scalar<T>::to_f32(rnd_mul<T>(y.e[j], c.e[j])) + Scalar<T>::to_f32(rnd_mul<T>(x.e[j], sn.e[j])));
    }
    *reinterpret_cast<u32x4*>(arr + ro) = ox.u;
    *reinterpret_cast<u32x4*>(arr + embed + ro) = oy.u;
  }
}

enum { ACT_SILU = 0, ACT_GELU = 1, ACT_GELU_TANH = 2, ACT_GELU_NEW = 3, ACT_GELU_FAST = 4, ACT_GELU_QUICK = 5 };

template <typename T, int ACT>
__device__ __forceinline__ T act_fn(T xv) {
  const float f = Scalar<T>::to_f32(xv);
  if constexpr (ACT == ACT_SILU) {
    return silu_rnd<T>(xv);  // activation_kernels.cu:27-30
  } else if constexpr (ACT == ACT_GELU) {
    return Scalar<T>::from_f32(f * 0.5f * (1.0f + erff(f * 0.70710678118654752440f)));  // :33-40
  } else if constexpr (ACT == ACT_GELU_TANH) {
    const float beta = 1.41421356237309504880f * 1.12837916709551257390f * 0.5f;  // :43-53
    const float inner = beta * (f + 0.044715f * (f * f * f));
    return Scalar<T>::from_f32(0.5f * f * (1.0f + tanhf(inner)));
  } else if constexpr (ACT == ACT_GELU_NEW) {  // :113-118, scalar_t arithmetic step by step
    const T x3t = rnd_mul<T>(rnd_mul<T>(xv, xv), xv);
    const float x3 = Scalar<T>::to_f32(x3t);
    const T inner = Scalar<T>::from_f32(f + Scalar<T>::to_f32(Scalar<T>::from_f32(0.044715f * x3)));
    const T t = Scalar<T>::from_f32(tanhf(Scalar<T>::to_f32(Scalar<T>::from_f32(0.79788456f * Scalar<T>::to_f32(inner)))));
    const T half_x = rnd_mul<T>(Scalar<T>::from_f32(0.5f), xv);
    const T one_t = Scalar<T>::from_f32(1.0f + Scalar<T>::to_f32(t));
    return rnd_mul<T>(half_x, one_t);
  } else if constexpr (ACT == ACT_GELU_FAST) {  // :121-127
    const T a = Scalar<T>::from_f32(f * 0.79788456f);
    const T b = Scalar<T>::from_f32(1.0f + Scalar<T>::to_f32(rnd_mul<T>(Scalar<T>::from_f32(0.044715f * f), xv)));
    const T t = Scalar<T>::from_f32(tanhf(Scalar<T>::to_f32(rnd_mul<T>(a, b))));
    const T half_x = rnd_mul<T>(Scalar<T>::from_f32(0.5f), xv);
    const T one_t = Scalar<T>::from_f32(1.0f + Scalar<T>::to_f32(t));
    return rnd_mul<T>(half_x, one_t);
  } else {
    return Scalar<T>::from_f32(f / (1.0f + expf(-1.702f * f)));  // :130-133
  }
}

// act_and_mul (activation_kernels.cu:12-24): out[t, i] = ACT(in[t, i]) * in[t, d + i]
template <typename T, int ACT>
__global__ void act_and_mul_kernel(T* __restrict__ out, const T* __restrict__ in, int d, float* __restrict__ absmax) {
  __shared__ float smem[17];
  const int64_t tok = blockIdx.x;
  const T* x = in + tok * 2 * d;
  float amax = 0.f;
  bool done = false;
  if constexpr (sizeof(T) == 2) {
    if ((d & 7) == 0) {
      union V { u32x4 u; T e[8]; };
      for (int v = threadIdx.x; v < d / 8; v += blockDim.x) {
        V a, b, o;
        a.u = *reinterpret_cast<const u32x4*>(x + v * 8);
        b.u = *reinterpret_cast<const u32x4*>(x + d + v * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o.e[j] = rnd_mul<T>(act_fn<T, ACT>(a.e[j]), b.e[j]);
          amax = fmaxf(amax, fabsf(Scalar<T>::to_f32(o.e[j])));
        }
        *reinterpret_cast<u32x4*>(out + tok * d + v * 8) = o.u;
      }
      done = true;
    }
  }
  if (!done) {
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
      const T o = rnd_mul<T>(act_fn<T, ACT>(x[i]), x[d + i]);
      out[tok * d + i] = o;
      amax = fmaxf(amax, fabsf(Scalar<T>::to_f32(o)));
    }
  }
  if (absmax != nullptr) block_max_store(amax, smem, absmax + blockIdx.x);
}

template <typename T, int ACT>
__global__ void activation_kernel(T* __restrict__ out, const T* __restrict__ in, int d) {
  const int64_t tok = blockIdx.x;
  for (int i = threadIdx.x; i < d; i += blockDim.x) out[tok * d + i] = act_fn<T, ACT>(in[tok * d + i]);
}

template <typename T>
int launch_rms(void* out, void* input, void* residual, const void* weight, float eps, int num_tokens, int hidden,
               bool fused, hipStream_t stream, float* absmax = nullptr) {
  const bool vec = sizeof(T) == 2 && hidden % 8 == 0 &&
                   (((uintptr_t)out | (uintptr_t)input | (uintptr_t)weight | (uintptr_t)residual) % 16 == 0) &&
                   hidden / 8 <= 1024 * 2;
  if (vec) {
    const int nvec = hidden / 8;
    int threads = std::min(1024, ((nvec + 63) / 64) * 64);
    if (nvec > 256 && nvec <= 2048) threads = std::min(1024, ((nvec / 2 + 63) / 64) * 64);  // 2 vectors per thread
    const int vpt = (nvec + threads - 1) / threads;
#define NMX_RMS(F, V) rms_norm_vec_kernel<T, F, V><<<num_tokens, threads, 0, stream>>>((T*)out, (T*)input, (T*)residual, (const T*)weight, eps, hidden, absmax)
    if (fused) { if (vpt == 1) NMX_RMS(true, 1); else NMX_RMS(true, 2); }
    else { if (vpt == 1) NMX_RMS(false, 1); else NMX_RMS(false, 2); }
#undef NMX_RMS
  } else {
    const int threads = std::min(1024, ((hidden + 63) / 64) * 64);
    if (fused) rms_norm_kernel<T, true><<<num_tokens, threads, 0, stream>>>((T*)out, (T*)input, (T*)residual, (const T*)weight, eps, hidden, absmax);
    else rms_norm_kernel<T, false><<<num_tokens, threads, 0, stream>>>((T*)out, (T*)input, (T*)residual, (const T*)weight, eps, hidden, absmax);
  }
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

template <typename T>
int launch_act(void* out, const void* in, int num_tokens, int d, int act, bool gated, hipStream_t stream, float* absmax = nullptr) {
  const int work = (gated && sizeof(T) == 2 && d % 8 == 0) ? d / 8 : d;
  const int threads = std::min(1024, std::max(64, ((work + 63) / 64) * 64));
#define NMX_ACT(A)                                                                                      \
  if (gated) act_and_mul_kernel<T, A><<<num_tokens, threads, 0, stream>>>((T*)out, (const T*)in, d, absmax);    \
  else activation_kernel<T, A><<<num_tokens, threads, 0, stream>>>((T*)out, (const T*)in, d)
  switch (act) {
    case ACT_SILU: NMX_ACT(ACT_SILU); break;
    case ACT_GELU: NMX_ACT(ACT_GELU); break;
    case ACT_GELU_TANH: NMX_ACT(ACT_GELU_TANH); break;
    case ACT_GELU_NEW: NMX_ACT(ACT_GELU_NEW); break;
    case ACT_GELU_FAST: NMX_ACT(ACT_GELU_FAST); break;
    case ACT_GELU_QUICK: NMX_ACT(ACT_GELU_QUICK); break;
    default: NMX_CHECK(false, NMX_ERR_INVALID_ARG, "unknown activation %d", act);
  }
#undef NMX_ACT
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}


// (block_sum / sum_partials8: nmx_common.h - marlin_decode_kernel<NORM> computes the same norm in its prologue)
// fused_add_rms_norm on x = round(sum_s partial[s]): residual += x; out = norm(residual) * weight (layernorm_kernels.cu:258-291)
template <typename T, int VPT>
__global__ void rms_norm_splitk_kernel(T* __restrict__ out, const float* __restrict__ partial, int splits, const float* __restrict__ sa,
                                       const float* __restrict__ sb, float* __restrict__ absmax, T* __restrict__ residual,
                                       const T* __restrict__ weight, float eps, int hidden, int64_t slab) {
  __shared__ float smem[17];
  const int64_t row = (int64_t)blockIdx.x * hidden;
  const int nvec = hidden / 8;
  union V { u32x4 u; T e[8]; };
  V x[VPT], w[VPT];
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int v = threadIdx.x + k * blockDim.x;
    if (v < nvec) w[k].u = *reinterpret_cast<const u32x4*>(weight + v * 8);
  }
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int v = threadIdx.x + k * blockDim.x;
    if (v < nvec) {
      V r;  // requested BEFORE the slab sum consumes its loads: one memory round trip instead of two
      r.u = *reinterpret_cast<const u32x4*>(residual + row + v * 8);
      sum_partials8<T>(partial, splits, slab, row + v * 8, x[k].e, sa, sb);
#pragma unroll
      for (int j = 0; j < 8; ++j) x[k].e[j] = Scalar<T>::from_f32(Scalar<T>::to_f32(x[k].e[j]) + Scalar<T>::to_f32(r.e[j]));
      *reinterpret_cast<u32x4*>(residual + row + v * 8) = x[k].u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = Scalar<T>::to_f32(x[k].e[j]);
        var += f * f;
      }
    }
  }
  var = block_sum(var, smem);
  const float sc = rsqrtf(var / (float)hidden + eps);
  float amax = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int v = threadIdx.x + k * blockDim.x;
    if (v < nvec) {
      V o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o.e[j] = rnd_mul<T>(Scalar<T>::from_f32(Scalar<T>::to_f32(x[k].e[j]) * sc), w[k].e[j]);
        amax = fmaxf(amax, fabsf(Scalar<T>::to_f32(o.e[j])));
      }
      *reinterpret_cast<u32x4*>(out + row + v * 8) = o.u;
    }
  }
  if (absmax != nullptr) block_max_store(amax, smem, absmax + blockIdx.x);
}

// silu_and_mul on x = round(sum_s partial[s]) (activation_kernels.cu:12-30)
template <typename T>
__global__ void silu_and_mul_splitk_kernel(T* __restrict__ out, const float* __restrict__ partial, int splits, int d, int64_t slab,
                                           const float* __restrict__ sa, const float* __restrict__ sb, float* __restrict__ absmax) {
  __shared__ float smem[17];
  const int64_t tok = blockIdx.x;
  union V { u32x4 u; T e[8]; };
  float amax = 0.f;
  for (int v = threadIdx.x; v < d / 8; v += blockDim.x) {
    V a, b, o;
    sum_partials8<T>(partial, splits, slab, tok * 2 * d + v * 8, a.e, sa, sb);
    sum_partials8<T>(partial, splits, slab, tok * 2 * d + d + v * 8, b.e, sa, sb);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o.e[j] = rnd_mul<T>(act_fn<T, ACT_SILU>(a.e[j]), b.e[j]);
      amax = fmaxf(amax, fabsf(Scalar<T>::to_f32(o.e[j])));
    }
    *reinterpret_cast<u32x4*>(out + tok * d + v * 8) = o.u;
  }
  if (absmax != nullptr) block_max_store(amax, smem, absmax + blockIdx.x);
}

// rotary_embedding (NeoX, rot_dim == head_size) on the q and k heads of a fused qkv row + reshape_and_cache of the
// (rotated) k and v heads, in one pass over the row (pos_encoding_kernels.cu:10-96, cache_kernels.cu:153-278); the row
// comes from fp16 / bf16 qkv (splits == 1, rotated in place) or from split-K partial sums (written to qkv once).
// One thread = 8 consecutive rotary indices of one head: the x run and the y run (8 elements each).
template <typename T, int KV>
__global__ void rope_cache_kernel(const int64_t* __restrict__ positions, T* __restrict__ qkv, const float* __restrict__ partial,
                                  int splits, int64_t slab, const float* __restrict__ sa, const float* __restrict__ sb, const T* __restrict__ cos_sin_cache, void* __restrict__ key_cache,
                                  void* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping, int num_heads,
                                  int num_kv_heads, int head_size, int block_size, float kv_scale) {
  using cache_t = typename std::conditional<KV == NMX_KV_AUTO, T, uint8_t>::type;
  constexpr int X = 16 / sizeof(cache_t);  // elements per 16-byte K chunk
  const int64_t tok = blockIdx.x;
  const int embed = head_size / 2, vper = embed / 8;
  const int heads = num_heads + 2 * num_kv_heads;
  const int64_t row = tok * (int64_t)heads * head_size;
  const int64_t pos = positions[tok];
  const int64_t slot = slot_mapping[tok];
  const int64_t block_idx = slot / block_size, block_off = slot % block_size;
  const T* cache = cos_sin_cache + pos * head_size;
  union V { u32x4 u; T e[8]; };
  // a token's heads are spread over gridDim.y workgroups of 128 threads: at decode batch sizes one workgroup per token
  // leaves most CUs idle and every thread is a chain of dependent loads (position -> cos / sin, slot -> cache address)
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < heads * vper; i += blockDim.x * gridDim.y) {
    const int head = i / vper, ro = (i % vper) * 8;
    const int64_t base = row + (int64_t)head * head_size;
    V x, y, c, sn;
    const bool is_v = head >= num_heads + num_kv_heads;
    if (!is_v) {  // cos / sin requested BEFORE the slab sums consume their loads (a dependent round trip less)
      c.u = *reinterpret_cast<const u32x4*>(cache + ro);
      sn.u = *reinterpret_cast<const u32x4*>(cache + embed + ro);
    }
    if (splits > 1) {
      sum_partials8<T>(partial, splits, slab, base + ro, x.e, sa, sb);
      sum_partials8<T>(partial, splits, slab, base + embed + ro, y.e, sa, sb);
    } else {
      x.u = *reinterpret_cast<const u32x4*>(qkv + base + ro);
      y.u = *reinterpret_cast<const u32x4*>(qkv + base + embed + ro);
    }
    if (!is_v) {
      V ox, oy;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ox.e[j] = Scalar<T>::from_f32(Scalar<T>::to_f32(rnd_mul<T>(x.e[j], c.e[j])) - Scalar<T>::to_f32(rnd_mul<T>(y.e[j], sn.e[j])));
        oy.e[j] = Scalar<T>::from_f32(Scalar<T>::to_f32(rnd_mul<T>(y.e[j], c.e[j])) + Scalar<T>::to_f32(rnd_mul<T>(x.e[j], sn.e[j])));
      }
      x = ox;
      y = oy;
    }
    if (!is_v || splits > 1) {
      *reinterpret_cast<u32x4*>(qkv + base + ro) = x.u;
      *reinterpret_cast<u32x4*>(qkv + base + embed + ro) = y.u;
    }
    if (head < num_heads || slot < 0) continue;  // q heads / padding tokens: nothing to cache
    const int kvh = is_v ? head - num_heads - num_kv_heads : head - num_heads;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const V& val = half == 0 ? x : y;
      const int d0 = half == 0 ? ro : embed + ro;
      if (!is_v) {
        cache_t* kc = reinterpret_cast<cache_t*>(key_cache) +
                      (((block_idx * num_kv_heads + kvh) * (head_size / X) + d0 / X) * block_size + block_off) * X + d0 % X;
        if constexpr (KV == NMX_KV_AUTO) {
          *reinterpret_cast<u32x4*>(kc) = val.u;
        } else {
          union { u32x2 u; uint8_t b[8]; } q;
#pragma unroll
          for (int j = 0; j < 8; ++j) q.b[j] = f32_to_fp8_sat<KV>(Scalar<T>::to_f32(val.e[j]) / kv_scale);
          *reinterpret_cast<u32x2*>(kc) = q.u;
        }
      } else {
        cache_t* vc = reinterpret_cast<cache_t*>(value_cache) + ((block_idx * num_kv_heads + kvh) * head_size + d0) * block_size + block_off;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if constexpr (KV == NMX_KV_AUTO) vc[(int64_t)j * block_size] = val.e[j];
          else vc[(int64_t)j * block_size] = f32_to_fp8_sat<KV>(Scalar<T>::to_f32(val.e[j]) / kv_scale);
        }
      }
    }
  }
}

}  // namespace

#define NMX_DISPATCH_DT(dtype, CALL)                                                     \
  switch (dtype) {                                                                       \
    case NMX_F32: { using T = float; return CALL; }                                      \
    case NMX_F16: { using T = f16; return CALL; }                                        \
    case NMX_BF16: { using T = bf16; return CALL; }                                      \
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "unsupported dtype code %d", dtype);  \
  }

extern "C" int nmx_rms_norm(void* out, const void* input, const void* weight, float epsilon, int num_tokens,
                            int hidden_size, int dtype, nmx_stream_t stream) {
  if (num_tokens == 0) return NMX_OK;
  NMX_CHECK(hidden_size > 0, NMX_ERR_INVALID_ARG, "hidden_size must be > 0");
  NMX_DISPATCH_DT(dtype, launch_rms<T>(out, const_cast<void*>(input), nullptr, weight, epsilon, num_tokens, hidden_size, false, (hipStream_t)stream));
}

extern "C" int nmx_fused_add_rms_norm(void* input, void* residual, const void* weight, float epsilon, int num_tokens,
                                      int hidden_size, int dtype, nmx_stream_t stream) {
  if (num_tokens == 0) return NMX_OK;
  NMX_CHECK(hidden_size > 0, NMX_ERR_INVALID_ARG, "hidden_size must be > 0");
  NMX_DISPATCH_DT(dtype, launch_rms<T>(input, input, residual, weight, epsilon, num_tokens, hidden_size, true, (hipStream_t)stream));
}

template <typename T>
static int launch_rotary(const int64_t* positions, void* query, void* key, const void* cache, const int64_t* offsets,
                         int rot_dim, int64_t q_stride, int64_t k_stride, int num_tokens, int num_heads,
                         int num_kv_heads, int head_size, int is_neox, hipStream_t stream) {
  const int work = (num_heads + num_kv_heads) * rot_dim / 2;
  const bool vec = is_neox && sizeof(T) == 2 && rot_dim % 16 == 0 && head_size % 8 == 0 && q_stride % 8 == 0 && k_stride % 8 == 0 &&
                   (((uintptr_t)query | (uintptr_t)key | (uintptr_t)cache) % 16 == 0);
  if (vec) {
    const int nvec = work / 8;
    const int vthreads = std::min(512, std::max(64, ((nvec + 63) / 64) * 64));
    rotary_neox_vec_kernel<T><<<num_tokens, vthreads, 0, stream>>>(positions, (T*)query, (T*)key, (const T*)cache, offsets, rot_dim,
                                                                 q_stride, k_stride, num_heads, num_kv_heads, head_size);
    NMX_LAUNCH_CHECK();
    return NMX_OK;
  }
  const int threads = std::min(512, std::max(64, ((work + 63) / 64) * 64));
  if (is_neox)
    rotary_kernel<T, true><<<num_tokens, threads, 0, stream>>>(positions, (T*)query, (T*)key, (const T*)cache, offsets, rot_dim,
                                                               q_stride, k_stride, num_heads, num_kv_heads, head_size);
  else
    rotary_kernel<T, false><<<num_tokens, threads, 0, stream>>>(positions, (T*)query, (T*)key, (const T*)cache, offsets, rot_dim,
                                                                q_stride, k_stride, num_heads, num_kv_heads, head_size);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int nmx_rotary_embedding(const int64_t* positions, void* query, void* key, const void* cos_sin_cache,
                                    const int64_t* cos_sin_cache_offsets, int rot_dim, int64_t query_stride,
                                    int64_t key_stride, int num_tokens, int num_heads, int num_kv_heads,
                                    int head_size, int is_neox, int dtype, nmx_stream_t stream) {
  if (num_tokens == 0) return NMX_OK;
  NMX_CHECK(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size, NMX_ERR_INVALID_ARG,
            "rot_dim = %d must be even and <= head_size = %d", rot_dim, head_size);
  NMX_DISPATCH_DT(dtype, launch_rotary<T>(positions, query, key, cos_sin_cache, cos_sin_cache_offsets, rot_dim, query_stride,
                                          key_stride, num_tokens, num_heads, num_kv_heads, head_size, is_neox, (hipStream_t)stream));
}

// Producer-side absmax (dynamic fp8 activation quantisation, fp8.py:340-359): the same kernels, plus absmax[t] = max |out[t, :]|
// of the rounded outputs; nmx_scaled_fp8_quant_partials turns the num_tokens maxima into the tensor scale.
extern "C" int nmx_rms_norm_absmax(void* out, const void* input, const void* weight, float epsilon, int num_tokens,
                                   int hidden_size, int dtype, float* absmax, nmx_stream_t stream) {
  if (num_tokens == 0) return NMX_OK;
  NMX_CHECK(hidden_size > 0 && absmax != nullptr, NMX_ERR_INVALID_ARG, "hidden_size must be > 0 and absmax non-null");
  NMX_DISPATCH_DT(dtype, launch_rms<T>(out, const_cast<void*>(input), nullptr, weight, epsilon, num_tokens, hidden_size, false, (hipStream_t)stream, absmax));
}

extern "C" int nmx_fused_add_rms_norm_absmax(void* input, void* residual, const void* weight, float epsilon, int num_tokens,
                                             int hidden_size, int dtype, float* absmax, nmx_stream_t stream) {
  if (num_tokens == 0) return NMX_OK;
  NMX_CHECK(hidden_size > 0 && absmax != nullptr, NMX_ERR_INVALID_ARG, "hidden_size must be > 0 and absmax non-null");
  NMX_DISPATCH_DT(dtype, launch_rms<T>(input, input, residual, weight, epsilon, num_tokens, hidden_size, true, (hipStream_t)stream, absmax));
}

extern "C" int nmx_act_and_mul_absmax(void* out, const void* input, int num_tokens, int d, int act, int dtype, float* absmax,
                                      nmx_stream_t stream) {
  if (num_tokens == 0 || d == 0) return NMX_OK;
  NMX_CHECK(absmax != nullptr, NMX_ERR_INVALID_ARG, "absmax must be non-null");
  NMX_DISPATCH_DT(dtype, launch_act<T>(out, input, num_tokens, d, act, true, (hipStream_t)stream, absmax));
}

extern "C" int nmx_act_and_mul(void* out, const void* input, int num_tokens, int d, int act, int dtype,
                               nmx_stream_t stream) {
  if (num_tokens == 0 || d == 0) return NMX_OK;
  NMX_DISPATCH_DT(dtype, launch_act<T>(out, input, num_tokens, d, act, true, (hipStream_t)stream));
}

extern "C" int nmx_activation(void* out, const void* input, int num_tokens, int d, int act, int dtype,
                              nmx_stream_t stream) {
  if (num_tokens == 0 || d == 0) return NMX_OK;
  NMX_DISPATCH_DT(dtype, launch_act<T>(out, input, num_tokens, d, act, false, (hipStream_t)stream));
}

// ---- fused consumers of deferred split-K partials (no reference counterpart: the reference runs the ops one by one;
//      results are bit-identical to nmx_fused_add_rms_norm / nmx_act_and_mul / nmx_rotary_embedding + nmx_reshape_and_cache
//      applied to the reduced GEMM output) -----------------------------------------------------------------------------
static int add_rms_norm_splitk_common(void* input_out, const float* partial, int splits, const float* sa, const float* sb,
                                      void* residual, const void* weight, float epsilon, int num_tokens, int hidden_size,
                                      int dtype, float* absmax, hipStream_t stream) {
  if (num_tokens == 0) return NMX_OK;
  NMX_CHECK((sa == nullptr) == (sb == nullptr), NMX_ERR_INVALID_ARG, "splitk consumer: both scales or neither");
  NMX_CHECK(dtype == NMX_F16 || dtype == NMX_BF16, NMX_ERR_UNSUPPORTED, "fused_add_rms_norm_splitk: fp16 / bf16 only");
  NMX_CHECK(NMX_SPLITK_COUNT(splits) >= 2 && partial != nullptr, NMX_ERR_INVALID_ARG, "fused_add_rms_norm_splitk needs >= 2 partial slabs");
  NMX_CHECK(hidden_size % 8 == 0 && hidden_size / 8 <= 2048 &&
                (((uintptr_t)input_out | (uintptr_t)partial | (uintptr_t)residual | (uintptr_t)weight) % 16 == 0),
            NMX_ERR_INVALID_ARG, "fused_add_rms_norm_splitk: hidden %% 8 == 0 (<= 16384) and 16-byte aligned operands");
  const int nvec = hidden_size / 8;
  int threads = std::min(1024, ((nvec + 63) / 64) * 64);
  if (nvec > 256) threads = std::min(1024, ((nvec / 2 + 63) / 64) * 64);
  const int vpt = (nvec + threads - 1) / threads;
  const int64_t slab = (int64_t)num_tokens * hidden_size;
#define NMX_RS(T, V) rms_norm_splitk_kernel<T, V><<<num_tokens, threads, 0, stream>>>((T*)input_out, partial, splits, sa, sb, absmax, (T*)residual, (const T*)weight, epsilon, hidden_size, slab)
  if (dtype == NMX_F16) { if (vpt == 1) NMX_RS(f16, 1); else NMX_RS(f16, 2); }
  else { if (vpt == 1) NMX_RS(bf16, 1); else NMX_RS(bf16, 2); }
#undef NMX_RS
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int nmx_fused_add_rms_norm_splitk(void* input_out, const float* partial, int splits, void* residual,
                                             const void* weight, float epsilon, int num_tokens, int hidden_size, int dtype,
                                             nmx_stream_t stream) {
  return add_rms_norm_splitk_common(input_out, partial, splits, nullptr, nullptr, residual, weight, epsilon, num_tokens, hidden_size,
                                    dtype, nullptr, (hipStream_t)stream);
}

// ... of a deferred fp8 scaled_mm: x = round(sa[0] * (sb[0] * sum_s partial[s])) (its epilogue), and optionally the per-token
// |max| of the normed output for nmx_scaled_fp8_quant_partials (absmax may be null)
extern "C" int nmx_fused_add_rms_norm_splitk_scaled(void* input_out, const float* partial, int splits, const float* sa,
                                                    const float* sb, void* residual, const void* weight, float epsilon,
                                                    int num_tokens, int hidden_size, int dtype, float* absmax,
                                                    nmx_stream_t stream) {
  return add_rms_norm_splitk_common(input_out, partial, splits, sa, sb, residual, weight, epsilon, num_tokens, hidden_size, dtype,
                                    absmax, (hipStream_t)stream);
}

static int silu_and_mul_splitk_common(void* out, const float* partial, int splits, const float* sa, const float* sb, int num_tokens,
                                      int d, int dtype, float* absmax, hipStream_t stream) {
  if (num_tokens == 0 || d == 0) return NMX_OK;
  NMX_CHECK((sa == nullptr) == (sb == nullptr), NMX_ERR_INVALID_ARG, "splitk consumer: both scales or neither");
  NMX_CHECK(dtype == NMX_F16 || dtype == NMX_BF16, NMX_ERR_UNSUPPORTED, "silu_and_mul_splitk: fp16 / bf16 only");
  NMX_CHECK(NMX_SPLITK_COUNT(splits) >= 2 && partial != nullptr && d % 8 == 0 && (((uintptr_t)out | (uintptr_t)partial) % 16 == 0),
            NMX_ERR_INVALID_ARG, "silu_and_mul_splitk: >= 2 slabs, d %% 8 == 0, 16-byte aligned operands");
  const int threads = std::min(1024, std::max(64, ((d / 8 + 63) / 64) * 64));
  const int64_t slab = (int64_t)num_tokens * 2 * d;
  if (dtype == NMX_F16) silu_and_mul_splitk_kernel<f16><<<num_tokens, threads, 0, stream>>>((f16*)out, partial, splits, d, slab, sa, sb, absmax);
  else silu_and_mul_splitk_kernel<bf16><<<num_tokens, threads, 0, stream>>>((bf16*)out, partial, splits, d, slab, sa, sb, absmax);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int nmx_silu_and_mul_splitk(void* out, const float* partial, int splits, int num_tokens, int d, int dtype,
                                       nmx_stream_t stream) {
  return silu_and_mul_splitk_common(out, partial, splits, nullptr, nullptr, num_tokens, d, dtype, nullptr, (hipStream_t)stream);
}

extern "C" int nmx_silu_and_mul_splitk_scaled(void* out, const float* partial, int splits, const float* sa, const float* sb,
                                              int num_tokens, int d, int dtype, float* absmax, nmx_stream_t stream) {
  return silu_and_mul_splitk_common(out, partial, splits, sa, sb, num_tokens, d, dtype, absmax, (hipStream_t)stream);
}

static int rope_cache_common(const int64_t* positions, void* qkv, const float* partial, int splits, const float* sa, const float* sb,
                             const void* cos_sin_cache, void* key_cache, void* value_cache, const int64_t* slot_mapping,
                             int num_tokens, int num_heads, int num_kv_heads, int head_size, int block_size, int dtype, int kv_dtype,
                             float kv_scale, hipStream_t stream) {
  if (num_tokens == 0) return NMX_OK;
  NMX_CHECK((sa == nullptr) == (sb == nullptr), NMX_ERR_INVALID_ARG, "splitk consumer: both scales or neither");
  NMX_CHECK(dtype == NMX_F16 || dtype == NMX_BF16, NMX_ERR_UNSUPPORTED, "rope_reshape_and_cache: fp16 / bf16 only");
  NMX_CHECK(head_size % 16 == 0 && num_heads > 0 && num_kv_heads > 0, NMX_ERR_INVALID_ARG,
            "rope_reshape_and_cache: head_size %% 16 == 0 (NeoX rotary over the whole head)");
  NMX_CHECK((splits == 1) || (NMX_SPLITK_COUNT(splits) >= 2 && partial != nullptr), NMX_ERR_INVALID_ARG, "bad split count %d", splits);
  NMX_CHECK((((uintptr_t)qkv | (uintptr_t)partial | (uintptr_t)cos_sin_cache | (uintptr_t)key_cache) % 16 == 0), NMX_ERR_INVALID_ARG,
            "rope_reshape_and_cache: operands must be 16-byte aligned");
  const int heads = num_heads + 2 * num_kv_heads;
  const int nvec = heads * (head_size / 16);
  const int threads = std::min(128, std::max(64, ((nvec + 63) / 64) * 64));
  const dim3 grid(num_tokens, std::min(8, ceil_div(nvec, threads)));
  const int64_t slab = (int64_t)num_tokens * heads * head_size;
#define NMX_RC(T, KVC) rope_cache_kernel<T, KVC><<<grid, threads, 0, stream>>>(positions, (T*)qkv, partial, splits, slab, sa, sb, (const T*)cos_sin_cache, key_cache, value_cache, slot_mapping, num_heads, num_kv_heads, head_size, block_size, kv_scale)
#define NMX_RC_T(T)                                                                       \
  switch (kv_dtype) {                                                                     \
    case NMX_KV_AUTO: NMX_RC(T, NMX_KV_AUTO); break;                                      \
    case NMX_KV_FP8_E4M3: NMX_RC(T, NMX_KV_FP8_E4M3); break;                              \
    case NMX_KV_FP8_E5M2: NMX_RC(T, NMX_KV_FP8_E5M2); break;                              \
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "unsupported kv cache dtype %d", kv_dtype); \
  }
  if (dtype == NMX_F16) { NMX_RC_T(f16) } else { NMX_RC_T(bf16) }
#undef NMX_RC_T
#undef NMX_RC
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int nmx_rope_reshape_and_cache(const int64_t* positions, void* qkv, const float* partial, int splits,
                                          const void* cos_sin_cache, void* key_cache, void* value_cache,
                                          const int64_t* slot_mapping, int num_tokens, int num_heads, int num_kv_heads,
                                          int head_size, int block_size, int dtype, int kv_dtype, float kv_scale,
                                          nmx_stream_t stream) {
  return rope_cache_common(positions, qkv, partial, splits, nullptr, nullptr, cos_sin_cache, key_cache, value_cache, slot_mapping,
                           num_tokens, num_heads, num_kv_heads, head_size, block_size, dtype, kv_dtype, kv_scale, (hipStream_t)stream);
}

extern "C" int nmx_rope_reshape_and_cache_scaled(const int64_t* positions, void* qkv, const float* partial, int splits,
                                                 const float* sa, const float* sb, const void* cos_sin_cache, void* key_cache,
                                                 void* value_cache, const int64_t* slot_mapping, int num_tokens, int num_heads,
                                                 int num_kv_heads, int head_size, int block_size, int dtype, int kv_dtype,
                                                 float kv_scale, nmx_stream_t stream) {
  return rope_cache_common(positions, qkv, partial, splits, sa, sb, cos_sin_cache, key_cache, value_cache, slot_mapping, num_tokens,
                           num_heads, num_kv_heads, head_size, block_size, dtype, kv_dtype, kv_scale, (hipStream_t)stream);
}
