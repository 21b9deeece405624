// Common device/host helpers for libnmx_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <type_traits>

#include "../../include/nmx.h"

// ---- error plumbing -------------------------------------------------------------------------------------
void nmx_set_error(const char* fmt, ...);

#define NMX_CHECK(cond, code, ...)  \
  do {                              \
    if (!(cond)) {                  \
      nmx_set_error(__VA_ARGS__);   \
      return (code);                \
    }                               \
  } while (0)

#define NMX_HIP(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      nmx_set_error("HIP error %d (%s) at %s:%d", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); \
      return NMX_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

#define NMX_LAUNCH_CHECK() NMX_HIP(hipGetLastError())

// ---- tuning overrides (sweeps and tests only; never needed for correct results) ------------------------------------
// The environment is read ONCE, when the library is loaded; nmx_tuning_set() (include/nmx.h) changes a value afterwards.
// `splits` arguments of the split-K consumers / reduce entries: bits 0..7 = number of slabs, NMX_SPLITK_F16 = the slabs hold
// fp16 instead of fp32 partial sums (round 3: the M > 64 Marlin kernels with fp16 outputs write them - half the slab traffic; the
// reference's own global reduce passes fp16 partials between blocks too, gptq_marlin.cu global_reduce)
#define NMX_SPLITK_F16 0x100
#define NMX_SPLITK_COUNT(s) ((s) & 0xff)

enum NmxTune {
  NMX_TUNE_GEMM_CFG = 0, NMX_TUNE_GEMM_LEAN, NMX_TUNE_GEMM_LARGE, NMX_TUNE_GEMM_LARGE_NGRP, NMX_TUNE_GEMM_WIDE,
  NMX_TUNE_ATTN_NW, NMX_TUNE_PREFILL_GQ, NMX_TUNE_MM_NO_LDS, NMX_TUNE_MM_NT, NMX_TUNE_AWQ_NO_RING, NMX_TUNE_GPTQ_NO_RING,
  NMX_TUNE_GPTQ_NT, NMX_TUNE_ATTN_FP8W, NMX_TUNE_MM_TILE, NMX_TUNE_GEMM_XCD_SPLIT, NMX_TUNE_GEMM_DMA, NMX_TUNE_SLAB_F32, NMX_TUNE_ATTN_PART, NMX_TUNE_GEMM_NORM_ROWS, NMX_TUNE_GEMM_ATTN, NMX_TUNE_COUNT
};
__attribute__((visibility("hidden"))) const char* nmx_tune(int id);  // value, or nullptr when unset

// ---- vector types ---------------------------------------------------------------------------------------
typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#include <float.h>
#define NMX_WAVE 64

// ---- scalar conversions (device) ------------------------------------------------------------------------
template <typename T> struct Scalar;
template <> struct Scalar<float> {
  static __device__ __forceinline__ float to_f32(float v) { return v; }
  static __device__ __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Scalar<f16> {
  static __device__ __forceinline__ float to_f32(f16 v) { return (float)v; }
  static __device__ __forceinline__ f16 from_f32(float v) { return (f16)v; }
};
template <> struct Scalar<bf16> {
  static __device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
  static __device__ __forceinline__ bf16 from_f32(float v) { return (bf16)v; }  // v_cvt_pk_bf16_f32: RNE, NaN-safe
};

template <typename T> __device__ __forceinline__ T rnd_mul(T a, T b) {  // scalar_t * scalar_t -> scalar_t
  return Scalar<T>::from_f32(Scalar<T>::to_f32(a) * Scalar<T>::to_f32(b));
}
// silu in the reference's arithmetic (activation_kernels.cu:27-30): float math, result rounded to scalar_t
template <typename T> __device__ __forceinline__ T silu_rnd(T xv) {
  const float f = Scalar<T>::to_f32(xv);
  return Scalar<T>::from_f32(f / (1.0f + expf(-f)));
}

// fp8 (OCP) byte -> float
__device__ __forceinline__ float fp8_e4m3_to_f32(uint8_t v) {
  return __builtin_amdgcn_cvt_f32_fp8((uint32_t)v, 0);
}
__device__ __forceinline__ float fp8_e5m2_to_f32(uint8_t v) {
  return __builtin_amdgcn_cvt_f32_bf8((uint32_t)v, 0);
}
// float -> fp8 byte, round-to-nearest-even, saturating to max finite (reference: __NV_SATFINITE)
__device__ __forceinline__ uint8_t f32_to_fp8_e4m3_sat(float f) {
  // v_cvt_pk_fp8_f32 saturates when the FP16_OVFL-independent clamp is applied by us (NaN propagates)
  f = __builtin_isnan(f) ? f : __builtin_fminf(__builtin_fmaxf(f, -448.0f), 448.0f);
  return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(f, f, 0, false) & 0xff);
}
__device__ __forceinline__ uint8_t f32_to_fp8_e5m2_sat(float f) {
  f = __builtin_isnan(f) ? f : __builtin_fminf(__builtin_fmaxf(f, -57344.0f), 57344.0f);
  return (uint8_t)(__builtin_amdgcn_cvt_pk_bf8_f32(f, f, 0, false) & 0xff);
}
template <int KV> __device__ __forceinline__ float fp8_to_f32(uint8_t v) {
  if constexpr (KV == NMX_KV_FP8_E4M3) return fp8_e4m3_to_f32(v);
  else return fp8_e5m2_to_f32(v);
}
template <int KV> __device__ __forceinline__ uint8_t f32_to_fp8_sat(float f) {
  if constexpr (KV == NMX_KV_FP8_E4M3) return f32_to_fp8_e4m3_sat(f);
  else return f32_to_fp8_e5m2_sat(f);
}

// ---- wave-level reductions ------------------------------------------------------------------------------
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// sum over the workgroup (two barriers; smem: 17 floats); waves that pass 0 do not change the result
__device__ __forceinline__ float block_sum(float v, float* smem) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  float t = (threadIdx.x < nw) ? smem[threadIdx.x] : 0.f;
  if (wave == 0) {
    t = wave_reduce_sum(t);
    if (lane == 0) smem[16] = t;
  }
  __syncthreads();
  return smem[16];
}


// ---- consumers of DEFERRED split-K partial sums ---------------------------------------------------------------------
// A Marlin-family GEMM that splits K across workgroups leaves fp32 slabs partial[s][row][col]; instead of a reduce launch
// (read the slabs, write fp16, then the next element-wise op reads that again) the op that consumes the GEMM output sums
// them while loading its row: one dependent launch and one fp16 round trip less per GEMM. The sum runs in the order of
// splitk_reduce_kernel (s = 0, 1, ...) and is rounded to scalar_t before any further arithmetic, so every result is
// bit-identical to the unfused op sequence.
// sa / sb (both or neither): per-tensor scales of a deferred fp8 scaled_mm, applied as its epilogue does - sa * (sb * sum)
// (quant_ops.hip mm_epilogue4) - before the rounding to scalar_t.
template <typename T>
__device__ __forceinline__ void sum_partials8(const float* __restrict__ partial, int splits, int64_t slab, int64_t off, T (&e)[8],
                                              const float* __restrict__ sa = nullptr, const float* __restrict__ sb = nullptr) {
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  const int ns = NMX_SPLITK_COUNT(splits);
  // The slabs are fetched in batches of up to 8 loads issued back to back (round 3, late: with one load per loop iteration
  // every slab cost a full memory round trip - a 14-slab down_proj row took ~14 of them); the additions keep the order
  // s = 0, 1, ... of splitk_reduce_kernel, so nothing changes in the result. Slots past the count re-read the last slab.
  constexpr int BATCH = 8;
  if (splits & NMX_SPLITK_F16) {  // (uniform) fp16 slabs: 16 bytes = the 8 elements
    const f16* ph = reinterpret_cast<const f16*>(partial);
    for (int s0 = 0; s0 < ns; s0 += BATCH) {
      union { u32x4 u; f16 h[8]; } v[BATCH];
#pragma unroll
      for (int i = 0; i < BATCH; ++i) v[i].u = *reinterpret_cast<const u32x4*>(ph + min(s0 + i, ns - 1) * slab + off);
#pragma unroll
      for (int i = 0; i < BATCH; ++i) {
        if (s0 + i < ns) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            a0[j] += (float)v[i].h[j];
            a1[j] += (float)v[i].h[4 + j];
          }
        }
      }
    }
  } else {
    for (int s0 = 0; s0 < ns; s0 += BATCH) {
      f32x4 t0[BATCH], t1[BATCH];
#pragma unroll
      for (int i = 0; i < BATCH; ++i) {
        const float* src = partial + min(s0 + i, ns - 1) * slab + off;
        t0[i] = *reinterpret_cast<const f32x4*>(src);
        t1[i] = *reinterpret_cast<const f32x4*>(src + 4);
      }
#pragma unroll
      for (int i = 0; i < BATCH; ++i) {
        if (s0 + i < ns) {
          a0 += t0[i];
          a1 += t1[i];
        }
      }
    }
  }
  if (sa != nullptr) {
    const float va = sa[0], vb = sb[0];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x0 = va * (vb * a0[j]), x1 = va * (vb * a1[j]);
      asm volatile("" : "+v"(x0), "+v"(x1));  // fp32 rounding step of its own, as in mm_epilogue4 (no fusion with the conversion)
      a0[j] = x0;
      a1[j] = x1;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    e[j] = Scalar<T>::from_f32(a0[j]);
    e[4 + j] = Scalar<T>::from_f32(a1[j]);
  }
}


// COHERENT: the partials were written by other workgroups of the SAME launch with write-through stores; they are read with
// agent-scope (sc1) loads, which do not hit a stale line of this CU's L1 or this XCD's L2.
template <bool COHERENT> __device__ __forceinline__ float ld_f32(const float* p) {
  if constexpr (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COHERENT, typename scalar_t> __device__ __forceinline__ scalar_t ld_elem(const scalar_t* p) {
  if constexpr (COHERENT) {  // the aligned dword that holds the element
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint32_t w = __hip_atomic_load(reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint16_t h = (a & 2) ? (uint16_t)(w >> 16) : (uint16_t)(w & 0xffff);
    return __builtin_bit_cast(scalar_t, h);
  } else {
    return *p;
  }
}

// ---- the same reduce for at most 8 partitions, FOUR consecutive head dimensions of one (sequence, head) per lane, every load of a
// lane issued before the first use (one memory round trip): marlin_decode_kernel<ATTN>'s prologue, where one wave has several heads
// and rows to reduce and v2_reduce_head's wave-per-head form would walk them one after the other. Bit-identical to
// v2_reduce_head: the same per-element expressions; the sum of the rescale factors in the order wave_reduce_sum gives partitions
// sitting in lanes 0 .. 7 - ((r0 + r4) + (r2 + r6)) + ((r1 + r5) + (r3 + r7)) - and the maximum is order-free.
template <typename scalar_t> struct V2Vec4 {
  float ml[8], es[8];
  u32x2 t[8];
};
template <typename scalar_t, bool COHERENT = false>
__device__ __forceinline__ void v2_vec4_load(V2Vec4<scalar_t>& r, const float* __restrict__ exp_sums, const float* __restrict__ max_logits,
                                             const scalar_t* __restrict__ tp, int np, int head_size, int d0) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int jj = min(j, max(np - 1, 0));
    r.ml[j] = ld_f32<COHERENT>(max_logits + jj);
    r.es[j] = ld_f32<COHERENT>(exp_sums + jj);
    if constexpr (COHERENT)
      r.t[j] = __builtin_bit_cast(u32x2, __hip_atomic_load(reinterpret_cast<const uint64_t*>(tp + (int64_t)jj * head_size + d0), __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT));
    else
      r.t[j] = *reinterpret_cast<const u32x2*>(tp + (int64_t)jj * head_size + d0);
  }
}
template <typename scalar_t>
__device__ __forceinline__ u32x2 v2_vec4_math(const V2Vec4<scalar_t>& r, int np) {
  if (np <= 1) return r.t[0];
  float m = -FLT_MAX;
#pragma unroll
  for (int j = 0; j < 8; ++j) if (j < np) m = fmaxf(m, r.ml[j]);
  float resc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    resc[j] = (j < np) ? r.es[j] * __expf(r.ml[j] - m) : 0.f;
    // a value of its own, as in v2_reduce_head (where it goes through LDS): without this hipcc contracts the product into the
    // additions of the sum below (fma: one rounding less) and the last bit of the result can differ
    asm volatile("" : "+v"(resc[j]));
  }
  const float gsum = ((resc[0] + resc[4]) + (resc[2] + resc[6])) + ((resc[1] + resc[5]) + (resc[3] + resc[7]));
  const float inv = __fdividef(1.f, gsum + 1e-6f);
  union { u32x2 u; scalar_t e[4]; } o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < np) {
        union { u32x2 u; scalar_t e[4]; } v;
        v.u = r.t[j];
        acc += Scalar<scalar_t>::to_f32(v.e[i]) * resc[j] * inv;
      }
    }
    o.e[i] = Scalar<scalar_t>::from_f32(acc);
  }
  return o.u;
}

// ---- v2 reduce of one (sequence, head) by ONE wave: follows attention_kernels.cu:567-669. Shared by the reduce kernel and by
// the in-kernel reduce of the last-arriving partition (below), so the two forms write the same bits. resc: np floats of LDS. ----
template <typename scalar_t, bool COHERENT = false>
__device__ __forceinline__ void v2_reduce_head(scalar_t* __restrict__ o, const float* __restrict__ exp_sums,
                                               const float* __restrict__ max_logits, const scalar_t* __restrict__ tp, int np,
                                               int head_size, float* __restrict__ absmax_dst, float* resc, int lane) {
  float amax = 0.f;
  {
    // at most 8 partitions (every fine-partition launch, and 512-token partitions up to 4,096 tokens): the four-dimensions-per-lane
    // form, all loads of a lane in one round trip - the SAME function marlin_decode_kernel<ATTN> runs in its prologue, so the reduce
    // launch and the fused form agree bit for bit by construction
    if (sizeof(scalar_t) == 2 && np <= 8 && head_size % 4 == 0) {  // (16-bit outputs: four of them are the lane's 8 bytes)
      for (int c = lane; c < head_size / 4; c += 64) {
        V2Vec4<scalar_t> r;
        v2_vec4_load<scalar_t, COHERENT>(r, exp_sums, max_logits, tp, np, head_size, 4 * c);
        union { u32x2 u; scalar_t e[4]; } ov;
        ov.u = v2_vec4_math<scalar_t>(r, np);
        *reinterpret_cast<u32x2*>(o + 4 * c) = ov.u;
#pragma unroll
        for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fabsf(Scalar<scalar_t>::to_f32(ov.e[i])));
      }
      if (absmax_dst != nullptr) {
        amax = wave_reduce_max(amax);
        if (lane == 0) *absmax_dst = amax;
      }
      return;
    }
  }
  if (np <= 1) {
    for (int i = lane; i < head_size; i += 64) {
      const scalar_t v = ld_elem<COHERENT>(tp + i);
      o[i] = v;
      amax = fmaxf(amax, fabsf(Scalar<scalar_t>::to_f32(v)));
    }
    if (absmax_dst != nullptr) {
      amax = wave_reduce_max(amax);
      if (lane == 0) *absmax_dst = amax;
    }
    return;
  }
  float m = -FLT_MAX;
  for (int i = lane; i < np; i += 64) m = fmaxf(m, ld_f32<COHERENT>(max_logits + i));
  m = wave_reduce_max(m);
  float gsum = 0.f;
  for (int i = lane; i < np; i += 64) {
    const float r = ld_f32<COHERENT>(exp_sums + i) * __expf(ld_f32<COHERENT>(max_logits + i) - m);
    resc[i] = r;
    gsum += r;
  }
  gsum = wave_reduce_sum(gsum);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the wave's own LDS writes before its reads (one wave: no s_barrier)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float inv = __fdividef(1.f, gsum + 1e-6f);
  for (int d = lane; d < head_size; d += 64) {
    float acc = 0.f;
    // (the partitions' values in batches of 8 loads issued back to back - with one load per iteration every partition cost a
    //  memory round trip; the additions keep the order j = 0, 1, ...)
    for (int j0 = 0; j0 < np; j0 += 8) {
      scalar_t v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = ld_elem<COHERENT>(tp + (int64_t)min(j0 + i, np - 1) * head_size + d);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (j0 + i < np) acc += Scalar<scalar_t>::to_f32(v[i]) * resc[j0 + i] * inv;
    }
    const scalar_t ov = Scalar<scalar_t>::from_f32(acc);
    o[d] = ov;
    amax = fmaxf(amax, fabsf(Scalar<scalar_t>::to_f32(ov)));
  }
  if (absmax_dst != nullptr) {
    amax = wave_reduce_max(amax);
    if (lane == 0) *absmax_dst = amax;
  }
}


static inline int nmx_dtype_size(int dt) { return dt == NMX_F32 ? 4 : 2; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
