// KV-cache write / copy / swap / fp8-convert kernels for gfx950.
// Replaces csrc/cache_kernels.cu of the reference (reshape_and_cache :153-278, reshape_and_cache_flash :207-314,
// copy_blocks :69-148, swap_blocks :24-63, convert_fp8 :318-389). Pure index math + byte moves: bit-exact.
//
// All kernels are HBM/latency bound. Layout notes (reference vllm/attention/ops/paged_attn.py:37-59):
//   K cache [NB, H, D/x, BS, x]  (x = 16 B worth of elements), V cache [NB, H, D, BS].
// For one token the K write is H * D/x chunks of 16 contiguous bytes (one 16-B store per lane); the V write is a
// scatter of single elements with stride BS (inherent to the layout the attention kernel reads).
#include "nmx_common.h"

namespace {

// One workgroup per token. Thread i handles 16-B chunk i of the token's [H*D] key row (x elements) and the same
// x elements of the value row.
template <typename scalar_t, typename cache_t, int KV>
__global__ void reshape_and_cache_kernel(const scalar_t* __restrict__ key, const scalar_t* __restrict__ value,
                                         cache_t* __restrict__ key_cache, cache_t* __restrict__ value_cache,
                                         const int64_t* __restrict__ slot_mapping, int64_t key_stride,
                                         int64_t value_stride, int num_heads, int head_size, int block_size, int x,
                                         float kv_scale) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;  // padding token
  const int64_t block_idx = slot / block_size;
  const int64_t block_off = slot % block_size;
  const int n = num_heads * head_size;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int h = i / head_size;
    const int d = i % head_size;
    const int64_t tk = block_idx * num_heads * (head_size / x) * block_size * x +
                       (int64_t)h * (head_size / x) * block_size * x + (int64_t)(d / x) * block_size * x +
                       block_off * x + (d % x);
    const int64_t tv = block_idx * num_heads * head_size * block_size + (int64_t)h * head_size * block_size +
                       (int64_t)d * block_size + block_off;
    const scalar_t k = key[token * key_stride + i];
    const scalar_t v = value[token * value_stride + i];
    if constexpr (KV == NMX_KV_AUTO) {
      key_cache[tk] = k;
      value_cache[tv] = v;
    } else {
      key_cache[tk] = f32_to_fp8_sat<KV>(Scalar<scalar_t>::to_f32(k) / kv_scale);
      value_cache[tv] = f32_to_fp8_sat<KV>(Scalar<scalar_t>::to_f32(v) / kv_scale);
    }
  }
}

// Vectorised variant for the common case (auto dtype, 16-B aligned rows): each thread moves one 16-B K chunk
// with a single dwordx4 load/store and scatters the same x value elements.
template <typename scalar_t>
__global__ void reshape_and_cache_vec_kernel(const scalar_t* __restrict__ key, const scalar_t* __restrict__ value,
                                             scalar_t* __restrict__ key_cache, scalar_t* __restrict__ value_cache,
                                             const int64_t* __restrict__ slot_mapping, int64_t key_stride,
                                             int64_t value_stride, int num_heads, int head_size, int block_size) {
  constexpr int X = 16 / sizeof(scalar_t);
  const int64_t token = blockIdx.x;
  const int chunks_per_head = head_size / X;
  const int nchunks = num_heads * chunks_per_head;
  // the token's key / value chunks do not depend on the slot: request the first ones before the slot_mapping read so
  // that the two round trips overlap (the kernel is pure latency: 4 KB per token)
  u32x4 kv0 = {0, 0, 0, 0}, vv0 = {0, 0, 0, 0};
  if ((int)threadIdx.x < nchunks) {
    kv0 = *reinterpret_cast<const u32x4*>(key + token * key_stride + (int64_t)threadIdx.x * X);
    vv0 = *reinterpret_cast<const u32x4*>(value + token * value_stride + (int64_t)threadIdx.x * X);
  }
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;
  const int64_t block_idx = slot / block_size;
  const int64_t block_off = slot % block_size;
  for (int c = threadIdx.x; c < nchunks; c += blockDim.x) {
    const int h = c / chunks_per_head;
    const int xc = c % chunks_per_head;
    const bool first = c == (int)threadIdx.x;
    const u32x4 kv = first ? kv0 : *reinterpret_cast<const u32x4*>(key + token * key_stride + (int64_t)c * X);
    const int64_t tk = ((block_idx * num_heads + h) * chunks_per_head + xc) * block_size * X + block_off * X;
    *reinterpret_cast<u32x4*>(key_cache + tk) = kv;
    union { u32x4 v; scalar_t e[X]; } vv;
    vv.v = first ? vv0 : *reinterpret_cast<const u32x4*>(value + token * value_stride + (int64_t)c * X);
    const int64_t tv = ((block_idx * num_heads + h) * head_size + (int64_t)xc * X) * block_size + block_off;
#pragma unroll
    for (int j = 0; j < X; ++j) value_cache[tv + (int64_t)j * block_size] = vv.e[j];
  }
}

template <typename T>
__global__ void reshape_and_cache_flash_kernel(const T* __restrict__ key, const T* __restrict__ value,
                                               T* __restrict__ k_cache, T* __restrict__ v_cache,
                                               const int64_t* __restrict__ slot_mapping, int64_t block_stride,
                                               int64_t key_stride, int64_t value_stride, int num_heads,
                                               int head_size, int block_size) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;
  const int64_t block_idx = slot / block_size;
  const int64_t block_off = slot % block_size;
  const int n = num_heads * head_size;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int64_t tgt = block_idx * block_stride + block_off * n + i;
    k_cache[tgt] = key[token * key_stride + i];
    v_cache[tgt] = value[token * value_stride + i];
  }
}

// grid (num_pairs, num_layers, 2 {K,V}); 16-B vector copy, block_bytes % 16 == 0
__global__ void copy_blocks_kernel(void* const* __restrict__ key_ptrs, void* const* __restrict__ value_ptrs,
                                   const int64_t* __restrict__ block_mapping, int64_t block_vecs) {
  const int pair = blockIdx.x;
  const int layer = blockIdx.y;
  u32x4* base = reinterpret_cast<u32x4*>(blockIdx.z == 0 ? key_ptrs[layer] : value_ptrs[layer]);
  const int64_t src = block_mapping[2 * pair];
  const int64_t dst = block_mapping[2 * pair + 1];
  const u32x4* s = base + src * block_vecs;
  u32x4* d = base + dst * block_vecs;
  for (int64_t i = threadIdx.x; i < block_vecs; i += blockDim.x) d[i] = s[i];
}

__global__ void copy_blocks_bytes_kernel(void* const* __restrict__ key_ptrs, void* const* __restrict__ value_ptrs,
                                         const int64_t* __restrict__ block_mapping, int64_t block_bytes) {
  const int pair = blockIdx.x;
  const int layer = blockIdx.y;
  uint8_t* base = reinterpret_cast<uint8_t*>(blockIdx.z == 0 ? key_ptrs[layer] : value_ptrs[layer]);
  const int64_t src = block_mapping[2 * pair];
  const int64_t dst = block_mapping[2 * pair + 1];
  for (int64_t i = threadIdx.x; i < block_bytes; i += blockDim.x) base[dst * block_bytes + i] = base[src * block_bytes + i];
}

template <typename scalar_t, int KV, bool TO_FP8>
__global__ void convert_fp8_kernel(void* __restrict__ dst, const void* __restrict__ src, int64_t numel, float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    if constexpr (TO_FP8) {
      const float v = Scalar<scalar_t>::to_f32(reinterpret_cast<const scalar_t*>(src)[i]);
      reinterpret_cast<uint8_t*>(dst)[i] = f32_to_fp8_sat<KV>(v / scale);
    } else {
      float v = fp8_to_f32<KV>(reinterpret_cast<const uint8_t*>(src)[i]) * scale;
      // keep the multiply and the narrowing conversion separate: hipcc otherwise fuses them into
      // v_fma_mixlo_f16(scale, v, +0), which turns -0 into +0 (the reference keeps the sign of zero)
      asm volatile("" : "+v"(v));
      reinterpret_cast<scalar_t*>(dst)[i] = Scalar<scalar_t>::from_f32(v);
    }
  }
}

template <typename scalar_t>
int launch_reshape_and_cache(const void* key, const void* value, void* key_cache, void* value_cache,
                             const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
                             int block_size, int x, int64_t key_stride, int64_t value_stride, int kv_dtype,
                             float kv_scale, hipStream_t stream) {
  const int n = num_heads * head_size;
  dim3 grid(num_tokens);
  if (kv_dtype == NMX_KV_AUTO) {
    constexpr int X = 16 / sizeof(scalar_t);
    const bool vec_ok = (x == X) && (head_size % X == 0) && (key_stride % X == 0) && (value_stride % X == 0) &&
                        (((uintptr_t)key | (uintptr_t)value | (uintptr_t)key_cache) % 16 == 0);
    if (vec_ok) {
      const int nchunks = n / X;
      dim3 block(std::min(std::max(64, ((nchunks + 63) / 64) * 64), 256));
      reshape_and_cache_vec_kernel<scalar_t><<<grid, block, 0, stream>>>(
          (const scalar_t*)key, (const scalar_t*)value, (scalar_t*)key_cache, (scalar_t*)value_cache, slot_mapping,
          key_stride, value_stride, num_heads, head_size, block_size);
    } else {
      dim3 block(std::min(((n + 63) / 64) * 64, 512));
      reshape_and_cache_kernel<scalar_t, scalar_t, NMX_KV_AUTO><<<grid, block, 0, stream>>>(
          (const scalar_t*)key, (const scalar_t*)value, (scalar_t*)key_cache, (scalar_t*)value_cache, slot_mapping,
          key_stride, value_stride, num_heads, head_size, block_size, x, kv_scale);
    }
  } else {
    dim3 block(std::min(((n + 63) / 64) * 64, 512));
    if (kv_dtype == NMX_KV_FP8_E4M3)
      reshape_and_cache_kernel<scalar_t, uint8_t, NMX_KV_FP8_E4M3><<<grid, block, 0, stream>>>(
          (const scalar_t*)key, (const scalar_t*)value, (uint8_t*)key_cache, (uint8_t*)value_cache, slot_mapping,
          key_stride, value_stride, num_heads, head_size, block_size, x, kv_scale);
    else
      reshape_and_cache_kernel<scalar_t, uint8_t, NMX_KV_FP8_E5M2><<<grid, block, 0, stream>>>(
          (const scalar_t*)key, (const scalar_t*)value, (uint8_t*)key_cache, (uint8_t*)value_cache, slot_mapping,
          key_stride, value_stride, num_heads, head_size, block_size, x, kv_scale);
  }
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

}  // namespace

extern "C" int nmx_reshape_and_cache(const void* key, const void* value, void* key_cache, void* value_cache,
                                     const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
                                     int block_size, int x, int64_t key_stride, int64_t value_stride, int dtype,
                                     int kv_dtype, float kv_scale, nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NMX_CHECK(kv_dtype == NMX_KV_AUTO || kv_dtype == NMX_KV_FP8_E4M3 || kv_dtype == NMX_KV_FP8_E5M2,
            NMX_ERR_UNSUPPORTED, "Unsupported data type of kv cache: %d", kv_dtype);
  NMX_CHECK(num_tokens >= 0 && num_heads > 0 && head_size > 0 && block_size > 0 && x > 0 && head_size % x == 0,
            NMX_ERR_INVALID_ARG, "reshape_and_cache: bad shape (tokens=%d heads=%d head_size=%d block=%d x=%d)",
            num_tokens, num_heads, head_size, block_size, x);
  if (num_tokens == 0) return NMX_OK;
  switch (dtype) {
    case NMX_F32:
      return launch_reshape_and_cache<float>(key, value, key_cache, value_cache, slot_mapping, num_tokens, num_heads,
                                             head_size, block_size, x, key_stride, value_stride, kv_dtype, kv_scale, stream);
    case NMX_F16:
      return launch_reshape_and_cache<f16>(key, value, key_cache, value_cache, slot_mapping, num_tokens, num_heads,
                                           head_size, block_size, x, key_stride, value_stride, kv_dtype, kv_scale, stream);
    case NMX_BF16:
      return launch_reshape_and_cache<bf16>(key, value, key_cache, value_cache, slot_mapping, num_tokens, num_heads,
                                            head_size, block_size, x, key_stride, value_stride, kv_dtype, kv_scale, stream);
    default:
      NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "Unsupported input type of kv cache: %d", dtype);
  }
}

extern "C" int nmx_reshape_and_cache_flash(const void* key, const void* value, void* k_cache, void* v_cache,
                                           const int64_t* slot_mapping, int num_tokens, int num_heads,
                                           int head_size, int block_size, int64_t block_stride, int64_t key_stride,
                                           int64_t value_stride, int elem_size, nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NMX_CHECK(elem_size == 2 || elem_size == 4, NMX_ERR_UNSUPPORTED, "reshape_and_cache_flash: element size %d", elem_size);
  if (num_tokens == 0) return NMX_OK;
  const int n = num_heads * head_size;
  dim3 grid(num_tokens), block(std::min(((n + 63) / 64) * 64, 512));
  if (elem_size == 2)
    reshape_and_cache_flash_kernel<uint16_t><<<grid, block, 0, stream>>>(
        (const uint16_t*)key, (const uint16_t*)value, (uint16_t*)k_cache, (uint16_t*)v_cache, slot_mapping,
        block_stride, key_stride, value_stride, num_heads, head_size, block_size);
  else
    reshape_and_cache_flash_kernel<uint32_t><<<grid, block, 0, stream>>>(
        (const uint32_t*)key, (const uint32_t*)value, (uint32_t*)k_cache, (uint32_t*)v_cache, slot_mapping,
        block_stride, key_stride, value_stride, num_heads, head_size, block_size);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int nmx_copy_blocks(void* const* key_cache_ptrs, void* const* value_cache_ptrs,
                               const int64_t* block_mapping, int num_layers, int num_pairs, int64_t block_bytes,
                               nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (num_layers == 0 || num_pairs == 0) return NMX_OK;
  NMX_CHECK(block_bytes > 0, NMX_ERR_INVALID_ARG, "copy_blocks: block_bytes = %lld", (long long)block_bytes);
  dim3 grid(num_pairs, num_layers, 2);
  if (block_bytes % 16 == 0) {
    const int64_t vecs = block_bytes / 16;
    dim3 block((int)std::min<int64_t>(1024, ((vecs + 63) / 64) * 64));
    copy_blocks_kernel<<<grid, block, 0, stream>>>(key_cache_ptrs, value_cache_ptrs, block_mapping, vecs);
  } else {
    copy_blocks_bytes_kernel<<<grid, dim3(256), 0, stream>>>(key_cache_ptrs, value_cache_ptrs, block_mapping, block_bytes);
  }
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int nmx_swap_blocks(const void* src, void* dst, const int64_t* block_mapping_host, int num_pairs,
                               int64_t block_bytes, int copy_kind, nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  hipMemcpyKind kind;
  switch (copy_kind) {
    case NMX_COPY_D2D: kind = hipMemcpyDeviceToDevice; break;
    case NMX_COPY_D2H: kind = hipMemcpyDeviceToHost; break;
    case NMX_COPY_H2D: kind = hipMemcpyHostToDevice; break;
    default: NMX_CHECK(false, NMX_ERR_INVALID_ARG, "Invalid device combination");
  }
  for (int i = 0; i < num_pairs; ++i) {
    const int64_t s = block_mapping_host[2 * i], d = block_mapping_host[2 * i + 1];
    NMX_HIP(hipMemcpyAsync((char*)dst + d * block_bytes, (const char*)src + s * block_bytes, block_bytes, kind, stream));
  }
  return NMX_OK;
}

extern "C" int nmx_convert_fp8(void* dst, const void* src, int64_t numel, float scale, int dtype, int kv_dtype,
                               int to_fp8, nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NMX_CHECK(kv_dtype == NMX_KV_FP8_E4M3 || kv_dtype == NMX_KV_FP8_E5M2, NMX_ERR_UNSUPPORTED,
            "Unsupported data type: %d", kv_dtype);
  if (numel == 0) return NMX_OK;
  const int threads = 256;
  const int blocks = (int)std::min<int64_t>((numel + threads - 1) / threads, 2048);
#define NMX_CVT(T, KV)                                                                                   \
  do {                                                                                                   \
    if (to_fp8) convert_fp8_kernel<T, KV, true><<<blocks, threads, 0, stream>>>(dst, src, numel, scale); \
    else convert_fp8_kernel<T, KV, false><<<blocks, threads, 0, stream>>>(dst, src, numel, scale);       \
  } while (0)
#define NMX_CVT_KV(T)                                          \
  do {                                                         \
    if (kv_dtype == NMX_KV_FP8_E4M3) NMX_CVT(T, NMX_KV_FP8_E4M3); \
    else NMX_CVT(T, NMX_KV_FP8_E5M2);                          \
  } while (0)
  switch (dtype) {
    case NMX_F32: NMX_CVT_KV(float); break;
    case NMX_F16: NMX_CVT_KV(f16); break;
    case NMX_BF16: NMX_CVT_KV(bf16); break;
    default: NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "convert_fp8: unsupported dtype %d", dtype);
  }
#undef NMX_CVT_KV
#undef NMX_CVT
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}
