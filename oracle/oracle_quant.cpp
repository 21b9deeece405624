// TEST INFRASTRUCTURE ONLY — CPU oracle for the quantized-linear ops.
//
// Plain C++ restatement of what the reference's quantized GEMM kernels compute:
//   unpack (format-specific integer index math) -> dequantise to scalar_t -> fp32-accumulated matmul.
// The reference has no CPU kernel for these ops (cmake/cpu_extension.cmake:94-100); its own tests use
// "Python fake-quant + dense matmul" as the expected value (tests/kernels/test_marlin_gemm.py:153-172).
// This file restates the formats from the CUDA sources / Python packers cited at each function, and is
// pinned by golden vectors produced with the reference's Python utilities (tests/golden/gen_golden.py).
// Unpinned (no reference kernel test exists): awq_*, gptq_gemm/gptq_shuffle — see DESIGN.md.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "numfmt.h"

using namespace orc;

namespace {

// Inverse of the Marlin weight permutation (vllm/model_executor/layers/quantization/utils/marlin_perms.py:16-50
// + marlin_utils.py:25-57; element map in SURVEY.md appendix A.2).
// packed: [K/16, N*16/pf] int32.  q_out: [K, N] uint8 (raw codes, zero-point not removed).
void marlin_unpack(const int32_t* packed, int K, int N, int bits, uint8_t* q_out) {
  const int pf = 32 / bits;
  const int row_words = N * 16 / pf;
  const uint32_t mask = (1u << bits) - 1;
  // position p in a 1024-element chunk -> source element perm[p] of the (16 x 64) tile image
  // perm built exactly like get_perms()
  std::vector<int> perm;
  perm.reserve(1024);
  for (int i = 0; i < 32; ++i) {
    int perm1[8];
    int c = 0;
    const int col = i / 4;
    for (int block = 0; block < 2; ++block) {
      const int rows[4] = {2 * (i % 4), 2 * (i % 4) + 1, 2 * (i % 4 + 4), 2 * (i % 4 + 4) + 1};
      for (int r = 0; r < 4; ++r) perm1[c++] = 16 * rows[r] + col + 8 * block;
    }
    for (int j = 0; j < 4; ++j)
      for (int e = 0; e < 8; ++e) perm.push_back(perm1[e] + 256 * j);
  }
  static const int il4[8] = {0, 2, 4, 6, 1, 3, 5, 7};
  static const int il8[4] = {0, 2, 1, 3};
  const int ilen = bits == 4 ? 8 : 4;
  std::vector<int> permi(1024);
  for (int g = 0; g < 1024 / ilen; ++g)
    for (int e = 0; e < ilen; ++e) permi[g * ilen + e] = perm[g * ilen + (bits == 4 ? il4[e] : il8[e])];

#pragma omp parallel for
  for (int kt = 0; kt < K / 16; ++kt) {
    const int32_t* row = packed + (int64_t)kt * row_words;
    for (int64_t pos = 0; pos < (int64_t)N * 16; ++pos) {
      const uint32_t w = (uint32_t)row[pos / pf];
      const uint8_t v = (w >> (bits * (pos % pf))) & mask;
      // position pos of the permuted row came from element src of the tiled row
      const int64_t chunk = pos / 1024;
      const int src = permi[pos % 1024];
      const int64_t tiled = chunk * 1024 + src;  // index into [N/16 tiles][16 k][16 n]
      const int64_t ntile = tiled / 256;
      const int kin = (tiled % 256) / 16, nin = tiled % 16;
      q_out[((int64_t)kt * 16 + kin) * N + ntile * 16 + nin] = v;
    }
  }
}

// inverse of marlin_permute_scales (marlin_utils.py:60-69; scale_perm / scale_perm_single in marlin_perms.py:40-47)
void marlin_unpermute_scales(const void* s_in, int num_groups, int N, int dt, bool grouped, std::vector<float>& s_out) {
  s_out.resize((size_t)num_groups * N);
  int sp[64], sps[32];
  {
    int c = 0;
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 8; ++j) sp[c++] = i + 8 * j;
    c = 0;
    static const int o[8] = {0, 1, 8, 9, 16, 17, 24, 25};
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 8; ++j) sps[c++] = 2 * i + o[j];
  }
  const int L = grouped ? 64 : 32;
  const int* p = grouped ? sp : sps;
  const int64_t total = (int64_t)num_groups * N;
  for (int64_t base = 0; base < total; base += L)
    for (int e = 0; e < L; ++e) s_out[base + p[e]] = ld(s_in, base + e, dt);  // out[e] = in[p[e]]  => in[p[e]] = out[e]
}

// blocked fp32 matmul: C[M,N] = A[M,K] (float) * W[K,N] (float), result rounded to dt
void matmul_f32(const float* A, const float* W, int M, int N, int K, void* out, int dt_out) {
#pragma omp parallel for
  for (int nb = 0; nb < N; nb += 256) {
    const int ne = std::min(N, nb + 256);
    std::vector<float> acc((size_t)M * 256);
    std::fill(acc.begin(), acc.end(), 0.f);
    for (int k = 0; k < K; ++k) {
      const float* wr = W + (int64_t)k * N;
      for (int m = 0; m < M; ++m) {
        const float a = A[(int64_t)m * K + k];
        float* ar = acc.data() + (size_t)m * 256;
        for (int n = nb; n < ne; ++n) ar[n - nb] += a * wr[n];
      }
    }
    for (int m = 0; m < M; ++m)
      for (int n = nb; n < ne; ++n) st(out, (int64_t)m * N + n, dt_out, acc[(size_t)m * 256 + n - nb]);
  }
}

}  // namespace

extern "C" {

// ---- generic helpers exposed for tests ----
void orc_matmul(void* out, const void* a, const void* w, int M, int N, int K, int dt_a, int dt_w, int dt_out) {
  std::vector<float> A((size_t)M * K), W((size_t)K * N);
  for (int64_t i = 0; i < (int64_t)M * K; ++i) A[i] = ld(a, i, dt_a);
  for (int64_t i = 0; i < (int64_t)K * N; ++i) W[i] = ld(w, i, dt_w);
  matmul_f32(A.data(), W.data(), M, N, K, out, dt_out);
}

void orc_marlin_unpack(const int32_t* packed, int K, int N, int bits, uint8_t* q_out) {
  marlin_unpack(packed, K, N, bits, q_out);
}

// Dequantise a Marlin-format weight to dense scalar_t [K, N] in the *sorted* row order
// (gptq_marlin.cu:410-1363: w = (q - 2^(bits-1)) * s[g(k), n]; g(k) = k / group or g_idx[k]).
void orc_gptq_marlin_dequant(void* w_out, const int32_t* b_q_weight, const void* b_scales, const int32_t* g_idx,
                             int K, int N, int bits, int num_groups, int has_act_order, int is_k_full, int dt) {
  std::vector<uint8_t> q((size_t)K * N);
  marlin_unpack(b_q_weight, K, N, bits, q.data());
  // grouped scales use scale_perm; channel-wise (one row) use scale_perm_single (gptq_marlin.py:47-56)
  const bool grouped = num_groups > 1;
  std::vector<float> s;
  marlin_unpermute_scales(b_scales, num_groups, N, dt, grouped, s);
  const int zp = 1 << (bits - 1);
  const int group_size = (num_groups > 1) ? K / num_groups : K;
#pragma omp parallel for
  for (int k = 0; k < K; ++k) {
    int g;
    if (has_act_order) g = g_idx[k];
    else g = (num_groups > 1) ? k / group_size : 0;
    (void)is_k_full;
    for (int n = 0; n < N; ++n) {
      const float v = (float)((int)q[(int64_t)k * N + n] - zp) * s[(int64_t)g * N + n];
      st(w_out, (int64_t)k * N + n, dt, v);
    }
  }
}

// gptq_marlin_gemm (gptq_marlin.cu:1735-1868): C = A[:, perm] * dequant(B)
void orc_gptq_marlin_gemm(void* c, const void* a, const int32_t* b_q_weight, const void* b_scales,
                          const int32_t* g_idx, const int32_t* perm, int M, int N, int K, int bits, int num_groups,
                          int has_act_order, int is_k_full, int dt) {
  std::vector<uint16_t> wbuf;  // dense weight in scalar dtype
  std::vector<float> W((size_t)K * N), A((size_t)M * K);
  std::vector<uint8_t> tmp((size_t)K * N * dt_size(dt));
  orc_gptq_marlin_dequant(tmp.data(), b_q_weight, b_scales, g_idx, K, N, bits, num_groups, has_act_order, is_k_full, dt);
  for (int64_t i = 0; i < (int64_t)K * N; ++i) W[i] = ld(tmp.data(), i, dt);
  for (int m = 0; m < M; ++m)
    for (int k = 0; k < K; ++k) {
      const int src = has_act_order ? perm[k] : k;  // permute_cols_kernel, gptq_marlin.cu:345-394
      A[(int64_t)m * K + k] = ld(a, (int64_t)m * K + src, dt);
    }
  matmul_f32(A.data(), W.data(), M, N, K, c, dt);
}

// fp8_marlin_gemm (fp8/fp8_marlin.cu:1212-1308): weight byte = e4m3fn bit pattern, channel-wise scale
void orc_fp8_marlin_gemm(void* c, const void* a, const int32_t* b_q_weight, const void* b_scales, int M, int N, int K,
                         int dt) {
  std::vector<uint8_t> q((size_t)K * N);
  marlin_unpack(b_q_weight, K, N, 8, q.data());
  std::vector<float> s;
  marlin_unpermute_scales(b_scales, 1, N, dt, false, s);
  std::vector<float> W((size_t)K * N), A((size_t)M * K);
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) W[(int64_t)k * N + n] = rnd(e4m3_to_float(q[(int64_t)k * N + n]), dt);
  for (int64_t i = 0; i < (int64_t)M * K; ++i) A[i] = ld(a, i, dt);
  // scale applied to the fp32 accumulators (channel-wise 8-bit path, gptq_marlin.cu:1298-1323 analogue)
  std::vector<float> C((size_t)M * N);
  matmul_f32(A.data(), W.data(), M, N, K, C.data(), DT_F32);
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) st(c, (int64_t)m * N + n, dt, C[(int64_t)m * N + n] * s[n]);
}

// awq_dequantize (awq/gemm_kernels.cu:367-431, awq/dequantize.cuh:17-98):
// W[k, 8c+j] = (nib(qweight[k,c], o_j) - nib(qzeros[k/G,c], o_j)) * scales[k/G, 8c+j], o = [0,4,1,5,2,6,3,7]
void orc_awq_dequantize(uint16_t* w_out, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, int K,
                        int N, int G) {
  static const int order[8] = {0, 4, 1, 5, 2, 6, 3, 7};
  const int NC = N / 8;
#pragma omp parallel for
  for (int k = 0; k < K; ++k)
    for (int c = 0; c < NC; ++c) {
      const uint32_t q = (uint32_t)qweight[(int64_t)k * NC + c];
      const uint32_t z = (uint32_t)qzeros[(int64_t)(k / G) * NC + c];
      for (int j = 0; j < 8; ++j) {
        const int qi = (q >> (4 * order[j])) & 0xf;
        const int zi = (z >> (4 * order[j])) & 0xf;
        const float s = half_to_float(scales[(int64_t)(k / G) * N + 8 * c + j]);
        // sub.f16x2 (exact) then fma.rn.f16x2 with zero addend == one fp16 rounding of the product
        w_out[(int64_t)k * N + 8 * c + j] = float_to_half((float)(qi - zi) * s);
      }
    }
}

// awq_gemm (awq/gemm_kernels.cu:492-549): fp16 only
void orc_awq_gemm(uint16_t* c, const uint16_t* a, const int32_t* qweight, const uint16_t* scales,
                  const int32_t* qzeros, int M, int N, int K, int G) {
  std::vector<uint16_t> w((size_t)K * N);
  orc_awq_dequantize(w.data(), qweight, scales, qzeros, K, N, G);
  orc_matmul(c, a, w.data(), M, N, K, DT_F16, DT_F16, DT_F16);
}

// gptq dequant (gptq/q_gemm.cu:1387-1417 reconstruct_gptq_kernel): w = half(q - (z + 1)) * s
// qweight [K/pf, N] (element k at bits (k % pf) * bits of row k / pf), qzeros [groups, N/pf] packed along N,
// g_idx [K] or null (then group = k / (K / groups)). bits in {2, 3, 4, 8}: both tensors are contiguous bit streams of
// BITS-wide fields (3-bit: 32 fields per 3 words, fields 10 and 21 straddle a word boundary - MatrixView_q3_row,
// gptq/matrix_view.cuh).
static inline uint32_t gptq_field(const int32_t* base, int64_t stride, int idx, int bits) {
  const int64_t pos = (int64_t)bits * idx;
  const int64_t w = pos >> 5;
  const int sh = (int)(pos & 31);
  uint32_t v = (uint32_t)base[w * stride] >> sh;
  if (sh + bits > 32) v |= (uint32_t)base[(w + 1) * stride] << (32 - sh);
  return v & ((1u << bits) - 1u);
}
void orc_gptq_dequantize(uint16_t* w_out, const int32_t* qweight, const int32_t* qzeros, const uint16_t* scales,
                         const int32_t* g_idx, int K, int N, int groups, int bits) {
  const int gs = K / groups;
  const int64_t zwords = (int64_t)N * bits / 32;
#pragma omp parallel for
  for (int k = 0; k < K; ++k) {
    const int g = g_idx ? g_idx[k] : k / gs;
    for (int n = 0; n < N; ++n) {
      const int q = (int)gptq_field(qweight + n, N, k, bits);
      const int z = (int)gptq_field(qzeros + (int64_t)g * zwords, 1, n, bits) + 1;
      const float s = half_to_float(scales[(int64_t)g * N + n]);
      w_out[(int64_t)k * N + n] = float_to_half((float)(q - z) * s);
    }
  }
}

void orc_gptq_gemm(uint16_t* c, const uint16_t* a, const int32_t* qweight, const int32_t* qzeros,
                   const uint16_t* scales, const int32_t* g_idx, int M, int N, int K, int groups, int bits) {
  std::vector<uint16_t> w((size_t)K * N);
  orc_gptq_dequantize(w.data(), qweight, qzeros, scales, g_idx, K, N, groups, bits);
  orc_matmul(c, a, w.data(), M, N, K, DT_F16, DT_F16, DT_F16);
}

// scaled_fp8_quant (fp8/common.cu:24-125): dynamic: scale = max|x| / 448 ; out = e4m3fn(clamp(x * (1/scale), +-448))
void orc_scaled_fp8_quant(uint8_t* out, const void* x, float* scale, int64_t numel, int dt, int dynamic) {
  if (dynamic) {
    float m = 0.f;
    for (int64_t i = 0; i < numel; ++i) m = std::max(m, std::fabs(ld(x, i, dt)));
    *scale = m / 448.0f;
  }
  const float inv = 1.0f / *scale;
  for (int64_t i = 0; i < numel; ++i) {
    float v = ld(x, i, dt) * inv;
    v = std::fmax(-448.0f, std::fmin(v, 448.0f));
    out[i] = float_to_e4m3_sat(v);
  }
}

// static / dynamic-per-token int8 quant (compressed_tensors/int8_quant_kernels.cu:8-71)
void orc_scaled_int8_quant(int8_t* out, const void* x, float* scales, int num_tokens, int hidden, int dt,
                           int dynamic) {
  for (int t = 0; t < num_tokens; ++t) {
    if (dynamic) {
      float m = 0.f;
      for (int i = 0; i < hidden; ++i) m = std::max(m, std::fabs(ld(x, (int64_t)t * hidden + i, dt)));
      scales[t] = m / 127.0f;
      const float ts = 127.0f / m;
      for (int i = 0; i < hidden; ++i) {
        float r = std::nearbyintf(ld(x, (int64_t)t * hidden + i, dt) * ts);
        out[(int64_t)t * hidden + i] = (int8_t)std::max(-128.f, std::min(127.f, r));
      }
    } else {
      const float s = scales[0];
      for (int i = 0; i < hidden; ++i) {
        float r = std::nearbyintf(ld(x, (int64_t)t * hidden + i, dt) / s);
        out[(int64_t)t * hidden + i] = (int8_t)std::max(-128.f, std::min(127.f, r));
      }
    }
  }
}

// cutlass_scaled_mm semantics (cutlass_w8a8/scaled_mm_entry.cu:47-100; tests/kernels/test_cutlass.py:35-47):
// out = cast(a_scale (.) (A @ B) (.) b_scale) (+ bias). A [M,K] row-major, B given column-major: b_t is [N,K] row-major.
// is_fp8 != 0: operands are e4m3fn bytes; else int8.
void orc_scaled_mm(void* out, const uint8_t* a, const uint8_t* b_t, const float* a_scales, int a_per_row,
                   const float* b_scales, int b_per_col, const void* bias, int M, int N, int K, int is_fp8,
                   int dt_out) {
#pragma omp parallel for
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      float acc = 0.f;
      if (is_fp8) {
        for (int k = 0; k < K; ++k) acc += e4m3_to_float(a[(int64_t)m * K + k]) * e4m3_to_float(b_t[(int64_t)n * K + k]);
      } else {
        int32_t iacc = 0;
        for (int k = 0; k < K; ++k) iacc += (int32_t)(int8_t)a[(int64_t)m * K + k] * (int32_t)(int8_t)b_t[(int64_t)n * K + k];
        acc = (float)iacc;
      }
      float v = a_scales[a_per_row ? m : 0] * (b_scales[b_per_col ? n : 0] * acc);
      v = rnd(v, dt_out);
      if (bias) v = v + ld(bias, n, dt_out);
      st(out, (int64_t)m * N + n, dt_out, v);
    }
}

}  // extern "C"
