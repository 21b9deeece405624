// TEST INFRASTRUCTURE ONLY — CPU oracle for the element-wise ops (RMSNorm, rotary embedding, gated activations).
// Restates the reference CUDA kernels step by step, including where intermediate values are rounded to scalar_t.
// Pinned against the reference's own CPU backend (csrc/cpu/{layernorm,pos_encoding,activation}.cpp in oracle/_ref)
// through tests/golden fixtures. Never linked into the product.
#include <cmath>
#include <cstdint>

#include "numfmt.h"

using namespace orc;

extern "C" {

// csrc/layernorm_kernels.cu:22-46 (rms_norm_kernel) / :258-291 (generic fused_add_rms_norm_kernel)
void orc_rms_norm(void* out, void* input, void* residual, const void* weight, float eps, int num_tokens, int hidden,
                  int dt, int fused_add) {
#pragma omp parallel for
  for (int t = 0; t < num_tokens; ++t) {
    const int64_t row = (int64_t)t * hidden;
    float var = 0.f;
    for (int i = 0; i < hidden; ++i) {
      float x = ld(input, row + i, dt);
      if (fused_add) {
        x = rnd(x + ld(residual, row + i, dt), dt);
        st(residual, row + i, dt, x);
      }
      var += x * x;
    }
    const float s = 1.0f / std::sqrt(var / (float)hidden + eps);
    for (int i = 0; i < hidden; ++i) {
      const float x = fused_add ? ld(residual, row + i, dt) : ld(input, row + i, dt);
      const float n = rnd(x * s, dt);
      st(out, row + i, dt, n * ld(weight, i, dt));
    }
  }
}

// csrc/pos_encoding_kernels.cu:10-122
void orc_rotary_embedding(const int64_t* positions, void* query, void* key, const void* cos_sin_cache,
                          const int64_t* offsets, int rot_dim, int64_t q_stride, int64_t k_stride, int num_tokens,
                          int num_heads, int num_kv_heads, int head_size, int is_neox, int dt) {
  const int embed = rot_dim / 2;
  for (int64_t t = 0; t < num_tokens; ++t) {
    int64_t pos = positions[t];
    if (offsets) pos += offsets[t];
    const int64_t cb = pos * rot_dim;
    for (int which = 0; which < 2; ++which) {
      void* arr = which == 0 ? query : key;
      const int heads = which == 0 ? num_heads : num_kv_heads;
      const int64_t stride = which == 0 ? q_stride : k_stride;
      for (int h = 0; h < heads; ++h)
        for (int ro = 0; ro < embed; ++ro) {
          const int64_t base = t * stride + (int64_t)h * head_size;
          const int xi = is_neox ? ro : 2 * ro;
          const int yi = is_neox ? embed + ro : 2 * ro + 1;
          const float c = ld(cos_sin_cache, cb + ro, dt), s = ld(cos_sin_cache, cb + embed + ro, dt);
          const float x = ld(arr, base + xi, dt), y = ld(arr, base + yi, dt);
          st(arr, base + xi, dt, rnd(x * c, dt) - rnd(y * s, dt));
          st(arr, base + yi, dt, rnd(y * c, dt) + rnd(x * s, dt));
        }
    }
  }
}

static float act(float f, int kind, int dt) {
  switch (kind) {
    case 0: return rnd(f / (1.0f + std::exp(-f)), dt);                                    // silu
    case 1: return rnd(f * 0.5f * (1.0f + std::erf(f * 0.70710678118654752440f)), dt);    // gelu
    case 2: {                                                                             // gelu_tanh
      const float beta = 1.41421356237309504880f * 1.12837916709551257390f * 0.5f;
      return rnd(0.5f * f * (1.0f + std::tanh(beta * (f + 0.044715f * (f * f * f)))), dt);
    }
    case 3: {  // gelu_new, scalar_t arithmetic step by step (activation_kernels.cu:113-118)
      const float x3 = rnd(rnd(f * f, dt) * f, dt);
      const float inner = rnd(f + rnd(0.044715f * x3, dt), dt);
      const float t = rnd(std::tanh(rnd(0.79788456f * inner, dt)), dt);
      return rnd(rnd(0.5f * f, dt) * rnd(1.0f + t, dt), dt);
    }
    case 4: {  // gelu_fast (:121-127)
      const float a = rnd(f * 0.79788456f, dt);
      const float b = rnd(1.0f + rnd(rnd(0.044715f * f, dt) * f, dt), dt);
      const float t = rnd(std::tanh(rnd(a * b, dt)), dt);
      return rnd(rnd(0.5f * f, dt) * rnd(1.0f + t, dt), dt);
    }
    default: return rnd(f / (1.0f + std::exp(-1.702f * f)), dt);                          // gelu_quick
  }
}

// csrc/activation_kernels.cu:12-24 (gated) / :83-93 (plain)
void orc_activation(void* out, const void* in, int num_tokens, int d, int kind, int gated, int dt) {
#pragma omp parallel for
  for (int t = 0; t < num_tokens; ++t)
    for (int i = 0; i < d; ++i) {
      if (gated) {
        const float a = act(ld(in, (int64_t)t * 2 * d + i, dt), kind, dt);
        st(out, (int64_t)t * d + i, dt, a * ld(in, (int64_t)t * 2 * d + d + i, dt));
      } else {
        st(out, (int64_t)t * d + i, dt, act(ld(in, (int64_t)t * d + i, dt), kind, dt));
      }
    }
}

}  // extern "C"
