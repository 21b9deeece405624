// TEST INFRASTRUCTURE ONLY — part of the CPU oracle, never linked into the product library.
//
// Scalar number-format conversions used by the oracle: IEEE fp16, bfloat16, OCP fp8 e4m3fn / e5m2.
// All float->narrow conversions are round-to-nearest-even. fp8 conversions saturate to the largest
// finite value (the reference's CUDA path uses __NV_SATFINITE:
// csrc/quantization/fp8/nvidia/quant_utils.cuh:456-486).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// ---- fp16 ----
static inline float half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1f;
  uint32_t man = h & 0x3ffu;
  if (exp == 0) {
    if (man == 0) return u2f(sign);
    // subnormal: man * 2^-24
    float v = (float)man * 5.9604644775390625e-08f;
    return sign ? -v : v;
  }
  if (exp == 31) return u2f(sign | 0x7f800000u | (man << 13));
  return u2f(sign | ((exp + 112) << 23) | (man << 13));
}

static inline uint16_t float_to_half(float f) {
  uint32_t x = f2u(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) {  // inf / nan
    return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0));
  }
  if (ax >= 0x477ff000u) {  // >= 65520 -> inf (RNE)
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax < 0x38800000u) {  // < 2^-14: subnormal half or zero
    if (ax < 0x33000000u) return (uint16_t)sign;  // < 2^-25 -> 0 (2^-25 itself ties to even = 0)
    // value = m * 2^-24, round to nearest even
    float a = u2f(ax);
    float scaled = a * 16777216.0f;  // * 2^24, exact
    float r = std::nearbyintf(scaled);  // default rounding mode = RNE
    return (uint16_t)(sign | (uint32_t)r);
  }
  uint32_t mant = ax & 0x7fffffu;
  uint32_t exp = (ax >> 23) - 112;
  uint32_t h = (exp << 10) | (mant >> 13);
  uint32_t rem = mant & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
  return (uint16_t)(sign | h);
}

// ---- bf16 ----
static inline float bf16_to_float(uint16_t b) { return u2f((uint32_t)b << 16); }
static inline uint16_t float_to_bf16(float f) {
  uint32_t x = f2u(f);
  if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u);  // quiet nan
  uint32_t lsb = (x >> 16) & 1;
  x += 0x7fffu + lsb;
  return (uint16_t)(x >> 16);
}

// ---- fp8 e4m3fn (OCP): bias 7, no inf, max 448, nan = 0x7f/0xff ----
static inline float e4m3_to_float(uint8_t v) {
  uint32_t sign = v & 0x80u;
  uint32_t exp = (v >> 3) & 0xf;
  uint32_t man = v & 0x7;
  float r;
  if (exp == 0) {
    r = (float)man * 0.001953125f;  // man * 2^-9
  } else if (exp == 15 && man == 7) {
    r = NAN;
  } else {
    r = std::ldexp((float)(8 + man), (int)exp - 10);  // (1 + man/8) * 2^(exp-7)
  }
  return sign ? -r : r;
}

static inline uint8_t float_to_e4m3_sat(float f) {
  uint8_t sign = (f2u(f) >> 24) & 0x80u;
  if (std::isnan(f)) return sign | 0x7f;
  float a = std::fabs(f);
  if (a >= 448.0f) return sign | 0x7e;  // saturate to max finite (also covers inf)
  if (a < 0.015625f) {                  // < 2^-6: subnormal range, quantum 2^-9
    float r = std::nearbyintf(a * 512.0f);
    return sign | (uint8_t)r;  // r in [0, 8]; r == 8 encodes exp=1, man=0 which is exactly 2^-6
  }
  int e;
  float m = std::frexp(a, &e);  // a = m * 2^e, m in [0.5, 1)
  // a = (2m) * 2^(e-1); mantissa 3 bits: q = round(2m * 8) in [8, 16]
  float q = std::nearbyintf(m * 16.0f);
  int ee = e - 1;
  if (q >= 16.0f) { q = 8.0f; ee++; }
  int biased = ee + 7;
  uint32_t code = ((uint32_t)biased << 3) | ((uint32_t)q - 8);
  if (code > 0x7e) code = 0x7e;
  return sign | (uint8_t)code;
}

// ---- fp8 e5m2: bias 15, has inf (0x7c) / nan; max finite 57344 ----
static inline float e5m2_to_float(uint8_t v) {
  return half_to_float((uint16_t)v << 8);
}

static inline uint8_t float_to_e5m2_sat(float f) {
  uint8_t sign = (f2u(f) >> 24) & 0x80u;
  if (std::isnan(f)) return sign | 0x7f;
  float a = std::fabs(f);
  if (a >= 57344.0f) return sign | 0x7b;
  if (a < 6.103515625e-05f) {  // < 2^-14, quantum 2^-16
    float r = std::nearbyintf(a * 65536.0f);
    return sign | (uint8_t)r;
  }
  int e;
  float m = std::frexp(a, &e);
  float q = std::nearbyintf(m * 8.0f);  // 2 mantissa bits: q in [4, 8]
  int ee = e - 1;
  if (q >= 8.0f) { q = 4.0f; ee++; }
  int biased = ee + 15;
  uint32_t code = ((uint32_t)biased << 2) | ((uint32_t)q - 4);
  if (code > 0x7b) code = 0x7b;
  return sign | (uint8_t)code;
}

// ---- generic load/store by dtype code: 0 = f32, 1 = f16, 2 = bf16 ----
enum { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2 };
// kv cache encodings: 0 = same as scalar dtype ("auto"), 1 = fp8 e4m3, 2 = fp8 e5m2
enum { KV_AUTO = 0, KV_E4M3 = 1, KV_E5M2 = 2 };

static inline int dt_size(int dt) { return dt == DT_F32 ? 4 : 2; }

static inline float ld(const void* p, int64_t i, int dt) {
  switch (dt) {
    case DT_F32: return ((const float*)p)[i];
    case DT_F16: return half_to_float(((const uint16_t*)p)[i]);
    default: return bf16_to_float(((const uint16_t*)p)[i]);
  }
}
static inline void st(void* p, int64_t i, int dt, float v) {
  switch (dt) {
    case DT_F32: ((float*)p)[i] = v; break;
    case DT_F16: ((uint16_t*)p)[i] = float_to_half(v); break;
    default: ((uint16_t*)p)[i] = float_to_bf16(v); break;
  }
}
// round a float through the scalar dtype (what "cast to scalar_t" does)
static inline float rnd(float v, int dt) {
  switch (dt) {
    case DT_F32: return v;
    case DT_F16: return half_to_float(float_to_half(v));
    default: return bf16_to_float(float_to_bf16(v));
  }
}

static inline float fp8_to_float(uint8_t v, int kv) {
  return kv == KV_E4M3 ? e4m3_to_float(v) : e5m2_to_float(v);
}
static inline uint8_t float_to_fp8(float f, int kv) {
  return kv == KV_E4M3 ? float_to_e4m3_sat(f) : float_to_e5m2_sat(f);
}

}  // namespace orc
