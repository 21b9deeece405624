// Zero-point int4 formats: AWQ (awq_gemm, awq_dequantize) and GPTQ / exllama (gptq_gemm, gptq_shuffle) for gfx950.
// Replaces csrc/quantization/awq/gemm_kernels.cu and csrc/quantization/gptq/q_gemm.cu (2 / 3 / 4 / 8 bit) of the reference.
// fp16 only, like the reference (awq.py:43-44, gptq.py).
//
// Both formats are consumed as stored (no repack pass); w = (q - z) * s is evaluated in fp16 exactly as the reference
// does (exact integer difference, one fp16 rounding of the product), then fed to MFMA 16x16x32 with fp32 accumulation.
//  * GPTQ  qweight [K/8, N]: one int32 = 8 consecutive k of one column = one MFMA operand fragment. A lane loads 16 B
//          = 4 adjacent columns; after gptq_shuffle the nibble order (even k in the low half, odd k in the high half)
//          makes (q & 0x000f000f) yield the (k, k+1) pair in operand order.
//  * AWQ   qweight [K, N/8]: one int32 = 8 columns of one k. A lane gathers the 8 k-rows of its 8-column chunk (8 dword
//          loads), converts nibble PAIRS (two columns at once) and transposes pairs of k with v_perm_b32.
// Workgroup = 4 waves splitting K (LDS reduce); grid.y splits K further (fp32 partial slabs + reduce kernel).
#include <stdlib.h>

#include "nmx_common.h"

namespace {

__device__ __forceinline__ uint32_t and_or(uint32_t q, uint32_t mask, uint32_t magic) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(q), "s"(mask), "v"(magic));
  return r;
}
__device__ __forceinline__ f16x2 h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ uint32_t u32(f16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

struct ZpParams {
  const f16* a;            // [M, K]
  const uint32_t* qweight;
  const uint32_t* qzeros;
  const f16* scales;
  const int32_t* g_idx;    // gptq: [K] or null
  const int32_t* q_perm;   // gptq exllama act-order: [K] or null (A column gather)
  f16* c;
  float* partial;
  int M, N, K, G;          // G = group size
  int k_splits;
  int shuffled;            // gptq: weights went through gptq_shuffle
};

// activation fragment: lane (g, li) <- A[m0 + li][k .. k+7], natural k order, optional column gather
template <bool GATHER>
__device__ __forceinline__ u32x4 load_a_frag(const ZpParams& p, int m, int k) {
  u32x4 v = {0, 0, 0, 0};
  if (m < p.M && k < p.K) {
    if constexpr (!GATHER) {
      v = *reinterpret_cast<const u32x4*>(p.a + (int64_t)m * p.K + k);
    } else {
      union { f16 h[8]; u32x4 u; } t;
#pragma unroll
      for (int e = 0; e < 8; ++e) t.h[e] = p.a[(int64_t)m * p.K + p.q_perm[k + e]];
      v = t.u;
    }
  }
  return v;
}

template <int MT, int NACC>
__device__ __forceinline__ void reduce_waves(f32x4 (&acc)[MT][NACC], char* smem, int wave, int lane) {
  f32x4* red = reinterpret_cast<f32x4*>(smem);  // [3][MT][NACC][64]
  if (wave > 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < NACC; ++t) red[(((wave - 1) * MT + mt) * NACC + t) * 64 + lane] = acc[mt][t];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[mt][t] += red[((w * MT + mt) * NACC + t) * 64 + lane];
  }
}

// ---- AWQ ------------------------------------------------------------------------------------------------------------
// grid (N / 128, k_splits, ceil(M / (16 MT))); wave = 16 chunks (128 columns) x K-slice
template <int MT>
__global__ __launch_bounds__(256) void awq_gemm_kernel(const ZpParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int NC = p.N / 8;
  const int c = blockIdx.x * 16 + li;         // this lane's 8-column chunk
  const bool c_ok = c < NC;
  const int cc = c_ok ? c : 0;
  const int m0 = blockIdx.z * 16 * MT;
  f32x4 acc[MT][8];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[mt][e] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int steps = p.K / 32;
  const int workers = p.k_splits * 4;
  const int per = (steps + workers - 1) / workers;
  const int s0 = min((blockIdx.y * 4 + wave) * per, steps), s1 = min(s0 + per, steps);
  const uint32_t MAGIC = 0x64006400u;
  int cur_grp = -1;
  uint32_t zmag[4] = {0, 0, 0, 0};
  uint32_t sc[4] = {0, 0, 0, 0};
  for (int s = s0; s < s1; ++s) {
    const int kb = s * 32 + 8 * g;
    uint32_t r[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) r[jj] = p.qweight[(int64_t)(kb + jj) * NC + cc];
    const int grp = (s * 32) / p.G;  // G % 32 == 0: one group per k-step
    if (grp != cur_grp) {
      cur_grp = grp;
      const uint32_t z = p.qzeros[(int64_t)grp * NC + cc];
      const u32x4 sv = *reinterpret_cast<const u32x4*>(p.scales + (int64_t)grp * p.N + 8 * cc);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        zmag[d] = and_or(z >> (4 * d), 0x000f000fu, MAGIC);  // (1024 + z_2d, 1024 + z_2d+1): nibbles d and d + 4
        sc[d] = sv[d];                                        // (s_2d, s_2d+1): columns 8c + 2d, 8c + 2d + 1
      }
    }
    u32x4 af[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = load_a_frag<false>(p, m0 + 16 * mt + li, kb);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      // x[jj] = ((q - z) * s) for columns (2d, 2d+1) at k = kb + jj  (awq/gemm_kernels.cu:401-424: sub then fma with 0)
      uint32_t x[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) x[jj] = u32((h2(and_or(r[jj] >> (4 * d), 0x000f000fu, MAGIC)) - h2(zmag[d])) * h2(sc[d]));
      u32x4 w_lo, w_hi;  // operand fragments of column 2d (low halves) and 2d + 1 (high halves)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        w_lo[q] = __builtin_amdgcn_perm(x[2 * q + 1], x[2 * q], 0x05040100u);
        w_hi[q] = __builtin_amdgcn_perm(x[2 * q + 1], x[2 * q], 0x07060302u);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt][2 * d] = mfma_f16(w_lo, af[mt], acc[mt][2 * d]);
        acc[mt][2 * d + 1] = mfma_f16(w_hi, af[mt], acc[mt][2 * d + 1]);
      }
    }
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  reduce_waves<MT, 8>(acc, smem, wave, lane);
  if (wave != 0) return;
  // D[n-slot][m]: lane (g, li = m) holds rows 4 g + r = chunks c0 + 4 g + r; tile e = column e of the chunk
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int chunk = blockIdx.x * 16 + 4 * g + r;
      if (chunk >= NC) continue;
      if (p.k_splits == 1) {
        union { f16 h[8]; u32x4 u; } o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.h[e] = (f16)acc[mt][e][r];
        *reinterpret_cast<u32x4*>(p.c + (int64_t)m * p.N + 8 * chunk) = o.u;
      } else {
        float* dst = p.partial + ((int64_t)blockIdx.y * p.M + m) * p.N + 8 * chunk;
        *reinterpret_cast<f32x4*>(dst) = f32x4{acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{acc[mt][4][r], acc[mt][5][r], acc[mt][6][r], acc[mt][7][r]};
      }
    }
  }
}

// ---- AWQ, group % 128 == 0 and K % 128 == 0: same arithmetic as awq_gemm_kernel with two 128-k units of operands in
// flight per wave (see gptq_gemm_ring_kernel). grid (N / 128, k_splits, ceil(M / (16 MT))).
template <int MT>
__global__ __launch_bounds__(256, 2) void awq_gemm_ring_kernel(const ZpParams p) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int NC = p.N / 8, K = p.K;
  const int c = blockIdx.x * 16 + li;  // this lane's 8-column chunk
  const int cc = c < NC ? c : 0;
  const int m0 = blockIdx.z * 16 * MT;
  f32x4 acc[MT][8];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[mt][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int units = K / 128;
  const int workers = p.k_splits * 4;
  const int per = (units + workers - 1) / workers;
  const int u0 = min(((int)blockIdx.y * 4 + wave) * per, units), u1 = min(u0 + per, units);
  const int groups = K / p.G;
  const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.qweight), 0, K * NC * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.a), 0, p.M * K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.qzeros), 0, groups * NC * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.scales), 0, groups * p.N * 2, 0x00020000);
  const int q_voff = (8 * g * NC + cc) * 4;  // k-row 8 g (+ jj through the scalar offset) of the 32-k step, this lane's chunk
  int a_voff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + li;
    a_voff[mt] = m < p.M ? (m * K + 8 * g) * 2 : (int)0x7ff00000;
  }
  struct Unit { uint32_t r[4][8]; u32x4 a[4][MT]; uint32_t z; u32x4 s; };
  auto load_step = [&](int u, int ks, Unit& U) {
    u = min(u, units - 1);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) U.r[ks][jj] = __builtin_amdgcn_raw_buffer_load_b32(rs_q, q_voff, (u * 128 + 32 * ks + jj) * NC * 4, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) U.a[ks][mt] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff[mt], (u * 128 + 32 * ks) * 2, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto load_group = [&](int u, Unit& U) {
    u = min(u, units - 1);
    const int grp = (u * 128) / p.G;
    U.z = __builtin_amdgcn_raw_buffer_load_b32(rs_z, cc * 4, grp * NC * 4, 0);
    U.s = __builtin_amdgcn_raw_buffer_load_b128(rs_s, cc * 16, grp * p.N * 2, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto load_unit = [&](int u, Unit& U) {
    load_group(u, U);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) load_step(u, ks, U);
  };
  const uint32_t MAGIC = 0x64006400u;
  auto compute_unit = [&](Unit& U, int next, bool keep) {
    const uint32_t z = U.z;
    const u32x4 sv = U.s;
    load_group(next, U);
    uint32_t zmag[4], sc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      zmag[d] = and_or(z >> (4 * d), 0x000f000fu, MAGIC);  // (1024 + z_2d, 1024 + z_2d+1): nibbles d and d + 4
      sc[d] = keep ? sv[d] : 0u;                            // (s_2d, s_2d+1); a dropped unit contributes (q - z) * 0
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint32_t x[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) x[jj] = u32((h2(and_or(U.r[ks][jj] >> (4 * d), 0x000f000fu, MAGIC)) - h2(zmag[d])) * h2(sc[d]));
        u32x4 w_lo, w_hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          w_lo[q] = __builtin_amdgcn_perm(x[2 * q + 1], x[2 * q], 0x05040100u);
          w_hi[q] = __builtin_amdgcn_perm(x[2 * q + 1], x[2 * q], 0x07060302u);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[mt][2 * d] = mfma_f16(w_lo, U.a[ks][mt], acc[mt][2 * d]);
          acc[mt][2 * d + 1] = mfma_f16(w_hi, U.a[ks][mt], acc[mt][2 * d + 1]);
        }
      }
      load_step(next, ks, U);
    }
  };
  if (u0 < u1) {
    Unit ua, ub;
    load_unit(u0, ua);
    load_unit(u0 + 1, ub);
    for (int u = u0; u < u1; u += 2) {
      compute_unit(ua, u + 2, true);
      compute_unit(ub, u + 3, u + 1 < u1);
    }
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  reduce_waves<MT, 8>(acc, smem, wave, lane);
  if (wave != 0) return;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int chunk = blockIdx.x * 16 + 4 * g + r;
      if (chunk >= NC) continue;
      if (p.k_splits == 1) {
        union { f16 h[8]; u32x4 u; } o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.h[e] = (f16)acc[mt][e][r];
        *reinterpret_cast<u32x4*>(p.c + (int64_t)m * p.N + 8 * chunk) = o.u;
      } else {
        float* dst = p.partial + ((int64_t)blockIdx.y * p.M + m) * p.N + 8 * chunk;
        *reinterpret_cast<f32x4*>(dst) = f32x4{acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{acc[mt][4][r], acc[mt][5][r], acc[mt][6][r], acc[mt][7][r]};
      }
    }
  }
}

// awq_dequantize (awq/gemm_kernels.cu:367-431): one thread per packed word -> 8 fp16
__global__ void awq_dequantize_kernel(const uint32_t* __restrict__ qweight, const f16* __restrict__ scales,
                                      const uint32_t* __restrict__ qzeros, f16* __restrict__ out, int K, int NC, int G) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)K * NC) return;
  const int k = idx / NC, c = idx % NC;
  const uint32_t q = qweight[idx], z = qzeros[(int64_t)(k / G) * NC + c];
  const u32x4 sv = *reinterpret_cast<const u32x4*>(scales + ((int64_t)(k / G) * NC + c) * 8);
  u32x4 o;
#pragma unroll
  for (int d = 0; d < 4; ++d)
    o[d] = u32((h2(and_or(q >> (4 * d), 0x000f000fu, 0x64006400u)) - h2(and_or(z >> (4 * d), 0x000f000fu, 0x64006400u))) * h2(sv[d]));
  *reinterpret_cast<u32x4*>(out + idx * 8) = o;
}

// ---- GPTQ ---------------------------------------------------------------------------------------------------------
// grid (N / 64, k_splits, ceil(M / (16 MT))); lane (g, li): 16-B load = columns n0 + 4 li .. + 3 of packed row ks*4 + g
template <int MT, bool SHUFFLED, bool GATHER, bool PER_ROW_GROUP>
__global__ __launch_bounds__(256) void gptq_gemm_kernel(const ZpParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int n0 = blockIdx.x * 64;
  const int nl = n0 + 4 * li;  // first of this lane's 4 columns
  const int m0 = blockIdx.z * 16 * MT;
  f32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int steps = p.K / 32;
  const int workers = p.k_splits * 4;
  const int per = (steps + workers - 1) / workers;
  const int s0 = min((blockIdx.y * 4 + wave) * per, steps), s1 = min(s0 + per, steps);
  const uint32_t MAGIC = 0x64006400u;
  int cur_grp = -1;
  uint32_t zmag[4] = {0, 0, 0, 0}, sc[4] = {0, 0, 0, 0};
  auto load_group = [&](int grp) {
    // qzeros [groups, N/8]: this lane's 4 columns share one word (4 li % 8 = 0 or 4); stored value is z - 1
    const uint32_t zw = p.qzeros[(int64_t)grp * (p.N / 8) + nl / 8] >> (4 * (nl & 7));
    const u32x2 sv = *reinterpret_cast<const u32x2*>(p.scales + (int64_t)grp * p.N + nl);
    union { u32x2 u; f16 h[4]; } su;
    su.u = sv;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t z = ((zw >> (4 * t)) & 0xf) + 1;  // q_gemm.cu:1408 (zero + 1)
      zmag[t] = 0x64006400u + z * 0x00010001u;
      union { f16 h[2]; uint32_t u; } pk;
      pk.h[0] = su.h[t];
      pk.h[1] = su.h[t];
      sc[t] = pk.u;
    }
  };
  for (int s = s0; s < s1; ++s) {
    const int kb = s * 32 + 8 * g;  // this lane's 8 k-rows = packed row kb / 8
    const u32x4 qv = *reinterpret_cast<const u32x4*>(p.qweight + (int64_t)(kb / 8) * p.N + nl);
    if constexpr (!PER_ROW_GROUP) {
      const int grp = kb / p.G;  // lane-dependent only if G < 32 (not supported: G % 32 == 0 checked on the host)
      const int grp_u = (s * 32) / p.G;
      if (grp_u != cur_grp) {
        cur_grp = grp_u;
        load_group(grp_u);
      }
      (void)grp;
    }
    u32x4 af[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      af[mt] = load_a_frag<GATHER>(p, m0 + 16 * mt + li, kb);
      if constexpr (!SHUFFLED) {
        // natural nibble order: (q & 0x000f000f) pairs k with k + 4 -> reorder the activations the same way:
        // operand slots (0..7) <- k offsets (0,4,1,5,2,6,3,7)
        const u32x4 a = af[mt];
        af[mt][0] = __builtin_amdgcn_perm(a[2], a[0], 0x05040100u);  // (k0, k4)
        af[mt][1] = __builtin_amdgcn_perm(a[2], a[0], 0x07060302u);  // (k1, k5)
        af[mt][2] = __builtin_amdgcn_perm(a[3], a[1], 0x05040100u);  // (k2, k6)
        af[mt][3] = __builtin_amdgcn_perm(a[3], a[1], 0x07060302u);  // (k3, k7)
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t q = qv[t];
      u32x4 wf;
      if constexpr (!PER_ROW_GROUP) {
#pragma unroll
        for (int d = 0; d < 4; ++d) wf[d] = u32((h2(and_or(q >> (4 * d), 0x000f000fu, MAGIC)) - h2(zmag[t])) * h2(sc[t]));
      } else {
        // act-order without reordering (g_idx per k-row): every element has its own group
        union { u32x4 u; f16 h[8]; } w;
#pragma unroll
        for (int d = 0; d < 4; ++d) w.u[d] = and_or(q >> (4 * d), 0x000f000fu, MAGIC);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          // slot e of the operand: shuffled: k offset e; natural: offsets (0,4,1,5,2,6,3,7)
          const int koff = SHUFFLED ? e : ((e & 1) * 4 + (e >> 1));
          const int grp = p.g_idx[min(kb + koff, p.K - 1)];
          const int z = ((p.qzeros[(int64_t)grp * (p.N / 8) + (nl + t) / 8] >> (4 * ((nl + t) & 7))) & 0xf) + 1;
          const f16 sv = p.scales[(int64_t)grp * p.N + nl + t];
          w.h[e] = (f16)((float)(w.h[e] - (f16)(1024 + z))) * sv;
        }
        wf = w.u;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][t] = mfma_f16(wf, af[mt], acc[mt][t]);
    }
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  reduce_waves<MT, 4>(acc, smem, wave, lane);
  if (wave != 0) return;
  // D rows 4 g + r = n-slot -> columns n0 + 4 (4 g + r) + t
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * (4 * g + r);
      if (n >= p.N) continue;
      if (p.k_splits == 1) {
        union { f16 h[4]; u32x2 u; } o;
#pragma unroll
        for (int t = 0; t < 4; ++t) o.h[t] = (f16)acc[mt][t][r];
        *reinterpret_cast<u32x2*>(p.c + (int64_t)m * p.N + n) = o.u;
      } else {
        *reinterpret_cast<f32x4*>(p.partial + ((int64_t)blockIdx.y * p.M + m) * p.N + n) =
            f32x4{acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
      }
    }
  }
}

// ---- GPTQ 4-bit, plain groups (no act-order gather, group % 128 == 0, K % 128 == 0): the decode path ------------------
// Same arithmetic and operand order as gptq_gemm_kernel, but the loop above requests a k-step's operands and uses them
// at once - every iteration pays a full memory round trip (32 us for gate_up at M = 1). Here a wave keeps two 128-k
// units (4 weight loads + 4 MT activation loads + zero word + scale quad each) in flight: compiler-visible buffer loads
// in one fixed order (prologue = loop order, branch-free pair loop, an odd last unit dropped through a zero scale), the
// same scheme as marlin_decode_kernel. grid (N / 64, k_splits, ceil(M / (16 MT))), 4 waves = 4 K slices.
template <int MT, bool SHUFFLED, bool NT>
__global__ __launch_bounds__(256, 2) void gptq_gemm_ring_kernel(const ZpParams p) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int n0 = blockIdx.x * 64;
  const int nl = n0 + 4 * li;  // first of this lane's 4 columns
  const int m0 = blockIdx.z * 16 * MT;
  const int N = p.N, K = p.K;
  f32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int units = K / 128;
  const int workers = p.k_splits * 4;
  const int per = (units + workers - 1) / workers;
  const int u0 = min(((int)blockIdx.y * 4 + wave) * per, units), u1 = min(u0 + per, units);
  const int groups = K / p.G;

  const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.qweight), 0, (K / 8) * N * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.a), 0, p.M * K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.qzeros), 0, groups * (N / 8) * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.scales), 0, groups * N * 2, 0x00020000);
  const int q_voff = (g * N + nl) * 4;  // packed row 4 ks + g of the unit's 16, this lane's 4 columns
  int a_voff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + li;
    a_voff[mt] = m < p.M ? (m * K + 8 * g) * 2 : (int)0x7ff00000;  // beyond the descriptor: zeros
  }
  const int z_voff = (nl / 8) * 4, s_voff = nl * 2;

  struct Unit { u32x4 q[4]; u32x4 a[4][MT]; uint32_t z; u32x2 s; };
  auto load_step = [&](int u, int ks, Unit& U) {
    u = min(u, units - 1);  // past the slice: the last unit again (never used)
    // aux 2 = non-temporal when this launch has one row block (M <= 16 MT): the weights are then read exactly once
    U.q[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs_q, q_voff, (u * 16 + 4 * ks) * N * 4, NT ? 2 : 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) U.a[ks][mt] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff[mt], (u * 128 + 32 * ks) * 2, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto load_group = [&](int u, Unit& U) {
    u = min(u, units - 1);
    const int grp = (u * 128) / p.G;
    U.z = __builtin_amdgcn_raw_buffer_load_b32(rs_z, z_voff, grp * (N / 8) * 4, 0);
    U.s = __builtin_amdgcn_raw_buffer_load_b64(rs_s, s_voff, grp * N * 2, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto load_unit = [&](int u, Unit& U) {
    load_group(u, U);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) load_step(u, ks, U);
  };
  const uint32_t MAGIC = 0x64006400u;
  auto compute_unit = [&](Unit& U, int next, bool keep) {
    // qzeros: this lane's 4 columns share one word (4 li % 8 = 0 or 4); stored value is z - 1 (q_gemm.cu:1408)
    const uint32_t zw = U.z >> (4 * (nl & 7));
    union { u32x2 u; f16 h[4]; } su;
    su.u = U.s;
    load_group(next, U);
    uint32_t zmag[4], sc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t z = ((zw >> (4 * t)) & 0xf) + 1;
      zmag[t] = 0x64006400u + z * 0x00010001u;
      union { f16 h[2]; uint32_t u; } pk;
      pk.h[0] = su.h[t];
      pk.h[1] = su.h[t];
      sc[t] = keep ? pk.u : 0u;  // a dropped unit contributes (q - z) * 0
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 af[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        af[mt] = U.a[ks][mt];
        if constexpr (!SHUFFLED) {
          const u32x4 a = af[mt];
          af[mt][0] = __builtin_amdgcn_perm(a[2], a[0], 0x05040100u);  // (k0, k4)
          af[mt][1] = __builtin_amdgcn_perm(a[2], a[0], 0x07060302u);  // (k1, k5)
          af[mt][2] = __builtin_amdgcn_perm(a[3], a[1], 0x05040100u);  // (k2, k6)
          af[mt][3] = __builtin_amdgcn_perm(a[3], a[1], 0x07060302u);  // (k3, k7)
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t q = U.q[ks][t];
        u32x4 wf;
#pragma unroll
        for (int d = 0; d < 4; ++d) wf[d] = u32((h2(and_or(q >> (4 * d), 0x000f000fu, MAGIC)) - h2(zmag[t])) * h2(sc[t]));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][t] = mfma_f16(wf, af[mt], acc[mt][t]);
      }
      load_step(next, ks, U);
    }
  };
  if (u0 < u1) {
    Unit ua, ub;
    load_unit(u0, ua);
    load_unit(u0 + 1, ub);
    for (int u = u0; u < u1; u += 2) {
      compute_unit(ua, u + 2, true);
      compute_unit(ub, u + 3, u + 1 < u1);
    }
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  reduce_waves<MT, 4>(acc, smem, wave, lane);
  if (wave != 0) return;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * (4 * g + r);
      if (n >= N) continue;
      if (p.k_splits == 1) {
        union { f16 h[4]; u32x2 u; } o;
#pragma unroll
        for (int t = 0; t < 4; ++t) o.h[t] = (f16)acc[mt][t][r];
        *reinterpret_cast<u32x2*>(p.c + (int64_t)m * N + n) = o.u;
      } else {
        *reinterpret_cast<f32x4*>(p.partial + ((int64_t)blockIdx.y * p.M + m) * N + n) =
            f32x4{acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
      }
    }
  }
}

// ---- GPTQ 2 / 3 / 8-bit (gptq/q_gemm.cu:329-700 gemm kernels, :1386-1470 reconstruct) ------------------------------
// Legacy bit widths, kept simple: weights stay in the checkpoint's sequential packing (element k of a column is a
// BITS-wide field of a contiguous bit stream over k: 16 / 4 per word for 2 / 8 bit, 32 per 3 words for 3 bit; qzeros
// the same along n), gptq_shuffle only applies the act-order row permutation. Every lane extracts the 8 x 4 codes of
// its MFMA fragment with plain loads (the 4-bit kernel above is the tuned one).
template <int BITS>
__device__ __forceinline__ uint32_t gptq_field(const uint32_t* __restrict__ base, int64_t stride, int idx) {
  // field idx of the bit stream whose words are base[0], base[stride], base[2 stride], ...
  if constexpr (BITS == 3) {
    const int pos = 3 * (idx & 31), w = 3 * (idx >> 5) + (pos >> 5), sh = pos & 31;
    uint32_t v = base[(int64_t)w * stride] >> sh;
    if (sh > 29) v |= base[(int64_t)(w + 1) * stride] << (32 - sh);
    return v & 7u;
  } else {
    constexpr int PF = 32 / BITS;
    return (base[(int64_t)(idx / PF) * stride] >> (BITS * (idx % PF))) & ((1u << BITS) - 1u);
  }
}

template <int BITS, int MT, bool GATHER, bool PER_ROW_GROUP>
__global__ __launch_bounds__(256) void gptq_gemm_bits_kernel(const ZpParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int n0 = blockIdx.x * 64;
  const int nl = n0 + 4 * li;
  const int m0 = blockIdx.z * 16 * MT;
  f32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int steps = p.K / 32;
  const int workers = p.k_splits * 4;
  const int per = (steps + workers - 1) / workers;
  const int s0 = min((blockIdx.y * 4 + wave) * per, steps), s1 = min(s0 + per, steps);
  const int zwords = p.N * BITS / 32;  // words per qzeros row
  for (int s = s0; s < s1; ++s) {
    const int kb = s * 32 + 8 * g;
    u32x4 af[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = load_a_frag<GATHER>(p, m0 + 16 * mt + li, kb);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = nl + t;
      union { u32x4 u; f16 h[8]; } w;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = kb + e;
        const int grp = PER_ROW_GROUP ? p.g_idx[k] : k / p.G;
        const int q = (int)gptq_field<BITS>(p.qweight + n, p.N, k);
        const int z = (int)gptq_field<BITS>(p.qzeros + (int64_t)grp * zwords, 1, n) + 1;  // q_gemm.cu:1408 (zero + 1)
        w.h[e] = (f16)(float)(q - z) * p.scales[(int64_t)grp * p.N + n];
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][t] = mfma_f16(w.u, af[mt], acc[mt][t]);
    }
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  reduce_waves<MT, 4>(acc, smem, wave, lane);
  if (wave != 0) return;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * (4 * g + r);
      if (n >= p.N) continue;
      if (p.k_splits == 1) {
        union { f16 h[4]; u32x2 u; } o;
#pragma unroll
        for (int t = 0; t < 4; ++t) o.h[t] = (f16)acc[mt][t][r];
        *reinterpret_cast<u32x2*>(p.c + (int64_t)m * p.N + n) = o.u;
      } else {
        *reinterpret_cast<f32x4*>(p.partial + ((int64_t)blockIdx.y * p.M + m) * p.N + n) =
            f32x4{acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
      }
    }
  }
}

// make_sequential for any bit width (q_gemm.cu:1602-1760): new row k' takes old row q_perm[k']; one thread per
// (32-row group, column) re-packs BITS words
template <int BITS>
__global__ void gptq_make_sequential_bits_kernel(const uint32_t* __restrict__ w, uint32_t* __restrict__ w_new,
                                                 const int32_t* __restrict__ q_perm, int groups32, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int grp = blockIdx.y;
  if (n >= N || grp >= groups32) return;
  uint32_t out[BITS];
#pragma unroll
  for (int i = 0; i < BITS; ++i) out[i] = 0;
  for (int i = 0; i < 32; ++i) {
    const uint32_t v = gptq_field<BITS>(w + n, N, q_perm[grp * 32 + i]);
    const int pos = BITS * i;
    out[pos >> 5] |= v << (pos & 31);
    if ((pos & 31) + BITS > 32) out[(pos >> 5) + 1] |= v >> (32 - (pos & 31));
  }
#pragma unroll
  for (int i = 0; i < BITS; ++i) w_new[(int64_t)(grp * BITS + i) * N + n] = out[i];
}

__global__ void zp_reduce_kernel(f16* __restrict__ c, const float* __restrict__ partial, int64_t mn4, int splits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mn4) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};  // four slabs in flight, summation order s = 0, 1, 2, ...
  int s = 0;
  for (; s + 4 <= splits; s += 4) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(partial + ((int64_t)(s + 0) * mn4 + i) * 4);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(partial + ((int64_t)(s + 1) * mn4 + i) * 4);
    const f32x4 v2 = *reinterpret_cast<const f32x4*>(partial + ((int64_t)(s + 2) * mn4 + i) * 4);
    const f32x4 v3 = *reinterpret_cast<const f32x4*>(partial + ((int64_t)(s + 3) * mn4 + i) * 4);
    acc += v0;
    acc += v1;
    acc += v2;
    acc += v3;
  }
  for (; s < splits; ++s) acc += *reinterpret_cast<const f32x4*>(partial + ((int64_t)s * mn4 + i) * 4);
  union { f16 h[4]; u32x2 u; } r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r.h[j] = (f16)acc[j];
  *reinterpret_cast<u32x2*>(c + i * 4) = r.u;
}

// gptq_shuffle, 4-bit (gptq/q_gemm.cu:1543-1553, qdq_4.cuh:15-31): nibbles 0..7 -> low half (0,2,4,6), high half (1,3,5,7)
__global__ void gptq_shuffle4_kernel(uint32_t* __restrict__ w, int64_t n_words) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_words) return;
  uint32_t qa = w[i], qb = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t a0 = qa & 0x0f, a1 = (qa & 0xf0) >> 4;
    qa >>= 8;
    qb |= (a1 << (j * 4 + 16));
    qb |= (a0 << (j * 4));
  }
  w[i] = qb;
}
// make_sequential, 4-bit (q_gemm.cu:1602-1640): new row k' takes old row q_perm[k']
__global__ void gptq_make_sequential4_kernel(const uint32_t* __restrict__ w, uint32_t* __restrict__ w_new,
                                             const int32_t* __restrict__ q_perm, int rows8, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y;
  if (n >= N || row >= rows8) return;
  uint32_t dst = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int src = q_perm[row * 8 + i];
    const uint32_t v = (w[(int64_t)(src >> 3) * N + n] >> (4 * (src & 7))) & 0xf;
    dst |= v << (4 * i);
  }
  w_new[(int64_t)row * N + n] = dst;
}

int pick_splits(int n_tiles, int m_blocks, int steps) {
  int splits = (n_tiles * m_blocks >= 256) ? 1 : ceil_div(256, n_tiles * m_blocks);
  return std::max(1, std::min(splits, std::min(std::max(1, steps / 8), 16)));
}

}  // namespace

// fp32 partial slabs that awq_gemm / gptq_gemm can use for (m, n): the same pick_splits() the launchers run, on the
// coarser of their two tilings (AWQ: 128-column tiles) and an unbounded K - an upper bound of the splits either takes,
// and 0 as soon as the tiles alone fill the chip (prefill: no scratch at all)
extern "C" int64_t nmx_zp_gemm_scratch_bytes(int m, int n) {
  if (m <= 0 || n <= 0) return 0;
  const int mt = m <= 16 ? 1 : 2;
  const int splits = pick_splits(ceil_div(n, 128), ceil_div(m, 16 * mt), 1 << 20);
  return splits > 1 ? (int64_t)splits * m * n * sizeof(float) : 0;
}

extern "C" int nmx_awq_gemm(const void* in_feats, const int32_t* kernel, const void* scaling_factors,
                            const int32_t* zeros, void* out, void* scratch, int64_t scratch_bytes, int m, int k, int oc,
                            int group_size, nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  // awq/gemm_kernels.cu:515-522 (std::invalid_argument in the reference)
  NMX_CHECK(oc % 64 == 0, NMX_ERR_INVALID_ARG, "OC is not multiple of cta_N = 64");
  NMX_CHECK(oc % 8 == 0, NMX_ERR_INVALID_ARG, "OC is not multiple of pack_num = 8");
  NMX_CHECK(group_size > 0 && group_size % 32 == 0, NMX_ERR_INVALID_ARG, "Group size should be a multiple of 32");
  NMX_CHECK(oc % group_size == 0, NMX_ERR_INVALID_ARG, "OC is not multiple of Group size");
  NMX_CHECK(k % 32 == 0, NMX_ERR_INVALID_ARG, "IC = %d must be a multiple of 32", k);
  if (m == 0) return NMX_OK;
  ZpParams p{(const f16*)in_feats, (const uint32_t*)kernel, (const uint32_t*)zeros, (const f16*)scaling_factors,
             nullptr, nullptr, (f16*)out, (float*)scratch, m, oc, k, group_size, 1, 0};
  const int mt = m <= 16 ? 1 : 2;
  const int n_tiles = ceil_div(oc, 128), m_blocks = ceil_div(m, 16 * mt);
  p.k_splits = pick_splits(n_tiles, m_blocks, k / 32);
  if (p.k_splits > 1 && (scratch == nullptr || scratch_bytes < (int64_t)p.k_splits * m * oc * 4))
    p.k_splits = scratch ? std::max<int>(1, (int)(scratch_bytes / ((int64_t)m * oc * 4))) : 1;
  dim3 grid(n_tiles, p.k_splits, m_blocks);
  const bool ring = group_size % 128 == 0 && k % 128 == 0 && (int64_t)k * oc / 2 < (1ll << 31) && (int64_t)m * k * 2 < (1ll << 31) &&
                    nmx_tune(NMX_TUNE_AWQ_NO_RING) == nullptr;
  if (ring) {
    while (p.k_splits > 1 && (k / 128) / (p.k_splits * 4) < 1) p.k_splits /= 2;  // >= 1 unit per wave (tiny N: CUs first)
    grid.y = p.k_splits;
    if (mt == 1) awq_gemm_ring_kernel<1><<<grid, 256, 3 * 1 * 8 * 64 * 16, stream>>>(p);
    else awq_gemm_ring_kernel<2><<<grid, 256, 3 * 2 * 8 * 64 * 16, stream>>>(p);
  } else if (mt == 1) awq_gemm_kernel<1><<<grid, 256, 3 * 1 * 8 * 64 * 16, stream>>>(p);
  else awq_gemm_kernel<2><<<grid, 256, 3 * 2 * 8 * 64 * 16, stream>>>(p);
  NMX_LAUNCH_CHECK();
  if (p.k_splits > 1) {
    const int64_t mn4 = (int64_t)m * oc / 4;
    zp_reduce_kernel<<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>((f16*)out, p.partial, mn4, p.k_splits);
    NMX_LAUNCH_CHECK();
  }
  return NMX_OK;
}

extern "C" int nmx_awq_dequantize(const int32_t* kernel, const void* scaling_factors, const int32_t* zeros, void* out,
                                  int in_c, int qout_c, int group_size, nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NMX_CHECK(group_size > 0 && in_c % group_size == 0, NMX_ERR_INVALID_ARG, "awq_dequantize: bad group size %d", group_size);
  const int64_t total = (int64_t)in_c * qout_c;
  if (total == 0) return NMX_OK;
  awq_dequantize_kernel<<<(unsigned)ceil_div64(total, 256), 256, 0, stream>>>((const uint32_t*)kernel, (const f16*)scaling_factors,
                                                                           (const uint32_t*)zeros, (f16*)out, in_c, qout_c, group_size);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

extern "C" int nmx_gptq_gemm(const void* a, const int32_t* qweight, const int32_t* qzeros, const void* scales,
                             const int32_t* g_idx, void* c, void* scratch, int64_t scratch_bytes, int m, int n, int k,
                             int groups, int use_exllama, int bit, nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NMX_CHECK(bit == 2 || bit == 3 || bit == 4 || bit == 8, NMX_ERR_INVALID_ARG, "gptq_gemm: bit must be 2, 3, 4 or 8 (got %d)", bit);
  NMX_CHECK(groups > 0 && k % groups == 0, NMX_ERR_INVALID_ARG, "gptq_gemm: K = %d not divisible by groups = %d", k, groups);
  const int G = k / groups;
  NMX_CHECK(k % 32 == 0 && n % 64 == 0, NMX_ERR_INVALID_ARG, "gptq_gemm: K %% 32 == 0 and N %% 64 == 0 required (K=%d N=%d)", k, n);
  if (m == 0) return NMX_OK;
  ZpParams p{(const f16*)a, (const uint32_t*)qweight, (const uint32_t*)qzeros, (const f16*)scales, nullptr, nullptr,
             (f16*)c, (float*)scratch, m, n, k, G, 1, use_exllama};
  // exllama (shuffled): g_idx argument is the row permutation q_perm (gptq.py:212-222) -> gather A columns, groups are
  // contiguous in the sorted order. Non-exllama: g_idx is the per-row group index in checkpoint order.
  const bool gather = use_exllama && g_idx != nullptr;
  const bool per_row = (!use_exllama && g_idx != nullptr) || (G % 32 != 0);
  if (gather) p.q_perm = g_idx;
  if (per_row) p.g_idx = g_idx;
  NMX_CHECK(!(per_row && g_idx == nullptr), NMX_ERR_UNSUPPORTED, "gptq_gemm: group size %d needs g_idx", G);
  const int mt = m <= 16 ? 1 : 2;
  const int n_tiles = n / 64, m_blocks = ceil_div(m, 16 * mt);
  p.k_splits = pick_splits(n_tiles, m_blocks, k / 32);
  if (p.k_splits > 1 && (scratch == nullptr || scratch_bytes < (int64_t)p.k_splits * m * n * 4))
    p.k_splits = scratch ? std::max<int>(1, (int)(scratch_bytes / ((int64_t)m * n * 4))) : 1;
  dim3 grid(n_tiles, p.k_splits, m_blocks);
  const size_t smem = (size_t)3 * mt * 4 * 64 * 16;
  if (bit != 4) {
    // 2 / 3 / 8 bit: sequential packing for both the exllama and the plain entry (gptq_shuffle only re-orders rows)
#define NMX_GPTQ_B(B_, MT_, GA, PR) gptq_gemm_bits_kernel<B_, MT_, GA, PR><<<grid, 256, smem, stream>>>(p)
#define NMX_GPTQ_BITS(B_)                                                                         \
  do {                                                                                            \
    if (mt == 1) { if (gather) NMX_GPTQ_B(B_, 1, true, false); else if (per_row) NMX_GPTQ_B(B_, 1, false, true); else NMX_GPTQ_B(B_, 1, false, false); } \
    else { if (gather) NMX_GPTQ_B(B_, 2, true, false); else if (per_row) NMX_GPTQ_B(B_, 2, false, true); else NMX_GPTQ_B(B_, 2, false, false); }         \
  } while (0)
    if (bit == 2) NMX_GPTQ_BITS(2);
    else if (bit == 3) NMX_GPTQ_BITS(3);
    else NMX_GPTQ_BITS(8);
#undef NMX_GPTQ_BITS
#undef NMX_GPTQ_B
    NMX_LAUNCH_CHECK();
    if (p.k_splits > 1) {
      const int64_t mn4 = (int64_t)m * n / 4;
      zp_reduce_kernel<<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>((f16*)c, p.partial, mn4, p.k_splits);
      NMX_LAUNCH_CHECK();
    }
    return NMX_OK;
  }
  if (!gather && !per_row && G % 128 == 0 && k % 128 == 0 && (int64_t)k * n / 2 < (1ll << 31) && (int64_t)m * k * 2 < (1ll << 31) &&
      nmx_tune(NMX_TUNE_GPTQ_NO_RING) == nullptr) {
    // keep >= 2 units (256 k) per wave so that the ring has something to overlap
    while (p.k_splits > 1 && (k / 128) / (p.k_splits * 4) < 2) p.k_splits /= 2;
    grid.y = p.k_splits;
    const bool nt = m_blocks == 1 && nmx_tune(NMX_TUNE_GPTQ_NT) != nullptr;  // measured: no gain (2.70 vs 2.65 ms at batch 1): off
#define NMX_RING(MT_, SH) do { if (nt) gptq_gemm_ring_kernel<MT_, SH, true><<<grid, 256, smem, stream>>>(p); else gptq_gemm_ring_kernel<MT_, SH, false><<<grid, 256, smem, stream>>>(p); } while (0)
    if (use_exllama) { if (mt == 1) NMX_RING(1, true); else NMX_RING(2, true); }
    else { if (mt == 1) NMX_RING(1, false); else NMX_RING(2, false); }
#undef NMX_RING
    NMX_LAUNCH_CHECK();
    if (p.k_splits > 1) {
      const int64_t mn4 = (int64_t)m * n / 4;
      zp_reduce_kernel<<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>((f16*)c, p.partial, mn4, p.k_splits);
      NMX_LAUNCH_CHECK();
    }
    return NMX_OK;
  }
#define NMX_GPTQ(MT_, SH, GA, PR) gptq_gemm_kernel<MT_, SH, GA, PR><<<grid, 256, smem, stream>>>(p)
#define NMX_GPTQ_MT(SH, GA, PR) do { if (mt == 1) NMX_GPTQ(1, SH, GA, PR); else NMX_GPTQ(2, SH, GA, PR); } while (0)
  if (use_exllama) {
    if (gather) NMX_GPTQ_MT(true, true, false);
    else if (per_row) NMX_GPTQ_MT(true, false, true);
    else NMX_GPTQ_MT(true, false, false);
  } else {
    if (per_row) NMX_GPTQ_MT(false, false, true);
    else NMX_GPTQ_MT(false, false, false);
  }
#undef NMX_GPTQ_MT
#undef NMX_GPTQ
  NMX_LAUNCH_CHECK();
  if (p.k_splits > 1) {
    const int64_t mn4 = (int64_t)m * n / 4;
    zp_reduce_kernel<<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>((f16*)c, p.partial, mn4, p.k_splits);
    NMX_LAUNCH_CHECK();
  }
  return NMX_OK;
}

extern "C" int nmx_gptq_shuffle(int32_t* q_weight, int32_t* tmp, const int32_t* q_perm, int size_k, int size_n, int bit,
                                nmx_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NMX_CHECK(bit == 2 || bit == 3 || bit == 4 || bit == 8, NMX_ERR_INVALID_ARG, "gptq_shuffle: bit must be 2, 3, 4 or 8 (got %d)", bit);
  if (bit != 4) {
    NMX_CHECK(size_k % 32 == 0, NMX_ERR_INVALID_ARG, "gptq_shuffle: K = %d must be a multiple of 32", size_k);
    if (q_perm == nullptr || size_k == 0 || size_n == 0) return NMX_OK;  // sequential packing is the kernel's format
    NMX_CHECK(tmp != nullptr, NMX_ERR_INVALID_ARG, "gptq_shuffle: act-order needs a temporary buffer");
    const int groups32 = size_k / 32;
    dim3 grid(ceil_div(size_n, 128), groups32);
    if (bit == 2) gptq_make_sequential_bits_kernel<2><<<grid, 128, 0, stream>>>((const uint32_t*)q_weight, (uint32_t*)tmp, q_perm, groups32, size_n);
    else if (bit == 3) gptq_make_sequential_bits_kernel<3><<<grid, 128, 0, stream>>>((const uint32_t*)q_weight, (uint32_t*)tmp, q_perm, groups32, size_n);
    else gptq_make_sequential_bits_kernel<8><<<grid, 128, 0, stream>>>((const uint32_t*)q_weight, (uint32_t*)tmp, q_perm, groups32, size_n);
    NMX_LAUNCH_CHECK();
    NMX_HIP(hipMemcpyAsync(q_weight, tmp, (int64_t)groups32 * bit * size_n * 4, hipMemcpyDeviceToDevice, stream));
    return NMX_OK;
  }
  NMX_CHECK(size_k % 8 == 0, NMX_ERR_INVALID_ARG, "gptq_shuffle: K = %d must be a multiple of 8", size_k);
  const int rows8 = size_k / 8;
  const int64_t words = (int64_t)rows8 * size_n;
  if (words == 0) return NMX_OK;
  if (q_perm != nullptr) {
    NMX_CHECK(tmp != nullptr, NMX_ERR_INVALID_ARG, "gptq_shuffle: act-order needs a temporary buffer");
    dim3 grid(ceil_div(size_n, 128), rows8);
    gptq_make_sequential4_kernel<<<grid, 128, 0, stream>>>((const uint32_t*)q_weight, (uint32_t*)tmp, q_perm, rows8, size_n);
    NMX_LAUNCH_CHECK();
    NMX_HIP(hipMemcpyAsync(q_weight, tmp, words * 4, hipMemcpyDeviceToDevice, stream));
  }
  gptq_shuffle4_kernel<<<(unsigned)ceil_div64(words, 256), 256, 0, stream>>>((uint32_t*)q_weight, words);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}
