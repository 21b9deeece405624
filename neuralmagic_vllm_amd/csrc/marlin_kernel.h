#pragma once
// Marlin-format W4A16 / W8A16 / fp8-W8A16 GEMMs (dense and 2:4-sparse) and the GPTQ->Marlin repack for gfx950.
//
// Replaces csrc/quantization/gptq_marlin/{gptq_marlin.cu, gptq_marlin_repack.cu},
// csrc/quantization/marlin/dense/marlin_cuda_kernel.cu and csrc/quantization/fp8/fp8_marlin.cu of the reference.
//
// The op contract hands over weights in the *Marlin layout* (built for NVIDIA mma.m16n8k16 fragments). It turns out
// to be directly consumable by MFMA 16x16x32 with zero re-layout:
//   * one lane's 16-byte load of a Marlin row (k-tile kt, 64-column group, chunk i = 4 c + m) holds, for the 8
//     columns {c, c+8} + 16 j (j = 0..3), the 4 k-rows {2m, 2m+1, 2m+8, 2m+9} of the 16-row k-tile;
//   * taking k-tiles 2 ks and 2 ks + 1 gives the lane 8 k-values for each of 8 columns = eight MFMA operand
//     fragments, provided the activation operand uses the same k order inside each 32-k step
//     (slot (g, jj) <-> k = 16 (jj >> 2) + 2 g + {0, 1, 8, 9}[jj & 3]);
//   * a wave's 64 lanes (c = lane & 7, m = lane >> 4, column-group = (lane >> 3) & 1) cover 1 KiB of contiguous
//     HBM per load instruction: fully coalesced, every fetched bit is used exactly once.
// So the weight stream goes HBM -> VGPR -> dequant (2 VALU per packed pair) -> MFMA with no LDS round trip and no
// repack pass. Activations (tiny, L2-resident) are staged per wave into LDS in fragment order.
//
// This file holds the "skinny" kernel (M <= 64 rows per pass, HBM-bound regime of decode). Rows are processed in
// blocks of 16 * MT; K is split over the 4 waves of a workgroup (LDS tree reduce) and over gridDim.y workgroups
// (fp32 partial slabs + a small reduce kernel).
//
// Algorithmic bytes per call: K*N*bits/8 (weights) + groups*N*2 (scales) + 2*M*K + 2*M*N.
#include <stdlib.h>
#include <type_traits>

#include "nmx_common.h"
#include "marlin_wide_api.h"

namespace {

// Timing ablations for tools/ablate_gemm.sh (results are WRONG when set; never defined in the product build):
// bit 0 skip MFMAs, 1 skip dequant, 2 skip LDS fragment reads, 3 skip workgroup barriers, 4 skip weight waits,
// bit 5 skip the epilogue stores, 6 skip activation loads, 7 skip weight loads, 8 skip LDS fragment writes
#ifndef NMX_ABLATE
#define NMX_ABLATE 0
#endif

// Non-temporal hint on the weight loads of marlin_decode_kernel (every byte read once: +4-7 %; in marlin_gemm_kernel the
// same hint measured 3-10 % SLOWER, with one row block too, and is not used there); 0 = plain loads
#ifndef NMX_W_NT
#define NMX_W_NT 1
#endif

#ifndef NMX_SKINNY_NT
#define NMX_SKINNY_NT 2  // marlin_gemm_kernel tiles that only run with one row block (16-row tiles; the split-free 128-column 8-wave tile): non-temporal weight loads, -1.5..-4 %
#endif

constexpr int kSubSteps = 4;  // 32-k steps per activation staging sub-chunk (128 k)

enum WeightKind { W_INT4 = 0, W_INT8 = 1, W_FP8 = 2 };

template <typename scalar_t>
__device__ __forceinline__ f32x4 mfma_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (__is_same(scalar_t, f16)) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
}

// (q & mask) | magic in ONE VALU op. hipcc splits the C expression into v_and + v_or because a VOP3 on gfx950 cannot
// carry two literals; with the mask in an SGPR and the magic number in a VGPR it is a single v_and_or_b32.
__device__ __forceinline__ uint32_t and_or(uint32_t q, uint32_t mask, uint32_t magic) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(q), "s"(mask), "v"(magic));
  return r;
}

// Every vector-memory load of this file is a compiler-visible __builtin_amdgcn_raw_buffer_load_* (per-lane byte offset in a
// VGPR, the wave-uniform running offset in an SGPR, loads past the descriptor's range return 0): hipcc counts the vmcnt
// waits itself and never reads a destination early. (Rounds 1-2 issued the ring of marlin_gemm_kernel as inline-asm
// loads with "+v" tied destinations and hand-counted waits; under register pressure the allocator can re-home such an
// operand with a copy made BEFORE the load has landed. Removed in round 3.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ uint32_t h2_bits(f16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ f16x2 bits_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  union { bf16 h[2]; uint32_t u; } r;
  r.h[0] = (bf16)lo;
  r.h[1] = (bf16)hi;
  return r.u;
}

// ---- dequantisation of one packed dword into two operand dwords --------------------------------------------
// int4, fp16: q holds (after the optional >> 8 for the "+8 column" half) nibbles n0 n1 . . n4 n5 . .
//   (n0, n4) = k-rows (2m, 2m+1), (n1, n5) = k-rows (2m+8, 2m+9).
// Exact integer -> fp16 conversion with the 0x6400 exponent trick (same constants the reference relies on,
// gptq_marlin.cu:162-181): (q & 0x000f000f) | 0x64006400 = 1024 + v ; (q & 0x00f000f0) | 0x64006400 = 1024 + 16 v.
template <typename scalar_t, int KIND>
struct Dequant;

template <>
struct Dequant<f16, W_INT4> {
  // s2 = (s, s) packed fp16 scale for this column, or 1.0
  static __device__ __forceinline__ void run(uint32_t q, uint32_t s2, bool scaled, uint32_t& w01, uint32_t& w23) {
    const f16x2 a = bits_h2(and_or(q, 0x000f000fu, 0x64006400u)) - bits_h2(0x64086408u);                       // v - 8
    const f16x2 b = bits_h2(and_or(q, 0x00f000f0u, 0x64006400u)) * bits_h2(0x2c002c00u) + bits_h2(0xd480d480u);  // /16 - 72
    if (scaled) {
      w01 = h2_bits(a * bits_h2(s2));
      w23 = h2_bits(b * bits_h2(s2));
    } else {
      w01 = h2_bits(a);
      w23 = h2_bits(b);
    }
  }
};

// int4 with a per-(group, column) zero point (AWQ): zneg = (-(1024 + z), -(1024 + z)), zhi = (-(64 + z), -(64 + z)).
// (1024 + q) - (1024 + z) and (1024 + 16 q) / 16 - (64 + z) are exact: the same two operations per pair as the symmetric form.
__device__ __forceinline__ void dequant_zp_f16(uint32_t q, uint32_t s2, bool scaled, uint32_t zneg, uint32_t zhi, uint32_t& w01, uint32_t& w23) {
  const f16x2 a = bits_h2(and_or(q, 0x000f000fu, 0x64006400u)) + bits_h2(zneg);
  const f16x2 b = bits_h2(and_or(q, 0x00f000f0u, 0x64006400u)) * bits_h2(0x2c002c00u) + bits_h2(zhi);
  if (scaled) {
    w01 = h2_bits(a * bits_h2(s2));
    w23 = h2_bits(b * bits_h2(s2));
  } else {
    w01 = h2_bits(a);
    w23 = h2_bits(b);
  }
}

template <>
struct Dequant<f16, W_INT8> {
  // bytes b0 b1 b2 b3 = v0 v2 v1 v3 (k-rows 2m, 2m+8, 2m+1, 2m+9); zero point 128
  static __device__ __forceinline__ void run(uint32_t q, uint32_t s2, bool scaled, uint32_t& w01, uint32_t& w23) {
    const f16x2 a = bits_h2(and_or(q, 0x00ff00ffu, 0x64006400u)) - bits_h2(0x64806480u);         // 1024 + b - 1152
    const f16x2 b = bits_h2(and_or(q >> 8, 0x00ff00ffu, 0x64006400u)) - bits_h2(0x64806480u);
    if (scaled) {
      w01 = h2_bits(a * bits_h2(s2));
      w23 = h2_bits(b * bits_h2(s2));
    } else {
      w01 = h2_bits(a);
      w23 = h2_bits(b);
    }
  }
};

template <>
struct Dequant<f16, W_FP8> {
  // e4m3fn byte -> fp16: move sign, shift exponent/mantissa into place, fix the bias with * 2^8
  // (same construction as fp8/fp8_marlin.cu:132-196)
  static __device__ __forceinline__ uint32_t cvt(uint32_t t) {  // t = 0x00XX00YY
    const uint32_t r = ((t << 8) & 0x80008000u) | ((t << 7) & 0x3f803f80u);
    return h2_bits(bits_h2(r) * bits_h2(0x5c005c00u));  // * 256
  }
  static __device__ __forceinline__ void run(uint32_t q, uint32_t s2, bool scaled, uint32_t& w01, uint32_t& w23) {
    w01 = cvt(q & 0x00ff00ffu);
    w23 = cvt((q >> 8) & 0x00ff00ffu);
    if (scaled) {
      w01 = h2_bits(bits_h2(w01) * bits_h2(s2));
      w23 = h2_bits(bits_h2(w23) * bits_h2(s2));
    }
  }
};

// bf16 has no packed arithmetic on gfx950: go through fp32 (exact integer, one rounding at the end).
// s2 carries the fp32 scale bits for bf16.
template <>
struct Dequant<bf16, W_INT4> {
  static __device__ __forceinline__ void run(uint32_t q, uint32_t sbits, bool scaled, uint32_t& w01, uint32_t& w23) {
    const float s = scaled ? __builtin_bit_cast(float, sbits) : 1.0f;
    const float v0 = (float)(int)(q & 0xf) - 8.f, v1 = (float)(int)((q >> 16) & 0xf) - 8.f;
    const float v2 = (float)(int)((q >> 4) & 0xf) - 8.f, v3 = (float)(int)((q >> 20) & 0xf) - 8.f;
    w01 = pack_bf16(v0 * s, v1 * s);
    w23 = pack_bf16(v2 * s, v3 * s);
  }
};
template <>
struct Dequant<bf16, W_INT8> {
  static __device__ __forceinline__ void run(uint32_t q, uint32_t sbits, bool scaled, uint32_t& w01, uint32_t& w23) {
    const float s = scaled ? __builtin_bit_cast(float, sbits) : 1.0f;
    const float v0 = (float)(int)(q & 0xff) - 128.f, v1 = (float)(int)((q >> 16) & 0xff) - 128.f;
    const float v2 = (float)(int)((q >> 8) & 0xff) - 128.f, v3 = (float)(int)((q >> 24) & 0xff) - 128.f;
    w01 = pack_bf16(v0 * s, v1 * s);
    w23 = pack_bf16(v2 * s, v3 * s);
  }
};
template <>
struct Dequant<bf16, W_FP8> {
  static __device__ __forceinline__ void run(uint32_t q, uint32_t sbits, bool scaled, uint32_t& w01, uint32_t& w23) {
    const float s = scaled ? __builtin_bit_cast(float, sbits) : 1.0f;
    const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(q, false);  // bytes 0,1 = v0, v2
    const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8(q, true);   // bytes 2,3 = v1, v3
    w01 = pack_bf16(lo[0] * s, hi[0] * s);
    w23 = pack_bf16(lo[1] * s, hi[1] * s);
  }
};

// Lane (g, hi, c8) loaded the four words (sub-tiles j = 0..3) of chunk 4 c8 + g of k-tile 2 ks + hi; it needs words
// 2 hi, 2 hi + 1 of BOTH k-tiles. Its partner lane ^ 8 (same g and c8, other hi) holds exactly the missing pair and
// needs the pair this lane does not: one DPP row rotation by 8 per word.
__device__ __forceinline__ void split_pair(u32x4 r, bool hi, u32x2& q0, u32x2& q1) {
  const uint32_t s0 = hi ? r[0] : r[2], s1 = hi ? r[1] : r[3];  // what the partner wants
  const uint32_t t0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0x128, 0xf, 0xf, false);  // row_ror:8
  const uint32_t t1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0x128, 0xf, 0xf, false);
  q0 = hi ? u32x2{t0, t1} : u32x2{r[0], r[1]};
  q1 = hi ? u32x2{r[2], r[3]} : u32x2{t0, t1};
}

struct GemmParams {
  const void* a;           // [M, K]
  const int32_t* b;        // Marlin-packed weight (2:4: compressed non-zeros, Marlin-24 permutation)
  const void* meta;        // 2:4 only: [K/32, 2N] int16 CUTLASS-reordered 2-bit indices
  const void* scales;      // [num_groups, N] Marlin-permuted
  int partial_f16 = 0;     // marlin_wide_kernel, fp16 outputs: the K-split slabs hold fp16 (NMX_SPLITK_F16)
  const void* zeros;       // ZP kernels only: [num_groups, N] fp16 -(1024 + z), permuted like the grouped scales
  const int32_t* g_idx;    // [K] or null
  const int32_t* perm;     // [K] or null
  void* c;                 // [M, N] scalar_t
  float* partial;          // [k_splits, M, N] fp32 (k_splits > 1)
  int M, N, K;
  int num_groups, group_size;  // group_size = K for channel-wise
  int k_splits;
  int slow_act_order;      // act-order with partial K: per-row scale lookup
  int defer_reduce;        // k_splits > 1: leave the fp32 slabs in `partial` for the consumer (no reduce launch)
  void* act_out = nullptr; // gate_up + silu_and_mul in one launch: [M, N / 2]. A launch without a K split of marlin_wide,
                           // of marlin_gemm_kernel's 16-row shape or of marlin_decode_kernel writes silu(c[:, :N/2]) *
                           // c[:, N/2:] here INSTEAD of c and the host sets act_done; every other path leaves it alone
  int act_done = 0;        // host side only
  int xcd_split = 1;       // marlin_gemm_kernel: 2 / 4 / 8 K splits are placed one per group of XCDs (NMX_GEMM_XCD_SPLIT=0: off)
  // marlin_decode_kernel<NORM> (round 3, late): the A operand is fused_add_rms_norm of a deferred GEMM's K-split slabs,
  // computed in every workgroup's prologue (a is unused): x = round(sum_s norm_partial[s]) + norm_res_in; A = norm(x) * norm_weight;
  // workgroup (0, 0) also writes x to norm_res_out (a buffer of its own: other workgroups still read norm_res_in)
  const float* norm_partial = nullptr;  // [norm_splits][M][K] fp32, or fp16 with NMX_SPLITK_F16 set in norm_splits
  int norm_splits = 0;
  const void* norm_res_in = nullptr;    // [M, K] scalar_t
  void* norm_res_out = nullptr;         // [M, K] scalar_t
  const void* norm_weight = nullptr;    // [K] scalar_t
  float norm_eps = 0.f;
  // marlin_decode_kernel<ATTN> (round 3, late): the A operand is the v2 reduce of paged attention's partition results - row m =
  // sequence m, k = head * 128 + d - computed by every wave for the heads of its own K slice (a is unused)
  const float* attn_exp_sums = nullptr;   // [M, heads, max_parts]
  const float* attn_max_logits = nullptr; // [M, heads, max_parts]
  const void* attn_tmp = nullptr;         // [M, heads, max_parts, 128] scalar_t
  const int32_t* attn_seq_lens = nullptr; // [M]
  int attn_part_size = 0, attn_max_parts = 0, attn_heads = 0;
};
constexpr int kNormMaxRows = 4;         // rows the norm-fused kernel can take (every workgroup recomputes the norm of all rows)
// threads / vectors per thread of rms_norm_splitk_kernel for a hidden size (elementwise.hip add_rms_norm_splitk_common): the
// norm-fused GEMM prologue sums the squares over the same threads in the same order, so the two forms agree bit for bit
inline __host__ __device__ int norm_threads(int hidden) {
  const int nvec = hidden / 8;
  int t = ((nvec + 63) / 64) * 64;
  if (nvec > 256) t = ((nvec / 2 + 63) / 64) * 64;
  return t < 1024 ? t : 1024;
}

// ---- the GEMM kernel -------------------------------------------------------------------------------------------
// Workgroup = 4 waves. Wave w owns 64-column group (w % NG) of the workgroup's 64*NG columns and K-slice (w / NG) of
// the workgroup's K range (KW = 4 / NG slices, reduced through LDS at the end). Rows: 16*MT per workgroup.
//   M <= 16 : MT = 1, NG = 1  (each wave streams its own K-slice of one 64-column group; no barriers)
//   M <= 32 : MT = 2, NG = 2
//   larger  : MT = 4, NG = 4  (256-column tiles: the activation tile is shared by the 4 waves through LDS so that
//                              activation re-reads from L2 stay ~1x the weight bytes)
// Weight stream: lane (g, li), c8 = li & 7, hi = li >> 3 reads from chunk 4 c8 + g of its 64-column group the two
// words of tiles j = 2 hi, 2 hi + 1 (8 B, int4; 16 B for 8-bit) of BOTH k-tiles of a 32-k step: the 64 lanes cover
// the group's 512 B (1 KiB) per k-tile exactly once, and every lane owns all four MFMA operand fragments it needs
// (column c8 + 8 x + 32 hi for x = 0..3) without any cross-lane traffic. A register ring keeps PF k-steps in flight.
// Activations are double-buffered in LDS in fragment order.
// grid (ceil(N / (64 NG)), k_splits, ceil(M / (16 MT))), block 256
// MODE 0: channel-wise scales, MODE 1: group size a multiple of 128 (one scale row per sub-chunk) — both
// branch-free in the main loop so that hipcc keeps counted vmcnt waits and the weight ring stays in flight;
// MODE 2: generic (group sizes 32 / 64, act-order column gather, per-row group lookup).
//
// SP = true: 2:4-sparse weights (gptq_marlin_24_gemm). One packed k-tile row then spans 32 dense k; the lane's word
// pair holds, for each of its four 16-column tiles, the two kept values of quads g and g + 4 of the k-tile, and the
// hardware sparse MFMA (v_smfmac_f32_16x16x32_f16) consumes them directly together with the 2-bit positions from
// the metadata tensor: the compressed operand is never expanded. Second ring slot = the lane's 16 B of metadata.
// Second launch bound = minimum waves per SIMD: without it hipcc spreads the MT = 4 accumulators over VGPRs + AGPRs
// (> 256 registers) and a CU holds ONE workgroup, so every stall of a wave is exposed and a grid just above 256
// workgroups runs as two rounds.
// W8 = true: 8 waves per workgroup (twice the K slices for the same column groups): two waves per SIMD are then
// resident BY CONSTRUCTION and fill each other's stalls, without doubling the cross-workgroup K splits (and their
// fp32 partial traffic) that two co-resident 4-wave workgroups would need.
// ZP = true (fp16, int4, MODE 1): per-(group, column) zero points from p.zeros (AWQ weights repacked into the Marlin layout).
template <typename scalar_t, int KIND, int MT, int NG, int MODE, bool SP = false, bool W8 = false, bool ZP = false>
__global__ __launch_bounds__(W8 ? 512 : 256, 2) void marlin_gemm_kernel(const GemmParams p) {
  static_assert(!SP || (__is_same(scalar_t, f16) && KIND != W_FP8), "2:4 path: fp16, int4 / int8 weights");
  static_assert(!ZP || (__is_same(scalar_t, f16) && KIND == W_INT4 && MODE == 1 && !SP), "zero points: fp16, int4, 128-multiple groups");
  constexpr bool I4 = (KIND == W_INT4);
  constexpr bool GENERIC = (MODE == 2);
  constexpr int SUB = I4 ? 4 : 2;                     // 32-k steps per sub-chunk
  constexpr int PF = 2 * SUB;                         // k-steps of weights in flight per wave
  constexpr int NTILE = 4;                            // 16-column MFMA tiles per wave (one 64-column group)
  constexpr int KW = (W8 ? 8 : 4) / NG;               // K slices per workgroup
  constexpr int ROWS = 16 * MT;
  constexpr int WORDS64 = I4 ? 128 : 256;             // int32 per (k-tile, 64-column group)
  constexpr int ABUF = SUB * 4 * ROWS * 16;           // bytes of one activation buffer
  constexpr int NPIECE = ROWS * SUB * 4;              // 16-B activation pieces per sub-chunk
  constexpr int PIECES = (NPIECE + 64 * NG - 1) / (64 * NG);  // per thread

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: K-slice bounds, loop trip counts and load offsets stay in SGPRs
  const int g = lane >> 4;
  const int li = lane & 15;
  const int c8 = li & 7;
  const int hi = li >> 3;
  const int ng = wave % NG;
  const int kslice = wave / NG;

  const int N = p.N, K = p.K, M = p.M;
  // blockIdx.x enumerates (column tile, row block) so that the row blocks of one column tile are 8 workgroup ids
  // apart: consecutive ids go to consecutive XCDs, so they land on the SAME XCD a few dispatches apart and the second
  // row block finds the tile's weights in that XCD's L2 instead of fetching them from HBM again.
  const int m_blocks = (p.M + ROWS - 1) / ROWS;
  // With 2 / 4 / 8 K splits the XCD (= linear workgroup id % 8; gridDim.x is a multiple of 8) selects the K SPLIT as well:
  // 8 / splits XCDs share one split, each of them a disjoint set of column tiles. A split's slice of the ACTIVATIONS is then
  // fetched by 8 / splits XCDs instead of all 8 (down_proj at M = 256: 7.3 MB of activations were fetched 8 x = 59 of the
  // launch's 101 MB), the weights of a (tile, split) stay with one XCD as before.
  int tile_x, block_m, split_id;
  if (p.xcd_split && (p.k_splits == 2 || p.k_splits == 4 || p.k_splits == 8)) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int per = 8 / p.k_splits;  // XCDs per split
    const int q = lin >> 3;
    split_id = (lin & 7) / per;
    block_m = q % m_blocks;
    tile_x = (q / m_blocks) * per + (lin & 7) % per;
  } else {
    const int bx_group = blockIdx.x / (8 * m_blocks), bx_r = blockIdx.x % (8 * m_blocks);
    tile_x = bx_group * 8 + (bx_r & 7);
    block_m = bx_r >> 3;
    split_id = blockIdx.y;
  }
  if (tile_x * NG * 64 >= p.N) return;      // padding workgroup (column tiles are rounded up to a multiple of 8)
  // this wave's 64-column group. Fused silu_and_mul (p.act_out; host: NG >= 2, no K split, (N / 2) % (32 NG) == 0): the tile
  // is 32 NG gate columns plus the 32 NG up columns N / 2 further right (column groups ng >= NG / 2), paired in the epilogue.
  const bool fuse_act = NG > 1 && p.act_out != nullptr;
  const int n0 = fuse_act ? (ng >= NG / 2 ? N / 2 : 0) + (tile_x * (NG / 2) + ng % (NG > 1 ? NG / 2 : 1)) * 64 : (tile_x * NG + ng) * 64;
  const bool col_ok = n0 < N;
  const int m0 = block_m * ROWS;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_a = smem + (size_t)kslice * 2 * ABUF;  // [2][ABUF] double buffer of this K-slice

  // ---- this K-slice's range of sub-chunks ----
  const int total_steps = (K + 31) / 32;
  const int total_sub = (total_steps + SUB - 1) / SUB;
  const int nworkers = p.k_splits * KW;
  const int sub_per = (total_sub + nworkers - 1) / nworkers;
  const int worker = split_id * KW + kslice;
  const int sub_begin = min(worker * sub_per, total_sub);
  const int sub_end = min(sub_begin + sub_per, total_sub);
  // all K-slices of a workgroup run the same number of iterations (barriers): the longest one
  const int n_iter = (NG > 1) ? sub_per : (sub_end - sub_begin);

  // ---- weight addressing: lane (g, li) reads chunk 4 c8 + g of its 64-column group from k-tile row 2 ks + hi ----
  const int ktiles = SP ? K / 32 : K / 16;  // rows of the packed weight tensor
  const int64_t row_words = (int64_t)N * 16 / (I4 ? 8 : 4);
  const int32_t* bw = p.b + (int64_t)((col_ok ? n0 : 0) / 64) * WORDS64 + (4 * c8 + g) * (I4 ? 4 : 8) + (I4 ? 2 : 4) * hi;

  const bool grouped = GENERIC ? (p.num_groups > 1) : (MODE == 1);
  const bool slow_act = GENERIC && p.slow_act_order;
  const scalar_t* sc = reinterpret_cast<const scalar_t*>(p.scales);
  // grouped scales (scale_perm): position 8 c8 + b holds column c8 + 8 b; this lane's tiles are b = 4 hi + x
  const int64_t scale_off = (int64_t)((col_ok ? n0 : 0) / 64) * 64 + 8 * c8 + 4 * hi;

  uint32_t s2[NTILE];
  uint32_t zc[ZP ? NTILE : 1];  // zero points of the lane's four tile columns as packed fp16 pairs -(1024 + z)
#pragma unroll
  for (int t = 0; t < NTILE; ++t) s2[t] = 0;
  int cur_group = -1;
  auto load_group_scales = [&](int grp) {
    union { u32x2 v; scalar_t e[4]; } raw;
    raw.v = *reinterpret_cast<const u32x2*>(sc + (int64_t)grp * N + scale_off);
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      const scalar_t sv = raw.e[t];
      if constexpr (__is_same(scalar_t, f16)) {
        union { f16 h[2]; uint32_t u; } pk;
        pk.h[0] = sv;
        pk.h[1] = sv;
        s2[t] = pk.u;
      } else {
        s2[t] = __builtin_bit_cast(uint32_t, (float)sv);
      }
    }
  };

  f32x4 acc[MT][NTILE];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NTILE; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const scalar_t* A = reinterpret_cast<const scalar_t*>(p.a);

  // one k-step of weights for this lane: its words of k-tile rows 2 ks and 2 ks + 1
  // (2:4: q0 = the k-tile row's words, q1 = the lane's 8 metadata int16: tile x = 2 p + q, k-half cc at 4 p + 2 cc + q)
  using bvec_t = typename std::conditional<I4, u32x2, u32x4>::type;
  using mvec_t = typename std::conditional<SP, u32x4, bvec_t>::type;
  // X4: dense int4, fixed-pattern loop - ONE 16-byte load per lane and k-step (chunk 4 c8 + g of k-tile 2 ks + hi, all
  // four words) instead of two 8-byte ones; split_pair() trades the unused half with lane ^ 8. A CU retires about
  // one vector-memory wave instruction per ~38 cycles whatever its width, so 1 KiB instead of 512 B per instruction
  // is what lifts the per-CU weight rate.
  constexpr bool X4 = I4 && !SP && !GENERIC;
  struct BStep { bvec_t q0; mvec_t q1; u32x4 raw; };
  // metadata: reordered int16 index of (k-tile kt, column n0 + 8 c8 + 2 hi + p + 4 q, k-half cc) is
  // 2 (kt N + n0 + 32 hi + 4 c8) + 4 p + 2 cc + q (format_24.py:21-50 solved for this lane's columns)
  const int64_t meta_off = ((int64_t)(col_ok ? n0 : 0) + 32 * hi + 4 * c8) * 2;  // int16 units
  auto load_b = [&](int kstep, BStep& r) {
    kstep = min(kstep, total_steps - 1);
    if constexpr (SP) {
      r.q0 = *reinterpret_cast<const bvec_t*>(bw + (int64_t)kstep * row_words);
      r.q1 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const int16_t*>(p.meta) + (int64_t)kstep * N * 2 + meta_off);
    } else {
      const int kt0 = 2 * kstep;
      const int kt1 = min(kt0 + 1, ktiles - 1);  // K % 32 == 16 tail: activations are zero-filled there
      r.q0 = *reinterpret_cast<const bvec_t*>(bw + (int64_t)kt0 * row_words);
      r.q1 = *reinterpret_cast<const bvec_t*>(bw + (int64_t)kt1 * row_words);
    }
  };
  // activation pieces: piece id = it * (64 NG) + ng * 64 + lane -> (row, 16-B chunk of the SUB*32-k slab)
  struct ARegs { u32x4 v[PIECES]; };
  auto load_a = [&](int sub, bool valid, ARegs& r) {
    const int kbase = sub * (SUB * 32);
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
      const int piece = it * (64 * NG) + ng * 64 + lane;
      const int row = piece / (SUB * 4);
      const int cc16 = piece % (SUB * 4);
      const int k = kbase + cc16 * 8;
      const int m = m0 + row;
      const bool ok = valid && piece < NPIECE && m < M && k < K;
      u32x4 v = {0, 0, 0, 0};
      if (GENERIC && p.perm != nullptr) {
        if (ok) {
          // act-order: A'[m][k] = A[m][perm[k]] (gptq_marlin.cu:345-394)
          union { scalar_t h[8]; u32x4 u; } gth;
#pragma unroll
          for (int e = 0; e < 8; ++e) gth.h[e] = A[(int64_t)m * K + p.perm[k + e]];
          v = gth.u;
        }
      } else {
        // branch-free: clamp the address, zero by select
        const int mc = min(m, M - 1), kc = min(k, K - 8);
        const u32x4 ld = *reinterpret_cast<const u32x4*>(A + (int64_t)mc * K + kc);
        v = ok ? ld : v;
      }
      r.v[it] = v;
    }
  };
  // 16-B chunk (ks = cc16 / 4, cc = cc16 % 4), dword e2 -> fragment (ks, g = e2, row), dword cc
  // 2:4: lane group g multiplies quads g and g + 4 of the k-tile (k = 4 g + jj, 16 + 4 g + jj): dword e2 of chunk
  // cc (k = 8 cc + 2 e2 + {0, 1}) -> fragment (ks, g = 2 (cc & 1) + (e2 >> 1), row), dword 2 (cc >> 1) + (e2 & 1)
  // Rows are stored at position row ^ (2 ks): the 16 pieces of a row (4 ks x 4 cc) would otherwise hit the same 4
  // banks from 4 lanes each; with the swizzle a ds_write_b32 of 32 lanes touches 32 different banks. The fragment
  // read permutes its 16 row-lanes the same way (still 16 distinct consecutive 16-byte slots).
  auto store_a = [&](const ARegs& r, char* buf) {
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
      const int piece = it * (64 * NG) + ng * 64 + lane;
      const int cc16 = piece % (SUB * 4);
      const int ks = cc16 >> 2, cc = cc16 & 3;
      const int row = (piece / (SUB * 4)) ^ (2 * ks);
      if (piece < NPIECE) {
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          if constexpr (SP)
            *reinterpret_cast<uint32_t*>(buf + (((ks * 4 + 2 * (cc & 1) + (e2 >> 1)) * ROWS + row) * 16) +
                                         4 * (2 * (cc >> 1) + (e2 & 1))) = r.v[it][e2];
          else
            *reinterpret_cast<uint32_t*>(buf + (((ks * 4 + e2) * ROWS + row) * 16) + 4 * cc) = r.v[it][e2];
        }
      }
    }
  };
  auto sync_slice = [&]() {
    if constexpr ((NMX_ABLATE & 8) != 0) __builtin_amdgcn_wave_barrier();
    else if constexpr (NG > 1) __syncthreads();
    else __builtin_amdgcn_wave_barrier();  // wave-private buffer: LDS ops of one wave complete in order
  };

  // MODE 1 with few row tiles: MFMA on UNSCALED integer-valued weights into per-group accumulators and apply the
  // group scale once per group in fp32 (16 MT FMAs per 128 k instead of 16 packed multiplies per 32 k). This is
  // sum_g s_g * (a . (q - 8)) exactly — slightly MORE accurate than the reference's fp16-rounded (q - 8) * s.
#ifndef NMX_ACC_SCALE_MAX_MT
#define NMX_ACC_SCALE_MAX_MT 2
#endif
  constexpr bool ACC_SCALE = (MODE == 1) && (MT <= NMX_ACC_SCALE_MAX_MT);
  f32x4 gacc[ACC_SCALE ? MT : 1][NTILE];
  float srow[ACC_SCALE ? NTILE : 1][4];

  // one 32-k step of MFMAs for this wave: weights `cur`, activation fragments from `abuf`.
  // GA = 0: accumulate (scaled weights) into acc; GA = 1: start a group in gacc; GA = 2: continue in gacc
  auto compute_step = [&](const BStep& cur, int ksl, int kstep, const char* abuf, auto ga_c) {
    constexpr int GA = decltype(ga_c)::value;
    if (GENERIC && grouped && !slow_act) {
      const int grp = min((kstep * 32) / p.group_size, p.num_groups - 1);
      if (grp != cur_group) {
        cur_group = grp;
        load_group_scales(grp);
      }
    }
    u32x4 af[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if constexpr ((NMX_ABLATE & 4) != 0) af[mt] = u32x4{(uint32_t)lane, (uint32_t)ksl, (uint32_t)mt, 0x3c003c00u};
      else af[mt] = *reinterpret_cast<const u32x4*>(abuf + (((ksl * 4 + g) * ROWS + mt * 16 + (li ^ (2 * ksl))) * 16));
    }

    if constexpr (SP) {
      // index registers: byte 0 (ABID 0) = positions for tile 2 p, byte 2 (ABID 2) = tile 2 p + 1; low nibble = quad
      // g (k-half 0), high nibble = quad g + 4 (k-half 1). Bits outside the selected byte are ignored.
      uint32_t xidx[2];
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const uint32_t m0 = cur.q1[2 * pp] >> (4 * g), m1 = (cur.q1[2 * pp + 1] >> (4 * g)) << 4;
        xidx[pp] = (m0 & 0x000f000fu) | (m1 & ~0x000f000fu);
      }
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        // tile x = t = 2 p + q: column 8 c8 + 2 hi + p + 4 q. int4: word p, block q 8 bits up; 8-bit: word 2 p + q
        const uint32_t w = I4 ? (cur.q0[t >> 1] >> (8 * (t & 1))) : cur.q0[t];
        uint32_t d0, d1;  // kept values of quad g, of quad g + 4
        Dequant<scalar_t, KIND>::run(w, s2[t], grouped && GA == 0, d0, d1);
        const f16x4 wa = __builtin_bit_cast(f16x4, u32x2{d0, d1});
        auto sm = [&](f32x4 cin, int mt) -> f32x4 {
          const f16x8 ab = __builtin_bit_cast(f16x8, af[mt]);
          if (t & 1) return __builtin_amdgcn_smfmac_f32_16x16x32_f16(wa, ab, cin, (int)xidx[t >> 1], 0, 2);
          return __builtin_amdgcn_smfmac_f32_16x16x32_f16(wa, ab, cin, (int)xidx[t >> 1], 0, 0);
        };
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if constexpr (GA == 0) acc[mt][t] = sm(acc[mt][t], mt);
          else if constexpr (GA == 1) gacc[ACC_SCALE ? mt : 0][t] = sm(f32x4{0.f, 0.f, 0.f, 0.f}, mt);
          else gacc[ACC_SCALE ? mt : 0][t] = sm(gacc[ACC_SCALE ? mt : 0][t], mt);
        }
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      // tile x = t: column c8 + 8 t + 32 hi. int4: word t >> 1 of the lane's pair, "+8 column" half 8 bits up;
      // 8-bit: word t of the lane's four
      uint32_t w0, w1;
      if constexpr (I4) {
        w0 = cur.q0[t >> 1] >> (8 * (t & 1));
        w1 = cur.q1[t >> 1] >> (8 * (t & 1));
      } else {
        w0 = cur.q0[t];
        w1 = cur.q1[t];
      }
      uint32_t d0, d1, d2, d3;
      u32x4 wf;
      if constexpr ((NMX_ABLATE & 2) != 0) {
        wf = u32x4{w0, w1, w0 ^ s2[t], w1};
      } else if constexpr (ZP) {
        const uint32_t zh = h2_bits(bits_h2(zc[t]) + bits_h2(0x63806380u));  // -(1024 + z) + 960 = -(64 + z), exact
        dequant_zp_f16(w0, s2[t], GA == 0, zc[t], zh, d0, d1);
        dequant_zp_f16(w1, s2[t], GA == 0, zc[t], zh, d2, d3);
        wf = u32x4{d0, d1, d2, d3};
      } else if (!slow_act) {
        Dequant<scalar_t, KIND>::run(w0, s2[t], grouped && GA == 0, d0, d1);
        Dequant<scalar_t, KIND>::run(w1, s2[t], grouped && GA == 0, d2, d3);
        wf = u32x4{d0, d1, d2, d3};
      } else {
        // act-order on a K-shard (is_k_full == false): every k-row carries its own group id
        // (gptq_marlin.cu:965-980 with g_idx). Dequantise unscaled, then scale element-wise.
        Dequant<scalar_t, KIND>::run(w0, 0, false, d0, d1);
        Dequant<scalar_t, KIND>::run(w1, 0, false, d2, d3);
        union { u32x4 u; scalar_t h[8]; } wv;
        wv.u = u32x4{d0, d1, d2, d3};
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int koff[4] = {0, 1, 8, 9};
          int k = kstep * 32 + 16 * (jj >> 2) + 2 * g + koff[jj & 3];
          k = min(k, K - 1);
          const int grp = p.g_idx[k];
          const float sv = Scalar<scalar_t>::to_f32(sc[(int64_t)grp * N + scale_off + t]);
          wv.h[jj] = Scalar<scalar_t>::from_f32(Scalar<scalar_t>::to_f32(wv.h[jj]) * sv);
        }
        wf = wv.u;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if constexpr ((NMX_ABLATE & 1) != 0) {
          acc[mt][t][0] += __builtin_bit_cast(float, wf[mt & 3] ^ af[mt][t & 3]);
          continue;
        }
        if constexpr (GA == 0) acc[mt][t] = mfma_16x16x32<scalar_t>(wf, af[mt], acc[mt][t]);
        else if constexpr (GA == 1) gacc[ACC_SCALE ? mt : 0][t] = mfma_16x16x32<scalar_t>(wf, af[mt], f32x4{0.f, 0.f, 0.f, 0.f});
        else gacc[ACC_SCALE ? mt : 0][t] = mfma_16x16x32<scalar_t>(wf, af[mt], gacc[ACC_SCALE ? mt : 0][t]);
      }
    }
#ifndef NMX_SCHED_PIPE
#define NMX_SCHED_PIPE 10  // VALU instructions ahead of the first MFMA of a k-step; 0 = leave the order to hipcc
#endif
#if NMX_SCHED_PIPE > 0
    // (measured at M = 256: gate_up 84.0 -> 78.9 us, down 45.9 -> 44.5; no change at M <= 64)
    // Ask the scheduler for the software pipeline the data flow allows inside one k-step: fragment reads first, the
    // first tile's dequantisation, then one MFMA followed by the VALU work that fits in its 16-cycle shadow (the next
    // tile's dequantisation), NTILE * MT times.
    if constexpr (!GENERIC && MT >= 2) {
      __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);       // DS reads: the activation fragments
      __builtin_amdgcn_sched_group_barrier(0x002, NMX_SCHED_PIPE, 0);  // VALU: first tile
#pragma unroll
      for (int i = 0; i < NTILE * MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // three VALU in its shadow
      }
    }
#endif
  };
  using GA0 = std::integral_constant<int, 0>;
  using GA1 = std::integral_constant<int, 1>;
  using GA2 = std::integral_constant<int, 2>;

  if constexpr (GENERIC) {
    // ---- compiler-scheduled loop (any group size, act-order gather) ----
    BStep ring[PF];
    ARegs areg;
    if (n_iter > 0) {
#pragma unroll
      for (int i = 0; i < PF; ++i) load_b(sub_begin * SUB + i, ring[i]);
      load_a(sub_begin, sub_begin < sub_end, areg);
      store_a(areg, lds_a);
    }
    sync_slice();
    auto body = [&](auto par_c, int it) {
      constexpr int PAR = decltype(par_c)::value;
      const int sub = sub_begin + it;
      const bool have_next = (sub + 1) < sub_end;
      load_a(sub + 1, have_next && (it + 1 < n_iter), areg);
#pragma unroll
      for (int ksl = 0; ksl < SUB; ++ksl) {
        const int kstep = sub * SUB + ksl;
        BStep cur = ring[PAR * SUB + ksl];
        load_b(kstep + PF, ring[PAR * SUB + ksl]);
        compute_step(cur, ksl, kstep, lds_a + PAR * ABUF, GA0{});
      }
      store_a(areg, lds_a + (PAR ^ 1) * ABUF);
      sync_slice();
    };
    for (int it = 0; it < n_iter; it += 2) {
      body(std::integral_constant<int, 0>{}, it);
      if (it + 1 < n_iter) body(std::integral_constant<int, 1>{}, it + 1);
    }
  } else {
    // ---- fixed-pattern loop: every vector-memory op is a compiler-visible buffer load issued in one periodic order
    // (pinned with sched_barrier), so that hipcc's own counted s_waitcnt vmcnt(N) keep PF k-steps of weights in flight.
    // Issue order per sub-chunk body:  [batch: PIECES activation loads (+ scale rows) for sub-chunk sub + AD]
    //                                  then per k-step [compute(kstep)] [load weights(kstep + PF) into the freed slot],
    //                                  then [write batch(sub + 1) to LDS] [barrier].
    // The activation tile is requested AD = 2 sub-chunks ahead (two register sets): an L2 round trip under a full
    // weight stream is ~1 us, about what a whole sub-chunk of compute takes, and with AD = 1 every iteration ended
    // on that latency (measured: a kernel with all compute removed still took 55 % of the full time).
    // Three things keep the waits counted (the rules marlin_decode_kernel / marlin_wide_kernel were built on): the
    // prologue queues its loads in exactly the loop's order; the loop runs whole PAIRS of bodies with nothing
    // conditional inside (an odd sub-chunk count ends in a peeled single body - a conditional second half would put a
    // path body 0 -> body 0 into the loop and merge its pending-load order into every wait); loads past the range are
    // clamped / out of the descriptor's range, never skipped.
    constexpr int AD = 2;
    constexpr int NSC = (MODE == 1) ? (ACC_SCALE ? 4 : 1) : 0;  // scale loads per batch
    BStep ring[PF];
    ARegs areg[AD];
    u32x2 sraw[AD][4];
    u32x2 zraw[AD];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      ring[i].q0 = bvec_t{};
      ring[i].q1 = mvec_t{};
      ring[i].raw = u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int d = 0; d < AD; ++d) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) areg[d].v[i] = u32x4{0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 4; ++i) sraw[d][i] = u32x2{0, 0};
      zraw[d] = u32x2{0, 0};
    }

    // fast modes require K % (32 SUB) == 0, so every k-tile row of an in-range k-step exists; rows past the end of
    // the matrix (prefetch beyond the slice) are out of the descriptor's range and read as zeros
    const int row_bytes = (int)(row_words * 4);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(p.b, (int64_t)ktiles * row_bytes);
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a, (int64_t)M * K * sizeof(scalar_t));
    const __amdgpu_buffer_rsrc_t rs_s = make_rsrc(p.scales, (int64_t)p.num_groups * N * sizeof(scalar_t));
    const __amdgpu_buffer_rsrc_t rs_m = make_rsrc(SP ? p.meta : p.b, SP ? (int64_t)ktiles * N * 4 : 0);
    const __amdgpu_buffer_rsrc_t rs_z = make_rsrc(ZP ? p.zeros : p.scales, ZP ? (int64_t)p.num_groups * N * sizeof(scalar_t) : 0);
    const int z_voff = (int)(scale_off * sizeof(scalar_t));  // weight layout: positions 8 c8 + 4 hi + t
    const int b_voff = X4 ? (int)((bw - p.b - 2 * hi) * 4) + hi * row_bytes : (int)((bw - p.b) * 4);
    const int m_voff = (int)(meta_off * 2);
    // 16-row tiles and the split-free 128-column 8-wave tile only run with one row block: non-temporal weight loads
    constexpr int W_AUX = (X4 && NMX_SKINNY_NT && (MT == 1 || (NMX_SKINNY_NT > 1 && MT == 4 && NG == 2 && W8))) ? 2 : 0;
    auto issue_b = [&](int kstep, BStep& r) {
      if constexpr ((NMX_ABLATE & 128) != 0) return;
      if constexpr (SP) {
        const int soff = kstep * row_bytes;  // wave-uniform (the K slice depends on the wave id only)
        if constexpr (I4) r.q0 = __builtin_amdgcn_raw_buffer_load_b64(rs_b, b_voff, soff, 0);
        else r.q0 = __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff, soff, 0);
        r.q1 = __builtin_amdgcn_raw_buffer_load_b128(rs_m, m_voff, kstep * N * 4, 0);
      } else {
        const int soff = 2 * kstep * row_bytes;
        if constexpr (X4) {
          r.raw = __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff, soff, W_AUX);
        } else if constexpr (I4) {
          r.q0 = __builtin_amdgcn_raw_buffer_load_b64(rs_b, b_voff, soff, 0);
          r.q1 = __builtin_amdgcn_raw_buffer_load_b64(rs_b, b_voff, soff + row_bytes, 0);
        } else {
          r.q0 = __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff, soff, 0);
          r.q1 = __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff, soff + row_bytes, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // the load keeps the program position it is given here
    };
    // per-lane byte offsets of the activation pieces inside a [ROWS x SUB*32] slab at (m0, k = 0)
    int a_voff[PIECES];
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
      const int piece = it * (64 * NG) + ng * 64 + lane;
      const int row = min(piece / (SUB * 4), ROWS - 1);
      const int cc16 = piece % (SUB * 4);
      // rows >= M (and the padding pieces of a short tile) get an offset beyond the descriptor's range: the
      // hardware returns zeros for them without touching memory
      a_voff[it] = (piece < NPIECE && (m0 + row) < M) ? (int)(((int64_t)(m0 + row) * K + cc16 * 8) * sizeof(scalar_t))
                                                      : (int)0x7ff00000;
    }
    // scale offsets. per-weight scaling: the lane's 4 tile columns c8 + 8 x + 32 hi -> positions 8 c8 + 4 hi + x;
    // per-group scaling of the accumulators: D row r of tile x is column 32 (g>>1) + 8 x + 4 (g&1) + r ->
    // position 8 (4 (g&1) + r) + 4 (g>>1) + x: four 8-byte loads (r = 0..3), x contiguous
    const int s_voff = ACC_SCALE ? (int)((((col_ok ? n0 : 0) / 64) * 64 + 32 * (g & 1) + 4 * (g >> 1)) * sizeof(scalar_t))
                                 : (int)(scale_off * sizeof(scalar_t));
    auto issue_batch = [&](int sub, auto set_c) {
      constexpr int SET = decltype(set_c)::value;
      const int kbase = sub * (SUB * 32);
      const int soff = kbase * (int)sizeof(scalar_t);  // wave-uniform
      if constexpr ((NMX_ABLATE & 64) == 0) {
#pragma unroll
        for (int it = 0; it < PIECES; ++it) areg[SET].v[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff[it], soff, 0);
      }
      if constexpr (MODE == 1) {
        const int grp = min(kbase / p.group_size, p.num_groups - 1);
#pragma unroll
        for (int r = 0; r < NSC; ++r)
          sraw[SET][r] = __builtin_amdgcn_raw_buffer_load_b64(rs_s, s_voff + 8 * r * (int)sizeof(scalar_t), grp * N * (int)sizeof(scalar_t), 0);
        if constexpr (ZP) zraw[SET] = __builtin_amdgcn_raw_buffer_load_b64(rs_z, z_voff, grp * N * (int)sizeof(scalar_t), 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // make the landed batch usable: zero the out-of-range pieces, write the fragments, unpack the scales
    auto land_batch = [&](int sub, bool valid, char* buf, auto set_c) {
      constexpr int SET = decltype(set_c)::value;
      // rows >= M and k >= K were read as zeros through the buffer descriptor (fast modes: K % (32 SUB) == 0 and
      // pieces beyond the tile are never stored); only a K slice that has run out of sub-chunks must be blanked
      // (a wave-uniform branch with no load inside: the pending-load order is the same on both paths)
      if (!valid) {
#pragma unroll
        for (int it = 0; it < PIECES; ++it) areg[SET].v[it] = u32x4{0, 0, 0, 0};
      }
      if constexpr ((NMX_ABLATE & 256) == 0) store_a(areg[SET], buf);
      if constexpr (ZP) {
        union { u32x2 v; f16 e[4]; } zr;
        zr.v = zraw[SET];
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
          union { f16 h[2]; uint32_t u; } pk;
          pk.h[0] = zr.e[t];
          pk.h[1] = zr.e[t];
          zc[t] = pk.u;
        }
      }
      if constexpr (ACC_SCALE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          union { u32x2 v; scalar_t e[4]; } raw;
          raw.v = sraw[SET][r];
#pragma unroll
          for (int t = 0; t < NTILE; ++t) {
            srow[t][r] = Scalar<scalar_t>::to_f32(raw.e[t]);
            // convert HERE: left to itself hipcc sinks the conversions to the end of the next body, keeps the raw rows
            // alive across the re-issue of their register set and copies the new rows into place at the loop end - behind
            // a vmcnt wait for loads that are only a few k-steps old
            asm volatile("" : "+v"(srow[t][r]));
          }
        }
      } else if constexpr (MODE == 1) {
        union { u32x2 v; scalar_t e[4]; } raw;
        raw.v = sraw[SET][0];
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
          const scalar_t sv = raw.e[t];
          if constexpr (__is_same(scalar_t, f16)) {
            union { f16 h[2]; uint32_t u; } pk;
            pk.h[0] = sv;
            pk.h[1] = sv;
            s2[t] = pk.u;
          } else {
            s2[t] = __builtin_bit_cast(uint32_t, (float)sv);
          }
        }
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    static_assert(PIECES == 1 || PIECES == 2 || PIECES == 4 || PIECES == 8, "unsupported activation piece count");
    // X4: the lane's 16 bytes are chunk 4 c8 + g of ONE k-tile; trade the unused half with lane ^ 8
    auto ready_b = [&](BStep& r) {
      if constexpr (X4 && std::is_same<bvec_t, u32x2>::value && std::is_same<mvec_t, u32x2>::value) split_pair(r.raw, hi != 0, r.q0, r.q1);
    };

    if (n_iter > 0) {
      // prologue in the steady-state pattern: [batch] [SUB weight loads] [batch] [SUB weight loads]
      issue_batch(sub_begin, S0{});
#pragma unroll
      for (int i = 0; i < SUB; ++i) issue_b(sub_begin * SUB + i, ring[i]);
      issue_batch(sub_begin + 1, S1{});
#pragma unroll
      for (int i = SUB; i < PF; ++i) issue_b(sub_begin * SUB + i, ring[i]);
      land_batch(sub_begin, sub_begin < sub_end, lds_a, S0{});
    }
    sync_slice();

    auto body = [&](auto par_c, int it) {
      constexpr int PAR = decltype(par_c)::value;
      const int sub = sub_begin + it;
      const bool have_next = ((sub + 1) < sub_end) && (it + 1 < n_iter);
      issue_batch(sub + 2, std::integral_constant<int, PAR>{});
#pragma unroll
      for (int ksl = 0; ksl < SUB; ++ksl) {
        const int kstep = sub * SUB + ksl;
        BStep& r = ring[PAR * SUB + ksl];
        ready_b(r);
        if constexpr (!ACC_SCALE) compute_step(r, ksl, kstep, lds_a + PAR * ABUF, GA0{});
        else if (ksl == 0) compute_step(r, ksl, kstep, lds_a + PAR * ABUF, GA1{});
        else compute_step(r, ksl, kstep, lds_a + PAR * ABUF, GA2{});
        issue_b(kstep + PF, r);
      }
      if constexpr (ACC_SCALE) {
        // one group (sub-chunk) done: acc += scale[column] * group accumulator
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < NTILE; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)  // an explicit fma: under -ffp-contract=fast hipcc fused this in one body of the pair and
              acc[mt][t][r] = __builtin_fmaf(srow[t][r], gacc[mt][t][r], acc[mt][t][r]);  // not in the other (last-bit differences between tile shapes)
      }
      land_batch(sub + 1, have_next, lds_a + (PAR ^ 1) * ABUF, std::integral_constant<int, PAR ^ 1>{});
      sync_slice();
    };
    int it = 0;
    for (; it + 1 < n_iter; it += 2) {
      body(std::integral_constant<int, 0>{}, it);
      body(std::integral_constant<int, 1>{}, it + 1);
    }
    if (it < n_iter) body(std::integral_constant<int, 0>{}, it);  // odd count: peeled last body
  }


  // ---- channel-wise scales are applied to the fp32 accumulators (rows of D = column slots 4 g + r) ----
  if (!grouped && !slow_act) {
    // scale_perm_single (marlin_perms.py:44-47): within a 32-column chunk position 8 (c/2) + (c%2) + 2 b' holds
    // column c + 8 b'. D row 4 g + r of tile x is column 32 (g >> 1) + 8 x + 4 (g & 1) + r of the 64 group.
    const int64_t base64 = (int64_t)((col_ok ? n0 : 0) / 64) * 64;
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int pos;
        if constexpr (SP) {
          // marlin_24_scale_perm_single is the identity; D row r of tile x = 2 p + q is column
          // 8 (4 (g & 1) + r) + 2 (g >> 1) + p + 4 q
          pos = 8 * (4 * (g & 1) + r) + 2 * (g >> 1) + (t >> 1) + 4 * (t & 1);
        } else {
          const int col = 32 * (g >> 1) + 8 * t + 4 * (g & 1) + r;
          const int cc = col & 7, b = col >> 3;
          pos = 32 * (b >> 2) + 8 * (cc >> 1) + (cc & 1) + 2 * (b & 3);
        }
        const float sv = Scalar<scalar_t>::to_f32(sc[base64 + pos]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][t][r] *= sv;
      }
    }
  }

  // ---- reduce the KW K-slices (tree through LDS); slice 0 writes ----
  constexpr int ACC_FLOATS = MT * NTILE * 64 * 4;
  float* red = reinterpret_cast<float*>(smem);
  if constexpr (KW > 1) {
#pragma unroll
    for (int stride = KW / 2; stride >= 1; stride >>= 1) {
      __syncthreads();
      if (kslice >= stride && kslice < 2 * stride) {
        float* dst = red + ((kslice - stride) * NG + ng) * ACC_FLOATS;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < NTILE; ++t) *reinterpret_cast<f32x4*>(dst + ((mt * NTILE + t) * 64 + lane) * 4) = acc[mt][t];
      }
      __syncthreads();
      if (kslice < stride) {
        const float* src = red + (kslice * NG + ng) * ACC_FLOATS;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < NTILE; ++t)
            acc[mt][t] += *reinterpret_cast<const f32x4*>(src + ((mt * NTILE + t) * 64 + lane) * 4);
      }
    }
  }
  if constexpr (NG > 1 && !SP) {
    if (fuse_act) {
      // silu_and_mul on the ROUNDED gate and up values, the arithmetic of act_and_mul_kernel (activation_kernels.cu:12-30)
      constexpr int HW = NG / 2;
      u32x2* ex = reinterpret_cast<u32x2*>(smem);
      __syncthreads();  // staging buffers / reduction slabs are free
      const int pair = (ng % HW) * (MT * NTILE * 64);
      if (kslice == 0 && ng >= HW) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < NTILE; ++t) {
            union { scalar_t h[4]; u32x2 u; } r;
#pragma unroll
            for (int j = 0; j < 4; ++j) r.h[j] = Scalar<scalar_t>::from_f32(acc[mt][t][j]);
            ex[pair + (mt * NTILE + t) * 64 + lane] = r.u;
          }
      }
      __syncthreads();
      if (kslice != 0 || ng >= HW || !col_ok) return;
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        const int n = n0 + 32 * (g >> 1) + 8 * t + 4 * (g & 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int m = m0 + mt * 16 + li;
          union { scalar_t h[4]; u32x2 u; } up, o;
          up.u = ex[pair + (mt * NTILE + t) * 64 + lane];
#pragma unroll
          for (int j = 0; j < 4; ++j) o.h[j] = rnd_mul<scalar_t>(silu_rnd<scalar_t>(Scalar<scalar_t>::from_f32(acc[mt][t][j])), up.h[j]);
          if (m < M) *reinterpret_cast<u32x2*>(reinterpret_cast<scalar_t*>(p.act_out) + (int64_t)m * (N / 2) + n) = o.u;
        }
      }
      return;
    }
  }
  if (kslice != 0 || !col_ok) return;
  if constexpr ((NMX_ABLATE & 32) != 0) {
    if (acc[0][0][0] != 12345.678f) return;
  }

  if constexpr (SP) {
    // lane (g, li): D row r of tile x = 2 p + q is column 8 (4 (g & 1) + r) + 2 (g >> 1) + p + 4 q: tiles q and 2 + q
    // are neighbouring columns -> 2-element stores
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + mt * 16 + li;
      if (m >= M) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int n = n0 + 8 * (4 * (g & 1) + r) + 2 * (g >> 1) + 4 * q;
          const float v0 = acc[mt][q][r], v1 = acc[mt][2 + q][r];
          if (p.k_splits == 1) {
            union { scalar_t h[2]; uint32_t u; } o;
            o.h[0] = Scalar<scalar_t>::from_f32(v0);
            o.h[1] = Scalar<scalar_t>::from_f32(v1);
            *reinterpret_cast<uint32_t*>(reinterpret_cast<scalar_t*>(p.c) + (int64_t)m * N + n) = o.u;
          } else {
            *reinterpret_cast<f32x2*>(p.partial + ((int64_t)split_id * M + m) * N + n) = f32x2{v0, v1};
          }
        }
      }
    }
    return;
  }
  // lane (g, li): D rows = 4 consecutive output columns 32 (g >> 1) + 8 t + 4 (g & 1) + r, D col = activation row li
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    const int n = n0 + 32 * (g >> 1) + 8 * t + 4 * (g & 1);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + mt * 16 + li;
      if (m >= M) continue;
      if (p.k_splits == 1) {
        union { scalar_t h[4]; u32x2 u; } r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.h[j] = Scalar<scalar_t>::from_f32(acc[mt][t][j]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<scalar_t*>(p.c) + (int64_t)m * N + n) = r.u;
      } else {
        *reinterpret_cast<f32x4*>(p.partial + ((int64_t)split_id * M + m) * N + n) = acc[mt][t];
      }
    }
  }
}

// ---- large-M (prefill) kernel ------------------------------------------------------------------------------------
// M > 128: the MFMA-bound regime. A 256-row x 256-column tile per 8-wave workgroup; every packed weight is fetched and
// dequantised ONCE per workgroup (each wave converts the 16-byte Marlin chunk pair it loaded into the four MFMA
// operand fragments exactly as the decode kernel does and parks them in LDS), then all eight waves (2 row halves x 4
// column groups, 128 x 64 outputs each = 32 accumulator tiles) read weight and activation fragments from LDS with
// ds_read_b128: 12 LDS reads per 32 MFMAs, ~2 dequant VALU per MFMA instead of ~4 in the decode kernel's row blocks.
// Stages of 64 k (two 32-k steps), LDS double-buffered (2 x 64 KiB), global loads of stage s+1 in flight during the
// MFMAs of stage s. Group scales (group % 64 == 0) are folded into the dequantised fp16 weights like the reference;
// channel-wise scales are applied to the fp32 accumulators. grid (N / 256 rounded up, k_splits, M / 256 rounded up).
// NGRP = 64-column groups per workgroup (4: 8 waves, 256 columns, LDS double-buffered, one workgroup per CU;
// 2: 4 waves, 128 columns, ONE LDS stage of 48 KiB so that two workgroups share a CU and one computes while the other
// stages - the phases of a barrier-synchronised workgroup otherwise leave the MFMA pipes idle half the time).
template <typename scalar_t, int KIND, int MODE, int NGRP>
__global__ __launch_bounds__(128 * NGRP, 2) void marlin_large_kernel(const GemmParams p) {
  constexpr bool I4 = (KIND == W_INT4);
  constexpr int BM = 256;
  constexpr int NTHR = 128 * NGRP;
  constexpr int APIECES = 2048 / NTHR;         // 16-byte activation pieces per thread and stage
  constexpr bool DBUF = (NGRP == 4);
  constexpr int WORDS64 = I4 ? 128 : 256;
  constexpr int W_IMG = 2 * NGRP * 4 * 64 * 16;  // bytes: [k-step][column group][tile][lane] x 16 B
  constexpr int A_IMG = 2 * 4 * BM * 16;         // bytes: [k-step][g][row] x 16 B
  constexpr int STAGE = W_IMG + A_IMG;
  using bvec_t = typename std::conditional<I4, u32x2, u32x4>::type;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15, c8 = li & 7, hi = li >> 3;
  const int wm = wave / NGRP, wn = wave % NGRP;   // compute role: rows 128 wm .., column group wn
  const int lw = wave % NGRP, kw = wave / NGRP;   // load role: column group lw, k-step kw of the stage
  const int N = p.N, K = p.K, M = p.M;
  const int nb = blockIdx.x * (64 * NGRP), m0 = blockIdx.z * BM;
  const int n_load = nb + 64 * lw;
  const bool load_ok = n_load < N;
  const int stages_total = K / 64;
  const int per = (stages_total + p.k_splits - 1) / p.k_splits;
  const int st_begin = min((int)blockIdx.y * per, stages_total), st_end = min(st_begin + per, stages_total);

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int64_t row_words = (int64_t)N * 16 / (I4 ? 8 : 4);
  const int32_t* bw = p.b + (int64_t)((load_ok ? n_load : 0) / 64) * WORDS64 + (4 * c8 + g) * (I4 ? 4 : 8) + (I4 ? 2 : 4) * hi;
  const scalar_t* sc = reinterpret_cast<const scalar_t*>(p.scales);
  const int64_t scale_off = (int64_t)(load_ok ? n_load : 0) + 8 * c8 + 4 * hi;
  const scalar_t* A = reinterpret_cast<const scalar_t*>(p.a);

  f32x4 acc[8][4];
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  bvec_t wq0, wq1;
  u32x4 areg[APIECES];
  uint32_t s2[4] = {0, 0, 0, 0};
  u32x2 sraw = {0, 0};
  auto load_stage = [&](int st) {
    const int kstep = st * 2 + kw;
    wq0 = *reinterpret_cast<const bvec_t*>(bw + (int64_t)(2 * kstep) * row_words);
    wq1 = *reinterpret_cast<const bvec_t*>(bw + (int64_t)(2 * kstep + 1) * row_words);
#pragma unroll
    for (int it = 0; it < APIECES; ++it) {
      const int piece = it * NTHR + (int)threadIdx.x;
      const int row = piece >> 3, cc8 = piece & 7;
      const int m = min(m0 + row, M - 1);
      const u32x4 v = *reinterpret_cast<const u32x4*>(A + (int64_t)m * K + st * 64 + cc8 * 8);
      areg[it] = (m0 + row < M) ? v : u32x4{0, 0, 0, 0};
    }
    if constexpr (MODE == 1) {
      // loaded every stage and unpacked in write_stage: a conditional reload would need its data (and so a full
      // vmcnt(0) drain of this stage's loads) before the MFMAs instead of after them
      const int grp = min((st * 64) / p.group_size, p.num_groups - 1);
      sraw = *reinterpret_cast<const u32x2*>(sc + (int64_t)grp * N + scale_off);
    }
  };
  auto write_stage = [&](char* buf) {
    if constexpr (MODE == 1) {
      union { u32x2 v; scalar_t e[4]; } raw;
      raw.v = sraw;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (__is_same(scalar_t, f16)) {
          union { f16 h[2]; uint32_t u; } pk;
          pk.h[0] = raw.e[t];
          pk.h[1] = raw.e[t];
          s2[t] = pk.u;
        } else {
          s2[t] = __builtin_bit_cast(uint32_t, (float)raw.e[t]);
        }
      }
    }
    // weights: the same chunk -> fragment conversion as the decode kernel (tile x = t: column c8 + 8 t + 32 hi)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint32_t w0, w1, d0, d1, d2, d3;
      if constexpr (I4) {
        w0 = wq0[t >> 1] >> (8 * (t & 1));
        w1 = wq1[t >> 1] >> (8 * (t & 1));
      } else {
        w0 = wq0[t];
        w1 = wq1[t];
      }
      Dequant<scalar_t, KIND>::run(w0, s2[t], MODE == 1, d0, d1);
      Dequant<scalar_t, KIND>::run(w1, s2[t], MODE == 1, d2, d3);
      *reinterpret_cast<u32x4*>(buf + (((kw * NGRP + lw) * 4 + t) * 64 + lane) * 16) = u32x4{d0, d1, d2, d3};
    }
    // activations: 16-byte piece (row, k-step ks, chunk cc), dword e2 -> fragment (ks, g = e2, row), dword cc
    char* ab = buf + W_IMG;
#pragma unroll
    for (int it = 0; it < APIECES; ++it) {
      const int piece = it * NTHR + (int)threadIdx.x;
      const int cc8 = piece & 7, ks = cc8 >> 2, cc = cc8 & 3;
      const int row = (piece >> 3) ^ (2 * ks);
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2)
        *reinterpret_cast<uint32_t*>(ab + ((ks * 4 + e2) * BM + row) * 16 + 4 * cc) = areg[it][e2];
    }
  };
  auto compute_stage = [&](const char* buf) {
    const char* ab = buf + W_IMG;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 wf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const u32x4*>(buf + (((ks * NGRP + wn) * 4 + t) * 64 + lane) * 16);
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const u32x4 af = *reinterpret_cast<const u32x4*>(ab + ((ks * 4 + g) * BM + wm * 128 + mt * 16 + (li ^ (2 * ks))) * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mt][t] = mfma_16x16x32<scalar_t>(wf[t], af, acc[mt][t]);
      }
    }
  };

  if (st_begin < st_end) {
    load_stage(st_begin);
    write_stage(smem);
  }
  __syncthreads();
  if constexpr (DBUF) {
    // branch-free body (the last iteration re-stages its own stage into the idle buffer) so that the scheduler can put
    // the next stage's global loads, dequantisation and LDS writes between this stage's MFMAs
    for (int st = st_begin; st < st_end; ++st) {
      const int par = (st - st_begin) & 1;
      load_stage(min(st + 1, st_end - 1));
      compute_stage(smem + par * STAGE);
      write_stage(smem + (par ^ 1) * STAGE);
      __syncthreads();
    }
  } else {
    // one LDS stage: the next stage's global loads fly during the MFMAs, then [barrier] write [barrier]
    for (int st = st_begin; st < st_end; ++st) {
      load_stage(min(st + 1, st_end - 1));
      compute_stage(smem);
      __syncthreads();
      write_stage(smem);
      __syncthreads();
    }
  }

  const int n0 = nb + 64 * wn;
  if (n0 >= N) return;
  if constexpr (MODE == 0) {
    // scale_perm_single: D row 4 g + r of tile t is column 32 (g >> 1) + 8 t + 4 (g & 1) + r of the 64-column group
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 32 * (g >> 1) + 8 * t + 4 * (g & 1) + r;
        const int cc = col & 7, b = col >> 3;
        const float sv = Scalar<scalar_t>::to_f32(sc[n0 + 32 * (b >> 2) + 8 * (cc >> 1) + (cc & 1) + 2 * (b & 3)]);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[mt][t][r] *= sv;
      }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = n0 + 32 * (g >> 1) + 8 * t + 4 * (g & 1);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int m = m0 + wm * 128 + mt * 16 + li;
      if (m >= M) continue;
      if (p.k_splits == 1) {
        union { scalar_t h[4]; u32x2 u; } r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.h[j] = Scalar<scalar_t>::from_f32(acc[mt][t][j]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<scalar_t*>(p.c) + (int64_t)m * N + n) = r.u;
      } else {
        *reinterpret_cast<f32x4*>(p.partial + ((int64_t)blockIdx.y * M + m) * N + n) = acc[mt][t];
      }
    }
  }
}


// ---- decode kernel for M <= 32 (int4, K % 128 == 0, channel-wise or group % 128 == 0 scales) -------------------------
// What the skinny kernel above pays at M <= 32 is not bandwidth but fixed cost: 64-96 workgroups x 4 waves x 8 KiB of
// weights in flight cannot cover the HBM latency (Little: ~70 KiB per CU are needed), the cure - more workgroups along
// K - buys a second dependent launch (the split-K reduce, ~5 us) and fp32 partial traffic, and the LDS staging of the
// activations needs barriers / a second in-order queue. This kernel removes all three for the small-M case:
//   * one workgroup = NW (8 or 16) waves on ONE 64-column group, each wave a K slice: up to 16 x 16 KiB in flight per
//     CU without any cross-workgroup split for K = 4096 (gridDim.y K splits remain for long K / tiny N);
//   * activations never touch LDS: lane (G, row) loads the 16 bytes k = 8 G .. 8 G + 7 of its row for the 32-k step
//     (one buffer_load_dwordx4 per 16 rows and k-step) and the 4 x 4 (register, lane group) transposition that turns
//     them into the lane's MFMA fragment {dword g, g + 4, g + 8, g + 12} is two v_permlane32_swap + two
//     v_permlane16_swap - no barrier anywhere in the main loop;
//   * every vector-memory op is a compiler-visible buffer intrinsic issued in one fixed order (weights, activations
//     and scale rows of a 128-k unit two units ahead), so hipcc's counted vmcnt keeps the whole ring in flight;
//   * group scales are applied to the fp32 group accumulators (16 MT FMAs per 128 k), the K slices are summed through
//     LDS with one barrier, 256 MT threads each adding NW float4s.
// grid (N / 64, k_splits, row blocks of 16 MT), block 64 NW. Weight / activation / fragment conventions are those of
// marlin_gemm_kernel.
__device__ __forceinline__ u32x4 frag_transpose(u32x4 v) {
  // (register e, lane group G) -> (register G, lane group e): e = 2 e1 + e0, G = 2 G1 + G0
  const auto s02 = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false);  // e1 <-> G1
  const auto s13 = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
  const auto p01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);  // e0 <-> G0
  const auto p23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
  return u32x4{p01[0], p01[1], p23[0], p23[1]};
}

template <typename scalar_t, int MT, int NW, bool GROUPED, bool WS, bool NORM = false, bool ATTN = false>
__global__ __launch_bounds__(64 * NW, (NW >= 16) ? 4 : 2) void marlin_decode_kernel(const GemmParams p) {
  static_assert(!(NORM || ATTN) || MT == 1, "the forms with a computed A operand take one 16-row tile");
  static_assert(!(NORM && ATTN), "one computed A operand at a time");
  constexpr bool LDSA = NORM || ATTN;  // the A operand is computed in the prologue and read from LDS
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15, c8 = li & 7, hi = li >> 3;
  const int N = p.N, K = p.K, M = p.M;
  // Fused silu_and_mul (p.act_out; host: no K split, N % 128 == 0, grid.x = N / 128): the lower half of the waves slices K
  // of gate column group blockIdx.x, the upper half of the up group N / 2 further right - the K slicing and the order of
  // the sums are those of the plain launch with NW / 2 waves, so the fused op is bit-identical to GEMM + silu_and_mul.
  const bool fuse_act = p.act_out != nullptr;
  const int nslice = fuse_act ? NW / 2 : NW;
  const int n0 = blockIdx.x * 64 + ((fuse_act && wave >= NW / 2) ? N / 2 : 0);
  // row blocks (M > 16 MT) re-read the weights: gridDim.x is a multiple of 8 for every Marlin shape of interest, so
  // the row blocks of one column group have equal workgroup id % 8 = the same XCD and the later one hits its L2
  const int m0 = blockIdx.z * (16 * MT);

  // K slice of this wave, in units of 128 k
  const int total_units = K / 128;
  const int workers = p.k_splits * nslice;
  const int per = (total_units + workers - 1) / workers;
  const int worker = blockIdx.y * nslice + (fuse_act ? wave % (NW / 2) : wave);
  const int u0 = min(worker * per, total_units), u1 = min(u0 + per, total_units);

  const int row_bytes = N * 8;  // one k-tile row of the packed tensor
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.b), 0, (K / 16) * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, M * K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.scales), 0, p.num_groups * N * 2, 0x00020000);
  // weights: ONE 16-byte load per lane and 32-k step - the whole chunk 4 c8 + g (all four 16-column sub-tiles) of
  // k-tile 2 ks + hi; the half the lane does not use is traded with lane ^ 8 (which holds the other k-tile) in
  // split_pair() below. A wave instruction then moves 1 KiB instead of 512 B: the CU's texture-address path retires
  // roughly one vector-memory wave instruction per ~38 cycles whatever its width (measured: tools/stream_probe.hip,
  // 16-byte vs 8-byte Marlin pattern), so bytes per instruction is what sets a CU's ingest rate.
  const int b_voff = ((n0 / 64) * 128 + (4 * c8 + g) * 4) * 4 + hi * row_bytes;
  int a_voff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + mt * 16 + li;
    a_voff[mt] = (m < M) ? (m * K + 8 * g) * 2 : 0x7ff00000;  // beyond the descriptor: reads as zero
  }
  // group scales. WS = false: applied to the fp32 group accumulators - D row r of tile t is column
  // 32 (g>>1) + 8 t + 4 (g&1) + r, stored (scale_perm) at position 8 (4 (g&1) + r) + 4 (g>>1) + t -> one 8-byte load
  // per r (fewest VALU ops, 4 vector-memory instructions per unit). WS = true: folded into the dequantised weights
  // like the reference does - the lane's four tile columns c8 + 8 t + 32 hi sit at positions 8 c8 + 4 hi + t -> ONE
  // 8-byte load per unit and no group accumulators (16 more packed multiplies per 32 k; the only form that fits the
  // 128 registers of the 16-wave shape).
  constexpr bool WSCALE = GROUPED && WS;
  constexpr int NS = !GROUPED ? 0 : (WSCALE ? 1 : 4);
  const int s_voff = WSCALE ? (n0 + 8 * c8 + 4 * hi) * 2 : (n0 + 32 * (g & 1) + 4 * (g >> 1)) * 2;

  struct Unit {
    u32x4 q[4];
    u32x4 a[4][MT];
    u32x2 s[NS > 0 ? NS : 1];
  };
  // loads of one 32-k step / of the scale rows of unit u (past the slice: the last unit again, never consumed)
  // NORM: the normalised rows live in LDS behind the reduction image - row r at r * XS (XS = 2 K + 64: four rows land in
  // different 64-byte bank quarters), row M is all zeros and serves the tile's unused rows
  // ATTN: a row holds only the workgroup's K slice (units abase .. abase + nslice * per), every wave fills and reads its own part
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int abase = ATTN ? (int)blockIdx.y * nslice * per : 0;  // first unit of the LDS rows
  const int XS = ATTN ? nslice * per * 256 + 64 : 2 * K + 64;
  char* const xl = smem + NW * MT * 4096;
  const char* const xl_lane = xl + min(li, M) * XS + 16 * g;
  // WQ / WA: the weight-side (vector-memory) and the activation-side half of a step; the NORM prologue issues them apart
  auto load_step = [&](int u, int ks, Unit& U, bool WQ = true, bool WA = true) {
    u = min(u, total_units - 1);
    // aux 2 = non-temporal: the weights are read once (this kernel is used with one row block; a constant, because a
    // branch around the load halves the wait counts hipcc can prove)
    if (WQ) U.q[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff, (u * 8 + 2 * ks) * row_bytes, NMX_W_NT ? 2 : 0);
    if constexpr (LDSA) {
      // (ATTN: past the wave's slice the last own unit again - the row holds nothing beyond the workgroup's slice)
      const int ua_ = ATTN ? min(u, max(u1 - 1, u0)) - abase : u;
      if (WA) U.a[ks][0] = *reinterpret_cast<const u32x4*>(xl_lane + (ua_ * 128 + ks * 32) * 2);
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        U.a[ks][mt] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff[mt], (u * 128 + ks * 32) * 2, 0);
    }
    // pin the issue order: the prologue must queue the loads exactly as the loop does, or the (merged) wait counts
    // at the loop head degrade to those of the worse of the two orders
    __builtin_amdgcn_sched_barrier(0);
  };
  auto load_scales = [&](int u, Unit& U) {
    if constexpr (GROUPED) {
      u = min(u, total_units - 1);
      const int grp = (u * 128) / p.group_size;
#pragma unroll
      for (int r = 0; r < NS; ++r) U.s[r] = __builtin_amdgcn_raw_buffer_load_b64(rs_s, s_voff + 16 * r, grp * N * 2, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto load_unit = [&](int u, Unit& U, bool WQ = true, bool WA = true) {  // same order as compute_unit re-issues them
    if constexpr (WSCALE) { if (WQ) load_scales(u, U); }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) load_step(u, ks, U, WQ, WA);
    if constexpr (!WSCALE) { if (WQ) load_scales(u, U); }
  };

  f32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // consumes unit U and re-issues each of its register sets for unit `next` as soon as it is free
  auto compute_unit = [&](Unit& U, int next, float keep) {
    f32x4 gacc[WSCALE ? 1 : MT][4];
    uint32_t s2[4] = {0, 0, 0, 0};
    if constexpr (WSCALE) {
      union { u32x2 v; scalar_t e[4]; } raw;
      raw.v = U.s[0];
      load_scales(next, U);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (__is_same(scalar_t, f16)) {
          union { f16 h[2]; uint32_t u; } pk;
          pk.h[0] = raw.e[t];
          pk.h[1] = raw.e[t];
          s2[t] = keep != 0.f ? pk.u : 0u;
        } else {
          s2[t] = __builtin_bit_cast(uint32_t, (float)raw.e[t] * keep);
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 af[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[mt] = frag_transpose(U.a[ks][mt]);
      u32x2 q0, q1;  // words 2 hi, 2 hi + 1 of k-tiles 2 ks and 2 ks + 1
      split_pair(U.q[ks], hi != 0, q0, q1);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t w0 = q0[t >> 1] >> (8 * (t & 1)), w1 = q1[t >> 1] >> (8 * (t & 1));
        uint32_t d0, d1, d2, d3;
        Dequant<scalar_t, W_INT4>::run(w0, s2[t], WSCALE, d0, d1);
        Dequant<scalar_t, W_INT4>::run(w1, s2[t], WSCALE, d2, d3);
        const u32x4 wf = {d0, d1, d2, d3};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if constexpr (WSCALE) acc[mt][t] = mfma_16x16x32<scalar_t>(wf, af[mt], acc[mt][t]);
          else gacc[mt][t] = mfma_16x16x32<scalar_t>(wf, af[mt], ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : gacc[mt][t]);
        }
      }
      load_step(next, ks, U);
    }
    if constexpr (WSCALE) {
    } else if constexpr (GROUPED) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        union { u32x2 v; scalar_t e[4]; } raw;
        raw.v = U.s[r];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float sv = Scalar<scalar_t>::to_f32(raw.e[t]) * keep;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt][t][r] = __builtin_fmaf(sv, gacc[mt][t][r], acc[mt][t][r]);
        }
      }
      load_scales(next, U);
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mt][t][r] = __builtin_fmaf(gacc[mt][t][r], keep, acc[mt][t][r]);
    }
  };

  Unit ua, ub;
  if constexpr (NORM) {
    // ---- the A operand: fused_add_rms_norm of the producer GEMM's K-split slabs, the arithmetic of rms_norm_splitk_kernel
    //      (elementwise.hip) thread for thread - its NT threads sum the squares of the same elements in the same order ----
    __shared__ float nsm[17];
    const int nvec = K / 8, NT = norm_threads(K);
    const int64_t slab = (int64_t)M * K;
    const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
    const scalar_t* res_in = reinterpret_cast<const scalar_t*>(p.norm_res_in);
    scalar_t* res_out = reinterpret_cast<scalar_t*>(p.norm_res_out);
    union V { u32x4 u; scalar_t e[8]; };
    V x[2], wv[2];
    const int tid = threadIdx.x;
    auto nload = [&](int m) {  // x = round(sum of the slabs) + residual, rounded: row m, this thread's (up to) two vectors
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int v = tid + k * NT;
        if (tid < NT && v < nvec) {
          V r;  // (requested before the slab sum consumes its loads)
          r.u = *reinterpret_cast<const u32x4*>(res_in + (int64_t)m * K + v * 8);
          sum_partials8<scalar_t>(p.norm_partial, p.norm_splits, slab, (int64_t)m * K + v * 8, x[k].e);
#pragma unroll
          for (int j = 0; j < 8; ++j) x[k].e[j] = Scalar<scalar_t>::from_f32(Scalar<scalar_t>::to_f32(x[k].e[j]) + Scalar<scalar_t>::to_f32(r.e[j]));
        }
      }
    };
    auto nmath = [&](int m) {
      float var = 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int v = tid + k * NT;
        if (tid < NT && v < nvec) {
          if (writer) *reinterpret_cast<u32x4*>(res_out + (int64_t)m * K + v * 8) = x[k].u;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float f = Scalar<scalar_t>::to_f32(x[k].e[j]);
            var += f * f;
          }
        }
      }
      var = block_sum(var, nsm);
      const float sc = rsqrtf(var / (float)K + p.norm_eps);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int v = tid + k * NT;
        if (tid < NT && v < nvec) {
          V o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o.e[j] = rnd_mul<scalar_t>(Scalar<scalar_t>::from_f32(Scalar<scalar_t>::to_f32(x[k].e[j]) * sc), wv[k].e[j]);
          *reinterpret_cast<u32x4*>(xl + m * XS + v * 16) = o.u;
        }
      }
    };
    // the weight stream of the first two units goes out FIRST (it does not depend on the norm): the norm's own loads queue
    // behind it and one wait covers both - issued the other way round, the sum of the slabs (which consumes its loads at
    // once) put a whole memory round trip in front of the first weight request
    if (u0 < u1) {
      load_unit(u0, ua, true, false);
      load_unit(u0 + 1, ub, true, false);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int v = tid + k * NT;
      if (tid < NT && v < nvec) wv[k].u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const scalar_t*>(p.norm_weight) + v * 8);
    }
    nload(0);
    for (int e = tid; e < (2 * K) / 16; e += 64 * NW) *reinterpret_cast<u32x4*>(xl + M * XS + e * 16) = u32x4{0, 0, 0, 0};
    nmath(0);
    for (int m = 1; m < M; ++m) {
      nload(m);
      nmath(m);
    }
    __syncthreads();
    if (u0 < u1) {
      load_unit(u0, ua, false, true);
      load_unit(u0 + 1, ub, false, true);
    }
  }
  if constexpr (ATTN) {
    // ---- the A operand: paged attention's v2 reduce (v2_reduce_head, the device function of the reduce kernel: same bits) of
    //      the heads this wave's K slice covers - unit u = head u (head size 128), row m = sequence m. Wave-private: a wave
    //      reads back only what it wrote; the one barrier publishes the zero row. ----
    if (u0 < u1) {
      load_unit(u0, ua, true, false);
      load_unit(u0 + 1, ub, true, false);
    }
    for (int e = threadIdx.x; e < XS / 16; e += 64 * NW) *reinterpret_cast<u32x4*>(xl + M * XS + e * 16) = u32x4{0, 0, 0, 0};
    // lane c of the wave's (u1 - u0) x 32 chunks: head u0 + c / 32, dimensions 4 (c % 32) .. + 3; rows four at a time, every load
    // of the four rows in flight before the first is consumed (v2_vec4_*: at most 8 partitions, the host checks)
    const int nchunks = max(u1 - u0, 0) * 32;
    for (int c = lane; c < nchunks; c += 64) {
      const int u = u0 + (c >> 5), d0 = 4 * (c & 31);
      for (int mb = 0; mb < M; mb += 4) {
        V2Vec4<scalar_t> r[4];
        int npv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = min(mb + i, M - 1);
          npv[i] = (p.attn_seq_lens[m] + p.attn_part_size - 1) / p.attn_part_size;
          const int64_t pb = ((int64_t)m * p.attn_heads + u) * p.attn_max_parts;
          v2_vec4_load<scalar_t>(r[i], p.attn_exp_sums + pb, p.attn_max_logits + pb, reinterpret_cast<const scalar_t*>(p.attn_tmp) + pb * 128,
                                 npv[i], 128, d0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (mb + i < M) *reinterpret_cast<u32x2*>(xl + (mb + i) * XS + (u - abase) * 256 + d0 * 2) = v2_vec4_math<scalar_t>(r[i], npv[i]);
      }
    }
    __syncthreads();
    if (u0 < u1) {
      load_unit(u0, ua, false, true);
      load_unit(u0 + 1, ub, false, true);
    }
  }
  if (u0 < u1) {
    if constexpr (!LDSA) {
      load_unit(u0, ua);
      load_unit(u0 + 1, ub);
    }
    // always whole pairs, branch-free: an odd last unit computes on the (reloaded) final unit and is dropped through
    // its weight 0. A conditional second half would make the wait counts at the loop head cover the path that skipped
    // it (half the ring); a peeled tail costs a third copy of the body and its registers.
    for (int u = u0; u < u1; u += 2) {
      compute_unit(ua, u + 2, 1.0f);
      compute_unit(ub, u + 3, (u + 1 < u1) ? 1.0f : 0.0f);
    }
  }

  // ---- sum the NW K slices through LDS: thread e < 256 MT owns float4 e of the workgroup's [MT][4][64] tile image ----
  f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) red[(wave * MT * 4 + mt * 4 + t) * 64 + lane] = acc[mt][t];
  __syncthreads();
  if (fuse_act) {
    const int gate0 = blockIdx.x * 64;
    for (int e = threadIdx.x; e < 256 * MT; e += 64 * NW) {
      f32x4 sg = red[e], su = red[(NW / 2) * MT * 256 + e];
#pragma unroll
      for (int w = 1; w < NW / 2; ++w) {
        sg += red[w * MT * 256 + e];
        su += red[(NW / 2 + w) * MT * 256 + e];
      }
      const int mt = e >> 8, t = (e >> 6) & 3, el = e & 63, eg = el >> 4;
      const int m = m0 + mt * 16 + (el & 15);
      const int col = 32 * (eg >> 1) + 8 * t + 4 * (eg & 1);
      if (m >= M) continue;
      if constexpr (!GROUPED) {
        const scalar_t* sc = reinterpret_cast<const scalar_t*>(p.scales) + gate0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = col + r, cc = c & 7, b = c >> 3;
          const int pos = 32 * (b >> 2) + 8 * (cc >> 1) + (cc & 1) + 2 * (b & 3);
          sg[r] *= Scalar<scalar_t>::to_f32(sc[pos]);
          su[r] *= Scalar<scalar_t>::to_f32(sc[N / 2 + pos]);
          asm volatile("" : "+v"(sg[r]), "+v"(su[r]));  // see the plain epilogue below
        }
      }
      union { scalar_t h[4]; u32x2 u; } o;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        o.h[r] = rnd_mul<scalar_t>(silu_rnd<scalar_t>(Scalar<scalar_t>::from_f32(sg[r])), Scalar<scalar_t>::from_f32(su[r]));
      *reinterpret_cast<u32x2*>(reinterpret_cast<scalar_t*>(p.act_out) + (int64_t)m * (N / 2) + gate0 + col) = o.u;
    }
    return;
  }
  for (int e = threadIdx.x; e < 256 * MT; e += 64 * NW) {
    f32x4 sum = red[e];
#pragma unroll
    for (int w = 1; w < NW; ++w) sum += red[w * MT * 256 + e];
    const int mt = e >> 8, t = (e >> 6) & 3, el = e & 63, eg = el >> 4;
    const int m = m0 + mt * 16 + (el & 15);
    const int col = 32 * (eg >> 1) + 8 * t + 4 * (eg & 1);  // + r, within the 64-column group
    if (m >= M) continue;
    if constexpr (!GROUPED) {
      // channel-wise scales (every K split scales its own partial); scale_perm_single: position
      // 32 (b >> 2) + 8 (cc >> 1) + (cc & 1) + 2 (b & 3) holds column cc + 8 b
      const scalar_t* sc = reinterpret_cast<const scalar_t*>(p.scales) + n0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = col + r, cc = c & 7, b = c >> 3;
        sum[r] *= Scalar<scalar_t>::to_f32(sc[32 * (b >> 2) + 8 * (cc >> 1) + (cc & 1) + 2 * (b & 3)]);
        // keep the fp32 product a value of its own: hipcc otherwise fuses this multiply with the fp16 conversion below
        // (v_fma_mixlo_f16, ONE rounding) here but not in the fused epilogue above, where the rounded value is used again
        // as a float - the two forms of the op must round alike
        asm volatile("" : "+v"(sum[r]));
      }
    }
    if (p.k_splits == 1) {
      union { scalar_t h[4]; u32x2 u; } o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o.h[r] = Scalar<scalar_t>::from_f32(sum[r]);
      *reinterpret_cast<u32x2*>(reinterpret_cast<scalar_t*>(p.c) + (int64_t)m * N + n0 + col) = o.u;
    } else {
      *reinterpret_cast<f32x4*>(p.partial + ((int64_t)blockIdx.y * M + m) * N + n0 + col) = sum;
    }
  }
}

// out[m][n] = cast(sum_s partial[s][m][n]); 4 columns per thread
template <typename scalar_t>
__global__ void splitk_reduce_kernel(scalar_t* __restrict__ c, const float* __restrict__ partial, int64_t mn4, int splits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mn4) return;
  // four slabs in flight per thread (a one-load-per-iteration loop pays a full L2 / MALL round trip per split);
  // summation order stays s = 0, 1, 2, ... so the result does not depend on the batching
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (splits & NMX_SPLITK_F16) {  // fp16 slabs (the M > 64 kernels): 8 bytes = the 4 elements, same order of summation
    const f16* ph = reinterpret_cast<const f16*>(partial);
    const int ns = NMX_SPLITK_COUNT(splits);
    for (int q = 0; q < ns; ++q) {
      union { u32x2 u; f16 h[4]; } v;
      v.u = *reinterpret_cast<const u32x2*>(ph + ((int64_t)q * mn4 + i) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += (float)v.h[j];
    }
    union { scalar_t h[4]; u32x2 u; } r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r.h[j] = Scalar<scalar_t>::from_f32(acc[j]);
    *reinterpret_cast<u32x2*>(c + i * 4) = r.u;
    return;
  }
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
  union { scalar_t h[4]; u32x2 u; } r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r.h[j] = Scalar<scalar_t>::from_f32(acc[j]);
  *reinterpret_cast<u32x2*>(c + i * 4) = r.u;
}

// ---- GPTQ -> Marlin repack (gptq_marlin_repack.cu:32-260; element map SURVEY.md appendix A.2) ----------------
// one thread per output int32
template <int BITS>
__global__ void marlin_repack_kernel(const uint32_t* __restrict__ in, const int32_t* __restrict__ perm,
                                     uint32_t* __restrict__ out, int size_k, int size_n) {
  constexpr int PF = 32 / BITS;
  constexpr int WORDS64 = 1024 / PF;  // words per (k-tile, 64-column group)
  const int64_t row_words = (int64_t)size_n * 16 / PF;
  const int64_t total = (int64_t)(size_k / 16) * row_words;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int kt = idx / row_words;
  const int w = idx % row_words;
  const int ng = w / WORDS64;
  const int wi = w % WORDS64;
  int i, j, part;
  if constexpr (BITS == 4) { i = wi >> 2; j = wi & 3; part = 0; }
  else { i = wi >> 3; j = (wi >> 1) & 3; part = wi & 1; }
  const int col = i >> 2;
  const int row0 = 2 * (i & 3);
  const int rows[4] = {row0, row0 + 1, row0 + 8, row0 + 9};
  auto fetch = [&](int e) -> uint32_t {  // element e of v[0..7]
    int k = kt * 16 + rows[e & 3];
    const int n = ng * 64 + 16 * j + col + 8 * (e >> 2);
    if (perm != nullptr) k = perm[k];
    const uint32_t word = in[(int64_t)(k / PF) * size_n + n];
    return (word >> (BITS * (k % PF))) & ((1u << BITS) - 1);
  };
  uint32_t r = 0;
  if constexpr (BITS == 4) {
    const int il[8] = {0, 2, 4, 6, 1, 3, 5, 7};
#pragma unroll
    for (int pz = 0; pz < 8; ++pz) r |= fetch(il[pz]) << (4 * pz);
  } else {
    const int il[4] = {0, 2, 1, 3};
#pragma unroll
    for (int pz = 0; pz < 4; ++pz) r |= fetch(4 * part + il[pz]) << (8 * pz);
  }
  out[idx] = r;
}

struct GemmCfg { int mt, ng, splits, w8; };

// Tile shape and K splits, fitted to MI355X sweeps (tools/gemm_sweep.py, tools/check_cfg.py; Llama-3-8B and 70B/TP8
// shapes). What the measurements say:
//  * every wave is instruction-issue bound (dequant VALU + MFMA issue, ~450 cycles per 32-k step at MT = 4), so time
//    ~ k-steps per wave once every CU has work; a dependent split-K reduce launch costs ~4-5 us, a workgroup's
//    prologue + epilogue ~5 us;
//  * M <= 16: 4 waves split K inside the workgroup; split across workgroups only when a wave would run more than
//    ~32 k-steps or fewer than ~128 workgroups exist;
//  * 16 < M <= 64 and N < 16 K: 32-row tiles (two row blocks re-read the weights through L2) beat one 64-row tile -
//    twice the workgroups, half the accumulators, fp32 group scaling;
//  * otherwise 64-row x 256-column tiles with 8 waves (2 K slices) per workgroup.
GemmCfg pick_cfg(int M, int N, int K) {
  GemmCfg c;
  c.w8 = 0;
  const int n64 = N / 64;
  if (M <= 16) { c.mt = 1; c.ng = 1; }
  else if (M <= 32 || (M <= 64 && n64 < 256)) { c.mt = 2; c.ng = 2; }
  else { c.mt = 4; c.ng = 4; c.w8 = 1; }
  // one 64-row block and >= 256 column groups: 128-column tiles (8 waves = 2 column groups x 4 K slices) fill the chip
  // without any cross-workgroup K split - no fp32 partials, no reduce launch (gate_up at M = 64: 29.8 us vs 30.5-33.6)
  if (c.mt == 4 && M <= 64 && n64 >= 448 && K <= 8192) c.ng = 2;
  if (const char* e = nmx_tune(NMX_TUNE_GEMM_CFG)) {  // tuning override: "mt,ng,splits[,w8]"
    int a = 0, b = 0, s = 0, w = 0;
    const int got = sscanf(e, "%d,%d,%d,%d", &a, &b, &s, &w);
    const bool w8 = got == 4 && w != 0 && a == 4;
    if (got >= 3 && (a == 1 || a == 2 || a == 4) && (b == 1 || b == 2 || b == 4) && s >= 1 &&
        !(a == 4 && b == 1) && !(a == 2 && b == 1)) {
      c.mt = a; c.ng = b; c.splits = s;
      c.w8 = w8 ? 1 : 0;
      return c;
    }
  }
  if (c.w8 && c.ng == 4) {
    // between one and two 8-wave workgroups per CU (gate_up at M = 256: 448) the second round runs three quarters
    // empty; 4-wave workgroups are all resident at two per CU instead (79.3 vs 83.8 us)
    const int u8 = ceil_div(N, 256) * ceil_div(M, 64);
    if (u8 > 256 && u8 <= 512) c.w8 = 0;
  }
  const int kw = (c.w8 ? 8 : 4) / c.ng;
  const int total_steps = ceil_div(K, 32);
  const int total_sub = ceil_div(total_steps, 4);
  const int units = ceil_div(N, 64 * c.ng) * ceil_div(M, 16 * c.mt);
  auto friendly = [&](int s) { return s == 1 || total_sub % (s * kw) == 0; };  // equal K slices
  auto ks = [&](int s) { return total_steps / (s * kw); };                      // k-steps per wave
  int sp = 1;
  if (c.mt == 1) {
    int want = std::max(1, (ks(1) + 16) / 32);
    if (units < 64) {
      for (int s = 1; s <= 16; ++s)
        if (friendly(s) && units * s >= 128 && ks(s) >= 8) { want = std::max(want, s); break; }
    }
    for (int s = 1; s <= std::min(want, 16); ++s)
      if (friendly(s)) sp = s;
  } else if (units < 200) {
    const int cap = c.w8 ? 256 : 512;  // workgroups resident at once
    for (int s = 1; s <= 8; ++s)
      if (friendly(s) && ks(s) >= 16 && units * s <= cap) sp = s;
    if (units * sp < 192) {
      for (int s = sp + 1; s <= 16; ++s)
        if (friendly(s) && ks(s) >= 8 && units * s >= 256) { if (units * s <= cap) sp = s; break; }
    }
    if (c.mt == 2) {
      // 32-row tiles: the fewest splits that give >= 192 workgroups with <= 32 k-steps per wave (qkv at M = 64: 2
      // splits 16.8 us, 4 splits 18.0; down keeps 8)
      for (int s = 1; s <= 16; s *= 2)
        if (friendly(s) && units * s >= 192 && ks(s) <= 32 && ks(s) >= 8 && units * s <= cap) { sp = s; break; }
    }
  }
  c.splits = sp;
  return c;
}

template <typename scalar_t, int KIND, int MT, int NG, int MODE, bool SP, bool W8 = false, bool ZP = false>
int launch_cfg(const GemmParams& p, hipStream_t stream) {
  constexpr int SUB = (KIND == W_INT4) ? 4 : 2;
  constexpr int KW = (W8 ? 8 : 4) / NG;
  const size_t stage = (size_t)KW * 2 * SUB * 4 * (16 * MT) * 16;
  const size_t red = (KW > 1) ? (size_t)(KW / 2) * NG * MT * 4 * 64 * 4 * sizeof(float) : 0;
  const size_t smem = std::max(stage, red);
  // x = (column tiles rounded up to the 8 XCDs) x row blocks, see the kernel's blockIdx decoding
  dim3 grid(ceil_div(ceil_div(p.N, 64 * NG), 8) * 8 * ceil_div(p.M, 16 * MT), p.k_splits, 1);
  auto kern = marlin_gemm_kernel<scalar_t, KIND, MT, NG, MODE, SP, W8, ZP>;
  if (smem > 64 * 1024)
    NMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  kern<<<grid, W8 ? 512 : 256, smem, stream>>>(p);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

template <typename scalar_t, int KIND, int MODE, bool SP, bool ZP = false>
int launch_mode(const GemmParams& p, const GemmCfg& cfg, hipStream_t stream) {
  if (cfg.w8 && cfg.ng == 2) return launch_cfg<scalar_t, KIND, 4, 2, MODE, SP, true, ZP>(p, stream);
  if (cfg.w8) return launch_cfg<scalar_t, KIND, 4, 4, MODE, SP, true, ZP>(p, stream);
  if (cfg.mt == 1 && cfg.ng == 1) return launch_cfg<scalar_t, KIND, 1, 1, MODE, SP, false, ZP>(p, stream);
  if (cfg.mt == 1 && cfg.ng == 2) return launch_cfg<scalar_t, KIND, 1, 2, MODE, SP, false, ZP>(p, stream);
  if (cfg.mt == 1) return launch_cfg<scalar_t, KIND, 1, 4, MODE, SP, false, ZP>(p, stream);
  if (cfg.mt == 2 && cfg.ng == 4) return launch_cfg<scalar_t, KIND, 2, 4, MODE, SP, false, ZP>(p, stream);
  if (cfg.mt == 2) return launch_cfg<scalar_t, KIND, 2, 2, MODE, SP, false, ZP>(p, stream);
  if (cfg.mt == 4 && cfg.ng == 2) return launch_cfg<scalar_t, KIND, 4, 2, MODE, SP, false, ZP>(p, stream);
  return launch_cfg<scalar_t, KIND, 4, 4, MODE, SP, false, ZP>(p, stream);
}

// K splits of the large-M kernel: enough workgroups for the 256 CUs, stages of 64 k
inline int large_splits(int M, int N, int K) {
  const int units = ceil_div(N, 256) * ceil_div(M, 256), stages = K / 64;
  int sp = 1;
  while (units * sp < 192 && sp * 2 <= 8 && stages / (sp * 2) >= 8) sp *= 2;
  return sp;
}
inline bool use_large(const GemmParams& p, bool sp24) {
  if (sp24 || p.M <= 128 || p.perm != nullptr || p.slow_act_order || p.K % 64 != 0 || p.N % 64 != 0) return false;
  if (p.num_groups > 1 && p.group_size % 64 != 0) return false;
  if ((int64_t)p.M * p.K * 2 >= (1ll << 31) || (int64_t)p.K * p.N >= (1ll << 31)) return false;
  if (const char* e = nmx_tune(NMX_TUNE_GEMM_LARGE)) return atoi(e) != 0;
  // measured (bench.py --sweep): the 256 x 256 tiles win once they alone fill the chip (>= 192 workgroups without K
  // splits: gate_up from M = 512, qkv at M = 2048); narrower matrices stay on the 64-row row-block path
  return ceil_div(p.N, 256) * ceil_div(p.M, 256) >= 192;
}

template <typename scalar_t, int KIND>
int launch_large(GemmParams& p, void* scratch, int64_t scratch_bytes, hipStream_t stream) {
  p.k_splits = large_splits(p.M, p.N, p.K);
  if (p.k_splits > 1) {
    const int64_t per = (int64_t)p.M * p.N * sizeof(float);
    const int fit = scratch == nullptr ? 1 : (int)std::min<int64_t>(8, scratch_bytes / per);
    while (p.k_splits > std::max(1, fit)) p.k_splits /= 2;
  }
  p.partial = reinterpret_cast<float*>(scratch);
  int ngrp = 4;  // measured: 0.75-0.90 PFLOP/s at M = 2048 vs 0.65-0.84 for the two-workgroups-per-CU shape
  if (const char* e = nmx_tune(NMX_TUNE_GEMM_LARGE_NGRP)) ngrp = atoi(e) == 2 ? 2 : 4;
  const size_t smem = (ngrp == 4 ? 2 : 1) * (size_t)(2 * ngrp * 4 * 64 * 16 + 2 * 4 * 256 * 16);
  dim3 grid(ceil_div(p.N, 64 * ngrp), p.k_splits, ceil_div(p.M, 256));
#define NMX_LAUNCH_LARGE(MODE_, NGRP_)                                                                              \
  {                                                                                                                 \
    auto kern = marlin_large_kernel<scalar_t, KIND, MODE_, NGRP_>;                                                   \
    NMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                (int)smem));                                                                        \
    kern<<<grid, 128 * NGRP_, smem, stream>>>(p);                                                                    \
  }
  if (p.num_groups > 1) {
    if (ngrp == 4) NMX_LAUNCH_LARGE(1, 4) else NMX_LAUNCH_LARGE(1, 2)
  } else {
    if (ngrp == 4) NMX_LAUNCH_LARGE(0, 4) else NMX_LAUNCH_LARGE(0, 2)
  }
#undef NMX_LAUNCH_LARGE
  NMX_LAUNCH_CHECK();
  if (p.k_splits > 1 && !p.defer_reduce) {
    const int64_t mn4 = (int64_t)p.M * p.N / 4;
    splitk_reduce_kernel<scalar_t><<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>(
        reinterpret_cast<scalar_t*>(p.c), p.partial, mn4, p.k_splits);
    NMX_LAUNCH_CHECK();
  }
  return NMX_OK;
}

// ---- decode kernel dispatch ---------------------------------------------------------------------------------------
struct DecodeCfg { int mt, nw, splits, ws; };  // nw = 0: not applicable; ws: group scales folded into the weights

// Where marlin_decode_kernel applies and how it is launched. NMX_GEMM_LEAN = "nw,splits[,mt[,ws]]" forces a configuration
// (nw = 0 disables the kernel) for sweeps.
inline DecodeCfg pick_decode_cfg(int M, int N, int K) {
  DecodeCfg c{0, 0, 1, 1};
  if (M > 64 || K % 128 != 0 || N % 64 != 0) return c;
  c.mt = M <= 16 ? 1 : 2;
  const int units = K / 128, groups = N / 64;
  // Measured (tools/lean_sweep.py, 32-launch graph chains, Llama-3-8B shapes): a CU retires ~one vector-memory wave
  // instruction per 38 cycles, so what matters is the number of CUs streaming, not the waves per CU - 4-wave
  // workgroups, K split across workgroups until ~256 of them exist, and the split-K reduce launch is cheaper than
  // leaving CUs idle. The kernel wins where few column groups exist and the rows fit one tile (activations are read
  // per wave, so their instruction count grows with M): o / qkv at M <= 16 (7.2 / 8.7 us vs 10.2 / 10.7 at M = 1),
  // o at M <= 32. Long K (down), wide N (gate_up) and M > 32 stay on marlin_gemm_kernel.
  const bool wins = K <= 8192 && ((M <= 16 && groups <= 128) || (M <= 32 && groups <= 64));
  if (wins) {
    c.nw = 4;
    while (groups * c.splits * 2 <= 256 && units % (c.nw * c.splits * 2) == 0 && units / (c.nw * c.splits * 2) >= 2)
      c.splits *= 2;
  } else if (K <= 8192 && M <= 8 && groups >= 256) {
    // wide N at batch <= 8 (gate_up): one 4-wave workgroup per column group, no split: 16.7 vs 18.6 us at M = 1,
    // 17.9 vs 18.6 at M = 8 (at M = 16 the per-wave activation loads make it lose: 21.0 vs 20.7)
    c.nw = 4;
  }
  if (const char* e = nmx_tune(NMX_TUNE_GEMM_LEAN)) {
    int nw = 0, sp = 1, mt = 0, ws = 1;
    const int got = sscanf(e, "%d,%d,%d,%d", &nw, &sp, &mt, &ws);
    if (got >= 1 && (nw == 0 || nw == 4 || nw == 8 || nw == 16)) {
      c.nw = nw;
      c.splits = (got >= 2 && sp >= 1 && sp <= 32) ? sp : 1;
      if (got >= 3 && (mt == 1 || mt == 2)) c.mt = mt;
      if (got >= 4) c.ws = ws != 0;
    }
  }
  return c;
}

// Shapes the norm-fused form (marlin_decode_kernel<NORM>) serves: fp16 / bf16, plain int4 layout (the caller knows), at most
// kNormMaxRows rows, the decode kernel's 4-wave shape (with_act: its unsplit gate | up form), and a hidden size whose
// rms_norm_splitk_kernel thread count fits the workgroup (bit identity with the unfused sequence)
inline bool decode_norm_supported(int M, int N, int K, int num_groups, bool with_act) {
  // Rows served by default: ONE. Every workgroup pulls a row's slabs + residual through its CU (80 - 150 KB at ~64 B/clk:
  // 0.6 - 1.1 us per row), which the saved launch (~3.7 us) pays for once, not two to four times - measured on the decode
  // step: batch 1 2.31 -> 2.19 ms, batch 2 2.34 -> 2.42, batch 4 2.38 -> 2.84 (gpurun_out/norm_ab.log). NMX_GEMM_NORM_ROWS
  // raises the limit (tests: the multi-row prologue stays covered).
  int max_rows = 1;
  if (const char* e = nmx_tune(NMX_TUNE_GEMM_NORM_ROWS)) max_rows = std::min(std::max(atoi(e), 0), kNormMaxRows);
  if (M < 1 || M > max_rows || K % 128 != 0 || N % 64 != 0 || K / 8 > 2 * 1024) return false;
  if (!(num_groups == 1 || (K / num_groups) % 128 == 0)) return false;
  const DecodeCfg c = pick_decode_cfg(M, N, K);
  if (c.nw != 4 || c.mt != 1) return false;
  const int nt = norm_threads(K);
  if ((K / 8 + nt - 1) / nt > 2) return false;
  if (with_act) return c.splits == 1 && N % 128 == 0 && nt <= 512;
  return nt <= 256;
}

// LDS bytes of the computed-A forms behind the reduction image (marlin_decode_kernel: rows of XS bytes + one zero row; ATTN: the
// workgroup's K slice only + the waves' rescale scratch)
inline size_t decode_lds_a_bytes(const GemmParams& p, int nw, bool norm, bool attn) {
  if (norm) return (size_t)(p.M + 1) * (2 * p.K + 64);
  if (attn) {
    const int units = p.K / 128, workers = p.k_splits * nw, per = (units + workers - 1) / workers;
    return (size_t)(p.M + 1) * (nw * per * 256 + 64);
  }
  return 0;
}

// Shapes the attention-reduce form (marlin_decode_kernel<ATTN>) serves: o_proj right behind paged_attention_v2's partition
// launch - K = heads x 128 (one 128-k unit per head), at most 16 sequences, the decode kernel's 4-wave shape, LDS for the rows
inline bool decode_attn_supported(int M, int N, int K, int num_groups, int heads, int head_size, int max_parts) {
  // Rows served by default: NONE. With the reduce kernel on the same one-round-trip arithmetic (v2_vec4_*) the fused form is level
  // with reduce launch + GEMM - decode step at batch 1 / 2 / 4: 2.171 / 2.318 / 2.393 ms fused, 2.177 / 2.315 / 2.370 unfused
  // (gpurun_out/attn_fuse_ab.log); against the older wave-per-head reduce kernel it had been 2.5 % ahead at batch 1. NMX_GEMM_ATTN =
  // rows to serve (at most 16) switches it on (tests, A/B).
  int max_rows = 0;
  if (const char* e = nmx_tune(NMX_TUNE_GEMM_ATTN)) max_rows = std::min(std::max(atoi(e), 0), 16);
  if (M < 1 || M > max_rows || head_size != 128 || heads * 128 != K || K % 128 != 0 || N % 64 != 0 || max_parts < 1 || max_parts > 8) return false;
  if (!(num_groups == 1 || (K / num_groups) % 128 == 0)) return false;
  const DecodeCfg c = pick_decode_cfg(M, N, K);
  if (c.nw != 4 || c.mt != 1) return false;
  GemmParams q;
  q.M = M; q.K = K; q.k_splits = c.splits; q.attn_max_parts = max_parts;
  return (size_t)4 * 4096 + decode_lds_a_bytes(q, 4, false, true) <= 64 * 1024;
}

template <typename scalar_t, int MT, int NW, bool WS, bool NORM = false, bool ATTN = false>
int launch_decode_cfg(const GemmParams& p, hipStream_t stream) {
  const size_t smem = (size_t)NW * MT * 4096 + decode_lds_a_bytes(p, NW, NORM, ATTN);
  dim3 grid(p.act_out != nullptr ? p.N / 128 : p.N / 64, p.k_splits, ceil_div(p.M, 16 * MT));
#define NMX_LAUNCH_DECODE(GROUPED_)                                                                                  \
  {                                                                                                                  \
    auto kern = marlin_decode_kernel<scalar_t, MT, NW, GROUPED_, WS, NORM, ATTN>;                                            \
    if (smem > 64 * 1024)                                                                                            \
      NMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                  (int)smem));                                                                       \
    kern<<<grid, 64 * NW, smem, stream>>>(p);                                                                        \
  }
  if (p.num_groups > 1) NMX_LAUNCH_DECODE(true) else NMX_LAUNCH_DECODE(false)
#undef NMX_LAUNCH_DECODE
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

template <typename scalar_t>
int launch_decode(GemmParams& p, const DecodeCfg& cfg, void* scratch, int64_t scratch_bytes, hipStream_t stream) {
  p.k_splits = cfg.splits;
  if (p.k_splits > 1) {
    const int64_t per = (int64_t)p.M * p.N * sizeof(float);
    const int fit = scratch == nullptr ? 1 : (int)std::min<int64_t>(p.k_splits, scratch_bytes / per);
    p.k_splits = std::max(1, fit);  // never allocate here: graph capture
  }
  p.partial = reinterpret_cast<float*>(scratch);
  int rc;
  // the 16-wave shape exists where it fits 128 registers per lane without spilling: 16 rows, fp16
  constexpr bool HAS16 = __is_same(scalar_t, f16);
  const bool ws = cfg.ws || p.num_groups == 1;
  // gate_up + silu_and_mul in one launch: the unsplit 4-wave shape run as 4 + 4 waves (gate | up column group)
  const bool fuse = p.act_out != nullptr && p.k_splits == 1 && cfg.nw == 4 && cfg.mt == 1 && p.N % 128 == 0;
  void* const act_out = p.act_out;
  p.act_out = nullptr;  // the kernels below read it as "fused mode"
  if (p.attn_tmp != nullptr) {
    // attention-reduce A operand: the 4-wave shape (callers ask decode_attn_supported() first)
    if (cfg.mt == 1 && cfg.nw == 4 && act_out == nullptr && (p.k_splits == 1 || p.defer_reduce) &&
        (size_t)4 * 4096 + decode_lds_a_bytes(p, 4, false, true) <= 64 * 1024)
      return ws ? launch_decode_cfg<scalar_t, 1, 4, true, false, true>(p, stream) : launch_decode_cfg<scalar_t, 1, 4, false, false, true>(p, stream);
    NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "attention-reduce gptq_marlin_gemm: shape not served (M = %d, N = %d, K = %d)", p.M, p.N, p.K);
  }
  if (p.norm_partial != nullptr) {
    // norm-fused A operand: the two shapes the batch <= 4 decode step uses (callers ask decode_norm_supported() first)
    if (fuse) {
      p.act_out = act_out;
      rc = ws ? launch_decode_cfg<scalar_t, 1, 8, true, true>(p, stream) : launch_decode_cfg<scalar_t, 1, 8, false, true>(p, stream);
      p.act_done = 1;
      return rc;
    }
    if (cfg.mt == 1 && cfg.nw == 4 && act_out == nullptr && (p.k_splits == 1 || p.defer_reduce))
      return ws ? launch_decode_cfg<scalar_t, 1, 4, true, true>(p, stream) : launch_decode_cfg<scalar_t, 1, 4, false, true>(p, stream);
    NMX_CHECK(false, NMX_ERR_UNSUPPORTED, "norm-fused gptq_marlin_gemm: shape not served (M = %d, N = %d, K = %d)", p.M, p.N, p.K);
  }
  if (fuse) {
    p.act_out = act_out;
    rc = ws ? launch_decode_cfg<scalar_t, 1, 8, true>(p, stream) : launch_decode_cfg<scalar_t, 1, 8, false>(p, stream);
    p.act_done = 1;
    return rc;
  }
  if (cfg.mt == 1) {
    if (cfg.nw == 16 && HAS16) {
      if constexpr (HAS16) rc = launch_decode_cfg<scalar_t, 1, 16, true>(p, stream);
      else rc = NMX_ERR_UNSUPPORTED;
    } else if (cfg.nw >= 8) {
      rc = ws ? launch_decode_cfg<scalar_t, 1, 8, true>(p, stream) : launch_decode_cfg<scalar_t, 1, 8, false>(p, stream);
    } else {
      rc = ws ? launch_decode_cfg<scalar_t, 1, 4, true>(p, stream) : launch_decode_cfg<scalar_t, 1, 4, false>(p, stream);
    }
  } else {
    if (cfg.nw >= 8)
      rc = ws ? launch_decode_cfg<scalar_t, 2, 8, true>(p, stream) : launch_decode_cfg<scalar_t, 2, 8, false>(p, stream);
    else
      rc = ws ? launch_decode_cfg<scalar_t, 2, 4, true>(p, stream) : launch_decode_cfg<scalar_t, 2, 4, false>(p, stream);
  }
  if (rc != NMX_OK) return rc;
  if (p.k_splits > 1 && !p.defer_reduce) {
    const int64_t mn4 = (int64_t)p.M * p.N / 4;
    splitk_reduce_kernel<scalar_t><<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>(
        reinterpret_cast<scalar_t*>(p.c), p.partial, mn4, p.k_splits);
    NMX_LAUNCH_CHECK();
  }
  return NMX_OK;
}

template <typename scalar_t, int KIND, bool SP = false>
int launch_skinny(GemmParams& p, void* scratch, int64_t scratch_bytes, hipStream_t stream) {
  if constexpr (!SP && KIND == W_INT4) {
    // small M, plain int4 (no act-order, channel-wise or 128-multiple groups): the barrier-free decode kernel
    const bool plain = p.perm == nullptr && !p.slow_act_order && (p.num_groups == 1 || p.group_size % 128 == 0) &&
                       (int64_t)p.M * p.K * 2 < (1ll << 31) && (int64_t)p.K * p.N < (1ll << 31) &&
                       (int64_t)p.num_groups * p.N * 2 < (1ll << 31);
    if (plain) {
      const DecodeCfg dc = pick_decode_cfg(p.M, p.N, p.K);
      if (dc.nw != 0) return launch_decode<scalar_t>(p, dc, scratch, scratch_bytes, stream);
    }
  }
  if constexpr (!SP && KIND == W_INT4) {
    // 64 < M, fp16, plain layout: both operands by LDS-DMA (marlin_dma.hip)
    int dsplits = 1;
    if (p.perm == nullptr && !p.slow_act_order &&
        nmx_dma_pick(p.M, p.N, p.K, p.num_groups, p.group_size, KIND, __is_same(scalar_t, bf16) ? 1 : 0, &dsplits)) {
      NmxWideCall call;
      call.a = p.a; call.b = p.b; call.scales = p.scales; call.c = p.c; call.scratch = scratch; call.scratch_bytes = scratch_bytes;
      call.M = p.M; call.N = p.N; call.K = p.K; call.num_groups = p.num_groups; call.group_size = p.group_size;
      call.kind = KIND; call.is_bf16 = 0;
      call.defer_reduce = p.defer_reduce;
      call.act_out = p.act_out;
      const int rc = nmx_dma_run(call, dsplits, stream);
      p.k_splits = call.splits_done;
      p.act_done = call.act_done;
      return rc;
    }
  }
  if constexpr (!SP) {
    // M > 64, plain layout: 128-row wave tiles (marlin_wide.hip)
    NmxWideCfg wc;
    if (p.perm == nullptr && !p.slow_act_order && nmx_wide_pick(p.M, p.N, p.K, p.num_groups, p.group_size, &wc, KIND)) {
      NmxWideCall call;
      call.a = p.a; call.b = p.b; call.scales = p.scales; call.c = p.c; call.scratch = scratch; call.scratch_bytes = scratch_bytes;
      call.M = p.M; call.N = p.N; call.K = p.K; call.num_groups = p.num_groups; call.group_size = p.group_size;
      call.kind = KIND; call.is_bf16 = __is_same(scalar_t, bf16) ? 1 : 0;
      call.defer_reduce = p.defer_reduce;
      call.act_out = p.act_out;
      const int rc = nmx_wide_launch(call, wc, stream);
      p.k_splits = call.splits_done;
      p.act_done = call.act_done;
      return rc;
    }
    if (use_large(p, false)) return launch_large<scalar_t, KIND>(p, scratch, scratch_bytes, stream);
  }
  if constexpr (SP) {
    // 2:4-sparse, M > 64: the wide tiles with the sparse MFMA (marlin_wide_kernel<SP = true>, round 3)
    NmxWideCfg wc;
    if (nmx_wide_pick(p.M, p.N, p.K, p.num_groups, p.group_size, &wc, KIND, true)) {
      NmxWideCall call;
      call.a = p.a; call.b = p.b; call.scales = p.scales; call.c = p.c; call.scratch = scratch; call.scratch_bytes = scratch_bytes;
      call.M = p.M; call.N = p.N; call.K = p.K; call.num_groups = p.num_groups; call.group_size = p.group_size;
      call.kind = KIND; call.is_bf16 = 0;
      call.defer_reduce = p.defer_reduce;
      call.meta = p.meta;
      const int rc = nmx_wide_launch(call, wc, stream);
      p.k_splits = call.splits_done;
      return rc;
    }
  }
  GemmCfg cfg = pick_cfg(p.M, p.N, p.K);
  p.k_splits = cfg.splits;
  if (p.k_splits > 1) {
    const int64_t need = (int64_t)p.k_splits * p.M * p.N * sizeof(float);
    if (scratch == nullptr || scratch_bytes < need) {
      // degrade to the number of splits that fits (never allocate here: graph capture)
      int fit = scratch == nullptr ? 1 : (int)(scratch_bytes / ((int64_t)p.M * p.N * sizeof(float)));
      p.k_splits = std::max(1, std::min(p.k_splits, fit));
    }
  }
  p.partial = reinterpret_cast<float*>(scratch);
  if (const char* e = nmx_tune(NMX_TUNE_GEMM_XCD_SPLIT)) p.xcd_split = atoi(e) != 0;
  int rc;
  const int sub_k = (KIND == W_INT4) ? 128 : 64;
  const bool generic = p.perm != nullptr || p.slow_act_order || (p.num_groups > 1 && p.group_size % 128 != 0) ||
                       (p.K % sub_k != 0) || ((int64_t)p.M * p.K * 2 >= (1ll << 31)) || ((int64_t)p.K * p.N >= (1ll << 31));
  // gate_up + silu_and_mul in one launch (8 < M <= 16 here; larger M: marlin_wide, M <= 8: the decode kernel): the unsplit
  // 16-row shape runs with two column groups - the tile's gate columns and the matching up columns - on 8 waves, K sliced
  // exactly as in the plain launch (bit-identical sums). Measured: batch 16 2.92 -> 2.74 ms; the 32-row shape as 2 + 2
  // column groups on 8 waves LOSES to two 4-wave workgroups per CU + the activation launch (batch 32 3.53 -> 3.65 ms).
  void* const act_out = p.act_out;
  p.act_out = nullptr;  // the kernel reads it as "fused mode"
  if constexpr (!SP && KIND == W_INT4) {
    const bool shape16 = cfg.mt == 1 && cfg.ng == 1 && !cfg.w8;
    if (act_out != nullptr && p.k_splits == 1 && !generic && shape16 && p.N % 128 == 0) {
      p.act_out = act_out;
      p.act_done = 1;
      if (p.num_groups > 1) return launch_cfg<scalar_t, KIND, 1, 2, 1, false, true>(p, stream);
      return launch_cfg<scalar_t, KIND, 1, 2, 0, false, true>(p, stream);
    }
  }
  if (generic) rc = launch_mode<scalar_t, KIND, 2, SP>(p, cfg, stream);
  else if (p.num_groups > 1) rc = launch_mode<scalar_t, KIND, 1, SP>(p, cfg, stream);
  else rc = launch_mode<scalar_t, KIND, 0, SP>(p, cfg, stream);
  if (rc != NMX_OK) return rc;
  if (p.k_splits > 1 && !p.defer_reduce) {
    const int64_t mn4 = (int64_t)p.M * p.N / 4;
    splitk_reduce_kernel<scalar_t><<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>(
        reinterpret_cast<scalar_t*>(p.c), p.partial, mn4, p.k_splits);
    NMX_LAUNCH_CHECK();
  }
  return NMX_OK;
}

}  // namespace
