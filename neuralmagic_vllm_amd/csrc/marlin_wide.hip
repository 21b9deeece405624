// marlin_wide_kernel: the Marlin-format W4A16 / W8A16 / fp8-W8A16 GEMM for M > 64 (the MFMA-bound regime of large decode
// batches and prefill). Same op contract as marlin_kernel.h (reference: csrc/quantization/gptq_marlin/gptq_marlin.cu:1735-1868,
// large-M tile table :1395-1412), different work decomposition:
//
//   * every wave owns a 128-row x 64-column output tile (8 x 4 MFMA 16x16x32 accumulators = 128 registers) and streams
//     the packed weights of its 64 columns HBM -> VGPR -> dequant -> MFMA A operand exactly like marlin_gemm_kernel
//     (no LDS round trip for weights), but each dequantised fragment now feeds 8 MFMAs instead of 4: ~60 dequant VALU
//     per 32 MFMAs, which fits in the issue slots the MFMAs leave free (2 per 16x16x32);
//   * a workgroup is 8 waves = WM row halves x WN column groups x WK K-slices. The K-slices of a workgroup dequantise
//     DIFFERENT weights (no redundant conversion) and are summed through LDS at the end, so a 256 x 128 tile fills the
//     chip on wide matrices (gate_up: 224 workgroups) with no fp32 partials in HBM and no reduce launch;
//   * activations are staged per K-slice into LDS in MFMA fragment order, 64 k per stage, double-buffered, from ONE
//     register set that is refilled right after it has been written (loads of stage s + 2 fly during stage s + 1);
//   * MFMAs and the int4 -> fp16 conversion are ordered asm statements ([MFMA, 2 VALU] pairs); loads are compiler-visible
//     buffer intrinsics issued in one fixed periodic order, so hipcc's own counted vmcnt waits keep them in flight.
//
// Algorithmic bytes per call: K*N*bits/8 + groups*N*2 + 2*M*K + 2*M*N; flops 2*M*N*K.
#include "marlin_kernel.h"
#include "marlin_wide_api.h"

namespace {

// Timing ablations (tools/wide_ablate.sh; results are WRONG when set, never defined in the product build): bit 0 skip
// MFMAs, 1 skip dequant, 2 skip LDS fragment reads, 3 skip barriers, 4 skip LDS writes, 5 skip activation loads,
// 6 skip weight loads
#ifndef NMX_WABLATE
#define NMX_WABLATE 0
#endif
#ifndef NMX_WIDE_ROT
#define NMX_WIDE_ROT 1   // rotate every workgroup's K order (see stage_of)
#endif
#ifndef NMX_WIDE_RING
#define NMX_WIDE_RING(MT) ((MT) == 4 ? 8 : 4)   // weight ring slots (k-steps) of the 64-row / 128-row wave tiles
#endif
#ifndef NMX_WIDE_NT
#define NMX_WIDE_NT 0    // non-temporal weight loads
#endif

// ---- single instructions as ordered asm statements ------------------------------------------------------------------
// hipcc's scheduler bunches the dequantisation in front of the MFMAs and makes every group of four MFMAs wait for a
// just-issued ds_read (measured: 0.36 of the MFMA rate). asm volatile statements keep their program order, so the
// instruction stream below is the one written: [MFMA, 2 VALU] pairs - tools/probes/valu_mfma_probe.hip measures two
// plain or packed-f16 VALU per v_mfma_f32_16x16x32 as nearly free (+8 %) and every further one at ~4 cycles.
__device__ __forceinline__ void a_mfma_f16(f32x4& acc, const u32x4& a, const u32x4& b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void a_mfma_bf16(f32x4& acc, const u32x4& a, const u32x4& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void a_and_or(uint32_t& d, uint32_t q, uint32_t mask_s, uint32_t magic_v) {
  asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(q), "s"(mask_s), "v"(magic_v));
}
__device__ __forceinline__ void a_pk_add(uint32_t& d, uint32_t c_s) { asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(d) : "s"(c_s)); }
__device__ __forceinline__ void a_pk_fma(uint32_t& d, uint32_t b_s, uint32_t c_v) {
  asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(d) : "s"(b_s), "v"(c_v));
}
// d *= (s.half[H], s.half[H]): the scale row stays packed as loaded, op_sel broadcasts one half to both lanes
template <int H>
__device__ __forceinline__ void a_pk_mul_h(uint32_t& d, uint32_t s_v) {
  if constexpr (H == 0) asm volatile("v_pk_mul_f16 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(d) : "v"(s_v));
  else asm volatile("v_pk_mul_f16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(d) : "v"(s_v));
}
// zero-point forms of the two fix-up operations: the constant comes from half H of a packed row, broadcast to both lanes
template <int H>
__device__ __forceinline__ void a_pk_add_h(uint32_t& d, uint32_t z_v) {
  if constexpr (H == 0) asm volatile("v_pk_add_f16 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(d) : "v"(z_v));
  else asm volatile("v_pk_add_f16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(d) : "v"(z_v));
}
template <int H>
__device__ __forceinline__ void a_pk_fma_h(uint32_t& d, uint32_t b_s, uint32_t c_v) {
  if constexpr (H == 0) asm volatile("v_pk_fma_f16 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(d) : "s"(b_s), "v"(c_v));
  else asm volatile("v_pk_fma_f16 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "+v"(d) : "s"(b_s), "v"(c_v));
}
__device__ __forceinline__ void a_lshr8(uint32_t& d, uint32_t q) { asm volatile("v_lshrrev_b32 %0, 8, %1" : "=v"(d) : "v"(q)); }

// Whole-line stores of a wave's (16 MT) x 64 tile of 16-bit outputs (round 3, late; marlin_dma.hip store_tile16 has the
// measurements: the accumulator layout gives a lane 4 - 2:4: 2 - consecutive columns of a row, i.e. 16 rows x 32 (8) bytes per
// wave instruction and 4 - 16 x the write requests of whole lines). The lanes park their pieces in a wave-private LDS image with
// a 144-byte row stride; flush_tile16 reads 16 bytes per lane back (8 lanes = one row's 128-byte line) and stores them.
constexpr int kTRow = 144;
template <int MT>
__device__ __forceinline__ void flush_tile16(const char* img, uint16_t* dst, int64_t ld, int rows_valid, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private image: the wave's own LDS operations complete in order
  const int r0 = lane >> 3, c = lane & 7;
#pragma unroll
  for (int i = 0; i < 2 * MT; ++i) {
    const int row = 8 * i + r0;
    const u32x4 d = *reinterpret_cast<const u32x4*>(img + row * kTRow + 16 * c);
    if (row < rows_valid) *reinterpret_cast<u32x4*>(dst + (int64_t)row * ld + 8 * c) = d;
  }
}

struct WFrag { uint32_t w[4][4]; };  // four MFMA operand fragments (tile t, dword)

// Operation j of the int4 -> fp16 conversion of one k-step (two packed words per tile pair; tile t = column
// c8 + 8 t + 32 hi uses word t >> 1, the "+8 column" half sits 8 bits up). Per tile 12 operations (13 for odd tiles):
// [2 shifts] 4 x and_or, 2 x (pk_add, pk_fma) = exact (v - 8), then (SCALED) 4 x pk_mul by the group scale - the same
// arithmetic as Dequant<f16, W_INT4>. The order keeps dependent operations >= 4 apart.
template <bool SCALED>
struct DqPlan {
  static constexpr int PER_EVEN = SCALED ? 12 : 8, PER_ODD = PER_EVEN + 2, TOTAL = 2 * (PER_EVEN + PER_ODD);
};
// ZPV: per-column zero points - zc = the packed row of -(1024 + z), zh = the same + 960 = -(64 + z) (both exact in fp16)
template <bool SCALED, int J, bool ZPV = false>
__device__ __forceinline__ void dq_op(const u32x2& q0, const u32x2& q1, const u32x2& sc, WFrag& f, uint32_t (&tmp)[2],
                                      uint32_t magic, uint32_t neg72, const u32x2& zc = u32x2{0, 0}, const u32x2& zh = u32x2{0, 0}) {
  using P = DqPlan<SCALED>;
  constexpr int PAIR = P::PER_EVEN + P::PER_ODD;
  constexpr int tp = J / PAIR, r = J % PAIR;             // tile pair, op inside the pair
  constexpr bool odd = r >= P::PER_EVEN;
  constexpr int t = 2 * tp + (odd ? 1 : 0);
  constexpr int o = odd ? r - P::PER_EVEN - 2 : r;       // op inside the tile (-2, -1 = the shifts)
  if constexpr (odd && o == -2) a_lshr8(tmp[0], q0[tp]);
  else if constexpr (odd && o == -1) a_lshr8(tmp[1], q1[tp]);
  else if constexpr (o < 4) {
    const uint32_t src = odd ? tmp[o >> 1] : ((o >> 1) ? q1[tp] : q0[tp]);
    a_and_or(f.w[t][o], src, (o & 1) ? 0x00f000f0u : 0x000f000fu, magic);
  } else if constexpr (o < 8) {
    constexpr int e = o - 4;
    if constexpr (ZPV) {
      if constexpr ((e & 1) == 0) a_pk_add_h<t & 1>(f.w[t][e], zc[t >> 1]);              // (1024 + v) - (1024 + z)
      else a_pk_fma_h<t & 1>(f.w[t][e], 0x2c002c00u, zh[t >> 1]);                         // (1024 + 16 v) / 16 - (64 + z)
    } else if constexpr ((e & 1) == 0) a_pk_add(f.w[t][e], 0xe408e408u);       // (1024 + v) - 1032
    else a_pk_fma(f.w[t][e], 0x2c002c00u, neg72);                        // (1024 + 16 v) / 16 - 72
  } else {
    a_pk_mul_h<t & 1>(f.w[t][o - 8], sc[t >> 1]);  // scale of tile t = half t & 1 of word t >> 1 of the scale row
  }
}

// SP = true (round 3): 2:4-sparse weights (gptq_marlin_24_gemm; fp16, int4 / int8) on the same tile machinery. A k-step is
// then ONE row of the packed tensor (32 dense k: the lane's words hold, per column tile, the two kept values of quads g and
// g + 4) plus the lane's 16 bytes of metadata; the MFMA is v_smfmac_f32_16x16x64_f16 on the compressed operands of BOTH
// k-steps of a stage (index register built from the metadata, 16 bits selected per tile); the activations are staged in
// fragments (k-step, g') = k 4 g' .. 4 g' + 3 and 16 + 4 g' .. 16 + 4 g' + 3; an accumulator row r of tile x = 2 p + q is column
// 8 (4 (g & 1) + r) + 2 (g >> 1) + p + 4 q of the 64-column group - all exactly as marlin_gemm_kernel<SP = true> has them
// (marlin_kernel.h; reference marlin_24_cuda_kernel.cu:111-860, common/mma.h:39-82). Half the conversion work and half
// the weight bytes of the dense launch per flop; compiler-scheduled conversion (no hand-placed plan).
// ZP = true (round 3; fp16, int4, grouped): per-(group, column) zero points from p.zeros - AWQ weights repacked into the Marlin
// layout (marlin_zp_gemm.hip) - with the arithmetic of marlin_gemm_kernel<ZP>: (1024 + q) - (1024 + z) exactly, one rounding by the
// scale. The zero row travels with the scale row (same positions); the hand-placed conversion plan takes its two fix-up constants from it (dq_op<.., ZPV>).
template <typename scalar_t, int KIND, int MODE, int WM, int WN, int WK, int MTP, bool SP = false, bool ZP = false>
__global__ __launch_bounds__(64 * WM * WN * WK, 2) void marlin_wide_kernel(const GemmParams p) {
  static_assert(!SP || (__is_same(scalar_t, f16) && KIND != W_FP8), "2:4 path: fp16, int4 / int8 weights");
  static_assert(!ZP || (__is_same(scalar_t, f16) && KIND == W_INT4 && MODE == 1 && !SP), "zero points: fp16, int4, grouped scales");
  constexpr bool I4 = (KIND == W_INT4);
  constexpr bool FAST = I4 && __is_same(scalar_t, f16) && !SP;  // hand-placed conversion; other kinds: compiler-scheduled
  constexpr bool SCALED = (MODE == 1);
  constexpr int MT = MTP;                       // 16-row MFMA tiles per wave: 8 (128-row wave tile) or 4 (64 rows, M <= 64)
  constexpr int NTILE = 4;                      // 16-column MFMA tiles per wave
  constexpr int BM = 16 * MT * WM;
  constexpr int TS = 64 * WM * WN;              // threads of one K-slice
  constexpr int NA = BM * 8 / TS;               // 16-byte activation pieces per thread and 64-k stage
  constexpr int A_IMG = BM * 128;               // bytes of one stage image [k-step 2][g 4][row BM][16 B]
  constexpr int WORDS64 = I4 ? 128 : 256;       // int32 per (k-tile, 64-column group)
  // load issue order per iteration: [batch(s + 2)] [W(2 s + 5)] [W(2 s + 6)]; hipcc derives the counted vmcnt waits
  static_assert((NA == 2 || NA == 4 || NA == 8) && NA <= MT, "activation pieces per thread");
  using bvec_t = typename std::conditional<I4, u32x2, u32x4>::type;
  using mvec_t = typename std::conditional<SP, u32x4, bvec_t>::type;  // second half of a k-step: k-tile row 2 ks + 1 / the metadata

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15, c8 = li & 7, hi = li >> 3;
  const int wn = wave % WN, wm = (wave / WN) % WM, wk = wave / (WN * WM);
  const int N = p.N, K = p.K, M = p.M;

  // blockIdx.x -> (column tile, row block); the row blocks of one column tile are 8 ids apart = same XCD (placement is
  // a speed matter only: the second row block then finds the tile's weights in that XCD's L2)
  const int m_blocks = (M + BM - 1) / BM;
  // With 2 / 4 / 8 K splits the XCD (= linear workgroup id % 8) selects the split too, as in marlin_gemm_kernel: a split's
  // slice of the activations is then fetched by 8 / splits XCDs instead of all 8.
  int tile_x, block_m, split_id;
  if (p.xcd_split && (p.k_splits == 2 || p.k_splits == 4 || p.k_splits == 8)) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int per_split = 8 / p.k_splits;
    const int q = lin >> 3;
    split_id = (lin & 7) / per_split;
    block_m = q % m_blocks;
    tile_x = (q / m_blocks) * per_split + (lin & 7) % per_split;
  } else {
    const int bx_group = blockIdx.x / (8 * m_blocks), bx_r = blockIdx.x % (8 * m_blocks);
    tile_x = bx_group * 8 + (bx_r & 7);
    block_m = bx_r >> 3;
    split_id = blockIdx.y;
  }
  if (tile_x * WN * 64 >= N) return;            // padding workgroup (column tiles are rounded up to a multiple of 8)
  // Fused silu_and_mul (p.act_out, host guarantees no K split and (N / 2) % (64 WN) == 0): the tile is 32 WN gate columns
  // PLUS the 32 WN up columns N / 2 further right - column groups wn < WN / 2 stream gate weights, the others the
  // matching up weights, and the epilogue pairs them through LDS.
  const bool fuse_act = !SP && p.act_out != nullptr;  // (the host never asks the 2:4 launch for the fused activation)
  const int n0 = fuse_act ? (wn >= WN / 2 ? N / 2 : 0) + (tile_x * (WN / 2) + (wn % (WN / 2))) * 64 : (tile_x * WN + wn) * 64;
  const bool col_ok = n0 < N;
  const int m0 = block_m * BM;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* abuf = smem + (size_t)wk * 2 * A_IMG;

  // ---- K range of this slice, in 64-k stages; every slice of the workgroup runs `per` iterations (shared barriers) ----
  const int total_stages = K / 64;
  const int workers = p.k_splits * WK;
  const int per = (total_stages + workers - 1) / workers;
  const int worker = split_id * WK + wk;
  const int st_begin = min(worker * per, total_stages), st_end = min(st_begin + per, total_stages);
  const int nst = st_end - st_begin;             // stages this slice really has (the rest of `per` multiply zeros)
  // Every column tile reads the SAME activation rows; workgroups that walk K in the same order hit the same few L2 lines
  // at the same time (28-32 requesters per line and XCD). Each workgroup therefore starts its K walk at its own offset
  // and wraps around: the sum is the same set of products, accumulated in a rotated order.
  // The row blocks of one column tile walk in the SAME order: they then stream the tile's weights together and the second
  // one hits L2 (with a per-row-block offset every row block fetched the weights from HBM again: rocprofv3 FETCH_SIZE
  // 150 MB per gate_up launch at M = 256 against 61 MB of operands).
  // The offset is a function of the PLAIN column tile (64 WN columns), periodic over the two halves of N when they hold whole
  // tiles: the fused gate | up tile then walks K in the order both of its halves have in the plain launch, and the fused
  // op stays bit-identical to GEMM + silu_and_mul.
  const int n_tiles = (N + 64 * WN - 1) / (64 * WN);
  const int period = (N % (128 * WN) == 0) ? n_tiles / 2 : n_tiles;
  const int tile_plain = fuse_act ? tile_x >> 1 : tile_x;
  const int rot = (NMX_WIDE_ROT && nst > 1) ? (int)(((int64_t)(tile_plain % period) * nst) / period) % nst : 0;
  auto stage_of = [&](int rel) {                 // absolute stage of this slice's rel-th iteration (clamped past the end)
    int r = min(rel, max(nst - 1, 0)) + rot;
    r = r >= nst ? r - nst : r;
    return min(st_begin + max(r, 0), total_stages - 1);
  };

  // ---- descriptors and per-lane offsets ----
  const int ktiles = SP ? K / 32 : K / 16;       // rows of the packed tensor (2:4: one row = 32 dense k)
  const int row_bytes = N * (I4 ? 8 : 16);       // one k-tile row of the packed tensor
  // Every vector-memory load is a compiler-visible buffer intrinsic: hipcc counts the waits itself and never reads a
  // destination early. (Inline-asm loads with "+v" destinations were tried first: under register pressure the allocator
  // re-homed a tied operand with a copy made BEFORE the load had landed - rare wrong scale rows.) The asm MFMA / VALU
  // statements around them have side effects, so the loads keep the program positions they are given here.
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.b), 0, ktiles * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, M * K * (int)sizeof(scalar_t), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.scales), 0, p.num_groups * N * (int)sizeof(scalar_t), 0x00020000);
  // weights: lane (g, hi, c8) reads words 2 hi, 2 hi + 1 (int4; 4 words for 8-bit) of chunk 4 c8 + g of both k-tile rows
  const int b_voff = (((col_ok ? n0 : 0) / 64) * WORDS64 + (4 * c8 + g) * (I4 ? 4 : 8) + (I4 ? 2 : 4) * hi) * 4;
  const __amdgpu_buffer_rsrc_t rs_z =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ZP ? p.zeros : p.scales), 0, ZP ? p.num_groups * N * (int)sizeof(scalar_t) : 0, 0x00020000);
  // 2:4 metadata: the lane's 8 reordered int16 (tile x = 2 p + q, k-half cc at 4 p + 2 cc + q) start at int16 index
  // 2 (row N + n0 + 32 hi + 4 c8) (format_24.py:21-50 solved for this lane's columns, as in marlin_gemm_kernel)
  const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(SP ? p.meta : (const void*)p.b), 0, SP ? ktiles * N * 4 : 0, 0x00020000);
  const int m_voff = (((col_ok ? n0 : 0) + 32 * hi + 4 * c8) * 2) * 2;
  // activations: piece i of this thread = row i * (TS / 8) + (tis >> 3), 16-byte chunk cc8 = tis & 7 of the 128-byte
  // stage row: a wave instruction reads 8 rows x one whole 128-byte line
  const int tis = (wave % (WM * WN)) * 64 + lane;
  const int cc8 = tis & 7, aks = cc8 >> 2, acc_ = cc8 & 3;
  // rows >= M (and every row of a stage past this slice's range) get a voffset beyond the descriptor's range: the
  // hardware returns zeros without touching memory. Piece i is row (tis >> 3) + i * (TS / 8): valid iff i * (TS / 8) < a_lim.
  const int a_base = (int)(((int64_t)(m0 + (tis >> 3)) * K + cc8 * 8) * sizeof(scalar_t));
  const int a_lim = M - m0 - (tis >> 3);
  const int a_step = (TS / 8) * K * (int)sizeof(scalar_t);  // wave-uniform: goes into the scalar offset
  // LDS image: fragment (k-step ks, lane group g, row) = 16 bytes at ((ks * 4 + g) * BM + (row ^ 4 ks)) * 16; dword e2 of
  // piece (row, ks, cc) is dword cc of fragment (ks, g = e2, row). The row ^ 4 ks swizzle puts the two k-steps that one
  // ds_write_b32 wave-half covers on disjoint banks; the 16 row-lanes of a fragment read stay 16 distinct slots.
  // 2:4: lane group g multiplies quads g and g + 4 of the k-step: dword e2 of chunk cc (k = 8 cc + 2 e2 + {0, 1}) is dword
  // 2 (cc >> 1) + (e2 & 1) of fragment (ks, g = 2 (cc & 1) + (e2 >> 1), row)
  char* const w_base = SP ? abuf + ((aks * 4 + 2 * (acc_ & 1)) * BM + ((tis >> 3) ^ (4 * aks))) * 16 + 8 * (acc_ >> 1)
                          : abuf + ((aks * 4) * BM + ((tis >> 3) ^ (4 * aks))) * 16 + 4 * acc_;
  // scales (MODE 1): the lane's four tile columns c8 + 8 t + 32 hi sit at positions 8 c8 + 4 hi + t (scale_perm)
  const int s_voff = (int)((((col_ok ? n0 : 0) / 64) * 64 + 8 * c8 + 4 * hi) * sizeof(scalar_t));

  struct BStep { bvec_t q0; mvec_t q1; };
  // k-step j of this slice lives in ring[j % RD]. 128-row wave tiles: 4 slots (2 stages ahead; the registers are the
  // accumulators'). 64-row wave tiles have registers to spare and run where the weight stream is the bound (M <= 64 ...
  // 256 on small matrices): 8 slots = 7 k-steps (7 KiB per wave, 56 KiB per CU) of weights in flight - tools/probes/
  // l2_ingest_probe.hip needs ~64 KiB per CU to saturate HBM.
  constexpr int RD = NMX_WIDE_RING(MT);
  BStep ring[RD];
  u32x4 areg[NA];
  u32x2 sraw = {0, 0};
  u32x2 scc = {0, 0}, scn = {0, 0};              // packed scale rows (4 halves = the lane's 4 tiles) of the current / next stage
  u32x2 zraw = {0, 0}, zcc = {0, 0}, zcn = {0, 0};  // (ZP) the matching rows of -(1024 + z)
  WFrag wfa, wfb;                                // dequantised fragments of the even / odd k-step of a stage
  // 2:4: one v_smfmac_f32_16x16x64_f16 per (row tile, column tile) and STAGE - twice the products per matrix-pipe cycle of the
  // 32-k form (tools/probes/smfmac_rate_probe.hip: every fp16 MFMA form issues at ~16-17 cycles). Its operands are the two
  // k-steps' compressed values side by side - {ks 0: quad g, quad g + 4, ks 1: quad g, quad g + 4}, index fields in the same
  // order (16 bits per tile, the half selected with ABID) - but the 64-k form pairs them with the activation operand ACROSS
  // lane groups (one-hot experiments, tools/probes/run_smfmac_probe.py: pair pa of lane group ga multiplies 4-k slot
  // 2 (ga >> 1) + (pa & 1) of activation lane group 2 (ga & 1) + (pa >> 1)). Activation lane group gb therefore supplies, of
  // k-step gb & 1, the staged fragments (gb >> 1) and (gb >> 1) + 2 - the staging layout itself is the 32-k form's.
  struct SFrag { uint32_t a[4][4]; uint32_t idx[2]; };
  SFrag sfa, sfb;                                // (SP) operands of the current / the next stage, roles swap every stage
#pragma unroll
  for (int i = 0; i < RD; ++i) { ring[i].q0 = bvec_t{}; ring[i].q1 = mvec_t{}; }
#pragma unroll
  for (int i = 0; i < NA; ++i) areg[i] = u32x4{0, 0, 0, 0};

  f32x4 acc[MT][NTILE];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NTILE; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // The loads below take ABSOLUTE stages. The prologue gets them from stage_of(); the main loop keeps the stages of the
  // walk positions it will need next in SGPRs (wpos[], advanced once per iteration by walk_advance) - stage_of() per load
  // was 29 scalar instructions per k-step, a quarter of the loop's instruction stream at 64 rows.
  auto issue_w = [&](int stage, int ks, BStep& r) {  // k-step ks of absolute stage `stage`
    const int soff = (SP ? 1 : 2) * (2 * stage + ks) * row_bytes;  // never past the tensor
    if constexpr ((NMX_WABLATE & 64) != 0) return;
    if constexpr (SP) {
      if constexpr (I4) r.q0 = __builtin_amdgcn_raw_buffer_load_b64(rs_b, b_voff, soff, 0);
      else r.q0 = __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff, soff, 0);
      r.q1 = __builtin_amdgcn_raw_buffer_load_b128(rs_m, m_voff, (2 * stage + ks) * N * 4, 0);
    } else if constexpr (I4) {
      r.q0 = __builtin_amdgcn_raw_buffer_load_b64(rs_b, b_voff, soff, NMX_WIDE_NT ? 2 : 0);
      r.q1 = __builtin_amdgcn_raw_buffer_load_b64(rs_b, b_voff, soff + row_bytes, NMX_WIDE_NT ? 2 : 0);
    } else {
      r.q0 = __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff, soff, NMX_WIDE_NT ? 2 : 0);
      r.q1 = __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff, soff + row_bytes, NMX_WIDE_NT ? 2 : 0);
    }
  };
  // group of a stage = stage / (group_size / 64) as a multiply-high with the rounded-up reciprocal of 2^31 (branch-free, also
  // for one stage per group; exact while stage * stages_per_group < 2^31)
  const uint32_t gs_stages = SCALED ? (uint32_t)max(p.group_size / 64, 1) : 1u;
  const uint32_t gs_inv = (0x80000000u + gs_stages - 1) / gs_stages;
  auto scale_soff = [&](int stage) {
    const int grp = min((int)__umulhi(2u * (uint32_t)stage, gs_inv), p.num_groups - 1);
    return grp * N * (int)sizeof(scalar_t);
  };
  // batch(rel): the activation pieces of the rel-th stage of the walk and the scale row of stage rel + 1
  auto load_piece = [&](int i, int stage, bool in_range) {  // activation piece i of absolute stage `stage`
    if constexpr ((NMX_WABLATE & 32) != 0) return;
    const int soff = stage * 64 * (int)sizeof(scalar_t);
    const int lim = in_range ? a_lim : 0;
    areg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, i * (TS / 8) < lim ? a_base : (int)0x7ff00000, soff + i * a_step, 0);
  };
  auto load_scale = [&](int stage) {  // the scale row (and zero row) of absolute stage `stage`
    if constexpr (SCALED) sraw = __builtin_amdgcn_raw_buffer_load_b64(rs_s, s_voff, scale_soff(stage), 0);
    if constexpr (ZP) zraw = __builtin_amdgcn_raw_buffer_load_b64(rs_z, s_voff, scale_soff(stage), 0);
  };
  auto issue_batch = [&](int st) {  // prologue: walk position st
#pragma unroll
    for (int i = 0; i < NA; ++i) load_piece(i, stage_of(st), st < nst);
    load_scale(stage_of(st + 1));
  };
  // walk positions st + 2 .. st + 1 + RD / 2 of the iteration that runs stage st (wpos[0]: its activation batch, wpos[1]: its
  // scale row, wpos[RD / 2 - 2], wpos[RD / 2 - 1]: its two weight k-steps)
  constexpr int NCUR = RD / 2;
  int wpos[NCUR];
#pragma unroll
  for (int j = 0; j < NCUR; ++j) wpos[j] = stage_of(2 + j);
  auto walk_advance = [&](int st) {  // after the iteration of stage st: positions st + 3 .. st + 2 + RD / 2
    int nx = wpos[NCUR - 1] + 1;
    nx = nx >= st_begin + nst ? st_begin : nx;
    nx = st + 2 + NCUR < nst ? nx : wpos[NCUR - 1];  // past the end of the slice: stay on its last stage (never consumed)
#pragma unroll
    for (int j = 0; j + 1 < NCUR; ++j) wpos[j] = wpos[j + 1];
    wpos[NCUR - 1] = nx;
  };
  auto scale_operand = [&](u32x2 v, int t) -> uint32_t {  // what Dequant<>::run expects: (s, s) fp16 pair / fp32 bits
    union { u32x2 v; scalar_t e[4]; } raw;
    raw.v = v;
    if constexpr (__is_same(scalar_t, f16)) {
      union { f16 h[2]; uint32_t u; } pk;
      pk.h[0] = raw.e[t];
      pk.h[1] = raw.e[t];
      return pk.u;
    } else {
      return __builtin_bit_cast(uint32_t, (float)raw.e[t]);
    }
  };
  // LDS writes of activation piece i into buffer `buf` (fragment order)
  auto write_piece = [&](int i, int buf) {
    if constexpr ((NMX_WABLATE & 16) != 0) return;
    char* wb = w_base + buf * A_IMG + i * (TS / 8) * 16;
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      if constexpr (SP) *reinterpret_cast<uint32_t*>(wb + (e2 >> 1) * BM * 16 + 4 * (e2 & 1)) = areg[i][e2];
      else *reinterpret_cast<uint32_t*>(wb + e2 * BM * 16) = areg[i][e2];
    }
  };
  auto stage_barrier = [&]() {
    if constexpr ((NMX_WABLATE & 8) != 0) __builtin_amdgcn_wave_barrier();
    else __syncthreads();
  };

  // whole k-step conversion, compiler-scheduled (prologue; and every k-step of the kinds without a hand-placed plan)
  auto dequant_cxx = [&](const BStep& r, const u32x2& sc, WFrag& f, const u32x2& zr = u32x2{0, 0}) {
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      const uint32_t s2t = SCALED ? scale_operand(sc, t) : 0u;
      uint32_t w0, w1;
      if constexpr (I4) {
        w0 = r.q0[t >> 1] >> (8 * (t & 1));
        w1 = r.q1[t >> 1] >> (8 * (t & 1));
      } else {
        w0 = r.q0[t];
        w1 = r.q1[t];
      }
      if constexpr ((NMX_WABLATE & 2) != 0) {
        f.w[t][0] = w0; f.w[t][1] = w1; f.w[t][2] = w0 ^ s2t; f.w[t][3] = w1;
        continue;
      }
      if constexpr (ZP) {
        const uint32_t zneg = scale_operand(zr, t);                                      // (-(1024 + z), -(1024 + z))
        const uint32_t zhi = h2_bits(bits_h2(zneg) + bits_h2(0x63806380u));              // + 960 = -(64 + z), exact
        dequant_zp_f16(w0, s2t, true, zneg, zhi, f.w[t][0], f.w[t][1]);
        dequant_zp_f16(w1, s2t, true, zneg, zhi, f.w[t][2], f.w[t][3]);
      } else {
        Dequant<scalar_t, KIND>::run(w0, s2t, SCALED, f.w[t][0], f.w[t][1]);
        Dequant<scalar_t, KIND>::run(w1, s2t, SCALED, f.w[t][2], f.w[t][3]);
      }
    }
  };

  auto dequant_sp64 = [&](const BStep& r0, const BStep& r1, const u32x2& sc, SFrag& f) {
    if constexpr (SP) {
      uint32_t x0[2], x1[2];
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {  // per k-step: byte 0 = tile 2 p, byte 2 = tile 2 p + 1 (low nibble quad g, high nibble quad g + 4)
        x0[pp] = ((r0.q1[2 * pp] >> (4 * g)) & 0x000f000fu) | (((r0.q1[2 * pp + 1] >> (4 * g)) << 4) & 0x00f000f0u);
        x1[pp] = ((r1.q1[2 * pp] >> (4 * g)) & 0x000f000fu) | (((r1.q1[2 * pp + 1] >> (4 * g)) << 4) & 0x00f000f0u);
        f.idx[pp] = x0[pp] | (x1[pp] << 8);   // half 0 = tile 2 p: {ks 0 byte, ks 1 byte}, half 1 = tile 2 p + 1
      }
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        const uint32_t s2t = SCALED ? scale_operand(sc, t) : 0u;
        const uint32_t w0 = I4 ? (r0.q0[t >> 1] >> (8 * (t & 1))) : r0.q0[t];
        const uint32_t w1 = I4 ? (r1.q0[t >> 1] >> (8 * (t & 1))) : r1.q0[t];
        Dequant<scalar_t, KIND>::run(w0, s2t, SCALED, f.a[t][0], f.a[t][1]);
        Dequant<scalar_t, KIND>::run(w1, s2t, SCALED, f.a[t][2], f.a[t][3]);
      }
    }
  };
  const uint32_t magic = 0x64006400u, neg72 = 0xd480d480u;
  const char* const r_base0 = abuf + (g * BM + wm * 16 * MT + li) * 16;
  const char* const r_base1 = abuf + ((4 + g) * BM + wm * 16 * MT + (li ^ 4)) * 16;

  // One k-step of a stage: 32 MFMAs on fragments `cur` x the 8 activation fragments of (buf, KS); in their shadows the
  // conversion of the NEXT k-step's packed words `nxt` into `out`, the fragment reads 4 row tiles ahead (KS = 0: running
  // on into k-step 1's fragments), and (LAND) the wait for + LDS writes of the next stage's activation batch.
  auto kstep_block = [&](auto ks_c, auto land_c, const WFrag& cur, const BStep& nxt, const u32x2& s2, WFrag& out,
                         u32x4 (&af)[MT + 4], int buf, int st, BStep& refill, const u32x2& z2 = u32x2{0, 0}) {
    constexpr int KS = decltype(ks_c)::value;
    constexpr bool LAND = decltype(land_c)::value;
    const char* rb = (KS == 0 ? r_base0 : r_base1) + buf * A_IMG;
    const char* rb1 = r_base1 + buf * A_IMG;
    uint32_t tmp[2];
    if constexpr (!FAST) dequant_cxx(nxt, s2, out, z2);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      u32x4 wq[NTILE];
#pragma unroll
      for (int t = 0; t < NTILE; ++t) wq[t] = u32x4{cur.w[t][0], cur.w[t][1], cur.w[t][2], cur.w[t][3]};
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        constexpr int dummy = 0;
        const int i = mt * NTILE + t;
        // builtin MFMA here: the compiler pads the VALU-write -> MFMA-read distance of its own conversion code
        if constexpr ((NMX_WABLATE & 1) != 0) acc[mt][t][0] += __builtin_bit_cast(float, wq[t][mt & 3] ^ af[mt][t & 3]);
        else acc[mt][t] = mfma_16x16x32<scalar_t>(wq[t], af[mt], acc[mt][t]);
        (void)dummy; (void)i;
      }
      // fragment read 4 row tiles ahead: af[mt + 4] of this k-step, or (KS = 0) af[mt - 4] of k-step 1
      if constexpr ((NMX_WABLATE & 4) == 0) {
        if (mt + 4 < MT) af[mt + 4] = *reinterpret_cast<const u32x4*>(rb + (mt + 4) * 256);
        else if constexpr (KS == 0) af[MT + (mt + 4 - MT)] = *reinterpret_cast<const u32x4*>(rb1 + (mt + 4 - MT) * 256);
      }
      // (LAND) this stage's LDS writes of the next stage's activations, each piece's registers refilled for stage
      // st + 2 one step later; the k-step's packed words (ring slot `refill`, consumed by the previous block) are
      // re-requested here too: the loads are SPREAD over the MFMA rows instead of issued as one burst per iteration -
      // a CU retires roughly one vector-memory wave instruction per ~38 cycles, and a burst of 13 per wave from all
      // eight waves at the same point of the loop left the matrix pipes idle meanwhile
      if constexpr (LAND) {
        constexpr int STEP = MT / NA;
        if (mt % STEP == 0) write_piece(mt / STEP, buf ^ 1);
        if (mt % STEP == 0 && mt >= STEP) load_piece(mt / STEP - 1, wpos[0], st + 2 < nst);
        if (mt == 1) issue_w(wpos[NCUR - 2], 1, refill);  // k-step 2 st + 1 + RD
      }
    }
    if constexpr (LAND) load_piece(NA - 1, wpos[0], st + 2 < nst);
  };
  // FAST path: the same block with the conversion operations placed two per MFMA
  auto kstep_block_fast = [&](auto ks_c, auto land_c, const WFrag& cur, const BStep& nxt, const u32x2& s2, WFrag& out,
                              u32x4 (&af)[MT + 4], int buf, int st, BStep& refill, const u32x2& z2 = u32x2{0, 0}) {
    constexpr int KS = decltype(ks_c)::value;
    constexpr bool LAND = decltype(land_c)::value;
    const char* rb = (KS == 0 ? r_base0 : r_base1) + buf * A_IMG;
    const char* rb1 = r_base1 + buf * A_IMG;
    uint32_t tmp[2];
    u32x4 wq[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) wq[t] = u32x4{cur.w[t][0], cur.w[t][1], cur.w[t][2], cur.w[t][3]};
    constexpr int NOPS = DqPlan<SCALED>::TOTAL;
    constexpr int OPM = (NOPS + MT * NTILE - 1) / (MT * NTILE);  // conversion operations per MFMA: 2 (MT = 8) / 3-4 (MT = 4)
    u32x2 zh2 = {0, 0};
    if constexpr (ZP) {
      zh2[0] = h2_bits(bits_h2(z2[0]) + bits_h2(0x63806380u));
      zh2[1] = h2_bits(bits_h2(z2[1]) + bits_h2(0x63806380u));
    }
    auto ops = [&](auto i_c) {  // the conversion operations behind MFMA number i
      constexpr int I = decltype(i_c)::value;
      if constexpr ((NMX_WABLATE & 2) == 0) {
        if constexpr (I4) {
          if constexpr (OPM * I < NOPS) dq_op<SCALED, OPM * I, ZP>(nxt.q0, nxt.q1, s2, out, tmp, magic, neg72, z2, zh2);
          if constexpr (OPM * I + 1 < NOPS) dq_op<SCALED, OPM * I + 1, ZP>(nxt.q0, nxt.q1, s2, out, tmp, magic, neg72, z2, zh2);
          if constexpr (OPM > 2 && OPM * I + 2 < NOPS) dq_op<SCALED, OPM * I + 2, ZP>(nxt.q0, nxt.q1, s2, out, tmp, magic, neg72, z2, zh2);
          if constexpr (OPM > 3 && OPM * I + 3 < NOPS) dq_op<SCALED, OPM * I + 3, ZP>(nxt.q0, nxt.q1, s2, out, tmp, magic, neg72, z2, zh2);
        }
      }
    };
    auto row = [&](auto mt_c) {
      constexpr int mt = decltype(mt_c)::value;
      a_mfma_f16(acc[mt][0], wq[0], af[KS == 0 ? mt : (mt < 4 ? MT + mt : mt)]);
      ops(std::integral_constant<int, mt * 4 + 0>{});
      a_mfma_f16(acc[mt][1], wq[1], af[KS == 0 ? mt : (mt < 4 ? MT + mt : mt)]);
      ops(std::integral_constant<int, mt * 4 + 1>{});
      a_mfma_f16(acc[mt][2], wq[2], af[KS == 0 ? mt : (mt < 4 ? MT + mt : mt)]);
      ops(std::integral_constant<int, mt * 4 + 2>{});
      a_mfma_f16(acc[mt][3], wq[3], af[KS == 0 ? mt : (mt < 4 ? MT + mt : mt)]);
      ops(std::integral_constant<int, mt * 4 + 3>{});
      if constexpr ((NMX_WABLATE & 4) == 0) {
        if constexpr (mt + 4 < MT) af[mt + 4] = *reinterpret_cast<const u32x4*>(rb + (mt + 4) * 256);
        else if constexpr (KS == 0) af[MT + (mt + 4 - MT)] = *reinterpret_cast<const u32x4*>(rb1 + (mt + 4 - MT) * 256);
      }
      if constexpr (LAND) {  // see kstep_block
        constexpr int STEP = MT / NA;
        if constexpr (mt % STEP == 0) write_piece(mt / STEP, buf ^ 1);
        if constexpr (mt % STEP == 0 && mt >= STEP) load_piece(mt / STEP - 1, wpos[0], st + 2 < nst);
        if constexpr (mt == 1) issue_w(wpos[NCUR - 2], 1, refill);  // k-step 2 st + 1 + RD
      }
    };
    row(std::integral_constant<int, 0>{});
    row(std::integral_constant<int, 1>{});
    row(std::integral_constant<int, 2>{});
    row(std::integral_constant<int, 3>{});
    if constexpr (MT > 4) {
      row(std::integral_constant<int, 4>{});
      row(std::integral_constant<int, 5>{});
      row(std::integral_constant<int, 6>{});
      row(std::integral_constant<int, 7>{});
    }
    if constexpr (LAND) load_piece(NA - 1, wpos[0], st + 2 < nst);
  };

  // 2:4 stage: MT x 4 sparse MFMAs of 64 k on `cur`, in their shadow the conversion of the NEXT stage's two k-steps (ring slots
  // ra, rb -> `out`), the fragment reads two row tiles ahead, and the next stage's activation batch + the ring refills exactly
  // as kstep_block places them
  auto stage_sp = [&](const SFrag& cur, SFrag& out, BStep& ra, BStep& rb, const u32x2& s2, int buf, int st) {
    if constexpr (SP) {
      typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));
      typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
      const char* rb0 = abuf + buf * A_IMG + (((4 * (g & 1) + (g >> 1)) * BM + wm * 16 * MT + (li ^ (4 * (g & 1)))) * 16);
      const char* rb1 = rb0 + 2 * BM * 16;
      u32x4 a0[MT], a1[MT];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        a0[mt] = *reinterpret_cast<const u32x4*>(rb0 + mt * 256);
        a1[mt] = *reinterpret_cast<const u32x4*>(rb1 + mt * 256);
      }
      dequant_sp64(ra, rb, s2, out);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const u32x8 b8 = u32x8{a0[mt][0], a0[mt][1], a0[mt][2], a0[mt][3], a1[mt][0], a1[mt][1], a1[mt][2], a1[mt][3]};
        const f16x16 bb = __builtin_bit_cast(f16x16, b8);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
          const f16x8 wa = __builtin_bit_cast(f16x8, u32x4{cur.a[t][0], cur.a[t][1], cur.a[t][2], cur.a[t][3]});
          if (t & 1) acc[mt][t] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(wa, bb, acc[mt][t], (int)cur.idx[t >> 1], 0, 1);
          else acc[mt][t] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(wa, bb, acc[mt][t], (int)cur.idx[t >> 1], 0, 0);
        }
        if (mt + 2 < MT) {
          a0[mt + 2] = *reinterpret_cast<const u32x4*>(rb0 + (mt + 2) * 256);
          a1[mt + 2] = *reinterpret_cast<const u32x4*>(rb1 + (mt + 2) * 256);
        }
        constexpr int STEP = MT / NA;
        if (mt % STEP == 0) write_piece(mt / STEP, buf ^ 1);
        if (mt % STEP == 0 && mt >= STEP) load_piece(mt / STEP - 1, wpos[0], st + 2 < nst);
        if (mt == 1) issue_w(wpos[NCUR - 1], 0, ra);  // k-steps 2 st + 2 + RD, 2 st + 3 + RD: stage st + 1 + RD / 2
        // keeps the fragment reads two row tiles ahead (hipcc otherwise hoists all 2 MT of them to the top: 64 registers, spills)
        __builtin_amdgcn_sched_barrier(0);
      }
      load_piece(NA - 1, wpos[0], st + 2 < nst);
      issue_w(wpos[NCUR - 1], 1, rb);
    }
  };

  // ---- prologue, in the steady-state issue order ----
  if constexpr (SCALED) scc = __builtin_amdgcn_raw_buffer_load_b64(rs_s, s_voff, scale_soff(stage_of(0)), 0);
  if constexpr (ZP) zcc = __builtin_amdgcn_raw_buffer_load_b64(rs_z, s_voff, scale_soff(stage_of(0)), 0);
  issue_w(stage_of(0), 0, ring[0]);
  issue_batch(0);
#pragma unroll
  for (int j = 1; j <= RD - 2; ++j) issue_w(stage_of(j >> 1), j & 1, ring[j]);
#pragma unroll
  for (int i = 0; i < NA; ++i) write_piece(i, 0);
  scn = sraw;
  zcn = zraw;
  if constexpr (SP) {
    dequant_sp64(ring[0], ring[1], scc, sfa);
    issue_w(stage_of((RD + 1) >> 1), 1, ring[1]);  // slot 1 is free again: k-step RD + 1
  } else dequant_cxx(ring[0], scc, wfa, zcc);
  // the same order as an iteration issues them: hipcc merges the pending-load state of this path and of the loop's back
  // edge at the loop head, and any difference turns the first waits of the body into vmcnt(0)
  issue_w(stage_of((RD - 1) >> 1), 1, ring[RD - 1]);
  issue_batch(1);
  issue_w(stage_of(RD >> 1), 0, ring[0]);
  stage_barrier();

  using KS0 = std::integral_constant<int, 0>;
  using KS1 = std::integral_constant<int, 1>;
  using NOLAND = std::integral_constant<bool, false>;
  using DOLAND = std::integral_constant<bool, true>;
  auto body = [&](auto par_c, int it) {
    constexpr int PH = decltype(par_c)::value;    // = it % (RD / 2): ring slots 2 PH + {1, 2}
    constexpr int PAR = PH & 1;                   // LDS buffer
    const int st = it;  // relative stage of the walk
    BStep& r1 = ring[(2 * PH + 1) % RD];
    BStep& r2 = ring[(2 * PH + 2) % RD];
    if constexpr (SP) {
      BStep& ra = ring[(2 * PH + 2) % RD];
      BStep& rb = ring[(2 * PH + 3) % RD];
      if constexpr (PAR == 0) stage_sp(sfa, sfb, ra, rb, scn, PAR, st);
      else stage_sp(sfb, sfa, ra, rb, scn, PAR, st);
      scc = scn;
      scn = sraw;
      load_scale(wpos[1]);
      walk_advance(st);
      stage_barrier();
      return;
    }
    u32x4 af[MT + 4];  // 0..7: k-step 0, 8..11: first four of k-step 1 (then 4..7 again)
    if constexpr ((NMX_WABLATE & 4) == 0) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const u32x4*>(r_base0 + PAR * A_IMG + mt * 256);
    } else {
#pragma unroll
      for (int mt = 0; mt < MT + 4; ++mt) af[mt] = u32x4{(uint32_t)lane, (uint32_t)it, (uint32_t)mt, 0x3c003c00u};
    }
    if constexpr (FAST) {
      kstep_block_fast(KS0{}, NOLAND{}, wfa, r1, scc, wfb, af, PAR, st, r1, zcc);
      kstep_block_fast(KS1{}, DOLAND{}, wfb, r2, scn, wfa, af, PAR, st, r1, zcn);
    } else {
      kstep_block(KS0{}, NOLAND{}, wfa, r1, scc, wfb, af, PAR, st, r1, zcc);
      // k-step 1 of the generic path reads its fragments 4..7 into af[4..7] and 0..3 from af[8..11]
      u32x4 af1[MT + 4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) af1[mt] = af[MT + mt];
      kstep_block(KS1{}, DOLAND{}, wfb, r2, scn, wfa, af1, PAR, st, r1, zcn);
    }
    scc = scn;
    scn = sraw;
    zcc = zcn;
    zcn = zraw;
    load_scale(wpos[1]);               // scale row of walk position st + 3
    issue_w(wpos[NCUR - 1], 0, r2);    // k-step 2 st + 2 + RD
    walk_advance(st);
    stage_barrier();
  };
  // Always whole pairs of iterations (an odd `per` runs one more, on zero activations): with a conditional second half
  // there is a control-flow path from body 0 straight back to body 0, and hipcc's wait counts at the loop head then
  // assume that order of pending loads too (vmcnt(1) / vmcnt(0) in front of the first k-step's conversion).
  // (8 slots: whole quadruples - a ring slot index must be a compile-time constant, registers cannot be indexed)
  for (int it = 0; it < per; it += RD / 2) {
    body(std::integral_constant<int, 0>{}, it);
    body(std::integral_constant<int, 1>{}, it + 1);
    if constexpr (RD == 8) {
      body(std::integral_constant<int, 2>{}, it + 2);
      body(std::integral_constant<int, 3>{}, it + 3);
    }
  }
  // the s_nop covers the MFMA -> VALU read distance that hipcc does not know about (the MFMAs are asm statements)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    asm volatile("s_nop 7" : "+v"(acc[mt][0]), "+v"(acc[mt][1]), "+v"(acc[mt][2]), "+v"(acc[mt][3]));

  // ---- channel-wise scales on the fp32 accumulators (D row 4 g + r of tile t = column 32 (g >> 1) + 8 t + 4 (g & 1) + r;
  //      scale_perm_single: position 32 (b >> 2) + 8 (cc >> 1) + (cc & 1) + 2 (b & 3) holds column cc + 8 b) ----
  if constexpr (MODE == 0) {
    const scalar_t* sc = reinterpret_cast<const scalar_t*>(p.scales) + (col_ok ? n0 : 0);
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 32 * (g >> 1) + 8 * t + 4 * (g & 1) + r;
        const int cc = col & 7, b = col >> 3;
        // (2:4: marlin_24_scale_perm_single is the identity; D row r of tile x = 2 p + q is column 8 (4 (g & 1) + r) + 2 (g >> 1) + p + 4 q)
        const int pos = SP ? 8 * (4 * (g & 1) + r) + 2 * (g >> 1) + (t >> 1) + 4 * (t & 1) : 32 * (b >> 2) + 8 * (cc >> 1) + (cc & 1) + 2 * (b & 3);
        const float sv = Scalar<scalar_t>::to_f32(sc[pos]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][t][r] *= sv;
      }
  }

  // ---- sum the WK K-slices (tree through LDS); slice 0 stores ----
  constexpr int ACC_FLOATS = MT * NTILE * 64 * 4;
  if constexpr (WK > 1) {
    float* red = reinterpret_cast<float*>(smem);
    const int wslot = wave % (WM * WN);
#pragma unroll
    for (int stride = WK / 2; stride >= 1; stride >>= 1) {
      __syncthreads();
      if (wk >= stride && wk < 2 * stride) {
        float* dst = red + ((wk - stride) * (WM * WN) + wslot) * ACC_FLOATS;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < NTILE; ++t) *reinterpret_cast<f32x4*>(dst + ((mt * NTILE + t) * 64 + lane) * 4) = acc[mt][t];
      }
      __syncthreads();
      if (wk < stride) {
        const float* src = red + (wk * (WM * WN) + wslot) * ACC_FLOATS;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < NTILE; ++t) acc[mt][t] += *reinterpret_cast<const f32x4*>(src + ((mt * NTILE + t) * 64 + lane) * 4);
      }
    }
  }
  if (fuse_act) {
    // silu_and_mul on the fp16 / bf16 ROUNDED gate and up values, the arithmetic of act_and_mul_kernel (elementwise.hip;
    // reference activation_kernels.cu:12-30): out = scalar_t(silu(float(gate))) * up, rounded once more
    constexpr int HW = WN / 2;
    u32x2* ex = reinterpret_cast<u32x2*>(smem);
    __syncthreads();  // stage buffers / reduction slabs are free
    const int pair = (wm * HW + wn % HW) * (MT * NTILE * 64);
    if (wk == 0 && wn >= HW) {
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
    u32x2 ov[MT][NTILE];
    if (wk == 0 && wn < HW) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
          union { scalar_t h[4]; u32x2 u; } up, o;
          up.u = ex[pair + (mt * NTILE + t) * 64 + lane];
#pragma unroll
          for (int j = 0; j < 4; ++j) o.h[j] = rnd_mul<scalar_t>(silu_rnd<scalar_t>(Scalar<scalar_t>::from_f32(acc[mt][t][j])), up.h[j]);
          ov[mt][t] = o.u;
        }
    }
    __syncthreads();  // the exchange buffer becomes the transpose images
    if (wk != 0 || wn >= HW || !col_ok) return;
    char* img = smem + (wm * HW + wn) * (16 * MT * kTRow);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < NTILE; ++t)
        *reinterpret_cast<u32x2*>(img + (16 * mt + li) * kTRow + 2 * (32 * (g >> 1) + 8 * t + 4 * (g & 1))) = ov[mt][t];
    flush_tile16<MT>(img, reinterpret_cast<uint16_t*>(p.act_out) + (int64_t)(m0 + wm * 16 * MT) * (N / 2) + n0, N / 2,
                     M - m0 - wm * 16 * MT, lane);
    return;
  }
  // 16-bit outputs (the result, or an fp16 slab of a K split) leave as whole lines through the wave's transpose image
  const bool out16 = p.k_splits == 1 || p.partial_f16;
  if constexpr (WK > 1) __syncthreads();  // the reduction slabs become the transpose images (WK = 1: the loop's last barrier freed the stage buffers)
  if (wk != 0 || !col_ok) return;
  char* const timg = smem + (wm * WN + wn) * (16 * MT * kTRow);
  uint16_t* const tdst = p.k_splits == 1 ? reinterpret_cast<uint16_t*>(p.c) + (int64_t)(m0 + wm * 16 * MT) * N + n0
                                         : reinterpret_cast<uint16_t*>(p.partial) + ((int64_t)split_id * M + m0 + wm * 16 * MT) * N + n0;

  if constexpr (SP) {
    if (out16) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            union { scalar_t h[2]; uint32_t u; } o;  // (p.partial_f16 implies scalar_t = f16)
            o.h[0] = Scalar<scalar_t>::from_f32(acc[mt][q][r]);
            o.h[1] = Scalar<scalar_t>::from_f32(acc[mt][2 + q][r]);
            *reinterpret_cast<uint32_t*>(timg + (16 * mt + li) * kTRow + 2 * (8 * (4 * (g & 1) + r) + 2 * (g >> 1) + 4 * q)) = o.u;
          }
      flush_tile16<MT>(timg, tdst, N, M - m0 - wm * 16 * MT, lane);
      return;
    }
    // tiles q and 2 + q are neighbouring columns -> 2-element stores (as marlin_gemm_kernel<SP>)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + wm * 16 * MT + mt * 16 + li;
      if (m >= M) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int n = n0 + 8 * (4 * (g & 1) + r) + 2 * (g >> 1) + 4 * q;
          const float v0 = acc[mt][q][r], v1 = acc[mt][2 + q][r];
          if (p.k_splits == 1) {
            union { scalar_t h[2]; uint32_t u; } o;
            o.h[0] = Scalar<scalar_t>::from_f32(v0);
            o.h[1] = Scalar<scalar_t>::from_f32(v1);
            *reinterpret_cast<uint32_t*>(reinterpret_cast<scalar_t*>(p.c) + (int64_t)m * N + n) = o.u;
          } else if (p.partial_f16) {
            union { f16 h[2]; uint32_t u; } o;
            o.h[0] = (f16)v0;
            o.h[1] = (f16)v1;
            *reinterpret_cast<uint32_t*>(reinterpret_cast<f16*>(p.partial) + ((int64_t)split_id * M + m) * N + n) = o.u;
          } else {
            *reinterpret_cast<f32x2*>(p.partial + ((int64_t)split_id * M + m) * N + n) = f32x2{v0, v1};
          }
        }
    }
    return;
  }
  // lane (g, li): D rows = 4 consecutive output columns 32 (g >> 1) + 8 t + 4 (g & 1) + r, D col = activation row li
  if (out16) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        union { scalar_t h[4]; u32x2 u; } r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.h[j] = Scalar<scalar_t>::from_f32(acc[mt][t][j]);
        *reinterpret_cast<u32x2*>(timg + (16 * mt + li) * kTRow + 2 * (32 * (g >> 1) + 8 * t + 4 * (g & 1))) = r.u;
      }
    flush_tile16<MT>(timg, tdst, N, M - m0 - wm * 16 * MT, lane);
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + wm * 16 * MT + mt * 16 + li;
    if (m >= M) continue;
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      const int n = n0 + 32 * (g >> 1) + 8 * t + 4 * (g & 1);
      if (p.k_splits == 1) {
        union { scalar_t h[4]; u32x2 u; } r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.h[j] = Scalar<scalar_t>::from_f32(acc[mt][t][j]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<scalar_t*>(p.c) + (int64_t)m * N + n) = r.u;
      } else if (p.partial_f16) {
        union { f16 h[4]; u32x2 u; } r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.h[j] = (f16)acc[mt][t][j];
        *reinterpret_cast<u32x2*>(reinterpret_cast<f16*>(p.partial) + ((int64_t)split_id * M + m) * N + n) = r.u;
      } else {
        *reinterpret_cast<f32x4*>(p.partial + ((int64_t)split_id * M + m) * N + n) = acc[mt][t];
      }
    }
  }
}

template <typename scalar_t, int KIND, int MODE, int WM, int WN, int WK, int MT = 8, bool SP = false, bool ZP = false>
int launch_wide_cfg(const GemmParams& p, hipStream_t stream) {
  constexpr int BM = 16 * MT * WM;
  const size_t stage = (size_t)WK * 2 * BM * 128;
  const size_t red = (WK > 1) ? (size_t)(WK / 2) * WM * WN * MT * 4 * 64 * 4 * sizeof(float) : 0;
  const size_t ex = (size_t)WM * (WN / 2) * MT * 4 * 64 * 8;  // fused silu_and_mul: the up halves as fp16 / bf16
  const size_t timg = (size_t)WM * WN * 16 * MT * kTRow;       // transpose images of the output stores
  const size_t smem = std::max(std::max(std::max(stage, red), ex), timg);
  dim3 grid(ceil_div(ceil_div(p.N, 64 * WN), 8) * 8 * ceil_div(p.M, BM), p.k_splits, 1);
  auto kern = marlin_wide_kernel<scalar_t, KIND, MODE, WM, WN, WK, MT, SP, ZP>;
  if (smem > 64 * 1024)
    NMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  kern<<<grid, 64 * WM * WN * WK, smem, stream>>>(p);
  NMX_LAUNCH_CHECK();
  return NMX_OK;
}

template <typename scalar_t, int KIND, int MODE>
int launch_wide_shape(const GemmParams& p, const NmxWideCfg& c, hipStream_t stream) {
  if (c.mt == 4) {  // 64-row wave tiles (M <= 64): int4 only
    if constexpr (KIND == W_INT4) {
      if (c.wn == 2) return launch_wide_cfg<scalar_t, KIND, MODE, 1, 2, 4, 4>(p, stream);
      return launch_wide_cfg<scalar_t, KIND, MODE, 1, 4, 2, 4>(p, stream);
    } else {
      return NMX_ERR_UNSUPPORTED;
    }
  }
  if (c.wm == 2 && c.wn == 2) return launch_wide_cfg<scalar_t, KIND, MODE, 2, 2, 2>(p, stream);
  if (c.wm == 2 && c.wn == 4) return launch_wide_cfg<scalar_t, KIND, MODE, 2, 4, 1>(p, stream);
  if (c.wm == 1 && c.wn == 2) return launch_wide_cfg<scalar_t, KIND, MODE, 1, 2, 4>(p, stream);
  return launch_wide_cfg<scalar_t, KIND, MODE, 1, 4, 2>(p, stream);
}

// 2:4-sparse launches: fp16, int4 / int8; 128-row wave tiles as 1 x 4 x 2 or 1 x 2 x 4 waves, 64-row wave tiles (int4) as 1 x 2 x 4
template <int KIND>
int launch_wide_sparse(const GemmParams& p, const NmxWideCfg& c, hipStream_t stream) {
  if (p.num_groups > 1) {
    if (c.mt == 4) {
      if constexpr (KIND == W_INT4) return launch_wide_cfg<f16, KIND, 1, 1, 2, 4, 4, true>(p, stream);
      else return NMX_ERR_UNSUPPORTED;
    }
    if (c.wn == 2) return launch_wide_cfg<f16, KIND, 1, 1, 2, 4, 8, true>(p, stream);
    return launch_wide_cfg<f16, KIND, 1, 1, 4, 2, 8, true>(p, stream);
  }
  if (c.mt == 4) {
    if constexpr (KIND == W_INT4) return launch_wide_cfg<f16, KIND, 0, 1, 2, 4, 4, true>(p, stream);
    else return NMX_ERR_UNSUPPORTED;
  }
  if (c.wn == 2) return launch_wide_cfg<f16, KIND, 0, 1, 2, 4, 8, true>(p, stream);
  return launch_wide_cfg<f16, KIND, 0, 1, 4, 2, 8, true>(p, stream);
}

// zero-point launches (AWQ on the Marlin layout): fp16 int4, grouped; 128-row wave tiles as 1 x 4 x 2 or 1 x 2 x 4 waves
int launch_wide_zp(const GemmParams& p, const NmxWideCfg& c, hipStream_t stream) {
  // 128-row wave tiles as 1 x 4 x 2 waves only: 1 x 2 x 4 spills 12 registers with the zero rows on top, and the 64-row wave
  // tiles measured level with / behind the row-block kernel on the 70B / TP = 8 shapes (o 13.2 vs 13.9 us, down 26.9 vs 25.0)
  return launch_wide_cfg<f16, W_INT4, 1, 1, 4, 2, 8, false, true>(p, stream);
}

template <typename scalar_t, int KIND>
int launch_wide_kind(const GemmParams& p, const NmxWideCfg& c, hipStream_t stream) {
  if constexpr (KIND == W_FP8) return launch_wide_shape<scalar_t, KIND, 0>(p, c, stream);  // fp8 Marlin: channel-wise scales only
  else {
    if (p.num_groups > 1) return launch_wide_shape<scalar_t, KIND, 1>(p, c, stream);
    return launch_wide_shape<scalar_t, KIND, 0>(p, c, stream);
  }
}

// NMX_GEMM_WIDE = "wm,wn,splits" forces a configuration, "0" disables the kernel (sweeps and tests)
struct WideEnv {
  bool set = false, off = false;
  int wm = 0, wn = 0, splits = 0, mt = 8;
};
WideEnv wide_env() {
  WideEnv w;
  const char* e = nmx_tune(NMX_TUNE_GEMM_WIDE);
  if (e == nullptr) return w;
  int a = 0, b = 0, s = 0, m = 8;
  const int got = sscanf(e, "%d,%d,%d,%d", &a, &b, &s, &m);
  if (got == 1 && a == 0) { w.off = true; return w; }
  if (got >= 3 && (a == 1 || a == 2) && (b == 2 || b == 4) && s >= 1 && s <= 32) {
    w.set = true; w.wm = a; w.wn = b; w.splits = s;
    w.mt = (got == 4 && m == 4 && a == 1) ? 4 : 8;
  }
  return w;
}

}  // namespace

bool nmx_wide_pick(int M, int N, int K, int num_groups, int group_size, NmxWideCfg* cfg, int kind, bool sparse) {
  const WideEnv env = wide_env();
  if (env.off || K % 64 != 0 || N % 64 != 0) return false;
  // M <= 64: the 64-row instantiation wins only where 128-column tiles alone fill the chip and K is short (gate_up at
  // 32 < M <= 64: 29.3 vs 34.5 us); qkv / o / down need K splits and stay on the row-block / decode kernels
  const bool small_ok = M > 32 && K <= 8192 && ceil_div(N, 128) >= 192;
  // (2:4-sparse, round 3: also qkv-like matrices whose 128-column tiles x 4 K splits fill the chip - 12.9 vs 15.2 us at M = 64;
  // o_proj, 32 tiles, stays on the row-block kernel: 11.6 vs 10.4)
  const bool small_sparse = sparse && M > 32 && K <= 4096 && ceil_div(N, 128) * 4 >= 160 && ceil_div(N, 128) * 4 <= 256;
  if (M <= 64 && (kind != W_INT4 || !(env.set ? env.mt == 4 : (small_ok || small_sparse)))) return false;  // 64-row tiles: int4 only
  // 2:4-sparse (round 3): M > 64, K a multiple of 64; group sizes -1 / 128 are the op's own restriction
  if (sparse && (M <= 32 || kind == W_FP8)) return false;
  if (num_groups > 1 && group_size % 64 != 0) return false;
  if ((int64_t)M * K * 2 >= (1ll << 31) || (int64_t)K * N >= (1ll << 31) || (int64_t)num_groups * N * 2 >= (1ll << 31)) return false;
  NmxWideCfg c;
  const int stages = K / 64;
  c.mt = 8;
  if (env.set) {
    c.wm = env.wm; c.wn = env.wn; c.splits = env.splits; c.mt = (M <= 64 || kind == W_INT4) ? env.mt : 8;
  } else if (M <= 64) {
    c.wm = 1; c.wn = 2; c.splits = small_ok ? 1 : 4; c.mt = 4;
  } else if (kind == W_INT4 && M <= 256 && K <= 4096 && N <= 8192) {
    // small matrices (qkv, o) at 64 < M <= 256: 64-row x 128-column tiles, no K split when they alone give >= 192
    // workgroups, else two (tools/lean_sweep.py: qkv 27.1 vs 31.0 us at M = 256, 21.6 vs 25.2 at 128; o 22.6 vs 24.2, 19.7 vs 23.5)
    // K splits (2 or 4) while all of them still fit the chip in one round (<= 256 workgroups); above that the second round's tail
    // costs more than idle CUs do (deferred reduce, tools/lean_sweep.py LEAN_SWEEP_DEFER=1: qkv at M = 192, 144 tiles: 23.6 us
    // unsplit vs 30.5 split; o at M = 256, 128 tiles: 22.5 unsplit vs 17.3 split)
    // (o at M = 80 .. 128, 64 tiles: 4 splits 12.3-12.8 us vs 15.5-15.7 with 2)
    c.wm = 1; c.wn = 2; c.mt = 4;
    const int units64 = ceil_div(N, 128) * ceil_div(M, 64);
    c.splits = 1;
    while (c.splits < 4 && units64 * c.splits * 2 <= 256 && stages / (c.splits * 2 * 4) >= 4) c.splits *= 2;
  } else if (M <= 128 && K <= 8192 && ceil_div(N, 128) >= 192) {
    // gate_up-like at 64 < M <= 128: one row block, 128-column tiles fill the chip without a split (41.0 vs 47.9 us)
    c.wm = 1; c.wn = 2; c.splits = 1;
    // (2:4-sparse: 256-column tiles with two K splits - 35.9 vs 45.2 us at M = 128, row-block kernel 42.7)
    if (sparse) { c.wn = 4; c.splits = 2; }
  } else {
    // Measured (tools/lean_sweep.py, 32-launch graph chains over distinct weights, Llama-3-8B shapes, M = 128 .. 2048;
    // gpurun_out/wide_sweep.log): 128-row x 256-column tiles with two K slices per workgroup win 13-23 % over the 64-row
    // row-block kernel once they alone give (nearly) every CU a workgroup - gate_up from M = 256, qkv from 1024, o / down
    // from 2048 - and, with K splits across workgroups, on long K (down from M = 512: 9 %). Below that the row-block
    // kernel's four waves per SIMD hide the staging latency better and it stays in charge.
    c.wm = 1;
    c.wn = 4;
    const int units = ceil_div(N, 256) * ceil_div(M, 128);
    c.splits = 1;
    if (units < 192) {
      // long K, few tiles: K splits across workgroups down to 14 stages per wave (down_proj at M = 256: 32 tiles x 8 splits,
      // 36.1 us against 42.8 on the row-block kernel, deferred reduce; M = 192: 34.1 vs 38.4; at M = 128 the 16 tiles do
      // not fill the chip and the row-block kernel stays)
      if (K < 8192 || units < 32) return false;
      while (units * c.splits * 2 <= 256 && stages / (c.splits * 2 * 2) >= 14) c.splits *= 2;
      if (units * c.splits < 192) return false;
    }
  }
  if (sparse) {  // instantiated shapes: 1 x 4 x 2 and 1 x 2 x 4 waves (128-row wave tiles), 1 x 2 x 4 with 64-row wave tiles (int4)
    if (c.wm != 1) { c.wm = 1; }
    if (c.mt == 4 && (kind != W_INT4 || c.wn != 2)) c.mt = 8;
  }
  c.wk = 8 / (c.wm * c.wn);
  while (c.splits > 1 && stages / (c.splits * c.wk) < 1) c.splits /= 2;
  *cfg = c;
  return true;
}

int nmx_wide_launch(NmxWideCall& call, const NmxWideCfg& cfg, hipStream_t stream) {
  GemmParams p;
  p.a = call.a; p.b = call.b; p.meta = call.meta; p.zeros = call.zeros; p.scales = call.scales; p.g_idx = nullptr; p.perm = nullptr; p.c = call.c;
  p.M = call.M; p.N = call.N; p.K = call.K; p.num_groups = call.num_groups; p.group_size = call.group_size;
  p.slow_act_order = 0;
  if (const char* e = nmx_tune(NMX_TUNE_GEMM_XCD_SPLIT)) p.xcd_split = atoi(e) != 0;
  p.defer_reduce = call.defer_reduce;
  p.k_splits = cfg.splits;
  // fused silu_and_mul epilogue: only a launch whose workgroups own whole K (the slabs of a K split belong to the consumer)
  call.act_done = (call.meta == nullptr && call.zeros == nullptr && call.act_out != nullptr && cfg.splits == 1 && call.N % 2 == 0 && (call.N / 2) % (64 * cfg.wn) == 0) ? 1 : 0;
  p.act_out = call.act_done ? call.act_out : nullptr;
  if (p.k_splits > 1) {  // never allocate here (graph capture): degrade to the splits that fit
    const int64_t per = (int64_t)p.M * p.N * sizeof(float);
    const int fit = call.scratch == nullptr ? 1 : (int)std::min<int64_t>(p.k_splits, call.scratch_bytes / per);
    p.k_splits = std::max(1, fit);
  }
  p.partial = reinterpret_cast<float*>(call.scratch);
  // fp16 outputs: the slabs of a K split hold fp16 partial sums (NMX_SPLITK_F16; NMX_SLAB_F32=1 keeps fp32 for A/B runs)
  p.partial_f16 = (p.k_splits > 1 && !call.is_bf16 && nmx_tune(NMX_TUNE_SLAB_F32) == nullptr) ? 1 : 0;
  int rc;
#ifdef NMX_WIDE_MIN  // experiment builds: fp16 int4 only (compile time)
  rc = launch_wide_kind<f16, W_INT4>(p, cfg, stream);
#else
  if (call.zeros != nullptr) {
    if (call.is_bf16 || call.kind != W_INT4 || call.num_groups <= 1 || cfg.wm != 1 || !(cfg.mt == 8 && cfg.wn == 4))
      return NMX_ERR_UNSUPPORTED;
    rc = launch_wide_zp(p, cfg, stream);
  } else if (call.meta != nullptr) {
    if (call.is_bf16 || call.kind == W_FP8) return NMX_ERR_UNSUPPORTED;
    rc = call.kind == W_INT4 ? launch_wide_sparse<W_INT4>(p, cfg, stream) : launch_wide_sparse<W_INT8>(p, cfg, stream);
  } else
#define NMX_WIDE_KIND(T)                                                            \
  switch (call.kind) {                                                              \
    case W_INT4: rc = launch_wide_kind<T, W_INT4>(p, cfg, stream); break;            \
    case W_INT8: rc = launch_wide_kind<T, W_INT8>(p, cfg, stream); break;            \
    default: rc = launch_wide_kind<T, W_FP8>(p, cfg, stream); break;                 \
  }
  if (call.is_bf16) { NMX_WIDE_KIND(bf16) } else { NMX_WIDE_KIND(f16) }
#undef NMX_WIDE_KIND
#endif
  if (rc != NMX_OK) return rc;
  const int coded = p.k_splits | (p.partial_f16 ? NMX_SPLITK_F16 : 0);
  call.splits_done = coded;
  if (p.k_splits > 1 && !p.defer_reduce) {
    const int64_t mn4 = (int64_t)p.M * p.N / 4;
    if (call.is_bf16)
      splitk_reduce_kernel<bf16><<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>(reinterpret_cast<bf16*>(p.c), p.partial, mn4, coded);
    else
      splitk_reduce_kernel<f16><<<(unsigned)ceil_div64(mn4, 256), 256, 0, stream>>>(reinterpret_cast<f16*>(p.c), p.partial, mn4, coded);
    NMX_LAUNCH_CHECK();
  }
  return NMX_OK;
}
