// marlin_dma_kernel: the Marlin-format W4A16 GEMM for 64 < M (fp16, int4, plain layout) with BOTH operands delivered by
// LDS-DMA (round 3). Same op contract as marlin_kernel.h / marlin_wide.hip (reference: csrc/quantization/gptq_marlin/
// gptq_marlin.cu:1735-1868, large-batch tile table :1395-1412); what changed against marlin_wide_kernel is how the operands
// reach the MFMAs:
//
//   * marlin_wide_kernel streams the packed weights HBM -> VGPR in every wave and stages the activations through VGPRs with
//     a dword scatter into LDS (16 ds_write_b32 + 9 vector-memory loads per wave and 64-k stage), because the Marlin k order
//     inside a 16-row k-tile ({2m, 2m+1, 2m+8, 2m+9} per 32-bit word) was matched on the ACTIVATION side. Its timing
//     ablations put the vector-memory loads + the waits on them at a quarter of the kernel.
//   * Here the activations keep their NATURAL k order, so a lane's MFMA B fragment is 16 contiguous bytes of its row and a
//     whole 128-row x 64-k tile is ONE `buffer_load_dwordx4 ... lds` per 8 rows (no VGPR, no ds_write, no VALU). The k order
//     is matched on the WEIGHT side instead: the MFMA k-step is not "32 consecutive k" but "rows 8p .. 8p+7 of each of the
//     FOUR k-tiles of a 64-k stage" (p = 0, 1: two MFMA passes per stage); lane group g then owns k-tile g, a lane's 8 k
//     values are 16 g + 8 p + 0..7 (contiguous: chunk 2g + p of the activation row) and come out of the four words
//     m = 0..3 of its chunk column with ONE mask per pass (0x000f000f for p = 0, 0x00f000f0 for p = 1 - wave-uniform).
//     The packed tiles are copied to LDS as stored (two 1-KiB DMA instructions per 64-column group and stage) and each
//     lane reads its 64 bytes back with four ds_read_b128.
//   * Nothing in the loop returns data to a VGPR from global memory, so no wave ever waits on `vmcnt` for its own operands:
//     three LDS buffers per K-group, the DMAs of stage s + 2 (weights: s + 3) are issued at the top of stage s, ONE
//     `s_waitcnt vmcnt(N)` (counted, never 0) + `s_barrier` per stage.
//   * 8 waves = 4 column groups x 2 K-groups; a wave owns a 128-row x 64-column tile (8 x 4 MFMA 16x16x32 accumulators),
//     dequantises its own 64 columns (no redundant conversion inside the workgroup: 1.75 VALU per MFMA, placed two per MFMA
//     as ordered asm statements like marlin_wide's FAST path), the two K-groups work on alternate halves of the K range and
//     are summed through LDS at the end. Cross-workgroup K splits leave fp32 slabs for the fused consumers (3.6).
//
// Numerics: exactly the reference's weight values (exact (q - 8), ONE fp16 rounding of (q - 8) * s for grouped scales,
// channel-wise scales on the fp32 accumulators), fp32 accumulation; only the summation order over k differs from the
// other kernels.
// Algorithmic bytes per call: K*N/2 + groups*N*2 + 2*M*K + 2*M*N; flops 2*M*N*K.
#include <stdio.h>
#include <stdlib.h>

#include "nmx_common.h"
#include "marlin_wide_api.h"

namespace {

// Timing ablations (tools/dma_ablate.sh; results are WRONG when set, never defined in the product build): bit 0 skip MFMAs,
// 1 skip dequant, 2 skip A fragment reads, 3 skip DMA issue, 4 skip barriers, 5 wait one batch later (data race), 6 every
// DMA re-fetches the K-group's first stage (cache-hot sources: issue + LDS-write cost without memory latency), 10 / 11 / 12:
// no in-loop DMA of the packed words / the activations / the scales (the prologue still fills the ring: nothing is dead code)
#ifndef NMX_DABLATE
#define NMX_DABLATE 0
#endif

constexpr int kBM = 128;                     // rows per workgroup
constexpr int kNBUF = 3;                     // LDS ring slots per K-group
constexpr int kWN = 4, kWK = 2;              // 64-column groups x K-groups = 8 waves
constexpr int kAImg = kBM * 128;             // [row][64 k fp16], chunk c of row r at slot c ^ a_swz(r)
constexpr int kWImg = kWN * 2048;            // per column group [k-tile pair q 2][m 4][k-tile & 1][c8 8] x 16 B
constexpr int kSImg = kWN * 256;             // per column group 64 fp16 scales (+ the 128 bytes the upper lanes deposit)
constexpr int kSlot = kAImg + kWImg + kSImg;


struct DmaParams {
  const void* a;
  const int32_t* b;
  const void* scales;
  void* c;
  float* partial;
  void* act_out;
  int M, N, K, num_groups, group_size, k_splits, xcd_split;
  int partial_f16;  // the K-split slabs hold fp16 (NMX_SPLITK_F16)
};

__device__ __forceinline__ void a_mfma_f16(f32x4& acc, const u32x4& a, const u32x4& b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void a_and_or(uint32_t& d, uint32_t q, uint32_t mask_s, uint32_t magic_v) {
  asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(q), "s"(mask_s), "v"(magic_v));
}
__device__ __forceinline__ void a_pk_add(uint32_t& d, uint32_t c_s) { asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(d) : "s"(c_s)); }
__device__ __forceinline__ void a_pk_fma(uint32_t& d, uint32_t b_s, uint32_t c_v) {
  asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(d) : "s"(b_s), "v"(c_v));
}
__device__ __forceinline__ void a_pk_mul(uint32_t& d, uint32_t s_v) { asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(d) : "v"(s_v)); }
__device__ __forceinline__ void a_lshr(uint32_t& d, uint32_t sh_v) { asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(d) : "v"(sh_v)); }

// Chunk c (16 bytes = 8 k) of activation row r sits at slot c ^ a_swz(r) of the row's 128 bytes. Found by exhaustive search over
// the XOR-linear maps of (r & 15): with it the 16 lanes of every ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...: rows li of
// lane groups g and g + 1, chunks 2 g + p) hit 16 different 16-byte bank slots in both passes (the first version, r & 7,
// measured SQ_LDS_BANK_CONFLICT = 41 % of the LDS cycles).
__device__ __host__ __forceinline__ constexpr int a_swz(int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 2); }

// A wave's 128 x 64 tile of 16-bit outputs stored as WHOLE 128-byte lines (round 3, late). The accumulator layout gives a lane 4
// consecutive columns of one row: stored from there, a wave instruction is 16 rows x 32 bytes, and the K-scan ablations priced the
// output stores of a gate_up launch at 6.1 us (2,048 partial-line write requests per workgroup, all workgroups at once). Through a
// wave-private LDS image with a 144-byte row stride (conflict-free ds_write_b64 / aligned ds_read_b128) a lane stores 16 bytes and
// eight lanes cover a row's line: 512 requests per workgroup. v[mt][t] = the 4 values of row 16 mt + li, columns 16 t + 4 g ..;
// dst = the tile's row 0 / column 0, ld in elements, rows >= rows_valid are not stored.
#ifndef NMX_DMA_TSTORE
#define NMX_DMA_TSTORE 1   // 0: the first form of the stores (8 bytes per lane straight from the accumulator layout), for A/B builds
#endif
constexpr int kTRow = 144;
__device__ __forceinline__ void store_tile16(char* img, const u32x2 (&v)[8][4], f16* dst, int64_t ld, int rows_valid, int lane) {
  const int g = lane >> 4, li = lane & 15;
  if constexpr (NMX_DMA_TSTORE == 0) {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (16 * mt + li < rows_valid) *reinterpret_cast<u32x2*>(dst + (int64_t)(16 * mt + li) * ld + 16 * t + 4 * g) = v[mt][t];
    return;
  }
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<u32x2*>(img + (16 * mt + li) * kTRow + 32 * t + 8 * g) = v[mt][t];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private image: the wave's own LDS operations complete in order
  const int r0 = lane >> 3, c = lane & 7;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = 8 * i + r0;
    const u32x4 d = *reinterpret_cast<const u32x4*>(img + row * kTRow + 16 * c);
    if (row < rows_valid) *reinterpret_cast<u32x4*>(dst + (int64_t)row * ld + 8 * c) = d;
  }
}

struct WFrag { uint32_t w[4][4]; };   // [tile t][register m]: MFMA A operand of output columns 16 t + (lane & 15)
struct WRaw { uint32_t r[4][4]; };    // [chunk m][word t] of k-tile g, already shifted right by 8 b

// LS = true: loader / consumer split - waves 4..7 only issue the DMAs (of ALL stages: one K-group), waves 0..3 (one per SIMD)
// only compute. A wave that issues a `buffer_load ... lds` is held at the instruction for ~100-200 cycles under load; with
// every wave doing both, those holds came out of the MFMA issue time (timing ablation: 15-19 of 64 us on gate_up at M = 256).
template <bool SCALED, bool LS>
__global__ __launch_bounds__(512, 2) void marlin_dma_kernel(const DmaParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave & (kWN - 1), kg = wave / kWN;
  const int g = lane >> 4, li = lane & 15, hb = (lane >> 3) & 1, c8 = lane & 7;
  const int N = p.N, K = p.K, M = p.M;

  // blockIdx.x -> (column tile, row block, K split) exactly as marlin_wide_kernel: row blocks of a column tile 8 ids apart
  // (same XCD: the second one finds the tile's weights in that L2), K splits tied to XCD groups
  const int m_blocks = (M + kBM - 1) / kBM;
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
  if (tile_x * kWN * 64 >= N) return;  // padding workgroup (column tiles are rounded up to a multiple of 8)
  if constexpr ((NMX_DABLATE & 256) != 0) { if (p.M > 0) return; }  // dispatch cost only
  // fused silu_and_mul (host: no K split, (N / 2) % 128 == 0): column groups wn < 2 stream gate weights, the others the
  // matching up weights N / 2 further right; paired through LDS in the epilogue
  const bool fuse_act = p.act_out != nullptr;
  const int n0 = fuse_act ? (wn >= kWN / 2 ? N / 2 : 0) + (tile_x * (kWN / 2) + (wn % (kWN / 2))) * 64 : (tile_x * kWN + wn) * 64;
  const bool col_ok = n0 < N;
  const int nl = col_ok ? n0 : 0;  // column group used for the loads
  const int m0 = block_m * kBM;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WKE = LS ? 1 : kWK;                      // K-groups that split the workgroup's K range
  const bool loader = LS ? kg == 1 : true, consumer = LS ? kg == 0 : true;
  char* const ring = smem + (LS ? 0 : kg) * kNBUF * kSlot;

  // ---- K range of this K-group in 64-k stages; every K-group of the workgroup runs `per` iterations (shared barriers) ----
  const int total_stages = K / 64;
  const int workers = p.k_splits * WKE;
  const int per = (total_stages + workers - 1) / workers;
  const int worker = split_id * WKE + (LS ? 0 : kg);
  const int st_begin = min(worker * per, total_stages), st_end = min(st_begin + per, total_stages);
  const int nst = st_end - st_begin;

  // ---- DMA sources ----
  const int row_bytes = N * 8;  // one k-tile row of the packed tensor
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, M * K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.b), 0, (K / 16) * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.scales), 0, p.num_groups * N * 2, 0x00020000);
  // activations: instruction j of this wave deposits rows 8 q .. 8 q + 7 (q = 4 wn + j), lane = (row r = lane >> 3, slot = lane & 7)
  // and fetches chunk slot ^ (r & 7) of its row (the swizzle lives on the SOURCE address: the LDS side of a DMA is lane-linear)
  int a_voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 8 * (4 * wn + j) + (lane >> 3);
    a_voff[j] = (min(m0 + r, M - 1) * K + 8 * ((lane & 7) ^ a_swz(r))) * 2;  // rows past M: a valid row again, never stored
  }
  // weights: instruction q (k-tiles 2 q, 2 q + 1 of the stage), lane = (m = lane >> 4, k-tile & 1 = (lane >> 3) & 1, c8): chunk 4 c8 + m
  const int w_voff = ((lane >> 3) & 1) * row_bytes + (nl / 64) * 512 + (4 * (lane & 7) + (lane >> 4)) * 16;
  // scales: 4-byte DMA, lanes 0..31 = the 64 scales of this column group (lanes 32..63 deposit the same dwords behind them)
  const int s_voff = (nl + 2 * (lane & 31)) * 2;
  const int gs_shift = SCALED ? (31 - __builtin_clz((unsigned)max(p.group_size / 64, 1))) : 0;  // stages per group: a power of two
  const int st_last = min(st_begin + max(nst - 1, 0), total_stages - 1);
  // Every column tile reads the SAME activation rows; workgroups that walk K in the same order ask for the same few L2 lines at
  // the same time. Each workgroup therefore starts its K walk at its own offset and wraps around (the same products, summed in a
  // rotated order) - as marlin_wide_kernel does, with the offset a function of the PLAIN column tile and periodic over the two
  // halves of N, so that a fused gate | up tile walks K in the order both of its halves have in the plain launch (the fused op
  // stays bit-identical to GEMM + silu_and_mul). Row blocks of a column tile share the offset (the second one hits L2).
#ifndef NMX_DMA_ROT
#define NMX_DMA_ROT 1
#endif
  const int n_tiles_r = (N + 64 * kWN - 1) / (64 * kWN);
  const int period_r = (N % (128 * kWN) == 0) ? n_tiles_r / 2 : n_tiles_r;
  const int tile_plain = fuse_act ? tile_x >> 1 : tile_x;
  const int rot = (NMX_DMA_ROT && nst > 1) ? (int)(((int64_t)(tile_plain % max(period_r, 1)) * nst) / max(period_r, 1)) % nst : 0;
  auto stage_abs = [&](int rel) {  // absolute stage of walk position rel (past the range: some valid stage, never consumed)
    int r = min(rel, max(nst - 1, 0)) + rot;
    r = r >= nst ? r - nst : r;
    return min(st_begin + max(r, 0), total_stages - 1);
  };
  // DMA instruction J (0, 1: packed words; 2 .. 5: activations; 6: scales) of a batch: weights + scales of absolute stage sw
  // into ring slot SW, activations of stage sa into slot SA (positions past the range: the last stage again, never consumed)
  auto issue_one = [&](auto j_c, auto sw_c, auto sa_c, int sw, int sa) {
    constexpr int J = decltype(j_c)::value, SW = decltype(sw_c)::value, SA = decltype(sa_c)::value;
    if constexpr ((NMX_DABLATE & 8) != 0) return;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (J < 2) {
      auto* dw = (__attribute__((address_space(3))) char*)(ring + SW * kSlot + kAImg + wn * 2048 + J * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, dw, 16, w_voff, (4 * sw + 2 * J) * row_bytes, 0, 0);
    } else if constexpr (J < 6) {
      auto* da = (__attribute__((address_space(3))) char*)(ring + SA * kSlot + (4 * wn + J - 2) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, da, 16, a_voff[J - 2], sa * 128, 0, 0);
    } else if constexpr (SCALED) {
      const int grp = min(sw >> gs_shift, p.num_groups - 1);
      auto* ds = (__attribute__((address_space(3))) char*)(ring + SW * kSlot + kAImg + kWImg + wn * 256);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, ds, 4, s_voff, grp * N * 2, 0, 0);
    }
#endif
  };
  auto issue = [&](auto sw_c, auto sa_c, int rel_w, int rel_a) {  // a whole batch at once (prologue)
    const int sw = stage_abs(rel_w), sa = stage_abs(rel_a);
    issue_one(std::integral_constant<int, 0>{}, sw_c, sa_c, sw, sa);
    issue_one(std::integral_constant<int, 1>{}, sw_c, sa_c, sw, sa);
    issue_one(std::integral_constant<int, 2>{}, sw_c, sa_c, sw, sa);
    issue_one(std::integral_constant<int, 3>{}, sw_c, sa_c, sw, sa);
    issue_one(std::integral_constant<int, 4>{}, sw_c, sa_c, sw, sa);
    issue_one(std::integral_constant<int, 5>{}, sw_c, sa_c, sw, sa);
    issue_one(std::integral_constant<int, 6>{}, sw_c, sa_c, sw, sa);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  constexpr int NDMA = 6 + (SCALED ? 1 : 0);  // DMA instructions per wave and iteration

  // ---- LDS read addresses ----
  // A fragment (pass p, row tile mt): row 16 mt + li, chunk 2 g + p -> slot (2 g + p) ^ a_swz(li)
  const int a_rd0 = li * 128 + (((2 * g) ^ a_swz(li)) * 16), a_rd1 = li * 128 + (((2 * g + 1) ^ a_swz(li)) * 16);
  // packed words: k-tile g -> instruction q = g >> 1, position (g & 1) * 8 + c8; chunk m at + 256 m
  const int w_rd = kAImg + wn * 2048 + (g >> 1) * 1024 + ((g & 1) * 8 + c8) * 16;
  const int s_rd = kAImg + kWImg + wn * 256 + c8 * 16;

  const uint32_t magic = 0x64006400u, neg72 = 0xd480d480u;
  const uint32_t shv = 8u * (uint32_t)hb;                      // the "+8 column" half of a word sits 8 bits up
  const uint32_t ssel = hb ? 0x03020302u : 0x01000100u;        // v_perm selector: this lane's half of a scale dword, twice

  f32x4 acc[8][4];
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  WRaw raw;
  WFrag wf0, wf1;
  uint32_t s2[4] = {0, 0, 0, 0};

  auto read_raw = [&](const char* buf) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(buf + w_rd + 256 * m);
#pragma unroll
      for (int t = 0; t < 4; ++t) raw.r[m][t] = v[t];
    }
  };
  auto read_scales = [&](const char* buf) {
    if constexpr (SCALED) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(buf + s_rd);  // dword t = scales of columns 16 t + c8 (+ 8)
#pragma unroll
      for (int t = 0; t < 4; ++t) s2[t] = __builtin_amdgcn_perm(v[t], v[t], ssel);
    }
  };
  // conversion operation J (0 .. 47) of pass P into `out`: per tile 4 x and_or, 4 x fix (exact q - 8), 4 x scale
  // (channel-wise: 32 operations, the scale goes on the accumulators)
  auto dq = [&](auto p_c, auto j_c, WFrag& out) {
    constexpr int P = decltype(p_c)::value, J = decltype(j_c)::value;
    constexpr int PER = SCALED ? 12 : 8;
    constexpr int t = J / PER, o = J % PER;
    if constexpr ((NMX_DABLATE & 2) != 0) { if constexpr (o < 4) out.w[t][o] = raw.r[o][t]; return; }
    if constexpr (t < 4) {
      if constexpr (o < 4) a_and_or(out.w[t][o], raw.r[o][t], P == 0 ? 0x000f000fu : 0x00f000f0u, magic);
      else if constexpr (o < 8) {
        if constexpr (P == 0) a_pk_add(out.w[t][o - 4], 0xe408e408u);          // (1024 + q) - 1032
        else a_pk_fma(out.w[t][o - 4], 0x2c002c00u, neg72);                    // (1024 + 16 q) / 16 - 72
      } else a_pk_mul(out.w[t][o - 8], s2[t]);
    }
  };
  constexpr int NOPS = SCALED ? 48 : 32;
  // whole-fragment conversion, prologue only
  auto dq_all = [&](auto p_c, WFrag& out) {
    auto run = [&](auto self, auto j_c) {
      constexpr int J = decltype(j_c)::value;
      if constexpr (J < NOPS) { dq(p_c, j_c, out); self(self, std::integral_constant<int, J + 1>{}); }
    };
    run(run, std::integral_constant<int, 0>{});
  };
  auto shift_raw = [&]() {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t) a_lshr(raw.r[m][t], shv);
  };

  // ---- prologue: walk positions 0 and 1 of the activations, 0 .. 2 of the weights; counts as in the loop ----
  if (loader) {
    issue(I1{}, I0{}, 1, 0);
    issue(I0{}, I0{}, 0, 0);   // the second copy of A(0) is harmless; keeps every batch NDMA instructions long
    issue(I2{}, I1{}, 2, 1);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
  if constexpr ((NMX_DABLATE & 16) == 0) __builtin_amdgcn_s_barrier();
  read_scales(ring);
  read_raw(ring);
  shift_raw();
  dq_all(std::integral_constant<int, 0>{}, wf0);

  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  auto body = [&](auto cur_c, int it) {
    constexpr int CUR = decltype(cur_c)::value, NXT = (CUR + 1) % kNBUF, PRV = (CUR + 2) % kNBUF;
    const char* const buf = ring + CUR * kSlot;
    const char* const nbuf = ring + NXT * kSlot;
    // stage it (and the weights of it + 1) have landed for every wave; everyone is done with the slot refilled below
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NMX_DABLATE & 32) ? 2 * NDMA : NDMA) : "memory");
    if constexpr ((NMX_DABLATE & 16) == 0) __builtin_amdgcn_s_barrier();
    u32x4 af[12];  // 0..7: pass 0 (chunk 2 g), 8..11: the first four of pass 1 (then 4..7 again)
    if constexpr ((NMX_DABLATE & 4) == 0) {
      if (consumer) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const u32x4*>(buf + a_rd0 + mt * 2048);
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < 12; ++mt) af[mt] = u32x4{(uint32_t)lane, (uint32_t)it, (uint32_t)mt, 0x3c003c00u};
    }
    // This iteration's DMA batch (weights of walk position it + 3 into this slot's packed-word image - read out during the
    // last stage -, activations of it + 2 into the slot released at the barrier) is NOT issued here in one burst: all eight
    // waves stalled on the vector-memory issue together right after the barrier and the matrix pipes idled meanwhile
    // (timing ablation: 19 of gate_up's 66 us at M = 256). One instruction goes out behind every second MFMA row, the two
    // K-groups (= the two waves of a SIMD) on alternate rows.
    const int sw_n = (NMX_DABLATE & 64) ? st_begin : stage_abs(it + 3), sa_n = (NMX_DABLATE & 64) ? st_begin : stage_abs(it + 2);
    auto dma_at = [&](auto r_c) {  // R = 0 .. 15: MFMA row of the stage
      constexpr int R = decltype(r_c)::value;
      constexpr bool skip = ((NMX_DABLATE & 1024) != 0 && (R >> 1) < 2) || ((NMX_DABLATE & 2048) != 0 && (R >> 1) >= 2 && (R >> 1) < 6) ||
                            ((NMX_DABLATE & 4096) != 0 && (R >> 1) == 6);  // in-loop DMA issue only: the prologue still fills the ring
      if constexpr (!LS && (R >> 1) < NDMA && !skip) {
        if (kg == (R & 1)) {
          __builtin_amdgcn_sched_barrier(0);
          issue_one(std::integral_constant<int, (R >> 1)>{}, std::integral_constant<int, CUR>{}, std::integral_constant<int, PRV>{}, sw_n, sa_n);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if ((LS && loader) || (!LS && it >= nst)) {  // (a K-group that has run out of stages still issues its batch: uniform counted waits)
      issue(std::integral_constant<int, CUR>{}, std::integral_constant<int, PRV>{}, it + 3, it + 2);
    }
    if (it < nst && consumer) {
      u32x4 wq[4];
      auto mma = [&](f32x4& c, const u32x4& a, const u32x4& b) {
        if constexpr ((NMX_DABLATE & 1) == 0) a_mfma_f16(c, a, b);
        else c[0] += __builtin_bit_cast(float, a[1] ^ b[2]);
      };
      // ---- pass 0: rows 8 g' .. of every k-tile with p = 0; in its shadow the conversion of pass 1's fragments ----
#pragma unroll
      for (int t = 0; t < 4; ++t) wq[t] = u32x4{wf0.w[t][0], wf0.w[t][1], wf0.w[t][2], wf0.w[t][3]};
      auto ops0 = [&](auto i_c) {  // the two conversion operations behind MFMA i of pass 0
        constexpr int I = decltype(i_c)::value;
        if constexpr (2 * I < NOPS) dq(P1{}, std::integral_constant<int, 2 * I>{}, wf1);
        if constexpr (2 * I + 1 < NOPS) dq(P1{}, std::integral_constant<int, 2 * I + 1>{}, wf1);
      };
      auto pass0 = [&](auto self, auto mt_c) {
        constexpr int mt = decltype(mt_c)::value;
        if constexpr (mt < 8) {
          mma(acc[mt][0], wq[0], af[mt]); ops0(std::integral_constant<int, 4 * mt + 0>{});
          mma(acc[mt][1], wq[1], af[mt]); ops0(std::integral_constant<int, 4 * mt + 1>{});
          mma(acc[mt][2], wq[2], af[mt]); ops0(std::integral_constant<int, 4 * mt + 2>{});
          mma(acc[mt][3], wq[3], af[mt]); ops0(std::integral_constant<int, 4 * mt + 3>{});
          if constexpr ((NMX_DABLATE & 4) == 0) {
            if constexpr (mt + 4 < 8) af[mt + 4] = *reinterpret_cast<const u32x4*>(buf + a_rd0 + (mt + 4) * 2048);
            else af[8 + (mt - 4)] = *reinterpret_cast<const u32x4*>(buf + a_rd1 + (mt - 4) * 2048);
          }
          // the packed words and scales of stage it + 1 (landed: same wait as above), once wf1 is complete (rows 6, 7)
          if constexpr (mt == 6) { read_scales(nbuf); read_raw(nbuf); }
          dma_at(std::integral_constant<int, mt>{});
          self(self, std::integral_constant<int, mt + 1>{});
        }
      };
      pass0(pass0, std::integral_constant<int, 0>{});
      // ---- pass 1; in its shadow the shift + pass-0 conversion of stage it + 1 ----
#pragma unroll
      for (int t = 0; t < 4; ++t) wq[t] = u32x4{wf1.w[t][0], wf1.w[t][1], wf1.w[t][2], wf1.w[t][3]};
      auto ops1 = [&](auto i_c) {
        constexpr int I = decltype(i_c)::value;
        if constexpr (I < 8) {  // 16 shifts, two per MFMA
          constexpr int e0 = 2 * I, e1 = 2 * I + 1;
          a_lshr(raw.r[e0 / 4][e0 % 4], shv);
          a_lshr(raw.r[e1 / 4][e1 % 4], shv);
        } else {
          constexpr int J = 2 * (I - 8);
          if constexpr (J < NOPS) dq(P0{}, std::integral_constant<int, J>{}, wf0);
          if constexpr (J + 1 < NOPS) dq(P0{}, std::integral_constant<int, J + 1>{}, wf0);
        }
      };
      auto pass1 = [&](auto self, auto mt_c) {
        constexpr int mt = decltype(mt_c)::value;
        if constexpr (mt < 8) {
          constexpr int ai = mt < 4 ? 8 + mt : mt;
          mma(acc[mt][0], wq[0], af[ai]); ops1(std::integral_constant<int, 4 * mt + 0>{});
          mma(acc[mt][1], wq[1], af[ai]); ops1(std::integral_constant<int, 4 * mt + 1>{});
          mma(acc[mt][2], wq[2], af[ai]); ops1(std::integral_constant<int, 4 * mt + 2>{});
          mma(acc[mt][3], wq[3], af[ai]); ops1(std::integral_constant<int, 4 * mt + 3>{});
          if constexpr ((NMX_DABLATE & 4) == 0) {
            if constexpr (mt + 4 < 8) af[mt + 4] = *reinterpret_cast<const u32x4*>(buf + a_rd1 + (mt + 4) * 2048);
          }
          dma_at(std::integral_constant<int, 8 + mt>{});
          self(self, std::integral_constant<int, mt + 1>{});
        }
      };
      pass1(pass1, std::integral_constant<int, 0>{});
    }
  };
  for (int it = 0; it < per; it += 3) {  // `per` is the same for every wave of the workgroup: the barriers pair up
    body(I0{}, it);
    if (it + 1 >= per) break;
    body(I1{}, it + 1);
    if (it + 2 >= per) break;
    body(I2{}, it + 2);
  }
  // nothing may land in LDS after this point (clamped DMAs past the range included): the ring becomes the reduction buffer
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // the s_nop covers the MFMA -> VALU read distance that hipcc does not know about (the MFMAs are asm statements)
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) asm volatile("s_nop 7" : "+v"(acc[mt][0]), "+v"(acc[mt][1]), "+v"(acc[mt][2]), "+v"(acc[mt][3]));
  if constexpr ((NMX_DABLATE & 512) != 0) { if (acc[0][0][0] != 12345.678f) return; }  // no epilogue

  // ---- sum the two K-groups through LDS; K-group 0 stores ----
  __syncthreads();
  if constexpr (!LS) {
    float* red = reinterpret_cast<float*>(smem);
    if (kg == 1) {
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(red + ((wn * 32 + mt * 4 + t) * 64 + lane) * 4) = acc[mt][t];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mt][t] += *reinterpret_cast<const f32x4*>(red + ((wn * 32 + mt * 4 + t) * 64 + lane) * 4);
    }
  }
  // ---- channel-wise scales on the fp32 sums (D row 4 g + r of tile t = column 16 t + 4 g + r of the group;
  //      scale_perm_single: position 32 (b >> 2) + 8 (cc >> 1) + (cc & 1) + 2 (b & 3) holds column cc + 8 b) ----
  if constexpr (!SCALED) {
    if (kg == 0) {
      const f16* sc = reinterpret_cast<const f16*>(p.scales) + nl;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = 16 * t + 4 * g + r;
          const int cc = col & 7, bb = col >> 3;
          const float sv = (float)sc[32 * (bb >> 2) + 8 * (cc >> 1) + (cc & 1) + 2 * (bb & 3)];
#pragma unroll
          for (int mt = 0; mt < 8; ++mt) {
            acc[mt][t][r] *= sv;
            asm volatile("" : "+v"(acc[mt][t][r]));  // a rounding step of its own in every form of the op (see marlin_decode_kernel)
          }
        }
    }
  }
  if (fuse_act) {
    // silu_and_mul on the fp16-ROUNDED gate and up values, the arithmetic of act_and_mul_kernel (reference
    // activation_kernels.cu:12-30): out = f16(silu(float(gate))) * up, rounded once more
    constexpr int HW = kWN / 2;
    u32x2* ex = reinterpret_cast<u32x2*>(smem);
    __syncthreads();  // reduction slabs are free
    const int pair = (wn % HW) * (32 * 64);
    if (kg == 0 && wn >= HW) {
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          union { f16 h[4]; u32x2 u; } r;
#pragma unroll
          for (int j = 0; j < 4; ++j) r.h[j] = (f16)acc[mt][t][j];
          ex[pair + (mt * 4 + t) * 64 + lane] = r.u;
        }
    }
    __syncthreads();
    u32x2 ov[8][4];
    if (kg == 0 && wn < HW) {
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          union { f16 h[4]; u32x2 u; } up, o;
          up.u = ex[pair + (mt * 4 + t) * 64 + lane];
#pragma unroll
          for (int j = 0; j < 4; ++j) o.h[j] = rnd_mul<f16>(silu_rnd<f16>((f16)acc[mt][t][j]), up.h[j]);
          ov[mt][t] = o.u;
        }
    }
    __syncthreads();  // the exchange buffer becomes the transpose images
    if (kg != 0 || wn >= HW || !col_ok) return;
    store_tile16(smem + wn * (128 * kTRow), ov, reinterpret_cast<f16*>(p.act_out) + (int64_t)m0 * (N / 2) + n0, N / 2, M - m0, lane);
    return;
  }
  if constexpr (!LS) __syncthreads();  // the reduction slabs become the transpose images (LS: nothing was staged there)
  if (kg != 0 || !col_ok) return;
  if constexpr ((NMX_DABLATE & 8192) != 0) { if (acc[0][0][0] != 12345.678f) return; }  // no output stores (the K-group reduce stays)
  if (p.k_splits == 1 || p.partial_f16) {
    // 16-bit outputs (the result, or an fp16 slab): whole 128-byte lines through the wave's transpose image
    u32x2 ov[8][4];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        union { f16 h[4]; u32x2 u; } r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.h[j] = (f16)acc[mt][t][j];
        ov[mt][t] = r.u;
      }
    f16* dst = p.k_splits == 1 ? reinterpret_cast<f16*>(p.c) + (int64_t)m0 * N + n0
                               : reinterpret_cast<f16*>(p.partial) + ((int64_t)split_id * M + m0) * N + n0;
    store_tile16(smem + wn * (128 * kTRow), ov, dst, N, M - m0, lane);
    return;
  }
  // fp32 slabs (NMX_SLAB_F32): lane (g, li): D rows = the 4 consecutive output columns 16 t + 4 g + r, D col = activation row li
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int m = m0 + mt * 16 + li;
    if (m >= M) continue;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = n0 + 16 * t + 4 * g;
      *reinterpret_cast<f32x4*>(p.partial + ((int64_t)split_id * M + m) * N + n) = acc[mt][t];
    }
  }
}


// ---- producer / consumer form (round 3, late) -----------------------------------------------------------------------------
// marlin_dma_kernel's valid timing ablations (tools/dma_ablate.sh) showed its cost classes ADD: skeleton
// (LDS reads, barrier, waits) + conversion VALU + MFMA + DMA issue, ~4.2 instructions per MFMA in every wave, and two waves
// of the SAME program on a SIMD overlap their MFMA and VALU phases only by accident. A SIMD co-issues an MFMA and a VALU
// instruction only from DIFFERENT waves - so here the two waves of a SIMD are different programs:
//   * waves 0..3 (CONSUMERS, one per SIMD; wave wn owns the 128 x 64 tile of column group wn over the WHOLE K range of the
//     workgroup) run no conversion at all: per 64-k stage 16 activation-fragment reads + 8 weight-fragment reads
//     (ds_read_b128) + 64 MFMAs, and they issue the stage's four activation DMAs of their row quarter;
//   * waves 4..7 (PRODUCERS, the other wave of each SIMD; wave wn serves column group wn) DMA the packed words + scales,
//     read them back one stage later, run the whole int4 -> fp16 conversion (112 VALU per stage) and park the eight MFMA A
//     fragments of the NEXT stage in an LDS fragment buffer in consumer lane order (8 ds_write_b128, linear: no conflicts).
// One barrier per stage; ring of three stage slots (activations + packed words), two fragment buffers. No K-groups inside the
// workgroup (the consumers walk all of its stages), so no LDS reduction in the epilogue; cross-workgroup K splits as before.
constexpr int kFBuf = kWN * 8 * 64 * 16;      // [column group][fragment 4 p + t][lane] x 16 B = 32 KiB

#if (NMX_DABLATE & 65536)
#define PC_MMA(c, a, b) (c)[0] += __builtin_bit_cast(float, (a)[1] ^ (b)[2])
#else
#define PC_MMA(c, a, b) a_mfma_f16(c, a, b)
#endif
template <bool SCALED>
__global__ __launch_bounds__(512, 2) void marlin_pc_kernel(const DmaParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave & (kWN - 1);
  const bool producer = wave >= kWN;
  const int g = lane >> 4, li = lane & 15, hb = (lane >> 3) & 1, c8 = lane & 7;
  const int N = p.N, K = p.K, M = p.M;

  const int m_blocks = (M + kBM - 1) / kBM;
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
  if (tile_x * kWN * 64 >= N) return;  // padding workgroup
  const bool fuse_act = p.act_out != nullptr;
  const int n0 = fuse_act ? (wn >= kWN / 2 ? N / 2 : 0) + (tile_x * (kWN / 2) + (wn % (kWN / 2))) * 64 : (tile_x * kWN + wn) * 64;
  const bool col_ok = n0 < N;
  const int nl = col_ok ? n0 : 0;
  const int m0 = block_m * kBM;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const fbuf = smem + kNBUF * kSlot;

  const int total_stages = K / 64;
  const int per = (total_stages + p.k_splits - 1) / p.k_splits;
  const int st_begin = min(split_id * per, total_stages), st_end = min(st_begin + per, total_stages);
  const int nst = st_end - st_begin;
  const int st_last = min(st_begin + max(nst - 1, 0), total_stages - 1);

  const int row_bytes = N * 8;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, M * K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.b), 0, (K / 16) * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.scales), 0, p.num_groups * N * 2, 0x00020000);
  int a_voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 8 * (4 * wn + j) + (lane >> 3);
    a_voff[j] = (min(m0 + r, M - 1) * K + 8 * ((lane & 7) ^ a_swz(r))) * 2;
  }
  const int w_voff = ((lane >> 3) & 1) * row_bytes + (nl / 64) * 512 + (4 * (lane & 7) + (lane >> 4)) * 16;
  const int s_voff = (nl + 2 * (lane & 31)) * 2;
  const int gs_shift = SCALED ? (31 - __builtin_clz((unsigned)max(p.group_size / 64, 1))) : 0;
  constexpr int NW = 2 + (SCALED ? 1 : 0);  // producer DMA instructions per stage (consumers: 4)

  auto issue_a1 = [&](auto j_c, auto sa_c, int rel) {  // consumer: activation DMA j of walk position rel into slot SA
    constexpr int J = decltype(j_c)::value, SA = decltype(sa_c)::value;
#if defined(__HIP_DEVICE_COMPILE__)
    const int sa = min(st_begin + rel, st_last);
    auto* da = (__attribute__((address_space(3))) char*)(ring + SA * kSlot + (4 * wn + J) * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, da, 16, a_voff[J], sa * 128, 0, 0);
#endif
  };
  auto issue_a = [&](auto sa_c, int rel) {
    issue_a1(std::integral_constant<int, 0>{}, sa_c, rel);
    issue_a1(std::integral_constant<int, 1>{}, sa_c, rel);
    issue_a1(std::integral_constant<int, 2>{}, sa_c, rel);
    issue_a1(std::integral_constant<int, 3>{}, sa_c, rel);
  };
  auto issue_w = [&](auto sw_c, int rel) {  // producer: packed words (+ scales) of walk position rel into slot SW
    constexpr int SW = decltype(sw_c)::value;
#if defined(__HIP_DEVICE_COMPILE__)
    const int sw = min(st_begin + rel, st_last);
    auto* dw = (__attribute__((address_space(3))) char*)(ring + SW * kSlot + kAImg + wn * 2048);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, dw, 16, w_voff, (4 * sw) * row_bytes, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, dw + 1024, 16, w_voff, (4 * sw + 2) * row_bytes, 0, 0);
    if constexpr (SCALED) {
      const int grp = min(sw >> gs_shift, p.num_groups - 1);
      auto* ds = (__attribute__((address_space(3))) char*)(ring + SW * kSlot + kAImg + kWImg + wn * 256);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, ds, 4, s_voff, grp * N * 2, 0, 0);
    }
#endif
  };

  const int a_rd0 = li * 128 + (((2 * g) ^ a_swz(li)) * 16), a_rd1 = li * 128 + (((2 * g + 1) ^ a_swz(li)) * 16);
  const int w_rd = kAImg + wn * 2048 + (g >> 1) * 1024 + ((g & 1) * 8 + c8) * 16;
  const int s_rd = kAImg + kWImg + wn * 256 + c8 * 16;
  const int f_off = (wn * 8 * 64 + lane) * 16;  // this lane's 16 bytes of fragment 0 of its column group; fragment f at + 1024 f

  const uint32_t magic = 0x64006400u, neg72 = 0xd480d480u;
  const uint32_t shv = 8u * (uint32_t)hb;
  const uint32_t ssel = hb ? 0x03020302u : 0x01000100u;

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  // ---- producer: packed words of ring slot SLOT -> the 8 fp16 fragments of that stage in fragment buffer fb ----
  auto produce = [&](auto slot_c, char* fb) {
    constexpr int SLOT = decltype(slot_c)::value;
    const char* buf = ring + SLOT * kSlot;
    uint32_t s2[4] = {0, 0, 0, 0};
    if constexpr (SCALED) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(buf + s_rd);
#pragma unroll
      for (int t = 0; t < 4; ++t) s2[t] = __builtin_amdgcn_perm(v[t], v[t], ssel);
    }
    uint32_t raw[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(buf + w_rd + 256 * m);
#pragma unroll
      for (int t = 0; t < 4; ++t) raw[m][t] = v[t] >> shv;
    }
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        uint32_t w[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if constexpr ((NMX_DABLATE & 16384) != 0) { w[m] = raw[m][t] ^ s2[t]; continue; }  // ablation: no conversion
          a_and_or(w[m], raw[m][t], pp == 0 ? 0x000f000fu : 0x00f000f0u, magic);
          if (pp == 0) a_pk_add(w[m], 0xe408e408u);          // (1024 + q) - 1032
          else a_pk_fma(w[m], 0x2c002c00u, neg72);           // (1024 + 16 q) / 16 - 72
          if constexpr (SCALED) a_pk_mul(w[m], s2[t]);
        }
        *reinterpret_cast<u32x4*>(fb + f_off + (4 * pp + t) * 1024) = u32x4{w[0], w[1], w[2], w[3]};
      }
    }
  };

  // ---- producers: their own loop (same barrier count as the consumers'), then out through the epilogue's barriers ----
  if (producer) {
    issue_w(I0{}, 0);
    issue_w(I1{}, 1);
    issue_w(I2{}, 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NW) : "memory");  // this wave's own DMA of stage 0
    produce(I0{}, fbuf);
    auto pbody = [&](auto cur_c, int it) {
      constexpr int CUR = decltype(cur_c)::value, NXT = (CUR + 1) % kNBUF;
      // the fragments of stage it are written (lgkmcnt), the packed words of it + 1 have landed (this wave's own DMA)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NW) : "memory");
      __builtin_amdgcn_s_barrier();
      issue_w(std::integral_constant<int, CUR>{}, it + 3);  // this slot's packed words were converted last iteration
      produce(std::integral_constant<int, NXT>{}, fbuf + ((it + 1) & 1) * kFBuf);
    };
    for (int it = 0; it < per; it += 3) {
      pbody(I0{}, it);
      if (it + 1 >= per) break;
      pbody(I1{}, it + 1);
      if (it + 2 >= per) break;
      pbody(I2{}, it + 2);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (fuse_act) __syncthreads();
    return;
  }

  // ---- consumers ----
  f32x4 acc[8][4];
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  issue_a(I0{}, 0);
  issue_a(I1{}, 1);

  // Software pipeline across the barrier: the first fragment reads of a stage can only be issued behind its barrier, and with
  // ONE consumer wave per SIMD nothing else feeds the matrix pipe meanwhile - so the last two MFMA rows of pass 1 of every
  // stage are held back (operands stay in registers: t_af, t_wq) and run behind the NEXT stage's barrier, right after that
  // stage's first eight fragment reads have been issued. Pass 1's weight fragments are read during pass 0.
  u32x4 t_af[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
  u32x4 t_wq[4] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
  auto tail = [&]() {  // rows 6, 7 of pass 1 of the previous stage (zeros before the first stage: adds nothing)
#pragma unroll
    for (int t = 0; t < 4; ++t) PC_MMA(acc[6][t], t_wq[t], t_af[0]);
#pragma unroll
    for (int t = 0; t < 4; ++t) PC_MMA(acc[7][t], t_wq[t], t_af[1]);
  };
  auto body = [&](auto cur_c, int it) {
    constexpr int CUR = decltype(cur_c)::value, PRV = (CUR + 2) % kNBUF;
    // the activation rows of stage it this wave issued have landed; behind the barrier everybody's rows and the producers'
    // fragments of stage it are visible
    if constexpr ((NMX_DABLATE & 32768) == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* const fb_cur = fbuf + (it & 1) * kFBuf;
    const char* const buf = ring + CUR * kSlot;
    if (it >= nst) {  // out of stages (uneven split): keep the DMA / barrier pattern; flush the held-back rows once
      issue_a(std::integral_constant<int, PRV>{}, it + 2);
      tail();
      t_af[0] = u32x4{0, 0, 0, 0};
      t_af[1] = u32x4{0, 0, 0, 0};
      return;
    }
    u32x4 af[12];
    u32x4 wq[4], wq1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) wq[t] = *reinterpret_cast<const u32x4*>(fb_cur + f_off + t * 1024);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const u32x4*>(buf + a_rd0 + mt * 2048);
    __builtin_amdgcn_sched_barrier(0);
    tail();
    __builtin_amdgcn_sched_barrier(0);
    auto pass0 = [&](auto self, auto mt_c) {
      constexpr int mt = decltype(mt_c)::value;
      if constexpr (mt < 8) {
        PC_MMA(acc[mt][0], wq[0], af[mt]);
        PC_MMA(acc[mt][1], wq[1], af[mt]);
        PC_MMA(acc[mt][2], wq[2], af[mt]);
        PC_MMA(acc[mt][3], wq[3], af[mt]);
        if constexpr (mt + 4 < 8) af[mt + 4] = *reinterpret_cast<const u32x4*>(buf + a_rd0 + (mt + 4) * 2048);
        else af[8 + (mt - 4)] = *reinterpret_cast<const u32x4*>(buf + a_rd1 + (mt - 4) * 2048);
        if constexpr (mt < 4) wq1[mt] = *reinterpret_cast<const u32x4*>(fb_cur + f_off + (4 + mt) * 1024);  // pass 1's weight fragments
        __builtin_amdgcn_sched_barrier(0);  // the fragment reads stay 4 rows ahead of their MFMAs (hoisted to the top they spill)
        if constexpr ((mt & 1) == 0 && (NMX_DABLATE & 32768) == 0) {  // the four activation DMAs of stage it + 2, one behind every second MFMA row
          issue_a1(std::integral_constant<int, (mt >> 1)>{}, std::integral_constant<int, PRV>{}, it + 2);
          __builtin_amdgcn_sched_barrier(0);
        }
        self(self, std::integral_constant<int, mt + 1>{});
      }
    };
    pass0(pass0, std::integral_constant<int, 0>{});
    auto pass1 = [&](auto self, auto mt_c) {
      constexpr int mt = decltype(mt_c)::value;
      if constexpr (mt < 6) {  // rows 6, 7 are held back (tail)
        constexpr int ai = mt < 4 ? 8 + mt : mt;
        PC_MMA(acc[mt][0], wq1[0], af[ai]);
        PC_MMA(acc[mt][1], wq1[1], af[ai]);
        PC_MMA(acc[mt][2], wq1[2], af[ai]);
        PC_MMA(acc[mt][3], wq1[3], af[ai]);
        if constexpr (mt + 4 < 8) af[mt + 4] = *reinterpret_cast<const u32x4*>(buf + a_rd1 + (mt + 4) * 2048);
        __builtin_amdgcn_sched_barrier(0);
        self(self, std::integral_constant<int, mt + 1>{});
      }
    };
    pass1(pass1, std::integral_constant<int, 0>{});
    t_af[0] = af[6];
    t_af[1] = af[7];
#pragma unroll
    for (int t = 0; t < 4; ++t) t_wq[t] = wq1[t];
  };
  for (int it = 0; it < per; it += 3) {
    body(I0{}, it);
    if (it + 1 >= per) break;
    body(I1{}, it + 1);
    if (it + 2 >= per) break;
    body(I2{}, it + 2);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  tail();  // the last stage's held-back rows
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) asm volatile("s_nop 7" : "+v"(acc[mt][0]), "+v"(acc[mt][1]), "+v"(acc[mt][2]), "+v"(acc[mt][3]));
  __syncthreads();  // nothing lands in LDS any more; the ring is free for the epilogue

  if constexpr (!SCALED) {
    {
      const f16* sc = reinterpret_cast<const f16*>(p.scales) + nl;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = 16 * t + 4 * g + r;
          const int cc = col & 7, bb = col >> 3;
          const float sv = (float)sc[32 * (bb >> 2) + 8 * (cc >> 1) + (cc & 1) + 2 * (bb & 3)];
#pragma unroll
          for (int mt = 0; mt < 8; ++mt) {
            acc[mt][t][r] *= sv;
            asm volatile("" : "+v"(acc[mt][t][r]));
          }
        }
    }
  }
  if (fuse_act) {
    constexpr int HW = kWN / 2;
    u32x2* ex = reinterpret_cast<u32x2*>(smem);
    const int pair = (wn % HW) * (32 * 64);
    if (wn >= HW) {
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          union { f16 h[4]; u32x2 u; } r;
#pragma unroll
          for (int j = 0; j < 4; ++j) r.h[j] = (f16)acc[mt][t][j];
          ex[pair + (mt * 4 + t) * 64 + lane] = r.u;
        }
    }
    __syncthreads();
    if (wn >= HW || !col_ok) return;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int m = m0 + mt * 16 + li;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        union { f16 h[4]; u32x2 u; } up, o;
        up.u = ex[pair + (mt * 4 + t) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) o.h[j] = rnd_mul<f16>(silu_rnd<f16>((f16)acc[mt][t][j]), up.h[j]);
        if (m < M) *reinterpret_cast<u32x2*>(reinterpret_cast<f16*>(p.act_out) + (int64_t)m * (N / 2) + n0 + 16 * t + 4 * g) = o.u;
      }
    }
    return;
  }
  if (!col_ok) return;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int m = m0 + mt * 16 + li;
    if (m >= M) continue;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = n0 + 16 * t + 4 * g;
      if (p.k_splits == 1) {
        union { f16 h[4]; u32x2 u; } r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.h[j] = (f16)acc[mt][t][j];
        *reinterpret_cast<u32x2*>(reinterpret_cast<f16*>(p.c) + (int64_t)m * N + n) = r.u;
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

}  // namespace

// One launch of marlin_dma_kernel with `splits` K splits across workgroups (the caller has sized the scratch). The split-K
// reduce launch / deferral stays with the caller (nmx_wide_launch).
static int nmx_dma_launch(NmxWideCall& call, int splits, int xcd_split, hipStream_t stream) {
  DmaParams p;
  p.a = call.a; p.b = call.b; p.scales = call.scales; p.c = call.c; p.partial = reinterpret_cast<float*>(call.scratch);
  p.M = call.M; p.N = call.N; p.K = call.K; p.num_groups = call.num_groups; p.group_size = call.group_size;
  p.k_splits = splits; p.xcd_split = xcd_split;
  p.partial_f16 = (splits > 1 && nmx_tune(NMX_TUNE_SLAB_F32) == nullptr) ? 1 : 0;  // fp16 outputs only reach this kernel
  call.act_done = (call.act_out != nullptr && splits == 1 && call.N % 2 == 0 && (call.N / 2) % (64 * kWN) == 0) ? 1 : 0;
  p.act_out = call.act_done ? call.act_out : nullptr;
  const size_t ring = (size_t)kWK * kNBUF * kSlot;
  const size_t red = (size_t)kWN * 32 * 64 * 4 * sizeof(float);
  const size_t smem = std::max(ring, red);
  dim3 grid(ceil_div(ceil_div(p.N, 64 * kWN), 8) * 8 * ceil_div(p.M, kBM), splits, 1);
  bool ls = false;  // NMX_GEMM_DMA = "splits,1": the loader / consumer split; "splits,2": marlin_pc_kernel
  int mode = 0;
  if (const char* e = nmx_tune(NMX_TUNE_GEMM_DMA)) { int a = 0, b = 0; if (sscanf(e, "%d,%d", &a, &b) == 2) mode = b; }
  ls = mode == 1;
  if (mode == 2) {
    const size_t pc_smem = (size_t)kNBUF * kSlot + 2 * kFBuf;
    if (call.num_groups > 1) {
      auto kern = marlin_pc_kernel<true>;
      NMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pc_smem));
      kern<<<grid, 512, pc_smem, stream>>>(p);
    } else {
      auto kern = marlin_pc_kernel<false>;
      NMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pc_smem));
      kern<<<grid, 512, pc_smem, stream>>>(p);
    }
    NMX_LAUNCH_CHECK();
    call.splits_done = splits | (p.partial_f16 ? NMX_SPLITK_F16 : 0);
    return NMX_OK;
  }
#define NMX_DMA_LAUNCH(SC, LSV)                                                                                                   \
  {                                                                                                                               \
    auto kern = marlin_dma_kernel<SC, LSV>;                                                                                       \
    NMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));     \
    kern<<<grid, 512, smem, stream>>>(p);                                                                                         \
  }
  if (call.num_groups > 1) { if (ls) NMX_DMA_LAUNCH(true, true) else NMX_DMA_LAUNCH(true, false) }
  else { if (ls) NMX_DMA_LAUNCH(false, true) else NMX_DMA_LAUNCH(false, false) }
#undef NMX_DMA_LAUNCH
  NMX_LAUNCH_CHECK();
  call.splits_done = splits | (p.partial_f16 ? NMX_SPLITK_F16 : 0);
  return NMX_OK;
}


// fp16 int4, channel-wise or 64-multiple groups, N a multiple of 64, K of 64; 32-bit offsets
static bool dma_supported(int M, int N, int K, int num_groups, int group_size, int kind, int is_bf16) {
  if (kind != 0 || is_bf16) return false;
  if (K % 64 != 0 || N % 64 != 0 || M < 1) return false;
  if (num_groups > 1 && (group_size % 64 != 0 || ((group_size / 64) & (group_size / 64 - 1)) != 0)) return false;  // 64, 128, 256, ...
  if ((int64_t)M * K * 2 >= (1ll << 31) || (int64_t)K * N >= (1ll << 31) || (int64_t)num_groups * N * 2 >= (1ll << 31)) return false;
  return true;
}

bool nmx_dma_pick(int M, int N, int K, int num_groups, int group_size, int kind, int is_bf16, int* splits) {
  if (!dma_supported(M, N, K, num_groups, group_size, kind, is_bf16)) return false;
  const int stages = K / 64;
  int forced = -1;
  if (const char* e = nmx_tune(NMX_TUNE_GEMM_DMA)) forced = atoi(e);
  if (forced == 0) return false;
  int sp = 1;
  if (forced > 0) {
    sp = forced;
    while (sp > 1 && stages / (sp * kWK) < 2) sp /= 2;  // at least two stages per K-group
    *splits = sp;
    return true;
  }
  // Default rule, from tools/lean_sweep.py (32-launch graph chains over distinct weights, deferred reduce; gpurun_out/
  // dma_sweep_s*.txt, M = 128 .. 2048 on the four Llama-3-8B shapes): the kernel runs where its 128 x 256 tiles times 1 / 2 / 4 /
  // 8 K splits give 160 .. 256 workgroups with >= 8 stages per K-group - there it is 8-16 % faster than marlin_wide_kernel /
  // the 64-row tiles on the matrices that need K splits (down at M = 256: 35.4 vs 38.6-41.6 us, qkv at M = 512: 37.1 vs 44.4, o at
  // M = 512 / 1024: 26.0 / 39.7 vs 28.8 / 45.8, down at M = 512: 59.4 vs 68.3) and level (+-2 %) where the tiles alone fill the
  // chip (gate_up from M = 256, everything at M = 2048). One row block (M <= 128) and short K per split (o at M = 256: 8
  // splits x 4 stages 18.3-19.4 vs 17.3-18.8 us) stay with the older kernels. (Round 3, late: a 64-row form of this kernel - MT = 4,
  // the conversion ops four per MFMA - was built and measured for M <= 64: a K-group stage costs ~2,100 cycles against ~3,000 for
  // 128 rows, so it only levels with marlin_gemm_kernel - qkv 13.3 vs 13.2 us, gate_up 26.3 vs 27.4, o 11.9 vs 9.4, down 21.5 vs
  // 19.0 at M = 64 - and the batch-64 step was slower with it, 4.69 vs 4.56 ms, because two K splits lose gate_up's fused
  // activation epilogue; removed again. gpurun_out/dma_small2.txt is quoted in profiles/r03_dma_sweep.txt.)
  if (M <= 128) return false;
  // batch <= 256 of the decode step: only the long-K matrices. In the step the slabs of a K split are summed by the consumer op,
  // so MORE splits than the older dispatch takes cost there what they save here (bench.py A/B at batch 256 with qkv on 4
  // splits instead of 2: step 11.15 vs 11.09 ms); down_proj runs 8 splits either way and is 8-10 % faster on this kernel.
  // (round 3, late, with fp16 slabs and whole-line stores: qkv on 4 splits here is 22.8 vs 24.8 us on the 64-row wide tiles, and the
  // step still does not gain - 11.11 vs 11.09 ms; unchanged)
  if (M <= 256 && K < 8192) return false;
  // an explicit override of one of the older kernels (sweeps, their tests) keeps this one out of the way
  if (nmx_tune(NMX_TUNE_GEMM_WIDE) != nullptr || nmx_tune(NMX_TUNE_GEMM_CFG) != nullptr || nmx_tune(NMX_TUNE_GEMM_LARGE) != nullptr) return false;
  const int tiles = ceil_div(N, 64 * kWN) * ceil_div(M, kBM);
  while (sp < 8 && tiles * sp * 2 <= 256 && stages / (sp * 2 * kWK) >= 8) sp *= 2;
  if (tiles * sp < 160) return false;
  *splits = sp;
  return true;
}

int nmx_dma_run(NmxWideCall& call, int splits, hipStream_t stream) {
  if (splits > 1) {  // never allocate here (graph capture): degrade to the splits that fit
    const int64_t per = (int64_t)call.M * call.N * sizeof(float);
    const int fit = call.scratch == nullptr ? 1 : (int)std::min<int64_t>(splits, call.scratch_bytes / per);
    splits = std::max(1, fit);
  }
  int xcd = 1;
  if (const char* e = nmx_tune(NMX_TUNE_GEMM_XCD_SPLIT)) xcd = atoi(e) != 0;
  const int rc = nmx_dma_launch(call, splits, xcd, stream);
  if (rc != NMX_OK) return rc;
  if (splits > 1 && !call.defer_reduce)
    return nmx_splitk_reduce(call.c, reinterpret_cast<const float*>(call.scratch), call.splits_done, call.M, call.N, NMX_F16, (nmx_stream_t)stream);
  return NMX_OK;
}
