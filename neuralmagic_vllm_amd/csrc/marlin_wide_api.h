#pragma once
// Internal (not part of the C-ABI) hand-over between marlin_gemm.hip and marlin_wide.hip: the wide-tile kernel is its
// own translation unit so that the two heavy template families compile in parallel.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct NmxWideCall {
  const void* a;          // [M, K] fp16 / bf16
  const int32_t* b;       // Marlin-packed weight
  const void* scales;     // [num_groups, N], Marlin-permuted
  void* c;                // [M, N]
  void* scratch;          // fp32 split-K partials (may be null)
  int64_t scratch_bytes;
  int M, N, K;
  int num_groups, group_size;
  int kind;               // WeightKind
  int is_bf16;
  int defer_reduce;       // leave split-K partials in scratch (no reduce launch)
  int splits_done;        // out: K splits the launch used
  void* act_out = nullptr;  // [M, N / 2]: fuse silu_and_mul into the epilogue when the launch has no K split (see act_done)
  int act_done = 0;         // out: act_out was written (and c was not)
  const void* meta = nullptr;  // 2:4-sparse weights (gptq_marlin_24_gemm): the reordered metadata tensor; b is then the compressed tensor
  const void* zeros = nullptr; // AWQ on the Marlin layout (awq_marlin_gemm): [num_groups, N] fp16 -(1024 + z), permuted like the scales
};

// tile configuration of marlin_wide_kernel: wm x wn x wk waves, `splits` K splits across workgroups
struct NmxWideCfg { int wm, wn, wk, splits, mt; };  // mt = 16-row tiles per wave (8; 4 = 64-row wave tiles)

// true when the wide kernel handles this problem (M large enough, plain layout); fills the configuration
__attribute__((visibility("hidden"))) bool nmx_wide_pick(int M, int N, int K, int num_groups, int group_size, NmxWideCfg* cfg, int kind = 0, bool sparse = false);  // kind: WeightKind (0 = int4)
__attribute__((visibility("hidden"))) int nmx_wide_launch(NmxWideCall& call, const NmxWideCfg& cfg, hipStream_t stream);

// marlin_dma_kernel (marlin_dma.hip): fp16 int4 GEMM with both operands delivered by LDS-DMA. nmx_dma_pick: true when it
// should serve this problem (NMX_GEMM_DMA=0 disables it, =N forces it with N K splits); nmx_dma_run launches it (+ the
// split-K reduce unless call.defer_reduce) and fills call.splits_done / call.act_done.
__attribute__((visibility("hidden"))) bool nmx_dma_pick(int M, int N, int K, int num_groups, int group_size, int kind, int is_bf16, int* splits);
__attribute__((visibility("hidden"))) int nmx_dma_run(NmxWideCall& call, int splits, hipStream_t stream);
