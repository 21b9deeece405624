#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_marlin_dma_gpu.py tests/test_dispatch_fuzz_gpu.py tests/test_fused_gpu.py -m gpu -x -q > gpurun_out/r3_check3_tests.log 2>&1
tail -4 gpurun_out/r3_check3_tests.log
LEAN_SWEEP_DEFER=1 LEAN_SWEEP_ONLY="D:auto;X:1;X:2;X:4;X:8" timeout -k 10 600 python3 tools/lean_sweep.py 128 256 > gpurun_out/dma_sweep1.txt 2>&1
cat gpurun_out/dma_sweep1.txt
