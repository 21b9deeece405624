#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_marlin_dma_gpu.py -m gpu -x -q > gpurun_out/r3_check4_tests.log 2>&1
tail -3 gpurun_out/r3_check4_tests.log
LEAN_SWEEP_DEFER=1 LEAN_SWEEP_ONLY="${SWEEP_CFGS:-D:auto;X:1;X:2;X:4;X:8}" timeout -k 10 600 python3 tools/lean_sweep.py ${SWEEP_MS:-256} 2>&1 | grep -v amdgpu.ids > gpurun_out/dma_sweep_${1:-x}.txt
cat gpurun_out/dma_sweep_${1:-x}.txt
