#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_marlin_dma_gpu.py -m gpu -x -q 2>&1 | tail -2
for lib in "" exp/libnmx_norot.so; do
  if [ -n "$lib" ]; then export NMX_LIB_PATH=$root/$lib; else unset NMX_LIB_PATH; fi
  LEAN_SWEEP_DEFER=1 LEAN_SWEEP_SHAPES=gate_up,down LEAN_SWEEP_ONLY="D:auto;X:1;X:8" timeout -k 10 300 python3 tools/lean_sweep.py 256 2>&1 | grep -v amdgpu.ids
done > gpurun_out/dma_rot.txt
cat gpurun_out/dma_rot.txt
unset NMX_LIB_PATH
timeout -k 10 300 python3 bench.py --steps 10 > gpurun_out/bench_r3c.log 2>&1; tail -1 gpurun_out/bench_r3c.log | cut -c1-1500
