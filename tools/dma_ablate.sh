#!/bin/bash
# GPU box: timing ablations of marlin_dma_kernel (results wrong by construction): one line per exp/libnmx_dab*.so
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for lib in "" $(ls exp/libnmx_dab*.so | sort -V); do
  if [ -n "$lib" ]; then export NMX_LIB_PATH=$root/$lib; fi
  LEAN_SWEEP_DEFER=1 LEAN_SWEEP_SHAPES=${1:-o,gate_up} LEAN_SWEEP_ONLY="${ABL_CFG:-X:1}" timeout -k 10 200 python3 tools/lean_sweep.py 256 2>&1 | grep -v amdgpu.ids
done
