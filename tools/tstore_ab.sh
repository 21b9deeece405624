#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for lib in $root/neuralmagic_vllm_amd/libnmx_hip.so $root/exp/libnmx_tstore0.so; do
  echo "== $(basename $lib)"
  NMX_LIB_PATH=$lib LEAN_SWEEP_DEFER=1 LEAN_SWEEP_ONLY="X:1" LEAN_SWEEP_SHAPES=gate_up timeout -k 10 200 python3 $root/tools/lean_sweep.py 256 2>&1 | grep -v amdgpu.ids | sed "s/^ *//" | cut -c1-95
  NMX_LIB_PATH=$lib LEAN_SWEEP_DEFER=1 LEAN_SWEEP_ONLY="X:8;X:4" LEAN_SWEEP_SHAPES=down,o,qkv timeout -k 10 200 python3 $root/tools/lean_sweep.py 256 2>&1 | grep -v amdgpu.ids | sed "s/^ *//" | cut -c1-95
done; done
