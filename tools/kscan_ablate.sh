#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
for m in 0 256 512 8192; do
  lib=$root/exp/libnmx_dab$m.so; [ "$m" = 0 ] && lib=$root/neuralmagic_vllm_amd/libnmx_hip.so
  echo "== DABLATE $m"
  NMX_LIB_PATH=$lib LEAN_SWEEP_SET=kscan LEAN_SWEEP_DEFER=1 LEAN_SWEEP_ONLY="X:1" timeout -k 10 200 python3 $root/tools/lean_sweep.py 256 2>&1 | grep -v amdgpu.ids | sed "s/^ *//" | cut -c1-90
done
