#!/bin/bash
# GPU box: tools/mm_sweep.py against prebuilt scaled_mm_tile_kernel experiment variants (exp/libnmx_tab<mask>.so)
# usage: tools/mm_ablate.sh "<Ms>" "<cfgs>" "<shapes>" "<masks>"
root=${GRAFT_REPO_ROOT:-/root/repo}
for m in $4; do
  lib=$root/exp/libnmx_tab$m.so; [ "$m" = 0 ] && lib=$root/neuralmagic_vllm_amd/libnmx_hip.so
  echo "== ablate mask $m"
  NMX_LIB_PATH=$lib timeout -k 10 300 python3 $root/tools/mm_sweep.py "$1" "$2" "$3" 2>&1 | grep -v amdgpu.ids
done
