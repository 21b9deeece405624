#!/bin/bash
# usage (GPU box): tools/wide_times.sh M "CFG CFG ..."  -> timing lines for the four Llama-3-8B shapes
M=$1; shift
for shape in qkv o gate_up down; do
  for cfg in "$@"; do timeout -k 10 120 python3 tools/gemm_time.py $shape $M $cfg 2>&1 | grep -v amdgpu.ids || exit 1; done
done
