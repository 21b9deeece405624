#!/bin/bash
# usage (GPU box): tools/wide_libs.sh SHAPE M CFG  -> one timing line per exp/libnmx_*.so
for lib in $(ls exp/libnmx_*.so | sort -V); do
  NMX_LIB_PATH=$PWD/$lib timeout -k 10 120 python3 tools/gemm_time.py $1 $2 $3 2>&1 | grep -v amdgpu.ids || exit 1
done
