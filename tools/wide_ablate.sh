#!/bin/bash
# usage (GPU box): tools/wide_ablate.sh SHAPE M CFG  -> one timing line per exp/libnmx_ab*.so
for lib in $(ls exp/libnmx_ab*.so | sort -V); do
  NMX_LIB_PATH=$PWD/$lib timeout -k 10 120 python3 tools/gemm_time.py $1 $2 $3 || exit 1
done
