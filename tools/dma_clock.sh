#!/bin/bash
# GPU box: in-kernel clock of marlin_dma_kernel variants (exp/libnmx_dab{128+x}.so: bit 7 = stamps)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for lib in $(ls exp/libnmx_dab1[23]*.so | sort -V); do
  for cfg in "X:1" "X:1,1"; do
    echo "== $lib $1 $cfg"
    NMX_LIB_PATH=$root/$lib timeout -k 10 100 python3 tools/gemm_one.py ${1:-gate_up} 256 "$cfg" 40 2>&1 | grep "dma dbg" | tail -2
  done
done
