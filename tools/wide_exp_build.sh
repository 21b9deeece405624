#!/bin/bash
# Builds experiment variants of libnmx_hip.so that differ only in marlin_wide.o (fp16 int4 instantiations only):
#   tools/wide_exp_build.sh NAME "<-D flags>"   ->  exp/libnmx_NAME.so   (travels to the GPU box; git-ignored)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/exp
name=$1; flags=$2
cs=$root/neuralmagic_vllm_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-function -Wno-unused-variable \
  -DNMX_WIDE_MIN $flags -c $cs/marlin_wide.hip -o $root/exp/wide_$name.o
objs=$(ls $cs/*.o | grep -v marlin_wide.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/exp/libnmx_$name.so $objs $root/exp/wide_$name.o
rm -f $root/exp/wide_$name.o
echo built exp/libnmx_$name.so
