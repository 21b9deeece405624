#!/bin/bash
# Builds experiment variants of libnmx_hip.so that differ only in one translation unit (third argument, default marlin_dma):
#   tools/dma_exp_build.sh NAME "<-D flags>"   ->  exp/libnmx_NAME.so   (travels to the GPU box; git-ignored)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/exp
name=$1; flags=$2; EXP_TU=${3:-marlin_dma}
cs=$root/neuralmagic_vllm_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-function -Wno-unused-variable \
  $flags -c $cs/$EXP_TU.hip -o $root/exp/x_$name.o
objs=$(ls $cs/*.o | grep -v $EXP_TU.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/exp/libnmx_$name.so $objs $root/exp/x_$name.o
rm -f $root/exp/x_$name.o
echo built exp/libnmx_$name.so
