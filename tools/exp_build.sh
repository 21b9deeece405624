#!/bin/bash
# GPU box: builds an experimental libnmx_hip.so variant of the GEMM translation units with extra -D flags into /tmp
# and runs a command against it (NMX_LIB_PATH). usage: [EXP_SRCS="a.hip b.hip"] tools/exp_build.sh "<-D flags>" <command...>
set -o pipefail
flags=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
src=$root/neuralmagic_vllm_amd/csrc
lib=/tmp/libnmx_exp_$$.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast $flags -shared -o $lib \
  ${EXP_SRCS:-$src/marlin_gemm.hip $src/nmx_runtime.hip} 2>/dev/null || { echo "exp build failed"; exit 1; }
NMX_LIB_PATH=$lib "$@"
