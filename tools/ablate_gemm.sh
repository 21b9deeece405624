#!/bin/bash
# GPU box: builds timing-ablation variants of the GEMM kernel and times one configuration under rocprofv3.
# usage: tools/ablate_gemm.sh SHAPE M CFG "0 1 2 4 8 16 32 ..."   -> gpurun_out/ablate_<shape>_<M>_<cfg>.txt
set -o pipefail
shape=$1; M=$2; cfg=$3; masks=$4
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ablate_${shape}_${M}_${cfg//,/-}.txt
: > $out
cd /tmp && export TMPDIR=/tmp
src=$root/neuralmagic_vllm_amd/csrc
for m in $masks; do
  lib=/tmp/libnmx_ab$m.so
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -DNMX_ABLATE=$m -shared -o $lib \
     $src/marlin_gemm.hip $src/nmx_runtime.hip 2>/dev/null || { echo "build failed $m" >> $out; continue; }
  rm -rf /tmp/ab_$m
  NMX_LIB_PATH=$lib timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab_$m -- python3 $root/tools/gemm_one.py $shape $M $cfg 20 > /tmp/ab_$m.log 2>&1
  python3 - $m /tmp/ab_$m >> $out <<'PY'
import csv, glob, sys
m, d = sys.argv[1], sys.argv[2]
v = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "marlin_gemm_kernel" in r["Kernel_Name"]:
            v.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
v.sort()
print(f"ablate={int(m):3d} n={len(v)} median={v[len(v)//2]/1e3 if v else -1:.2f} us")
PY
done
cat $out
