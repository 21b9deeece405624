#!/bin/bash
# GPU box: times prebuilt ablation variants (exp/libnmx_ab<mask>.so, built on the CPU box) of one GEMM configuration from
# the rocprofv3 kernel trace (host-side event timing is launch-bound for kernels this short).
# usage: tools/ablate_prebuilt.sh SHAPE M CFG "0 1 2 ..."   -> gpurun_out/ablate_<shape>_<M>.txt
set -o pipefail
shape=$1; M=$2; cfg=$3; masks=$4
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ablate_${shape}_${M}.txt
: > $out
cd /tmp && export TMPDIR=/tmp
for m in $masks; do
  rm -rf /tmp/ab_$m
  NMX_LIB_PATH=$root/exp/libnmx_ab$m.so timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab_$m -- python3 $root/tools/gemm_one.py $shape $M $cfg 20 > /tmp/ab_$m.log 2>&1
  python3 - $m /tmp/ab_$m $shape $M >> $out <<'PY'
import csv, glob, sys, collections
m, d = sys.argv[1], sys.argv[2]
v = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "marlin" in n or "splitk" in n:
            v[n.split("<")[0].split("(")[0][-28:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
s = " ".join(f"{k}: n={len(x)} med={sorted(x)[len(x)//2]/1e3:.2f}us" for k, x in v.items())
print(f"{sys.argv[3]} M={sys.argv[4]} ablate={int(m):3d} {s}")
PY
done
cat $out
