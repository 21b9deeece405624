#!/bin/bash
# GPU box: kernel-trace timing of the prebuilt scaled_mm_tile_kernel ablation variants (exp/libnmx_tab<mask>.so)
# usage: tools/ablate_mm.sh M N K "0 1 2 ..."
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for m in $4; do
  rm -rf /tmp/ab_$m
  lib=$root/exp/libnmx_tab$m.so; [ "$m" = 0 ] && lib=$root/neuralmagic_vllm_amd/libnmx_hip.so
  NMX_LIB_PATH=$lib timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab_$m -- python3 $root/tools/mm_one.py $1 $2 $3 fp8 20 > /tmp/ab_$m.log 2>&1
  python3 - $m /tmp/ab_$m <<'PY'
import csv, glob, sys
v = []
for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scaled_mm_tile" in r["Kernel_Name"]:
            v.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
v.sort()
print(f"ablate={int(sys.argv[1]):3d} n={len(v)} median={v[len(v)//2]/1e3 if v else -1:.2f} us")
PY
done
