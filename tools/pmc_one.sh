#!/bin/bash
# usage (GPU box): tools/pmc_one.sh TAG SHAPE M CFG   -> gpurun_out/pmc_TAG.txt : kernel duration + SQ counters
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/p1 -- python3 $root/tools/gemm_one.py "$@" > $out/pmc_${tag}_run.log 2>&1
timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/p2 -- python3 $root/tools/gemm_one.py "$@" >> $out/pmc_${tag}_run.log 2>&1
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/p3 -- python3 $root/tools/gemm_one.py "$@" >> $out/pmc_${tag}_run.log 2>&1
rm -rf /tmp/p4; timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/p4 -- python3 $root/tools/gemm_one.py "$@" >> $out/pmc_${tag}_run.log 2>&1
python3 - "$out/pmc_${tag}.txt" <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[1], "w")
def rows(d, suf):
    r = []
    for f in glob.glob(d + "/**/*" + suf, recursive=True):
        r += list(csv.DictReader(open(f)))
    return r
dur = collections.defaultdict(list)
for r in rows("/tmp/p1", "kernel_trace.csv"):
    dur[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in dur.items():
    v.sort()
    print(f"{k:70} n={len(v)} median={v[len(v)//2]/1e3:.2f}us min={v[0]/1e3:.2f}", file=out)
for d in ("/tmp/p2", "/tmp/p3", "/tmp/p4"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(d, "counter_collection.csv"):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        if "marlin" not in k and "reduce" not in k and "large" not in k:
            continue
        print(k, file=out)
        for name, v in sorted(c.items()):
            print(f"   {name:28} {sum(v)/len(v):16.0f}", file=out)
out.close()
print(open(sys.argv[1]).read())
PY
