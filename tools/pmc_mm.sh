#!/bin/bash
# usage (GPU box): tools/pmc_mm.sh TAG LIB CFG M N K   -> gpurun_out/pmcmm_TAG.txt : scaled_mm kernel durations + SQ counters
# (LIB = path of the library variant or "-" for the in-tree one; CFG = NMX_MM_TILE value or "D")
set -o pipefail
tag=$1; lib=$2; cfg=$3; shift 3
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
[ "$lib" != "-" ] && export NMX_LIB_PATH=$lib
[ "$cfg" != "D" ] && export NMX_MM_TILE=$cfg
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/q1 /tmp/q2 /tmp/q3 /tmp/q4 /tmp/q5
log=$out/pmcmm_${tag}_run.log
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/q1 -- python3 $root/tools/mm_one.py "$@" fp8 20 > $log 2>&1
timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/q2 -- python3 $root/tools/mm_one.py "$@" fp8 20 >> $log 2>&1
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/q3 -- python3 $root/tools/mm_one.py "$@" fp8 20 >> $log 2>&1
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/q4 -- python3 $root/tools/mm_one.py "$@" fp8 20 >> $log 2>&1
timeout -k 10 120 rocprofv3 --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d /tmp/q5 -- python3 $root/tools/mm_one.py "$@" fp8 20 >> $log 2>&1
python3 - "$out/pmcmm_${tag}.txt" <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[1], "w")
def rows(d, suf):
    r = []
    for f in glob.glob(d + "/**/*" + suf, recursive=True):
        r += list(csv.DictReader(open(f)))
    return r
dur = collections.defaultdict(list)
for r in rows("/tmp/q1", "kernel_trace.csv"):
    if "scaled_mm" in r["Kernel_Name"]:
        dur[r["Kernel_Name"][:90]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in dur.items():
    v.sort()
    print(f"{k:90} n={len(v)} median={v[len(v)//2]/1e3:.2f}us min={v[0]/1e3:.2f}", file=out)
for d in ("/tmp/q2", "/tmp/q3", "/tmp/q4", "/tmp/q5"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(d, "counter_collection.csv"):
        acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        if "scaled_mm_tile" not in k:
            continue
        print(k, file=out)
        for name, v in sorted(c.items()):
            print(f"   {name:28} {sum(v)/len(v):16.0f}", file=out)
out.close()
print(open(sys.argv[1]).read())
PY
