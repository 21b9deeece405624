#!/bin/bash
# usage (GPU box): tools/pmc_mm2.sh TAG LIB CFG M N K   -> gpurun_out/pmcmm2_TAG.txt : memory-path counters (TA / TCP / TCC) of scaled_mm
set -o pipefail
tag=$1; lib=$2; cfg=$3; shift 3
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
[ "$lib" != "-" ] && export NMX_LIB_PATH=$lib
[ "$cfg" != "D" ] && export NMX_MM_TILE=$cfg
cd /tmp && export TMPDIR=/tmp
log=$out/pmcmm2_${tag}_run.log; : > $log
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_REQUEST_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1)); rm -rf /tmp/r$i
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d /tmp/r$i -- python3 $root/tools/mm_one.py "$@" fp8 12 >> $log 2>&1
done
rm -rf /tmp/r0; timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/r0 -- python3 $root/tools/mm_one.py "$@" fp8 12 >> $log 2>&1
python3 - "$out/pmcmm2_${tag}.txt" $i <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[1], "w")
def rows(d, suf):
    r = []
    for f in glob.glob(d + "/**/*" + suf, recursive=True):
        r += list(csv.DictReader(open(f)))
    return r
dur = collections.defaultdict(list)
for r in rows("/tmp/r0", "kernel_trace.csv"):
    if "scaled_mm" in r["Kernel_Name"]:
        dur[r["Kernel_Name"][:90]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in dur.items():
    v.sort()
    print(f"{k:90} n={len(v)} median={v[len(v)//2]/1e3:.2f}us min={v[0]/1e3:.2f}", file=out)
for i in range(1, int(sys.argv[2]) + 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(f"/tmp/r{i}", "counter_collection.csv"):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        if "scaled_mm_tile" not in k and "scaled_mm_bdirect" not in k:
            continue
        for name, v in sorted(c.items()):
            print(f"   {name:44} {sum(v)/len(v):16.0f}", file=out)
out.close()
print(open(sys.argv[1]).read())
PY
