#!/bin/bash
# GPU box: the end-of-round result table (BASELINE.md / DESIGN.md section 5): every bench config at its batch points.
# usage: tools/result_table.sh TAG   -> gpurun_out/results_TAG.txt (one "config batch tok/s ms" line per run)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/results_$1.txt
: > $out
run() {
  cfg=$1; b=$2; shift 2
  timeout -k 10 300 python3 $root/bench.py --config $cfg --batch $b --steps 10 --no-cpu-baseline "$@" > /tmp/rt.log 2>&1
  python3 - "$cfg $*" $b >> $out <<'PY'
import json, sys
ln = [l for l in open("/tmp/rt.log") if l.startswith("{")]
if not ln:
    print(sys.argv[1], sys.argv[2], "FAILED"); sys.exit(0)
r = json.loads(ln[-1])
k = r.get("kernels", {})
ks = " ".join(f"{n}={v['us']}" for n, v in k.items())
print(f"{sys.argv[1]:24} batch {int(sys.argv[2]):4d}  {r['value']:9.1f} tok/s  {r['ms_per_step']:7.3f} ms  frac={r['roofline']['frac']:.3f}  {ks}")
PY
}
for b in 1 8 16 32 64 128 256; do run int4 $b; done
for b in 1 64 256; do run int4 $b --no-fuse; done
for b in 64 256; do run sparse24 $b; done
for b in 64 256; do run fp8 $b; done
for b in 64 256; do run awq70b-tp8rank $b; done
for b in 1 64 256; do run gptq-exllama $b; done
cat $out
