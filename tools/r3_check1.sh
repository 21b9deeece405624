#!/bin/bash
# GPU box: round-3 first check - the GPU tests of the files touched by the ring-load rewrite + a short result table
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests/test_marlin_gpu.py tests/test_marlin24_gpu.py tests/test_zp_gpu.py tests/test_marlin_decode_gpu.py \
  tests/test_custom_ar_gpu.py tests/test_quant_gpu.py tests/test_tp_rccl_gpu.py tests/test_dispatch_fuzz_gpu.py tests/test_fused_gpu.py \
  tests/test_marlin_wide_gpu.py -m gpu -x -q > gpurun_out/r3_check1_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_check1_tests.log
tail -5 gpurun_out/r3_check1_tests.log
out=$root/gpurun_out/results_r3a.txt
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
for b in 1 16 32 64 128 256; do run int4 $b; done
for b in 64 256; do run sparse24 $b; done
run awq70b-tp8rank 64
cat $out
