#!/bin/bash
# GPU box: full GPU test suite + short result table (after the DMA kernel became the default for M > 128)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3_check5_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_check5_tests.log
tail -5 gpurun_out/r3_check5_tests.log
out=$root/gpurun_out/results_r3b.txt
: > $out
run() {
  cfg=$1; b=$2; shift 2
  timeout -k 10 300 python3 $root/bench.py --config $cfg --batch $b --steps 10 --no-cpu-baseline "$@" > /tmp/rt.log 2>&1
  python3 - "$cfg $* dma=${NMX_GEMM_DMA:-default}" $b >> $out <<'PY'
import json, sys
ln = [l for l in open("/tmp/rt.log") if l.startswith("{")]
if not ln:
    print(sys.argv[1], sys.argv[2], "FAILED"); sys.exit(0)
r = json.loads(ln[-1])
k = r.get("kernels", {})
ks = " ".join(f"{n}={v['us']}" for n, v in k.items())
print(f"{sys.argv[1]:30} batch {int(sys.argv[2]):4d}  {r['value']:9.1f} tok/s  {r['ms_per_step']:7.3f} ms  frac={r['roofline']['frac']:.3f}  {ks}")
PY
}
for b in 128 192 256; do run int4 $b; done
export NMX_GEMM_DMA=0; run int4 256; unset NMX_GEMM_DMA
run int4 256
export NMX_GEMM_DMA=0; run int4 256; unset NMX_GEMM_DMA
cat $out
