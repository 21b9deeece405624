# GPU box: parity of the kernels that split K + A/B of NMX_GEMM_XCD_SPLIT (timing, FETCH_SIZE per launch)
# usage: tools/ab_xcd_split.sh "SHAPE M" ...   (default: down 256, qkv 256, o 256)
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_fused_gpu.py tests/test_marlin_decode_gpu.py tests/test_marlin_wide_gpu.py -x -q -m gpu > gpurun_out/t_xcd.log 2>&1 || { tail -30 gpurun_out/t_xcd.log; exit 1; }
tail -1 gpurun_out/t_xcd.log
shapes=("$@"); [ ${#shapes[@]} -eq 0 ] && shapes=("down 256" "qkv 256" "o 256")
for v in 1 0; do
  for c in "${shapes[@]}"; do
    echo "xcd_split=$v $(NMX_GEMM_XCD_SPLIT=$v timeout -k 10 120 python3 tools/gemm_time.py $c auto 2>&1 | grep -v amdgpu.ids)"
  done
done
for b in 128 256; do
  for v in 1 0; do
    echo "batch $b xcd_split=$v: $(NMX_GEMM_XCD_SPLIT=$v timeout -k 10 200 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])')"
  done
done
cd /tmp && export TMPDIR=/tmp
for c in "${shapes[@]}"; do
for v in 1 0; do
  rm -rf /tmp/pf$v
  NMX_GEMM_XCD_SPLIT=$v timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf$v -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py $c auto 10 > /dev/null 2>&1
  python3 - /tmp/pf$v $v "$c" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE": acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "marlin" in k: print(f"{sys.argv[3]} xcd_split={sys.argv[2]} {k} FETCH x2 = {2 * sum(v) / len(v) / 1e3:.1f} MB per launch (n={len(v)})")
PY
done
done
