set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_marlin_wide_gpu.py tests/test_fused_gpu.py -x -q -m gpu -k "wide or gate_up" > gpurun_out/t_ring.log 2>&1 || { tail -30 gpurun_out/t_ring.log; exit 1; }
tail -2 gpurun_out/t_ring.log
for c in "gate_up 64 auto" "gate_up 48 auto" "qkv 256 auto" "o 256 auto" "qkv 128 auto" "o 128 auto"; do
  echo "== $c"; tools/wide_libs.sh $c || exit 1
done
for b in 64 256; do
  for l in ring4 ring8; do
    echo "batch $b $l: $(NMX_LIB_PATH=$PWD/exp/libnmx_$l.so timeout -k 10 200 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])')"
  done
done
