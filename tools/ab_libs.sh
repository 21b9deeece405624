# GPU box: same-box A/B of every exp/libnmx_*.so (built by hand / tools/wide_exp_build.sh): parity of the wide-tile and fused
# paths with the in-tree library first, then the whole decode step per library.   usage: [BATCHES="64 256"] bash tools/ab_libs.sh
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_marlin_wide_gpu.py tests/test_fused_gpu.py tests/test_dispatch_fuzz_gpu.py -x -q -m gpu > gpurun_out/t_ab.log 2>&1 || { tail -30 gpurun_out/t_ab.log; exit 1; }
tail -1 gpurun_out/t_ab.log
for b in ${BATCHES:-64 256}; do
  for l in $(ls exp/libnmx_*.so | sort -V); do
    echo "batch $b $l: $(NMX_LIB_PATH=$PWD/$l timeout -k 10 200 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])')"
  done
done
