# GPU box: parity of the one-launch gate_up + silu_and_mul op, then same-box A/B of the decode step with / without it
# usage: [BATCHES="1 8 64"] bash tools/ab_act_fuse.sh
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_fused_gpu.py -x -q -m gpu -k "gate_up" > gpurun_out/t_gateup.log 2>&1 || { tail -30 gpurun_out/t_gateup.log; exit 1; }
tail -2 gpurun_out/t_gateup.log
for b in ${BATCHES:-64 128 256}; do
  for f in "" "--no-act-fuse"; do
    echo "batch $b $f: $(timeout -k 10 200 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline $f 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])')"
  done
done
