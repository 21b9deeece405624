#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of bench.py for each batch given; per-kernel table into gpurun_out/stats_<batch>.txt
# usage: tools/stats_only.sh "<bench args>" batch...
set -o pipefail
root=${GRAFT_REPO_ROOT:-/root/repo}
extra=$1; shift
cd /tmp && export TMPDIR=/tmp
for b in "$@"; do
  rm -rf /tmp/ps_$b
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$b -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --batch $b $extra > $root/gpurun_out/stats_$b.log 2>&1 || exit 1
  python3 - $(find /tmp/ps_$b -name "*kernel_stats.csv" | head -1) > $root/gpurun_out/stats_$b.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print(f'{r["Name"][:90]:90s} n={int(r["Calls"]):6d} avg={float(r["AverageNs"])/1e3:8.2f}us share={float(r["TotalDurationNs"])/tot:.3f}')
PY
  tail -1 $root/gpurun_out/stats_$b.log
done
