#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of `bench.py` and writes the summaries
# under gpurun_out/profiles_<tag>/ (copy what should be judged into profiles/).
# usage: tools/profile_round.sh r01 [bench args...]
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write /tmp/prof_sq
args="--steps 3 --warmup 1 --no-cpu-baseline $@"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $root/bench.py $args > $out/bench_stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python3 $root/bench.py $args > $out/bench_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python3 $root/bench.py $args > $out/bench_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/prof_sq -- python3 $root/bench.py $args > $out/bench_sq.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
python3 $root/tools/pmc_summary.py /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write $out/${tag}_kernel_summary.json $out/bench_stats.log /tmp/prof_sq | tee $out/${tag}_kernel_summary.txt
