#!/bin/bash
# GPU box (usage: tools/round_check.sh [TAG]): full GPU test suite, then the round profile (kernel stats + PMC passes of bench.py at the default batch)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3_round_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_round_tests.log
tail -4 gpurun_out/r3_round_tests.log
bash tools/profile_round.sh ${1:-r03} > gpurun_out/r3_profile.log 2>&1
tail -30 gpurun_out/r3_profile.log
