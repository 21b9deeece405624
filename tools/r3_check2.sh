#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
python3 tools/dbg_fused.py > gpurun_out/dbg_fused_new.log 2>&1
NMX_LIB_PATH=$root/exp/libnmx_r02.so python3 tools/dbg_fused.py > gpurun_out/dbg_fused_r02.log 2>&1
python3 - <<'PY' >> gpurun_out/dbg_fused_new.log 2>&1
import torch
a=torch.load("gpurun_out/dbg_new.pt"); b=torch.load("gpurun_out/dbg_r02.pt")
print("plain new==r02", torch.equal(a["out"], b["out"]), "fused new==r02", torch.equal(a["one"], b["one"]))
PY
cat gpurun_out/dbg_fused_new.log gpurun_out/dbg_fused_r02.log
timeout -k 10 900 python3 -m pytest tests/test_marlin_dma_gpu.py -m gpu -x -q > gpurun_out/r3_dma_tests.log 2>&1
tail -15 gpurun_out/r3_dma_tests.log
