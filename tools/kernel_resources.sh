#!/bin/bash
# Prints VGPR / spill / LDS / occupancy per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_resources.sh neuralmagic_vllm_amd/csrc/marlin_gemm.hip
f=$1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -c "$f" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur={}
for line in sys.stdin:
    m=re.search(r"remark: [^ ]+ +(.*?): +(.*?) \[-Rpass", line)
    if not m:
        m=re.search(r"(Function Name|Name): (\S+)", line)
        if m:
            if cur: print(cur)
            cur={"name":m.group(2)[:90]}
        continue
' ; /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Name:|VGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*(Name: [^ ]*|VGPRs: [0-9]+|VGPRs Spill: [0-9]+|SGPRs Spill: [0-9]+|ScratchSize \[bytes\/lane\]: [0-9]+|Occupancy \[waves\/SIMD\]: [0-9]+|LDS Size \[bytes\/block\]: [0-9]+).*/\1/' | paste -sd' ' | sed 's/Name: /\n/g' | awk 'NF' | sed -E 's/_ZN12_GLOBAL__N_1[0-9]+//' | cut -c1-200
