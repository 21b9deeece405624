#!/usr/bin/env python3
"""Compact per-kernel resource table from `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr saved to a file).
usage: tools/res_summary.py res.txt [name-regex]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
cur = None
rows = []
for line in txt.splitlines():
    m = re.search(r"remark: +(?:Function )?Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    if cur is None:
        continue
    for key, rx in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                    ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)")):
        m = re.search(rx, line)
        if m:
            cur[key] = int(m.group(1))
names = [r["name"] for r in rows]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, dem):
    d = re.sub(r"\(anonymous namespace\)::", "", d)
    d = re.sub(r"^void ", "", d)
    d = re.sub(r"\(.*\)$", "", d)
    if pat and not pat.search(d):
        continue
    print(f"{d[:110]:110s} v={r.get('vgpr')} a={r.get('agpr')} s={r.get('sgpr')} scr={r.get('scratch')} spill={r.get('vspill')} occ={r.get('occ')}")
