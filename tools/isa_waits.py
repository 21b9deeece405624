#!/usr/bin/env python3
"""For one kernel in a hipcc -S file: the s_waitcnt vmcnt(N) values, buffer loads and MFMAs per basic block of its loops.
usage: tools/isa_waits.py file.s mangled-name-substring"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and key in l and re.match(r"^_Z\w+:", l))
end = next(i for i in range(start + 1, len(src)) if src[i].startswith(".Lfunc_end"))
blk = None
stats = {}
order = []
for l in src[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blk = m.group(1)
        stats[blk] = dict(n=0, mfma=0, vmem=0, lds=0, valu=0, salu=0, waits=[], branch=None)
        order.append(blk)
        continue
    if blk is None or not l.startswith("\t") or l.startswith("\t."):
        continue
    ins = l.strip().split()[0]
    s = stats[blk]
    s["n"] += 1
    if "mfma" in ins or "smfmac" in ins:
        s["mfma"] += 1
    elif ins.startswith("buffer_") or ins.startswith("global_"):
        s["vmem"] += 1
    elif ins.startswith("ds_"):
        s["lds"] += 1
    elif ins.startswith("v_"):
        s["valu"] += 1
    elif ins == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", l)
        if m:
            s["waits"].append(int(m.group(1)))
    elif ins.startswith("s_cbranch") or ins == "s_branch":
        s["branch"] = l.strip().split()[-1]
    elif ins.startswith("s_"):
        s["salu"] += 1
for b in order:
    s = stats[b]
    if s["mfma"] or s["vmem"] > 2:
        print(f"{b:12s} n={s['n']:5d} mfma={s['mfma']:4d} vmem={s['vmem']:3d} lds={s['lds']:3d} valu={s['valu']:4d} salu={s['salu']:4d} -> {s['branch']}  vmcnt={s['waits']}")
