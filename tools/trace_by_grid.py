"""Per (kernel, grid size) duration statistics from a rocprofv3 --kernel-trace csv (tells the shapes of one kernel apart).
usage: python3 tools/trace_by_grid.py <dir or kernel_trace.csv> [name filter]"""
import collections
import csv
import glob
import os
import sys

path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            key = (r["Kernel_Name"][:70], r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"),
                   r.get("Workgroup_Size_X", "?"))
            acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{k[0]:70} grid=({k[1]},{k[2]},{k[3]}) wg={k[4]} n={len(v):5d} median={v[len(v)//2]:8.2f} us min={v[0]:8.2f}")
