cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pf8
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf8 -- python3 $GRAFT_REPO_ROOT/bench.py --config fp8 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pf8/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scaled_mm" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            acc[(r["Kernel_Name"][:75], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "n=%d" % len(v), "fetch MB (x2 corrected) = %.1f" % (2 * 1024 * sum(v) / len(v) / 1e6))
PY
