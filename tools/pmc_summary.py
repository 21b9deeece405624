"""Post-processes rocprofv3 output directories (kernel trace + one PMC counter per pass) into a per-kernel summary:
average duration, launches, FETCH_SIZE / WRITE_SIZE per launch with the gfx950 correction of MI355X_MICROARCH.md
(FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> x2; rocprofv3 reports both in KiB).

usage: python3 tools/pmc_summary.py STATS_DIR FETCH_DIR WRITE_DIR OUT.json [BENCH_LOG [SQ_DIR]]
(BENCH_LOG: stdout of the profiled bench.py run; its JSON line names the workload the summary belongs to)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def rows(d, suffix):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def short(name):
    n = name
    for key in ("marlin_dma_kernel", "marlin_pc_kernel", "marlin_gemm_kernel", "marlin_wide_kernel", "marlin_decode_kernel", "marlin_large_kernel", "paged_attention_kernel", "paged_attention_fp8w_kernel", "paged_attention_v2_reduce_kernel", "splitk_reduce_kernel",
                "rms_norm_splitk_kernel", "silu_and_mul_splitk_kernel", "rope_cache_kernel", "prefill_attention_shared_kernel", "prefill_attention_kernel",
                "rms_norm_vec_kernel", "rotary_kernel", "act_and_mul_kernel", "reshape_and_cache_vec_kernel",
                "reshape_and_cache_kernel", "Cijk", "gptq_gemm_kernel", "awq_gemm_kernel", "scaled_mm_tile_kernel", "scaled_mm_lds_kernel",
                "scaled_mm_reduce_kernel", "scaled_mm_kernel", "fp8_quant_kernel", "fp8_absmax_kernel", "moe_scaled_mm_kernel"):
        if key in n:
            return key
    return n[:60]


def main():
    stats_dir, fetch_dir, write_dir, out = sys.argv[1:5]
    dur = defaultdict(list)
    for r in rows(stats_dir, "kernel_trace.csv"):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    cnt = {}
    for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        acc = defaultdict(list)
        for r in rows(d, "counter_collection.csv"):
            if r.get("Counter_Name") == name:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        cnt[name] = acc
    sq = defaultdict(lambda: defaultdict(list))
    if len(sys.argv) > 6:
        for r in rows(sys.argv[6], "counter_collection.csv"):
            sq[short(r["Kernel_Name"])][r.get("Counter_Name")].append(float(r["Counter_Value"]))
    total = sum(sum(v) for v in dur.values())
    summary = {}
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        e = {"launches": len(v), "avg_us": sum(v) / len(v) / 1e3, "share_of_gpu_time": sum(v) / total}
        f, w = cnt["FETCH_SIZE"].get(k), cnt["WRITE_SIZE"].get(k)
        if f:
            e["fetch_bytes_per_launch_corrected"] = 2.0 * 1024.0 * sum(f) / len(f)  # KiB -> B, x2 (gfx950 half-count)
        if w:
            e["write_bytes_per_launch"] = 1024.0 * sum(w) / len(w)
        c = sq.get(k)
        if c and c.get("SQ_BUSY_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            mean = lambda name: sum(c[name]) / len(c[name])
            # SQ_BUSY_CYCLES is summed over the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
            e["mfma_busy_cycles_per_launch"] = mean("SQ_VALU_MFMA_BUSY_CYCLES")
            e["mfma_util"] = mean("SQ_VALU_MFMA_BUSY_CYCLES") / (mean("SQ_BUSY_CYCLES") / 32.0 * 1024.0)
            if c.get("SQ_WAVE_CYCLES"):
                wc = mean("SQ_WAVE_CYCLES")
                for name, key in (("SQ_WAIT_ANY", "wave_cycles_waiting"), ("SQ_WAIT_INST_ANY", "wave_cycles_issue_stalled"),
                                  ("SQ_ACTIVE_INST_ANY", "wave_cycles_issuing")):
                    if c.get(name):
                        e[key] = mean(name) / wc
        summary[k] = e
    if len(sys.argv) > 5:
        for line in open(sys.argv[5]):
            if line.startswith("{") and '"metric"' in line:
                c = json.loads(line)["config"]
                summary["_workload"] = {"batch": c["batch_per_gpu"], "ctx": c["context"], "layers": c["layers"]}
    json.dump(summary, open(out, "w"), indent=1)
    for k, e in [kv for kv in summary.items() if not kv[0].startswith('_')][:14]:
        print(f"{k:36} n={e['launches']:6d} avg={e['avg_us']:9.2f} us share={e['share_of_gpu_time']:.3f} "
              f"fetch={e.get('fetch_bytes_per_launch_corrected', 0) / 1e6:9.2f} MB write={e.get('write_bytes_per_launch', 0) / 1e6:8.2f} MB"
              + (f" mfma_util={e['mfma_util']:.3f}" if "mfma_util" in e else ""))


if __name__ == "__main__":
    main()
