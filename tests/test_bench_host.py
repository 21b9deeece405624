"""Host-side checks of bench.py that need no GPU: the multi-GPU launcher refuses to start more ranks than GPUs are visible
(and says so in one JSON line), and the TP sharding arithmetic of the 70B AWQ config follows SURVEY section 8e."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_without_gpus_reports_and_fails():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert "error" in rec and rec["n_gpus"] == 0


def test_tp_shard_of_llama3_70b():
    sys.path.insert(0, ROOT)
    import bench
    cfg = dict(bench.VARIANTS["awq70b"]["model"])
    full_heads, full_kv, inter = cfg["heads"], cfg["kv_heads"], cfg["inter"]
    sh = bench.tp_shard(cfg, 8)
    assert sh["heads"] == full_heads // 8 and sh["kv_heads"] == max(1, full_kv // 8) and sh["inter"] == inter // 8
    assert sh["hidden"] == cfg["hidden"]  # row-parallel outputs are all-reduced to the full hidden size
    sh16 = bench.tp_shard(cfg, 16) if full_heads % 16 == 0 else None
    if sh16 is not None:
        assert sh16["kv_heads"] == 1  # kv heads are replicated once tp exceeds their count
