"""-m gpu: bench.py as the driver runs it -- one JSON line on stdout with the contract's fields -- on one GPU, and the
N-rank code path (communicator, native gather, per-rank breakdown) rehearsed as a world of one under the launcher."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = dict(os.environ)
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    env.update(extra)
    return env


def test_bench_line_has_the_contract_fields():
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--cpu-seconds", "8"],
                          capture_output=True, text=True, env=_clean_env(), timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout                        # exactly ONE line on stdout
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "scaling_detail"):
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 4 and rec["warmup"] == 1 and rec["unit"] == "Mrays/s"
    assert rec["metric"].startswith("Mrays/sec + ms/frame, scene_import_dragon.json @1920x1080")
    assert "workload" in rec["config"] and "model" not in rec["config"]
    assert rec["vs_baseline"] is None and rec["higher_is_better"] is True and rec["dtype"] == "f64"
    assert rec["value"] > 1000 and abs(rec["value"] - rec["config"]["rays_per_frame"] / rec["ms_per_step"] / 1e3) < 0.02 * rec["value"]
    roof = rec["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert 0 < roof["frac"] <= 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["kernel"].startswith("ct_walk_") and roof["launches_timed"] >= 4
    # frac is the undisturbed launch (what profiles/ reproduces); the in-flight figure beside it, never better
    assert roof["launch_ms_inflight"] >= 0.9 * roof["launch_ms"] and 0 < roof["frac_inflight"] <= roof["frac"] * 1.1
    assert abs(roof["achieved"] - roof["bytes_per_launch"] / roof["launch_ms"] / 1e6) < 0.01 * roof["achieved"]
    assert roof["limiter"]["what"] == "valu-issue"
    ref = roof["reference_bytes"]
    assert ref["primary_walk"]["bytes"] + ref["shadow_walk"]["bytes"] <= ref["per_frame"]
    assert ref["primary_walk"]["rays"] + ref["shadow_walk"]["rays"] == rec["config"]["reference_rays_per_frame"]
    assert len(rec["build_id"]["libctgpu"]) == 16 and len(rec["build_id"]["libcthost"]) == 16
    cpu = rec["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["unit"] == "Mrays/s" and cpu["sample"]
    assert rec["parity"]["differing_pixels"] == 0 and rec["parity"]["parity_max_lsb"] <= 1
    ranks = rec["scaling_detail"]["ranks"]
    assert len(ranks) == 1 and ranks[0]["rays"] == rec["config"]["rays_per_frame"]


def test_rank_code_path_as_a_world_of_one():
    """torch.distributed.run with one rank + CT_BENCH_FORCE_DIST: process group, unique id broadcast, ctgpu_comm_init,
    ctgpu_render_sharded (ncclGather on the frame's stream), the reductions and the per-rank gather of bench.py."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--nproc-per-node", "1",
           os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    proc = subprocess.run(cmd, capture_output=True, text=True, env=_clean_env(CT_BENCH_FORCE_DIST="1"), timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    rec = json.loads([l for l in proc.stdout.splitlines() if l.lstrip().startswith('{"metric"')][-1])
    assert "issued by libctgpu.so" in rec["config"]["collective"]
    rank0 = rec["scaling_detail"]["ranks"][0]
    assert rank0["trace_ms"] > 0 and rank0["gather_ms"] >= 0 and rank0["assemble_ms"] > 0 and rank0["d2h_ms"] > 0
