"""bench.py prints one JSON line with the keys its contract names (run here on a small frame)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--width", "320", "--height", "180",
                        "--spp", "32", "--cpu-spp", "1"], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in out, key
    assert out["unit"] == "Mpaths/s" and out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1
    assert out["higher_is_better"] is True and out["scaling"] == "strong" and out["vs_baseline"] is None and out["data"] == "synthetic"
    assert "workload" in out["config"] and "model" not in out["config"]
    assert out["value"] > 0 and abs(out["value"] - 320 * 180 * 32 / (out["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * out["value"]
    roof = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    # the bound names the measured limiter: HBM, a matrix-core contraction, the L2s' gather rate, or (this kernel) vector instruction issue
    assert roof["bound"] in ("hbm", "mfma", "l2", "valu")
    assert roof["kernel"] == "k_sdf_march" and roof["bound"] == "valu" and roof["peak"] > 0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert 0.0 < roof["frac"] <= 1.0
    cpu = out["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1
    # the GPU frame against the CPU reference frame at the same sample count (here 1 spp on a small frame: only the keys)
    for key in ("gpu_vs_cpu", "cpu_seedA_vs_seedB", "box8_rel", "crop", "accepted", "accepted_rmse"):
        assert key in out["rmse"], key


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: two child processes, one per rank (both on cuda:0 here, the
    all-reduce over gloo), the 32-spp frame divided between them; rank 0 prints the line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "1", "--warmup", "1", "--width", "320",
                        "--height", "180", "--spp", "32", "--no-cpu-baseline"], capture_output=True, text=True, timeout=280,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and "REHEARSAL" in out["data"]
    assert "32 spp per frame (16 per GPU)" in out["config"]["workload"]
    assert out["weak"]["spp_per_frame"] == 64 and out["weak"]["value"] > 0
    assert abs(out["value"] - 320 * 180 * 32 / (out["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * out["value"]
