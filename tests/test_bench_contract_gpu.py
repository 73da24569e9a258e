"""bench.py prints ONE JSON line with the fields the driver reads (small problem here; the default run is the 100k x 20k one)."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--m", "6000", "--n", "3000",
           "--k", "64", "--traffic", "0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "iterations/s" and d["value"] > 0 and d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-9)
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    assert "traffic" in r and r["launches_timed"] >= 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["unit"] == d["unit"]
    assert d["with_mae"]["value"] > 0 and d["updates_only"]["value"] > 0 and d["alt"]["value"] > 0
    # the GPU against the fp64 oracle (one update from the final state, exact on a sample), not only against itself
    assert d["checks"]["oracle_step_rel_U"] <= 1e-4 and d["checks"]["oracle_step_rel_V"] <= 1e-4
    assert len(d["repeat"]["legs_of_K_steps"]) == 3
    # every other BASELINE.json configuration has a number on the same line
    sec = d["secondary"]
    assert sec["c1_penalty_fit"]["fit_ms"] > 0 and sec["c1_penalty_fit"]["counts_TP_FP_FN_TN"] == sec["c1_penalty_fit"]["reference_counts"]
    assert sec["c2_wnmf_real"]["iterations_per_s"] > 0 and 0 < sec["c2_wnmf_real"]["roofline"]["frac"] < 1
    assert sec["c5_threshold_line_search"]["iterations_per_s"] > 0
    assert c["cores"] == os.cpu_count() and c["reassociated"]["value"] > 0
    # the pre-heating phase is declared, and what a cold process delivers is on the line next to `value`
    assert d["config"]["preheat_iterations"] >= d["steps"] + d["warmup"] and d["cold_start"]["value"] > 0
    for name in ("xf_f32_tiled", "residual_sums_f32_tiled"):
        assert sec["c2_wnmf_real"]["kernels"][name]["us_per_launch"] > 0


def test_bench_without_preheat():
    """--preheat 0: the run proper starts cold; no cold_start block then."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--m", "3000", "--n", "2000",
           "--k", "64", "--traffic", "0", "--secondary", "0", "--cpu-rows", "0", "--preheat", "0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert "cold_start" not in d and "preheat_iterations" not in d["config"] and d["final"]["iter"] == 4
