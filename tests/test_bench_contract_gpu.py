"""GPU: `bench.py` prints exactly ONE JSON line on stdout carrying the contract's fields (a small configuration)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", *extra],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines          # RCCL banners and warnings must not reach stdout
    return json.loads(lines[0])


def test_sor_line_has_the_contract_fields():
    d = _run("--n", "300000", "--extent", "10", "--no-secondary")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "Msplats/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and "300000" in d["config"]["workload"]
    assert abs(d["value"] - 300000 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.02 * d["value"]
    r = d["roofline"]
    assert r["kernel"] == "knn_brick_kernel" and r["kernel_ms"] > 0 and r["bound"] in ("hbm", "valu")
    hbm = r["hbm"] if "hbm" in r else r
    for key in ("achieved", "peak", "unit", "frac", "traffic"):
        assert key in hbm, key
    assert hbm["unit"] == "GB/s" and hbm["peak"] == 8000.0 and abs(hbm["frac"] - hbm["achieved"] / hbm["peak"]) < 1e-4
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] >= 1 and c["mask_identical_to_gpu"] is True


def test_slab_pipeline_line_one_rank():
    d = _run("--n", "300000", "--extent", "10", "--no-secondary", "--no-cpu-baseline", "--exchange", "slab")
    assert "slab" in d["config"]["parallelism"] and d["survivors_rank0"] > 0
