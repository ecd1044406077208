"""The benchmark contract on a real GPU: `bench.py` prints exactly one JSON line on stdout with
the agreed keys, for the headline op and for the point-sampling op."""

from __future__ import annotations

import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
        "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _run(*args):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", *args],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_headline_line():
    d = _run()
    assert KEYS <= set(d)
    assert d["metric"] == "msdeformattn_fwd_bwd_sampled_points_per_sec" and d["unit"] == "points/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert d["config"]["workload"] == "cityscapes_1024x2048_fp32_b8" and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1
    assert d["value"] > 1e9  # more than a G sampled points per second
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "points/s"


def test_point_sample_line():
    d = _run("--op", "point_sample")
    assert KEYS <= set(d)
    assert d["metric"] == "point_sample_fwd_bwd_points_per_sec"
    assert d["config"]["workload"] == "point_sample_masks_256x512_p12544"
    assert 0 < d["roofline"]["frac"] < 1 and d["cpu_baseline"]["kind"] == "port"
