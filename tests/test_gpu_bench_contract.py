"""-m gpu: `python bench.py` prints ONE JSON line that carries what the driver's contract asks for (metric / value / unit / n_gpus / steps /
warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the `roofline` and `cpu_baseline`
objects, with the timed registrations checked against the oracle, binned ahead of their registration and -- N = 1 -- chained on the device."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-secondary", "--cpu-sample", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "rank 0 prints exactly one line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "icp_registrations_per_sec" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and "os1_128_2m" in d["config"]["workload"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-6
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and 0.0 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert d["parity_iteration_counts_and_histograms_equal"] is True
    assert max(d["parity_vs_oracle_m_rad"]) < 1e-8
    assert d["host"]["binned_ahead_timed_steps"] >= 4  # (the timed scans were binned behind their copies, beside the registration before them)
    # N = 1: the timed steps are one so_icp_register_sequence call; all but the first start behind the registration in front of them
    assert d["config"]["entry"] == "chained" and d["host"]["chained_timed_steps"] >= 4
