"""bench.py's JSON line without a GPU: the parts of it that are plain arithmetic - the roofline object of the dominant kernel
(keys and fractions as the measurement contract names them, SURVEY 8.5's per-unit figures) and the PMC traffic entry it quotes
from profiles/pmc_traffic.json - so that an edit of bench.py cannot silently drop a field the driver's line is checked for."""
import importlib.util
import json
import os
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_roofline_object(bench):
    cs = 33_600_000 * 256                       # channel-samples of one 16 s x 256-channel launch
    t = {"k1_ms": 3.8, "k1_chan_samples": cs}
    case = types.SimpleNamespace(count=256, cfg=types.SimpleNamespace(duration_s=16.0))
    tr = bench.pmc_traffic("config4", case)
    assert tr and tr["traffic_bytes"] > 3.4e9 and "profiles/" in tr["source"]          # at least the 3.44 GB of y it stores
    assert bench.pmc_traffic("config4", types.SimpleNamespace(count=32, cfg=case.cfg)) is None      # quoted only where it was measured
    r = bench.roofline_of(t, tr)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "hbm_algorithmic", "hbm_physical"):
        assert k in r, k
    assert r["unit"] == "TFLOP/s" and r["peak"] == bench.VALU_PEAK_TFLOPS and r["kernel"] == "k_chanfir"
    assert abs(r["achieved"] - cs * 30.0 / 3.8e-3 / 1e12) < 0.01 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # 4 B per channel-sample in + 8 / oversample out
    a = r["hbm_algorithmic"]
    assert a["bytes_per_chan_sample"] == pytest.approx(4.4) and a["bytes_per_launch"] == pytest.approx(cs * 4.4)
    assert a["achieved"] == pytest.approx(cs * 4.4 / 3.8e-3 / 1e9, rel=1e-3) and a["peak"] == 8000.0
    p = r["hbm_physical"]
    assert p["achieved"] == pytest.approx(tr["traffic_bytes"] / 3.8e-3 / 1e9, rel=1e-3) and p["traffic_over_algorithmic"] < 0.2
    json.dumps(r)
    assert bench.roofline_of(t, None)["hbm_physical"] is None


def test_workload_names_are_baseline_configs(bench):
    from dumpvdl2_amd import workloads
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert bench.WORKLOAD_INDEX == {"config2": 1, "config3": 2, "config4": 3, "config5": 4} and len(base["configs"]) == 5
    for name, idx in bench.WORKLOAD_INDEX.items():
        cfg = getattr(workloads, name)(0.1)
        want = {1: 8, 2: 64, 3: 256, 4: 256}[idx]
        assert len(cfg.freqs) == want and str(want) in base["configs"][idx]
        assert cfg.oversample == 20                    # 2.1 MS/s
