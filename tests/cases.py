"""Seeded inputs shared by the CPU and GPU parity tests (same factories as tests/golden/make_golden.py)."""
import functools
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import CASES  # noqa: E402
from dumpvdl2_amd import synth  # noqa: E402


@functools.lru_cache(maxsize=None)
def load(name):
    """-> (cfg, iq int16 array, tx bursts, golden dict)"""
    cfg = CASES[name]()
    iq, bursts = synth.synthesize(cfg)
    with open(os.path.join(ROOT, "tests", "golden", name + ".json")) as f:
        gold = json.load(f)
    assert hashlib.sha1(iq.tobytes()).hexdigest() == gold["iq_sha1"], \
        f"{name}: regenerated IQ differs from the one the golden answers were made for"
    return cfg, iq, bursts, gold


def golden_frames_as_dicts(gold):
    return gold["frames"]


def check_against_golden(frames, counters, gold, tol_db=0.05, tol_ppm=0.01, label="", exact_diagnostics=True):
    """frames: decoder output dicts (with octets); gold: committed oracle answers."""
    got = sorted(frames, key=lambda f: (f["chan"], f["burst_ord"], f["idx"]))
    want = gold["frames"]
    assert len(got) == len(want), f"{label}: {len(got)} frames, golden has {len(want)}"
    for g, w in zip(got, want):
        assert (g["chan"], g["burst_ord"], g["idx"]) == (w["chan"], w["burst_ord"], w["idx"])
        assert len(g["octets"]) == w["len"] and hashlib.sha1(g["octets"]).hexdigest() == w["sha1"], f"{label}: octets differ in {w}"
        for k in ("synd_weight", "datalen_octets", "num_fec_corrections", "sync_sample", "end_sample"):
            assert g[k] == w[k], f"{label}: {k} {g[k]} != {w[k]}"
        assert abs(g["frame_pwr_dbfs"] - w["frame_pwr_dbfs"]) <= tol_db
        assert abs(g["nf_pwr_dbfs"] - w["nf_pwr_dbfs"]) <= tol_db
        assert abs(g["ppm_error"] - w["ppm_error"]) <= tol_ppm
    if counters is not None:
        assert_counters_equal(counters, gold["counters"], label, exact_diagnostics)


N_REFERENCE_COUNTERS = 18      # the reference's statsd counters; the last two are this repo's own diagnostics


def assert_counters_equal(got, want, label="", exact_diagnostics=True):
    """The 18 counters the reference itself keeps must be identical per channel.  demod.ppm_reject and
    demod.slicer_neg_idx have no reference counterpart; on the GPU path they may differ by marginal events
    (a leaked preamble whose metric sits within the filter's rounding noise of the threshold)."""
    got = [list(c) for c in got]
    assert [c[:N_REFERENCE_COUNTERS] for c in got] == [c[:N_REFERENCE_COUNTERS] for c in want], f"{label}: reference counters differ"
    if exact_diagnostics:
        assert got == want, f"{label}: diagnostic counters differ"
    else:
        for k in (18, 19):
            a = sum(c[k] for c in got); b = sum(c[k] for c in want)
            assert abs(a - b) <= max(3, 0.01 * b), f"{label}: diagnostic counter {k}: {a} vs {b}"


def with_silence(cfg, iq, bursts):
    """The capture with exact silence (zero samples) in front of it, behind it and - where no burst of any channel is on the
    air - cut into it: the filter output then decays through the denormal range to exact (signed) zeros.  A burst is never cut:
    what a receiver makes of the ringing after a cut is implementation noise, in the reference too."""
    x = np.array(iq, dtype=np.int16).reshape(-1, 2).copy()
    n = x.shape[0]
    sps = 10 * cfg.oversample
    busy = np.zeros(n, dtype=bool)
    for b in bursts:
        a = max(0, int(b.start_sample) - 20 * sps)
        e = min(n, int(b.start_sample) + (120 + (int(b.tl_bits) + 8 * 60) // 3) * sps)
        busy[a:e] = True
    free = np.flatnonzero(np.diff(np.concatenate(([True], busy, [True])).astype(np.int8)))   # edges of the free stretches
    for a, e in zip(free[0::2], free[1::2]):
        if e - a >= 40000: x[a + 5000:e - 5000] = 0
    pad = np.zeros((150000, 2), dtype=np.int16)
    return np.concatenate([pad, x, pad[:120000]]).reshape(-1)
