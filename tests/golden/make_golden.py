"""Generates tests/golden/*.json: oracle outputs for seeded synthetic inputs.

Run in the build container (python tests/golden/make_golden.py).  The inputs are regenerated
from their seeds by the tests on either box; only the oracle's answers are stored here.
Each frame is stored as (chan, burst_ord, idx, len, sha1(octets), integer metadata, floats).
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from dumpvdl2_amd import synth, workloads  # noqa: E402

CASES = {
    "config2_1s": lambda: workloads.config2(1.0),
    "config3_0p6s": lambda: workloads.config3(0.6),
    "config4_0p4s": lambda: workloads.config4(0.4),
    "config5_0p4s": lambda: workloads.config5(0.4),
    "dirty25k_1s": lambda: synth.SynthConfig(centerfreq=workloads.CENTER, freqs=synth.channel_plan(8, workloads.CENTER, 25000),
                                             oversample=20, duration_s=1.0, seed=11),
    "os10_noisy_1s": lambda: synth.SynthConfig(centerfreq=workloads.CENTER, freqs=[workloads.CENTER, workloads.CENTER + 50000],
                                               oversample=10, duration_s=1.0, seed=5, noise_sigma=0.02, amplitude=0.2),
}


def frame_record(f):
    return dict(chan=f["chan"], burst_ord=f["burst_ord"], idx=f["idx"], len=len(f["octets"]),
                sha1=hashlib.sha1(f["octets"]).hexdigest(), synd_weight=f["synd_weight"],
                datalen_octets=f["datalen_octets"], num_fec_corrections=f["num_fec_corrections"],
                sync_sample=f["sync_sample"], end_sample=f["end_sample"],
                frame_pwr_dbfs=round(f["frame_pwr_dbfs"], 4), nf_pwr_dbfs=round(f["nf_pwr_dbfs"], 4),
                ppm_error=round(f["ppm_error"], 5))


def run_case(cfg):
    iq, bursts = synth.synthesize(cfg)
    o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
    o.process(iq.view(np.uint8), block_bytes=1 << 24, nthreads=4)
    fr = sorted(o.frames(), key=lambda f: (f["chan"], f["burst_ord"], f["idx"]))
    return dict(iq_sha1=hashlib.sha1(iq.tobytes()).hexdigest(), n_tx_bursts=len(bursts),
                frames=[frame_record(f) for f in fr],
                counters=[list(o.counters(c).values()) for c in range(len(cfg.freqs))])


if __name__ == "__main__":
    out = os.path.dirname(os.path.abspath(__file__))
    for name, mk in CASES.items():
        rec = run_case(mk())
        with open(os.path.join(out, name + ".json"), "w") as f:
            json.dump(rec, f, separators=(",", ":"))
        print(name, len(rec["frames"]), "frames", rec["iq_sha1"][:12])
