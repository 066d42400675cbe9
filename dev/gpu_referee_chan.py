"""development aid: one channel of a bench workload on the device with the referee, against the oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dumpvdl2_amd import synth, vdl2hip, workloads
from oracle import pyoracle as po
name, dur, ch = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
cfg = getattr(workloads, name)(dur)
iq, _ = synth.synthesize(cfg); raw = iq.view(np.uint8)
o = po.Oracle(cfg.centerfreq, [cfg.freqs[ch]], oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
o.process(raw, block_bytes=1 << 24, nthreads=2)
fo = [(f["burst_ord"], f["idx"], f["sync_sample"], f["end_sample"], len(f["octets"])) for f in o.frames()]
print("oracle", fo, list(o.counters(0).values())[:18])
for env in ({}, {"VDL2HIP_SEG_MAX": "1"}):
    os.environ.pop("VDL2HIP_SEG_MAX", None); os.environ.update(env)
    for kinds in (0, 1, 7):
        rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, 1, cfg.rx_max_ppm, max_block_bytes=raw.size, chan_first=ch, chan_count=1)
        rx.debug_option("referee", 1 if kinds else 0); rx.debug_option("ref_kinds", kinds or 7)
        rx.feed(raw); fr = rx.drain()
        got = [(f["burst_ord"], f["idx"], f["sync_sample"], f["end_sample"], len(f["octets"])) for f in fr]
        s = rx.stats()
        print(env, "kinds", kinds, "same" if got == fo else got, list(rx.counters(ch).values())[:18], "scans", s["referee_candidate_scans"], s["referee_header_scans"], s["referee_symbol_scans"], "seg", s["seg_adopted"], s["seg_walked"], flush=True)
        rx.close()
