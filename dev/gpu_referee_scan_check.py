"""development aid: the referee's scan against the oracle's stream over a whole channel of a bench workload, stretch by stretch"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dumpvdl2_amd import synth, vdl2hip, workloads
from oracle import pyoracle as po
name, dur, chans = sys.argv[1], float(sys.argv[2]), [int(c) for c in sys.argv[3].split(",")]
cfg = getattr(workloads, name)(dur)
iq, _ = synth.synthesize(cfg); raw = iq.view(np.uint8)
D = raw.size // 4 // cfg.oversample
o = po.Oracle(cfg.centerfreq, [cfg.freqs[c] for c in chans], oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
tr = o.trace_all(D + 4); o.process(raw, block_bytes=1 << 24, nthreads=8); tr = tr[:, :D, :]
rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, 1, cfg.rx_max_ppm, max_block_bytes=raw.size)
rx.debug_option("referee", 0)
rx.feed(raw); rx.drain()
for i, c in enumerate(chans):
    nbad = 0; first = None
    for lo in list(range(0, 8192, 256)) + list(range(8192, D - 256, 8192)):
        assert rx.exact_window(c, lo, lo + 255)
        got = rx.read_decimated(c, lo, 256)
        if got.tobytes() != tr[i, lo:lo + 256].tobytes():
            nbad += 1
            if first is None: first = (lo, int(np.flatnonzero((got != tr[i, lo:lo + 256]).any(axis=1))[0]), float(np.abs(got - tr[i, lo:lo + 256]).max()))
    print(f"channel {c}: {nbad} stretches differ from the oracle; first {first}", flush=True)
print(rx.stats())
