import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, cases
from dumpvdl2_amd import vdl2hip
name = sys.argv[1] if len(sys.argv) > 1 else "config4_0p4s"
cfg, iq, bursts, gold = cases.load(name)
rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, 1, cfg.rx_max_ppm, max_block_bytes=iq.nbytes)
rx.feed(iq); fr = rx.drain()
for c in range(len(cfg.freqs)):
    a = list(rx.counters(c).values()); b = gold["counters"][c]
    if a != b:
        print("chan", c, {vdl2hip.COUNTER_NAMES[i]: (a[i], b[i]) for i in range(len(a)) if a[i] != b[i]})
