"""development aid: after a whole-capture feed with the referee on, where is a channel's decimated stream the oracle's own (made exact), where the channeliser's, where neither?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dumpvdl2_amd import synth, vdl2hip, workloads
from oracle import pyoracle as po
name, dur, ch = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
cfg = getattr(workloads, name)(dur)
iq, _ = synth.synthesize(cfg); raw = iq.view(np.uint8)
D = raw.size // 4 // cfg.oversample
o = po.Oracle(cfg.centerfreq, [cfg.freqs[ch]], oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
tr = o.trace_all(D + 4); o.process(raw, block_bytes=1 << 24, nthreads=2); tr = tr[0, :D, :]
rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, 1, cfg.rx_max_ppm, max_block_bytes=raw.size)
rx.debug_option("ref_kinds", int(sys.argv[4]) if len(sys.argv) > 4 else 1)
rx.feed(raw); fr = rx.drain()
y = rx.read_decimated(ch, 0, D)
same = (y == tr).all(axis=1)
err = np.sqrt(((y.astype(np.float64) - tr) ** 2).sum(axis=1)); mag = np.sqrt((tr.astype(np.float64) ** 2).sum(axis=1))
loc = mag.copy()
for k in range(1, 4): loc[k:] = np.maximum(loc[k:], mag[:-k])
bad = (~same) & (err > 1e-3 * np.maximum(loc, 1e-12)) & (np.arange(D) > 2000)
def runs(m):
    idx = np.flatnonzero(m)
    if idx.size == 0: return []
    cut = np.flatnonzero(np.diff(idx) > 1)
    st = np.concatenate(([idx[0]], idx[cut + 1])); en = np.concatenate((idx[cut], [idx[-1]]))
    return list(zip(st.tolist(), en.tolist()))
ex = runs(same & (np.arange(D) > 2000))
print("frames of the channel:", [(f["burst_ord"], f["idx"], f["sync_sample"], len(f["octets"])) for f in fr if f["chan"] == ch])
print("stretches identical to the oracle:", [(a, b) for a, b in ex if b - a >= 20][:20])
print("stretches that are neither:", runs(bad)[:20], "worst relative", float((err / np.maximum(loc, 1e-12))[2000:].max()))
print(rx.stats())
print("device counters", list(rx.counters(ch).values()))
print("oracle counters", list(o.counters(0).values()))
