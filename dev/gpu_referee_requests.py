"""development aid: where in a feed do the decisions lie that the walk hands to the referee?  (a bench workload in blocks of `block_s` seconds)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dumpvdl2_amd import synth, vdl2hip, workloads
name, dur, block_s = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
cfg = getattr(workloads, name)(dur)
iq, _ = synth.synthesize(cfg); raw = iq.view(np.uint8)
bs = int(block_s * 105000 * cfg.oversample) * 4
rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, 1, cfg.rx_max_ppm, max_block_bytes=bs)
f = rx.L.vdl2hip_debug_read_requests; f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
out = np.zeros((8192, 4), dtype=np.int64)
allr = []
for i in range(0, raw.size, bs):
    blk = raw[i:i + bs]; rx.feed(blk); rx.sync()
    n = f(rx.h, out.ctypes.data, 8192)
    D = blk.size // 4 // cfg.oversample
    allr.append((i // bs, n, out[:min(n, 8192)].copy(), D))
rx.drain()
for k, n, r, D in allr[:12]:
    off = r[:, 2] - r[:, 3]
    print(f"feed {k}: {n} requests; kinds {np.bincount(r[:, 1], minlength=3).tolist()}; offsets in the feed (D = {D}): quartiles {np.percentile(off, [0, 25, 50, 75, 100]).astype(int).tolist() if n else []}; "
          f"channels {len(set(r[:, 0].tolist()))}; first {[(int(a), int(b), int(c - d)) for a, b, c, d in r[:6]]}")
tot = sum(n for _, n, _, _ in allr)
keys = set()
for _, _, r, _ in allr: keys |= {(int(a), int(c)) for a, _, c, _ in r}
print(f"total {tot} requests over {len(allr)} feeds, {len(keys)} distinct (chan, n); stats {rx.stats()}")
