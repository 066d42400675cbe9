import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dumpvdl2_amd import vdl2hip, synth, workloads
cfg = workloads.config4(2.0)
iq, _ = synth.synthesize(cfg)
raw = iq.view(np.uint8)
BLK = 320000
for cap in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "320000,8388608").split(",")]:
    rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=cap)
    for k in range(0, 10 * BLK, BLK): rx.feed(raw[k:k + BLK]); rx.drain_packed()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for k in range(0, 40 * BLK, BLK): rx.feed(raw[k:k + BLK]); n += rx.drain_packed()[0]
    dt = time.perf_counter() - t0
    print(f"max_block_bytes {cap}: {dt / 40 * 1e3:.3f} ms per 320 000-byte block, frames {n}", flush=True)
    rx.close()
