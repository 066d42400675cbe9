import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dumpvdl2_amd import vdl2hip, synth, workloads
cfg = workloads.config4(2.0)
iq, _ = synth.synthesize(cfg)
raw = iq.view(np.uint8)
BLK = 320000
rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=BLK)
for k in range(0, 10 * BLK, BLK): rx.feed(raw[k:k + BLK]); rx.drain_packed()
for gap_ms in (0, 0.5, 1, 2, 5, 20, 38):
    tot = 0.0
    for k in range(0, 40 * BLK, BLK):
        t0 = time.perf_counter(); rx.feed(raw[k:k + BLK]); rx.drain_packed(); tot += time.perf_counter() - t0
        if gap_ms:
            t1 = time.perf_counter()
            if os.environ.get("SPIN"):
                while time.perf_counter() - t1 < gap_ms * 1e-3: pass
            else: time.sleep(gap_ms * 1e-3)
    print(f"idle gap {gap_ms} ms between blocks ({'spin' if os.environ.get('SPIN') else 'sleep'}): feed + drain {tot / 40 * 1e3:.3f} ms per block", flush=True)
