"""Throughput of the drop-in way of using the library: the reference's own block size (FILE_BUFSIZE = 320 000 bytes, dumpvdl2.h:48),
one block at a time, every frame delivered before the next block is read (the reference's blocking process_buf_*() semantics:
vdl2hip_feed() + vdl2hip_drain() per block, drain lag 0) - and the same blocks with one / two blocks of lag.
usage: python dev/gpu_dropin_rate.py [config2|config3|config4] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dumpvdl2_amd import vdl2hip, synth, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "config4"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
cfg = getattr(workloads, name)(secs)
iq, bursts = synth.synthesize(cfg)
raw = iq.view(np.uint8)
want = sum(len(b.frames) for b in bursts if b.decodable)
BLK = 320000
for lag in (0, 1, 2):
    rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=BLK)
    rx.set_drain_lag(lag)
    for k in range(0, 20 * BLK, BLK):          # warm
        rx.feed(raw[k:k + BLK]); rx.drain_packed()
    rx.close()
    rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=BLK)
    rx.set_drain_lag(lag)
    n = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(0, raw.size, BLK):
        rx.feed(raw[k:k + BLK])
        n += rx.drain_packed()[0]
    rx.set_drain_lag(0); n += rx.drain_packed()[0]
    dt = time.perf_counter() - t0
    nblk = (raw.size + BLK - 1) // BLK
    print(f"{name} {len(cfg.freqs)} channels, {secs:g} s in {nblk} blocks of {BLK} bytes, drain lag {lag}: {dt * 1e3:.1f} ms = {dt / nblk * 1e3:.3f} ms per block, "
          f"{raw.size / 4 / dt / 1e6:.1f} MS/s = {raw.size / 4 / dt / 2.1e6:.1f}x real time; frames {n} (sent {want})")
    rx.close()
