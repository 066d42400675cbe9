"""What handing the reference's own blocks (FILE_BUFSIZE = 320 000 bytes, dumpvdl2.h:48) to the GPU k at a time costs per block:
the same capture fed in pieces of k x 320 000 bytes from pageable memory, one vdl2hip_feed() + drain per piece, drain lag 0 / 1 / 2.
(The drop-in adapter collects k blocks before it feeds: csrc/dropin.c, VDL2HIP_DROPIN_BATCH.)
usage: python dev/gpu_block_batch.py [config2|config3|config4] [seconds] [k,k,...] [lag,lag,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dumpvdl2_amd import vdl2hip, synth, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "config4"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
ks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8,12,16,24,32").split(",")]
lags = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0,1,2").split(",")]
cfg = getattr(workloads, name)(secs)
iq, bursts = synth.synthesize(cfg)
raw = iq.view(np.uint8)
BLK = 320000
nblk = (raw.size + BLK - 1) // BLK
ref_frames = None
for k in ks:
    piece = k * BLK
    for lag in lags:
        rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=piece)
        rx.set_drain_lag(lag)
        for o in range(0, min(raw.size, max(20 * BLK, 6 * piece)), piece):      # warm
            rx.feed(raw[o:o + piece]); rx.drain_packed()
        rx.close()
        rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=piece)
        rx.set_drain_lag(lag)
        n = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for o in range(0, raw.size, piece):
            rx.feed(raw[o:o + piece])
            n += rx.drain_packed()[0]
        rx.set_drain_lag(0); n += rx.drain_packed()[0]
        dt = time.perf_counter() - t0
        st = rx.stats()
        rx.close()
        if ref_frames is None: ref_frames = n
        print(f"{name} {len(cfg.freqs)} ch, {secs:g} s = {nblk} blocks, {k:2d} blocks per feed, lag {lag}: {dt / nblk * 1e3:.3f} ms per block "
              f"({raw.size / 4 / dt / 2.1e6:.0f}x real time); frames {n}{'' if n == ref_frames else ' != ' + str(ref_frames)}; scans {st.get('referee_scans', '?')}", flush=True)
