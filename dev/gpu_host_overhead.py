"""Development probe: host time spent inside vdl2hip_feed_device / vdl2hip_drain_packed per step of the bench loop
(three blocks in flight), next to the step time - tells whether the loop is GPU- or host-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dumpvdl2_amd import vdl2hip, synth, workloads
cfg = workloads.config2(16.0)
iq, _ = synth.synthesize(cfg)
buf = torch.from_numpy(iq).cuda()
rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, 1, cfg.rx_max_ppm, max_block_bytes=iq.nbytes)
rx.set_drain_lag(2)
for _ in range(4):
    rx.feed_device(buf.data_ptr(), iq.nbytes); rx.drain_packed()
torch.cuda.synchronize()
tf = td = 0.0; n = 30
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter(); rx.feed_device(buf.data_ptr(), iq.nbytes); b = time.perf_counter(); rx.drain_packed(); c = time.perf_counter()
    tf += b - a; td += c - b
rx.set_drain_lag(0); rx.drain_packed(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"step {dt / n * 1e3:.3f} ms; host inside feed {tf / n * 1e3:.3f} ms, inside drain (incl. waiting for the GPU) {td / n * 1e3:.3f} ms")
