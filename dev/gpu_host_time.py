"""Where does the host spend a rank-sized receiver's step?  Times vdl2hip_feed_device() and vdl2hip_drain_packed() separately over K steps
(block resident, drain lag 5) for a 32-channel shard of config4.  usage: python dev/gpu_host_time.py [first_channel] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from dumpvdl2_amd import synth, vdl2hip, workloads
first = int(sys.argv[1]) if len(sys.argv) > 1 else 96
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = workloads.config4(16.0)
path = "/tmp/vdl2_config4_16.npy"
if os.path.exists(path):
    iq = np.load(path)
else:
    iq, _ = synth.synthesize(cfg); np.save(path, iq)
nbytes = iq.size * 2
dev = torch.from_numpy(iq).to("cuda:0")
rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=nbytes, chan_first=first, chan_count=32)
for _ in range(8):
    rx.feed_device(dev.data_ptr(), nbytes); rx.drain_packed()
for lag in [int(x) for x in os.environ.get("LAGS", "0,1,2,3,4,5").split(",")]:
    rx.set_drain_lag(lag)
    for _ in range(6):
        rx.feed_device(dev.data_ptr(), nbytes); rx.drain_packed()
    torch.cuda.synchronize()
    tf = td = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        a = time.perf_counter(); rx.feed_device(dev.data_ptr(), nbytes); b = time.perf_counter(); rx.drain_packed(); c = time.perf_counter()
        tf += b - a; td += c - b
    rx.set_drain_lag(0); rx.drain_packed(); torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"shard {first}..{first + 31} lag {lag}: {tot / steps * 1e3:.3f} ms per step; inside feed_device {tf / steps * 1e3:.3f} ms, inside drain_packed {td / steps * 1e3:.3f} ms", flush=True)
