"""vdl2hip_group_* with 8 members on ONE device (the rig the multi-GPU C path is tested on): ms per 16 s x 256-channel step in both exchange
forms, host time inside vdl2hip_group_feed_pinned(), against the one receiver of all 256 channels.  usage: python dev/gpu_group_one_device.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from dumpvdl2_amd import synth, vdl2hip as vh, workloads
from util import truth_is_subset
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = workloads.config4(16.0)
iq, bursts = synth.synthesize(cfg)
host = torch.from_numpy(iq); pin = host.pin_memory(); nbytes = iq.nbytes
want = sum(len(b.frames) for b in bursts if b.decodable)
rx = vh.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vh.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=nbytes)
rx.set_drain_lag(vh.MAX_DRAIN_LAG)
for _ in range(4): rx.feed_pinned(pin.data_ptr(), nbytes); rx.drain_packed()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): rx.feed_pinned(pin.data_ptr(), nbytes); rx.drain_packed()
rx.set_drain_lag(0); rx.drain_packed(); torch.cuda.synchronize()
print(f"one receiver, 256 channels: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step", flush=True)
rx.close()
for members in (8, 2):
    g = vh.ReceiverGroup(cfg.centerfreq, list(cfg.freqs), [0] * members, cfg.oversample, vh.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=nbytes)
    for form in ("allgather", "broadcast"):
        g.set_exchange(form); g.set_drain_lag(0)
        g.feed_pinned(pin.data_ptr(), nbytes); fr = g.drain()
        assert truth_is_subset(bursts, fr) == 0 and len(fr) == want, (form, len(fr), want)
        g.set_drain_lag(vh.MAX_DRAIN_LAG)
        for _ in range(3): g.feed_pinned(pin.data_ptr(), nbytes); g.drain_count()
        g.set_drain_lag(0); g.drain_count(); g.sync(); g.set_drain_lag(vh.MAX_DRAIN_LAG)
        torch.cuda.synchronize(); t0 = time.perf_counter(); tc = 0.0
        for _ in range(steps):
            a = time.perf_counter(); g.feed_pinned(pin.data_ptr(), nbytes); tc += time.perf_counter() - a
            g.drain_count()
        g.set_drain_lag(0); g.drain_count(); g.sync(); torch.cuda.synchronize()
        print(f"group of {members} on one device, {form}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step, {tc / steps * 1e3:.3f} ms inside vdl2hip_group_feed_pinned(); frames {len(fr)} = sent", flush=True)
    g.close()
