"""What one referee scan costs (kernels.h: ref_exact_window_dev), alone on the device: wall time of vdl2hip_debug_exact_window() for
stretches of 256 decimated samples with run-ups of 2^16 .. 2^18 input samples (one wavefront; the call's launch + wait is ~30 us)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from dumpvdl2_amd import vdl2hip
cfg, iq, _, _ = cases.load("config2_1s")
raw = iq.view(np.uint8)
rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=raw.size)
rx.feed(raw); rx.drain()
D = raw.size // 4 // cfg.oversample
for warm in (1 << 15, 1 << 16, 1 << 17):
    rx.debug_option("ref_warm", warm); rx.feed(raw[:4000]); rx.drain()   # (the run-up travels in the feed's hook)
    ts = []
    for i in range(12):
        lo = 40000 + 3072 * i + (warm >> 8)          # (a new stretch every time: the list of stretches done would answer a repeat)
        t0 = time.perf_counter(); ok = rx.exact_window(i % len(cfg.freqs), lo, lo + 200); ts.append((time.perf_counter() - t0) * 1e3)
        assert ok
    print(f"run-up {warm}: {np.median(ts):.3f} ms per scan (min {min(ts):.3f}, max {max(ts):.3f}); {np.median(ts) * 1e6 / (warm + 5120):.2f} ns per input sample", flush=True)
# many wavefronts at once (the device then runs at its working clock; one wavefront per SIMD up to 1024)
for warm in (1 << 16,):
    rx.debug_option("ref_warm", warm); rx.feed(raw[:4000]); rx.drain()   # (the run-up travels in the feed's hook)
    for count in (64, 1024, 4096):
        base = 30000 + (warm >> 8) + 7 * count
        done, ms = rx.exact_window_many(0, base, base + 200, count, 256)
        print(f"run-up {warm}, {count} scans at once: {ms:.3f} ms for the kernel ({done} done); {ms * 1e6 / (warm + 5120):.2f} ns per input sample and wavefront", flush=True)
print(rx.stats())
# ... and side by side (k_ref_scan_multi: 32 requests per workgroup)
rng = np.random.default_rng(3)
rx.debug_option("ref_warm", 1 << 17); rx.feed(raw[:4000]); rx.drain()
for count in (1, 32, 160, 1024):
    chans = rng.integers(0, len(cfg.freqs), count); los = rng.integers(20000, D - 2000, count); his = los + 255
    ran, ms = rx.scan_multi(chans, los, his)
    print(f"side by side, run-up 131072: {count} requests ({ran} scans run) in {ms:.3f} ms", flush=True)
