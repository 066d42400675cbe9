"""Development probe: builds libvdl2hip with -DVDL2_K5_PROF into /tmp and prints the max / mean cycles
the burst decoder spends per phase (slice, octets, deinterleave, RS, reserialise, unstuff+emit)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dumpvdl2_amd import build, vdl2hip, synth, workloads
lib = "/tmp/libvdl2hip_prof.so"
subprocess.check_call([build.hipcc_path()] + build.FLAGS + ["-DVDL2_K5_PROF", "-o", lib, os.path.join(build.CSRC, "vdl2hip.hip")])
L = vdl2hip.load_library(lib)
cfg = workloads.config2(4.0)
iq, bursts = synth.synthesize(cfg)
rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), 20, 1, 0.0, max_block_bytes=iq.nbytes)
rx.set_profiling(2)
rx.feed(iq); fr = rx.drain()
a = (C.c_ulonglong * 16)()
print("rc", L.vdl2hip_debug_k5_prof(a), "frames", len(fr), "bursts", len(bursts), "burst_ms", rx.stats()["burst_ms"], "walk_ms", rx.stats()["walk_ms"])
names = ["slice", "octets", "deinterleave", "rs", "reserialise", "unstuff+emit"]
for i, n in enumerate(names):
    print(f"{n:14s} max {a[i]:10d} cycles   mean {a[8 + i] / max(1, len(bursts)):12.0f}")

b = (C.c_ulonglong * 16)()
print("k4 rc", L.vdl2hip_debug_k4_prof(b))
names4 = ["state load/store", "stale batch eval", "account_evals", "bitmap hop", "(fire setup)", "fire handling", "header", "burst emit/tail"]
for i, n in enumerate(names4):
    print(f"{n:18s} total {b[i]:12d} cycles over {b[8 + i]:6d} marks  ({b[i] / 8 / 2.1e6:8.3f} ms per channel at 2.1 GHz)")
