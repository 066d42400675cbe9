"""Development aid: isolated per-stage kernel times of one workload (drain lag 0: nothing overlaps), and with VDL2_K5_PROF builds
the walker's per-phase cycle counters.  usage: python dev/gpu_stage_times.py [config4] [seconds] [reps]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dumpvdl2_amd import vdl2hip, synth, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "config4"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = getattr(workloads, name)(secs)
iq, bursts = synth.synthesize(cfg)
t = torch.from_numpy(iq).cuda()
rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, 1, cfg.rx_max_ppm, max_block_bytes=iq.nbytes)
rx.set_profiling(2)
rx.feed_device(t.data_ptr(), iq.nbytes); rx.drain_packed()
s0 = rx.stats()
for _ in range(reps):
    rx.feed_device(t.data_ptr(), iq.nbytes); rx.drain_packed()
s = rx.stats()
d = {k: (s[k] - s0[k]) / reps for k in ("chanfir_ms", "phase_ms", "sync_ms", "walk_ms", "nf_ms", "burst_ms")}
print(name, secs, "s:", {k: round(v, 4) for k, v in d.items()}, "seg adopted/walked per feed", (s["seg_adopted"] - s0["seg_adopted"]) / reps, (s["seg_walked"] - s0["seg_walked"]) / reps,
      "bursts/feed", (s["bursts"] - s0["bursts"]) / reps, "env", {k: v for k, v in os.environ.items() if k.startswith("VDL2HIP_")})
L = rx.L
try:
    L.vdl2hip_debug_k4_prof
    have = True
except AttributeError:
    have = False
if have:
    a = (C.c_ulonglong * 16)()
    L.vdl2hip_debug_k5_prof(a)
    names5 = ["1 slice symbols (atan2)", "2 descramble/pack", "3 de-interleave", "4 Reed-Solomon", "5 un-stuff", "6 frames out"]
    tot5 = sum(a[8 + i] for i in range(6)) or 1
    print("K5 phases (lane-0 cycles summed over all bursts; share; longest single):")
    for i in range(6):
        print(f"   {names5[i]:26s} {a[8 + i]:14d} {100.0 * a[8 + i] / tot5:5.1f}%  max={a[i]}")
    L.vdl2hip_debug_k4_prof(a)
    names = ["state load/store", "stale batch eval", "account_evals", "bitmap hop", "(fire setup)", "fire handling", "header", "burst emit/tail"]
    tot = sum(a[i] for i in range(8)) or 1
    print("K4 phases (cycles summed over all walker waves; share; count):")
    for i in range(8):
        print(f"   {i} {names[i]:18s} {a[i]:14d} {100.0 * a[i] / tot:5.1f}%  n={a[8 + i]}")
