"""Development aid: per-stage kernel times of one workload, alone (drain lag 0: nothing overlaps) and in the pipeline (lag 2), and - with
a -DVDL2_K5_PROF build (dev/gpu_batch.sh k5prof:<workload>) - the walker's and the burst decoder's shader clocks per phase (sums kept
in wave-uniform registers, written once per wavefront: vdl2_core.h).   usage: python dev/gpu_stage_times.py [config4] [seconds] [reps]"""
import os, sys, pickle, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dev"))
import numpy as np
import torch
from dumpvdl2_amd import vdl2hip, workloads
import gpu_variants
name = sys.argv[1] if len(sys.argv) > 1 else "config4"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = getattr(workloads, name)(secs)
path = gpu_variants.synth_to_tmp(name, secs)
iq = np.load(path + ".npy")
t = torch.from_numpy(iq).cuda()
rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, 1, cfg.rx_max_ppm, max_block_bytes=iq.nbytes)
L = rx.L
have = hasattr(L, "vdl2hip_debug_k4_prof")
K5N = ["1 slice (atan2)", "2 descramble/pack", "3 de-interleave", "4 Reed-Solomon", "5 un-stuff", "6a frames: list+reserve", "6b frames: records", "6c frames: octets"]
K4N = ["0 entry", "1 stale batch eval", "2 log_evals", "3 bitmap hop", "4 (fire branch)", "5 fire LANE0", "6 header", "7 tail", "8 fire gather", "9 park (no fire)"]


NFN = ["0 first chunk (64-way search)", "1 stage chunk list", "2 per-update search", "3 positions of a batch", "4 wait for the batch's loads", "5 hypot + LDS store", "6 recurrence (64 steps)", "7 store"]
have_nf = hasattr(L, "vdl2hip_debug_nf_prof")


def prof(reset):
    a5 = (C.c_ulonglong * 16)(); a4 = (C.c_ulonglong * 24)(); an = (C.c_ulonglong * 16)()
    L.vdl2hip_debug_k5_prof(a5, reset); L.vdl2hip_debug_k4_prof(a4, reset)
    if have_nf:
        L.vdl2hip_debug_nf_prof(an, reset)
    return list(a5), list(a4), list(an)


for label, lag in (("alone (one block in flight)", 0), ("in the pipeline (three blocks in flight)", 2)):
    rx.set_profiling(2); rx.set_drain_lag(lag)
    for _ in range(3):
        rx.feed_device(t.data_ptr(), iq.nbytes); rx.drain_packed()
    rx.set_drain_lag(0); rx.drain_packed(); rx.set_drain_lag(lag)
    if have:
        prof(1)
    s0 = rx.stats()
    for _ in range(reps):
        rx.feed_device(t.data_ptr(), iq.nbytes); rx.drain_packed()
    rx.set_drain_lag(0); rx.drain_packed()
    s = rx.stats()
    d = {k: (s[k] - s0[k]) / reps for k in ("chanfir_ms", "sync_ms", "walk_ms", "nf_ms", "burst_ms")}
    nb = (s["bursts"] - s0["bursts"]) / reps
    print(f"{name} {secs:g} s, {label}:", {k: round(v, 4) for k, v in d.items()}, "bursts/feed", nb, "env", {k: v for k, v in os.environ.items() if k.startswith("VDL2HIP_")})
    if have:
        a5, a4, an = prof(0)
        nb5 = a5[8] or 1
        tot5 = sum(a5[:8]) or 1
        print(f"  K5: {a5[8]} bursts decoded, {tot5 / nb5:.0f} shader clocks per burst")
        for i in range(8):
            print(f"     {K5N[i]:26s} {a5[i] / nb5:10.0f} clk/burst {100.0 * a5[i] / tot5:5.1f}%")
        tot4 = sum(a4[:10]) or 1
        print(f"  K4: {tot4 / reps:.4g} shader clocks per feed summed over all walker waves")
        for i in range(10):
            n = a4[12 + i] or 1
            print(f"     {K4N[i]:20s} {100.0 * a4[i] / tot4:5.1f}%  n/feed={a4[12 + i] / reps:9.0f}  clk each={a4[i] / n:9.0f}")
        if have_nf and an[8]:
            totn = sum(an[:8]) or 1
            print(f"  NF replay: {an[8] / reps:.0f} groups of 32 updates per feed, {totn / an[8]:.0f} shader clocks per group")
            for i in range(8):
                print(f"     {NFN[i]:32s} {an[i] / an[8]:10.0f} clk/group {100.0 * an[i] / totn:5.1f}%")
