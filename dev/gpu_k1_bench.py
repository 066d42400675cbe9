"""Development micro-benchmark of the channeliser (K1) alone-ish: feeds noise for C channels and reports the
HIP-event time of k_chanfir per launch.  usage: python dev/gpu_k1_bench.py [C] [seconds] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dumpvdl2_amd import vdl2hip, synth
if os.environ.get('VDL2HIP_LIB'):
    vdl2hip.load_library(os.environ['VDL2HIP_LIB'])
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cf = 136975000
freqs = synth.channel_plan(C, cf, max(8000, min(100000, 2000000 // C)))
if os.environ.get("K1BENCH_SPACING"):      # e.g. 0: every channel on the centre frequency -> all lanes of a gather hit one LUT entry (no bank conflicts)
    sp = int(os.environ["K1BENCH_SPACING"]); freqs = [cf + (k - C // 2) * sp for k in range(C)]
n = int(secs * 2100000)
iq = (torch.randn(2 * n, device="cuda") * 300).to(torch.int16)
rx = vdl2hip.Receiver(cf, freqs, 20, 1, 3.0, max_block_bytes=iq.numel() * 2)
rx.set_profiling(2)
rx.feed_device(iq.data_ptr(), iq.numel() * 2); rx.drain_packed()      # cold launch, not counted
s0 = rx.stats()
for _ in range(reps):
    rx.feed_device(iq.data_ptr(), iq.numel() * 2)
    rx.drain_packed()
s = rx.stats()
s = {k: s[k] - s0[k] for k in s}
ms = s["chanfir_ms"] / s["chanfir_launches"]
cs = s["chan_samples"] / s["chanfir_launches"]
print(os.environ.get("VDL2HIP_LIB", "default"), os.environ.get("VDL2HIP_CR", ""), f"C={C} secs={secs}: k_chanfir {ms:.4f} ms/launch, {cs / ms * 1e3:.3e} chan-samples/s, algorithmic {cs * 4.4 / ms / 1e6:.1f} GB/s = {cs * 4.4 / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s | "
      f"K2 {s['phase_ms'] / reps:.3f} K3 {s['sync_ms'] / reps:.3f} K4 {s['walk_ms'] / reps:.3f} nf {s['nf_ms'] / reps:.3f} K5 {s['burst_ms'] / reps:.3f} ms")
