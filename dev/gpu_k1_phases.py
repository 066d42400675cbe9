"""Development aid: where a channeliser workgroup spends its time (build with -DVDL2_K1_PROF: dev/gpu_run.sh <tag> k1phases:<C>).
Wave 0 of every workgroup stamps s_memtime at the phase boundaries of each tile; this prints shader clocks per tile and phase."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dumpvdl2_amd import vdl2hip, synth
L = vdl2hip.load_library(os.environ['VDL2HIP_LIB'])
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
reps = 3
cf = 136975000
freqs = synth.channel_plan(C, cf, max(8000, min(100000, 2000000 // C)))
n = int(secs * 2100000)
iq = (torch.randn(2 * n, device="cuda") * 300).to(torch.int16)
rx = vdl2hip.Receiver(cf, freqs, 20, 1, 3.0, max_block_bytes=iq.numel() * 2)
rx.set_profiling(1)
a = (ctypes.c_ulonglong * 16)()
for _ in range(2):
    rx.feed_device(iq.data_ptr(), iq.numel() * 2); rx.drain_packed()
L.vdl2hip_debug_k1_prof(a, 1)
s0 = rx.stats()
for _ in range(reps):
    rx.feed_device(iq.data_ptr(), iq.numel() * 2); rx.drain_packed()
s1 = rx.stats()
L.vdl2hip_debug_k1_prof(a, 0)
ms = (s1["chanfir_ms"] - s0["chanfir_ms"]) / reps
names = ["prologue (tables, first prefetch)", "staging: convert + LDS stores + next tile's loads issued", "barrier after staging", "sample loop + block updates",
         "scan + outputs + carry", "barrier before the tile is overwritten", "look-back wait (+ barrier)", "fix-up of the first tile"]
wgs = a[8] or 1
tiles = a[9] or 1
tot = sum(a[k] for k in range(8))
print(f"k_chanfir with the probe: {ms:.3f} ms/launch, C={C}; {wgs // reps} workgroups/launch, {tiles / wgs:.2f} tiles per workgroup; shader clocks of wave 0, per tile (phases 1-5) or per workgroup (0, 6, 7)")
for k in range(8):
    cnt = wgs if k in (0, 6, 7) else tiles
    print(f"  {k} {names[k]:58s} {a[k] / cnt:9.0f} clocks each = {100.0 * a[k] / tot:5.1f} % of the workgroup's life")
print(f"  workgroup life {tot / wgs:.0f} clocks = {tot / wgs / 2.3e3:.1f} us at 2.3 GHz")
