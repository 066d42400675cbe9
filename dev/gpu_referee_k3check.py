"""development aid: the sync kernels' verdicts (candidate bits, referee marks, tabulated metric) of one channel against the host build's on the device's own samples"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
import numpy as np
import pyhostsim
from dumpvdl2_amd import synth, vdl2hip, workloads
name, dur, ch = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
cfg = getattr(workloads, name)(dur)
iq, _ = synth.synthesize(cfg); raw = iq.view(np.uint8)
D = raw.size // 4 // cfg.oversample
rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, 1, cfg.rx_max_ppm, max_block_bytes=raw.size)
rx.debug_option("ref_kinds", 0)            # marks are made, no scan is run: the stream stays the channeliser's
rx.feed(raw); rx.drain()
y = rx.read_decimated(ch, 0, D)
pf, cand = rx.read_sync(ch, 0, D)
hs = pyhostsim.HostSim([cfg.freqs[ch]], cfg.rx_max_ppm, cap_log2=int(np.ceil(np.log2(D + 70000))))
hs.set_two_tier(True); hs.set_exact(y[None, :, :])
hs.feed(y[None, :, :])
pfh, candh = hs.read_sync(0, 0, D)
both = (np.abs(pfh[:, 0]) < 999) & (np.abs(pf[:, 0]) < 999) & (pfh[:, 0] != 12345.0)
print("candidate bits: device", int(cand.sum()), "host", int(candh.sum()), "differ at", np.flatnonzero(cand != candh)[:20].tolist())
comp = np.flatnonzero((pfh[:, 0] != 12345.0) & (pfh[:, 0] < 999) & (np.arange(D) > 200))
dm = np.flatnonzero((np.abs(pf[comp, 0]) != np.abs(pfh[comp, 0])) | (pf[comp, 1] != pfh[comp, 1]))
print("samples the host tabulates:", comp.size, "; metric/slope differ on", dm.size, comp[dm][:10].tolist(), [(float(pf[i, 0]), float(pfh[i, 0])) for i in comp[dm][:5]])
ms = np.flatnonzero(np.signbit(pf[comp, 0]) != np.signbit(pfh[comp, 0]))
print("marks differ on", ms.size, comp[ms][:20].tolist(), [(float(pf[i, 0]), float(pfh[i, 0]), int(cand[i]), int(candh[i])) for i in comp[ms][:8]])
