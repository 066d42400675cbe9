"""How far the decimated stream moves when EVERY channeliser workgroup takes the look-back fall-back (the segment-start state
recomputed as a sum over the previous segment's last tile instead of taken from the scan: kernels.h), and when the segment fix-up
runs as a kernel of its own (VDL2HIP_NO_FUSE) - per golden capture, over the channel's peak.  No PyTorch.  usage: python dev/gpu_fallback_diff.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.modules["torch"] = None
import numpy as np  # noqa: E402
from dumpvdl2_amd import vdl2hip  # noqa: E402
import cases  # noqa: E402

for name in ("config2_1s", "config4_0p4s", "os10_noisy_1s"):
    cfg, iq, _, gold = cases.load(name)
    D = iq.size // 2 // cfg.oversample
    ys = {}
    for tag, opt in (("normal", None), ("fallback", ("force_timeout", 1)), ("no_fuse", ("no_fuse", 1))):
        rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=iq.size * 2)
        if opt:
            rx.debug_option(*opt)
        rx.feed(iq)
        fr = rx.drain()
        cases.check_against_golden(fr, None, gold, label=f"{name}/{tag}")
        ys[tag] = np.stack([np.asarray(rx.read_decimated(c, 0, D), dtype=np.float64).reshape(-1, 2) for c in range(min(len(cfg.freqs), 16))])
        st = rx.stats()
        rx.close()
        if tag == "fallback":
            assert st["front_sync_timeouts"] > 0
    peak = np.abs(ys["normal"]).max()
    print(f"{name}: fall-back vs normal: max |diff| / peak = {np.abs(ys['fallback'] - ys['normal']).max() / peak:.3e};  "
          f"separate fix-up kernel vs fused: {np.abs(ys['no_fuse'] - ys['normal']).max() / peak:.3e}", flush=True)
