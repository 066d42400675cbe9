"""debugging aid: one fuzz seed through the device with the referee on, against the oracle, in detail"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
import numpy as np
import fuzz_gpu
from dumpvdl2_amd import synth, vdl2hip
from oracle import pyoracle as po
seed, profile = int(sys.argv[1]), sys.argv[2]
cfg, _ = fuzz_gpu.make_cfg(seed, profile)
iq, _ = synth.synthesize(cfg); raw = iq.view(np.uint8); nch = len(cfg.freqs)
o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
o.process(raw, block_bytes=1 << 24, nthreads=8)
fo = o.frames(); co = [list(o.counters(c).values())[:18] for c in range(nch)]
key = lambda f: (f["chan"], f["burst_ord"], f["idx"])
for chunk in (raw.size, 1 << 20, 320000):
    for referee in (0, 1):
        rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, 1, cfg.rx_max_ppm, max_block_bytes=raw.size)
        rx.debug_option("referee", referee)
        for k in range(0, raw.size, chunk):
            rx.feed(raw[k:k + chunk])
        fr = rx.drain(); cg = [list(rx.counters(c).values())[:18] for c in range(nch)]
        a = {key(f): f for f in fo}; b = {key(f): f for f in fr}
        bad = []
        for k in sorted(set(a) | set(b)):
            if k not in a or k not in b: bad.append((k, "missing in " + ("oracle" if k not in a else "device"))); continue
            for fld in ("octets", "synd_weight", "datalen_octets", "num_fec_corrections", "sync_sample", "end_sample"):
                if a[k][fld] != b[k][fld]:
                    bad.append((k, fld, len(a[k]["octets"]) if fld == "octets" else a[k][fld], len(b[k]["octets"]) if fld == "octets" else b[k][fld])); break
        cb = [c for c in range(nch) if co[c] != cg[c]]
        s = rx.stats()
        print(f"chunk {chunk} referee {referee}: {len(fo)} / {len(fr)} frames, differing {bad[:6]}, counters differ on channels {cb}, scans {s['referee_scans']} cached {s['referee_cached']} refused {s['referee_refused']}", flush=True)
        rx.close()
print("--- which kind of request breaks it (1 candidate, 2 header, 4 symbols)")
for kinds in (1, 2, 4, 3):
    rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, 1, cfg.rx_max_ppm, max_block_bytes=raw.size)
    rx.debug_option("ref_kinds", kinds)
    rx.feed(raw)
    fr = rx.drain(); cg = [list(rx.counters(c).values())[:18] for c in range(nch)]
    a = {key(f): f for f in fo}; b = {key(f): f for f in fr}
    nbad = sum(1 for k in set(a) | set(b) if k not in a or k not in b or a[k]["octets"] != b[k]["octets"] or a[k]["num_fec_corrections"] != b[k]["num_fec_corrections"] or a[k]["sync_sample"] != b[k]["sync_sample"])
    s = rx.stats()
    print(f"kinds {kinds}: {len(fr)} frames, {nbad} differ, counters differ on {sum(co[c] != cg[c] for c in range(nch))} channels, scans {s['referee_scans']}", flush=True)
    rx.close()
