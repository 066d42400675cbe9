"""development aid: a bench workload (a few seconds) through the device with the referee, against the oracle - strictly (frames, timing,
integer metadata, the reference's 18 counters), with the kinds of requests switched on one by one"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dumpvdl2_amd import synth, vdl2hip, workloads
from oracle import pyoracle as po
name, dur = sys.argv[1], float(sys.argv[2])
kinds_list = [int(k) for k in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 7]
cfg = getattr(workloads, name)(dur)
iq, _ = synth.synthesize(cfg); raw = iq.view(np.uint8); nch = len(cfg.freqs)
o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
o.run(raw, mode=po.RUN_WORKQUEUE)
fo = o.frames(); co = [list(o.counters(c).values())[:18] for c in range(nch)]
key = lambda f: (f["chan"], f["burst_ord"], f["idx"])
a = {key(f): f for f in fo}
for kinds in kinds_list:
    rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, 1, cfg.rx_max_ppm, max_block_bytes=raw.size)
    rx.debug_option("referee", 1 if kinds else 0); rx.debug_option("ref_kinds", kinds or 7)
    t0 = time.perf_counter(); rx.feed(raw); fr = rx.drain(); dt = time.perf_counter() - t0
    cg = [list(rx.counters(c).values())[:18] for c in range(nch)]
    b = {key(f): f for f in fr}
    bad = []
    for k in sorted(set(a) | set(b)):
        if k not in a or k not in b: bad.append((k, "missing in " + ("oracle" if k not in a else "device"))); continue
        for fld in ("octets", "synd_weight", "datalen_octets", "num_fec_corrections", "sync_sample", "end_sample"):
            if a[k][fld] != b[k][fld]:
                bad.append((k, fld, len(a[k]["octets"]) if fld == "octets" else a[k][fld], len(b[k]["octets"]) if fld == "octets" else b[k][fld])); break
    dppm = max([abs(a[k]["ppm_error"] - b[k]["ppm_error"]) for k in a if k in b] or [0]); dnf = max([abs(a[k]["nf_pwr_dbfs"] - b[k]["nf_pwr_dbfs"]) for k in a if k in b] or [0])
    cb = [c for c in range(nch) if co[c] != cg[c]]
    s = rx.stats()
    print(f"{name} {dur}s kinds {kinds}: {len(fo)} / {len(fr)} frames, {len(bad)} differ {bad[:5]}, counters differ on {len(cb)} channels {cb[:6]}, max ppm diff {dppm:.2e}, nf diff {dnf:.2e} dB; "
          f"scans {s['referee_candidate_scans']}/{s['referee_header_scans']}/{s['referee_symbol_scans']} refused {s['referee_refused']}; {dt * 1e3:.1f} ms", flush=True)
    rx.close()
