"""tests/predict_gpu_parity.py's prediction for a bench workload (e.g. config4, 16 s, 256 channels): the channel filter in double
precision per channel (a process pool; the capture is shared by fork), the host build of the device logic behind it, against the
oracle with bench.py's own gate (tests/util.compare_at_full_size / compare_reference_counters).
usage: python dev/predict_workload.py [config4|config4_bursty|config5] [seconds] [procs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim")); sys.path.insert(0, os.path.join(ROOT, "dev"))
import numpy as np  # noqa: E402

G = {}


def _chan(c):
    import predict_gpu_parity as p
    y = p.exact_stream(G["cfg"], G["raw"], 1, G["A"], G["B"], [G["dphi"][c]], G["D"])
    return c, y[0]


def main():
    import multiprocessing as mp
    import pyhostsim
    from dumpvdl2_amd import synth, workloads
    from oracle import pyoracle as po
    from util import compare_at_full_size, compare_reference_counters
    name = sys.argv[1] if len(sys.argv) > 1 else "config4"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    cfg = getattr(workloads, name)(secs)
    t0 = time.time()
    iq, _ = synth.synthesize(cfg)
    raw = iq.view(np.uint8)
    nch = len(cfg.freqs)
    o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
    o.process(raw, block_bytes=1 << 24, nthreads=os.cpu_count() or 8)
    fo = o.frames()
    names = list(o.counters(0).keys())
    co = [list(o.counters(c).values()) for c in range(nch)]
    D = o.decimated_count(0)
    A, B = o.lpf()
    G.update(cfg=cfg, raw=raw, A=A, B=B, D=D, dphi=[o.dphi(c) for c in range(nch)])
    print(f"# {name} {secs:g} s: {nch} channels, {len(fo)} oracle frames, {D} decimated samples per channel ({time.time() - t0:.0f} s)", flush=True)
    y = np.zeros((nch, D, 2), dtype=np.float32)
    with mp.get_context("fork").Pool(procs) as pool:
        for c, yc in pool.imap_unordered(_chan, range(nch), chunksize=2):
            y[c] = yc
    print(f"# filter in double precision done ({time.time() - t0:.0f} s)", flush=True)
    cap = 21
    while (1 << cap) < D + 70000:
        cap += 1
    hs = pyhostsim.HostSim(list(cfg.freqs), cfg.rx_max_ppm, cap_log2=cap)
    hs.set_segments(16384, 10)
    hs.feed(y)
    got = hs.frames()
    cg = [list(hs.counters(c)) for c in range(nch)]
    st = compare_at_full_size(fo, got, label=name, max_tie_frac=0.5)
    which, nbad = compare_reference_counters(names, co, cg, label=name, strict=False, max_channels=nch)
    print(f"{name} {secs:g} s, exact-arithmetic channel filter + host build of the device logic vs the oracle: {st}; bookkeeping differences {which} on {nbad} channels "
          f"({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
