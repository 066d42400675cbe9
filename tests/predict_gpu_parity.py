"""CPU prediction of how often the GPU path's answer differs from the oracle's, over many random captures (tests/fuzz_gpu.py's seeds).

Since the channeliser's state went into normal form the GPU's decimated stream is the filter evaluated EXACTLY, to 2e-7 of the
peak; what separates it from the oracle's stream is the rounding noise of the reference's own sequential fp32 scan
(~1e-5 rms).  Everything behind the channeliser is bit-exact with the oracle on equal samples (tests/hostsim is the same source as
the device logic).  So the GPU's answer can be predicted without a GPU: the channel filter in DOUBLE precision (numpy / scipy, the
reference's table mixer and coefficients) -> float32 -> the host build of the device logic -> frames and counters, against the
oracle with tests/fuzz_gpu.py's own comparison.  A capture that differs here is one where a decision of the reference hinges on its
own rounding noise; no time-parallel implementation can be expected to agree on it.

Checked against the GPU on the twelve seeds of profiles/r04_parity_ab_state_basis.txt (--known).
usage: python tests/predict_gpu_parity.py --known | <seed0> <count> [procs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
import numpy as np  # noqa: E402

KNOWN = ((55, "plain"), (100, "plain"), (104, "extreme"), (145, "plain"), (175, "plain"), (179, "extreme"), (274, "plain"),
         (1001, "plain"), (1014, "extreme"), (1041, "extreme"), (1292, "plain"), (2274, "plain"))
GPU_NORMAL_FORM = {55: "ok", 100: "ok", 104: "ok", 145: "ok", 175: "differs", 179: "ok", 274: "differs", 1001: "ok", 1014: "differs",
                   1041: "ok", 1292: "ok", 2274: "ok"}          # profiles/r04_parity_ab_state_basis.txt, second line
PROFILES = ("plain", "extreme", "rejects")


def exact_stream(cfg, raw, fmt, A, B, dphis, D):
    """the channel filter of src/demod.c:302-329 in double precision: table mixer (sincosf_lut, entries float), 2-pole IIR, decimation"""
    from scipy.signal import lfilter
    os_ = cfg.oversample
    if fmt == 1:
        v = raw.view(np.int16).reshape(-1, 2).astype(np.float32) / np.float32(32768.0)
    else:
        v = (raw.reshape(-1, 2).astype(np.float32) - np.float32(127.5)) / np.float32(127.5)
    n = D * os_
    x = v[:n, 0].astype(np.float64) + 1j * v[:n, 1].astype(np.float64)
    i = np.arange(257, dtype=np.float32)
    ang = (np.float32(2.0) * np.float32(np.pi) * (i % 256) / np.float32(256.0)).astype(np.float32)
    sl = np.sin(ang.astype(np.float64)).astype(np.float32).astype(np.float64); cl = np.cos(ang.astype(np.float64)).astype(np.float32).astype(np.float64)
    b = [float(A[0]), float(A[1]), float(A[2])]; a = [1.0, -float(B[1]), -float(B[2])]
    idxn = np.arange(n, dtype=np.uint64)
    y = np.zeros((len(dphis), D, 2), dtype=np.float32)
    for c, dphi in enumerate(dphis):
        if dphi & 0xffffff:
            ph = ((idxn * np.uint64(dphi & 0xffffff)) & np.uint64(0xffffff)).astype(np.int64)
            k = ph >> 16; f = (ph & 0xffff).astype(np.float64) / 65536.0
            xm = x * ((cl[k] + (cl[k + 1] - cl[k]) * f) + 1j * (sl[k] + (sl[k + 1] - sl[k]) * f))
        else:
            xm = x
        z = lfilter(b, a, xm)[os_ - 1::os_][:D]
        y[c, :, 0] = z.real; y[c, :, 1] = z.imag
    return y


def run_seed(seed, profile, referee=False, prescan=False, pieces=1):
    """referee: the host build asks its referee - the oracle's own decimated stream stands in for the device's sequential scan - and the
    comparison is strict (no tie allowances, all 18 counters on every channel)"""
    import fuzz_gpu
    import pyhostsim
    from dumpvdl2_amd import synth, vdl2hip
    from oracle import pyoracle as po
    from util import compare_at_full_size, compare_reference_counters
    cfg, rng = fuzz_gpu.make_cfg(seed, profile)
    iq, _ = synth.synthesize(cfg)
    raw = iq.view(np.uint8)
    nch = len(cfg.freqs)
    rng2 = np.random.default_rng(seed + 7_000_000)
    fmt = vdl2hip.FMT_S16LE
    if seed >= 2000 and rng2.random() < 0.25:
        fmt = vdl2hip.FMT_U8
        raw = np.clip(np.rint(iq.astype(np.float64) / 256.0 + 127.5), 0, 255).astype(np.uint8)
    o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, sample_fmt=fmt, max_ppm=cfg.rx_max_ppm)
    tr = o.trace_all(raw.size // (4 if fmt == vdl2hip.FMT_S16LE else 2) // cfg.oversample + 4) if referee else None
    o.process(raw, block_bytes=1 << 24, nthreads=2)
    fo = o.frames()
    names = list(o.counters(0).keys())
    co = [list(o.counters(c).values()) for c in range(nch)]
    D = o.decimated_count(0)
    A, B = o.lpf()
    y = exact_stream(cfg, raw, fmt, A, B, [o.dphi(c) for c in range(nch)], D)
    hs = pyhostsim.HostSim(list(cfg.freqs), cfg.rx_max_ppm, cap_log2=21)
    hs.set_segments(6000, 8)
    if referee:
        hs.set_exact(tr[:, :D, :]); hs.set_prescan(prescan)      # (prescan: the device's VDL2HIP_REF_PRESCAN=1 - marked candidates' stretches made exact ahead of the walk)
    step = (D + pieces - 1) // pieces            # (pieces > 1: several feeds - the walk's state, its snapshot and the noted decisions cross feed boundaries)
    for k in range(0, D, step):
        hs.feed(np.ascontiguousarray(y[:, k:k + step, :]))
    got = hs.frames()
    cg = [list(hs.counters(c)) for c in range(nch)]
    rst = hs.referee_stats() if referee else {}
    hs.close(); o.close()
    label = f"seed {seed} {profile}"
    try:
        st = compare_at_full_size(fo, got, label=label, max_tie_frac=0.0 if referee else 0.02)
        which, nbad = compare_reference_counters(names, co, cg, label=label, strict=referee, max_channels=max(1, nch // 4))
        return seed, profile, "ok", {"frames": len(fo), "ties": st["timing_ties"], "nf_ties": st["nf_update_ties"], "bookkeeping_channels": nbad, "referee": rst}
    except AssertionError as e:
        return seed, profile, "differs", {"frames": len(fo), "why": str(e)[:200]}


REFEREE = False
PRESCAN = False
PIECES = 1


def _job(a):
    try:
        return run_seed(*a, referee=REFEREE, prescan=PRESCAN, pieces=PIECES)
    except Exception as e:  # noqa: BLE001
        return a[0], a[1], "error", {"why": f"{type(e).__name__}: {str(e)[:200]}"}


def main():
    global REFEREE, PRESCAN, PIECES
    if "--prescan" in sys.argv:          # ... with the scans ahead of the walk (VDL2HIP_REF_PRESCAN=1)
        PRESCAN = True; sys.argv.remove("--prescan")
    if "--pieces" in sys.argv:           # the capture in that many feeds
        i = sys.argv.index("--pieces"); PIECES = int(sys.argv[i + 1]); del sys.argv[i:i + 2]
    if "--referee" in sys.argv:          # the host build with its referee, compared strictly (what the device does by default)
        REFEREE = True; sys.argv.remove("--referee")
    if sys.argv[1] == "--known":
        agree = 0
        for seed, profile in KNOWN:
            s, p, verdict, info = _job((seed, profile))
            same = verdict == GPU_NORMAL_FORM[seed]
            agree += same
            print(f"seed {seed} {profile}: predicted {verdict}, GPU (normal form) {GPU_NORMAL_FORM[seed]}{'' if same else '   <-- prediction and GPU disagree'}  {info}", flush=True)
        print(f"prediction = GPU on {agree} of {len(KNOWN)} seeds")
        return
    seed0, count = int(sys.argv[1]), int(sys.argv[2])
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, (os.cpu_count() or 2) // 2)
    jobs = [(seed0 + i, PROFILES[i % 3]) for i in range(count)]
    import multiprocessing as mp
    t0 = time.time()
    tot = {"seeds": 0, "frames": 0, "ties": 0, "nf_ties": 0, "bookkeeping_channels": 0, "differ": 0, "errors": 0, "referee": REFEREE, "referee_windows": 0, "marked_candidates": 0}
    with mp.Pool(procs) as pool:
        for seed, profile, verdict, info in pool.imap(_job, jobs, chunksize=1):
            tot["seeds"] += 1
            if verdict == "ok":
                for k in ("frames", "ties", "nf_ties", "bookkeeping_channels"):
                    tot[k] += info[k]
                tot["referee_windows"] += info.get("referee", {}).get("exact_windows", 0); tot["marked_candidates"] += info.get("referee", {}).get("marked_candidates", 0)
            elif verdict == "differs":
                tot["differ"] += 1; tot["frames"] += info.get("frames", 0)
                print(f"seed {seed} {profile}: predicted to differ: {info['why']}", flush=True)
            else:
                tot["errors"] += 1
                print(f"seed {seed} {profile}: ERROR {info['why']}", flush=True)
    print(f"SUMMARY seeds {seed0}..{seed0 + count - 1} ({time.time() - t0:.0f} s, {procs} processes): {tot}", flush=True)


if __name__ == "__main__":
    main()
