"""How often is a referee scan NOT the reference's own samples, and does the product notice?

A scan (k_ref_scan_multi) re-runs the reference's fp32 recursion from a zero state 2^17 input samples before the stretch it is
asked for; it is the reference's trajectory from the sample on at which the two have become bit-identical.  Every scan
carries a witness trajectory (another start state); a scan that has not met its witness by the stretch's first output is listed and
run again from further back (VDL2HIP_REF_RETRY, default 2x).  This driver asks for N disjoint 256-sample stretches of a synthetic
capture (config3, 64 channels), all with a full run-up, in batches of 4096 side by side, and compares every stretch bit for bit
with the oracle's decimated stream:

  passes A17..A20  scans without the second try (the test hook's plain launch) with run-ups of 2^17 .. 2^20 input samples: how many
                   stretches differ, by how much, how many of those the witness flagged
  pass B           the same stretches on a fresh receiver with the product's run-up and second try: how many still differ

usage: python dev/gpu_scan_soundness.py [--seconds 4] [--scans 106496] [--out gpurun_out/scan_soundness.txt]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--scans", type=int, default=26 * 4096)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from dumpvdl2_amd import synth, vdl2hip, workloads
    from oracle import pyoracle as po
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    cfg = workloads.config3(args.seconds)
    iq, _ = synth.synthesize(cfg)
    raw = iq.view(np.uint8)
    nch = len(cfg.freqs)
    t0 = time.time()
    o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, sample_fmt=vdl2hip.FMT_S16LE, max_ppm=cfg.rx_max_ppm)
    D = raw.size // 4 // cfg.oversample
    tr = o.trace_all(D + 4)
    o.process(raw, block_bytes=1 << 24, nthreads=8)
    o.close()
    say(f"# capture: config3, {args.seconds:g} s, {nch} channels, {D} decimated samples per channel; oracle stream in {time.time() - t0:.0f} s")
    warm_dec = (1 << 17) // cfg.oversample
    los = np.arange((8 * warm_dec + 255) // 256 * 256, D - 512, 256, dtype=np.int64)       # every stretch has its whole run-up (and a retry's) in the capture
    pairs = np.stack(np.meshgrid(np.arange(nch), los, indexing="ij"), -1).reshape(-1, 2)
    rng = np.random.default_rng(20261001)
    rng.shuffle(pairs)
    n = min(args.scans, len(pairs))
    pairs = pairs[:n]
    say(f"# {n} disjoint stretches of 256 samples (of {len(los) * nch} possible), batches of {args.batch}")
    results = {}
    passes = [(f"A{k}: run-up 2^{k}, no second try", 1 << k, False) for k in (17, 18, 19, 20)] + [("A196608: the product's run-up, no second try", 196608, False), ("B: the product's run-up (196 608) and second try", 196608, True)]
    for label, warm, retry in passes:
        os.environ["VDL2HIP_REF_WARM"] = str(warm)
        rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=raw.size)
        rx.feed(raw)
        rx.drain()
        s0 = rx.stats()
        bad = []
        ran = 0
        ms_tot = 0.0
        for b in range(0, n, args.batch):
            pb = pairs[b:b + args.batch]
            r, ms = rx.scan_multi(pb[:, 0], pb[:, 1], pb[:, 1] + 255, retry=retry)
            ran += r
            ms_tot += ms
            for c, lo in pb:
                got = rx.read_decimated(int(c), int(lo), 256)
                want = tr[int(c), int(lo):int(lo) + 256]
                if got.tobytes() != want.tobytes():
                    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
                    bad.append((int(c), int(lo), float(d.max() / max(1e-30, np.abs(want).max())), int((d.sum(axis=1) > 0).sum()), float(np.abs(want).max())))
        s1 = rx.stats()
        d = {k: s1[k] - s0[k] for k in ("referee_scans", "referee_cached", "referee_refused", "referee_short", "referee_unmet", "referee_retried")}
        rx.close()
        results[label] = (bad, d)
        say(f"{label}: {ran} scans run in {ms_tot:.0f} ms of kernel time ({ms_tot / max(1, (n + args.batch - 1) // args.batch):.2f} ms per batch); "
            f"stretches that are NOT the oracle's bit for bit: {len(bad)} ({len(bad) / n:.2e}); counted unmet (published as they were): {d['referee_unmet']}; "
            f"listed and run again: {d['referee_retried']}; cached {d['referee_cached']}, refused {d['referee_refused']}, shortened run-ups {d['referee_short']}")
        if bad:
            rel = np.array([x[2] for x in bad]); nsm = np.array([x[3] for x in bad]); amp = np.array([x[4] for x in bad])
            chans = sorted(set(x[0] for x in bad))
            say(f"   differing stretches: largest |difference| / largest |sample| of the stretch: median {np.median(rel):.1e}, max {rel.max():.1e}; differing samples per stretch: median {int(np.median(nsm))} of 256; "
                f"largest |sample|: median {np.median(amp):.2e} (all stretches: {float(np.median(np.abs(tr[:, ::4096]).max(axis=2))):.2e}); on {len(chans)} channels")
        if label.startswith("A17") and bad:
            for x in bad[:12]:
                say(f"   e.g. channel {x[0]} at {x[1]}: rel {x[2]:.1e}, {x[3]} samples differ, peak {x[4]:.2e}")
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
