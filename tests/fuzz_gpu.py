"""Development aid (needs a GPU): random synthetic captures through libvdl2hip.so - fed in random pieces, so that long feeds
(speculative walk, back end on its own streams) and short ones (whole back end on the front stream, k_nf_burst) follow each other in
one stream, with a random drain lag - against the oracle.  The host-build fuzz (tests/fuzz_hostsim.py) proves the logic; this one
covers what only the device can get wrong: wavefront-scope phase syncs, stream hand-offs between feeds of different kinds, the
per-wavefront output shares and the compaction behind them.

usage: python tests/fuzz_gpu.py [seconds] [seed0] [plain|extreme|rejects|all]      one line per seed, a summary at the end;
exit status 1 if any seed differs.

Per seed: frames (octets, integer metadata) identical, floats within SURVEY 8.5's tolerances, burst timing identical except for
counted ties (tests/util.compare_at_full_size), the reference's 18 counters identical except for the failure bookkeeping of bursts
that deliver nothing (util.compare_reference_counters, not strict) - both exceptions are tallied in the summary."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dumpvdl2_amd import synth, vdl2hip
from oracle import pyoracle as po
from util import compare_at_full_size, compare_reference_counters


def make_cfg(seed, profile):
    rng = np.random.default_rng(seed)
    nch = int(rng.choice([1, 2, 3, 5, 8, 16, 24]))
    os_ = int(rng.choice([10, 13, 20, 20]))
    spacing = int(rng.choice([25000, 50000, 100000]))
    if nch * spacing > 0.8 * 105000 * os_:
        spacing = 25000
    if nch * spacing > 0.8 * 105000 * os_:
        nch = max(1, int(0.8 * 105000 * os_ / spacing))
    cfg = synth.SynthConfig(freqs=synth.channel_plan(nch, spacing=spacing), oversample=os_, duration_s=float(rng.uniform(0.5, 1.6)),
                            seed=seed, mean_gap_s=float(rng.choice([0.01, 0.05, 0.15])), max_payload=int(rng.choice([60, 300, 1000, 1980])),
                            noise_sigma=float(rng.choice([0.0005, 0.002, 0.01, 0.02])), error_injection=bool(rng.random() < 0.4),
                            invalid_frame_rate=float(rng.choice([0.0, 0.3])), max_ppm=float(rng.choice([0.5, 2.0, 8.0])),
                            rx_max_ppm=float(rng.choice([0.0, 0.0, 3.0])))
    if profile == "extreme":
        cfg.mean_gap_s = float(rng.choice([0.001, 0.004, 0.02])); cfg.max_frames = int(rng.choice([1, 3, 8]))
        cfg.amplitude = float(rng.choice([0.01, 0.05, 0.4])); cfg.first_burst_s = float(rng.choice([0.0, 0.0005, 0.02]))
        cfg.duration_s = float(rng.uniform(0.3, 2.0)); cfg.min_payload = int(rng.choice([9, 20]))
        cfg.noise_sigma = float(rng.choice([0.0005, 0.004, 0.012, 0.03]))
    elif profile == "rejects":
        n2 = int(rng.choice([3, 5, 8, 16]))
        cfg.freqs = synth.channel_plan(n2, spacing=int(rng.choice([8000, 12000, 25000])))
        cfg.rx_max_ppm = float(rng.choice([0.5, 1.0, 2.5])); cfg.max_ppm = float(rng.choice([0.3, 2.0, 6.0]))
        cfg.noise_sigma = float(rng.choice([0.0005, 0.002])); cfg.mean_gap_s = float(rng.choice([0.004, 0.02, 0.05]))
        if rng.random() < 0.6:
            cfg.tdm_slots = int(rng.choice([2, 4])); cfg.tdm_slot_s = float(rng.choice([0.02, 0.05])); cfg.max_payload = int(rng.choice([60, 300]))
    return cfg, rng


def run_seed(seed, profile):
    cfg, rng = make_cfg(seed, profile)
    iq, _ = synth.synthesize(cfg)
    raw = iq.view(np.uint8)
    nch = len(cfg.freqs)
    o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
    o.process(raw, block_bytes=1 << 24, nthreads=8)
    fo = o.frames()
    # the pieces: mostly a few long feeds with runs of short ones (the reference's own 320 000-byte blocks, odd sizes, tiny ones) between them
    style = int(rng.integers(0, 4))
    big = int(rng.choice([1 << 20, 3 << 20, 8 << 20]))
    rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=max(big, 1 << 20))
    lag = int(rng.integers(0, 3))
    rx.set_drain_lag(lag)
    got = []
    t, nfeeds, nsmall = 0, 0, 0
    while t < raw.size:
        r = rng.random()
        if style == 0:
            m = 320000
        elif style == 1:
            m = big
        elif r < 0.35:
            m = big if r < 0.2 else int(rng.integers(1 << 19, big + 1))
        elif r < 0.8:
            m = int(rng.choice([320000, 320000, 262144, 65536, 100000]))
        else:
            m = int(rng.integers(4, 40000))
        m = max(4, min(m, raw.size - t) & ~3) if raw.size - t >= 4 else raw.size - t
        if m <= 0:
            break
        rx.feed(raw[t:t + m]); t += m; nfeeds += 1
        nsmall += (m // 4 // cfg.oversample) < 32768
        if rng.random() < 0.7:
            got += vdl2hip.Receiver.unpack(*rx.drain_packed())
    rx.set_drain_lag(0)
    got += vdl2hip.Receiver.unpack(*rx.drain_packed())
    label = f"seed {seed} {profile}"
    st = compare_at_full_size(fo, got, label=label, max_tie_frac=0.02)
    names = list(o.counters(0).keys())
    co = [list(o.counters(c).values()) for c in range(nch)]
    cg = [list(rx.counters(c).values()) for c in range(nch)]
    which, nbad = compare_reference_counters(names, co, cg, label=label, strict=False, max_channels=max(1, nch // 4))
    s = rx.stats()
    assert s["overflow_feeds"] == 0, f"{label}: overflow"
    rx.close()
    return {"frames": len(fo), "ties": st["timing_ties"], "nf_ties": st["nf_update_ties"], "bookkeeping_channels": nbad, "feeds": nfeeds,
            "short_feeds": int(nsmall), "lag": lag, "nch": nch, "os": cfg.oversample, "fallbacks": s["front_sync_timeouts"]}


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    which = sys.argv[3] if len(sys.argv) > 3 else "all"
    profiles = ["plain", "extreme", "rejects"] if which == "all" else [which]
    t0 = time.time()
    tot = {"seeds": 0, "frames": 0, "ties": 0, "nf_ties": 0, "bookkeeping_channels": 0, "feeds": 0, "short_feeds": 0, "failed": 0}
    i = 0
    while time.time() - t0 < budget:
        seed, profile = seed0 + i, profiles[i % len(profiles)]
        i += 1
        try:
            r = run_seed(seed, profile)
        except AssertionError as e:
            tot["failed"] += 1
            print(f"seed {seed} {profile}: DIFFERS: {str(e)[:400]}", flush=True)
            continue
        tot["seeds"] += 1
        for k in ("frames", "ties", "nf_ties", "bookkeeping_channels", "feeds", "short_feeds"):
            tot[k] += r[k]
        print(f"seed {seed} {profile}: ok {r}", flush=True)
    print(f"SUMMARY ({time.time() - t0:.0f} s): {tot}", flush=True)
    sys.exit(1 if tot["failed"] else 0)


if __name__ == "__main__":
    main()
