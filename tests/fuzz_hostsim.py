"""Development aid (CPU only): random synthetic captures through the host-compiled device logic (tests/hostsim, with the
speculative walk, K3's two-tier rule and random chunking) against the oracle.  usage: python tests/fuzz_hostsim.py [n] [seed0] [extreme|rejects]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
import numpy as np
from dumpvdl2_amd import synth
from oracle import pyoracle as po
import pyhostsim
from util import assert_frames_equal


def run_seed(seed, extreme=False, rejects=False):
    """one random capture through hostsim (random segmentation, two-tier on/off, random chunking) against the oracle;
    returns a description string, raises AssertionError on any difference"""
    rng = np.random.default_rng(seed)
    nch = int(rng.choice([1, 2, 3, 5]))
    os_ = int(rng.choice([10, 13, 20]))
    spacing = int(rng.choice([25000, 50000, 100000]))
    cfg = synth.SynthConfig(freqs=synth.channel_plan(nch, spacing=spacing), oversample=os_, duration_s=float(rng.uniform(0.6, 1.6)),
                            seed=seed, mean_gap_s=float(rng.choice([0.01, 0.05, 0.15])), max_payload=int(rng.choice([60, 300, 1000, 1980])),
                            noise_sigma=float(rng.choice([0.0005, 0.002, 0.01, 0.02])), error_injection=bool(rng.random() < 0.4),
                            invalid_frame_rate=float(rng.choice([0.0, 0.3])), max_ppm=float(rng.choice([0.5, 2.0, 8.0])),
                            rx_max_ppm=float(rng.choice([0.0, 0.0, 3.0])))
    if extreme:      # back-to-back bursts, many frames per burst, loud and faint signals, a burst right at the start, longer runs
        cfg.mean_gap_s = float(rng.choice([0.001, 0.004, 0.02])); cfg.max_frames = int(rng.choice([1, 3, 8]))
        cfg.amplitude = float(rng.choice([0.01, 0.05, 0.4])); cfg.first_burst_s = float(rng.choice([0.0, 0.0005, 0.02]))
        cfg.duration_s = float(rng.uniform(0.3, 2.5)); cfg.min_payload = int(rng.choice([9, 20]))
        cfg.noise_sigma = float(rng.choice([0.0005, 0.004, 0.012, 0.03]))
    if rejects:      # dense channel plans behind the --max-ppm gate: leaked preambles that lock and are dropped, in clusters (the walker's reject chain)
        nch = int(rng.choice([3, 5, 8]))
        cfg.freqs = synth.channel_plan(nch, spacing=int(rng.choice([8000, 12000, 25000])))
        cfg.rx_max_ppm = float(rng.choice([0.5, 1.0, 2.5])); cfg.max_ppm = float(rng.choice([0.3, 2.0, 6.0]))
        cfg.noise_sigma = float(rng.choice([0.0005, 0.002])); cfg.mean_gap_s = float(rng.choice([0.004, 0.02, 0.05]))
        if rng.random() < 0.6:     # time-division slots as in the 256-channel workloads: a channel is silent while its neighbours send, so it locks on their leakage
            cfg.tdm_slots = int(rng.choice([2, 4])); cfg.tdm_slot_s = float(rng.choice([0.02, 0.05])); cfg.max_payload = int(rng.choice([60, 300]))
    iq, bursts = synth.synthesize(cfg)
    o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
    D = iq.size // 2 // cfg.oversample
    tr = o.trace_all(D + 4)
    o.process(iq.view(np.uint8), block_bytes=1 << 24, nthreads=4)
    D = o.decimated_count(0)
    hs = pyhostsim.HostSim(list(cfg.freqs), cfg.rx_max_ppm, cap_log2=19)
    seg_min = int(rng.choice([64, 300, 1500, 6000, 25000] if extreme else [64, 300, 1500, 6000])); hs.set_segments(seg_min, int(rng.integers(2, 33)))
    hs.set_two_tier(bool(rng.random() < 0.7))
    t = 0
    while t < D:
        m = min(D - t, int(rng.integers(2 * seg_min, max(2 * seg_min + 1, min(60 * seg_min, 400000)))) if rng.random() < 0.8 else int(rng.integers(1, 500)))
        hs.feed(tr[:, t:t + m, :]); t += m
    fo, fh = o.frames(), hs.frames()
    try:
        assert_frames_equal(fo, fh, label=f"seed {seed}")
        key = lambda f: (f["chan"], f["burst_ord"], f["idx"])
        so, sh_ = sorted(fo, key=key), sorted(fh, key=key)
        # fed with the oracle's own decimated samples the device logic is bit-exact, floats included
        assert [(f["nf_pwr_dbfs"], f["ppm_error"]) for f in so] == [(f["nf_pwr_dbfs"], f["ppm_error"]) for f in sh_], f"seed {seed}: nf/ppm differ"
        nch = len(cfg.freqs)
        co = [list(o.counters(c).values()) for c in range(nch)]; ch = [hs.counters(c) for c in range(nch)]
        # the 18 counters the reference keeps + demod.ppm_reject must be identical; demod.slicer_neg_idx (a diagnostic of this
        # implementation) is counted by the burst decoder, so symbols of a burst still incomplete at the end of the capture -
        # which the sequential oracle has already sliced - are not in it yet
        assert [c[:19] for c in co] == [c[:19] for c in ch], f"seed {seed}: counters differ"
        assert all(a[19] >= b[19] for a, b in zip(co, ch)), f"seed {seed}: slicer_neg_idx over-counted"
        assert po.avlc_counters(fh, nch) == [hs.avlc_counters(c) for c in range(nch)], f"seed {seed}: avlc counters differ"
        return f"ch={len(cfg.freqs)} os={os_} frames={len(fo)} bursts={len(bursts)} syncs={sum(c[0] for c in co)} ppm_rejects={sum(c[18] for c in co)} seg={hs.segment_stats()}"
    finally:
        hs.close()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    extreme = len(sys.argv) > 3 and sys.argv[3] == "extreme"
    rejects = len(sys.argv) > 3 and sys.argv[3] == "rejects"
    po.build()
    bad = 0
    for k in range(n):
        try:
            print(f"seed {seed0 + k}: ok  {run_seed(seed0 + k, extreme, rejects)}", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed0 + k}: MISMATCH {str(e)[:300]}", flush=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
