"""Development aid (needs a GPU): random synthetic captures through libvdl2hip.so - fed in random pieces, so that long feeds
(speculative walk, back end on its own streams) and short ones (whole back end on the front stream, k_nf_burst) follow each other in
one stream, with a random drain lag - against the oracle.  The host-build fuzz (tests/fuzz_hostsim.py) proves the logic; this one
covers what only the device can get wrong: wavefront-scope phase syncs, stream hand-offs between feeds of different kinds, the
per-wavefront output shares and the compaction behind them.

usage: python tests/fuzz_gpu.py [seconds] [seed0] [plain|extreme|rejects|all]      one line per seed, a summary at the end;
exit status 1 if any seed fails (see below).

Per seed: frames (octets, integer metadata), burst timing and the reference's 18 counters IDENTICAL to the oracle's on every channel,
floats within SURVEY 8.5's tolerances (tests/util.compare_at_full_size / compare_reference_counters, strict).  Any seed that differs
fails the run.

These captures are harsher than the bench workloads on purpose (noise up to the decoding threshold, injected symbol errors, bursts
cut off by the next one): they are full of decisions that hinge on one symbol, and the channeliser's samples differ from the
reference's by the reference's own rounding noise (~1e-5 rms).  Those decisions are the referee's: it takes them on the
reference's own samples.  With VDL2HIP_REFEREE=0 in the environment the run shows what it is there for: a seed in fifty or so then
differs from the oracle in a frame or a counter; the comparison then allows counted ties and the failure bookkeeping of bursts that
deliver nothing, and a seed that differs is decided, not shrugged off - the decimated stream is read back from the GPU and run
through the HOST build of the device logic (tests/hostsim, bit-exact with the oracle on the oracle's samples); if that reproduces
the GPU's frames, timing and counters exactly, everything behind the channeliser did on the GPU what the reference does with those
samples ("explained by the samples"); if not, it is a defect and the run fails.  Every fourth agreeing seed gets the same check."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
import numpy as np
from dumpvdl2_amd import synth, vdl2hip
from oracle import pyoracle as po
from util import compare_at_full_size, compare_reference_counters

STRICT = os.environ.get("VDL2HIP_REFEREE", "1") != "0"
EXACT = ("chan", "burst_ord", "idx", "octets", "synd_weight", "datalen_octets", "num_fec_corrections", "sync_sample", "end_sample")


class Differs(AssertionError):
    """the GPU's answer differs from the oracle's; .from_samples: the host build of the device logic, fed with the GPU's own
    decimated samples, gives the GPU's answer exactly - i.e. everything behind the channeliser did what the reference does with
    those samples, and the difference is the ~1e-5 by which the channeliser's samples differ from the reference's"""
    def __init__(self, msg, from_samples, rel):
        super().__init__(msg); self.from_samples, self.rel = from_samples, rel


def device_logic_on_device_samples(rx, cfg, D, got, cg):
    """-> (True/False, description): tests/hostsim on the decimated stream read back from the GPU against the GPU's frames and counters"""
    import pyhostsim
    nch = len(cfg.freqs)
    y = np.stack([rx.read_decimated(c, 0, D).reshape(-1, 2) for c in range(nch)])
    hs = pyhostsim.HostSim(list(cfg.freqs), cfg.rx_max_ppm, cap_log2=20)
    hs.set_segments(6000, 8)
    hs.feed(y)
    key = lambda f: tuple(f[k] for k in EXACT)
    a, b = sorted(key(f) for f in hs.frames()), sorted(key(f) for f in got)
    ch = [list(hs.counters(c))[:18] for c in range(nch)]
    cgg = [c[:18] for c in cg]
    if a != b:
        return False, f"frames: host-build-only {[(t[0], t[1], t[2], len(t[3])) + t[4:] for t in sorted(set(a) - set(b))][:3]} gpu-only {[(t[0], t[1], t[2], len(t[3])) + t[4:] for t in sorted(set(b) - set(a))][:3]}", y
    if ch != cgg:
        return False, f"counters: {[(c, ch[c], cgg[c]) for c in range(nch) if ch[c] != cgg[c]][:2]}", y
    return True, "", y


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


def run_seed(seed, profile, always_check=False, strict=None):
    strict = STRICT if strict is None else strict
    cfg, rng = make_cfg(seed, profile)
    iq, _ = synth.synthesize(cfg)
    raw = iq.view(np.uint8)
    nch = len(cfg.freqs)
    # seeds from 2000 on also vary the sample format (the RTL-SDR's offset-binary u8, process_buf_uchar) and the receiver: one
    # context, or a vdl2hip_group of 2-3 virtual shards on this GPU with either exchange form (seeds below 2000 stay what they were)
    rng2 = np.random.default_rng(seed + 7_000_000)
    fmt, sb = vdl2hip.FMT_S16LE, 4
    ndev, form = 1, None
    if seed >= 2000:
        if rng2.random() < 0.25:
            fmt, sb = vdl2hip.FMT_U8, 2
            raw = np.clip(np.rint(iq.astype(np.float64) / 256.0 + 127.5), 0, 255).astype(np.uint8)
        if rng2.random() < 0.25 and nch >= 3:
            ndev, form = int(rng2.choice([2, 3])), str(rng2.choice(["allgather", "broadcast"]))
    o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, sample_fmt=fmt, max_ppm=cfg.rx_max_ppm)
    o.process(raw, block_bytes=1 << 24, nthreads=8)
    fo = o.frames()
    # the pieces: mostly a few long feeds with runs of short ones (the reference's own 320 000-byte blocks, odd sizes, tiny ones) between them
    style = int(rng.integers(0, 4))
    big = int(rng.choice([1 << 20, 3 << 20, 8 << 20]))
    # (max_block_bytes >= the capture: the whole decimated stream then stays inside the device's history ring for the check below)
    if ndev == 1:
        rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, fmt, cfg.rx_max_ppm, max_block_bytes=max(big, raw.size))
        take = lambda: vdl2hip.Receiver.unpack(*rx.drain_packed())
    else:
        rx = vdl2hip.ReceiverGroup(cfg.centerfreq, list(cfg.freqs), [0] * ndev, cfg.oversample, fmt, cfg.rx_max_ppm, max_block_bytes=max(big, raw.size))
        rx.set_exchange(form)
        take = rx.drain
    lag = int(rng.integers(0, 3))
    # seeds from 3000 on (round 6): up to six feeds in flight, and every third capture in LONG pieces only (each several walk segments:
    # the walks then run ahead of the checks, second walks are compared with the next feed's start - vdl2hip.hip: launch_rest)
    all_long = False
    if seed >= 3000:
        lag = int(rng2.integers(0, vdl2hip.MAX_DRAIN_LAG + 1))
        all_long = rng2.random() < 0.34
    rx.set_drain_lag(lag)
    got = []
    t, nfeeds, nsmall = 0, 0, 0
    while t < raw.size:
        r = rng.random()
        if all_long:
            m = int(rng2.integers(700_000, 3_000_000)) * sb
        elif style == 0:
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
        nsmall += (m // sb // cfg.oversample) < 32768
        if rng.random() < 0.7:
            got += take()
    rx.set_drain_lag(0)
    got += take()
    label = f"seed {seed} {profile}"
    names = list(o.counters(0).keys())
    co = [list(o.counters(c).values()) for c in range(nch)]
    cg = [list(rx.counters(c).values()) for c in range(nch)]
    s = rx.stats()
    assert s["overflow_feeds"] == 0, f"{label}: overflow"
    D = o.decimated_count(0)
    try:
        st = compare_at_full_size(fo, got, label=label, max_tie_frac=0.0 if strict else 0.02)
        which, nbad = compare_reference_counters(names, co, cg, label=label, strict=strict, max_channels=max(1, nch // 4))
    except AssertionError as e:
        ok, why, y = device_logic_on_device_samples(rx, cfg, D, got, cg)
        tr = o2_trace(cfg, raw, D, fmt)
        rms = np.sqrt((tr.astype(np.float64) ** 2).sum(axis=2).mean(axis=1))
        rel = float((np.sqrt(((y.astype(np.float64) - tr) ** 2).sum(axis=2).mean(axis=1)) / rms).max())
        rx.close()
        raise Differs(str(e) + ("" if ok else f" | AND the device logic differs from its host build on the same samples: {why}"), ok, rel)
    check = None
    if always_check or seed % 4 == 0:      # every fourth seed that agrees with the oracle gets the same check of the back end against its host build
        ok, why, _ = device_logic_on_device_samples(rx, cfg, D, got, cg)
        assert ok, f"{label}: agrees with the oracle but the device logic differs from its host build on the device's samples: {why}"
        check = True
    rx.close()
    return {"frames": len(fo), "ties": st["timing_ties"], "nf_ties": st["nf_update_ties"], "bookkeeping_channels": nbad, "feeds": nfeeds,
            "short_feeds": int(nsmall), "lag": lag, "nch": nch, "os": cfg.oversample, "fallbacks": s["front_sync_timeouts"], "host_build_check": check,
            "fmt": "u8" if fmt == vdl2hip.FMT_U8 else "s16", "referee_scans": s.get("referee_scans", 0), "referee_refused": s.get("referee_refused", 0),
            "referee_unmet": s.get("referee_unmet", 0), "referee_rewalks": s.get("referee_rewalks", 0), "referee_redone_next": s.get("referee_redone_next", 0), "all_long": all_long, "receiver": "one context" if ndev == 1 else f"group of {ndev}, {form}"}


def o2_trace(cfg, raw, D, fmt):
    """the oracle's decimated samples of the capture (a second oracle run with tracing on)"""
    o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, sample_fmt=fmt, max_ppm=cfg.rx_max_ppm)
    tr = o.trace_all(D + 4)
    o.process(raw, block_bytes=1 << 24, nthreads=8)
    return tr[:, :D, :]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    which = sys.argv[3] if len(sys.argv) > 3 else "all"
    profiles = ["plain", "extreme", "rejects"] if which == "all" else [which]
    t0 = time.time()
    tot = {"seeds": 0, "frames": 0, "ties": 0, "nf_ties": 0, "bookkeeping_channels": 0, "feeds": 0, "short_feeds": 0, "failed": 0,
           "differ_from_oracle": 0, "of_those_explained_by_the_samples": 0, "host_build_checks": 0, "referee_scans": 0, "referee_refused": 0,
           "referee_unmet": 0, "referee_rewalks": 0, "referee_redone_next": 0, "strict": STRICT}
    i = 0
    while time.time() - t0 < budget:
        seed, profile = seed0 + i, profiles[i % len(profiles)]
        i += 1
        try:
            r = run_seed(seed, profile)
        except Differs as e:
            tot["seeds"] += 1; tot["differ_from_oracle"] += 1; tot["of_those_explained_by_the_samples"] += bool(e.from_samples)
            tot["failed"] += (not e.from_samples) or STRICT
            print(f"seed {seed} {profile}: DIFFERS from the oracle ({'device logic == its host build on the device samples' if e.from_samples else 'DEVICE LOGIC DEFECT'}; "
                  f"samples differ from the oracle's by {e.rel:.2e} rms relative): {str(e)[:500]}", flush=True)
            continue
        except AssertionError as e:
            tot["failed"] += 1
            print(f"seed {seed} {profile}: FAILED: {str(e)[:500]}", flush=True)
            continue
        tot["seeds"] += 1; tot["host_build_checks"] += bool(r.get("host_build_check"))
        for k in ("frames", "ties", "nf_ties", "bookkeeping_channels", "feeds", "short_feeds", "referee_scans", "referee_refused", "referee_unmet", "referee_rewalks", "referee_redone_next"):
            tot[k] += r[k]
        print(f"seed {seed} {profile}: ok {r}", flush=True)
    print(f"SUMMARY ({time.time() - t0:.0f} s): {tot}", flush=True)
    sys.exit(1 if tot["failed"] else 0)


if __name__ == "__main__":
    main()
