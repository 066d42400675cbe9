"""The source of the walker (K4) and burst-decoder (K5) kernels, compiled for the CPU by
tests/hostsim, against the oracle: fed with the oracle's own decimated samples it must
reproduce frames, timing, metadata and counters exactly, for any chunking of the stream."""
import numpy as np
import pytest

import cases
import pyhostsim
from util import assert_frames_equal


def run_both(oracle_mod, cfg, iq, chunks=None, cap_log2=None, segments=None, two_tier=False, reverse_lanes=False):
    C = len(cfg.freqs)
    o = oracle_mod.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
    D = iq.size // 2 // cfg.oversample
    tr = o.trace_all(D + 4)
    o.process(iq.view(np.uint8), block_bytes=1 << 24, nthreads=4)
    D = o.decimated_count(0)
    hs = pyhostsim.HostSim(list(cfg.freqs), cfg.rx_max_ppm, cap_log2=cap_log2 or int(np.ceil(np.log2(D + 70000))), reverse_lanes=reverse_lanes)
    if segments:
        hs.set_segments(*segments)
    if two_tier:
        hs.set_two_tier(True)
    if chunks is None:
        hs.feed(tr[:, :D, :])
    else:
        rng = np.random.default_rng(1); k = 0
        while k < D:
            m = min(D - k, int(rng.integers(*chunks))); hs.feed(tr[:, k:k + m, :]); k += m
    fo, fh = o.frames(), hs.frames()
    cnt_o = [list(o.counters(c).values()) for c in range(C)]
    cnt_h = [hs.counters(c) for c in range(C)]
    run_both.last_segment_stats = hs.segment_stats()
    run_both.last_two_tier_stats = hs.two_tier_stats()
    hs.close()
    return fo, fh, cnt_o, cnt_h


@pytest.mark.parametrize("name,chunks", [("config2_1s", None), ("config2_1s", (100, 30000)), ("config3_0p6s", None),
                                         ("config4_0p4s", (3000, 50000)), ("config5_0p4s", None),
                                         ("dirty25k_1s", (500, 20000)), ("os10_noisy_1s", (64, 5000))])
def test_device_logic_matches_oracle(oracle_mod, name, chunks):
    cfg, iq, _, _ = cases.load(name)
    fo, fh, co, ch = run_both(oracle_mod, cfg, iq, chunks, cap_log2=17 if chunks else None)
    assert_frames_equal(fo, fh, label=name)
    for a, b in zip(fo, sorted(fh, key=lambda f: (f["chan"], f["burst_ord"], f["idx"]))):
        pass
    assert co == ch
    # nf and ppm follow the reference's arithmetic exactly on identical input
    fo = sorted(fo, key=lambda f: (f["chan"], f["burst_ord"], f["idx"])); fh = sorted(fh, key=lambda f: (f["chan"], f["burst_ord"], f["idx"]))
    assert [f["nf_pwr_dbfs"] for f in fo] == [f["nf_pwr_dbfs"] for f in fh]
    assert [f["ppm_error"] for f in fo] == [f["ppm_error"] for f in fh]


@pytest.mark.parametrize("name,chunks,segments", [
    ("config2_1s", None, (5000, 32)), ("config2_1s", None, (1500, 32)), ("config2_1s", (20000, 60000), (700, 32)),
    ("config3_0p6s", None, (2500, 32)), ("config4_0p4s", None, (4000, 7)), ("config5_0p4s", None, (333, 32)),
    ("dirty25k_1s", None, (1000, 32)), ("dirty25k_1s", (3000, 40000), (450, 5)), ("os10_noisy_1s", None, (2000, 32)),
    ("os10_noisy_1s", None, (64, 32))])
def test_segmented_walk_matches_oracle(oracle_mod, name, chunks, segments):
    """The walk in speculative segments (k_walk_spec / k_walk_stitch) is exact: same frames, timing, counters and noise
    floor as the sequential FSM, wherever the segment boundaries fall (inside bursts, inside headers, in fresh intervals)."""
    cfg, iq, _, _ = cases.load(name)
    fo, fh, co, ch = run_both(oracle_mod, cfg, iq, chunks, cap_log2=18 if chunks else None, segments=segments)
    st = run_both.last_segment_stats
    assert st["adopted"] + st["walked"] > 0
    assert_frames_equal(fo, fh, label=name)
    assert co == ch
    fo = sorted(fo, key=lambda f: (f["chan"], f["burst_ord"], f["idx"])); fh = sorted(fh, key=lambda f: (f["chan"], f["burst_ord"], f["idx"]))
    assert [f["nf_pwr_dbfs"] for f in fo] == [f["nf_pwr_dbfs"] for f in fh]
    assert [f["ppm_error"] for f in fo] == [f["ppm_error"] for f in fh]
    print(name, segments, st)


def test_reference_wav(oracle_mod, golden_wav):
    cf = 136975000
    o = oracle_mod.Oracle(cf, [cf], oversample=10)
    tr = o.trace_all(len(golden_wav) // 40 + 4)
    o.process(golden_wav)
    D = o.decimated_count(0)
    hs = pyhostsim.HostSim([cf], 0.0, cap_log2=17)
    for k in range(0, D, 7001):
        hs.feed(tr[:, k:min(D, k + 7001), :])
    assert_frames_equal(o.frames(), hs.frames(), label="wav")


@pytest.mark.parametrize("name", ["config2_1s", "dirty25k_1s", "os10_noisy_1s", "config5_0p4s"])
def test_segmented_walk_random_geometry(oracle_mod, name):
    """Seeded sweep over segment lengths, segment counts and feed chunkings: the speculative walk must give the
    oracle's frames and counters for every geometry (boundaries land in bursts, headers, fresh intervals, feed ends)."""
    cfg, iq, _, _ = cases.load(name)
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    tot = {"adopted": 0, "walked": 0}
    for trial in range(5):
        seg_min = int(rng.choice([64, 150, 400, 1000, 2500, 6000]))
        seg_max = int(rng.integers(2, 33))
        chunks = None if rng.random() < 0.4 else (int(rng.integers(2 * seg_min, 6 * seg_min)), int(rng.integers(8 * seg_min, 40 * seg_min)))
        fo, fh, co, ch = run_both(oracle_mod, cfg, iq, chunks, cap_log2=19 if chunks else None, segments=(seg_min, seg_max))
        label = f"{name} seg_min={seg_min} seg_max={seg_max} chunks={chunks}"
        assert_frames_equal(fo, fh, label=label)
        assert co == ch, label
        st = run_both.last_segment_stats
        tot["adopted"] += st["adopted"]; tot["walked"] += st["walked"]
    assert tot["adopted"] > 0


@pytest.mark.parametrize("name,chunks,segments", [("config2_1s", None, None), ("config2_1s", (1, 3000), (700, 32)), ("config3_0p6s", None, (2500, 16)),
                                                  ("config4_0p4s", (3000, 50000), None), ("config5_0p4s", None, (1000, 32)),
                                                  ("dirty25k_1s", (500, 20000), None), ("os10_noisy_1s", (64, 5000), (64, 32))])
def test_two_tier_sync_metric_changes_nothing(oracle_mod, name, chunks, segments):
    """K3 stores the exact got_sync() value only where it can reach the walker (under the screening threshold, or 3 samples
    either side of such a place, or at the end of the data so far) and a cheaper screening value elsewhere.  Same rule on the
    CPU: frames, timing, counters, ppm and noise floor stay those of the oracle, for any chunking / segmentation, while only a
    small fraction of the samples gets the exact arithmetic."""
    cfg, iq, _, _ = cases.load(name)
    fo, fh, co, ch = run_both(oracle_mod, cfg, iq, chunks, cap_log2=17 if chunks else None, segments=segments, two_tier=True)
    assert_frames_equal(fo, fh, label=name)
    assert co == ch
    fo = sorted(fo, key=lambda f: (f["chan"], f["burst_ord"], f["idx"])); fh = sorted(fh, key=lambda f: (f["chan"], f["burst_ord"], f["idx"]))
    assert [f["nf_pwr_dbfs"] for f in fo] == [f["nf_pwr_dbfs"] for f in fh]
    assert [f["ppm_error"] for f in fo] == [f["ppm_error"] for f in fh]
    st = run_both.last_two_tier_stats
    assert st["total"] > 0 and st["exact"] < 0.2 * st["total"]
    print(name, st, st["exact"] / st["total"])


def test_silence_cut_into_a_capture(oracle_mod):
    """Stretches of exact zeros in the input (the filter output decays through the denormal range to exact zero) and a
    capture that ends in silence: the two-tier sync rule, the segmented walk and the noise-floor replay still reproduce the
    oracle bit for bit."""
    cfg, iq, bursts, _ = cases.load("config2_1s")
    x = cases.with_silence(cfg, iq, bursts)
    fo, fh, co, ch = run_both(oracle_mod, cfg, x.reshape(-1), chunks=(500, 40000), segments=(1500, 16), two_tier=True)
    assert len(fo) >= 4
    assert_frames_equal(fo, fh, label="silence cut in")
    assert co == ch
    key = lambda f: (f["chan"], f["burst_ord"], f["idx"])
    assert [f["nf_pwr_dbfs"] for f in sorted(fo, key=key)] == [f["nf_pwr_dbfs"] for f in sorted(fh, key=key)]


@pytest.mark.parametrize("seed", [5019, 5029, 5041, 1002, 1018])
def test_fuzz_regressions(oracle_mod, seed):
    """Seeds of tests/fuzz_hostsim.py kept as regressions.  5019/5029/5041: a sync fired just before a segment end while the
    stitcher was walking sequentially; the header phases gathered speculatively at the fire were bounded by that segment's end
    (zeros beyond it) and were still used when the next call, with a later end, decoded the header.  1002/1018: heavy
    cross-talk with many sequentially walked segments."""
    import fuzz_hostsim
    fuzz_hostsim.run_seed(seed)


@pytest.mark.parametrize("name,chunks,segments", [("config2_1s", (100, 30000), (700, 32)), ("config5_0p4s", None, (1000, 16)),
                                                  ("dirty25k_1s", None, None), ("os10_noisy_1s", (64, 5000), (500, 32))])
def test_lane_order_does_not_matter(oracle_mod, name, chunks, segments):
    """On the device the 64 lanes of a wave phase run together; the host build runs them one after the other.  Run backwards
    (-DVDL2_HOST_REVERSE_LANES) the walker, the stitcher, the noise-floor replay, the burst decoder and the frame finisher
    must give the same frames and counters - a phase in which one lane consumed what another lane produced would be a race
    on the GPU that the forward host order hides."""
    cfg, iq, _, _ = cases.load(name)
    fo, fh, co, ch = run_both(oracle_mod, cfg, iq, chunks, cap_log2=17 if chunks else None, segments=segments, two_tier=True, reverse_lanes=True)
    assert_frames_equal(fo, fh, label=name)
    assert co == ch


def test_unwrap_alternatives_short_cut_contains_the_brute_force_range():
    """sync_metric_unwrap_alts() (the exact sync tier's `full` form, receivers that scan ahead of the walk): the values the sync metric takes
    with one or two unwrap decisions the other way, worked out from the residuals instead of running the metric again - its range must
    contain what the metric gives with those decisions forced, and exceed it by no more than its own slack."""
    import ctypes as C
    import pyhostsim
    L = C.CDLL(pyhostsim.build())
    L.hostsim_check_unwrap_alts.argtypes = [C.c_int, C.c_uint, C.POINTER(C.c_double)]
    worst = C.c_double(0)
    assert L.hostsim_check_unwrap_alts(200000, 7, C.byref(worst)) == 0
    assert worst.value < 1.5, worst.value
