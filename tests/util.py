"""Shared helpers for the parity tests."""
import numpy as np

EXACT_KEYS = ("chan", "idx", "octets", "synd_weight", "datalen_octets", "num_fec_corrections", "burst_ord")
# tolerances from SURVEY.md 8.5 (floats are compared, not bit-matched: frame power and the noise floor are sums formed in another
# order)
TOL_DB = 0.05
TOL_PPM = 0.01


def frame_key(f):
    return (f["chan"], f["burst_ord"], f["idx"])


def assert_frames_equal(ref, got, exact_samples=True, label=""):
    """ref/got: lists of frame dicts.  Octets and integer metadata must match exactly."""
    ref = sorted(ref, key=frame_key)
    got = sorted(got, key=frame_key)
    assert len(ref) == len(got), f"{label}: frame count {len(got)} != reference {len(ref)}"
    for a, b in zip(ref, got):
        for k in EXACT_KEYS:
            assert a[k] == b[k], f"{label}: frame {frame_key(a)} field {k}: {b[k]!r} != {a[k]!r}"
        if exact_samples:
            assert a["sync_sample"] == b["sync_sample"] and a["end_sample"] == b["end_sample"], \
                f"{label}: frame {frame_key(a)} timing {b['sync_sample']},{b['end_sample']} != {a['sync_sample']},{a['end_sample']}"
        assert abs(a["frame_pwr_dbfs"] - b["frame_pwr_dbfs"]) <= TOL_DB, f"{label}: frame_pwr {a['frame_pwr_dbfs']} vs {b['frame_pwr_dbfs']}"
        assert abs(a["nf_pwr_dbfs"] - b["nf_pwr_dbfs"]) <= TOL_DB, f"{label}: nf_pwr {a['nf_pwr_dbfs']} vs {b['nf_pwr_dbfs']}"
        assert abs(a["ppm_error"] - b["ppm_error"]) <= TOL_PPM, f"{label}: ppm {a['ppm_error']} vs {b['ppm_error']}"


def compare_at_full_size(ref, got, label="", max_tie_frac=0.0):
    """The parity gate of SURVEY 8.5 for runs with thousands of bursts: (channel, burst ordinal, idx, octets), the integer metadata
    and the burst timing (sync_sample / end_sample - diagnostics of this repo, not reference metadata) identical, floats within
    tolerance (0.01 ppm, 0.05 dB).  No exceptions: decisions that hang on the reference's own rounding are taken on the reference's
    own samples (the referee).
    max_tie_frac > 0 is for diagnosing a run WITHOUT the referee (dev/ scripts, VDL2HIP_REFEREE=0): the channeliser's stream
    differs from the reference's sequential one by its rounding noise, and where calc_para_vertex lands within that of a rounding
    boundary the sync point moves by one or two decimated samples (a "tie": <= 2 samples, looser ppm bound, and the noise-floor update
    a frame sees may shift by one); such frames are counted and must stay under the fraction.  Returns the statistics."""
    ref = sorted(ref, key=frame_key)
    got = sorted(got, key=frame_key)
    assert len(ref) == len(got), f"{label}: frame count {len(got)} != reference {len(ref)}"
    ties = nf_ties = 0
    worst = {"frame_pwr_db": 0.0, "nf_pwr_db": 0.0, "ppm": 0.0}
    on_ties = {"sync_samples": 0, "end_samples": 0, "ppm": 0.0, "nf_pwr_db": 0.0}     # what the frames counted as ties REALLY differ by
    for a, b in zip(ref, got):
        for k in EXACT_KEYS:
            assert a[k] == b[k], f"{label}: frame {frame_key(a)} field {k}: {b[k]!r} != {a[k]!r}"
        tie = a["sync_sample"] != b["sync_sample"] or a["end_sample"] != b["end_sample"]
        if tie:
            ties += 1
            on_ties["sync_samples"] = max(on_ties["sync_samples"], abs(a["sync_sample"] - b["sync_sample"]))
            on_ties["end_samples"] = max(on_ties["end_samples"], abs(a["end_sample"] - b["end_sample"]))
            on_ties["ppm"] = max(on_ties["ppm"], abs(a["ppm_error"] - b["ppm_error"]))
            assert abs(a["sync_sample"] - b["sync_sample"]) <= 2 and abs(a["end_sample"] - b["end_sample"]) <= 2, \
                f"{label}: frame {frame_key(a)} timing {b['sync_sample']},{b['end_sample']} != {a['sync_sample']},{a['end_sample']}"
        assert abs(a["frame_pwr_dbfs"] - b["frame_pwr_dbfs"]) <= TOL_DB, f"{label}: frame_pwr {a['frame_pwr_dbfs']} vs {b['frame_pwr_dbfs']}"
        # the noise floor a frame reports is mag_nf after (evaluations so far) / 1000 updates (demod.c:240-243, decode.c:181): a tie
        # anywhere earlier on the channel (also at a preamble the --max-ppm gate dropped) shifts the evaluation grid, hence the
        # evaluation count, by one - a burst that synchronises right at an update then sees the value before / after it
        dnf = abs(a["nf_pwr_dbfs"] - b["nf_pwr_dbfs"])
        if dnf > TOL_DB:
            nf_ties += 1
            on_ties["nf_pwr_db"] = max(on_ties["nf_pwr_db"], dnf)
            assert dnf <= 1.5, f"{label}: nf_pwr {a['nf_pwr_dbfs']} vs {b['nf_pwr_dbfs']}"
        else:
            worst["nf_pwr_db"] = max(worst["nf_pwr_db"], dnf)
        assert abs(a["ppm_error"] - b["ppm_error"]) <= (0.5 if tie else TOL_PPM), f"{label}: ppm {a['ppm_error']} vs {b['ppm_error']} (tie={tie})"
        if not tie:
            worst["ppm"] = max(worst["ppm"], abs(a["ppm_error"] - b["ppm_error"]))
        worst["frame_pwr_db"] = max(worst["frame_pwr_db"], abs(a["frame_pwr_dbfs"] - b["frame_pwr_dbfs"]))
    lim = max(1, int(max_tie_frac * len(ref))) if max_tie_frac > 0 else 0
    assert ties <= lim and nf_ties <= lim, f"{label}: {ties} / {nf_ties} of {len(ref)} frames differ in burst timing / noise-floor update"
    return {"frames": len(ref), "timing_ties": ties, "nf_update_ties": nf_ties, "max_abs_diff": {k: round(v, 6) for k, v in worst.items()},
            "max_abs_diff_on_ties": {k: (round(v, 6) if isinstance(v, float) else v) for k, v in on_ties.items()}}


# failure bookkeeping of a burst that delivers nothing: how far the decoder got before it gave up (decode.c:266-334, 345-370)
BOOKKEEPING = ("decoder.blocks.processed", "decoder.blocks.fec_ok", "decoder.errors.fec_bad", "decoder.errors.unstuff",
               "decoder.errors.truncated_octets", "decoder.errors.bitstream")


def compare_reference_counters(names, per_channel_ref, per_channel_got, label="", strict=True, nref=18, max_channels=None):
    """The reference's 18 statsd counters, channel by channel.  strict: identical, all of them.  Otherwise the only exception allowed
    is the one DESIGN 5 describes - a lock on a neighbour's leaked preamble that the --max-ppm gate lets pass slices its symbols out
    of noise, delivers no frame in either implementation, and WHERE its decoding dies (which RS block first fails, which stuffing rule)
    can hinge on one symbol decision at the 1e-5 by which the time-parallel filter differs from the sequential one: on at most
    `max_channels` channels (default 1 %, at least 1) the BOOKKEEPING counters may differ by at most 2 each, every other counter
    (locks, header verdicts, delivered messages ...) stays identical.  Returns ({counter: summed difference}, channels that differ)."""
    which, nbad = {}, 0
    lim = max_channels if max_channels is not None else max(1, len(per_channel_ref) // 100)
    for ch, (co, cg) in enumerate(zip(per_channel_ref, per_channel_got)):
        if co[:nref] == cg[:nref]:
            continue
        assert not strict, f"{label}: reference counters of channel {ch} differ from the oracle's: {co} vs {cg}"
        nbad += 1
        for i in range(nref):
            if co[i] != cg[i]:
                assert names[i] in BOOKKEEPING and abs(co[i] - cg[i]) <= 2, f"{label}: channel {ch} counter {names[i]}: {cg[i]} vs oracle {co[i]}"
                which[names[i]] = which.get(names[i], 0) + abs(co[i] - cg[i])
    assert nbad <= lim, f"{label}: the failure bookkeeping differs on {nbad} channels (allowed {lim})"
    return which, nbad


def frames_multiset(frames):
    return sorted((f["chan"], f["idx"], f["octets"]) for f in frames)


def truth_is_subset(bursts, frames):
    """Every frame of every decodable transmitted burst must appear on its channel, in order."""
    from collections import defaultdict
    got = defaultdict(list)
    for f in sorted(frames, key=lambda f: (f["chan"], f["end_sample"], f["idx"])):
        got[f["chan"]].append(f["octets"])
    missing = 0
    for ch in set(b.chan for b in bursts):
        want = [fr for b in sorted((b for b in bursts if b.chan == ch), key=lambda b: b.start_sample) if b.decodable for fr in b.frames]
        have = got.get(ch, [])
        pos = 0
        for w in want:
            try:
                pos = have.index(w, pos) + 1
            except ValueError:
                missing += 1
    return missing


def run_oracle(po, cfg, raw, nthreads=4, trace=False):
    import numpy as np
    o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
    tr = None
    if trace:
        D = raw.size // 4 // cfg.oversample
        tr = o.trace_all(D + 4)
    o.process(np.ascontiguousarray(raw).view(np.uint8), block_bytes=1 << 24, nthreads=nthreads)
    fr = o.frames()
    return o, fr, tr
