"""Burst-format edge cases: synthetic transmitter -> oracle (encode -> decode round trips)."""
import numpy as np
import pytest

from dumpvdl2_amd import synth

CF = 136975000


def one_burst_iq(frames, os_=10, sep=False, errs=None, hflips=0, seed=1, amp=0.25, sigma=0.003, lead=3000):
    rng = np.random.default_rng(seed)
    bb = synth.build_burst(frames, rng, errs, hflips, separate_flags=sep)
    w = synth.modulate(bb.symbols, 10 * os_, start_phase=0.7)
    n = lead + w.size + 4000
    x = np.zeros(n, dtype=np.complex128)
    x[lead:lead + w.size] = amp * w * np.exp(1j * 2 * np.pi * 150.0 / (105000 * os_) * np.arange(w.size))
    x += sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    iq = np.empty(2 * n, dtype=np.float64)
    iq[0::2] = x.real; iq[1::2] = x.imag
    return np.clip(np.rint(iq * 32768), -32768, 32767).astype(np.int16), bb


def decode(oracle_mod, iq, os_=10):
    o = oracle_mod.Oracle(CF, [CF], oversample=os_)
    o.process(iq.view(np.uint8))
    return o, o.frames()


@pytest.mark.parametrize("nbytes", [9, 11, 25, 28, 29, 60, 65, 66, 240, 244, 245, 249, 250, 251, 495, 498, 1000, 1980])
def test_block_geometry_round_trip(oracle_mod, nbytes):
    # sizes straddle the FEC steps (3/31/68 octets, decode.c:124-133), the 249-octet block edge
    # (v2.5.1 fix, decode.c:244-245) and the 2-octet last block that carries no FEC at all
    rng = np.random.default_rng(nbytes)
    frames = [synth.make_avlc_frame(rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes())]
    iq, bb = one_burst_iq(frames, seed=nbytes)
    o, fr = decode(oracle_mod, iq)
    assert [f["octets"] for f in fr] == frames
    assert fr[0]["datalen_octets"] == bb.datalen_octets and fr[0]["num_fec_corrections"] == 0


def test_stuffing_heavy_payload(oracle_mod):
    frames = [synth.make_avlc_frame(bytes([0xFF] * 120)), synth.make_avlc_frame(bytes([0x7E, 0x7D, 0xFE, 0xFC] * 40))]
    for sep in (False, True):
        iq, _ = one_burst_iq(frames, sep=sep)
        _, fr = decode(oracle_mod, iq)
        assert [f["octets"] for f in fr] == frames and [f["idx"] for f in fr] == [0, 1]


def test_rs_corrections_are_counted(oracle_mod):
    rng = np.random.default_rng(3)
    frames = [synth.make_avlc_frame(rng.integers(0, 256, 600, dtype=np.uint8).tobytes())]
    iq, bb = one_burst_iq(frames, errs=[3, 2, 1], hflips=1, seed=3)
    o, fr = decode(oracle_mod, iq)
    assert [f["octets"] for f in fr] == frames
    assert fr[0]["num_fec_corrections"] == 6 and fr[0]["synd_weight"] == 1


def test_uncorrectable_block_drops_burst(oracle_mod):
    rng = np.random.default_rng(4)
    frames = [synth.make_avlc_frame(rng.integers(0, 256, 300, dtype=np.uint8).tobytes())]
    iq, bb = one_burst_iq(frames, errs=[4, 0], seed=4)
    o, fr = decode(oracle_mod, iq)
    assert fr == [] and o.counters(0)["decoder.errors.fec_bad"] == 1


def test_oversample_20_offset_channel(oracle_mod):
    cfg = synth.SynthConfig(centerfreq=CF, freqs=[CF + 25000, CF - 250000], oversample=20, duration_s=0.8, seed=21)
    iq, bursts = synth.synthesize(cfg)
    o = oracle_mod.Oracle(CF, list(cfg.freqs), oversample=20)
    o.process(iq.view(np.uint8))
    got = sorted((f["chan"], f["sync_sample"], f["idx"], f["octets"]) for f in o.frames())
    want = synth.expected_frames(bursts)
    assert [g[3] for g in got] == [w[3] for w in want] and [g[0] for g in got] == [w[0] for w in want]


def test_u8_samples(oracle_mod):
    cfg = synth.SynthConfig(centerfreq=CF, freqs=[CF], oversample=10, duration_s=0.6, seed=8, amplitude=0.3, noise_sigma=0.01)
    iq, bursts = synth.synthesize(cfg, dtype=np.uint8)
    o = oracle_mod.Oracle(CF, [CF], oversample=10, sample_fmt=oracle_mod.FMT_U8)
    o.process(iq)
    assert [f["octets"] for f in o.frames()] == [w[3] for w in synth.expected_frames(bursts)]
