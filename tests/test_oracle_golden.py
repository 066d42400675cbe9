"""Pins the CPU oracle to the reference's own test vector and recorded known answers."""
import numpy as np
import pytest


def test_reference_wav_two_frames(oracle_mod, golden_wav):
    # the reference's CI runs exactly this input and greps two strings (.github/workflows/build.yml:16-18,57-62);
    # intermediates below were recorded from the compiled reference in SURVEY.md section 4
    cf = 136975000
    o = oracle_mod.Oracle(cf, [cf], oversample=10)
    o.process(golden_wav)      # whole file incl. the 44-byte RIFF header, as the reference reads it
    fr = o.frames()
    assert [len(f["octets"]) for f in fr] == [314, 186]
    assert fr[0]["octets"][:12].hex() == "b2107684948a341f22544146" and fr[0]["octets"][-3:].hex() == "0a44bf"
    assert fr[1]["octets"][:12].hex() == "b2107684948a341f344d4554" and fr[1]["octets"][-3:].hex() == "0a3ef9"
    assert b" -RA BR OVC005\n" in fr[0]["octets"] and b" SLP135\n" in fr[1]["octets"]
    for f in fr:
        assert oracle_mod.crc16_x25(f["octets"]) == 0xF0B8          # avlc.c:40,177 good-FCS residual
        assert (f["synd_weight"], f["datalen_octets"], f["num_fec_corrections"]) == (0, 504, 0)
        assert abs(f["frame_pwr_dbfs"] - (-9.841)) < 1e-3 and abs(f["nf_pwr_dbfs"] - 1.799) < 1e-3
        assert f["sync_sample"] == 11972 + 3                        # "Preamble found at" prints samplenum - SYNC_SKIP
    c = o.counters(0)
    assert c["demod.sync.good"] == 1 and c["decoder.blocks.processed"] == 3 and c["decoder.blocks.fec_ok"] == 3
    assert c["decoder.msg.good"] == 2


def test_filter_coefficients_kat(oracle_mod):
    # strict-IEEE values recorded from the reference build
    import ctypes as C
    L = oracle_mod.lib()
    for os_, a0, b1, b2 in [(20, "0x1.0e48p-13", "0x1.f8215p+0", "-0x1.f08632p-1"),
                            (10, "0x1.0a28p-11", "0x1.f03d0ap+0", "-0x1.e1843cp-1")]:
        A = (C.c_float * 3)(); B = (C.c_float * 3)()
        L.vdl2o_chebyshev(C.c_float(8000.0 / (105000.0 * os_)), C.c_float(0.5), A, B)
        assert float(A[0]) == float.fromhex(a0) and float(A[2]) == float.fromhex(a0)
        assert float(A[1]) == 2 * float.fromhex(a0)
        assert float(B[1]) == float.fromhex(b1) and float(B[2]) == float.fromhex(b2)


def test_nco_step_fp32_rounding(oracle_mod):
    # demod.c:385 converts both frequencies to fp32 first: 250 000 Hz becomes 250 016 Hz
    cf = 136975000
    o = oracle_mod.Oracle(cf, [136725000, 137000000], oversample=20)
    fs = 2100000
    assert o.dphi(0) == int(np.float32(np.float32(cf) - np.float32(136725000)) / np.float32(fs) * np.float32(256.0) * np.float32(65536.0))
    assert abs(o.dphi(0) / 2 ** 24 * fs - 250016) < 1
    assert o.dphi(1) == (int(-25000 / fs * 2 ** 24) & 0xFFFFFFFF) or abs(((o.dphi(1) ^ 0xFFFFFFFF) + 1) / 2 ** 24 * fs - 25000) < 20


def test_header_code_table(oracle_mod):
    import ctypes as C
    L = oracle_mod.lib()
    # synd_weight row of decode.c:98-100 (data): weight of the pattern each syndrome corrects
    weight = [0, 1, 1, 2, 1, 2, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 2, 1, 2, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1]
    seen = set()
    for tl in (0, 1, 4029, 0x1FFFF):
        word = (tl << 5) | L.vdl2o_header_parity(tl)
        w = C.c_uint32(word)
        assert L.vdl2o_header_decode(C.byref(w)) == 0 and w.value == word
        for bit in range(25):                      # every single-bit error is corrected
            w = C.c_uint32(word ^ (1 << bit))
            s = L.vdl2o_header_decode(C.byref(w))
            assert w.value == word and weight[s] == 1
            seen.add(s)
    assert len(seen) == 25
    assert sorted(set(range(1, 32)) - seen) == [3, 5, 13, 18, 20, 23]   # the six double-error syndromes


def _expect_offset_tuned(frames, delta, crc16):
    """What the reference-held data says about the capture moved by `delta` Hz and received with centerfreq = f - delta:
    the CI's two frames (octets, FCS), S:0 L:504 F:0 and the burst power of the on-centre run (-9.841 dBFS, SURVEY 4: the filter
    passes the burst only if the NCO brought it back to 0 Hz), and a carrier offset that differs from the on-centre one
    (dphi = -0.005778 rad/symbol -> -0.0705 ppm) by exactly the error of the fp32 NCO step of demod.c:385."""
    import shift_wav as sw
    assert [len(f["octets"]) for f in frames] == [314, 186]
    assert frames[0]["octets"][:12].hex() == "b2107684948a341f22544146" and frames[0]["octets"][-3:].hex() == "0a44bf"
    assert frames[1]["octets"][:12].hex() == "b2107684948a341f344d4554" and frames[1]["octets"][-3:].hex() == "0a3ef9"
    assert b" -RA BR OVC005\n" in frames[0]["octets"] and b" SLP135\n" in frames[1]["octets"]
    cf = sw.CHANNEL - delta
    step = int(np.float32(np.float32(cf) - np.float32(sw.CHANNEL)) / np.float32(sw.FS) * np.float32(256.0) * np.float32(65536.0))
    nco_hz = step / 2 ** 24 * sw.FS                                  # what the NCO really shifts by (truncated 24-bit step of fp32-rounded frequencies)
    ppm = 10500 * -0.005778 / (2 * np.pi * sw.CHANNEL) * 1e6 + (delta + nco_hz) / sw.CHANNEL * 1e6
    for f in frames:
        assert crc16(f["octets"]) == 0xF0B8
        assert (f["synd_weight"], f["datalen_octets"], f["num_fec_corrections"]) == (0, 504, 0)
        assert abs(f["frame_pwr_dbfs"] - (-9.841)) < 0.01, f["frame_pwr_dbfs"]
        assert abs(f["ppm_error"] - ppm) < 0.005, (f["ppm_error"], ppm)
    return step


@pytest.mark.parametrize("delta", [25000, -250000, 100008, -412500])
def test_reference_wav_offset_tuned(oracle_mod, delta):
    """The NCO / mix branch (demod.c:58-72,200-203,312-317,385) against reference-held data: tests/golden/shift_wav.py moves
    the reference's capture off-centre, the receiver has to bring it back."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import shift_wav as sw
    assert sorted(sw.DELTAS) == sorted([25000, -250000, 100008, -412500])
    raw = sw.shifted(delta)
    o = oracle_mod.Oracle(sw.CHANNEL - delta, [sw.CHANNEL], oversample=10)
    o.process(raw)
    fr = o.frames()
    step = _expect_offset_tuned(fr, delta, oracle_mod.crc16_x25)
    assert o.dphi(0) == step & 0xFFFFFFFF
    # a mixer with the wrong sign leaves the burst 2*delta away: the 2-pole filter then takes >= 30 dB off it (on this noiseless
    # model capture the phase-only detector may still lock), so the power figure above is what pins the sign
    m = oracle_mod.Oracle(sw.CHANNEL + delta, [sw.CHANNEL], oversample=10)
    m.process(raw)
    assert all(f["frame_pwr_dbfs"] < -40 for f in m.frames())


# ---- the 2.1 MS/s path (oversample 20) and the u8 path against reference-held data: tests/golden/resample_wav.py ----
HOT_CASES = [(0, 0), (25000, 5), (-250000, 10), (100008, 15)]     # (offset of the capture in Hz, index of the tuned channel among 16)


def hot_plan(delta, pos, n=16, spacing=50000):
    """16 channels 50 kHz apart with the capture's channel at index `pos`; receiver centre `delta` below that channel"""
    import resample_wav as rw
    return rw.CHANNEL - delta, [rw.CHANNEL + (k - pos) * spacing for k in range(n)]


def _expect_resampled(frames, delta, fs, crc16, pwr_db=-9.841, tol_db=0.01, tol_ppm=0.015):
    """The reference-held answer for a capture derived from the reference's WAV (resample_wav.py): the CI's two frames, S:0 L:504
    F:0, the on-centre burst power (+ the u8 scale where it applies) and the carrier offset of SURVEY 4 (-0.0705 ppm) plus the
    error of the fp32-rounded, truncated NCO step of demod.c:385.  tol_ppm = 0.015 pins the step to 2 Hz at 137 MHz."""
    import resample_wav as rw
    assert [len(f["octets"]) for f in frames] == [314, 186]
    assert frames[0]["octets"][:12].hex() == "b2107684948a341f22544146" and frames[0]["octets"][-3:].hex() == "0a44bf"
    assert frames[1]["octets"][:12].hex() == "b2107684948a341f344d4554" and frames[1]["octets"][-3:].hex() == "0a3ef9"
    assert b" -RA BR OVC005\n" in frames[0]["octets"] and b" SLP135\n" in frames[1]["octets"]
    cf = rw.CHANNEL - delta
    step = int(np.float32(np.float32(cf) - np.float32(rw.CHANNEL)) / np.float32(fs) * np.float32(256.0) * np.float32(65536.0))
    nco_hz = step / 2 ** 24 * fs
    ppm = 10500 * -0.005778 / (2 * np.pi * rw.CHANNEL) * 1e6 + (delta + nco_hz) / rw.CHANNEL * 1e6
    for f in frames:
        assert crc16(f["octets"]) == 0xF0B8
        assert (f["synd_weight"], f["datalen_octets"], f["num_fec_corrections"]) == (0, 504, 0)
        assert abs(f["frame_pwr_dbfs"] - pwr_db) < tol_db, f["frame_pwr_dbfs"]
        assert abs(f["ppm_error"] - ppm) < tol_ppm, (f["ppm_error"], ppm)
    return step


def _golden_path():
    import sys, os
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if p not in sys.path:
        sys.path.insert(0, p)


@pytest.mark.parametrize("delta,pos", HOT_CASES)
def test_reference_wav_at_2100kHz(oracle_mod, delta, pos):
    """oversample 20 (input_lpf_init(2.1 MS/s), demod.c:367-370; the NCO at that rate) on the reference's capture interpolated to
    2.1 MS/s, alone and as one of 16 channels (the others lie 50 kHz apart: on this noiseless capture some of them decode the
    leaked burst too - whatever they do, the tuned channel must give the reference's answer)."""
    _golden_path()
    import resample_wav as rw
    raw = rw.upsampled2x(delta)
    o = oracle_mod.Oracle(rw.CHANNEL - delta, [rw.CHANNEL], oversample=20)
    o.process(raw)
    alone = o.frames()
    step = _expect_resampled(alone, delta, rw.FS2, oracle_mod.crc16_x25)
    assert o.dphi(0) == step & 0xFFFFFFFF
    c = o.counters(0)
    assert c["demod.sync.good"] == 1 and c["decoder.blocks.processed"] == 3 and c["decoder.blocks.fec_ok"] == 3 and c["decoder.msg.good"] == 2
    cf, freqs = hot_plan(delta, pos)
    o16 = oracle_mod.Oracle(cf, freqs, oversample=20)
    o16.process(raw)
    mine = [f for f in o16.frames() if f["chan"] == pos]
    _expect_resampled(mine, delta, rw.FS2, oracle_mod.crc16_x25)
    # same channel, same input: what it decodes does not depend on who else is configured
    assert [(f["octets"], f["sync_sample"], f["ppm_error"]) for f in mine] == [(f["octets"], f["sync_sample"], f["ppm_error"]) for f in alone]


def test_reference_wav_as_u8(oracle_mod):
    """process_buf_uchar()'s conversion (demod.c:339-354) on the reference's capture re-quantised to u8: same frames, the burst
    0.034 dB louder (the u8 table's scale), at 1.05 MS/s on the centre and at 2.1 MS/s off it."""
    _golden_path()
    import resample_wav as rw
    o = oracle_mod.Oracle(rw.CHANNEL, [rw.CHANNEL], oversample=10, sample_fmt=oracle_mod.FMT_U8)
    o.process(rw.as_u8())
    _expect_resampled(o.frames(), 0, rw.FS, oracle_mod.crc16_x25, pwr_db=-9.841 + rw.U8_GAIN_DB)
    d = 25000
    o = oracle_mod.Oracle(rw.CHANNEL - d, [rw.CHANNEL], oversample=20, sample_fmt=oracle_mod.FMT_U8)
    o.process(rw.as_u8(rw.upsampled2x(d)))
    _expect_resampled(o.frames(), d, rw.FS2, oracle_mod.crc16_x25, pwr_db=-9.841 + rw.U8_GAIN_DB)
