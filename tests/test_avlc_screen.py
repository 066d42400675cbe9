"""the AVLC front door (src/avlc.c:163-236 - minimum length, FCS residue, link addresses, direction
counters) restated in the oracle, pinned on the reference's own test vector and on known answers, and the device
implementation (finish_frame in vdl2_core.h, compiled for the CPU by tests/hostsim) against it."""
import numpy as np
import pytest

import cases
import pyhostsim


def test_crc16_known_answers(oracle_mod):
    # CRC-16/X.25 of "123456789" is 0x906E after the final inversion; crc16_ccitt() has no final inversion (crc.c:59-63)
    assert oracle_mod.crc16_x25(b"123456789") ^ 0xFFFF == 0x906E
    # a frame followed by its (inverted, LSB-first) FCS leaves the residue GOOD_FCS = 0xF0B8 (avlc.c:40)
    body = bytes(range(40))
    fcs = oracle_mod.crc16_x25(body) ^ 0xFFFF
    assert oracle_mod.crc16_x25(body + bytes([fcs & 0xFF, fcs >> 8])) == 0xF0B8


def test_reference_vector_frames_pass_the_front_door(oracle_mod, golden_wav):
    """Both frames of test/vdl2_model_16b_1050kHz.wav are valid AVLC: FCS good, HDLC address extension bits in place."""
    cf = 136975000
    o = oracle_mod.Oracle(cf, [cf], oversample=10)
    o.process(golden_wav)
    fr = o.frames()
    assert len(fr) == 2
    for f in fr:
        st, dst, src, d = oracle_mod.avlc_screen(f["octets"])
        oct_ = f["octets"]
        assert st == 0
        # ISO 3309 address extension bit: set only in the last octet of the address field; bit 0 of octets 0 and 4 carry
        # the A/G and C/R flags (parse_dlc_addr() drops them)
        assert [b & 1 for b in oct_[1:4]] == [0, 0, 0] and [b & 1 for b in oct_[5:8]] == [0, 0, 1]
        assert (dst >> 24) & 7 in (1, 4, 5, 7) and (src >> 24) & 7 in (1, 4, 5)
        assert d in (1, 2, 3, 4, 5, 6)
        # bit-reversal property of parse_dlc_addr(): the 27 address+type+status bits are the wire bits in transmission order
        wire = (oct_[0] >> 1) | (oct_[1] >> 1) << 7 | (oct_[2] >> 1) << 14 | (oct_[3] >> 1) << 21
        assert dst == int(format(wire, "028b")[::-1], 2)
    assert oracle_mod.avlc_counters(fr, 1)[0][:4] == [2, 0, 2, 0]


def test_front_door_classification(oracle_mod):
    def frame(dst_type, src_type, corrupt=False, short=False):
        def addr(a, t, last):
            v = a | t << 24                                   # status bit 0
            wire = int(format(v, "028b")[::-1], 2)            # transmitted LSB first = reversed
            b = [(wire >> (7 * i) & 0x7F) << 1 for i in range(4)]
            b[3] |= 1 if last else 0
            return bytes(b)
        body = addr(0x123456, dst_type, False) + addr(0xABCDEF, src_type, True) + b"\x03" + b"hello world"
        if short:
            body = body[:7]
        fcs = oracle_mod.crc16_x25(body) ^ 0xFFFF
        out = bytearray(body + bytes([fcs & 0xFF, fcs >> 8]))
        if corrupt:
            out[9] ^= 0x10
        return bytes(out)
    want = {(4, 1): 1, (5, 1): 1, (1, 1): 2, (7, 1): 3, (1, 4): 4, (1, 5): 4, (4, 5): 5, (5, 4): 5, (7, 4): 6, (2, 1): 0, (1, 2): 0, (4, 7): 0}
    for (dt, stp), d in want.items():
        st, dst, src, got = oracle_mod.avlc_screen(frame(dt, stp))
        assert (st, got) == (0, d), (dt, stp)
        assert dst == (0x123456 | dt << 24) and src == (0xABCDEF | stp << 24)
    assert oracle_mod.avlc_screen(frame(4, 1, corrupt=True))[0] == 2
    assert oracle_mod.avlc_screen(frame(4, 1, short=True))[0] == 1
    assert oracle_mod.avlc_screen(b"")[0] == 1


@pytest.mark.parametrize("name", ["config2_1s", "config5_0p4s", "dirty25k_1s"])
def test_device_front_door_matches_oracle(oracle_mod, name):
    cfg, iq, _, _ = cases.load(name)
    nch = len(cfg.freqs)
    o = oracle_mod.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
    D = iq.size // 2 // cfg.oversample
    tr = o.trace_all(D + 4)
    o.process(iq.view(np.uint8), block_bytes=1 << 24, nthreads=4)
    D = o.decimated_count(0)
    hs = pyhostsim.HostSim(list(cfg.freqs), cfg.rx_max_ppm, cap_log2=int(np.ceil(np.log2(D + 70000))))
    hs.feed(tr[:, :D, :])
    fh = hs.frames()
    assert len(fh) == len(o.frames()) > 0
    for f in fh:
        st, dst, src, _ = oracle_mod.avlc_screen(f["octets"])
        assert (f["avlc_status"], f["dst_addr"], f["src_addr"]) == (st, dst, src)
    assert [hs.avlc_counters(c) for c in range(nch)] == oracle_mod.avlc_counters(fh, nch)
    assert sum(f["avlc_status"] == 0 for f in fh) > 0
    hs.close()


def test_device_fcs_and_addresses_fuzz(oracle_mod):
    """finish_frame() (four octets per step from slicing tables, tail octets singly) against the oracle's bitwise FCS for every
    length 0..70 and random lengths up to the largest frame, valid and corrupted, with random address octets."""
    import ctypes as C
    from dumpvdl2_amd import synth
    L = C.CDLL(pyhostsim.build())
    L.hostsim_finish_frame.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_ulonglong)]
    rng = np.random.default_rng(11)
    acnt = (C.c_ulonglong * 10)()
    want_cnt = [0] * 10
    lens = list(range(0, 71)) + [int(x) for x in rng.integers(71, 2100, size=150)] + [2047, 2048, 2099]
    for n in lens:
        for variant in range(3):
            body = rng.integers(0, 256, size=max(n, 0), dtype=np.uint8).tobytes()
            if variant == 1 and n >= 3:
                body = synth.make_avlc_frame(body[:n - 2])                  # valid FCS, same total length
            elif variant == 2 and n >= 12:
                fr = bytearray(synth.make_avlc_frame(body[:n - 2])); fr[int(rng.integers(0, n))] ^= 1 << int(rng.integers(0, 8)); body = bytes(fr)
            dst = C.c_uint32(); src = C.c_uint32()
            st = L.hostsim_finish_frame(body, len(body), C.byref(dst), C.byref(src), acnt)
            ost, odst, osrc, odir = oracle_mod.avlc_screen(body)
            assert (st, dst.value, src.value) == (ost, odst, osrc), (n, variant)
            want_cnt[0] += 1
            if ost == 1: want_cnt[1] += 1
            elif ost == 2: want_cnt[3] += 1
            else:
                want_cnt[2] += 1
                if odir: want_cnt[3 + odir] += 1
    assert list(acnt) == want_cnt
    assert want_cnt[2] > 100 and want_cnt[3] > 100 and want_cnt[1] > 20
