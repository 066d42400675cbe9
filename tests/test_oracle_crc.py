"""The AVLC frame check sequence against the reference's own crc.c, compiled unmodified from /root/reference/src/crc.c into
oracle/_ref/libcrc_ref.so (oracle/Makefile): the oracle's CRC, the oracle's avlc_parse() front door, the table the device
uses (tables.h) and the device's sliced four-octets-per-step evaluation (vdl2_core.h:finish_frame, host build)."""
import ctypes as C
import os

import numpy as np
import pytest

import pyhostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libcrc_ref.so")


@pytest.fixture(scope="module")
def refcrc():
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/libcrc_ref.so not built (reference tree absent)")
    L = C.CDLL(REF)
    L.crc16_ccitt.restype = C.c_uint16
    L.crc16_ccitt.argtypes = [C.c_char_p, C.c_uint32, C.c_uint16]
    return L


def test_oracle_crc_is_the_reference_crc(refcrc, oracle_mod):
    rng = np.random.default_rng(77)
    assert refcrc.crc16_ccitt(b"123456789", 9, 0xFFFF) ^ 0xFFFF == 0x906E          # CRC-16/X.25 check value
    for n in list(range(0, 40)) + [249, 255, 1000, 2048]:
        for _ in range(8):
            b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            init = int(rng.integers(0, 65536)) if n % 3 else 0xFFFF
            assert oracle_mod.crc16_x25(b, init) == refcrc.crc16_ccitt(b, n, init)


def test_device_fcs_table_and_slicing_match_the_reference(refcrc, oracle_mod):
    H = C.CDLL(pyhostsim.build())
    pr = (C.c_float * 16)(); gray = (C.c_uint8 * 8)(); crc = (C.c_uint16 * 256)(); prbs = (C.c_uint8 * 64)(); gf = (C.c_uint8 * 8)()
    H.hostsim_misc_tables(pr, gray, crc, prbs, gf)
    # one table step from a zero register is the table entry itself (crc.c:59-63)
    assert [refcrc.crc16_ccitt(bytes([b]), 1, 0) for b in range(256)] == list(crc)
    # finish_frame(): FCS verdict of frames with a good / damaged FCS, every length class of the 4-octet slicing
    H.hostsim_finish_frame.restype = C.c_int
    H.hostsim_finish_frame.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_ulonglong)]
    rng = np.random.default_rng(78)
    from dumpvdl2_amd import synth
    for n in list(range(9, 30)) + [100, 101, 102, 103, 1023, 2047]:
        body = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        good = synth.make_avlc_frame(body)
        assert refcrc.crc16_ccitt(good, len(good), 0xFFFF) == 0xF0B8                   # avlc.c:40,177
        bad = bytearray(good); bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        for fr, want_ok in ((good, True), (bytes(bad), False)):
            dst = C.c_uint32(); src = C.c_uint32(); acnt = (C.c_ulonglong * 10)()
            st = H.hostsim_finish_frame(fr, len(fr), C.byref(dst), C.byref(src), acnt)
            ref_ok = refcrc.crc16_ccitt(fr, len(fr), 0xFFFF) == 0xF0B8
            assert ref_ok == want_ok
            assert (st == 0) == ref_ok if len(fr) >= 11 else st == 1
            assert (oracle_mod.avlc_screen(fr)[0] == 0) == (ref_ok and len(fr) >= 11)
