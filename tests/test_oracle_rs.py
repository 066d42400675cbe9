"""The oracle's RS(255,249) decoder against the reference's own libfec, compiled unmodified
from /root/reference/src/libfec into oracle/_ref/libfec_ref.so (oracle/Makefile)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libfec_ref.so")


@pytest.fixture(scope="module")
def libfec():
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/libfec_ref.so not built (reference tree absent)")
    L = C.CDLL(REF)
    L.init_rs_char.restype = C.c_void_p
    L.init_rs_char.argtypes = [C.c_int] * 6
    L.decode_rs_char.restype = C.c_int
    L.decode_rs_char.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int]
    rs = L.init_rs_char(8, 0x187, 120, 1, 6, 0)      # rs.c:28
    assert rs
    return L, rs


def ref_verify(L, rs, block, fec_octets):
    """rs_verify() of rs.c:32-49 on top of the reference decoder."""
    d = (C.c_uint8 * 255)(*block)
    if fec_octets == 0:
        return 0, bytes(d)
    n_era = 6 - fec_octets
    if n_era > 0:
        era = (C.c_int * 6)(*[249 + fec_octets + i for i in range(n_era)] + [0] * (6 - n_era))
        r = L.decode_rs_char(rs, d, era, n_era)
    else:
        r = L.decode_rs_char(rs, d, None, 0)
    return r, bytes(d)


def test_encoder_makes_codewords(oracle_mod, libfec):
    L, rs = libfec
    rng = np.random.default_rng(1)
    for _ in range(50):
        data = rng.integers(0, 256, 249, dtype=np.uint8)
        par = oracle_mod.rs_encode(data.tolist())
        r, out = ref_verify(L, rs, list(data) + list(par), 6)
        assert r == 0 and out == bytes(data) + par


@pytest.mark.parametrize("fec_octets", [6, 4, 2])
def test_decoder_matches_libfec(oracle_mod, libfec, fec_octets):
    L, rs = libfec
    rng = np.random.default_rng(100 + fec_octets)
    n_checked = 0
    for trial in range(1500):
        data = rng.integers(0, 256, 249, dtype=np.uint8)
        if trial % 7 == 0:
            data[int(rng.integers(1, 249)):] = 0            # short last block: trailing zeros
        par = list(oracle_mod.rs_encode(data.tolist()))
        block = list(data) + par[:fec_octets] + [0] * (6 - fec_octets)   # missing parity is zero-filled (decode.c:279-280)
        nerr = int(rng.integers(0, 6))                       # up to 5: beyond any capacity too
        for p in rng.choice(249 + fec_octets, size=nerr, replace=False):
            block[p] ^= int(rng.integers(1, 256))
        r_ref, out_ref = ref_verify(L, rs, block, fec_octets)
        r_or, out_or = oracle_mod.rs_decode(block, fec_octets)
        assert r_or == r_ref, f"return {r_or} != libfec {r_ref} (errors={nerr})"
        assert out_or == out_ref
        n_checked += 1
    assert n_checked == 1500


def test_random_garbage_blocks(oracle_mod, libfec):
    L, rs = libfec
    rng = np.random.default_rng(7)
    for _ in range(800):
        block = rng.integers(0, 256, 255, dtype=np.uint8).tolist()
        fec = int(rng.choice([6, 4, 2, 0]))
        if fec < 6:
            for i in range(249 + fec, 255):
                block[i] = 0
        r_ref, out_ref = ref_verify(L, rs, block, fec)
        r_or, out_or = oracle_mod.rs_decode(block, fec)
        assert (r_or, out_or) == (r_ref, out_ref)


def test_device_rs_stage_matches_libfec(oracle_mod, libfec):
    """The RS stage of the burst-decoder kernel (vdl2_core.h:rs_decode_row, built for the CPU by tests/hostsim)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
    import pyhostsim
    H = C.CDLL(pyhostsim.build())
    H.hostsim_rs_decode.restype = C.c_int
    H.hostsim_rs_decode.argtypes = [C.c_void_p, C.c_int]
    L, rs = libfec
    rng = np.random.default_rng(77)
    n_short = 0
    for trial in range(4000):
        fec = int(rng.choice([6, 6, 4, 2, 0]))
        if trial % 3 == 0:
            block = rng.integers(0, 256, 255, dtype=np.uint8).tolist()       # garbage
        else:
            data = rng.integers(0, 256, 249, dtype=np.uint8)
            block = list(data) + list(oracle_mod.rs_encode(data.tolist()))
            for p in rng.choice(249 + max(fec, 1), size=int(rng.integers(0, 6)), replace=False):
                block[p] ^= int(rng.integers(1, 256))
        for i in range(249 + fec, 255):
            block[i] = 0
        r_ref, out_ref = ref_verify(L, rs, block, fec)
        d = (C.c_uint8 * 255)(*block)
        r_dev = H.hostsim_rs_decode(d, fec)
        # what the stage owes the burst decoder: libfec's return value and the 249 data octets.  A short block (fec < 6) without
        # errors is recognised early (vdl2_core.h: the Berlekamp-Massey discrepancies of an erasures-only block are zero) and its
        # erased parity octets - which nothing reads - are then not filled in; every other block must come out as libfec leaves it
        assert r_dev == r_ref and bytes(d)[:249] == bytes(out_ref)[:249], f"trial {trial} fec {fec}"
        if not (fec < 6 and r_ref == 6 - fec):
            assert bytes(d) == bytes(out_ref), f"trial {trial} fec {fec}"
        else:
            n_short += 1
    assert n_short > 100
