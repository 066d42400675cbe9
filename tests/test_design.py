"""Host-side constants of the channeliser (dumpvdl2_amd/csrc/design.h, compiled for the CPU by tests/hostsim): the filter
design and the NCO against the oracle's restatement of chebyshev.c / demod.c, and the block form K1 evaluates against the
direct-form recurrence it replaces."""
import ctypes as C

import numpy as np
import pytest

import pyhostsim

K_FIX = 128
K_MAXOS = 32


class BlockForm(C.Structure):
    _fields_ = [("os", C.c_int), ("run", C.c_int), ("g0", C.c_float * K_MAXOS), ("g1", C.c_float * K_MAXOS),
                ("P", C.c_float * 4), ("c0", C.c_float), ("c1", C.c_float), ("c2", C.c_float),
                ("cP", (C.c_float * 2) * K_FIX), ("Ppow", (C.c_float * 4) * (K_FIX + 1)), ("Q", (C.c_float * 4) * 6),
                ("Qpow", (C.c_float * 4) * 64), ("basis", C.c_float * 4)]


@pytest.fixture(scope="module")
def hs():
    L = C.CDLL(pyhostsim.build())
    L.hostsim_nco_step.restype = C.c_uint32
    L.hostsim_nco_step.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.hostsim_design_lpf.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.hostsim_block_form.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(BlockForm)]
    assert L.hostsim_sizeof_blockform() == C.sizeof(BlockForm)
    return L


def lpf(hs, os_):
    A = (C.c_float * 3)(); B = (C.c_float * 3)()
    hs.hostsim_design_lpf(C.c_float(8000.0 / (105000.0 * os_)), C.c_float(0.5), A, B)     # input_lpf_init(), demod.c:367-370
    return A, B


@pytest.mark.parametrize("os_", [10, 13, 20])
def test_filter_design_and_nco_step_match_the_oracle(hs, oracle_mod, os_):
    cf = 136975000
    freqs = [cf - 400000, cf - 12500, cf, cf + 25000, cf + 987654]
    o = oracle_mod.Oracle(cf, freqs, oversample=os_)
    A, B = lpf(hs, os_)
    Ao, Bo = o.lpf()
    assert bytes(A) == Ao.tobytes() and bytes(B) == Bo.tobytes()
    for c, f in enumerate(freqs):
        assert hs.hostsim_nco_step(cf, f, 105000 * os_) & 0xFFFFFF == o.dphi(c) & 0xFFFFFF


def test_nco_lut_is_sincosf_lut(hs):
    """entry i = {sin, cos, (sin[i+1]-sin[i]) 2^-16, (cos[i+1]-cos[i]) 2^-16} of the reference's 256-entry table (demod.c:372-377)"""
    lut = np.zeros((256, 4), dtype=np.float32)
    hs.hostsim_nco_lut(lut.ctypes.data_as(C.c_void_p))
    i = np.arange(257, dtype=np.float64)
    ang = (2.0 * np.pi * i / 256.0).astype(np.float32)      # "2.0f * M_PI * (float)i / 256.0f": evaluated in double, narrowed for sincosf()
    s = np.sin(ang.astype(np.float64)).astype(np.float32); c = np.cos(ang.astype(np.float64)).astype(np.float32)
    s[256] = s[0]; c[256] = c[0]
    assert np.abs(lut[:, 0] - s[:256]).max() <= 1.2e-7 and np.abs(lut[:, 1] - c[:256]).max() <= 1.2e-7      # sincosf vs libm: 1 ulp
    assert np.allclose(lut[:, 2] * 65536.0, lut[np.r_[1:256, 0], 0] - lut[:, 0], atol=1e-7)
    assert np.allclose(lut[:, 3] * 65536.0, lut[np.r_[1:256, 0], 1] - lut[:, 1], atol=1e-7)


@pytest.mark.parametrize("os_,run", [(20, 2), (10, 2), (13, 2), (7, 2)])
def test_block_form_reproduces_the_direct_form(hs, os_, run):
    """y[n] = A0 x[n] + A1 x[n-1] + A2 x[n-2] + B1 y[n-1] + B2 y[n-2] (chebyshev.c / demod.c:58-79), decimated by os, against
    the block recurrence K1 runs: t_k = P t_{k-1} + sum_j (g0[j], g1[j]) x[os k + j], y_k = c0 t0 + c1 t1 + c2 x[last] - with the
    state in the normal form of the recursion matrix (design.h: derive_block_form), where the single-precision constants and the
    single-precision arithmetic both leave ~1e-7 of the signal level instead of the ~1e-4 of the (v[n], v[n-1]) basis."""
    A, B = lpf(hs, os_)
    bf = BlockForm()
    hs.hostsim_block_form(A, B, os_, run, C.byref(bf))
    a = np.array(A[:], dtype=np.float64); b = np.array(B[:], dtype=np.float64)
    rng = np.random.default_rng(os_)
    nblk = 600
    x = rng.standard_normal(nblk * os_)
    y = np.zeros_like(x)
    for n in range(len(x)):
        y[n] = a[0] * x[n] + (a[1] * x[n - 1] if n >= 1 else 0) + (a[2] * x[n - 2] if n >= 2 else 0) \
            + (b[1] * y[n - 1] if n >= 1 else 0) + (b[2] * y[n - 2] if n >= 2 else 0)
    want = y[os_ - 1::os_]
    P = np.array(bf.P[:], dtype=np.float64).reshape(2, 2)
    g0 = np.array(bf.g0[:os_], dtype=np.float64); g1 = np.array(bf.g1[:os_], dtype=np.float64)
    t = np.zeros(2); got = np.zeros(nblk)
    for k in range(nblk):
        blk = x[k * os_:(k + 1) * os_]
        t = P @ t + np.array([g0 @ blk, g1 @ blk])
        got[k] = bf.c0 * t[0] + bf.c1 * t[1] + bf.c2 * blk[-1]
    # fp32-rounded P, taps and c0..c2, arithmetic in double: in the normal form the rounded constants cost ~1e-7 of the signal
    # level (in the basis of the recursion itself, where P = [[15.1, -14.2], [14.7, -13.7]] at oversample 20, they cost ~1e-5)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    # ... and the same recurrence carried in single precision (every product and sum rounded) stays as close
    t32 = np.zeros(2, dtype=np.float32); got32 = np.zeros(nblk)
    P32 = P.astype(np.float32); g32 = np.stack([g0, g1]).astype(np.float32); c32 = np.array([bf.c0, bf.c1, bf.c2], dtype=np.float32)
    for k in range(nblk):
        blk = x[k * os_:(k + 1) * os_].astype(np.float32)
        t32 = (P32 @ t32 + g32 @ blk).astype(np.float32)
        got32[k] = np.float32(c32[0] * t32[0] + c32[1] * t32[1] + c32[2] * blk[-1])
    assert np.abs(got32 - want).max() <= 5e-6 * np.abs(want).max()
    # the tables the wave scan and the fix-ups use are powers of the same P
    # (derived in double from M = [[B1, B2], [1, 0]], P = T M^os T^-1, then rounded once - so compare with the double powers)
    r = np.sqrt(-b[2]); ct = b[1] / (2 * r); st = np.sqrt(1 - ct * ct)
    Ti = np.array([[r * ct, r * st], [1.0, 0.0]]); T = np.linalg.inv(Ti)          # columns of T^-1: Re and Im of M's eigenvector (lambda, 1)
    assert np.abs(np.array(bf.basis[:], dtype=np.float64).reshape(2, 2) - T).max() <= 1e-7 * np.abs(T).max()
    Pd = T @ np.linalg.matrix_power(np.array([[b[1], b[2]], [1.0, 0.0]]), os_) @ Ti
    assert np.abs(P - Pd).max() <= 6e-8 * np.abs(Pd).max()
    # a rotation scaled by |lambda|^os: nothing cancels in P t
    assert np.abs(Pd @ Pd.T - (r ** (2 * os_)) * np.eye(2)).max() <= 1e-9 * r ** (2 * os_)
    Pp = np.array([list(r_) for r_ in bf.Ppow], dtype=np.float64).reshape(-1, 2, 2)
    c01 = np.array([a[0] + a[2] / b[2], a[1] - a[2] * b[1] / b[2]]) @ Ti
    assert abs(bf.c0 - c01[0]) <= 1e-7 * abs(c01[0]) and abs(bf.c1 - c01[1]) <= 1e-7 * abs(c01[1]) and abs(bf.c2 + a[2] / b[2]) <= 1e-7 * abs(a[2] / b[2])
    acc = np.eye(2)
    for i in range(K_FIX + 1):
        assert np.abs(Pp[i] - acc).max() <= 1e-7 * max(1e-30, np.abs(acc).max()) + 1e-37
        if i < K_FIX:
            nxt = acc @ Pd
            cP = np.array(list(bf.cP[i]), dtype=np.float64)
            ref = c01 @ nxt
            assert np.abs(cP - ref).max() <= 1e-7 * max(1e-30, np.abs(nxt).max() * np.abs(c01).max()) + 1e-37
        acc = acc @ Pd
    P = Pd
    Q1 = np.linalg.matrix_power(P, run)
    for d in range(6):
        Qd = np.linalg.matrix_power(Q1, 2 ** d)
        assert np.abs(np.array(list(bf.Q[d])).reshape(2, 2) - Qd).max() <= 1e-7 * np.abs(Qd).max() + 1e-37
    for l in (0, 1, 5, 63):
        Ql = np.linalg.matrix_power(Q1, l + 1)
        assert np.abs(np.array(list(bf.Qpow[l])).reshape(2, 2) - Ql).max() <= 1e-7 * np.abs(Ql).max() + 1e-37
    # a start state has decayed below fp32 resolution after K_FIX blocks: what makes the one-step look-back exact
    assert np.abs(np.linalg.matrix_power(P, K_FIX)).max() < 1e-12


def test_sync_screening_never_hides_a_sub_threshold_metric(hs):
    """K3 redoes the exact got_sync() arithmetic only where the screening value is under 5.5 (threshold 4).  That is safe iff
    the screening value is within ~1 of the exact one wherever the exact one is small; it is within 0.2 everywhere here:
    random windows, clean preambles with carrier offsets up to +-3 rad/symbol (large unwrapped excursions), preambles with
    noise tuned to land around the threshold, and windows sitting right at the +-pi unwrap decision."""
    hs.hostsim_metric_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    q = np.array([0, 3, -3, 1, 1, 2, 0, 4, -3, 4, -2, 3, 1, -2, -3, 0], dtype=np.float64) * np.pi / 4     # demod.c:107-124

    def wrap(x):
        return (x + np.pi) % (2 * np.pi) - np.pi

    sets = [rng.uniform(-np.pi, np.pi, size=(200000, 16))]
    for sigma in (0.0, 0.05, 0.3, 0.5, 0.6, 0.8):
        n = 60000
        slope = rng.uniform(-3.0, 3.0, size=(n, 1)); off = rng.uniform(-np.pi, np.pi, size=(n, 1))
        sets.append(wrap(q[None, :] + off + slope * np.arange(16)[None, :] + sigma * rng.standard_normal((n, 16))))
    edge = wrap(q[None, :] + np.pi * np.arange(16)[None, :] * rng.choice([-1.0, 1.0], size=(50000, 1)) + 1e-3 * rng.standard_normal((50000, 16)))
    sets.append(edge)
    ph = np.ascontiguousarray(np.concatenate(sets).astype(np.float32))
    n = ph.shape[0]
    exact = np.zeros(n, dtype=np.float32); slope = np.zeros(n, dtype=np.float32); screen = np.zeros(n, dtype=np.float32)
    hs.hostsim_metric_pairs(ph.ctypes.data, n, exact.ctypes.data, slope.ctypes.data, screen.ctypes.data)
    assert np.isfinite(exact).all() and np.isfinite(screen).all()
    assert (exact < 4).sum() > 50000 and ((exact > 3) & (exact < 5)).sum() > 2000       # the interesting region is populated
    # a screening value of exactly 0 is the unwrap guard speaking: one of the window's tap differences lies within kScreenGuard of
    # +-pi, where the single-precision phases the kernel screens with could unwrap the other way - such a window goes to the
    # exact tier whatever it looks like (the "edge" set above is made of them)
    guarded = screen == 0.0
    assert guarded[-50000:].mean() > 0.001 and guarded[:200000].mean() < 1e-3
    err = np.abs(screen.astype(np.float64) - exact.astype(np.float64))[~guarded]
    assert err.max() < 0.2, err.max()
    assert not np.any((exact < 4.0) & (screen >= 5.5))
    # the early exit after 12 taps: that partial value is a lower bound (up to rounding) of the 16-tap screening value, and a
    # window that can matter (screening value under 5.5) is never dropped by the early test (under 5.8)
    hs.hostsim_metric_early.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    early = np.zeros(n, dtype=np.float32)
    hs.hostsim_metric_early(ph.ctypes.data, n, early.ctypes.data)
    assert hs.hostsim_screen_early_taps() == 12
    assert (early.astype(np.float64) - screen.astype(np.float64))[~guarded].max() < 0.2
    assert not np.any((screen < 5.5) & ~guarded & (early >= 5.8))
    # a guard that only trips in taps 12..15 is never seen when the window has already stopped after 12: harmless, the first 12
    # unwrapped errors do not depend on later decisions and already put the exact value over the threshold
    assert not np.any((exact < 4.0) & (early >= 5.8))
    assert (early[:200000] >= 5.8).mean() > 0.99                      # random windows: almost all stop early


def test_header_code_tables_are_the_reference_tables(hs, oracle_mod):
    """The (25,20) header code: parity-check rows, syndrome -> error-pattern table and syndrome weights, read from the
    reference's source where it lies (src/decode.c:55-100; skipped where /root/reference is not mounted), against the tables
    the device builds from the parity-check matrix (tables.h) and against the oracle's decoder."""
    import os, re
    path = "/root/reference/src/decode.c"
    if not os.path.exists(path):
        pytest.skip("reference tree not mounted here")
    src = open(path).read()

    def table(name):
        body = re.search(name + r"\s*\[[^\]]*\]\s*=\s*\{([^}]*)\}", src).group(1)
        return [int(t, 0) for t in re.findall(r"0b[01]+|\b\d+\b", body)]

    H_ref, fix_ref, w_ref = table("H"), table("syndtable"), table("synd_weight")
    assert len(H_ref) == 5 and len(fix_ref) == 32 and len(w_ref) == 32
    H = (C.c_uint32 * 5)(); fix = (C.c_uint32 * 32)(); w = (C.c_uint32 * 32)()
    hs.hostsim_header_tables(H, fix, w)
    assert list(H) == H_ref and list(fix) == fix_ref and list(w) == w_ref
    # the oracle corrects every table pattern (an error on the all-zero codeword) back to zero and reports its syndrome
    L = oracle_mod.lib()
    L.vdl2o_header_decode.restype = C.c_uint32
    L.vdl2o_header_decode.argtypes = [C.POINTER(C.c_uint32)]
    for s_, e in enumerate(fix_ref):
        word = C.c_uint32(e)
        assert L.vdl2o_header_decode(C.byref(word)) == s_ and word.value == 0


def test_other_tables_are_the_reference_constants(hs):
    """Preamble phases (demod.c:107-124), Gray map (demod.c:223), FCS table (crc.c:23-57), descrambler seed (decode.c:50) and RS
    field (rs.c:28), read from the reference's source where it lies, against what tables.h derives."""
    import os, re
    root = "/root/reference/src"
    if not os.path.exists(root):
        pytest.skip("reference tree not mounted here")
    pr = (C.c_float * 16)(); gray = (C.c_uint8 * 8)(); crc = (C.c_uint16 * 256)(); prbs = (C.c_uint8 * 64)(); gfe = (C.c_uint8 * 8)()
    hs.hostsim_misc_tables(pr, gray, crc, prbs, gfe)
    demod = open(os.path.join(root, "demod.c")).read()
    body = re.search(r"pr_phase\[PREAMBLE_SYMS\]\s*=\s*\{([^}]*)\}", demod).group(1)
    quarters = [int(m) for m in re.findall(r"(-?\d+)\s*\*\s*M_PI\s*/\s*4", body)]
    assert len(quarters) == 16
    assert [np.float32(q * np.pi / 4) for q in quarters] == [np.float32(x) for x in pr]
    g = [int(x) for x in re.search(r"graycode\[ARITY\]\s*=\s*\{([^}]*)\}", demod).group(1).split(",")]
    assert g == list(gray)
    crcsrc = open(os.path.join(root, "crc.c")).read()
    tab = [int(x, 16) for x in re.findall(r"0x[0-9A-Fa-f]{4}", re.search(r"crctable\[256\]\s*=\s*\{([^}]*)\}", crcsrc, re.S).group(1))]
    assert len(tab) == 256 and tab == list(crc)
    iv = int(re.search(r"#define\s+LFSR_IV\s+(0x[0-9a-fA-F]+)", open(os.path.join(root, "decode.c")).read()).group(1), 16)
    l, want = iv, []
    for _ in range(64):                                   # bitstream_descramble(), bitstream.c:94-107: x^15 + x + 1
        bit = (l ^ (l >> 14)) & 1
        l = (l >> 1) | (bit << 14)
        want.append(bit)
    assert want == list(prbs)
    m = re.search(r"init_rs_char\(\s*8\s*,\s*(0x[0-9a-fA-F]+)\s*,\s*(\d+)\s*,\s*1\s*,", open(os.path.join(root, "rs.c")).read())
    poly, fcr = int(m.group(1), 16), int(m.group(2))
    assert (poly, fcr) == (0x187, 120)
    x, exp = 1, []
    for _ in range(8):
        exp.append(x); x <<= 1
        if x & 0x100: x ^= poly
    assert exp == list(gfe)
