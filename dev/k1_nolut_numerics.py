"""CPU check of the table-free NCO formulation of K1 (the -DVDL2_K1_NOLUT experiment: commit 60803f7; measured on the GPU: correct, not
faster - profiles/r04_k1_table_free_nco_ab.txt) before it went to a GPU.

The reference mixes every input sample with a 256-entry, linearly interpolated sine table (sincosf_lut(), src/demod.c:58-72):
lut(phi) = e^{j theta_i} (1 + f (e^{j delta} - 1)), theta_i the cell start, f the position in the cell, delta = 2 pi / 256.
Against the exact carrier e^{j phi} that is  lut(phi) = e^{j phi} g(f),  g(f) = (1 + f (e^{j delta} - 1)) e^{-j f delta}
= 1 - (1 - cos delta) f (1 - f) + j O(delta^3 / 6 * 0.096): a real, per-sample amplitude dip of at most 7.5e-5 and a phase term
below 2.4e-7 rad.  So the block's tap sums can be taken with per-channel CONSTANT complex taps G[j] = hap[.] e^{j j dphi}
on amplitude-corrected samples x_j (1 + kappa (h_j^2 - 1/4)), h_j = f_j - 1/2, and rotated once per block by the carrier at the
block's first sample - no table look-up per sample.  This script measures, on a seeded capture, how far the decimated stream of
 (cur) today's block form with the interpolated table and
 (new) the table-free form
lie from the oracle's sequential scan, in single precision, and what the formula itself (double precision) leaves out.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle as po  # noqa: E402
import cases  # noqa: E402

F = np.float32


def lut_tables():
    i = np.arange(257, dtype=np.float32)
    ang = (F(2.0) * F(np.pi) * (i % 256) / F(256.0)).astype(np.float32)
    return np.sin(ang.astype(np.float64)).astype(np.float32), np.cos(ang.astype(np.float64)).astype(np.float32)


def block_form(A, B, os_):
    A0, A1, A2 = (float(a) for a in A)
    B1, B2 = float(B[1]), float(B[2])
    hap = np.zeros(os_ + 2)
    hap[1] = 1.0
    for n in range(1, os_ + 1):
        hap[n + 1] = B1 * hap[n] + B2 * (hap[n - 1] if n >= 2 else 0.0)
    H = lambda n: hap[n + 1]
    g0 = np.array([H(os_ - 1 - j) for j in range(os_)])
    g1 = np.array([H(os_ - 2 - j) for j in range(os_)])
    M = np.array([[B1, B2], [1.0, 0.0]])
    P = np.linalg.matrix_power(M, os_)
    c0, c1, c2 = A0 + A2 / B2, A1 - A2 * B1 / B2, -A2 / B2
    return g0, g1, P, (c0, c1, c2)


def run_blocks(acc0, acc1, mlast, P, c, dt):
    """the 2x2 recurrence over the blocks (sequential here; the kernel scans): acc = zero-start tap sums per block"""
    P = P.astype(dt)
    c0, c1, c2 = (dt(v) for v in c)
    t0 = dt(0) + 0j
    t1 = dt(0) + 0j
    y = np.zeros(len(acc0), dtype=np.complex128)
    cdt = np.complex64 if dt is np.float32 else np.complex128
    for k in range(len(acc0)):
        n0 = cdt(cdt(P[0, 0] * t0) + cdt(P[0, 1] * t1)) + cdt(acc0[k])
        n1 = cdt(cdt(P[1, 0] * t0) + cdt(P[1, 1] * t1)) + cdt(acc1[k])
        t0, t1 = cdt(n0), cdt(n1)
        y[k] = cdt(cdt(c0 * t0) + cdt(c1 * t1) + cdt(c2 * cdt(mlast[k])))
    return y


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "config2_1s"
    cfg, iq, bursts, gold = cases.load(name)
    os_ = cfg.oversample
    freqs = list(cfg.freqs)
    o = po.Oracle(cfg.centerfreq, freqs, oversample=os_)
    nin = iq.size // 2
    D = nin // os_
    tr = o.trace_all(D)
    o.process(iq.view(np.uint8), block_bytes=320000)
    A, B = o.lpf()
    g0, g1, P, c = block_form(A, B, os_)
    x = (iq.reshape(-1, 2).astype(np.float32) / F(32768.0))
    xc = (x[:, 0].astype(np.float64) + 1j * x[:, 1].astype(np.float64))[: D * os_]
    sin_l, cos_l = lut_tables()
    delta = 2 * np.pi / 256
    kappa = 1.0 - np.cos(delta)
    print(f"{name}: {len(freqs)} channels, oversample {os_}, {D} decimated samples; kappa = {kappa:.6e}")
    worst = {"cur": 0.0, "new": 0.0, "new_formula": 0.0, "new_with_phase": 0.0}
    for ch in range(len(freqs)):
        dphi = o.dphi(ch)
        ref = tr[ch, :D, 0].astype(np.float64) + 1j * tr[ch, :D, 1].astype(np.float64)
        peak = np.abs(ref).max()
        n = np.arange(D * os_, dtype=np.uint64)
        ph = ((n * np.uint64(dphi)) & np.uint64(0xffffff)).astype(np.uint32)
        idx = (ph >> 16).astype(np.int64)
        fr = (ph & 0xffff).astype(np.float32)
        # --- the reference's mixer in double (what both forms approximate) and in single (what "cur" does, FMAs apart)
        f64 = fr.astype(np.float64) / 65536.0
        s_d = sin_l[idx].astype(np.float64) + (sin_l[idx + 1].astype(np.float64) - sin_l[idx]) * f64
        c_d = cos_l[idx].astype(np.float64) + (cos_l[idx + 1].astype(np.float64) - cos_l[idx]) * f64
        if dphi == 0:
            s_d[:] = 0.0; c_d[:] = 1.0
        m_ref = xc * (c_d + 1j * s_d)
        # cur (single): taps in float, products rounded
        m32 = m_ref.astype(np.complex64)
        mb = m32.reshape(D, os_)
        acc0 = (mb * g0.astype(np.float32)).sum(axis=1, dtype=np.complex64)
        acc1 = (mb * g1.astype(np.float32)).sum(axis=1, dtype=np.complex64)
        y_cur = run_blocks(acc0, acc1, mb[:, -1], P, c, np.float32)
        # --- new: constant complex taps, amplitude-corrected samples, one rotation per block
        j = np.arange(os_, dtype=np.uint64)
        wj = np.exp(2j * np.pi * ((j * np.uint64(dphi)) & np.uint64(0xffffff)).astype(np.float64) / 2**24)
        G0, G1 = g0 * wj, g1 * wj
        h = (f64 - 0.5)
        for variant in ("new_formula", "new", "new_with_phase"):
            dt = np.float64 if variant == "new_formula" else np.float32
            cdt = np.complex128 if dt is np.float64 else np.complex64
            s = (1.0 + kappa * (h * h - 0.25))
            if dphi == 0:
                s[:] = 1.0
            xs = (xc * s)
            if variant == "new_with_phase":
                xs = xs * (1.0 + 1j * (delta**3) * h * (0.25 - h * h) / 3.0)
            xs = xs.astype(cdt).reshape(D, os_)
            S0 = (xs * G0.astype(cdt)).sum(axis=1, dtype=cdt)
            S1 = (xs * G1.astype(cdt)).sum(axis=1, dtype=cdt)
            pb = ph.reshape(D, os_)[:, 0]
            ib = (pb >> 16).astype(np.int64)
            r = (pb & 0xffff).astype(np.float64) * (delta / 65536.0)
            E = (cos_l[ib].astype(np.float64) + 1j * sin_l[ib].astype(np.float64)) * ((1 - r * r / 2 + r**4 / 24) + 1j * (r - r**3 / 6))
            if dphi == 0:
                E[:] = 1.0
            E = E.astype(cdt)
            a0 = (E * S0).astype(cdt); a1 = (E * S1).astype(cdt)
            ml = (E * (wj[-1].astype(cdt) * xs[:, -1]).astype(cdt)).astype(cdt)
            y_new = run_blocks(a0, a1, ml, P, c, dt)
            worst[variant] = max(worst[variant], np.abs(y_new - ref).max() / peak)
        worst["cur"] = max(worst["cur"], np.abs(y_cur - ref).max() / peak)
    print("max |y - oracle| / max |oracle| over the channels:")
    for k, v in worst.items():
        print(f"  {k:16s} {v:.3e}")


if __name__ == "__main__":
    main()
