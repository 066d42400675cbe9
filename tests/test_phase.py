"""The device atan2 (vdl2_core.h:atan2_f64, also what tests/hostsim runs) against libm."""
import ctypes as C
import math

import numpy as np

import pyhostsim


def test_atan2_matches_libm_after_narrowing():
    H = C.CDLL(pyhostsim.build())
    H.hostsim_phase.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    H.hostsim_atan2.restype = C.c_double
    H.hostsim_atan2.argtypes = [C.c_double, C.c_double]
    rng = np.random.default_rng(9)
    n = 4_000_000
    mag = np.exp(rng.uniform(-20, 2, n))
    ang = rng.uniform(-np.pi, np.pi, n)
    xy = np.empty((n, 2), dtype=np.float32)
    xy[:, 0] = mag * np.cos(ang); xy[:, 1] = mag * np.sin(ang)
    # the places where the reduction switches (ratio k/16, axes, diagonals) deserve their own samples
    edge = np.array([[1, 0], [0, 1], [-1, 0], [0, -1], [1, 1], [-1, 1], [1, -1], [-1, -1], [1e-30, 1], [1, 1e-30], [-1e-30, -1],
                     [3, 0.1875], [0.1875, 3], [1, 0.0625], [1, 0.062500004], [-2, 1e-38], [1e-38, -2], [-1, -0.0],
                     [5e-39, 1e-45], [1e-45, -5e-39]], dtype=np.float32)
    xy = np.concatenate([xy, edge, -edge])
    out = np.empty(len(xy), dtype=np.float32)
    H.hostsim_phase(xy.ctypes.data, out.ctypes.data, len(xy))
    ref = np.arctan2(xy[:, 1].astype(np.float64), xy[:, 0].astype(np.float64)).astype(np.float32)
    nz = ~((xy[:, 0] == 0) & (xy[:, 1] == 0))
    diff = out[nz] != ref[nz]
    assert diff.sum() <= 2, f"{diff.sum()} of {nz.sum()} phases differ from libm after narrowing"
    if diff.any():
        assert np.all(np.abs(out[nz][diff].astype(np.float64) - ref[nz][diff]) <= np.spacing(np.abs(ref[nz][diff])))
    for y, x in [(0.3, -2.0), (-7.5, 0.01), (1e-200, 1e-190), (5.0, 5.0), (-0.0, -1.0), (0.0, -1.0)]:
        a, b = H.hostsim_atan2(y, x), math.atan2(y, x)
        assert abs(a - b) <= 4 * np.spacing(abs(b)) and math.copysign(1, a) == math.copysign(1, b)


def test_screening_phase_is_within_its_guard():
    """phase_fast() (the sync kernel's screening tier; it works in turns) against the exact phase: the error of a difference of
    two such phases, plus the roundings on the way to the unwrap decision on both sides, must stay far inside kScreenGuard, the
    margin within which an unwrap decision is handed to the exact tier."""
    H = C.CDLL(pyhostsim.build())
    H.hostsim_phase.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    H.hostsim_phase_fast.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    H.hostsim_screen_guard.restype = C.c_float
    rng = np.random.default_rng(10)
    n = 2_000_000
    mag = np.exp(rng.uniform(-20, 2, n))
    ang = rng.uniform(-np.pi, np.pi, n)
    xy = np.empty((n, 2), dtype=np.float32)
    xy[:, 0] = mag * np.cos(ang); xy[:, 1] = mag * np.sin(ang)
    edge = np.array([[1, 0], [0, 1], [-1, 0], [0, -1], [1, 1], [-1, 1], [1, -1], [-1, -1], [1e-30, 1], [1, 1e-30], [-2, 1e-38],
                     [0, 0], [-1, -0.0], [5e-39, 1e-45]], dtype=np.float32)
    xy = np.concatenate([xy, edge])
    a = np.empty(len(xy), dtype=np.float32); b = np.empty(len(xy), dtype=np.float32)
    H.hostsim_phase(xy.ctypes.data, a.ctypes.data, len(xy))
    H.hostsim_phase_fast(xy.ctypes.data, b.ctypes.data, len(xy))
    assert np.abs(b).max() <= 0.5
    d = np.abs(a.astype(np.float64) / (2 * np.pi) - b.astype(np.float64))
    d = np.minimum(d, 1.0 - d)                             # +half a turn and -half a turn are the same direction
    assert d.max() < 1.2e-7, d.max()                       # turns (7.5e-7 rad)
    guard = H.hostsim_screen_guard()                       # turns
    assert abs(guard * 2 * np.pi - 2e-5) < 1e-6
    # two phase errors + rounding of their difference (|d| <= 1) and of d - k/8 (|u| < 2) + the reference's own three roundings
    # (two values under 8 rad, one difference under 16 rad) expressed in turns
    ref_side = (2 * 0.5 * np.spacing(np.float32(4.0)) + 0.5 * np.spacing(np.float32(8.0))) / (2 * np.pi)
    ours = 2 * d.max() + 0.5 * np.spacing(np.float32(0.5)) + 0.5 * np.spacing(np.float32(1.0))
    assert ours + ref_side < 0.25 * guard, (ours, ref_side, guard)
    # a zero sample has phase 0, whichever zero it is
    assert (b[len(xy) - 3] == 0) and abs(b[len(xy) - 2]) == 0.5
