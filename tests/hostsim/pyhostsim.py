"""ctypes driver of tests/hostsim/libhostsim.so (CPU build of the walker / burst-decoder kernels' source)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_LIB = os.path.join(_HERE, "libhostsim.so")
NUM_COUNTERS = 20


class OutFrame(C.Structure):
    _fields_ = [("chan", C.c_int32), ("idx", C.c_int32), ("len", C.c_uint32), ("pool_off", C.c_uint32),
                ("synd_weight", C.c_uint32), ("datalen_octets", C.c_uint32), ("num_fec_corrections", C.c_int32),
                ("frame_pwr_dbfs", C.c_float), ("nf_pwr_dbfs", C.c_float), ("ppm_error", C.c_float),
                ("burst_ord", C.c_int64), ("sync_sample", C.c_int64), ("end_sample", C.c_int64), ("nf_upd", C.c_int64),
                ("avlc_status", C.c_uint32), ("dst_addr", C.c_uint32), ("src_addr", C.c_uint32), ("pad_", C.c_uint32)]


def build(reverse_lanes=False):
    """reverse_lanes: compile with -DVDL2_HOST_REVERSE_LANES - the 64 lanes of every wave phase run in the opposite order, which
    must not change any result (a phase whose lanes depended on each other would be a race on the device)"""
    lib = _LIB.replace(".so", "_rev.so") if reverse_lanes else _LIB
    srcs = [os.path.join(_HERE, "hostsim.cpp")] + [os.path.join(_ROOT, "dumpvdl2_amd", "csrc", f) for f in ("vdl2_core.h", "tables.h", "design.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared"]
                              + (["-DVDL2_HOST_REVERSE_LANES"] if reverse_lanes else []) + ["-o", lib, srcs[0]])
    return lib


class HostSim:
    def __init__(self, freqs, max_ppm=0.0, cap_log2=21, reverse_lanes=False):
        self.L = C.CDLL(build(reverse_lanes))
        self.L.hostsim_create.restype = C.c_void_p
        self.L.hostsim_create.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.c_float, C.c_int]
        self.L.hostsim_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        self.L.hostsim_num_frames.restype = C.c_int64
        self.L.hostsim_num_frames.argtypes = [C.c_void_p]
        self.L.hostsim_frames.restype = C.POINTER(OutFrame)
        self.L.hostsim_frames.argtypes = [C.c_void_p]
        self.L.hostsim_pool.restype = C.POINTER(C.c_uint8)
        self.L.hostsim_pool.argtypes = [C.c_void_p]
        self.L.hostsim_counters.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]
        self.L.hostsim_avlc_counters.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]
        self.L.hostsim_destroy.argtypes = [C.c_void_p]
        self.L.hostsim_set_segments.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        self.L.hostsim_segment_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        assert self.L.hostsim_sizeof_outframe() == C.sizeof(OutFrame)
        self.n = len(freqs)
        self.h = self.L.hostsim_create(self.n, (C.c_uint32 * self.n)(*freqs), max_ppm, cap_log2)

    def feed(self, y):
        """y: float32 [nchan, D, 2]"""
        y = np.ascontiguousarray(y, dtype=np.float32)
        assert y.shape[0] == self.n and y.shape[2] == 2
        r = self.L.hostsim_feed(self.h, y.ctypes.data, y.shape[1])
        assert r >= 0, "hostsim output overflow"

    def frames(self):
        n = self.L.hostsim_num_frames(self.h)
        fr = self.L.hostsim_frames(self.h)
        pool = self.L.hostsim_pool(self.h)
        base = C.addressof(pool.contents) if n else 0
        out = []
        for i in range(n):
            f = fr[i]
            out.append(dict(chan=f.chan, idx=f.idx, octets=bytes(C.string_at(base + f.pool_off, f.len)) if f.len else b"",
                            synd_weight=f.synd_weight, datalen_octets=f.datalen_octets,
                            num_fec_corrections=f.num_fec_corrections, frame_pwr_dbfs=f.frame_pwr_dbfs,
                            nf_pwr_dbfs=f.nf_pwr_dbfs, ppm_error=f.ppm_error, burst_ord=f.burst_ord,
                            sync_sample=f.sync_sample, end_sample=f.end_sample,
                            avlc_status=f.avlc_status, dst_addr=f.dst_addr, src_addr=f.src_addr))
        return out

    def set_exact(self, y_exact):
        """referee on: y_exact float32 [nchan, D, 2] = the reference's own decimated samples (the oracle's trace) of the whole capture"""
        y = np.ascontiguousarray(y_exact, dtype=np.float32)
        assert y.shape[0] == self.n and y.shape[2] == 2
        self.L.hostsim_set_exact.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        self.L.hostsim_set_exact(self.h, y.ctypes.data, y.shape[1])

    def referee_stats(self):
        a = (C.c_int64 * 7)()
        self.L.hostsim_referee_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        self.L.hostsim_referee_stats(self.h, a)
        return {"marked_candidates": a[0], "candidate_bits": a[1], "exact_windows": a[2], "exact_samples": a[3], "walker_windows_first_feed": a[4],
                "decisions_checked": a[5], "channels_walked_again": a[6]}

    def set_prescan(self, on=True):
        """referee: the stretches around marked candidates are made exact before the walk (the device's default), or not"""
        self.L.hostsim_set_prescan.argtypes = [C.c_void_p, C.c_int]
        self.L.hostsim_set_prescan(self.h, 1 if on else 0)

    def set_optimistic(self, on=True):
        """referee mode: decisions within the margin are taken on the samples as they are and checked afterwards (the device's mode for long
        feeds; default) / the referee is asked on the spot"""
        self.L.hostsim_set_optimistic.argtypes = [C.c_void_p, C.c_int]
        self.L.hostsim_set_optimistic(self.h, int(on))

    def bursts(self):
        """debugging aid: every burst descriptor the walker has emitted, as (chan, sync_sample, t_first, nsym, tl_bits, syndrome, vdphi_err*1e6, prev_n)"""
        a = (C.c_int64 * (8 * 65536))()
        self.L.hostsim_bursts.restype = C.c_int64
        self.L.hostsim_bursts.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int64]
        n = self.L.hostsim_bursts(self.h, a, 65536)
        return [tuple(a[8 * i:8 * i + 8]) for i in range(n)]

    def read_sync(self, chan, first, count):
        """(pf [count, 2], cand [count]) as the sync stage left them (cf. vdl2hip.Receiver.read_sync)"""
        pf = np.zeros((count, 2), dtype=np.float32); cand = np.zeros(count, dtype=np.uint8)
        self.L.hostsim_read_sync.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        self.L.hostsim_read_sync(self.h, chan, first, count, pf.ctypes.data, cand.ctypes.data)
        return pf, cand

    def set_segments(self, seg_min, seg_max=32):
        """walk long feeds in speculative segments, as k_walk_spec / k_walk_stitch do on the device"""
        self.L.hostsim_set_segments(self.h, seg_min, seg_max)

    def set_two_tier(self, on=True):
        """compute the sync metric the way k_sync does: screening value everywhere, exact arithmetic only where it can matter"""
        self.L.hostsim_set_two_tier.argtypes = [C.c_void_p, C.c_int]
        self.L.hostsim_set_two_tier(self.h, int(on))

    def two_tier_stats(self):
        a = (C.c_int64 * 2)()
        self.L.hostsim_two_tier_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        self.L.hostsim_two_tier_stats(self.h, a)
        return {"exact": a[0], "total": a[1]}

    def segment_stats(self):
        a = (C.c_uint32 * 2)()
        self.L.hostsim_segment_stats(self.h, a)
        return {"adopted": a[0], "walked": a[1]}

    def counters(self, chan):
        a = (C.c_ulonglong * NUM_COUNTERS)()
        self.L.hostsim_counters(self.h, chan, a)
        return list(a)

    def avlc_counters(self, chan):
        a = (C.c_ulonglong * 10)()
        self.L.hostsim_avlc_counters(self.h, chan, a)
        return list(a)

    def close(self):
        if self.h:
            self.L.hostsim_destroy(self.h)
            self.h = None
