"""ctypes binding of the CPU oracle (oracle/libvdl2oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libvdl2oracle.so")
_LIB_FAST = os.path.join(_HERE, "libvdl2oracle_fast.so")     # the same source built -O3 -ffast-math, as upstream builds the reference

COUNTER_NAMES = [
    "demod.sync.good", "decoder.crc.good", "decoder.crc.bad", "decoder.errors.no_header",
    "decoder.errors.too_long", "decoder.errors.no_fec", "decoder.errors.data_truncated",
    "decoder.errors.fec_truncated", "decoder.errors.deinterleave_data",
    "decoder.errors.deinterleave_fec", "decoder.errors.fec_bad", "decoder.errors.bitstream",
    "decoder.errors.truncated_octets", "decoder.errors.unstuff", "decoder.blocks.processed",
    "decoder.blocks.fec_ok", "decoder.msg.good", "decoder.msg.good_loud",
    "demod.ppm_reject", "demod.slicer_neg_idx",
]
NUM_COUNTERS = len(COUNTER_NAMES)
FMT_U8, FMT_S16LE = 0, 1


class Frame(C.Structure):
    _fields_ = [
        ("chan", C.c_int32), ("freq", C.c_uint32), ("idx", C.c_int32), ("len", C.c_uint32),
        ("octets_off", C.c_uint64), ("synd_weight", C.c_uint32), ("datalen_octets", C.c_uint32),
        ("num_fec_corrections", C.c_int32), ("frame_pwr_dbfs", C.c_float), ("nf_pwr_dbfs", C.c_float),
        ("ppm_error", C.c_float), ("burst_ord", C.c_int64), ("sync_sample", C.c_int64),
        ("end_sample", C.c_int64),
    ]


def build(force=False):
    """Compile the oracle (and oracle/_ref when the reference tree is present)."""
    src_t = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("vdl2_oracle.c", "vdl2_oracle.h", "Makefile"))
    if force or any(not os.path.exists(p) or os.path.getmtime(p) < src_t for p in (_LIB, _LIB_FAST)):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None
_libs = {}


def lib(variant="strict"):
    """variant "strict" (default: the oracle proper) or "fast" (-O3 -ffast-math build of the same source)"""
    global _lib
    if variant != "strict":
        if variant not in _libs:
            build()
            _libs[variant] = _bind(C.CDLL(_LIB_FAST))
        return _libs[variant]
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(_LIB))
    return _lib


RUN_THREAD_PER_CHANNEL, RUN_WORKQUEUE = 1, 2


def _bind(L):
    L.vdl2o_run.restype = C.c_int
    L.vdl2o_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int]
    L.vdl2o_create.restype = C.c_void_p
    L.vdl2o_create.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.c_int, C.c_uint32, C.c_int, C.c_float]
    L.vdl2o_destroy.argtypes = [C.c_void_p]
    L.vdl2o_process.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    L.vdl2o_num_frames.restype = C.c_size_t
    L.vdl2o_num_frames.argtypes = [C.c_void_p]
    L.vdl2o_frames.restype = C.POINTER(Frame)
    L.vdl2o_frames.argtypes = [C.c_void_p]
    L.vdl2o_octets.restype = C.POINTER(C.c_uint8)
    L.vdl2o_octets.argtypes = [C.c_void_p]
    L.vdl2o_clear_frames.argtypes = [C.c_void_p]
    L.vdl2o_counters.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    L.vdl2o_get_lpf.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.vdl2o_get_dphi.restype = C.c_uint32
    L.vdl2o_get_dphi.argtypes = [C.c_void_p, C.c_int]
    L.vdl2o_get_sincos_lut.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.vdl2o_trace_decimated.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.vdl2o_trace_count.restype = C.c_size_t
    L.vdl2o_trace_count.argtypes = [C.c_void_p]
    L.vdl2o_trace_all.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.vdl2o_decimated_count.restype = C.c_int64
    L.vdl2o_decimated_count.argtypes = [C.c_void_p, C.c_int]
    L.vdl2o_rs_decode.restype = C.c_int
    L.vdl2o_rs_decode.argtypes = [C.c_void_p, C.c_int]
    L.vdl2o_rs_encode.argtypes = [C.c_void_p, C.c_void_p]
    L.vdl2o_header_decode.restype = C.c_uint32
    L.vdl2o_header_decode.argtypes = [C.POINTER(C.c_uint32)]
    L.vdl2o_header_parity.restype = C.c_uint32
    L.vdl2o_header_parity.argtypes = [C.c_uint32]
    L.vdl2o_crc16.restype = C.c_uint16
    L.vdl2o_crc16.argtypes = [C.c_void_p, C.c_uint32, C.c_uint16]
    L.vdl2o_chebyshev.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    return L


class Oracle:
    """One reference-equivalent receiver: nchan channels fed with raw IQ blocks."""

    def __init__(self, centerfreq, freqs, oversample=20, sample_fmt=FMT_S16LE, max_ppm=0.0, variant="strict"):
        self.L = lib(variant)
        self.freqs = list(freqs)
        arr = (C.c_uint32 * len(freqs))(*freqs)
        self.h = self.L.vdl2o_create(centerfreq, arr, len(freqs), oversample, sample_fmt, max_ppm)
        self._trace = None

    def close(self):
        if self.h:
            self.L.vdl2o_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def process(self, raw, block_bytes=320000, nthreads=1):
        """Feed raw bytes in block_bytes pieces like process_iq_file() (dumpvdl2.c:353-356)."""
        raw = np.ascontiguousarray(np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw.view(np.uint8).reshape(-1))
        n = raw.size
        base = raw.ctypes.data
        off = 0
        while off < n:
            m = min(block_bytes, n - off)
            self.L.vdl2o_process(self.h, base + off, m, nthreads)
            off += m

    def run(self, raw, block_bytes=320000, mode=RUN_THREAD_PER_CHANNEL, nthreads=0):
        """The whole capture with persistent threads (vdl2_oracle.h: vdl2o_run): mode RUN_THREAD_PER_CHANNEL is the reference's
        own threading (one thread per channel + producer, two barriers per block, serial conversion), RUN_WORKQUEUE a pool of
        `nthreads` workers with parallel conversion.  Same frames, in the same order, as process()."""
        raw = np.ascontiguousarray(np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw.view(np.uint8).reshape(-1))
        r = self.L.vdl2o_run(self.h, raw.ctypes.data, raw.size, block_bytes, mode, nthreads or (os.cpu_count() or 1))
        if r != 0:
            raise RuntimeError(f"vdl2o_run failed ({r})")

    def trace(self, chan, cap):
        self._trace = np.zeros((cap, 2), dtype=np.float32)
        self.L.vdl2o_trace_decimated(self.h, chan, self._trace.ctypes.data, cap)
        return self._trace

    def trace_all(self, cap):
        """Record (lp_re, lp_im) of every channel: returns array [nchan, cap, 2]."""
        self._trace_all = np.zeros((len(self.freqs), cap, 2), dtype=np.float32)
        self.L.vdl2o_trace_all(self.h, self._trace_all.ctypes.data, cap)
        return self._trace_all

    def decimated_count(self, chan=0):
        return self.L.vdl2o_decimated_count(self.h, chan)

    def trace_count(self):
        return self.L.vdl2o_trace_count(self.h)

    def frames(self, clear=True):
        n = self.L.vdl2o_num_frames(self.h)
        fr = self.L.vdl2o_frames(self.h)
        oc = self.L.vdl2o_octets(self.h)
        out = []
        for i in range(n):
            f = fr[i]
            out.append(dict(
                chan=f.chan, freq=f.freq, idx=f.idx,
                octets=bytes(C.string_at(C.addressof(oc.contents) + f.octets_off, f.len)) if f.len else b"",
                synd_weight=f.synd_weight, datalen_octets=f.datalen_octets,
                num_fec_corrections=f.num_fec_corrections, frame_pwr_dbfs=f.frame_pwr_dbfs,
                nf_pwr_dbfs=f.nf_pwr_dbfs, ppm_error=f.ppm_error, burst_ord=f.burst_ord,
                sync_sample=f.sync_sample, end_sample=f.end_sample))
        if clear:
            self.L.vdl2o_clear_frames(self.h)
        return out

    def counters(self, chan):
        a = (C.c_uint64 * NUM_COUNTERS)()
        self.L.vdl2o_counters(self.h, chan, a)
        return dict(zip(COUNTER_NAMES, list(a)))

    def lpf(self):
        A = (C.c_float * 3)(); B = (C.c_float * 3)()
        self.L.vdl2o_get_lpf(self.h, A, B)
        return np.array(A, dtype=np.float32), np.array(B, dtype=np.float32)

    def dphi(self, chan):
        return self.L.vdl2o_get_dphi(self.h, chan)


AVLC_COUNTER_NAMES = ["avlc.frames.processed", "avlc.errors.too_short", "avlc.frames.good", "avlc.errors.bad_fcs",
                      "avlc.msg.air2gnd", "avlc.msg.air2air", "avlc.msg.air2all", "avlc.msg.gnd2air", "avlc.msg.gnd2gnd",
                      "avlc.msg.gnd2all"]


def avlc_screen(octets):
    """(status, dst, src, dir) of avlc_parse()'s first checks: status 0 ok / 1 too short / 2 bad FCS"""
    L = lib()
    b = bytes(octets)
    dst = C.c_uint32(); src = C.c_uint32(); d = C.c_int()
    L.vdl2o_avlc_screen.restype = C.c_int
    L.vdl2o_avlc_screen.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    st = L.vdl2o_avlc_screen(b, len(b), C.byref(dst), C.byref(src), C.byref(d))
    return st, dst.value, src.value, d.value


def avlc_counters(frames, nchan):
    """the reference's per-channel avlc.* statsd counters for a list of frames (dicts with chan, octets)"""
    out = [[0] * len(AVLC_COUNTER_NAMES) for _ in range(nchan)]
    for f in frames:
        c = out[f["chan"]]
        c[0] += 1
        st, _, _, d = avlc_screen(f["octets"])
        if st == 1:
            c[1] += 1
        elif st == 2:
            c[3] += 1
        else:
            c[2] += 1
            if d:
                c[3 + d] += 1
    return out


def crc16_x25(data, init=0xFFFF):
    b = bytes(data)
    return lib().vdl2o_crc16(b, len(b), init)


def rs_encode(data249):
    d = (C.c_uint8 * 249)(*data249)
    p = (C.c_uint8 * 6)()
    lib().vdl2o_rs_encode(d, p)
    return bytes(p)


def rs_decode(block255, fec_octets):
    d = (C.c_uint8 * 255)(*block255)
    r = lib().vdl2o_rs_decode(d, fec_octets)
    return r, bytes(d)
