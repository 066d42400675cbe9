"""ctypes binding of libvdl2hip.so - the host-side mirror of the reference's
process_buf_*() -> avlc_decoder_queue_push() boundary (include/vdl2hip.h).

The library is the product; this module only passes pointers and sizes.
It raises if the shared library is missing or no HIP device exists: there is
no CPU implementation to fall back to.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvdl2hip.so")

FMT_U8, FMT_S16LE = 0, 1
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
AVLC_COUNTER_NAMES = [
    "avlc.frames.processed", "avlc.errors.too_short", "avlc.frames.good", "avlc.errors.bad_fcs",
    "avlc.msg.air2gnd", "avlc.msg.air2air", "avlc.msg.air2all", "avlc.msg.gnd2air", "avlc.msg.gnd2gnd", "avlc.msg.gnd2all",
]
NUM_AVLC_COUNTERS = len(AVLC_COUNTER_NAMES)
AVLC_OK, AVLC_TOO_SHORT, AVLC_BAD_FCS = 0, 1, 2
ABI_VERSION = 6
MAX_DRAIN_LAG = int(os.environ.get("VDL2HIP_PY_MAX_DRAIN_LAG", "5"))   # include/vdl2hip.h: VDL2HIP_MAX_DRAIN_LAG (the variable: development builds with another depth)
EXPORTS = [
    "vdl2hip_abi_version", "vdl2hip_strerror", "vdl2hip_create", "vdl2hip_destroy", "vdl2hip_feed",
    "vdl2hip_feed_device", "vdl2hip_sync", "vdl2hip_drain", "vdl2hip_counters", "vdl2hip_set_profiling",
    "vdl2hip_drain_packed", "vdl2hip_pack_raw_frame", "vdl2hip_get_stats", "vdl2hip_get_stats_sized", "vdl2hip_stream", "vdl2hip_set_drain_lag", "vdl2hip_get_lpf", "vdl2hip_get_nco_step", "vdl2hip_read_decimated",
    "vdl2hip_avlc_counters", "vdl2hip_set_avlc_filter", "vdl2hip_statsd_lines", "vdl2hip_feed_pinned",
    "vdl2hip_group_create", "vdl2hip_group_destroy", "vdl2hip_group_feed", "vdl2hip_group_feed_pinned", "vdl2hip_group_sync", "vdl2hip_group_drain",
    "vdl2hip_group_set_drain_lag", "vdl2hip_group_counters", "vdl2hip_group_avlc_counters", "vdl2hip_group_size", "vdl2hip_group_ctx",
    "vdl2hip_group_uses_rccl", "vdl2hip_group_set_exchange", "vdl2hip_group_exchange",
]


class Cfg(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("centerfreq", C.c_uint32), ("oversample", C.c_uint32),
                ("sample_fmt", C.c_uint32), ("nchan", C.c_uint32), ("freqs", C.POINTER(C.c_uint32)),
                ("max_ppm", C.c_float), ("device", C.c_int32), ("max_block_bytes", C.c_uint32),
                ("chan_first", C.c_uint32), ("chan_count", C.c_uint32)]


class CFrame(C.Structure):
    _fields_ = [("chan", C.c_uint32), ("freq", C.c_uint32), ("idx", C.c_int32), ("len", C.c_uint32),
                ("octets", C.POINTER(C.c_uint8)), ("synd_weight", C.c_uint32), ("datalen_octets", C.c_uint32),
                ("num_fec_corrections", C.c_int32), ("frame_pwr_dbfs", C.c_float), ("nf_pwr_dbfs", C.c_float),
                ("ppm_error", C.c_float), ("burst_ord", C.c_int64), ("sync_sample", C.c_int64),
                ("end_sample", C.c_int64), ("avlc_status", C.c_uint32), ("dst_addr", C.c_uint32), ("src_addr", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("feeds", C.c_uint64), ("input_samples", C.c_uint64), ("chan_samples", C.c_uint64),
                ("chanfir_launches", C.c_uint64), ("chanfir_ms", C.c_double), ("phase_ms", C.c_double),
                ("sync_ms", C.c_double), ("walk_ms", C.c_double), ("burst_ms", C.c_double), ("nf_ms", C.c_double),
                ("bursts", C.c_uint64), ("frames", C.c_uint64),
                ("seg_adopted", C.c_uint64), ("seg_walked", C.c_uint64), ("front_sync_timeouts", C.c_uint64),
                ("overflow_feeds", C.c_uint64), ("cold_start_feeds", C.c_uint64),
                ("referee_scans", C.c_uint64), ("referee_cached", C.c_uint64), ("referee_refused", C.c_uint64), ("referee_short", C.c_uint64), ("referee_rewalks", C.c_uint64),
                ("referee_candidate_scans", C.c_uint64), ("referee_header_scans", C.c_uint64), ("referee_symbol_scans", C.c_uint64),
                ("referee_redone_next", C.c_uint64), ("referee_unmet", C.c_uint64), ("referee_retried", C.c_uint64)]


class PackedFrame(C.Structure):
    _fields_ = [("frame", CFrame), ("octets_off", C.c_uint64)]


FRAME_CB = C.CFUNCTYPE(None, C.POINTER(CFrame), C.c_void_p)
_lib = None


def load_library(path: str = None):
    """path: default = the in-tree build; VDL2HIP_LIB in the environment points development runs at a variant build"""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("VDL2HIP_LIB") or LIB_PATH
    # A ROCm build of PyTorch carries its own libamdhip64.so.7 / libhsa-runtime64.so.1, the same sonames the system ROCm has: whichever
    # is loaded first serves the whole process.  Loaded after this library (which is linked against /opt/rocm), torch ends up on a
    # runtime it was not built with and the next vdl2hip_create() fails with a device error - so where torch is installed it goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing - build it with dumpvdl2_amd.build.build(); there is no CPU fallback")
    L = C.CDLL(path)
    L.vdl2hip_abi_version.restype = C.c_int
    # (development: VDL2HIP_LIB_ANY_ABI=1 lets dev/gpu_variants.py time an OLDER build beside this one - older structures are prefixes of these)
    if L.vdl2hip_abi_version() != ABI_VERSION and not (os.environ.get("VDL2HIP_LIB_ANY_ABI") and L.vdl2hip_abi_version() < ABI_VERSION):      # (the structures below are this version's: a library of another would be read or written out of bounds)
        raise RuntimeError(f"{path} has ABI version {L.vdl2hip_abi_version()}, this binding is for {ABI_VERSION}")
    L.vdl2hip_strerror.restype = C.c_char_p
    L.vdl2hip_strerror.argtypes = [C.c_int]
    L.vdl2hip_create.argtypes = [C.POINTER(Cfg), C.POINTER(C.c_void_p)]
    L.vdl2hip_destroy.argtypes = [C.c_void_p]
    L.vdl2hip_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.vdl2hip_feed_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.vdl2hip_feed_pinned.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.vdl2hip_sync.argtypes = [C.c_void_p]
    L.vdl2hip_drain.argtypes = [C.c_void_p, FRAME_CB, C.c_void_p]
    L.vdl2hip_drain_packed.argtypes = [C.c_void_p, C.POINTER(PackedFrame), C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.vdl2hip_pack_raw_frame.argtypes = [C.POINTER(CFrame), C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t]
    L.vdl2hip_counters.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
    L.vdl2hip_avlc_counters.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
    L.vdl2hip_set_avlc_filter.argtypes = [C.c_void_p, C.c_int]
    L.vdl2hip_statsd_lines.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.vdl2hip_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.vdl2hip_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.vdl2hip_set_drain_lag.argtypes = [C.c_void_p, C.c_int]
    L.vdl2hip_stream.restype = C.c_void_p
    L.vdl2hip_stream.argtypes = [C.c_void_p]
    L.vdl2hip_get_lpf.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.vdl2hip_get_nco_step.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.vdl2hip_read_decimated.argtypes = [C.c_void_p, C.c_uint32, C.c_int64, C.c_void_p, C.c_size_t]
    L.vdl2hip_group_create.argtypes = [C.POINTER(Cfg), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_void_p)]
    L.vdl2hip_group_destroy.argtypes = [C.c_void_p]
    L.vdl2hip_group_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.vdl2hip_group_feed_pinned.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.vdl2hip_group_sync.argtypes = [C.c_void_p]
    L.vdl2hip_group_drain.argtypes = [C.c_void_p, FRAME_CB, C.c_void_p]
    L.vdl2hip_group_set_drain_lag.argtypes = [C.c_void_p, C.c_int]
    L.vdl2hip_group_counters.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
    L.vdl2hip_group_avlc_counters.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
    L.vdl2hip_group_size.restype = C.c_uint32
    L.vdl2hip_group_size.argtypes = [C.c_void_p]
    L.vdl2hip_group_ctx.restype = C.c_void_p
    L.vdl2hip_group_ctx.argtypes = [C.c_void_p, C.c_uint32]
    L.vdl2hip_group_uses_rccl.argtypes = [C.c_void_p]
    if hasattr(L, "vdl2hip_group_set_exchange"):            # (absent from older builds loaded through VDL2HIP_LIB for comparisons)
        L.vdl2hip_group_set_exchange.argtypes = [C.c_void_p, C.c_int]
        L.vdl2hip_group_exchange.argtypes = [C.c_void_p]
    L.vdl2hip_group_ctx.restype = C.c_void_p
    L.vdl2hip_group_ctx.argtypes = [C.c_void_p, C.c_uint32]
    _lib = L
    return L


class Vdl2HipError(RuntimeError):
    pass


def pack_raw_frame(frame: dict, station_id: Optional[str] = None, tv_sec: int = 0, tv_usec: int = 0) -> bytes:
    """One record of the reference's raw-frame archive (`--output raw:binary:file` / `--raw-frames-file`)."""
    L = load_library()
    octs = frame["octets"]
    buf = (C.c_uint8 * max(1, len(octs)))(*octs)
    f = CFrame(frame["chan"], frame["freq"], frame["idx"], len(octs), C.cast(buf, C.POINTER(C.c_uint8)),
               frame["synd_weight"], frame["datalen_octets"], frame["num_fec_corrections"], frame["frame_pwr_dbfs"],
               frame["nf_pwr_dbfs"], frame["ppm_error"], frame.get("burst_ord", 0), frame.get("sync_sample", 0),
               frame.get("end_sample", 0), 0, 0, 0)
    out = (C.c_uint8 * (len(octs) + 600))()
    n = L.vdl2hip_pack_raw_frame(C.byref(f), station_id.encode() if station_id else None, tv_sec, tv_usec, out, len(out))
    if n < 0:
        raise Vdl2HipError(L.vdl2hip_strerror(n).decode())
    return bytes(out[:n])


class Receiver:
    """One multi-channel VDL2 receiver on one GPU (= the reference's set of demod threads)."""

    def __init__(self, centerfreq: int, freqs: Sequence[int], oversample: int = 20, sample_fmt: int = FMT_S16LE,
                 max_ppm: float = 0.0, device: int = 0, max_block_bytes: int = 320000,
                 chan_first: int = 0, chan_count: int = 0):
        self.L = load_library()
        self.freqs = list(freqs)
        self._freq_arr = (C.c_uint32 * len(self.freqs))(*self.freqs)
        cfg = Cfg(C.sizeof(Cfg), centerfreq, oversample, sample_fmt, len(self.freqs), self._freq_arr,
                  max_ppm, device, max_block_bytes, chan_first, chan_count)
        h = C.c_void_p()
        self._chk(self.L.vdl2hip_create(C.byref(cfg), C.byref(h)), "vdl2hip_create")
        self.h = h
        self.chan_first = chan_first
        self.chan_count = chan_count or (len(self.freqs) - chan_first)

    def _chk(self, r, what):
        if r < 0:
            raise Vdl2HipError(f"{what}: {self.L.vdl2hip_strerror(r).decode()} ({r})")
        return r

    def close(self):
        if getattr(self, "h", None):
            self.L.vdl2hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def feed(self, raw) -> None:
        """process_buf_uchar()/process_buf_short(): one block of raw IQ bytes from host memory."""
        a = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        self._chk(self.L.vdl2hip_feed(self.h, a.ctypes.data, a.size), "vdl2hip_feed")

    def feed_pinned(self, host_ptr: int, nbytes: int) -> None:
        """one block from page-locked host memory, queued without waiting for the copy (see vdl2hip.h for the lifetime rule)"""
        self._chk(self.L.vdl2hip_feed_pinned(self.h, C.c_void_p(host_ptr), nbytes), "vdl2hip_feed_pinned")

    def feed_device(self, dev_ptr: int, nbytes: int) -> None:
        self._chk(self.L.vdl2hip_feed_device(self.h, C.c_void_p(dev_ptr), nbytes), "vdl2hip_feed_device")

    def feed_tensor(self, t) -> None:
        """a block resident on this receiver's device (torch tensor of raw bytes / int16 values)"""
        self.feed_device(t.data_ptr(), t.numel() * t.element_size())

    def feed_pinned_tensor(self, t) -> None:
        """a block in page-locked host memory (torch tensor made with pin_memory())"""
        self.feed_pinned(t.data_ptr(), t.numel() * t.element_size())

    def set_drain_lag(self, lag: int) -> None:
        self._chk(self.L.vdl2hip_set_drain_lag(self.h, lag), "vdl2hip_set_drain_lag")

    def sync(self) -> None:
        self._chk(self.L.vdl2hip_sync(self.h), "vdl2hip_sync")

    def drain(self) -> List[dict]:
        out: List[dict] = []

        def cb(fp, _user):
            f = fp.contents
            out.append(dict(chan=f.chan, freq=f.freq, idx=f.idx,
                            octets=bytes(C.string_at(f.octets, f.len)) if f.len else b"",
                            synd_weight=f.synd_weight, datalen_octets=f.datalen_octets,
                            num_fec_corrections=f.num_fec_corrections, frame_pwr_dbfs=f.frame_pwr_dbfs,
                            nf_pwr_dbfs=f.nf_pwr_dbfs, ppm_error=f.ppm_error, burst_ord=f.burst_ord,
                            sync_sample=f.sync_sample, end_sample=f.end_sample,
                            avlc_status=f.avlc_status, dst_addr=f.dst_addr, src_addr=f.src_addr))

        self._cb = FRAME_CB(cb)
        self._chk(self.L.vdl2hip_drain(self.h, self._cb, None), "vdl2hip_drain")
        return out

    def drain_packed(self, cap_frames: int = 1 << 16, cap_octets: int = 1 << 24):
        """One call for all queued frames: returns (records ctypes array view, octets bytes).  Used where a Python
        callback per frame would dominate (bench.py)."""
        if getattr(self, "_pk", None) is None or len(self._pk) < cap_frames or len(self._po) < cap_octets:
            self._pk = (PackedFrame * cap_frames)()
            self._po = (C.c_uint8 * cap_octets)()
        used = C.c_size_t(0)
        n = self._chk(self.L.vdl2hip_drain_packed(self.h, self._pk, cap_frames, self._po, cap_octets, C.byref(used)), "vdl2hip_drain_packed")
        return n, self._pk, memoryview(self._po)[:used.value]

    @staticmethod
    def unpack(n, recs, octets) -> List[dict]:
        out = []
        for i in range(n):
            f = recs[i].frame; off = recs[i].octets_off
            out.append(dict(chan=f.chan, freq=f.freq, idx=f.idx, octets=bytes(octets[off:off + f.len]),
                            synd_weight=f.synd_weight, datalen_octets=f.datalen_octets,
                            num_fec_corrections=f.num_fec_corrections, frame_pwr_dbfs=f.frame_pwr_dbfs,
                            nf_pwr_dbfs=f.nf_pwr_dbfs, ppm_error=f.ppm_error, burst_ord=f.burst_ord,
                            sync_sample=f.sync_sample, end_sample=f.end_sample,
                            avlc_status=f.avlc_status, dst_addr=f.dst_addr, src_addr=f.src_addr))
        return out

    def avlc_counters(self, chan: int) -> dict:
        a = (C.c_uint64 * NUM_AVLC_COUNTERS)()
        self._chk(self.L.vdl2hip_avlc_counters(self.h, chan, a), "vdl2hip_avlc_counters")
        return dict(zip(AVLC_COUNTER_NAMES, list(a)))

    def set_avlc_filter(self, on: bool) -> None:
        self._chk(self.L.vdl2hip_set_avlc_filter(self.h, int(on)), "vdl2hip_set_avlc_filter")

    def statsd_lines(self, ns: str = "dumpvdl2", cap: int = 1 << 20) -> List[str]:
        """the reference's statsd counter traffic since the previous call, one "<ns>.<freq>.<counter>:<delta>|c" per line"""
        buf = C.create_string_buffer(cap)
        n = self._chk(self.L.vdl2hip_statsd_lines(self.h, ns.encode(), buf, cap), "vdl2hip_statsd_lines")
        return buf.raw[:n].decode().splitlines()

    def counters(self, chan: int) -> dict:
        a = (C.c_uint64 * NUM_COUNTERS)()
        self._chk(self.L.vdl2hip_counters(self.h, chan, a), "vdl2hip_counters")
        return dict(zip(COUNTER_NAMES, list(a)))

    def debug_option(self, name: str, value: int) -> None:
        """test hook of the library (not part of include/vdl2hip.h): "no_fuse", "force_timeout" """
        f = self.L.vdl2hip_debug_option
        f.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
        self._chk(f(self.h, name.encode(), value), "vdl2hip_debug_option")

    def exact_window(self, chan: int, n_lo: int, n_hi: int) -> bool:
        """test hook: the referee's scan over decimated samples n_lo..n_hi of one channel (True: done; read them with read_decimated)"""
        f = self.L.vdl2hip_debug_exact_window
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_int64, C.c_int64]
        return bool(self._chk(f(self.h, chan, n_lo, n_hi), "vdl2hip_debug_exact_window"))

    def exact_window_many(self, chan: int, n_lo: int, n_hi: int, count: int, stride: int):
        """test hook: `count` wavefronts scan at once (channel chan + b, the stretch moved on by stride * b) -> (stretches done, kernel ms)"""
        f = self.L.vdl2hip_debug_exact_window_many
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_int64, C.c_int64, C.c_uint32, C.c_int64, C.POINTER(C.c_float)]
        ms = C.c_float(0)
        return self._chk(f(self.h, chan, n_lo, n_hi, count, stride, C.byref(ms)), "vdl2hip_debug_exact_window_many"), ms.value

    def scan_multi(self, chans, los, his, retry=False):
        """test hook: the stretches (chans[i], los[i] .. his[i]) made exact side by side (k_ref_scan_multi) -> (scans run, kernel ms).
        retry: as the product's launches - a scan that has not met its witness is listed and run again from further back"""
        f = self.L.vdl2hip_debug_scan_multi2
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_float)]
        ch = np.ascontiguousarray(chans, dtype=np.int32); lo = np.ascontiguousarray(los, dtype=np.int64); hi = np.ascontiguousarray(his, dtype=np.int64)
        ms = C.c_float(0)
        return self._chk(f(self.h, ch.ctypes.data, lo.ctypes.data, hi.ctypes.data, len(ch), 1 if retry else 0, C.byref(ms)), "vdl2hip_debug_scan_multi2"), ms.value

    def read_sync(self, chan: int, first: int, count: int):
        """test hook: (pf [count, 2] = tabulated {pherr with the referee's mark as its sign, slope}, cand [count] candidate bits) of one channel"""
        f = self.L.vdl2hip_debug_read_sync
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_int64, C.c_size_t, C.c_void_p, C.c_void_p]
        pf = np.zeros((count, 2), dtype=np.float32); cand = np.zeros(count, dtype=np.uint8)
        n = self._chk(f(self.h, chan, first, count, pf.ctypes.data, cand.ctypes.data), "vdl2hip_debug_read_sync")
        return pf[:n], cand[:n]

    def set_profiling(self, level) -> None:
        """0/False off, 1/True: time the channeliser kernel only, 2: every stage (a few percent slower)"""
        self._chk(self.L.vdl2hip_set_profiling(self.h, int(level)), "vdl2hip_set_profiling")

    def stats(self) -> dict:
        s = Stats()
        self._chk(self.L.vdl2hip_get_stats(self.h, C.byref(s)), "vdl2hip_get_stats")
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def stream(self) -> int:
        return self.L.vdl2hip_stream(self.h) or 0

    def lpf(self):
        A = (C.c_float * 3)(); B = (C.c_float * 3)()
        self._chk(self.L.vdl2hip_get_lpf(self.h, A, B), "vdl2hip_get_lpf")
        return np.array(A, dtype=np.float32), np.array(B, dtype=np.float32)

    def nco_step(self, chan: int) -> int:
        d = C.c_uint32()
        self._chk(self.L.vdl2hip_get_nco_step(self.h, chan, C.byref(d)), "vdl2hip_get_nco_step")
        return d.value

    def read_decimated(self, chan: int, first: int, count: int) -> np.ndarray:
        buf = np.zeros((count, 2), dtype=np.float32)
        n = self._chk(self.L.vdl2hip_read_decimated(self.h, chan, first, buf.ctypes.data, count), "vdl2hip_read_decimated")
        return buf[:n]


def _frame_dict(f):
    return dict(chan=f.chan, freq=f.freq, idx=f.idx, octets=bytes(C.string_at(f.octets, f.len)) if f.len else b"",
                synd_weight=f.synd_weight, datalen_octets=f.datalen_octets, num_fec_corrections=f.num_fec_corrections,
                frame_pwr_dbfs=f.frame_pwr_dbfs, nf_pwr_dbfs=f.nf_pwr_dbfs, ppm_error=f.ppm_error, burst_ord=f.burst_ord,
                sync_sample=f.sync_sample, end_sample=f.end_sample, avlc_status=f.avlc_status, dst_addr=f.dst_addr, src_addr=f.src_addr)


class ReceiverGroup:
    """vdl2hip_group: one receiver spread over several GPUs of this process (the C-level multi-GPU path; the per-process form
    used by bench.py is Receiver + dist.ShardedFeeder).  `devices` may name a device more than once (virtual shards)."""

    def __init__(self, centerfreq: int, freqs: Sequence[int], devices: Sequence[int], oversample: int = 20,
                 sample_fmt: int = FMT_S16LE, max_ppm: float = 0.0, max_block_bytes: int = 320000):
        self.L = load_library()
        self.freqs = list(freqs)
        self._freq_arr = (C.c_uint32 * len(self.freqs))(*self.freqs)
        cfg = Cfg(C.sizeof(Cfg), centerfreq, oversample, sample_fmt, len(self.freqs), self._freq_arr, max_ppm, 0, max_block_bytes, 0, 0)
        dev = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        r = self.L.vdl2hip_group_create(C.byref(cfg), dev, len(devices), C.byref(h))
        if r < 0:
            raise Vdl2HipError(f"vdl2hip_group_create: {self.L.vdl2hip_strerror(r).decode()} ({r})")
        self.h = h

    def _chk(self, r, what):
        if r < 0:
            raise Vdl2HipError(f"{what}: {self.L.vdl2hip_strerror(r).decode()} ({r})")
        return r

    def feed(self, raw) -> None:
        a = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        self._chk(self.L.vdl2hip_group_feed(self.h, a.ctypes.data, a.size), "vdl2hip_group_feed")

    def feed_pinned(self, host_ptr: int, nbytes: int) -> None:
        """one block from page-locked host memory, queued without waiting for the copy"""
        self._chk(self.L.vdl2hip_group_feed_pinned(self.h, C.c_void_p(host_ptr), nbytes), "vdl2hip_group_feed_pinned")

    def sync(self) -> None:
        self._chk(self.L.vdl2hip_group_sync(self.h), "vdl2hip_group_sync")

    def set_drain_lag(self, lag: int) -> None:
        self._chk(self.L.vdl2hip_group_set_drain_lag(self.h, lag), "vdl2hip_group_set_drain_lag")

    def drain(self) -> List[dict]:
        out: List[dict] = []
        self._cb = FRAME_CB(lambda fp, _u: out.append(_frame_dict(fp.contents)))
        self._chk(self.L.vdl2hip_group_drain(self.h, self._cb, None), "vdl2hip_group_drain")
        return out

    def drain_count(self) -> int:
        """deliver (and drop) every finished frame without a callback: the number of frames (throughput loops)"""
        return self._chk(self.L.vdl2hip_group_drain(self.h, C.cast(None, FRAME_CB), None), "vdl2hip_group_drain")

    def counters(self, chan: int) -> dict:
        a = (C.c_uint64 * NUM_COUNTERS)()
        self._chk(self.L.vdl2hip_group_counters(self.h, chan, a), "vdl2hip_group_counters")
        return dict(zip(COUNTER_NAMES, list(a)))

    def size(self) -> int:
        return self.L.vdl2hip_group_size(self.h)

    def uses_rccl(self) -> bool:
        return bool(self.L.vdl2hip_group_uses_rccl(self.h))

    EXCHANGE_FORMS = {"allgather": 0, "broadcast": 1}

    def set_exchange(self, form: str) -> None:
        """how the next blocks reach the members: 'allgather' (striped ingest over every member's own PCIe link, default) or 'broadcast'"""
        self._chk(self.L.vdl2hip_group_set_exchange(self.h, self.EXCHANGE_FORMS[form]), "vdl2hip_group_set_exchange")

    def exchange(self) -> str:
        """what the last feed did"""
        return {0: "broadcast/peer-copy", 1: "broadcast/rccl", 2: "allgather/peer-copy", 3: "allgather/rccl"}.get(self.L.vdl2hip_group_exchange(self.h), "none yet")

    def stats(self) -> dict:
        """the members' statistics, summed"""
        tot = {}
        for i in range(self.size()):
            st = Stats()
            self._chk(self.L.vdl2hip_get_stats(C.c_void_p(self.L.vdl2hip_group_ctx(self.h, i)), C.byref(st)), "vdl2hip_get_stats")
            for k, _ in Stats._fields_:
                tot[k] = tot.get(k, 0) + getattr(st, k)
        return tot

    def read_decimated(self, chan: int, first: int, count: int):
        """`count` decimated (re, im) pairs of channel `chan` from the member that owns it (parity tests)"""
        for i in range(self.size()):
            ctx = self.L.vdl2hip_group_ctx(self.h, i)
            buf = np.empty(2 * count, dtype=np.float32)
            n = self.L.vdl2hip_read_decimated(C.c_void_p(ctx), chan, first, buf.ctypes.data, count)
            if n >= 0:
                return buf[:2 * n].reshape(-1, 2)
        raise Vdl2HipError("no member owns that channel")

    def close(self):
        if getattr(self, "h", None):
            self.L.vdl2hip_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
