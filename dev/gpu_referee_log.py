"""development aid (library built with -DVDL2_REF_DEBUG): the referee's event log of one channel, device against host build"""
import os, sys, ctypes as C, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dumpvdl2_amd import synth, vdl2hip, workloads
name, dur, ch = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
cfg = getattr(workloads, name)(dur)
iq, _ = synth.synthesize(cfg); raw = iq.view(np.uint8)
rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, 1, cfg.rx_max_ppm, max_block_bytes=raw.size)
rx.debug_option("ref_kinds", 1); rx.debug_option("ref_debug_chan", ch)
rx.feed(raw); rx.drain()
buf = (C.c_ulonglong * 4000)()
f = rx.L.vdl2hip_debug_ref_log; f.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_size_t]
n = f(rx.h, buf, 1000)
fl = lambda u: struct.unpack("f", struct.pack("I", u & 0xffffffff))[0]
rows = []
for i in range(n):
    e = buf[4 * i:4 * i + 4]
    rows.append((e[0] >> 56, e[0] & 0xffffffffffffff, fl(e[1]), fl(e[2]), fl(e[3])))
print(n, "entries")
for r in rows:
    if r[0] >= 2 or r[1] < 90000: print("reflog %d %d %.9g %.9g %.9g" % r)
print(list(rx.counters(ch).values()))
