"""development aid: dev/ref_short_runup.c over a bench capture - the residue of a zero-start scan of the reference's fp32 channel filter
against the reference's own trajectory, by distance from the start.
   python dev/ref_short_runup.py config4 2.0 [channels] [trials per channel]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dumpvdl2_amd import synth, workloads
from oracle import pyoracle as po
name, dur = sys.argv[1], float(sys.argv[2])
nchan = int(sys.argv[3]) if len(sys.argv) > 3 else 8
trials = int(sys.argv[4]) if len(sys.argv) > 4 else 40
so = "/tmp/ref_short_runup.so"
subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "dev", "ref_short_runup.c"), "-lm"])
L = C.CDLL(so)
cfg = getattr(workloads, name)(dur)
iq, _ = synth.synthesize(cfg)
raw = np.ascontiguousarray(iq.view(np.int16)); n = raw.size // 2
o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
A, B = o.lpf(); A = np.asarray(A, np.float32); B = np.asarray(B, np.float32)
i = np.arange(256, dtype=np.float32)
ang = (np.float32(2.0) * np.float32(np.pi) * i / np.float32(256.0)).astype(np.float32)
sin_t = np.concatenate([np.sin(ang.astype(np.float64)).astype(np.float32), [np.float32(0)]]).astype(np.float32); sin_t[256] = sin_t[0]
cos_t = np.concatenate([np.cos(ang.astype(np.float64)).astype(np.float32), [np.float32(1)]]).astype(np.float32); cos_t[256] = cos_t[0]
os_ = cfg.oversample; D = n // os_
P = lambda a: a.ctypes.data_as(C.c_void_p)
L.from_zero.restype = C.c_long
bin_, kmax = 256, 16384
nb = kmax // bin_
env = np.zeros(nb); env_abs = np.zeros(nb); coal = []
rng = np.random.default_rng(1)
chans = rng.choice(len(cfg.freqs), size=min(nchan, len(cfg.freqs)), replace=False)
per_trial = []
for c in chans:
    dphi = o.dphi(int(c)); mix = 1 if cfg.freqs[int(c)] != cfg.centerfreq else 0
    y = np.zeros(2 * D, np.float32)
    L.full(P(raw), C.c_long(n), C.c_int(os_), C.c_uint32(dphi), C.c_int(mix), P(sin_t), P(cos_t), P(A), P(B), P(y))
    for t in range(trials):
        s0 = int(rng.integers(2000, D - kmax - 10)) * os_
        e = np.zeros(nb); ea = np.zeros(nb)
        r = L.from_zero(P(raw), C.c_long(n), C.c_int(os_), C.c_uint32(dphi), C.c_int(mix), P(sin_t), P(cos_t), P(A), P(B), P(y), C.c_long(s0), C.c_long(kmax), C.c_int(bin_), P(e), P(ea))
        env = np.maximum(env, e); env_abs = np.maximum(env_abs, ea); coal.append(r); per_trial.append(e)
per_trial = np.array(per_trial)
print(f"{name} {dur}s os {os_}: {len(chans)} channels x {trials} starts; decimated samples until bit-identical: median {np.median(coal):.0f}, p90 {np.percentile(coal, 90):.0f}, max {max(coal)} (of {kmax} looked at)")
print("decimated samples after the start : worst |y - y_ref| / max|y_ref[n..n-3]| over all starts (p99 of starts), worst absolute")
for b in range(nb):
    if b < 8 or b % 8 == 0:
        print(f"  {b * bin_:6d} .. {b * bin_ + bin_ - 1:6d} : {env[b]:.3e} ({np.percentile(per_trial[:, b], 99):.3e})  {env_abs[b]:.3e}")
