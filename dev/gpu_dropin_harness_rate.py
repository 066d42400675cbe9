"""The drop-in adapter itself (csrc/dropin.c under tests/dropin_harness.c, the stand-in for an unmodified dumpvdl2 main()) on a
256-channel capture file: ms per 320 000-byte block with the blocks collected (default) and one by one (VDL2HIP_DROPIN_BATCH=1).
usage: python dev/gpu_dropin_harness_rate.py [config4] [seconds]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dumpvdl2_amd import build, synth, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "config4"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
cfg = getattr(workloads, name)(secs)
iq, _ = synth.synthesize(cfg)
d = tempfile.mkdtemp()
path = os.path.join(d, "cap.cs16"); iq.tofile(path)
exe = build.build_harness(os.path.join(d, "dropin_harness"))
outs = {}
for batch in (None, "32", "1"):
    env = dict(os.environ, HARNESS_TIMING="1", VDL2HIP_DROPIN_TIMING="1", HARNESS_MAX_PPM=str(cfg.rx_max_ppm))
    env.pop("VDL2HIP_DROPIN_BATCH", None)
    if batch: env["VDL2HIP_DROPIN_BATCH"] = batch
    p = subprocess.run([exe, path, str(cfg.oversample), str(cfg.centerfreq)] + [str(f) for f in cfg.freqs], capture_output=True, text=True, timeout=600, env=env)
    frames = []
    for l in p.stdout.splitlines():
        if l.startswith("FRAME"):
            kv = dict(t.split("=", 1) for t in l.split()[1:])
            frames.append((int(kv["freq"]), kv["octets"], int(kv["idx"]), int(kv["S"]), int(kv["L"]), int(kv["F"]), float(kv["pwr"]), float(kv["nf"]), float(kv["ppm"])))
    frames.sort()
    outs[batch] = frames
    t = [l for l in p.stderr.splitlines() if l.startswith("HARNESS")]
    for l in p.stderr.splitlines():
        if "dropin timing" in l: print("   ", l)
    print(f"{name} {len(cfg.freqs)} channels, {secs:g} s, VDL2HIP_DROPIN_BATCH={batch or 'default'}: {t[0] if t else p.stderr[-300:]}; frames {len(frames)}", flush=True)
def same(a, b):
    return len(a) == len(b) and all(x[:6] == y[:6] and all(abs(p - q) <= 0.0011 for p, q in zip(x[6:], y[6:])) for x, y in zip(a, b))
print("same frames whatever the collecting (octets and integer metadata identical, floats as printed within 0.001):", same(outs[None], outs["32"]) and same(outs[None], outs["1"]))
