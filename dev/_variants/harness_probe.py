import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dumpvdl2_amd import build, synth, workloads
cfg = workloads.config4(4.0)
iq, _ = synth.synthesize(cfg)
d = tempfile.mkdtemp()
path = os.path.join(d, "cap.cs16"); iq.tofile(path)
exe = build.build_harness(os.path.join(d, "dropin_harness"))
def run(label, nfreq, batch, prefix=(), extra_env=None):
    env = dict(os.environ, HARNESS_TIMING="1", VDL2HIP_DROPIN_TIMING="1", VDL2HIP_DROPIN_BATCH=batch)
    env.update(extra_env or {})
    p = subprocess.run(list(prefix) + [exe, path, str(cfg.oversample), str(cfg.centerfreq)] + [str(f) for f in cfg.freqs[:nfreq]], capture_output=True, text=True, timeout=600, env=env)
    t = [l for l in p.stderr.splitlines() if l.startswith("HARNESS")]
    tl = [l for l in p.stderr.splitlines() if "dropin timing" in l]
    print(label, "|", t[0] if t else p.stderr[-300:], flush=True)
    if tl: print("    ", tl[-1], flush=True)
print("rx_max_ppm", cfg.rx_max_ppm)
P = {"HARNESS_MAX_PPM": str(cfg.rx_max_ppm)}
run("256 channels, batch 1, --max-ppm as the bench", 256, "1", extra_env=P)
run("256 channels, batch 16, --max-ppm as the bench", 256, "16", extra_env=P)
run("256 channels, batch 32, --max-ppm as the bench", 256, "32", extra_env=P)
run("8 channels, batch 1, --max-ppm as the bench", 8, "1", extra_env=P)
