"""tools/vdl2hip_iqfile (the --iq-file work-alike, no barriers) on a 256-channel file of 64 s (config4's 16 s capture four times over): wall
clock with the blocks collected (default) and one by one (--blocks-per-feed 1), minus the start-up measured on an empty file.
usage: python dev/gpu_iqfile_rate.py"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dumpvdl2_amd import build, synth, workloads
cfg = workloads.config4(16.0)
iq, _ = synth.synthesize(cfg)
d = tempfile.mkdtemp()
path = os.path.join(d, "cap.cs16"); empty = os.path.join(d, "empty.cs16")
with open(path, "wb") as f:
    for _ in range(4): iq.tofile(f)
open(empty, "wb").close()
exe = build.build_cli(os.path.join(d, "vdl2hip_iqfile"))
base = [exe, "--sample-format", "S16_LE", "--oversample", str(cfg.oversample), "--centerfreq", str(cfg.centerfreq), "--max-ppm", str(cfg.rx_max_ppm)]
freqs = [str(f) for f in cfg.freqs]
def run(extra, file):
    t0 = time.perf_counter()
    p = subprocess.run(base + extra + ["--iq-file", file] + freqs, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    dt = time.perf_counter() - t0
    return dt, sum(1 for l in p.stdout.splitlines() if "[S:" in l), p.returncode
run([], empty)
t_start = min(run([], empty)[0] for _ in range(3))
nblk = os.path.getsize(path) / 320000
print(f"start-up (empty file): {t_start * 1e3:.0f} ms; file: {nblk:.0f} blocks of 320 000 bytes, 256 channels")
for label, extra in (("blocks collected (default)", []), ("--blocks-per-feed 64", ["--blocks-per-feed", "64"]), ("--blocks-per-feed 1", ["--blocks-per-feed", "1"])):
    dt, n, rc = run(extra, path)
    print(f"{label}: {dt:.2f} s wall, {(dt - t_start) / nblk * 1e3:.3f} ms per block after start-up ({os.path.getsize(path) / 4 / (dt - t_start) / 2.1e6:.0f}x real time), {n} frames, rc {rc}", flush=True)
