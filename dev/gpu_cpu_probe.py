"""What the GPU box's host gives the CPU baseline: cgroup CPU quota / cpuset, and the oracle's throughput against the number of
threads (work-queue runs over a 4 s x 256-channel sample, then the reference threading)."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/proc/loadavg"):
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, "-", e.__class__.__name__)
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
print(subprocess.run("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA|MHz'", shell=True, capture_output=True, text=True).stdout)
from oracle import pyoracle as po
from dumpvdl2_amd import synth, workloads
cfg = workloads.config4(4.0)
iq, _ = synth.synthesize(cfg)
raw = iq.view(np.uint8)
n = iq.size // 2
def run(mode, nth, blk=320000, variant="strict"):
    o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm, variant=variant)
    t = time.perf_counter(); o.run(raw, block_bytes=blk, mode=mode, nthreads=nth); dt = time.perf_counter() - t
    o.close(); return dt
for nth in (8, 16, 32, 64, 96, 128, 192, 256):
    dt = min(run(po.RUN_WORKQUEUE, nth, 1 << 22) for _ in range(2))
    print(f"workqueue {nth:3d} threads, 4 MiB blocks: {dt:.3f} s  {n / dt / 1e6:.2f} MS/s  {dt * nth / (n * 256) * 1e9:.1f} ns per channel-sample per thread")
for blk in (320000, 1 << 22):
    dt = min(run(po.RUN_THREAD_PER_CHANNEL, 0, blk) for _ in range(2))
    print(f"thread per channel (256+1), {blk}-byte blocks: {dt:.3f} s  {n / dt / 1e6:.2f} MS/s")
o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
t = time.perf_counter(); o.process(raw, block_bytes=320000, nthreads=256); dt = time.perf_counter() - t
print(f"spawn per block, 256 threads: {dt:.3f} s  {n / dt / 1e6:.2f} MS/s")
