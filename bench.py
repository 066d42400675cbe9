#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X-native VDL2 hot path.

Metric (BASELINE.json): IQ MS/s demodulated end-to-end.  One "step" = one pass of the whole hot path (K1 channeliser ...
K5 burst decoder, frames delivered to host memory) over one 16 s block of synthetic 2.1 MS/s cs16 IQ, demodulated on ALL
channels of the workload.  Default workload: north_star's 256-channel configuration (BASELINE configs[3], "config4") - the
largest configuration, which fits one GPU.  The same command runs at any N:

  N = 1   one GPU demodulates all 256 channels;
  N > 1   (one process per GPU, launched by torch.distributed.run) rank r demodulates channels [r*256/N, (r+1)*256/N) of
          the same stream (32 per GPU at N = 8); the only exchange puts every raw IQ block on every GPU with RCCL, inside
          the timed steps and overlapped with the demodulation of the previous block: broadcast from the ingest rank
          (north_star's form) or all-gather of per-rank stripes (each rank ingests 1/N of the block over its own PCIe
          link).  BOTH forms are timed bare and demodulating; `value` is the form `--exchange` names (`auto`: the faster
          bare exchange), the other one is printed beside it (`by_exchange`).

So total work is fixed ("scaling": "strong") and value(N=8) / value(N=1) is the 8-GPU speed-up north_star asks for.

`value` is the host-fed rate (cs16 block in page-locked host memory -> all frames of the block delivered):
the block crosses PCIe inside every timed step, overlapped with compute by the library's copy stream.  The K timed steps
are run `--repeats` times (default 3); `value` / `ms_per_step` are the MEDIAN repeat, all repeats are listed.  The same is
then done with the block already resident in HBM (`value_hbm_resident`).  At N = 1 the line also carries
  * `projected_scaling`: the rank-sized workload of the 8-GPU split (32 of the 256 channels, three ranks' shards, block
    resident in HBM as an RCCL exchange leaves it) timed on this GPU -> the compute-side ceiling t256 / max t32 of the
    speed-up, with the ingest bounds beside it;
  * `config.secondary`: the 64- and 8-channel configurations and a burst-dense variant of the 256-channel one (back-end
    sensitivity), each with per-stage kernel times.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import platform
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

OS = 20
# algorithmic bytes per channel-sample = 4 B (cs16 I+Q, charged once per CHANNEL, as the reference streams its
# buffer) + 8/os B (one float2 decimated output).  The kernel fetches the block once per XCD and shares it between channels
# through LDS/L2, so this figure exceeds the 8 TB/s peak by construction; it is kept because SURVEY 8.5 defines it.
ALGO_BYTES = 4.0 + 8.0 / OS
FLOP_PER_CHAN_SAMPLE = 30.0                       # LUT interpolation 6 + mix 6 + 2 x 9 IIR
HBM_PEAK_GBS = 8000.0                             # MI355X_MICROARCH.md: 8 TB/s spec
CLOCK_WARMUP_S = 0.05                             # untimed steps before every set of timed regions (see Case.timed)
VALU_PEAK_TFLOPS = 157.3                          # MI355X_MICROARCH.md: peak FP32 vector
WORKLOAD_INDEX = {"config2": 1, "config3": 2, "config4": 3, "config5": 4}


def cpu_info():
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        cc = subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    except Exception:  # noqa: BLE001
        cc = "gcc ?"
    return {"nproc": os.cpu_count(), "cpus_available": cpus_available(), "cpu_model": model or platform.processor(), "compiler": cc}


def cpus_available():
    """CPUs this process may actually use: the affinity mask, cut down to the cgroup's CPU-time quota where one is set (the GPU
    boxes show 256 hardware threads but run the container with cpu.max = 16 CPUs' worth of time: more runnable threads than that
    only add scheduling overhead)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()[:2]
            if q != "max":
                n = max(1, min(n, int(round(int(q) / int(per)))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = max(1, min(n, int(round(q / per))))
    except (OSError, ValueError):
        pass
    return n


class Case:
    """One workload on this rank: receiver over the rank's channel shard + the feeder that brings the blocks."""

    def __init__(self, name, duration, world, rank, local, torch, channels=None, iq=None, bursts=None, shard=None):
        from dumpvdl2_amd import synth, workloads, vdl2hip
        from dumpvdl2_amd import dist as vdist
        self.torch, self.vdist, self.vdl2hip = torch, vdist, vdl2hip
        self.name, self.world, self.rank, self.local = name, world, rank, local
        cfg = getattr(workloads, name)(duration)
        if name == "config2" and channels and channels != 8:
            cfg.freqs = synth.channel_plan(channels, cfg.centerfreq, max(8000, min(100000, 2000000 // channels)))
        self.cfg = cfg
        t0 = time.time()
        # every rank synthesises the capture itself (seeded: identical bytes, checked below) - each needs it in its own
        # page-locked memory for the striped ingest, and nothing large has to cross process boundaries
        if iq is None:
            iq, bursts = synth.synthesize(cfg)
        self.iq, self.bursts = iq, bursts
        self.t_synth = time.time() - t0
        self.nvals = self.iq.size
        self.nbytes = self.nvals * 2
        self.nsamples = self.nvals // 2
        self.C = len(cfg.freqs)
        # shard = (first, count): a rank-sized receiver on this GPU (projected_scaling); else the rank's own share
        self.first, self.count = shard if shard else vdist.shard_channels(self.C, world, rank)
        self.device = torch.device("cuda", local)
        # A receiver of few channels (a rank's share, config2, config3) has a short front: its step would be the walk's chain, and the
        # library puts the referee's scans AHEAD of the walk for receivers of <= 64 channels (vdl2hip_create).  With 256 channels
        # they cost more than they save and stay behind the walk.  This is the library's own choice (VDL2HIP_REF_PRESCAN overrides it);
        # the field below only reports it.
        env = os.environ.get("VDL2HIP_REF_PRESCAN")
        self.prescan = (env == "1") if env is not None else self.count <= 64
        # N > 1: a rank of few channels is handed the exchanged blocks TWO per feed in the timed loops (dist.ShardedFeeder, pairs): the chain
        # walk - scans - check is a cost per feed, and a rank's front is shorter than it (projected_scaling.two_blocks_per_feed measures the
        # same on one GPU).  VDL2_BENCH_PAIR=0 / 1 overrides.
        envp = os.environ.get("VDL2_BENCH_PAIR")
        self.pair = world > 1 and not shard and ((envp == "1") if envp is not None else self.count <= 64)
        self.rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm,
                                   device=local, max_block_bytes=self.nbytes * (2 if self.pair else 1), chan_first=self.first, chan_count=self.count)
        self.front = torch.cuda.ExternalStream(self.rx.stream(), device=self.device)
        self.host = torch.from_numpy(self.iq)

    def feeder(self, mode, source):
        return self.vdist.ShardedFeeder(self.rx, self.host, self.world, self.rank, mode=mode, source=source, device=self.device,
                                        front_stream=self.front, pairs=self.pair)

    def frames_of_step(self, feeder):
        """one synchronous step: every frame this rank's channels produce for one block"""
        self.rx.set_drain_lag(0)
        n, recs, octs = feeder.step()
        return self.vdl2hip.Receiver.unpack(n, recs, octs)

    def timed(self, feeder, steps, dist, repeats=1):
        """`repeats` times exactly `steps` steps in streaming mode (six blocks in flight: VDL2HIP_MAX_DRAIN_LAG + 1); every block fully delivered inside
        each timed region.  Returns the median repeat plus the list."""
        # the shader clock idles at ~100 MHz and needs some tens of ms of load to come up (profiles/r03_clocks_under_load.txt): after the CPU-side
        # pauses of this script (oracle gate, set-up of a receiver) a short region would otherwise start on a cold clock.  Untimed steps.
        t0 = time.perf_counter()
        while True:
            for _ in range(8):
                feeder.step()
            el = time.perf_counter() - t0
            if self.world > 1:         # every rank must do the same number of steps (each one is a collective)
                t = self.torch.tensor([el], dtype=self.torch.float64, device=self.device)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                el = float(t.item())
            if el >= CLOCK_WARMUP_S:
                break
        runs = [self._timed_once(feeder, steps, dist) for _ in range(max(1, repeats))]
        runs_sorted = sorted(runs, key=lambda r: r["dt"])
        med = dict(runs_sorted[len(runs_sorted) // 2])
        med["all_ms_per_step"] = [round(r["dt"] / steps * 1e3, 4) for r in runs]
        med["min_ms_per_step"] = round(runs_sorted[0]["dt"] / steps * 1e3, 4)
        med["k1_ms"] = statistics.median(r["k1_ms"] for r in runs)
        if "rank_dt" in runs[0]:
            med["rank_ms_per_step"] = [round(x / steps * 1e3, 4) for x in med["rank_dt"]]
        return med

    def _timed_once(self, feeder, steps, dist):
        torch = self.torch
        self.rx.set_profiling(1)           # start/stop events on the channeliser launch only: its duration is the roofline figure
        self.rx.set_drain_lag(self.vdl2hip.MAX_DRAIN_LAG)
        s0 = self.rx.stats()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nframes = 0
        for _ in range(steps):
            nframes += feeder.step(pair=self.pair)[0]
        if self.pair:
            feeder.flush()                  # (an odd K: the last block has no partner)
        self.rx.set_drain_lag(0)
        nframes += self.rx.drain_packed()[0]
        feeder.finish()
        torch.cuda.synchronize()
        t_own = time.perf_counter() - t0
        if self.world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        s1 = self.rx.stats()
        out = {}
        if self.world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            nf = torch.tensor([nframes], dtype=torch.int64, device=self.device)
            dist.all_reduce(nf)
            nframes = int(nf.item())
            mine = torch.tensor([t_own], dtype=torch.float64, device=self.device)
            every = [torch.zeros_like(mine) for _ in range(self.world)]
            dist.all_gather(every, mine)
            out["rank_dt"] = [float(x.item()) for x in every]      # each rank's own time to its last delivered frame (before the barrier)
        launches = max(1, s1["chanfir_launches"] - s0["chanfir_launches"])
        k1_ms = (s1["chanfir_ms"] - s0["chanfir_ms"]) / launches
        k1_cs = (s1["chan_samples"] - s0["chan_samples"]) / launches
        assert s1["overflow_feeds"] == s0["overflow_feeds"], "device-side buffer overflow"
        out.update({"dt": dt, "frames": nframes, "k1_ms": k1_ms, "k1_chan_samples": k1_cs,
                    # channeliser workgroups that stopped waiting for their predecessor and recomputed its state (0 with one process per GPU)
                    "lookback_fallbacks": s1["front_sync_timeouts"] - s0["front_sync_timeouts"],
                    # blocks fed to an idle receiver from page-locked memory (the first of a host-fed region): channelised in pieces behind the
                    # pieces of their copy; their channeliser launches are not part of k1_ms
                    "cold_start_feeds": s1.get("cold_start_feeds", 0) - s0.get("cold_start_feeds", 0),
                    "seg_adopted": (s1["seg_adopted"] - s0["seg_adopted"]) / steps, "seg_walked": (s1["seg_walked"] - s0["seg_walked"]) / steps})
        return out

    def stage_times(self, feeder, nstage=4):
        """per-stage kernel times (informational): a short untimed pass with every stage's launch timed"""
        self.rx.set_profiling(2)
        self.rx.set_drain_lag(self.vdl2hip.MAX_DRAIN_LAG)
        sa = self.rx.stats()
        for _ in range(nstage):
            feeder.step()
        self.rx.set_drain_lag(0)
        self.rx.drain_packed()
        feeder.finish()
        self.torch.cuda.synchronize()
        sb = self.rx.stats()
        self.rx.set_profiling(1)
        return {k: round((sb[k] - sa[k]) / nstage, 4) for k in ("chanfir_ms", "phase_ms", "sync_ms", "walk_ms", "nf_ms", "burst_ms")}

    def close(self):
        self.rx.close()


# ---------------------------------------------------------------------------------------------------------------------
def roofline_of(t, traffic):
    """The channeliser k_chanfir against what bounds it.  It is bound by VALU instruction issue, not by HBM: the block
    is fetched once per XCD and shared by the channels through LDS and L2, so that side leads; SURVEY 8.5's algorithmic-bytes
    figure (which charges the block once per channel and therefore exceeds the HBM peak by construction) and the physical HBM
    traffic of the same launch follow."""
    k1_ms, cs = t["k1_ms"], t["k1_chan_samples"]
    sec = k1_ms * 1e-3
    tf = cs * FLOP_PER_CHAN_SAMPLE / sec / 1e12 if k1_ms > 0 else 0.0
    ach = cs * ALGO_BYTES / sec / 1e9 if k1_ms > 0 else 0.0
    tr_bytes = traffic["traffic_bytes"] if traffic else None
    out = {"bound": "valu", "kernel": "k_chanfir", "achieved": round(tf, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": round(tf / VALU_PEAK_TFLOPS, 4), "traffic": tr_bytes,
           "flop_per_chan_sample": FLOP_PER_CHAN_SAMPLE, "avg_launch_ms": round(k1_ms, 5), "chan_samples_per_launch": cs,
           # the rates that bound the sample loop, measured in THIS run right after the timed regions (measured_rates(): the library's
           # vdl2hip_debug_ubench micro-kernels, chip at its working clock); None if the library does not export the hook
           "issue_rate_ceiling": None, "co_bound": None,
           "hbm_algorithmic": {"achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                               "bytes_per_chan_sample": ALGO_BYTES, "bytes_per_launch": cs * ALGO_BYTES,
                               "note": "SURVEY 8.5's accounting: the cs16 block charged once per channel (the reference's access pattern) + one "
                                       "float2 per decimated output.  NOT a physical bandwidth: the kernel fetches the block once per XCD, so "
                                       "this exceeds the HBM peak by construction"},
           "hbm_physical": None, "traffic_source": None}
    if tr_bytes:
        out["hbm_physical"] = {"achieved": round(tr_bytes / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(tr_bytes / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic_over_algorithmic": round(tr_bytes / (cs * ALGO_BYTES), 4)}
        out["traffic_source"] = traffic["source"]
    return out


def measured_rates(vh, device):
    """issue_rate_ceiling / co_bound of the roofline line from micro-kernels run in this process (csrc/ubench.inc, ~0.1 s): back-to-back
    FMAs, a random 16-byte LDS wave-gather, and the channeliser's sample loop in miniature with and without its gather"""
    import ctypes as C
    L = vh.load_library()
    if not hasattr(L, "vdl2hip_debug_ubench"):
        return None, None
    a = (C.c_double * 8)()
    L.vdl2hip_debug_ubench.argtypes = [C.c_int, C.POINTER(C.c_double)]
    if L.vdl2hip_debug_ubench(device, a) != 0:
        return None, None
    ghz = a[2] / 1e3
    issue = {"v_fma_f32": round(a[0], 1), "v_pk_fma_f32": round(a[1], 1), "unit": "TFLOP/s", "shader_clock_GHz": round(ghz, 3),
             "note": "back-to-back independent FMAs, 4 waves per SIMD, measured in this run (vdl2hip_debug_ubench, " + f"{a[7]:.0f} ms)"}
    co = {"pipe": "LDS gather", "clocks_per_random_16B_wave_gather_per_CU": round(a[3], 2),
          # one gather per channel-sample and wavefront, four SIMDs per LDS pipe: what the pipe allows each SIMD
          "lds_bound_clocks_per_chan_sample_wave_per_SIMD": round(4 * a[3], 1),
          "compute_units": int(a[6]),
          "note": "k_chanfir needs one LUT gather (ds_read_b128 at 64 unrelated entries) per channel-sample and wavefront; the CU's one LDS pipe serves "
                  "four SIMDs.  Both figures measured in this run (vdl2hip_debug_ubench); `k_chanfir_clocks_per_chan_sample_wave_per_SIMD` is the "
                  "kernel's own time in the same unit (all of it: sample loop, wave scan, staging, look-back).  The pipe runs level with VALU "
                  "issue but is not what binds: a build without any per-sample gather (round 4, profiles/r04_k1_table_free_nco_ab.txt) is exactly as fast"}
    return issue, co


def pmc_traffic(workload, case):
    """HBM traffic of K1 per launch.  PMC counters cannot be read from inside this process: the number is the one rocprofv3
    measured on this same command (dev/gpu_pmc_traffic.sh; summary under profiles/), a CONSTANT FROM A FILE, quoted only for
    the workload and shard size it was taken on - `traffic_source` in the line says so."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pt = json.load(f)
        for e in pt.get("k_chanfir_by_workload", []):
            w = e["workload"]
            if (w["name"], w["channels_per_gpu"], float(w["duration_s"])) == (workload, case.count, float(case.cfg.duration_s)):
                return {"traffic_bytes": e["traffic_bytes"],
                        "source": f"profiles/pmc_traffic.json <- {e.get('source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE')}: measured earlier with "
                                  f"rocprofv3 on this command, not in this run"}
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return None


def cpu_baseline_of(case, po, raw, first_pass_s, full=True):
    """CPU baseline on this box's host cores: the oracle restatement with the REFERENCE'S OWN THREADING (vdl2o_run: a persistent
    thread per channel + the producer, two barriers per 320 000-byte block, serial sample conversion - dumpvdl2.c:117-135,
    demod.c:300-301,356-365), best of 3 passes over the same block (the first pass is the parity gate's).  Beside it: the old
    spawn-threads-per-block figure, a work-queue variant (what a CPU implementation free to restructure would do) and the
    -O3 -ffast-math build upstream ships."""
    cfg = case.cfg

    def one(variant, how, **kw):
        o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm, variant=variant)
        t0 = time.perf_counter()
        if how == "spawn":
            o.process(raw, block_bytes=320000, nthreads=kw["nthreads"])
        else:
            o.run(raw, block_bytes=kw.get("block", 320000), mode=kw["mode"], nthreads=kw.get("nthreads", 0))
        dt = time.perf_counter() - t0
        o.close()
        return dt

    ncpu = cpus_available()
    ref = [first_pass_s] + ([one("strict", "run", mode=po.RUN_THREAD_PER_CHANNEL) for _ in range(2)] if full else [])
    best = min(ref)
    mss = lambda s: round(case.nsamples / s / 1e6, 4)       # noqa: E731
    nspt = lambda s, thr: round(s * min(thr, ncpu) / (case.nsamples * case.C) * 1e9, 2)     # noqa: E731  ns of one CPU per channel-sample
    out = {"value": mss(best), "unit": "MS/s", "cores": min(case.C + 1, ncpu), "threads": case.C + 1, "kind": "port",
           "threading": "reference: one persistent thread per channel + the producer, two pthread barriers of count N+1 per block, "
                        "serial sample conversion on the producer (dumpvdl2.c:117-135, demod.c:300-301,356-365)",
           "channel_MS_per_s": round(case.nsamples * case.C / best / 1e6, 1),
           "ns_per_channel_sample_per_cpu": nspt(best, case.C + 1),
           "passes_s": [round(x, 3) for x in ref], "best_of": len(ref),
           "sample": f"the same {cfg.duration_s:g} s x {case.C}-channel block, whole, per pass; CPU restatement of the reference (oracle/), "
                     f"320000-byte blocks as process_iq_file()",
           "flags": "-O2 -fno-fast-math -ffp-contract=off (oracle/Makefile)", **cpu_info()}
    out["note"] = (f"the container may use {ncpu} CPUs' worth of time (cgroup quota / affinity) of the {os.cpu_count()} the host shows; the reference's "
                   f"model needs {case.C + 1} runnable threads whatever the host offers, the work-queue figure uses {ncpu}")
    if not full:
        return out
    try:
        wq = [one("strict", "run", mode=po.RUN_WORKQUEUE, nthreads=ncpu, block=1 << 22) for _ in range(2)]
        # the figure to divide by for a GPU/CPU ratio: the best this CPU does with the same arithmetic, however it is threaded
        out["best_value"] = max(out["value"], mss(min(wq)))
        out["best_value_is"] = "the faster of the reference threading (`value`) and the work-queue variant: the CPU figure meant for speed-up ratios"
        out["workqueue"] = {"value": mss(min(wq)), "unit": "MS/s", "threads": ncpu, "block_bytes": 1 << 22, "passes_s": [round(x, 3) for x in wq],
                            "ns_per_channel_sample_per_cpu": nspt(min(wq), ncpu),
                            "note": "best-effort CPU: one persistent worker per available CPU, conversion spread over them, channels handed out "
                                    "dynamically, 4 MiB blocks"}
        s = one("strict", "spawn", nthreads=min(case.C, os.cpu_count() or 1))
        out["spawn_per_block"] = {"value": mss(s), "unit": "MS/s", "note": "round 2's figure: pthread_create/join of one thread per channel on EVERY block"}
        f = one("fast", "run", mode=po.RUN_WORKQUEUE, nthreads=ncpu, block=1 << 22)
        out["fast_math"] = {"value": mss(f), "unit": "MS/s", "flags": "-O3 -ffast-math (mirrors src/CMakeLists.txt:35-38)", "threading": "workqueue",
                            "threads": ncpu, "ns_per_channel_sample_per_cpu": nspt(f, ncpu)}
    except Exception as e:  # noqa: BLE001 - the extra variants are informational
        out["variants_error"] = str(e)[:200]
    return out


def oracle_gate(case, frames, po, label, strict_counters=True):
    """The oracle on the WHOLE block (all channels, all 16 s) on this box's host cores, reference threading; returns
    (verified dict, seconds of that pass)."""
    from util import truth_is_subset, compare_at_full_size, TOL_DB, TOL_PPM
    cfg = case.cfg
    missing = truth_is_subset(case.bursts, frames)
    want = sum(len(b.frames) for b in case.bursts if b.decodable)
    # with injected errors some bursts pushed past the nominal RS capacity still decode (both here and in the oracle)
    exact = not cfg.error_injection
    assert missing == 0 and (len(frames) == want if exact else len(frames) >= want), \
        f"{label}: {missing} transmitted frames missing, {len(frames)} decoded vs {want} sent"
    o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
    t0 = time.perf_counter()
    o.run(case.iq.view(np.uint8), block_bytes=320000, mode=po.RUN_THREAD_PER_CHANNEL)
    tc = time.perf_counter() - t0
    ofr = o.frames()
    cmp = compare_at_full_size(ofr, frames, label=label)            # (strict: burst timing identical, no tie allowances - the referee)
    # the reference's own 18 statsd counters per channel, identical on every channel (the last two of the 20 are this repo's
    # diagnostics: preambles dropped by --max-ppm and out-of-range slicer indices)
    from util import compare_reference_counters
    nref = 18
    names = list(o.counters(case.first).keys())
    co = [list(o.counters(ch).values()) for ch in range(case.first, case.first + case.count)]
    cg = [list(case.rx.counters(ch).values()) for ch in range(case.first, case.first + case.count)]
    which, nref_diff = compare_reference_counters(names, co, cg, label=label, strict=strict_counters, nref=nref)
    ndiff = sum(a[nref:] != b[nref:] for a, b in zip(co, cg))
    o.close()
    ties, nft = cmp["timing_ties"], cmp["nf_update_ties"]
    return {"tx_frames": want, "decoded": len(frames), "oracle_window_s": cfg.duration_s, "oracle_frames": len(ofr),
            # compare_at_full_size() has asserted octets, frame order and integer metadata of every frame; compare_reference_counters() the counters
            "frames_and_integer_metadata_identical": cmp["frames"] == len(ofr) == len(frames),
            "reference_counters": "identical on every channel (asserted)",
            # ... burst timing identical and the float metadata within SURVEY 8.5's tolerances on every frame (asserted: no tie allowances)
            "oracle_identical": ties == 0 and nft == 0 and nref_diff == 0,
            "oracle_parity_within_tolerance": True,
            "timing_ties": ties, "nf_update_ties": nft,
            "tolerances": {"frame_pwr_db": TOL_DB, "nf_pwr_db": TOL_DB, "ppm": TOL_PPM, "burst timing": "identical", "max_tie_fraction": 0},
            "max_abs_diff": cmp["max_abs_diff"],
            "referee": {k[8:]: v for k, v in case.rx.stats().items() if k.startswith("referee_")},
            "channels": case.count, "channels_with_reference_counters_identical": case.count - nref_diff, "channels_with_diagnostic_counter_diff": int(ndiff),
            "reference_counter_differences": which or None,
            "channels_with_frames": len({f["chan"] for f in frames})}, tc, OracleRef(ofr, names, co)


class OracleRef(list):
    """the oracle's frames of a whole block (a list, as before) plus its counters per channel of the receiver's share"""

    def __init__(self, frames, names, counters):
        super().__init__(frames)
        self.names, self.counters = names, counters


def pieces_gate(case, oref, label, seed=6):
    """The parity gate again with the block fed in 5-8 LONG pieces (each several walk segments), six in flight, drained as they complete:
    the path a streaming caller takes - feed i + 1's front beside feed i's walk, check and burst decoder, state carried from feed to
    feed, a channel walked again while the previous feed's burst decoder still runs (round 5's red test).  A fresh receiver (its counters
    start at zero); frames, burst timing, integer metadata and the reference's 18 counters must be the oracle's, as for the whole block."""
    from util import compare_at_full_size, compare_reference_counters
    vh = case.vdl2hip
    cfg = case.cfg
    rx = vh.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vh.FMT_S16LE, cfg.rx_max_ppm, device=case.local,
                     max_block_bytes=case.nbytes // 4, chan_first=case.first, chan_count=case.count)
    try:
        rng = np.random.default_rng(seed)
        npieces = int(rng.integers(5, 9))
        cuts = np.sort(rng.uniform(0.35, 1.0, npieces)); cuts = np.cumsum(cuts / cuts.sum())
        edges = [0] + [int(case.nsamples * c) for c in cuts[:-1]] + [case.nsamples]
        raw = case.iq.view(np.uint8)
        rx.set_drain_lag(vh.MAX_DRAIN_LAG)
        got = []
        for a, b in zip(edges[:-1], edges[1:]):
            rx.feed(raw[4 * a:4 * b])
            got += rx.drain()
        rx.set_drain_lag(0)
        got += rx.drain()
        mine = [f for f in oref if case.first <= f["chan"] < case.first + case.count]
        cmp = compare_at_full_size(mine, got, label=label)
        cg = [list(rx.counters(ch).values()) for ch in range(case.first, case.first + case.count)]
        which, nref_diff = compare_reference_counters(oref.names, oref.counters, cg, label=label, strict=True, nref=18)
        st = rx.stats()
        return {"pieces": npieces, "samples_per_piece": [b - a for a, b in zip(edges[:-1], edges[1:])], "in_flight": vh.MAX_DRAIN_LAG + 1,
                "frames": len(got), "oracle_frames": len(mine), "timing_ties": cmp["timing_ties"], "nf_update_ties": cmp["nf_update_ties"],
                "channels": case.count, "channels_with_reference_counters_identical": case.count - nref_diff,
                "oracle_identical": cmp["frames"] == len(mine) == len(got) and cmp["timing_ties"] == 0 and cmp["nf_update_ties"] == 0 and nref_diff == 0,
                "max_abs_diff": cmp["max_abs_diff"],
                "referee": {k[8:]: v for k, v in st.items() if k.startswith("referee_")}}
    finally:
        rx.close()


def measure_secondary(c2, name, oracle_check, args, dist, po):
    """one more configuration, measured like the headline one (one repeat): host-fed and HBM-resident K steps, per-stage kernel
    times, every transmitted frame recovered, optionally the oracle on the whole block"""
    from util import truth_is_subset
    shard = c2.count != c2.C
    fh = c2.feeder("broadcast", "host")
    fr2 = c2.frames_of_step(fh)
    mine = [b for b in c2.bursts if c2.first <= b.chan < c2.first + c2.count]
    miss = truth_is_subset(mine, fr2)
    want2 = sum(len(b.frames) for b in mine if b.decodable)
    # (with injected errors some bursts pushed past the nominal RS capacity still decode, here and in the oracle)
    assert miss == 0 and (len(fr2) >= want2 if c2.cfg.error_injection else len(fr2) == want2), f"{name}: {miss} transmitted frames missing, {len(fr2)} decoded vs {want2} sent"
    ver2 = pcs2 = None
    if oracle_check:
        ver2, _, oref2 = oracle_gate(c2, fr2, po, f"{name} oracle gate", strict_counters=True)
        pcs2 = pieces_gate(c2, oref2, f"{name} oracle gate, in pieces")
    fh.step(); c2.rx.sync()
    th = c2.timed(fh, args.steps, dist, 1)
    del fh
    fd = c2.feeder("broadcast", "hbm")
    fd.step(); fd.step(); c2.rx.set_drain_lag(0); c2.rx.drain_packed()
    td = c2.timed(fd, args.steps, dist, 1)
    st = c2.stage_times(fd)
    del fd
    rl = roofline_of(td, pmc_traffic(name, c2))
    step_hbm = td["dt"] / args.steps * 1e3
    return {"name": name + ("_shard" if shard else ""), "referee_scans_ahead_of_the_walk": c2.prescan,
            "workload": (f"configs[{WORKLOAD_INDEX[name]}] ({name})" if name in WORKLOAD_INDEX else name)
                        + f": {c2.C} channels in the air, {c2.count} decoded here"
                        + (f" (channels {c2.first}..{c2.first + c2.count - 1}: a rank's share at N = 8)" if shard else "") + f", {c2.cfg.duration_s:g} s",
            "value": round(c2.nsamples * args.steps / th["dt"] / 1e6, 3), "value_hbm_resident": round(c2.nsamples * args.steps / td["dt"] / 1e6, 3),
            "ms_per_step": round(th["dt"] / args.steps * 1e3, 4), "ms_per_step_hbm_resident": round(step_hbm, 4),
            "frames_per_step": th["frames"] / args.steps, "tx_frames_all_recovered": True, "verified": ver2, "verified_in_pieces": pcs2,
            "stage_ms_per_step": st,
            # is the burst-rate back end (walk, noise floor, burst decoder: own streams) hidden behind the sample-rate front?
            "front_ms": round(st["chanfir_ms"] + st["sync_ms"], 4), "step_minus_front_ms": round(step_hbm - st["chanfir_ms"] - st["sync_ms"], 4),
            "k_chanfir_ms": rl["avg_launch_ms"], "valu_frac": rl["frac"], "hbm_algorithmic_frac": rl["hbm_algorithmic"]["frac"]}


def group_from_c(case, args, torch, local, members=8):
    """vdl2hip_group_* (the multi-GPU path from plain C) with `members` virtual shards on this GPU: every transmitted frame recovered
    in both exchange forms, K timed steps each (six blocks in flight, frames drained without a callback)"""
    from util import truth_is_subset
    vh = case.vdl2hip
    cfg = case.cfg
    pin = case.host.pin_memory()
    out = {"members": members, "devices": [local] * members, "source": "page-locked host memory (vdl2hip_group_feed_pinned)", "forms": {},
           "note": "ONE GPU does the work of all members here (each decodes its share of the channels of every block), so ms_per_step is to be read "
                   "against t_all_channels_ms, not against t_rank_ms_max: it bounds what the C path adds when everything shares a device - 8 receivers' "
                   "48 streams on one GPU's hardware queues, 8 x ~10 launches per block from one host thread, the stripes' H2D copies and the "
                   "same-device copies that stand in for xGMI - and proves both exchange forms end to end; it is not a projection of 8 GPUs"}
    g = vh.ReceiverGroup(cfg.centerfreq, list(cfg.freqs), [local] * members, cfg.oversample, vh.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=case.nbytes)
    try:
        for form in ("allgather", "broadcast"):
            g.set_exchange(form)
            g.set_drain_lag(0)
            g.feed_pinned(pin.data_ptr(), case.nbytes)
            fr = g.drain()
            want = sum(len(b.frames) for b in case.bursts if b.decodable)
            assert truth_is_subset(case.bursts, fr) == 0 and (len(fr) >= want if cfg.error_injection else len(fr) == want), f"group {form}: frames missing"
            g.set_drain_lag(vh.MAX_DRAIN_LAG)
            for _ in range(3):
                g.feed_pinned(pin.data_ptr(), case.nbytes); g.drain_count()
            g.set_drain_lag(0); g.drain_count(); g.sync()
            g.set_drain_lag(vh.MAX_DRAIN_LAG)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            t_call = 0.0                                   # host time inside vdl2hip_group_feed_pinned() alone: it only queues work (8 members x ~14 launches + the exchange)
            for _ in range(args.steps):
                tc = time.perf_counter()
                g.feed_pinned(pin.data_ptr(), case.nbytes)
                t_call += time.perf_counter() - tc
                n += g.drain_count()
            g.set_drain_lag(0)
            n += g.drain_count()
            g.sync()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out["forms"][form] = {"ms_per_step": round(dt / args.steps * 1e3, 4), "value": round(case.nsamples * args.steps / dt / 1e6, 3),
                                  "frames_per_step": n / args.steps, "ran_as": g.exchange(), "host_ms_in_feed_call": round(t_call / args.steps * 1e3, 4)}
    finally:
        g.close()
        del pin
    return out


def dropin_block_rate(case, torch, seconds=2.0, block=320000):
    """The drop-in way of using the library (csrc/dropin.c under an unmodified dumpvdl2 --iq-file): the reference's own block size
    (FILE_BUFSIZE = 320 000 bytes = 80 000 samples, dumpvdl2.h:48) from pageable memory, one vdl2hip_feed() + drain per block.  Lag 0 is
    the blocking process_buf_*() semantics (every frame of a block delivered before the next block is handed over); lag 1 is what the
    adapter's split gives a file reader (the first channel's thread delivers block i while main() reads and hands over block i+1)."""
    vh = case.vdl2hip
    cfg = case.cfg
    raw = case.iq.view(np.uint8)[: int(seconds * 2100000) * 4]
    nblk = (raw.size + block - 1) // block
    out = {"workload": f"dropin_320kB_block: {case.C} channels, {seconds:g} s of the same capture in {nblk} blocks of {block} bytes from pageable memory, "
                       f"one feed + drain per block (lag0, lag1) or per 16 collected blocks (the adapter's default for a file)", "block_bytes": block, "blocks": nblk}
    # (blocks per feed, drain lag): the block on its own, blocking / one block late; and what csrc/dropin.c and tools/vdl2hip_iqfile do
    # by default for a producer that comes straight back (a file): 16 blocks collected per feed, the feed two before it delivered meanwhile
    for per_feed, lag, key in ((1, 0, "lag0"), (1, 1, "lag1"), (16, 2, "collected16_lag2")):
        piece = per_feed * block
        if per_feed > 1:                                   # (enough feeds for the rate to mean something: 8 s = 210 blocks = 14 feeds)
            raw = case.iq.view(np.uint8)[: int(4 * seconds * 2100000) * 4]
            nblk = (raw.size + block - 1) // block
        rx = vh.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vh.FMT_S16LE, cfg.rx_max_ppm, device=case.local, max_block_bytes=piece)
        rx.set_drain_lag(lag)
        for k in range(0, min(raw.size, max(20 * block, 4 * piece)), piece):          # warm
            rx.feed(raw[k:k + piece]); rx.drain_packed()
        rx.set_drain_lag(0); rx.drain_packed(); rx.set_drain_lag(lag)
        n = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(0, raw.size, piece):
            rx.feed(raw[k:k + piece])
            n += rx.drain_packed()[0]
        rx.set_drain_lag(0); n += rx.drain_packed()[0]
        dt = time.perf_counter() - t0
        rx.close()
        out[key] = {"blocks_per_feed": per_feed, "blocks": nblk, "ms_per_block": round(dt / nblk * 1e3, 4), "value": round(raw.size / 4 / dt / 1e6, 2), "x_real_time": round(raw.size / 4 / dt / 2.1e6, 1), "frames": n}
    out["note"] = "MS/s of IQ with all channels decoded; the 16 s blocks of the headline are what the throughput metric wants, this is what a live receiver or an unmodified --iq-file run sees"
    return out


def rank_two_blocks_per_feed(case, per, torch, local, steps, t_all_ms):
    """rank-sized receivers (ranks 0, 3, 7 of 8) fed TWO 16 s blocks per feed, block resident, six feeds in flight: ms per 16 s block"""
    vh = case.vdl2hip
    cfg = case.cfg
    dev2 = torch.from_numpy(np.concatenate([case.iq, case.iq])).to(f"cuda:{local}")
    nbytes2 = 2 * case.nbytes
    rows = []
    try:
        for r in (0, 3, 7):
            rx = vh.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vh.FMT_S16LE, cfg.rx_max_ppm, device=local, max_block_bytes=nbytes2, chan_first=r * per, chan_count=per)
            try:
                rx.set_drain_lag(vh.MAX_DRAIN_LAG)
                for _ in range(4):
                    rx.feed_device(dev2.data_ptr(), nbytes2); rx.drain_packed()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(steps):
                    rx.feed_device(dev2.data_ptr(), nbytes2); rx.drain_packed()
                rx.set_drain_lag(0); rx.drain_packed(); torch.cuda.synchronize()
                rows.append({"rank": r, "ms_per_16s_block": round((time.perf_counter() - t0) / steps / 2 * 1e3, 4)})
            finally:
                rx.close()
    finally:
        del dev2
    worst = max(x["ms_per_16s_block"] for x in rows)
    return {"what": f"the same rank-sized receivers fed TWO 16 s blocks per feed ({steps} feeds in one timed region): the fixed chain walk - scans - check of a feed is paid "
                    "once for 32 s of signal; what pairing the exchange's buffers would buy a rank (results one block later)",
            "shards": rows, "t_rank_ms_per_block_max": worst, "t_all_channels_ms": round(t_all_ms, 4), "compute_ceiling_speedup_at_8": round(t_all_ms / worst, 3),
            "note": "t_all_channels_ms is the 256-channel receiver on ONE block per feed (its front hides the chain: two per feed change nothing for it)"}


def h2d_ms(torch, host_pinned, device, iters=3):
    """one plain H2D copy of the block from page-locked memory: this rank's PCIe link, nothing else running"""
    dst = torch.empty_like(host_pinned, device=device)
    dst.copy_(host_pinned, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        dst.copy_(host_pinned, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=3, help="how many times the K timed steps are run (median reported)")
    ap.add_argument("--duration", type=float, default=16.0, help="seconds of 2.1 MS/s signal per step")
    ap.add_argument("--channels", type=int, default=8, help="channel count of config2 (experiments)")
    ap.add_argument("--workload", default="config4", choices=["config2", "config3", "config4", "config5", "config4_bursty"],
                    help="BASELINE configs[1..4]; default config4 = north_star's 256 channels")
    ap.add_argument("--exchange", default="auto", choices=["auto", "broadcast", "allgather"],
                    help="N>1: how each rank gets the raw IQ block (both are always measured; this picks `value`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N=1: skip the other configurations and the projected-scaling runs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from dumpvdl2_amd import vdl2hip  # noqa: F401
    from dumpvdl2_amd import dist as vdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal (tests/test_bench_rehearsal.py): the N > 1 control flow on a one-GPU box - every rank on device 0, gloo instead
    # of RCCL as the transport.  Never set by the driver; the numbers of such a run mean nothing.
    rehearsal = os.environ.get("VDL2_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
        # (several PROCESSES time-slicing one GPU: here the channeliser's look-back between workgroups does time out now and then and
        # takes its fall-back - config.lookback_fallbacks_per_step - which the parity gate below then covers)
    assert world == args.gpus or world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the library has no CPU path)"
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    case = Case(args.workload, args.duration, world, rank, local, torch, channels=args.channels)
    cfg = case.cfg
    if world > 1:        # every rank must hold the very same capture
        h = torch.tensor([int(np.bitwise_xor.reduce(case.iq.view(np.uint64))) & 0x7fffffffffffffff], dtype=torch.int64, device=case.device)
        lo, hi = h.clone(), h.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert int(lo.item()) == int(hi.item()), "ranks synthesised different captures"

    # ---- exchange: both forms measured bare, from both sources ----
    exchange_info = None
    mode = "broadcast"
    forms = ["broadcast"]
    if world > 1:
        forms = ["broadcast", "allgather"] if case.nbytes % world == 0 else ["broadcast"]
        exchange_info = {"uses_rccl": not rehearsal, "backend": dist.get_backend(), "nranks": dist.get_world_size()}
        try:       # which collective library this is, and that the ranks sit on different GPUs (the first N > 1 run is the proof of both)
            exchange_info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if not rehearsal else None
        except Exception as e:     # noqa: BLE001 - a missing version query must not stop a benchmark
            exchange_info["rccl_version"] = f"unavailable ({type(e).__name__})"
        try:
            prop = torch.cuda.get_device_properties(local)
            ident = f"{local}:{getattr(prop, 'uuid', '')}:{getattr(prop, 'pci_bus_id', '')}"
        except Exception:          # noqa: BLE001
            ident = str(local)
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        exchange_info["devices_by_rank"] = idents
        exchange_info["distinct_devices"] = len(set(idents))
        for src_kind in ("host", "hbm"):
            for m in forms:
                secs, ok = vdist.time_exchange(case.host, world, rank, m, src_kind, case.device)
                exchange_info[f"{m}_{src_kind}_ms"] = round(secs * 1e3, 4) if ok else None
        ok_forms = [m for m in forms if exchange_info.get(f"{m}_host_ms") is not None]
        assert ok_forms, "no exchange form delivered the block intact"
        if args.exchange == "auto":
            mode = min(ok_forms, key=lambda m: exchange_info[f"{m}_host_ms"])
        else:
            mode = args.exchange
            assert mode in ok_forms, f"--exchange {mode} is not available for this block length / world size"
        forms = ok_forms
        exchange_info["chosen"] = mode
        exchange_info["chosen_by"] = "fastest bare exchange from host memory (auto)" if args.exchange == "auto" else "--exchange"
        pin = case.host.pin_memory()
        mine = torch.tensor([h2d_ms(torch, pin, case.device)], dtype=torch.float64, device=case.device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        exchange_info["h2d_whole_block_ms_by_rank"] = [round(float(x.item()), 4) for x in every]   # [0] = the broadcast form's ingest cost
        del pin

    # ---- warm-up, with the parity gate on the first pass ----
    f_host = case.feeder(mode, "host")
    verified = verified_pieces = None
    cpu_baseline = None
    oracle_frames = None
    fr = case.frames_of_step(f_host)
    allfr = vdist.gather_frames(fr, dst=0) if world > 1 else fr
    if rank == 0 and not args.no_verify:
        from oracle import pyoracle as po
        # N > 1: rank 0 holds the merged frames of all ranks - the frame / metadata comparison covers all channels, the counter
        # comparison rank 0's own (the other ranks' counters stay on their GPUs)
        verified, tc, oracle_frames = oracle_gate(case, allfr, po, "bench oracle gate" + (" (all ranks)" if world > 1 else ""))
        if world == 1:
            verified_pieces = pieces_gate(case, oracle_frames, "bench oracle gate, in pieces")
        if world == 1 and not args.no_cpu_baseline:
            cpu_baseline = cpu_baseline_of(case, po, case.iq.view(np.uint8), tc)
    if world > 1:
        dist.barrier()
    for _ in range(max(0, args.warmup - 1)):
        f_host.step()
    case.rx.sync()

    # ---- timed region: exactly K steps, host-fed, `repeats` times ----
    t_host = case.timed(f_host, args.steps, dist, args.repeats)
    stage_ms = case.stage_times(f_host)
    # every timed region above pays the pipeline's fill and drain once (first H2D exposed, the last block's back end after the last
    # front): a fixed 2-4 ms, i.e. 0.1-0.2 ms per step at K = 20.  One long region (10 K steps, at least 100) shows the streaming rate.
    steady = None
    if world == 1 and not args.no_secondary:
        ks = max(100, 10 * args.steps)
        ts = case.timed(f_host, ks, dist, 1)
        steady = {"steps": ks, "value": round(case.nsamples * ks / ts["dt"] / 1e6, 3), "ms_per_step": round(ts["dt"] / ks * 1e3, 4),
                  "k_chanfir_ms": round(ts["k1_ms"], 5),
                  "note": "host-fed, one timed region of this many steps: the fill and drain of the pipeline (six blocks deep) (paid once per "
                          "region, whatever its length) weigh a tenth of what they do in the K-step regions `value` comes from"}
    del f_host
    # ---- the same with the block resident in HBM ----
    f_hbm = case.feeder(mode, "hbm")
    for _ in range(2):
        f_hbm.step()
    case.rx.set_drain_lag(0); case.rx.drain_packed()
    rs0 = case.rx.stats()
    t_hbm = case.timed(f_hbm, args.steps, dist, args.repeats)
    rs1 = case.rx.stats()
    stage_ms_hbm = case.stage_times(f_hbm)
    t_hbm_steady = None
    if world == 1 and not args.no_secondary:      # (one long region, as `steady_state` above: what the projection below divides by)
        ks = max(100, 10 * args.steps)
        t_hbm_steady = case.timed(f_hbm, ks, dist, 1)["dt"] / ks * 1e3
    # ---- what the referee costs: the same timed region with it switched off (the answer is then no longer the oracle's on every input) ----
    referee_ab = None
    if world == 1 and not args.no_secondary:
        nf = max(1, rs1["feeds"] - rs0["feeds"])
        per = lambda k: round((rs1[k] - rs0[k]) / nf, 2)
        case.rx.debug_option("referee", 0)
        for _ in range(3):
            f_hbm.step()
        case.rx.set_drain_lag(0); case.rx.drain_packed()
        t_off = case.timed(f_hbm, args.steps, dist, 1)
        case.rx.debug_option("referee", 1)
        f_hbm.step(); case.rx.set_drain_lag(0); case.rx.drain_packed()
        on_ms, off_ms = t_hbm["dt"] / args.steps * 1e3, t_off["dt"] / args.steps * 1e3
        referee_ab = {"ms_per_step_hbm_resident": round(on_ms, 4), "ms_per_step_hbm_resident_referee_off": round(off_ms, 4), "cost_frac": round(on_ms / off_ms - 1.0, 4),
                      "k_chanfir_ms": round(t_hbm["k1_ms"], 4), "k_chanfir_ms_referee_off": round(t_off["k1_ms"], 4),
                      "scans_per_step": per("referee_scans"), "candidate_scans_per_step": per("referee_candidate_scans"), "header_scans_per_step": per("referee_header_scans"),
                      "symbol_scans_per_step": per("referee_symbol_scans"), "channels_walked_again_per_step": per("referee_rewalks"), "refused": rs1["referee_refused"] - rs0["referee_refused"],
                      "what": "decisions within the margin of the channeliser's distance from the reference's fp32 scan are taken on the reference's own samples, "
                              "recomputed sequentially from the raw input (DESIGN 5); `value` is measured with it on"}
    del f_hbm
    measured = measured_rates(vdl2hip, local) if rank == 0 else (None, None)     # right behind the timed regions: the clock is up

    # ---- N > 1: the other exchange form, demodulating, beside the chosen one ----
    by_exchange = None
    if world > 1:
        by_exchange = {mode: {"value": round(case.nsamples * args.steps / t_host["dt"] / 1e6, 3), "ms_per_step": round(t_host["dt"] / args.steps * 1e3, 4),
                              "ms_per_step_hbm_resident": round(t_hbm["dt"] / args.steps * 1e3, 4)}}
        for m in forms:
            if m == mode:
                continue
            fo = case.feeder(m, "host")
            fo.step(); fo.step(); case.rx.set_drain_lag(0); case.rx.drain_packed()
            to = case.timed(fo, args.steps, dist, args.repeats)
            del fo
            fo = case.feeder(m, "hbm")
            fo.step(); fo.step(); case.rx.set_drain_lag(0); case.rx.drain_packed()
            tr = case.timed(fo, args.steps, dist, args.repeats)
            del fo
            by_exchange[m] = {"value": round(case.nsamples * args.steps / to["dt"] / 1e6, 3), "ms_per_step": round(to["dt"] / args.steps * 1e3, 4),
                              "ms_per_step_hbm_resident": round(tr["dt"] / args.steps * 1e3, 4), "rank_ms_per_step": to.get("rank_ms_per_step")}

    # ---- N = 1: what one rank of the 8-GPU split would do on this GPU -> ceiling of the speed-up ----
    projected = None
    secondary = []
    if world == 1 and not args.no_secondary:
        # one receiver at a time from here on: the HIP streams of an idle second receiver share the few hardware queues with the
        # active one's (its front and walk streams then serialise: measured +0.5 ms per step on the rank-sized workload)
        case.close()
    try:
      if world == 1 and not args.no_secondary and case.C % 8 == 0 and case.C >= 64:
        per = case.C // 8
        shards = []
        for r in (0, 3, 7):
            cs = Case(args.workload, args.duration, 1, 0, local, torch, iq=case.iq, bursts=case.bursts, shard=(r * per, per))
            fd = cs.feeder("broadcast", "hbm")
            got = cs.frames_of_step(fd)
            mine = [b for b in case.bursts if r * per <= b.chan < (r + 1) * per]
            from util import truth_is_subset
            assert truth_is_subset(mine, got) == 0, f"shard {r}: transmitted frames missing"
            vs_oracle = None
            if oracle_frames is not None:      # ... and the oracle's frames of exactly these channels (the whole-block pass of the parity gate above)
                from util import compare_at_full_size
                c = compare_at_full_size([f for f in oracle_frames if r * per <= f["chan"] < (r + 1) * per], got, label=f"shard {r} vs oracle")
                vs_oracle = {"frames": c["frames"], "timing_ties": c["timing_ties"], "nf_update_ties": c["nf_update_ties"], "max_abs_diff": c["max_abs_diff"]}
            fd.step(); fd.step(); cs.rx.set_drain_lag(0); cs.rx.drain_packed()
            ts = cs.timed(fd, args.steps, dist, args.repeats)
            st = cs.stage_times(fd)
            ks = max(100, 10 * args.steps)
            tl = cs.timed(fd, ks, dist, 1)
            shards.append({"rank": r, "channels": [r * per, (r + 1) * per - 1], "referee_scans_ahead_of_the_walk": cs.prescan, "ms_per_step": round(ts["dt"] / args.steps * 1e3, 4),
                           "min_ms_per_step": ts["min_ms_per_step"], "steady_ms_per_step": round(tl["dt"] / ks * 1e3, 4), "steady_steps": ks,
                           "k_chanfir_ms": round(ts["k1_ms"], 4), "stage_ms_per_step": st,
                           "frames_identical_to_the_oracle": vs_oracle})
            del fd
            cs.close()
        t256 = t_hbm["dt"] / args.steps * 1e3
        t32 = max(s["ms_per_step"] for s in shards)
        pin = case.host.pin_memory()
        h2d = h2d_ms(torch, pin, case.device)
        del pin
        projected = {"what": f"rank-sized workload of the 8-GPU split on this GPU: {per} of the {case.C} channels of the same block, block resident in HBM "
                             f"(as an RCCL exchange leaves it), six blocks in flight, the same K steps x {args.repeats} repeats (median)",
                     "t_all_channels_ms": round(t256, 4), "t_rank_ms_max": round(t32, 4), "shards": shards,
                     "compute_ceiling_speedup_at_8": round(t256 / t32, 3),
                     # the same from one long timed region each (max(100, 10 K) steps): a K-step region pays a feed's way through the device - four
                     # to six fronts for a rank-sized receiver - once, which at K = 20 is a fifth of a rank's step and a thirtieth of all channels'
                     "steady": None if t_hbm_steady is None else {"t_all_channels_ms": round(t_hbm_steady, 4), "t_rank_ms_max": max(s["steady_ms_per_step"] for s in shards),
                                                                   "compute_ceiling_speedup_at_8": round(t_hbm_steady / max(s["steady_ms_per_step"] for s in shards), 3)},
                     "h2d_whole_block_ms": round(h2d, 4),
                     "ingest_bounds": {"broadcast_from_one_host_link": {"ms_per_block": round(h2d, 4), "speedup_ceiling": round(t_host["dt"] / args.steps * 1e3 / max(h2d, t32), 3),
                                                                        "note": "north_star's literal form, host-fed: the whole block crosses ONE PCIe link per step"},
                                       "allgather_of_stripes": {"h2d_ms_per_rank": round(h2d / 8, 4), "speedup_ceiling": round(t_host["dt"] / args.steps * 1e3 / t32, 3),
                                                                "note": "each rank ingests 1/8 over its own link; the xGMI all-gather itself is not measurable on one GPU"}},
                     # link-rate bounds from the xGMI figures of MI355X_MICROARCH.md (7 links x ~153 GB/s per GPU, i.e. ~77 GB/s per direction
                     # and link): what an exchange cannot beat, NOT a measurement - one GPU cannot measure them
                     "xgmi_link_rate_bounds_ms": {"broadcast_one_link_per_hop": round(case.nbytes / 76.5e9 * 1e3, 3),
                                                  "allgather_into_each_gpu_over_7_links": round(case.nbytes * 7 / 8 / (7 * 76.5e9) * 1e3, 3),
                                                  "source": "spec link rates, not measured"},
                     "note": "projection from one GPU, not a measurement of 8; the driver's N = 8 run reports by_exchange / rank_ms_per_step"}
        # The chain walk -> scans -> check is a fixed cost per FEED: a rank that takes two blocks per feed (the exchange's buffers in pairs) pays
        # it once for 32 s of signal.  Timing only (the block twice in a row: what the second copy decodes is not looked at), its own try.
        try:
            projected["two_blocks_per_feed"] = rank_two_blocks_per_feed(case, per, torch, local, max(args.steps, 20), t_hbm_steady if t_hbm_steady is not None else t256)
            # `bench.py --gpus 8` hands its ranks (32 channels each) the exchanged blocks two per feed (Case.pair): the figure that goes with that run
            projected["as_bench_gpus_8_feeds_its_ranks"] = {"blocks_per_feed": 2, "compute_ceiling_speedup_at_8": projected["two_blocks_per_feed"]["compute_ceiling_speedup_at_8"],
                                                            "one_block_per_feed": projected["compute_ceiling_speedup_at_8"]}
        except Exception as e:  # noqa: BLE001
            projected["two_blocks_per_feed"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
    except Exception as e:  # noqa: BLE001 - informational block: a failure here must not take the headline line with it
        projected = {"error": f"{type(e).__name__}: {str(e)[:400]}"}

    if world == 1 and not args.no_secondary and args.workload == "config4":
        from oracle import pyoracle as po
        try:
            secondary.append(dropin_block_rate(case, torch))
        except Exception as e:  # noqa: BLE001
            secondary.append({"workload": "dropin_320kB_block", "error": f"{type(e).__name__}: {str(e)[:300]}"})
        for name, oracle_check in (("config3", True), ("config2", False), ("config4_bursty", True), ("config5", True)):
            # a failure in a SECONDARY configuration is reported in its entry, it does not take the headline line with it
            c2 = None
            try:
                c2 = Case(name, args.duration, 1, 0, local, torch)
                secondary.append(measure_secondary(c2, name, oracle_check and not args.no_verify, args, dist, po))
                iq2, b2 = c2.iq, c2.bursts
                c2.close()
                if name == "config4_bursty":      # ... and a rank's share of it at N = 8: does the back end stay hidden where the front is 8x shorter?
                    c2 = Case(name, args.duration, 1, 0, local, torch, iq=iq2, bursts=b2, shard=(96, 32))
                    secondary.append(measure_secondary(c2, name, False, args, dist, po))
                    c2.close()
            except Exception as e:  # noqa: BLE001
                secondary.append({"workload": name, "error": f"{type(e).__name__}: {str(e)[:400]}"})
                if c2 is not None:
                    c2.close()

    # ... and the 8-GPU split driven FROM C: vdl2hip_group_* over 8 members, all of them on this GPU ("virtual shards"), fed from
    # page-locked host memory, both exchange forms.  One GPU does the work of eight here, so the time per step is to be read against
    # t_all_channels_ms: what it shows is what the C path adds (8 x ~12 launches per block from one host thread, the stripes' H2D
    # copies and the peer copies that stand in for xGMI) - not a speed-up.  LAST of all: its 8 receivers create and destroy a hundred
    # streams, and the runtime's mapping of streams to its few hardware queues is not the same afterwards (receivers measured after it
    # in the same process ran 30-60 % slower than in a fresh one: profiles/r06_hw_queues_ab.txt against r06b).
    if isinstance(projected, dict) and "error" not in projected:
        try:
            projected["group_from_c"] = group_from_c(case, args, torch, local)
        except Exception as e:  # noqa: BLE001
            projected["group_from_c"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}

    if rank == 0:
        value = case.nsamples * args.steps / t_host["dt"] / 1e6
        value_hbm = case.nsamples * args.steps / t_hbm["dt"] / 1e6
        widx = WORKLOAD_INDEX.get(args.workload)
        out = {
            "metric": "IQ MS/s demodulated end-to-end",
            "value": round(value, 3), "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_host["dt"] / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": args.repeats, "clock_warmup_s": CLOCK_WARMUP_S, "value_is": f"median of {args.repeats} repeats of the K = {args.steps} timed steps",
            "ms_per_step_all_repeats": t_host["all_ms_per_step"], "ms_per_step_min": t_host["min_ms_per_step"],
            "steady_state": steady,
            "value_hbm_resident": round(value_hbm, 3), "ms_per_step_hbm_resident": round(t_hbm["dt"] / args.steps * 1e3, 4),
            "ms_per_step_hbm_resident_all_repeats": t_hbm["all_ms_per_step"],
            "parity": ("frames, integer metadata, burst timing and the reference's 18 counters identical to the CPU oracle on the whole block, every "
                       "channel; float metadata within SURVEY 8.5's tolerances (0.01 ppm, 0.05 dB); no tie allowances (config.verified)") if verified else "not checked (--no-verify)",
            "config": {"workload": (f"configs[{widx}] " if widx else "") + f"({args.workload}): synthetic 2.1 MS/s cs16 IQ, {cfg.duration_s:g} s per step, {case.C} VDL2 channels "
                                   f"in total, {case.count} per GPU; value = block in page-locked host memory -> frames in host memory "
                                   f"(H2D inside the step, overlapped); six blocks in flight",
                       "channels_total": case.C, "channels_per_gpu": case.count, "samples_per_step": case.nsamples,
                       "channel_MS_per_s": round(value * case.C, 1),
                       "realtime_channels_at_2.1MSps": round(value * case.C / 2.1, 1),
                       "frames_per_step": t_host["frames"] / args.steps,
                       "parallelism": (f"channels sharded x{world} ({case.count} per GPU), RCCL {mode} of every IQ block inside the timed steps"
                                       + (" [REHEARSAL: all ranks on one GPU over gloo - not a measurement]" if rehearsal else "")
                                       if world > 1 else "single GPU, all channels"),
                       "referee_scans_ahead_of_the_walk": case.prescan,
                       # N > 1, ranks of <= 64 channels: the exchanged blocks go to the receiver two per feed in the timed loops (dist.ShardedFeeder)
                       "blocks_per_feed": 2 if case.pair else 1,
                       "exchange": exchange_info,
                       "by_exchange": by_exchange,
                       "rank_ms_per_step": t_host.get("rank_ms_per_step"),
                       "stage_ms_per_step": stage_ms,
                       "stage_ms_per_step_hbm_resident": stage_ms_hbm,
                       "walk_segments_per_step": {"adopted": t_host["seg_adopted"], "walked_sequentially": t_host["seg_walked"]},
                       "lookback_fallbacks_per_step": t_host["lookback_fallbacks"] / args.steps,
                       "cold_start_feeds_in_the_timed_region": t_host["cold_start_feeds"],
                       "verified": verified,
                       "verified_in_pieces": verified_pieces,
                       "synth_s": round(case.t_synth, 1),
                       "secondary": secondary},
            "roofline": roofline_of(t_hbm, pmc_traffic(args.workload, case)),
        }
        # the parity gate's outcome as flat scalars (headline workload, then every secondary one that was checked against the oracle)
        def flat(prefix, v):
            out[prefix + "timing_ties"] = v["timing_ties"]; out[prefix + "nf_update_ties"] = v["nf_update_ties"]
            out[prefix + "channels_counters_identical"] = v["channels_with_reference_counters_identical"]
            out[prefix + "channels"] = v["channels"]
            out[prefix + "oracle_identical"] = bool(v["oracle_identical"])
            out[prefix + "referee_scans"] = v["referee"].get("scans"); out[prefix + "referee_refused"] = v["referee"].get("refused")
        def flat_pieces(prefix, v):
            out[prefix + "pieces"] = v["pieces"]; out[prefix + "pieces_timing_ties"] = v["timing_ties"]; out[prefix + "pieces_nf_update_ties"] = v["nf_update_ties"]
            out[prefix + "pieces_channels_counters_identical"] = v["channels_with_reference_counters_identical"]
            out[prefix + "pieces_oracle_identical"] = bool(v["oracle_identical"])
        try:
            if verified:
                flat("parity_", verified)
            if verified_pieces:
                flat_pieces("parity_", verified_pieces)
            for e in secondary:
                if isinstance(e, dict) and e.get("verified"):
                    flat("parity_" + e.get("name", "secondary") + "_", e["verified"])
                if isinstance(e, dict) and e.get("verified_in_pieces"):
                    flat_pieces("parity_" + e.get("name", "secondary") + "_", e["verified_in_pieces"])
        except Exception as e:  # noqa: BLE001 - the scalars repeat what config.verified holds; never lose the line over them
            out["parity_scalars_error"] = f"{type(e).__name__}: {str(e)[:200]}"
        if referee_ab is not None:
            out["referee"] = referee_ab
        if world > 1:
            out["rccl_version"] = exchange_info.get("rccl_version"); out["distinct_devices"] = exchange_info.get("distinct_devices")
            out["by_exchange"] = by_exchange
        out["roofline"]["issue_rate_ceiling"], out["roofline"]["co_bound"] = measured
        if measured[0] and measured[1]:
            rl = out["roofline"]
            items = rl["chan_samples_per_launch"] / 64.0 / (4 * measured[1]["compute_units"])          # wavefront-items per SIMD and launch
            rl["co_bound"]["k_chanfir_clocks_per_chan_sample_wave_per_SIMD"] = round(rl["avg_launch_ms"] * 1e-3 * measured[0]["shader_clock_GHz"] * 1e9 / items, 1)
        out["roofline"]["measured_in"] = "the HBM-resident timed region (HIP start/stop events attached to each k_chanfir launch; median of the repeats)"
        out["roofline"]["avg_launch_ms_host_fed"] = round(t_host["k1_ms"], 5)
        if projected is not None:
            out["projected_scaling"] = projected
        if cpu_baseline is not None:
            out["cpu_baseline"] = cpu_baseline
        print(json.dumps(out), flush=True)
    case.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
