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
          link).  `--exchange auto` (default) times both bare exchanges during warm-up and uses the faster one.

So total work is fixed ("scaling": "strong") and value(N=8) / value(N=1) is the 8-GPU speed-up north_star asks for.

`value` is the host-fed rate (cs16 block in page-locked host memory -> all frames of the block delivered):
the block crosses PCIe inside every timed step, overlapped with compute by the library's copy stream.  The same K steps
are then repeated with the block already resident in HBM (`value_hbm_resident`).  At N = 1 two smaller configurations
(64 and 8 channels) are measured the same way and reported under config.secondary.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import platform
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

OS = 20
# algorithmic bytes per channel-sample = 4 B (cs16 I+Q, read per channel as the reference does) + 8/os B (one
# float2 decimated output) - exactly what the channeliser reads and writes per channel-sample.
ALGO_BYTES = 4.0 + 8.0 / OS
FLOP_PER_CHAN_SAMPLE = 30.0                       # LUT interpolation 6 + mix 6 + 2 x 9 IIR
HBM_PEAK_GBS = 8000.0                             # MI355X_MICROARCH.md: 8 TB/s spec
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
    return {"nproc": os.cpu_count(), "cpu_model": model or platform.processor(), "compiler": cc}


class Case:
    """One workload on this rank: receiver over the rank's channel shard + the feeder that brings the blocks."""

    def __init__(self, name, duration, world, rank, local, torch, channels=None):
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
        self.iq, self.bursts = synth.synthesize(cfg)
        self.t_synth = time.time() - t0
        self.nvals = self.iq.size
        self.nbytes = self.nvals * 2
        self.nsamples = self.nvals // 2
        self.C = len(cfg.freqs)
        self.first, self.count = vdist.shard_channels(self.C, world, rank)
        self.device = torch.device("cuda", local)
        self.rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm,
                                   device=local, max_block_bytes=self.nbytes, chan_first=self.first, chan_count=self.count)
        self.front = torch.cuda.ExternalStream(self.rx.stream(), device=self.device)
        self.host = torch.from_numpy(self.iq)

    def feeder(self, mode, source):
        return self.vdist.ShardedFeeder(self.rx, self.host, self.world, self.rank, mode=mode, source=source, device=self.device,
                                        front_stream=self.front)

    def frames_of_step(self, feeder):
        """one synchronous step: every frame this rank's channels produce for one block"""
        self.rx.set_drain_lag(0)
        n, recs, octs = feeder.step()
        return self.vdl2hip.Receiver.unpack(n, recs, octs)

    def timed(self, feeder, steps, dist):
        """exactly `steps` steps in streaming mode (three blocks in flight); every block fully delivered inside the region"""
        torch = self.torch
        self.rx.set_profiling(1)           # start/stop events on the channeliser launch only: its duration is the roofline figure
        self.rx.set_drain_lag(2)
        s0 = self.rx.stats()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nframes = 0
        for _ in range(steps):
            nframes += feeder.step()[0]
        self.rx.set_drain_lag(0)
        nframes += self.rx.drain_packed()[0]
        feeder.finish()
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        s1 = self.rx.stats()
        if self.world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            nf = torch.tensor([nframes], dtype=torch.int64, device=self.device)
            dist.all_reduce(nf)
            nframes = int(nf.item())
        launches = max(1, s1["chanfir_launches"] - s0["chanfir_launches"])
        k1_ms = (s1["chanfir_ms"] - s0["chanfir_ms"]) / launches
        k1_cs = (s1["chan_samples"] - s0["chan_samples"]) / launches
        assert s1["front_sync_timeouts"] == 0 and s1["overflow_feeds"] == s0["overflow_feeds"], "device-side overflow or look-back timeout"
        return {"dt": dt, "frames": nframes, "k1_ms": k1_ms, "k1_chan_samples": k1_cs,
                "seg_adopted": (s1["seg_adopted"] - s0["seg_adopted"]) / steps, "seg_walked": (s1["seg_walked"] - s0["seg_walked"]) / steps}

    def stage_times(self, feeder, nstage=4):
        """per-stage kernel times (informational): a short untimed pass with every stage's launch timed"""
        self.rx.set_profiling(2)
        self.rx.set_drain_lag(2)
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


def roofline_of(t, case, traffic):
    k1_ms, cs = t["k1_ms"], t["k1_chan_samples"]
    ach = cs * ALGO_BYTES / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
    tf = cs * FLOP_PER_CHAN_SAMPLE / (k1_ms * 1e-3) / 1e12 if k1_ms > 0 else 0.0
    return {"bound": "hbm", "kernel": "k_chanfir", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes_per_chan_sample": ALGO_BYTES, "algorithmic_bytes_per_launch": cs * ALGO_BYTES,
            "avg_launch_ms": round(k1_ms, 5), "chan_samples_per_launch": cs,
            # the kernel is instruction-issue bound, not HBM bound: the same launch against the FP32 vector peak
            "valu": {"flop_per_chan_sample": FLOP_PER_CHAN_SAMPLE, "achieved": round(tf, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tf / VALU_PEAK_TFLOPS, 4)}}


def pmc_traffic(workload, case):
    """HBM traffic of K1 per launch: PMC counters cannot be read from inside this process; the number measured with
    rocprofv3 on this same command (tests/gpu_pmc_traffic.sh) is kept under profiles/ and quoted only for the workload and
    shard size it was taken on."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pt = json.load(f)
        for e in pt.get("k_chanfir_by_workload", []):
            w = e["workload"]
            if (w["name"], w["channels_per_gpu"], float(w["duration_s"])) == (workload, case.count, float(case.cfg.duration_s)):
                return e["traffic_bytes"]
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--duration", type=float, default=16.0, help="seconds of 2.1 MS/s signal per step")
    ap.add_argument("--channels", type=int, default=8, help="channel count of config2 (experiments)")
    ap.add_argument("--workload", default="config4", choices=["config2", "config3", "config4", "config5"],
                    help="BASELINE configs[1..4]; default config4 = north_star's 256 channels")
    ap.add_argument("--exchange", default="auto", choices=["auto", "broadcast", "allgather"],
                    help="N>1: how each rank gets the raw IQ block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N=1: skip the 64- and 8-channel configurations")
    ap.add_argument("--oracle-threads", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from dumpvdl2_amd import vdl2hip
    from dumpvdl2_amd import dist as vdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # development only (tests/gpu_r02_run_e.sh): rehearse the N > 1 control flow on a one-GPU box - every rank on device 0, gloo
    # instead of RCCL as the transport.  Never set by the driver; the numbers of such a run mean nothing.
    rehearsal = os.environ.get("VDL2_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
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

    # ---- exchange: measured, not assumed ----
    exchange_info = None
    mode = "broadcast"
    if world > 1:
        cands = ["broadcast", "allgather"] if args.exchange == "auto" else [args.exchange]
        if case.nbytes % world:
            cands = ["broadcast"]
        exchange_info = {}
        for src_kind in ("host", "hbm"):
            for m in cands:
                secs, ok = vdist.time_exchange(case.host, world, rank, m, src_kind, case.device)
                exchange_info[f"{m}_{src_kind}_ms"] = round(secs * 1e3, 4) if ok else None
        best = {m: exchange_info.get(f"{m}_host_ms") for m in cands}
        mode = min((m for m in cands if best[m] is not None), key=lambda m: best[m])
        exchange_info["chosen"] = mode

    # ---- warm-up, with the parity gate on the first pass ----
    f_host = case.feeder(mode, "host")
    verified = None
    cpu_baseline = None
    fr = case.frames_of_step(f_host)
    allfr = vdist.gather_frames(fr, dst=0) if world > 1 else fr
    if rank == 0 and not args.no_verify:
        from util import truth_is_subset, compare_at_full_size
        missing = truth_is_subset(case.bursts, allfr)
        want = sum(len(b.frames) for b in case.bursts if b.decodable)
        # with injected errors some bursts pushed past the nominal RS capacity still decode (both here and in the oracle)
        exact = not cfg.error_injection
        assert missing == 0 and (len(allfr) == want if exact else len(allfr) >= want), \
            f"parity gate: {missing} transmitted frames missing, {len(allfr)} decoded vs {want} sent"
        verified = {"tx_frames": want, "decoded": len(allfr)}
        # the oracle on the WHOLE block (all channels, all 16 s) on this box's host cores: the parity gate and, at N = 1, the
        # CPU baseline in one pass
        from oracle import pyoracle as po
        nth = args.oracle_threads or min(case.C, os.cpu_count() or 1)
        o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
        t0 = time.perf_counter()
        o.process(case.iq.view(np.uint8), block_bytes=320000, nthreads=nth)
        tc = time.perf_counter() - t0
        ofr = o.frames()
        cmp = compare_at_full_size(ofr, allfr, label="bench oracle gate")
        # the reference's own 18 statsd counters per channel (the last two are this repo's diagnostics: preambles dropped by
        # --max-ppm and out-of-range slicer indices, which a timing tie can move by one)
        nref, ndiff = 18, 0
        for ch in range(case.first, case.first + case.count):
            co, cg = list(o.counters(ch).values()), list(case.rx.counters(ch).values())
            assert co[:nref] == cg[:nref], f"reference counters of channel {ch} differ from the oracle's: {co} vs {cg}"
            ndiff += co[nref:] != cg[nref:]
        o.close()
        verified.update({"oracle_window_s": cfg.duration_s, "oracle_frames": len(ofr), "oracle_identical": True,
                         "burst_timing_ties": cmp["timing_ties"], "nf_update_ties": cmp["nf_update_ties"], "max_abs_diff": cmp["max_abs_diff"],
                         "channels_with_reference_counters_identical": case.count, "channels_with_diagnostic_counter_diff": int(ndiff)})
        if world == 1 and not args.no_cpu_baseline:
            ci = cpu_info()
            cpu_baseline = {"value": round(case.nsamples / tc / 1e6, 4), "unit": "MS/s", "cores": nth, "kind": "port",
                            "channel_MS_per_s": round(case.nsamples * case.C / tc / 1e6, 1),
                            "sample": f"the same {cfg.duration_s:g} s x {case.C}-channel block, one pass ({tc:.1f} s); CPU restatement of the "
                                      f"reference (oracle/), {nth} threads over the channels + serial sample conversion, 320000-byte blocks "
                                      f"as process_iq_file()",
                            "flags": "-O2 -fno-fast-math -ffp-contract=off (oracle/Makefile)", **ci}
            try:
                of = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm, variant="fast")
                n_fast = min(case.nbytes, 4 * cfg.sample_rate * 4)          # 4 s of the block: bounded
                t0 = time.perf_counter()
                of.process(case.iq.view(np.uint8)[:n_fast], block_bytes=320000, nthreads=nth)
                tf = time.perf_counter() - t0
                of.close()
                cpu_baseline["fast_math"] = {"value": round(n_fast / 4 / tf / 1e6, 4), "unit": "MS/s", "flags": "-O3 -ffast-math (mirrors src/CMakeLists.txt:35-38)",
                                             "sample": f"first {n_fast / 4 / cfg.sample_rate:g} s of the block"}
            except Exception as e:  # noqa: BLE001 - the fast-math build is optional
                cpu_baseline["fast_math"] = {"error": str(e)[:200]}
    if world > 1:
        dist.barrier()
    for _ in range(max(0, args.warmup - 1)):
        f_host.step()
    case.rx.sync()

    # ---- timed region: exactly K steps, host-fed ----
    t_host = case.timed(f_host, args.steps, dist)
    stage_ms = case.stage_times(f_host)
    del f_host
    # ---- the same K steps with the block resident in HBM ----
    f_hbm = case.feeder(mode, "hbm")
    for _ in range(2):
        f_hbm.step()
    case.rx.set_drain_lag(0); case.rx.drain_packed()
    t_hbm = case.timed(f_hbm, args.steps, dist)
    del f_hbm

    secondary = []
    if world == 1 and not args.no_secondary and args.workload == "config4":
        case.close()
        for name in ("config3", "config2"):
            c2 = Case(name, args.duration, 1, 0, local, torch)
            fh = c2.feeder("broadcast", "host")
            fr2 = c2.frames_of_step(fh)
            from util import truth_is_subset
            miss = truth_is_subset(c2.bursts, fr2)
            want2 = sum(len(b.frames) for b in c2.bursts if b.decodable)
            assert miss == 0 and len(fr2) == want2, f"{name}: {miss} transmitted frames missing, {len(fr2)} decoded vs {want2} sent"
            fh.step(); c2.rx.sync()
            th = c2.timed(fh, args.steps, dist)
            del fh
            fd = c2.feeder("broadcast", "hbm")
            fd.step(); fd.step(); c2.rx.set_drain_lag(0); c2.rx.drain_packed()
            td = c2.timed(fd, args.steps, dist)
            del fd
            rl = roofline_of(td, c2, pmc_traffic(name, c2))
            secondary.append({"workload": f"configs[{WORKLOAD_INDEX[name]}] ({name}): {c2.C} channels, {c2.cfg.duration_s:g} s",
                              "value": round(c2.nsamples * args.steps / th["dt"] / 1e6, 3),
                              "value_hbm_resident": round(c2.nsamples * args.steps / td["dt"] / 1e6, 3),
                              "ms_per_step": round(th["dt"] / args.steps * 1e3, 4), "ms_per_step_hbm_resident": round(td["dt"] / args.steps * 1e3, 4),
                              "frames_per_step": th["frames"] / args.steps, "tx_frames_all_recovered": True,
                              "k_chanfir_ms": rl["avg_launch_ms"], "roofline_frac": rl["frac"], "valu_frac": rl["valu"]["frac"]})
            c2.close()

    if rank == 0:
        value = case.nsamples * args.steps / t_host["dt"] / 1e6
        value_hbm = case.nsamples * args.steps / t_hbm["dt"] / 1e6
        widx = WORKLOAD_INDEX[args.workload]
        out = {
            "metric": "IQ MS/s demodulated end-to-end",
            "value": round(value, 3), "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_host["dt"] / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_hbm_resident": round(value_hbm, 3), "ms_per_step_hbm_resident": round(t_hbm["dt"] / args.steps * 1e3, 4),
            "config": {"workload": f"configs[{widx}] ({args.workload}): synthetic 2.1 MS/s cs16 IQ, {cfg.duration_s:g} s per step, {case.C} VDL2 channels "
                                   f"in total, {case.count} per GPU; value = block in page-locked host memory -> frames in host memory "
                                   f"(H2D inside the step, overlapped); three blocks in flight",
                       "channels_total": case.C, "channels_per_gpu": case.count, "samples_per_step": case.nsamples,
                       "channel_MS_per_s": round(value * case.C, 1),
                       "realtime_channels_at_2.1MSps": round(value * case.C / 2.1, 1),
                       "frames_per_step": t_host["frames"] / args.steps,
                       "parallelism": (f"channels sharded x{world} ({case.count} per GPU), RCCL {mode} of every IQ block inside the timed steps"
                                       + (" [REHEARSAL: all ranks on one GPU over gloo - not a measurement]" if rehearsal else "")
                                       if world > 1 else "single GPU, all channels"),
                       "exchange": exchange_info,
                       "stage_ms_per_step": stage_ms,
                       "walk_segments_per_step": {"adopted": t_host["seg_adopted"], "walked_sequentially": t_host["seg_walked"]},
                       "verified": verified,
                       "synth_s": round(case.t_synth, 1),
                       "secondary": secondary},
            "roofline": roofline_of(t_hbm, case, pmc_traffic(args.workload, case)),
        }
        out["roofline"]["measured_in"] = "the HBM-resident timed region (HIP start/stop events attached to each k_chanfir launch)"
        out["roofline"]["avg_launch_ms_host_fed"] = round(t_host["k1_ms"], 5)
        if cpu_baseline is not None:
            out["cpu_baseline"] = cpu_baseline
        print(json.dumps(out), flush=True)
    if not (world == 1 and secondary):
        case.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
