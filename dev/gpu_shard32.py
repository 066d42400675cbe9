"""The rank-sized workload of the 8-GPU split on ONE GPU: config4's 16 s block (256 channels in the air), demodulated by a
receiver that owns only `count` consecutive channels starting at `count * r` - what rank r of world 256/count runs.  Block
resident in HBM (what an RCCL broadcast / all-gather leaves behind), three blocks in flight, K timed steps, repeated.

usage: python dev/gpu_shard32.py [--count 32] [--ranks 0,3,7] [--steps 20] [--repeats 3] [--workload config4] [--json out.json]

Prints one JSON object: per rank ms/step (min / median of the repeats), the channeliser's own time per launch, per-stage kernel
times of a profiled pass, and the frames check (every transmitted frame of the shard's channels recovered).  Knobs of the
library (VDL2HIP_CR, VDL2HIP_K1_TILES, VDL2HIP_SEG_MAX ...) are taken from the environment and echoed."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from dumpvdl2_amd import synth, vdl2hip, workloads  # noqa: E402
from util import truth_is_subset  # noqa: E402


def run_rank(cfg, dev_block, nbytes, bursts, first, count, steps, repeats, check=True):
    rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm,
                          max_block_bytes=nbytes, chan_first=first, chan_count=count)
    out = {"chan_first": first, "chan_count": count}
    rx.set_drain_lag(0)
    rx.feed_device(dev_block.data_ptr(), nbytes)
    fr = vdl2hip.Receiver.unpack(*rx.drain_packed())
    if check:
        mine = [b for b in bursts if first <= b.chan < first + count]
        want = sum(len(b.frames) for b in mine if b.decodable)
        missing = truth_is_subset(mine, fr)
        assert missing == 0 and (len(fr) >= want if cfg.error_injection else len(fr) == want), (missing, len(fr), want)
        out["frames_per_step"] = len(fr)
    for _ in range(3):
        rx.feed_device(dev_block.data_ptr(), nbytes); rx.drain_packed()
    rx.set_profiling(1)
    times, k1 = [], []
    for _ in range(repeats):
        rx.set_drain_lag(vdl2hip.MAX_DRAIN_LAG)
        s0 = rx.stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            rx.feed_device(dev_block.data_ptr(), nbytes)
            rx.drain_packed()
        rx.set_drain_lag(0)
        rx.drain_packed()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / steps * 1e3)
        s1 = rx.stats()
        k1.append((s1["chanfir_ms"] - s0["chanfir_ms"]) / max(1, s1["chanfir_launches"] - s0["chanfir_launches"]))
        assert s1["front_sync_timeouts"] == s0["front_sync_timeouts"] and s1["overflow_feeds"] == s0["overflow_feeds"]
    out["ms_per_step"] = {"min": round(min(times), 4), "median": round(statistics.median(times), 4), "all": [round(t, 4) for t in times]}
    out["k_chanfir_ms"] = round(statistics.median(k1), 4)
    # per-stage kernel times, in the pipeline (every launch stamped: a few percent slower)
    rx.set_profiling(2); rx.set_drain_lag(vdl2hip.MAX_DRAIN_LAG)
    sa = rx.stats()
    n = 6
    for _ in range(n):
        rx.feed_device(dev_block.data_ptr(), nbytes); rx.drain_packed()
    rx.set_drain_lag(0); rx.drain_packed(); torch.cuda.synchronize()
    sb = rx.stats()
    out["stage_ms_in_pipeline"] = {k: round((sb[k] - sa[k]) / n, 4) for k in ("chanfir_ms", "sync_ms", "walk_ms", "nf_ms", "burst_ms")}
    # and alone (one block in flight): the latency chain of a single block
    rx.set_drain_lag(0)
    sa = rx.stats()
    t0 = time.perf_counter()
    for _ in range(n):
        rx.feed_device(dev_block.data_ptr(), nbytes); rx.drain_packed()
    torch.cuda.synchronize()
    out["ms_per_step_one_block_in_flight"] = round((time.perf_counter() - t0) / n * 1e3, 4)
    sb = rx.stats()
    out["stage_ms_alone"] = {k: round((sb[k] - sa[k]) / n, 4) for k in ("chanfir_ms", "sync_ms", "walk_ms", "nf_ms", "burst_ms")}
    rx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=32)
    ap.add_argument("--ranks", default="0,3,7")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--duration", type=float, default=16.0)
    ap.add_argument("--workload", default="config4")
    ap.add_argument("--json", default="")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--sweep", default="", help='e.g. "VDL2HIP_CR=1,2,4;VDL2HIP_K1_TILES=2,4,8": every combination (the library reads its knobs at create)')
    a = ap.parse_args()
    cfg = getattr(workloads, a.workload)(a.duration)
    import pickle
    cache = f"/tmp/vdl2_shard32_{a.workload}_{a.duration:g}.pkl"      # (several processes in one call: the capture is synthesised once)
    if os.path.exists(cache):
        with open(cache, "rb") as f:
            iq, bursts = pickle.load(f)
    else:
        iq, bursts = synth.synthesize(cfg)
        with open(cache, "wb") as f:
            pickle.dump((iq, bursts), f, protocol=4)
    dev = torch.from_numpy(iq).cuda()
    res = {"workload": a.workload, "duration_s": a.duration, "channels_in_the_air": len(cfg.freqs), "steps": a.steps, "repeats": a.repeats,
           "env": {k: v for k, v in os.environ.items() if k.startswith("VDL2HIP_")}, "ranks": []}
    ranks = [int(x) for x in a.ranks.split(",") if x != ""]
    if a.sweep:
        import itertools
        axes = [(kv.split("=")[0], kv.split("=")[1].split(",")) for kv in a.sweep.split(";") if kv]
        res["sweep"] = []
        for combo in itertools.product(*[v for _, v in axes]):
            env = {k: v for (k, _), v in zip(axes, combo)}
            for k, v in env.items():
                if v == "-":
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            for r in ranks:
                try:
                    o = run_rank(cfg, dev, iq.nbytes, bursts, r * a.count, a.count, a.steps, a.repeats, check=not a.no_check)
                    line = {"env": env, "rank": r, "ms_per_step": o["ms_per_step"], "k_chanfir_ms": o["k_chanfir_ms"], "pipe": o["stage_ms_in_pipeline"]}
                except Exception as e:  # noqa: BLE001
                    line = {"env": env, "rank": r, "error": str(e)[:200]}
                res["sweep"].append(line)
                print(json.dumps(line), file=sys.stderr, flush=True)
        for k, _ in axes:
            os.environ.pop(k, None)
    for r in ([] if a.sweep else ranks):
        res["ranks"].append({"rank": r, **run_rank(cfg, dev, iq.nbytes, bursts, r * a.count, a.count, a.steps, a.repeats, check=not a.no_check)})
    if res["ranks"]:
        res["max_over_ranks_ms_per_step"] = max(x["ms_per_step"]["median"] for x in res["ranks"])
    s = json.dumps(res)
    print(s, flush=True)
    if a.json:
        with open(a.json, "w") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
