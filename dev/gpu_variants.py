"""Variant sweep on ONE GPU box: the workloads are synthesised once (that is the expensive part: ~1 min each), then every
variant - a library build (VDL2HIP_LIB) plus environment switches - runs in a process of its own over the same captures:

  python dev/gpu_variants.py --out gpurun_out/x.jsonl --workloads config4,config4_bursty \
         --variant base --variant eager:VDL2HIP_BACKEND=eager --variant ph:@/tmp/vdl2hip_ph.so --variant ...

A variant is `name[:@lib.so][:ENV=VALUE]...`.  Per variant and workload, block resident in HBM, three blocks in flight:
ms per step (all channels and as the 32-channel shard 96..127 - a rank's share at N = 8), the channeliser's own time per
launch, per-stage kernel times in the pipeline, and the number of frames per step against the transmitted truth (a variant that
loses frames is wrong, whatever its speed).  One JSON line per (variant, workload) is appended to --out."""
import argparse
import json
import os
import pickle
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def synth_to_tmp(name, duration):
    import numpy as np
    from dumpvdl2_amd import synth, workloads
    path = f"/tmp/vdl2_{name}_{duration:g}"
    if not os.path.exists(path + ".npy"):
        cfg = getattr(workloads, name)(duration)
        iq, bursts = synth.synthesize(cfg)
        np.save(path + ".npy", iq)
        with open(path + ".pkl", "wb") as f:
            pickle.dump(bursts, f)
    return path


def measure(rx, torch, dev_block, nbytes, steps, repeats, stage_n=6):
    from dumpvdl2_amd import vdl2hip
    LAG = vdl2hip.MAX_DRAIN_LAG
    out = {}
    times, k1 = [], []
    rx.set_profiling(1)
    for _ in range(repeats):
        rx.set_drain_lag(LAG)
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
        assert s1["overflow_feeds"] == s0["overflow_feeds"]
    out["ms"] = round(statistics.median(times), 4)
    out["ms_all"] = [round(t, 4) for t in times]
    out["k1"] = round(statistics.median(k1), 4)
    rx.set_profiling(2); rx.set_drain_lag(LAG)
    sa = rx.stats()
    for _ in range(stage_n):
        rx.feed_device(dev_block.data_ptr(), nbytes); rx.drain_packed()
    rx.set_drain_lag(0); rx.drain_packed(); torch.cuda.synchronize()
    sb = rx.stats()
    out["stage"] = {k[:-3]: round((sb[k] - sa[k]) / stage_n, 4) for k in ("chanfir_ms", "sync_ms", "walk_ms", "nf_ms", "burst_ms")}
    rx.set_profiling(0)
    return out


def child(args):
    import numpy as np
    import torch
    from dumpvdl2_amd import vdl2hip, workloads
    from util import truth_is_subset
    res = {"variant": args.name, "workload": args.workload, "lib": os.environ.get("VDL2HIP_LIB", "in-tree"),
           "env": {k: v for k, v in os.environ.items() if k.startswith("VDL2HIP_") and k != "VDL2HIP_LIB"}}
    path = f"/tmp/vdl2_{args.workload}_{args.duration:g}"
    iq = np.load(path + ".npy")
    with open(path + ".pkl", "rb") as f:
        bursts = pickle.load(f)
    cfg = getattr(workloads, args.workload)(args.duration)
    nbytes = iq.size * 2
    dev_block = torch.from_numpy(iq).to("cuda:0")
    todo = []
    for part in args.parts.split(","):       # all | shard (= shard96) | shard<first channel>
        if part == "all":
            todo.append(("all", 0, len(cfg.freqs)))
        elif part.startswith("shard"):
            todo.append((part, int(part[5:] or 96), 32))
    for label, first, count in todo:
        rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm,
                              max_block_bytes=nbytes, chan_first=first, chan_count=count)
        rx.set_drain_lag(0)
        rx.feed_device(dev_block.data_ptr(), nbytes)
        fr = vdl2hip.Receiver.unpack(*rx.drain_packed())
        mine = [b for b in bursts if first <= b.chan < first + count]
        want = sum(len(b.frames) for b in mine if b.decodable)
        r = {"frames": len(fr), "tx_frames": want, "missing": truth_is_subset(mine, fr)}
        for _ in range(3):
            rx.feed_device(dev_block.data_ptr(), nbytes); rx.drain_packed()
        r.update(measure(rx, torch, dev_block, nbytes, args.steps, args.repeats))
        st = rx.stats()
        r["fallbacks"] = st["front_sync_timeouts"]
        r["referee"] = {k[8:]: st[k] for k in st if k.startswith("referee_")}; r["feeds"] = st["feeds"]
        res[label] = r
        rx.close()
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/variants.jsonl")
    ap.add_argument("--workloads", default="config4,config4_bursty")
    ap.add_argument("--variant", action="append", default=[])
    ap.add_argument("--duration", type=float, default=16.0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--parts", default="all,shard")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--synth-only", action="store_true")
    ap.add_argument("--name", default="")
    ap.add_argument("--workload", default="config4")
    args = ap.parse_args()
    if args.child:
        return child(args)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    for w in args.workloads.split(","):
        t0 = time.time()
        synth_to_tmp(w, args.duration)
        print(f"# {w}: capture ready ({time.time() - t0:.0f} s)", flush=True)
    if args.synth_only:
        return
    for v in args.variant or ["base"]:
        parts = v.split(":")
        env = dict(os.environ)
        for p in parts[1:]:
            if p.startswith("@"):
                env["VDL2HIP_LIB"] = p[1:]
            else:
                k, _, val = p.partition("=")
                env[k] = val
        for w in args.workloads.split(","):
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--name", parts[0], "--workload", w, "--duration", str(args.duration),
                   "--steps", str(args.steps), "--repeats", str(args.repeats), "--parts", args.parts]
            try:
                p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
                line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                rec = line[-1] if line else json.dumps({"variant": parts[0], "workload": w, "error": (p.stderr or "no output")[-600:]})
            except subprocess.TimeoutExpired:
                rec = json.dumps({"variant": parts[0], "workload": w, "error": "timeout"})
            with open(args.out, "a") as f:
                f.write(rec + "\n")
            j = json.loads(rec)
            if "error" in j:
                print(parts[0], w, "ERROR", j["error"][-300:], flush=True)
            else:
                print(parts[0], w, " ".join(f"{lab}: {j[lab]['ms']} ms (K1 {j[lab]['k1']}; {j[lab]['stage']}; frames {j[lab]['frames']}/{j[lab]['tx_frames']} missing {j[lab]['missing']}; referee {j[lab].get('referee')} in {j[lab].get('feeds')} feeds)"
                                            for lab in j if isinstance(j[lab], dict) and "ms" in j[lab]), flush=True)


if __name__ == "__main__":
    main()
