"""A/B of channeliser builds (first written for the table-free NCO experiment, history 60803f7, then the state-basis change) on ONE
GPU box, without PyTorch (its first import on a fresh box costs a minute or two): per library, in a process of its own,
  * the committed golden captures through the library -> frames / metadata / counters against tests/golden/*.json,
  * the decimated stream of a few channels saved for the parent to compare between the libraries,
  * the channeliser's own time per launch (HIP events of the launch) on a 256-channel 16 s block of noise and on a 32-channel one.
usage: python dev/gpu_k1_ab.py <lib_a.so> <lib_b.so> ...   (child: --child <lib>)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
CASES = ("config2_1s", "config4_0p4s", "os10_noisy_1s")


def child(lib):
    sys.modules["torch"] = None           # load_library() would import it first (runtime load order); this process never uses it
    import numpy as np
    from dumpvdl2_amd import vdl2hip, synth
    import cases
    vdl2hip.load_library(lib)
    tag = os.path.basename(lib).replace(".so", "")
    out = {"lib": lib}
    for name in CASES:
        cfg, iq, bursts, gold = cases.load(name)
        rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=iq.size * 2)
        rx.feed(iq)
        fr = rx.drain()
        cnt = [list(rx.counters(c).values()) for c in range(len(cfg.freqs))]
        try:
            cases.check_against_golden(fr, cnt, gold, label=f"{tag}/{name}", exact_diagnostics=False)
            out[name] = f"ok ({len(fr)} frames)"
        except AssertionError as e:
            out[name] = f"DIFFERS: {str(e)[:300]}"
        D = iq.size // 2 // cfg.oversample
        chans = sorted({0, 1, len(cfg.freqs) // 2, len(cfg.freqs) - 1})
        ys = np.stack([rx.read_decimated(c, 0, D) for c in chans])
        np.save(f"/tmp/k1ab_{tag}_{name}.npy", ys)
        rx.close()
        # ... and against the oracle's sequential scan of the same channels (the reference's own arithmetic)
        from oracle import pyoracle as po
        o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample)
        tr = o.trace_all(D)
        o.process(iq.view(np.uint8), block_bytes=320000)
        ref = tr[chans, :D, :]
        d = np.asarray(ys, dtype=np.float64).reshape(ref.shape) - ref
        peak = float(np.abs(ref).max())
        out[name + "_vs_oracle"] = {"max_over_peak": float(np.abs(d).max() / peak), "rms_over_peak": float(np.sqrt(np.mean(d * d) * 2) / peak)}
        o.close()
    rng = np.random.default_rng(5)
    for C, secs in ((256, 16.0), (32, 16.0)):
        n = int(secs * 2100000)
        iq = (rng.standard_normal(2 * n, dtype=np.float32) * 300).astype(np.int16)
        cf = 136975000
        freqs = synth.channel_plan(256, cf, 8000)
        rx = vdl2hip.Receiver(cf, freqs, 20, 1, 3.0, max_block_bytes=iq.size * 2, chan_first=0 if C == 256 else 96, chan_count=C)
        rx.set_profiling(1)
        for _ in range(3):                     # the first block of an idle receiver goes in pieces (cold start: not timed), and the clock has to come up
            rx.feed(iq); rx.drain_packed()
        rx.set_drain_lag(2)
        s0 = rx.stats()
        t0 = time.perf_counter()
        reps = 8
        for _ in range(reps):
            rx.feed(iq); rx.drain_packed()
        rx.set_drain_lag(0); rx.drain_packed()
        dt = (time.perf_counter() - t0) / reps * 1e3
        s1 = rx.stats()
        nl = max(1, s1["chanfir_launches"] - s0["chanfir_launches"])
        out[f"noise_{C}ch"] = {"k_chanfir_ms": round((s1["chanfir_ms"] - s0["chanfir_ms"]) / nl, 4), "launches": nl, "ms_per_step_pageable_feed": round(dt, 3),
                               "fallbacks": s1["front_sync_timeouts"]}
        rx.close()
    print(json.dumps(out), flush=True)


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2])
    import numpy as np
    libs = sys.argv[1:]
    first = None
    for lib in libs:
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], capture_output=True, text=True, timeout=150)
        except subprocess.TimeoutExpired:
            print(f"{lib}: TIMEOUT", flush=True)
            continue
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(f"{lib}: FAILED rc={p.returncode}\n{p.stderr[-1500:]}", flush=True)
            continue
        r = json.loads(line[-1])
        print(json.dumps(r), flush=True)
        if first is None:
            first = r
            continue
        ta, tb = (os.path.basename(x["lib"]).replace(".so", "") for x in (first, r))
        for name in CASES:          # each library's result is compared with the first one's as soon as it is in (a call may be cut short)
            a, b = np.load(f"/tmp/k1ab_{ta}_{name}.npy"), np.load(f"/tmp/k1ab_{tb}_{name}.npy")
            peak = float(np.abs(a).max())
            print(f"{name}: decimated stream {tb} vs {ta}: max |diff| / peak = {float(np.abs(a - b).max()) / peak:.3e} (peak {peak:.4f}, {a.shape})", flush=True)
        for k in ("noise_256ch", "noise_32ch"):
            print(f"{k}: k_chanfir {ta} {first[k]['k_chanfir_ms']} -> {tb} {r[k]['k_chanfir_ms']} ms ({(r[k]['k_chanfir_ms'] / first[k]['k_chanfir_ms'] - 1) * 100:+.1f} %)", flush=True)


if __name__ == "__main__":
    main()
