"""Parity A/B of two (or more) library builds on ONE GPU box, without PyTorch: how many decisions each build makes differently from
the oracle on the same captures -
  * config4_bursty (3 s, 256 channels: the lock-dense block) and config5 (2 s, injected errors): timing ties, noise-floor ties, channels whose
    failure bookkeeping differs (tests/util.py), the worst float differences;
  * the GPU-fuzz seeds that differed from the oracle in the round-4 batches (profiles/r04_gpu_fuzz.txt), re-run per build.
usage: python dev/gpu_parity_ab.py <lib_a.so> <lib_b.so> ...      (child: --child <lib>)"""
import json
import os
import pickle
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
WORK = (("config4_bursty", 3.0), ("config5", 2.0))
FUZZ = ((55, "plain"), (100, "plain"), (104, "extreme"), (145, "plain"), (175, "plain"), (179, "extreme"), (274, "plain"),
        (1001, "plain"), (1014, "extreme"), (1041, "extreme"), (1292, "plain"), (2274, "plain"))


def prepare():
    import numpy as np
    from dumpvdl2_amd import synth, workloads
    from oracle import pyoracle as po
    for name, secs in WORK:
        path = f"/tmp/pab_{name}"
        if os.path.exists(path + ".pkl"):
            continue
        t0 = time.time()
        cfg = getattr(workloads, name)(secs)
        iq, bursts = synth.synthesize(cfg)
        o = po.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=20, max_ppm=cfg.rx_max_ppm)
        o.process(iq.view(np.uint8), block_bytes=1 << 24, nthreads=min(len(cfg.freqs), os.cpu_count() or 8))
        fo = o.frames()
        names = list(o.counters(0).keys())
        co = [list(o.counters(c).values()) for c in range(len(cfg.freqs))]
        np.save(path + ".npy", iq)
        with open(path + ".pkl", "wb") as f:
            pickle.dump({"frames": fo, "names": names, "counters": co}, f)
        print(f"# {name} {secs:g} s: {len(fo)} oracle frames ({time.time() - t0:.0f} s)", flush=True)


def child(lib):
    sys.modules["torch"] = None
    import numpy as np
    from dumpvdl2_amd import vdl2hip, workloads
    from util import compare_at_full_size, compare_reference_counters
    vdl2hip.load_library(lib)
    out = {"lib": os.path.basename(lib)}
    for name, secs in WORK:
        cfg = getattr(workloads, name)(secs)
        iq = np.load(f"/tmp/pab_{name}.npy")
        with open(f"/tmp/pab_{name}.pkl", "rb") as f:
            ref = pickle.load(f)
        rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), 20, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=16_000_000)
        raw = iq.view(np.uint8)
        got, t, k = [], 0, 0
        while t < raw.size:
            m = min((2_000_000, 4_000_000)[k % 2], raw.size - t); k += 1
            rx.feed(raw[t:t + m]); t += m
            got += rx.drain()
        cg = [list(rx.counters(c).values()) for c in range(len(cfg.freqs))]
        try:
            st = compare_at_full_size(ref["frames"], got, label=name, max_tie_frac=0.5)
            which, nbad = compare_reference_counters(ref["names"], ref["counters"], cg, label=name, strict=False, max_channels=256)
            out[name] = {"frames": st["frames"], "timing_ties": st["timing_ties"], "nf_update_ties": st["nf_update_ties"],
                         "bookkeeping_channels": nbad, "bookkeeping": which, "max_abs_diff": st["max_abs_diff"], "on_ties": st["max_abs_diff_on_ties"]}
        except AssertionError as e:
            out[name] = f"FAILED: {str(e)[:300]}"
        rx.close()
    import fuzz_gpu
    fz = {}
    for seed, profile in FUZZ:
        try:
            r = fuzz_gpu.run_seed(seed, profile)
            fz[seed] = "ok" + (f" ({r['ties']} ties)" if r["ties"] else "")
        except fuzz_gpu.Differs as e:
            fz[seed] = f"differs ({'samples' if e.from_samples else 'DEFECT'}, {e.rel:.1e})"
        except AssertionError as e:
            fz[seed] = f"FAILED {str(e)[:120]}"
    out["fuzz"] = fz
    out["fuzz_agree"] = sum(1 for v in fz.values() if v.startswith("ok"))
    print(json.dumps(out), flush=True)


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2])
    prepare()
    for lib in sys.argv[1:]:
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], capture_output=True, text=True, timeout=200)
        except subprocess.TimeoutExpired:
            print(f"{lib}: TIMEOUT", flush=True)
            continue
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        print(line[-1] if line else f"{lib}: FAILED rc={p.returncode}\n{p.stderr[-1500:]}", flush=True)


if __name__ == "__main__":
    main()
