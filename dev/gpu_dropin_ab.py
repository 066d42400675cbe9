"""Drop-in block rate (dev/gpu_dropin_rate.py's measurement: 320 000-byte blocks, every frame delivered before the next block is fed)
of several library builds on one box, without PyTorch.  usage: python dev/gpu_dropin_ab.py <lib.so> ...   (child: --child <lib>)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BLK = 320000


def child(lib):
    sys.modules["torch"] = None
    import numpy as np
    from dumpvdl2_amd import vdl2hip, workloads
    vdl2hip.load_library(lib)
    iq = np.load("/tmp/dropin_ab.npy"); raw = iq.view(np.uint8)
    cfg = workloads.config4(float(os.environ.get("DROPIN_SECS", "2")))
    res = []
    for lag in (0, 1):
        best = None
        for rep in range(3):
            rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=BLK)
            rx.set_drain_lag(lag)
            n = 0
            for k in range(0, 20 * BLK, BLK):
                rx.feed(raw[k:k + BLK]); n += rx.drain_packed()[0]
            t0 = time.perf_counter()
            for k in range(20 * BLK, raw.size, BLK):
                rx.feed(raw[k:k + BLK]); n += rx.drain_packed()[0]
            rx.set_drain_lag(0); n += rx.drain_packed()[0]
            dt = (time.perf_counter() - t0) / ((raw.size - 20 * BLK + BLK - 1) // BLK)
            rx.close()
            best = dt if best is None else min(best, dt)
        res.append(f"lag {lag}: {best * 1e3:.4f} ms per block (best of 3), {n} frames")
    print(f"{os.path.basename(lib)}: " + "; ".join(res), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        import numpy as np
        from dumpvdl2_amd import synth, workloads
        iq, _ = synth.synthesize(workloads.config4(float(os.environ.get("DROPIN_SECS", "2"))))
        np.save("/tmp/dropin_ab.npy", iq)
        for lib in sys.argv[1:] * 2:                      # every build twice, interleaved: box-level drift shows
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], capture_output=True, text=True, timeout=60)
            print((p.stdout.strip() or f"{lib}: FAILED {p.stderr[-400:]}"), flush=True)
