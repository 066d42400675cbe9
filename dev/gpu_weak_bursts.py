"""config4 with and without its --max-ppm gate in the reference's own blocks (320 000 bytes), k per feed: without the gate idle channels
lock on to their neighbours' leakage and decode it - weak bursts, many symbols within the referee's margin.
usage: python dev/gpu_weak_bursts.py [max_ppm] [k,k,...] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dumpvdl2_amd import vdl2hip, synth, workloads
cfg = workloads.config4(float(sys.argv[3]) if len(sys.argv) > 3 else 4.0)
if os.environ.get("NOISE"):          # (weak bursts of another kind: the same capture at a lower signal-to-noise ratio; config4's is 26 dB)
    import dataclasses
    cfg = dataclasses.replace(cfg, noise_sigma=float(os.environ["NOISE"]))
iq, _ = synth.synthesize(cfg)
raw = iq.view(np.uint8)
BLK = 320000
ppm = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for k in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,16,105").split(",")]:
    piece = k * BLK
    rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, ppm, max_block_bytes=piece)
    rx.set_drain_lag(0 if k == 1 else 2)
    for o in range(0, min(raw.size, 4 * piece), piece): rx.feed(raw[o:o + piece]); rx.drain_packed()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for o in range(0, raw.size, piece): rx.feed(raw[o:o + piece]); n += rx.drain_packed()[0]
    rx.set_drain_lag(0); n += rx.drain_packed()[0]
    dt = time.perf_counter() - t0
    st = rx.stats()
    print(f"max_ppm {ppm}{(' noise ' + os.environ['NOISE']) if os.environ.get('NOISE') else ''}: {k} blocks per feed: {dt / (raw.size / BLK) * 1e3:.3f} ms per block, frames {n}, scans {st['referee_scans']} (candidate {st['referee_candidate_scans']}, header {st['referee_header_scans']}, symbol {st['referee_symbol_scans']}), rewalks {st['referee_rewalks']}", flush=True)
    rx.close()
