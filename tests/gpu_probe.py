"""Diagnostic run on a GPU box: compares every stage of the HIP path with the oracle and prints numbers.
Not a test; used while developing (python tests/gpu_probe.py > gpurun_out/probe.log)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from dumpvdl2_amd import synth, workloads, vdl2hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compare(name, raw, cf, freqs, os_, fmt, max_ppm=0.0, chunks=None, max_block=None):
    C = len(freqs)
    o = po.Oracle(cf, freqs, oversample=os_, sample_fmt=fmt, max_ppm=max_ppm)
    sb = 4 if fmt == 1 else 2
    N = raw.size // sb
    D = N // os_
    tr = o.trace_all(D + 4)
    t = time.time(); o.process(raw, block_bytes=1 << 24, nthreads=8); t_or = time.time() - t
    fo = o.frames()
    rx = vdl2hip.Receiver(cf, freqs, os_, fmt, max_ppm, max_block_bytes=max_block or raw.size)
    rx.set_profiling(2)
    t = time.time()
    if chunks is None:
        rx.feed(raw)
    else:
        rng = np.random.default_rng(3); k = 0
        while k < raw.size:
            m = min(raw.size - k, int(rng.integers(chunks[0], chunks[1])) * sb); rx.feed(raw[k:k + m]); k += m
    fg = rx.drain(); t_gpu = time.time() - t
    st = rx.stats()
    # stage 1: decimated samples
    worst = 0.0
    for c in range(min(C, 4)):
        yg = rx.read_decimated(c, max(0, D - 60000), 60000)
        first = max(0, D - 60000)
        yo = tr[c, first:first + len(yg)]
        err = np.abs(yg - yo).max(); ref = np.abs(yo).max()
        worst = max(worst, err / max(ref, 1e-30))
    key = lambda f: (f['chan'], f['burst_ord'], f['idx'])
    fo = sorted(fo, key=key); fg = sorted(fg, key=key)
    nbad = 0
    for a, b in zip(fo, fg):
        if any(a[k] != b[k] for k in ('chan', 'idx', 'octets', 'synd_weight', 'datalen_octets', 'num_fec_corrections', 'burst_ord', 'sync_sample', 'end_sample')):
            nbad += 1
            if nbad <= 3:
                print('   MISMATCH', {k: (a[k], b[k]) for k in ('chan', 'idx', 'burst_ord', 'sync_sample', 'end_sample', 'num_fec_corrections')}, len(a['octets']), len(b['octets']))
    dp = max([abs(a['frame_pwr_dbfs'] - b['frame_pwr_dbfs']) for a, b in zip(fo, fg)] or [0])
    dn = max([abs(a['nf_pwr_dbfs'] - b['nf_pwr_dbfs']) for a, b in zip(fo, fg)] or [0])
    dq = max([abs(a['ppm_error'] - b['ppm_error']) for a, b in zip(fo, fg)] or [0])
    cn = sum(list(o.counters(c).values()) != list(rx.counters(c).values()) for c in range(C))
    print(f"{name}: frames oracle {len(fo)} gpu {len(fg)} mismatched {nbad} | y rel err {worst:.2e} | dpwr {dp:.2e} dnf {dn:.2e} dppm {dq:.2e} | "
          f"counter-mismatch chans {cn} | oracle {t_or:.2f}s gpu wall {t_gpu:.3f}s | K1 {st['chanfir_ms']:.3f} K2 {st['phase_ms']:.3f} K3 {st['sync_ms']:.3f} "
          f"K4 {st['walk_ms']:.3f} K5 {st['burst_ms']:.3f} ms | chan-samples/s in K1 {st['chan_samples'] / max(st['chanfir_ms'], 1e-9) * 1e3:.3e}")
    if cn:
        for c in range(C):
            a = list(o.counters(c).values()); b = list(rx.counters(c).values())
            if a != b:
                print('   counters chan', c, a, b); break
    rx.close()
    return len(fo) == len(fg) and nbad == 0


if __name__ == '__main__':
    cf = 136975000
    wav = np.fromfile(os.path.join(ROOT, 'tests/golden/vdl2_model_16b_1050kHz.wav'), dtype=np.uint8)
    wav = wav[:wav.size - wav.size % 4]
    compare('golden wav os10 1ch', wav, cf, [cf], 10, 1)
    compare('golden wav chunked', wav, cf, [cf], 10, 1, chunks=(777, 40000))
    for nm, cfg, ch in [('c2 1s', workloads.config2(1.0), None), ('c2 1s chunked', workloads.config2(1.0), (5000, 300000)),
                        ('c3 0.5s', workloads.config3(0.5), None), ('c4 0.4s', workloads.config4(0.4), None),
                        ('c2 4s', workloads.config2(4.0), None)]:
        iq, _ = synth.synthesize(cfg)
        compare(nm, iq.view(np.uint8), cfg.centerfreq, list(cfg.freqs), cfg.oversample, 1, cfg.rx_max_ppm, chunks=ch)
    cfg = synth.SynthConfig(centerfreq=cf, freqs=[cf + 30000, cf - 60000], oversample=13, duration_s=0.6, seed=9)
    iq, _ = synth.synthesize(cfg)
    compare('os13 2ch', iq.view(np.uint8), cf, list(cfg.freqs), 13, 1)
    cfg = synth.SynthConfig(centerfreq=cf, freqs=[cf + 30000], oversample=16, duration_s=0.6, seed=10)
    iq, _ = synth.synthesize(cfg)
    compare('os16 generic 1ch', iq.view(np.uint8), cf, list(cfg.freqs), 16, 1)
    cfg = synth.SynthConfig(centerfreq=cf, freqs=[cf, cf + 40000], oversample=10, duration_s=0.6, seed=12, amplitude=0.3, noise_sigma=0.01)
    iq8, _ = synth.synthesize(cfg, dtype=np.uint8)
    compare('u8 os10 2ch', iq8, cf, list(cfg.freqs), 10, 0)
