"""Malformed / unusual HDLC framing inside otherwise valid bursts: leading flags, back-to-back flags,
missing closing flag, seven ones, flags too early, bit counts that are not whole octets, empty tails.
The wave-parallel un-stuffer (vdl2_core.h step 5) must follow bitstream_copy_next_frame() exactly."""
import numpy as np
import pytest

from dumpvdl2_amd import synth

CF = 136975000
FLAG = [0, 1, 1, 1, 1, 1, 1, 0]


def stuffed(nbytes, rng):
    return synth.hdlc_bits([rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes()])[8:-8].tolist()


def adversarial_bitstrings(rng, n):
    out = []
    for k in range(n):
        parts = []
        nseg = int(rng.integers(1, 7))
        for s in range(nseg):
            c = int(rng.integers(0, 12))
            if c == 0: parts += FLAG
            elif c == 1: parts += FLAG + FLAG
            elif c == 2: parts += stuffed(int(rng.integers(1, 60)), rng) + FLAG
            elif c == 3: parts += stuffed(int(rng.integers(1, 40)), rng)                      # no closing flag
            elif c == 4: parts += [1] * 7                                                       # abort sequence
            elif c == 5: parts += rng.integers(0, 2, int(rng.integers(1, 7))).tolist() + FLAG  # flag too early
            elif c == 6: parts += stuffed(int(rng.integers(1, 30)), rng) + [0, 1, 0] + FLAG    # not whole octets
            elif c == 7: parts += [0] * int(rng.integers(1, 20))
            elif c == 8: parts += [1, 1, 1, 1, 1, 0] * int(rng.integers(1, 6))                 # runs of stuffed zeros
            elif c == 9: parts += FLAG[:-1] + FLAG                                              # shared zero between flags
            elif c == 10: parts += stuffed(int(rng.integers(200, 400)), rng) + FLAG
            else: parts += rng.integers(0, 2, int(rng.integers(8, 200))).tolist()
        if len(parts) < 24:
            parts += FLAG * 3
        out.append(np.array(parts, dtype=np.uint8))
    return out


def make_stream(bitstrings, os_=10, seed=1):
    rng = np.random.default_rng(seed)
    sps = 10 * os_
    chunks = []
    for bits in bitstrings:
        bb = synth.build_burst([], rng, raw_bits=bits)
        w = synth.modulate(bb.symbols, sps, start_phase=float(rng.uniform(0, 6.28)))
        gap = np.zeros(int(rng.integers(3000, 6000)), dtype=np.complex128)
        chunks += [gap, 0.25 * w]
    x = np.concatenate(chunks + [np.zeros(5000, dtype=np.complex128)])
    x = x + 0.002 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))
    iq = np.empty(2 * x.size); iq[0::2] = x.real; iq[1::2] = x.imag
    return np.clip(np.rint(iq * 32768), -32768, 32767).astype(np.int16)


@pytest.fixture(scope="module")
def stream():
    rng = np.random.default_rng(2024)
    bs = adversarial_bitstrings(rng, 150)
    return make_stream(bs), len(bs)


def test_hostsim_unstuffer_matches_oracle(oracle_mod, stream):
    import pyhostsim
    from util import assert_frames_equal
    iq, nb = stream
    o = oracle_mod.Oracle(CF, [CF], oversample=10)
    D = iq.size // 2 // 10
    tr = o.trace_all(D + 4)
    o.process(iq.view(np.uint8), block_bytes=1 << 24)
    D = o.decimated_count(0)
    hs = pyhostsim.HostSim([CF], 0.0, cap_log2=int(np.ceil(np.log2(D + 70000))))
    hs.feed(tr[:, :D, :])
    fo, fh = o.frames(), hs.frames()
    c = o.counters(0)
    assert c["demod.sync.good"] >= nb - 2
    assert c["decoder.errors.unstuff"] > 5 and c["decoder.errors.truncated_octets"] > 5 and c["decoder.msg.good"] > 30
    assert any(len(f["octets"]) == 0 for f in fo)            # zero-length frames do occur and are pushed
    assert_frames_equal(fo, fh, label="adversarial")
    assert list(c.values()) == hs.counters(0)


@pytest.mark.gpu
def test_gpu_unstuffer_matches_oracle(oracle_mod, stream):
    from dumpvdl2_amd import vdl2hip
    from util import assert_frames_equal
    iq, nb = stream
    o = oracle_mod.Oracle(CF, [CF], oversample=10)
    o.process(iq.view(np.uint8), block_bytes=1 << 24)
    rx = vdl2hip.Receiver(CF, [CF], 10, vdl2hip.FMT_S16LE, max_block_bytes=iq.nbytes)
    rx.feed(iq)
    fg = rx.drain()
    assert_frames_equal(o.frames(), fg, label="adversarial gpu")
    assert list(o.counters(0).values()) == list(rx.counters(0).values())
