"""Synthetic VDL Mode 2 transmitter: AVLC frames -> D8PSK bursts -> wideband cs16 IQ.

The reference (dumpvdl2) is a receiver only and has no modulator, so this is
original code.  The burst format is the inverse of what the reference decodes
(reference file:line in each function).  It is used by the
tests and by bench.py to make seeded input with known ground truth (the list
of transmitted frames), so a decode can be checked without any oracle.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Sequence

import numpy as np

SYMBOL_RATE = 10500          # dumpvdl2.h:46
SPS = 10                     # dumpvdl2.h:44
RS_K, RS_N = 249, 255        # dumpvdl2.h:37-38
LFSR_IV = 0x6959             # decode.c:50
GRAY = np.array([0, 1, 3, 2, 6, 7, 5, 4], dtype=np.int64)   # demod.c:223  (phase step index -> 3 bits)
GRAY_INV = np.argsort(GRAY)                                    # 3-bit value -> phase step index
# phase steps (units of pi/4) of the 16-symbol unique word; their running sum is demod.c:107-124
PREAMBLE_STEPS = np.array([0, 3, 2, 4, 0, 1, 6, 4, 1, 7, 2, 5, 6, 5, 7, 3], dtype=np.int64)
HDR_H = [0x001FFF0, 0x07E1FE8, 0x18E61E4, 0x1B6A662, 0x0D3CAA1]   # decode.c:55-61


# --------------------------------------------------------------------------
# GF(2^8), poly 0x187, RS(255,249) generator roots alpha^120..alpha^125 (rs.c:28)
# --------------------------------------------------------------------------
def _gf_tables():
    exp = np.zeros(512, dtype=np.int64)
    log = np.zeros(256, dtype=np.int64)
    x = 1
    for i in range(255):
        exp[i] = x
        log[x] = i
        x <<= 1
        if x & 0x100:
            x ^= 0x187
    exp[255:510] = exp[0:255]
    return exp, log


_EXP, _LOG = _gf_tables()


def _gf_mul(a: int, b: int) -> int:
    if a == 0 or b == 0:
        return 0
    return int(_EXP[_LOG[a] + _LOG[b]])


def _rs_generator():
    g = [1]
    for i in range(6):
        r = int(_EXP[120 + i])
        ng = [0] * (len(g) + 1)
        for j, c in enumerate(g):          # (x + r) * g(x), g[j] = coeff of x^j
            ng[j + 1] ^= c
            ng[j] ^= _gf_mul(c, r)
        g = ng
    return g                                # degree 6, g[6] == 1


_RS_GEN = _rs_generator()


def rs_parity(block: Sequence[int]) -> List[int]:
    """6 parity symbols of a 249-symbol block; block[0] is the highest-degree coefficient
    (libfec/decode_rs.h:82-93 evaluates data[0] first)."""
    assert len(block) == RS_K
    rem = [0] * 6                           # rem[0] <-> x^5
    for d in block:
        fb = d ^ rem[0]
        rem = [rem[j + 1] ^ _gf_mul(fb, _RS_GEN[5 - j]) for j in range(5)] + [_gf_mul(fb, _RS_GEN[0])]
    return rem


def crc16_x25(data: bytes) -> int:
    """AVLC FCS register (reflected CCITT, init 0xFFFF) - avlc.c:177, crc.c:21-64."""
    c = 0xFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ 0x8408 if c & 1 else c >> 1
    return c


def make_avlc_frame(body: bytes) -> bytes:
    """body (addresses + LCF + payload) followed by the FCS, so that the receiver's residual is 0xF0B8."""
    fcs = crc16_x25(body) ^ 0xFFFF
    return body + bytes([fcs & 0xFF, fcs >> 8])


def fec_octets_for(n: int) -> int:          # decode.c:124-133
    return 0 if n < 3 else 2 if n < 31 else 4 if n < 68 else 6


def scramble_sequence(nbits: int) -> np.ndarray:
    """PRBS of bitstream.c:94-107 from the fixed IV."""
    out = np.zeros(nbits, dtype=np.uint8)
    l = LFSR_IV
    for i in range(nbits):
        bit = (l ^ (l >> 14)) & 1
        l = (l >> 1) | (bit << 14)
        out[i] = bit
    return out


_PRBS = scramble_sequence(17000)


def hdlc_bits(frames: Sequence[bytes], separate_flags: bool = False) -> np.ndarray:
    """Flag, then each frame LSB-first with zero insertion after five ones, flag after each
    frame (inverse of bitstream.c:109-150)."""
    flag = [0, 1, 1, 1, 1, 1, 1, 0]
    bits: List[int] = list(flag)
    for fi, fr in enumerate(frames):
        if separate_flags and fi > 0:
            bits += flag
        ones = 0
        for byte in fr:
            for j in range(8):
                b = (byte >> j) & 1
                bits.append(b)
                if b:
                    ones += 1
                    if ones == 5:
                        bits.append(0)
                        ones = 0
                else:
                    ones = 0
        bits += flag
    return np.array(bits, dtype=np.uint8)


@dataclasses.dataclass
class BurstBits:
    symbols: np.ndarray          # phase-step indices 0..7, one per symbol after the unique word
    tl_bits: int
    datalen_octets: int
    num_blocks: int
    injected_byte_errors: List[int]      # per RS block
    header_flips: int
    decodable: bool                       # False when an RS block was pushed past its capacity


def build_burst(frames: Sequence[bytes], rng: Optional[np.random.Generator] = None,
                byte_errors_per_block: Optional[Sequence[int]] = None, header_flips: int = 0,
                separate_flags: bool = False, raw_bits: Optional[np.ndarray] = None) -> BurstBits:
    """AVLC frames -> header + interleaved data + FEC, scrambled, as D8PSK phase steps.
    raw_bits (tests only) replaces the HDLC-framed bit string, e.g. to send malformed framing."""
    hb = hdlc_bits(frames, separate_flags) if raw_bits is None else np.asarray(raw_bits, dtype=np.uint8)
    tl = int(hb.size)
    assert tl <= 0x3FFF, "transmission length field limited by decode.c:45"
    noct = (tl + 7) // 8
    padded = np.zeros(noct * 8, dtype=np.uint8)
    padded[:tl] = hb
    data = np.packbits(padded.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).astype(np.int64)
    nblk = (noct + RS_K - 1) // RS_K
    last = noct - (nblk - 1) * RS_K                     # 1..249
    rows = []
    for r in range(nblk):
        blk = data[r * RS_K:(r + 1) * RS_K]
        full = np.zeros(RS_K, dtype=np.int64)
        full[:blk.size] = blk
        npar = 6 if r < nblk - 1 else fec_octets_for(last)
        par = rs_parity(list(full))[:npar]
        rows.append((list(blk), par))
    # channel errors are injected on the coded block (data + transmitted parity) before interleaving
    injected = [0] * nblk
    decodable = True
    if byte_errors_per_block is not None:
        assert rng is not None
        for r in range(nblk):
            k = int(byte_errors_per_block[r]) if r < len(byte_errors_per_block) else 0
            blk, par = rows[r]
            n = len(blk) + len(par)
            k = min(k, n)
            if k:
                pos = rng.choice(n, size=k, replace=False)
                for p in pos:
                    e = int(rng.integers(1, 256))
                    if p < len(blk):
                        blk[p] ^= e
                    else:
                        par[p - len(blk)] ^= e
            injected[r] = k
            if k > len(par) // 2:
                decodable = False
    # interleave: column-major over rows (inverse of decode.c:135-163)
    dcol: List[int] = []
    for col in range(RS_K):
        for r in range(nblk):
            if col < len(rows[r][0]):
                dcol.append(rows[r][0][col])
    fcol: List[int] = []
    for col in range(6):
        for r in range(nblk):
            if col < len(rows[r][1]):
                fcol.append(rows[r][1][col])
    assert len(dcol) == noct
    octs = np.array(dcol + fcol, dtype=np.uint8)
    body = np.unpackbits(octs.reshape(-1, 1), axis=1, bitorder="little").reshape(-1)
    # header: 3 reserved zeros, TL LSB-first in 17 bits, 5 parity bits (decode.c:209-222, 55-61)
    tl_rev = 0
    for i in range(17):
        if tl & (1 << i):
            tl_rev |= 1 << (16 - i)
    word = tl_rev << 5
    for i in range(5):
        if bin(word & HDR_H[i] & ~0x1F).count("1") & 1:
            word |= 1 << (4 - i)
    hdr = np.array([(word >> (24 - i)) & 1 for i in range(25)], dtype=np.uint8)
    if header_flips:
        assert rng is not None
        for p in rng.choice(np.arange(3, 25), size=header_flips, replace=False):
            hdr[p] ^= 1
    bits = np.concatenate([hdr, body])
    bits ^= _PRBS[:bits.size]
    pad = (-bits.size) % 3
    if pad:
        bits = np.concatenate([bits, np.zeros(pad, dtype=np.uint8)])
    tri = bits.reshape(-1, 3).astype(np.int64)
    g = tri[:, 0] * 4 + tri[:, 1] * 2 + tri[:, 2]
    return BurstBits(GRAY_INV[g], tl, noct, nblk, injected, header_flips, decodable)


# --------------------------------------------------------------------------
# waveform
# --------------------------------------------------------------------------
def raised_cosine(sps: int, alpha: float = 0.6, span: int = 4) -> np.ndarray:
    t = np.arange(-span * sps, span * sps + 1, dtype=np.float64) / sps
    with np.errstate(divide="ignore", invalid="ignore"):
        h = np.sinc(t) * np.cos(np.pi * alpha * t) / (1.0 - (2.0 * alpha * t) ** 2)
    sing = np.isclose(np.abs(2.0 * alpha * t), 1.0)
    h[sing] = (np.pi / 4.0) * np.sinc(1.0 / (2.0 * alpha))
    return h


def modulate(steps: np.ndarray, samples_per_symbol: int, ramp_symbols: int = 5,
             start_phase: float = 0.0) -> np.ndarray:
    """Complex baseband (unit symbol amplitude): ramp-up symbols at constant phase, unique word, payload."""
    all_steps = np.concatenate([np.zeros(ramp_symbols, dtype=np.int64), PREAMBLE_STEPS, steps])
    theta = start_phase + np.cumsum(all_steps) * (np.pi / 4.0)
    a = np.exp(1j * theta)
    span = 4
    p = raised_cosine(samples_per_symbol, 0.6, span)
    n = (a.size + 2 * span) * samples_per_symbol + 1
    up = np.zeros(n, dtype=np.complex128)        # impulse train, one symbol every sps samples
    up[span * samples_per_symbol + np.arange(a.size) * samples_per_symbol] = a
    return _fftconv_same(up, p)


def _fftconv_same(x: np.ndarray, h: np.ndarray) -> np.ndarray:
    n = x.size + h.size - 1
    nf = 1 << (n - 1).bit_length()
    y = np.fft.ifft(np.fft.fft(x, nf) * np.fft.fft(h, nf))[:n]
    off = (h.size - 1) // 2
    return y[off:off + x.size]


@dataclasses.dataclass
class TxBurst:
    chan: int
    start_sample: int            # first IQ sample the waveform touches
    frames: List[bytes]
    tl_bits: int
    datalen_octets: int
    injected_byte_errors: List[int]
    header_flips: int
    decodable: bool
    cfo_hz: float


@dataclasses.dataclass
class SynthConfig:
    centerfreq: int = 136975000
    freqs: Sequence[int] = (136975000,)
    oversample: int = 20
    duration_s: float = 1.0
    seed: int = 20260926
    amplitude: float = 0.05
    noise_sigma: float = 0.002
    mean_gap_s: float = 0.150
    min_payload: int = 20
    max_payload: int = 1000
    max_frames: int = 3
    max_ppm: float = 2.0                 # transmitter carrier offset, uniform +-ppm
    rx_max_ppm: float = 0.0              # receiver --max-ppm gate to use with this workload (demod.c:192); 0 = off
    tdm_slots: int = 0                   # >0: channel k transmits only in slot k % tdm_slots
    tdm_slot_s: float = 0.2
    tdm_pack: bool = False               # several bursts may follow each other inside the channel's slot (default: one per slot)
    error_injection: bool = False        # config 5: RS byte errors + header bit flips
    invalid_frame_rate: float = 0.0      # fraction of frames that are not valid AVLC (bad FCS or < 11 octets)
    first_burst_s: float = 0.02

    @property
    def sample_rate(self) -> int:
        return SYMBOL_RATE * SPS * self.oversample


def channel_plan(nchan: int, centerfreq: int = 136975000, spacing: int = 25000) -> List[int]:
    """Channel frequencies centred on centerfreq, offset by half a step so none is on-centre
    (SURVEY.md 8.5 config 2-4)."""
    return [centerfreq + (k - nchan // 2) * spacing + spacing // 2 for k in range(nchan)]


def _random_frames(rng: np.random.Generator, cfg: SynthConfig) -> List[bytes]:
    nfr = int(rng.integers(1, cfg.max_frames + 1))
    total = int(rng.integers(cfg.min_payload, cfg.max_payload + 1))
    cuts = sorted(rng.choice(np.arange(1, total), size=nfr - 1, replace=False).tolist()) if nfr > 1 and total > nfr else []
    sizes = [b - a for a, b in zip([0] + cuts, cuts + [total])]
    frames = []
    for s in sizes:
        body = rng.integers(0, 256, size=max(s, 9), dtype=np.uint8).tobytes()
        if cfg.invalid_frame_rate > 0 and rng.random() < cfg.invalid_frame_rate:
            # what avlc_parse() refuses (avlc.c:168-187): a frame shorter than 11 octets, or one whose FCS does not check
            frames.append(body[:int(rng.integers(1, 11))] if rng.random() < 0.4 else body + b"\x00\x00")
            continue
        frames.append(make_avlc_frame(body))
    return frames


def synthesize(cfg: SynthConfig, dtype=np.int16):
    """Return (iq, bursts): iq is interleaved I,Q int16 (2*N values), bursts the ground truth."""
    fs = cfg.sample_rate
    n = int(round(cfg.duration_s * fs))
    n -= n % 2
    sps = SPS * cfg.oversample
    acc = np.zeros(n, dtype=np.complex64)
    bursts: List[TxBurst] = []
    master = np.random.default_rng(cfg.seed)
    for k, f in enumerate(cfg.freqs):
        rng = np.random.default_rng(master.integers(0, 2 ** 63))
        t = cfg.first_burst_s + float(rng.exponential(cfg.mean_gap_s))
        while True:
            frames = _random_frames(rng, cfg)
            errs = None
            hflips = 0
            if cfg.error_injection:
                probe = build_burst(frames)
                errs = []
                for r in range(probe.num_blocks):
                    last = probe.datalen_octets - (probe.num_blocks - 1) * RS_K
                    npar = 6 if r < probe.num_blocks - 1 else fec_octets_for(last)
                    tcap = npar // 2
                    errs.append(tcap + 1 if (rng.random() < 0.05 and npar > 0) else int(rng.integers(0, tcap + 1)))
                u = rng.random()
                hflips = 2 if u < 0.02 else 1 if u < 0.12 else 0
            bb = build_burst(frames, rng, errs, hflips)
            wave = None
            if cfg.tdm_slots > 0:
                period = cfg.tdm_slots * cfg.tdm_slot_s
                slot0 = (k % cfg.tdm_slots) * cfg.tdm_slot_s
                stay = False
                if cfg.tdm_pack:
                    wave = modulate(bb.symbols, sps, start_phase=float(rng.uniform(0, 2 * np.pi)))
                    into = (t - slot0) % period              # position inside the channel's own period
                    stay = t >= slot0 and 0.002 <= into and into + wave.size / fs + 0.002 <= cfg.tdm_slot_s
                if not stay:
                    m = np.ceil((t - slot0) / period)
                    t = slot0 + max(m, 0) * period + 0.002
            if wave is None:
                wave = modulate(bb.symbols, sps, start_phase=float(rng.uniform(0, 2 * np.pi)))
            start = int(round(t * fs))
            if start + wave.size >= n:
                break
            cfo = float(rng.uniform(-cfg.max_ppm, cfg.max_ppm)) * 1e-6 * f
            w = 2.0 * np.pi * ((f - cfg.centerfreq) + cfo) / fs
            idx = np.arange(start, start + wave.size, dtype=np.float64)
            acc[start:start + wave.size] += (cfg.amplitude * wave * np.exp(1j * (w * idx))).astype(np.complex64)
            bursts.append(TxBurst(k, start, frames, bb.tl_bits, bb.datalen_octets, bb.injected_byte_errors,
                                  bb.header_flips,
                                  bb.decodable and (hflips == 0 or (hflips == 1 and bb.tl_bits <= 0x1FFF)), cfo))
            t = (start + wave.size) / fs + float(rng.exponential(cfg.mean_gap_s)) + 0.004
    nrng = np.random.default_rng(cfg.seed ^ 0x5EED)
    iq = np.empty(2 * n, dtype=np.float32)
    iq[0::2] = acc.real
    iq[1::2] = acc.imag
    del acc
    step = 1 << 22
    for o in range(0, 2 * n, step):     # noise in chunks keeps the peak footprint low
        m = min(step, 2 * n - o)
        iq[o:o + m] += nrng.standard_normal(m, dtype=np.float32) * np.float32(cfg.noise_sigma)
    if dtype == np.int16:
        out = np.clip(np.rint(iq * 32768.0), -32768, 32767).astype(np.int16)
    elif dtype == np.uint8:
        out = np.clip(np.rint(iq * 127.5 + 127.5), 0, 255).astype(np.uint8)
    else:
        raise ValueError("dtype must be int16 or uint8")
    return out, bursts


def expected_frames(bursts: Sequence[TxBurst]):
    """Ground truth as a sorted list of (chan, start_sample, idx, octets) for decodable bursts."""
    out = []
    for b in bursts:
        if not b.decodable:
            continue
        for i, fr in enumerate(b.frames):
            out.append((b.chan, b.start_sample, i, fr))
    return sorted(out)
