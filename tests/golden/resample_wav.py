"""Reference-held input for the 2.1 MS/s path (oversample 20: the channeliser instantiation the headline number is quoted on)
and for the u8 path (process_buf_uchar, src/demod.c:339-354).

The reference's only test vector, test/vdl2_model_16b_1050kHz.wav (copy: tests/golden/), is 1.05 MS/s cs16.  This script
derives two more captures from it - in float64, nothing but resampling / re-quantising, so what a receiver must get out of them
stays reference-held: the two frames the reference's CI greps for (.github/workflows/build.yml:16-18), FCS-good, 314 and 186
octets, S:0 L:504 F:0, the burst at -9.84 dBFS and the carrier offset of SURVEY 4 (-0.0705 ppm).

  upsampled2x(delta_hz)   the capture at 2.1 MS/s: band-limited interpolation by 2 (zero-padding of the spectrum: every original
                          sample is kept, the new ones lie on the only band-limited curve through them), optionally moved up by
                          delta_hz at the new rate (as shift_wav.py does at the old one), re-quantised to cs16.  Decode with
                          --oversample 20 (SDRplay / SoapySDR rate: src/sdrplay.h:22, src/soapysdr.h:23), centerfreq =
                          CHANNEL - delta_hz.
  as_u8(raw_cs16)         a cs16 capture re-quantised to the RTL-SDR's offset-binary u8: round(x / 256 + 127.5), i.e. the value
                          process_buf_uchar()'s table maps back to (x / 256) / 127.5 - the burst reads 20 log10(32768 / 32640)
                          = +0.034 dB louder than in cs16, nothing else changes.

The 44-byte RIFF header is 11 complex samples to the reference (dumpvdl2.c:353-356) and is treated like the rest.

Usage (the tests call the functions directly; nothing is written to disk):
    python tests/golden/resample_wav.py up2 <delta_hz> <out.cs16>
    python tests/golden/resample_wav.py u8 <out.cu8>
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
WAV = os.path.join(HERE, "vdl2_model_16b_1050kHz.wav")
FS = 1050000
FS2 = 2100000
CHANNEL = 136975000
U8_GAIN_DB = 20 * np.log10(32768.0 / 32640.0)      # (x/256)/127.5 against x/32768
# offsets of the 2.1 MS/s copies: on the centre (no mixing), and two of shift_wav.py's (one of them not a multiple of the fp32
# frequency grid of demod.c:385)
DELTAS2 = (0, 25000, -250000, 100008)


def _complex(raw):
    v = raw[:raw.size & ~3].view("<i2").astype(np.float64)
    return v[0::2] + 1j * v[1::2]


def _cs16(y):
    out = np.empty(2 * y.size, dtype="<i2")
    out[0::2] = np.clip(np.rint(y.real), -32768, 32767)
    out[1::2] = np.clip(np.rint(y.imag), -32768, 32767)
    return out.view(np.uint8)


def upsampled2x(delta_hz=0, raw=None):
    """-> uint8 array (cs16, 2.1 MS/s): decode with oversample 20, centerfreq = CHANNEL - delta_hz, freq = CHANNEL."""
    raw = np.fromfile(WAV, dtype=np.uint8) if raw is None else np.asarray(raw, dtype=np.uint8)
    x = _complex(raw)
    n = x.size
    X = np.fft.fft(x)
    Y = np.zeros(2 * n, dtype=np.complex128)
    h = n // 2
    Y[:h] = X[:h]
    Y[2 * n - (n - h):] = X[h:]
    if n % 2 == 0:                                  # the Nyquist bin belongs to both halves
        Y[h] = 0.5 * X[h]
        Y[2 * n - h] = 0.5 * X[h]
    y = np.fft.ifft(Y) * 2.0
    if delta_hz:
        k = np.arange(y.size, dtype=np.float64)
        y = y * np.exp(2j * np.pi * ((delta_hz / FS2 * k) % 1.0))
    return _cs16(y)


def as_u8(raw_cs16=None):
    """-> uint8 array (cu8, same rate): decode with --sample-format U8."""
    raw = np.fromfile(WAV, dtype=np.uint8) if raw_cs16 is None else np.asarray(raw_cs16, dtype=np.uint8)
    v = raw[:raw.size & ~3].view("<i2").astype(np.float64)
    return np.clip(np.rint(v / 256.0 + 127.5), 0, 255).astype(np.uint8)


if __name__ == "__main__":
    if sys.argv[1] == "up2":
        upsampled2x(int(sys.argv[2])).tofile(sys.argv[3])
    else:
        as_u8().tofile(sys.argv[2])
