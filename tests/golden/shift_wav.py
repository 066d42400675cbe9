"""Reference-held input for the offset-tuned branch (sincosf_lut + multiply + downmix_phi, src/demod.c:58-72,200-203,312-317,385).

The reference's only test vector, test/vdl2_model_16b_1050kHz.wav (copy: tests/golden/), sits on the centre frequency, so a
receiver that decodes it never runs its NCO.  This script moves the same capture to another place in the band - every complex
sample times exp(+j 2 pi delta n / fs) in float64, re-quantised to cs16 - so that a receiver tuned `delta` below the channel
(centerfreq = f - delta) has to mix it back down.  The expected answer stays reference-held: the two frames whose byte
patterns the reference's CI greps for (.github/workflows/build.yml:16-18), FCS-good, 314 and 186 octets, S:0 L:504 F:0.
A receiver whose NCO had the wrong sign, scale or phase-advance rule would mix the burst further away and decode nothing.

The 44-byte RIFF header is 11 complex samples to the reference (dumpvdl2.c:353-356) and is rotated like the rest.

Usage (the tests call shifted() directly; nothing is written to disk):
    python tests/golden/shift_wav.py <delta_hz> <out.cs16>
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
WAV = os.path.join(HERE, "vdl2_model_16b_1050kHz.wav")
FS = 1050000                       # --iq-file default: oversample 10 (dumpvdl2.c:840-845, dumpvdl2.h:49)
CHANNEL = 136975000                # the CLI's default frequency (dumpvdl2.c:1066-1070)
# offsets in Hz: one small, one that makes (float)centerfreq - (float)freq differ from the true offset (fp32 spacing at 137 MHz
# is 16 Hz, demod.c:385 / SURVEY A-4), one negative and large, one near the band edge
DELTAS = (25000, -250000, 100008, -412500)


def shifted(delta_hz, raw=None):
    """-> uint8 array: the capture moved up by delta_hz; decode it with centerfreq = CHANNEL - delta_hz, freq = CHANNEL."""
    raw = np.fromfile(WAV, dtype=np.uint8) if raw is None else np.asarray(raw, dtype=np.uint8)
    v = raw[:raw.size & ~3].view("<i2").astype(np.float64)
    x = v[0::2] + 1j * v[1::2]
    n = np.arange(x.size, dtype=np.float64)
    y = x * np.exp(2j * np.pi * ((delta_hz / FS * n) % 1.0))
    out = np.empty(2 * x.size, dtype="<i2")
    out[0::2] = np.clip(np.rint(y.real), -32768, 32767)
    out[1::2] = np.clip(np.rint(y.imag), -32768, 32767)
    return out.view(np.uint8)


if __name__ == "__main__":
    shifted(int(sys.argv[1])).tofile(sys.argv[2])
