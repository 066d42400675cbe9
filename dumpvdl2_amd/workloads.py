"""Named synthetic workloads = BASELINE.json configs[1..4], as SynthConfig factories.

Channel plans differ from SURVEY's first sketch where that sketch could not decode: with the
reference's 2-pole input filter a neighbour 25-50 kHz away leaks at only -20..-32 dB and the
phase-only preamble detector (demod.c:105-198) locks onto the leaked unique word, so the plans
below keep simultaneously active carriers >= 100 kHz apart (spacing or time-division slots),
and the dense plans run the receiver with the reference's own --max-ppm gate (demod.c:190-192),
which drops the locks an idle channel takes on a neighbour's burst (apparent offset =
(df mod 10.5 kHz) >> a few ppm).
"""
from .synth import SynthConfig, channel_plan

CENTER = 136975000


def config2(duration_s=16.0, seed=20260926 + 2):
    """8 channels, 2.1 MS/s cs16 - the configuration the headline metric is quoted on."""
    return SynthConfig(centerfreq=CENTER, freqs=channel_plan(8, CENTER, 100000), oversample=20,
                       duration_s=duration_s, seed=seed, amplitude=0.05, noise_sigma=0.002)


def config3(duration_s=16.0, seed=20260926 + 3):
    """64 channels at 25 kHz, 4 time-division slots (active carriers >= 100 kHz apart)."""
    return SynthConfig(centerfreq=CENTER, freqs=channel_plan(64, CENTER, 25000), oversample=20,
                       duration_s=duration_s, seed=seed, amplitude=0.02, noise_sigma=0.002,
                       tdm_slots=4, tdm_slot_s=0.3, rx_max_ppm=5.0)


def config4(duration_s=16.0, seed=20260926 + 4, error_injection=False):
    """256 channels at 8 kHz, 16 slots of 0.1 s (active carriers >= 128 kHz apart), short payloads."""
    return SynthConfig(centerfreq=CENTER, freqs=channel_plan(256, CENTER, 8000), oversample=20,
                       duration_s=duration_s, seed=seed, amplitude=0.01, noise_sigma=0.0005,
                       tdm_slots=16, tdm_slot_s=0.1, max_payload=300, mean_gap_s=0.05,
                       max_ppm=0.5, rx_max_ppm=2.5,
                       error_injection=error_injection)


def config5(duration_s=16.0, seed=20260926 + 5):
    """config4 with injected RS byte errors and header bit flips (FEC-heavy path)."""
    return config4(duration_s, seed, error_injection=True)


def config4_bursty(duration_s=16.0, seed=20260926 + 14):
    """Back-end stress: config4's channel plan and air time, but cut into four times as many, four times shorter bursts
    (payloads <= 60 octets, 4 ms mean gap): ~4x the synchronisations, headers, burst descriptors and frames per second that
    the walker (K4), the noise-floor replay (K4b) and the burst decoder (K5) have to get through per block."""
    return SynthConfig(centerfreq=CENTER, freqs=channel_plan(256, CENTER, 8000), oversample=20,
                       duration_s=duration_s, seed=seed, amplitude=0.01, noise_sigma=0.0005,
                       tdm_slots=16, tdm_slot_s=0.1, tdm_pack=True, min_payload=12, max_payload=60, max_frames=2, mean_gap_s=0.004,
                       max_ppm=0.5, rx_max_ppm=2.5)
