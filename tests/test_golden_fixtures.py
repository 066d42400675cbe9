"""The oracle, re-run on regenerated seeded inputs, reproduces the committed golden answers
(guards the generator, the oracle and the fixtures against drifting apart)."""
import numpy as np
import pytest

import cases


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_oracle_reproduces_golden(oracle_mod, name):
    cfg, iq, bursts, gold = cases.load(name)
    o = oracle_mod.Oracle(cfg.centerfreq, list(cfg.freqs), oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
    o.process(iq.view(np.uint8), block_bytes=320000, nthreads=2)     # the reference's FILE_BUFSIZE blocks
    cases.check_against_golden(o.frames(), [list(o.counters(c).values()) for c in range(len(cfg.freqs))], gold, 1e-3, 1e-4, name)
    assert gold["n_tx_bursts"] == len(bursts)
