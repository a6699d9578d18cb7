"""The device-side SplitMix64 generator (torch int64 ops) reproduces the oracle's stream bit for bit."""
import numpy as np
import torch

import oracle
from armadillocudalinearinterpolation_amd import synth


def test_splitmix_torch_matches_oracle():
    for seed in (0, 0x5EED0003, 0xFFFFFFFFFFFFFFF1):
        a = synth.splitmix_uniform(seed, 100003, torch.device("cpu"), chunk=4096).numpy()
        assert np.array_equal(a, oracle.splitmix_uniform(seed, 100003))
    assert a.min() >= 0.0 and a.max() < 1.0
