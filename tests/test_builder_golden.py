"""Both tree builders against the REFERENCE's answers for the adversarial sequences of tests/builder_cases.py
(tests/golden/builder_adversarial.npz: parent arrays from the unmodified reference's MinMatch): the host builder on
any machine, the device builder on the GPU -- not only against each other (tests/test_builder_gpu.py)."""
import os

import numpy as np
import pytest

import builder_cases
from relate_amd import api

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "builder_adversarial.npz"))


@pytest.mark.parametrize("name", sorted(builder_cases.CASES))
def test_host_builder_gives_the_reference_trees(name):
    N, mats = builder_cases.CASES[name]()
    b = api.Builder(N)
    for t, (d, prior) in enumerate(mats):
        assert np.array_equal(b.build(d, prior)[0], GOLD[name][t]), (name, t)
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(builder_cases.CASES))
def test_device_builder_gives_the_reference_trees(name):
    N, mats = builder_cases.CASES[name]()
    b = api.Builder(N, device=0)
    on_gpu = 0
    for t, (d, prior) in enumerate(mats):
        assert np.array_equal(b.build(d, prior)[0], GOLD[name][t]), (name, t)
        on_gpu += b.last_on_gpu
    b.close()
    # (a tree whose lists the kernel cannot hold -- the flat matrix's -- is the host's, from the carried state)
    assert on_gpu >= 1
