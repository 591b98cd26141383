"""Deterministic stand-in for a Synthesiser in 'stream_weight_balancing' mode: per-stream scores
that grow with the square of the stream weight, fixed per utterance name.  Used twice: by
tools/make_golden.py to drive the REFERENCE's balance_stream_weights.py (its loop is host logic
only), and by tests/test_hostprep.py to drive ours against the recorded trajectory."""
import zlib
import numpy as np


class StubSynthesiser(object):
    def __init__(self, config_fname=None):
        self.config = {'join_cost_weight': 1.0, 'greedy_search': False}
        self.stream_list_join = ['mag', 'real', 'imag', 'lf0']
        self.stream_list_target = ['mag', 'lf0']
        self.mode_of_operation = 'normal'
        self.verbose = True
        self.wj = np.ones(4)
        self.wt = np.ones(2)

    def get_sentence_set(self, set_name):
        assert set_name == 'tune'
        return ['utt_%02d' % i for i in range(14)]

    def set_join_weights(self, weights):
        self.wj = np.array(weights, dtype=float)

    def set_target_weights(self, weights):
        self.wt = np.array(weights, dtype=float)

    def get_tree_for_greedy_search(self):
        raise AssertionError('not a greedy configuration')

    def synth_utt(self, fname, synth_type='tune'):
        assert self.mode_of_operation == 'stream_weight_balancing' and synth_type == 'tune'
        rng = np.random.RandomState(zlib.crc32(fname.encode()) % (2 ** 31))
        L = int(rng.randint(20, 40))
        jbase = np.abs(rng.randn(L - 1, 4)) * np.array([3.0, 0.7, 1.9, 0.2])
        tbase = np.abs(rng.randn(L, 2)) * np.array([5.0, 0.4])
        jbase[rng.rand(L - 1, 4) < 0.15] = 0.0          # natural joins contribute exactly 0
        return (tbase * self.wt ** 2, jbase * self.wj ** 2)
