"""Small driver for PMC passes over the greedy step kernel: one 600-frame utterance at N = 1.5 M (100 steps)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
N, Dt, Dj, T, me = 1500000, 61, 151, 600, 6
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(np.full(Dt, 0.4), np.full(Dj, 0.05)); eng.set_greedy_layout(me, False, 0)
U = synthetic_targets(F_unw, T, seed=1) * 0.4
eng.greedy(U)
eng.close()
