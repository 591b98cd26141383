"""One B3 utterance (N = 1.5 M, me 6, 100 steps) through the hoisted float32 scan, for rocprofv3."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500000
Dt, Dj, T, me = 61, 151, 600, 6
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj); eng.set_greedy_layout(me, False, 0)
eng.set_option('greedy_mode', 1)
U = synthetic_targets(F_unw, T, seed=1) * wt
for _ in range(4):
    eng.greedy(U)
eng.close()
