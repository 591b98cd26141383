"""Greedy search timing at BASELINE configs 1/3 shapes (B1: N=65536, B3: N=1.5M; Dt=61, Dj=151, me=6)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
for N in (65536, 1500000):
    Dt, Dj, T, me = 61, 151, 600, 6
    F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
    wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
    eng = snickery_amd.HipSearchEngine(0)
    eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj); eng.set_greedy_layout(me, False, 0)
    U = synthetic_targets(F_unw, T, seed=1) * wt
    eng.greedy(U)
    eng.reset_timers()
    t0 = time.time()
    for _ in range(3):
        p = eng.greedy(U)
    dt = (time.time() - t0) / 3
    steps = T // me
    bytes_per_step = N * (Dj + Dt) * 4.0
    tm = eng.timers()['greedy_steps']
    print('N=%d: %.2f ms/utt (%d steps, %.1f us/step) -> %.0f frames/s; scan %.2f TB/s of HBM (algorithmic %.0f MB/step)' % (
        N, dt * 1e3, steps, tm[0] / tm[1] / steps * 1e3, T / dt, bytes_per_step / (tm[0] / tm[1] / steps * 1e-3) / 1e12, bytes_per_step / 1e6))
    eng.close()
