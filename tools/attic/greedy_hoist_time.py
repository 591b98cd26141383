"""Greedy search with the hoisted target term (greedy_hoist_kernels.hip) against the scans that compute it per step:
B1 / B3 shapes (N = 65 536 / 1.5 M; Dt 61, Dj 151, me 6), one utterance and a batch of six."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
sizes = [int(x) for x in sys.argv[1:]] or [65536, 1500000]
for N in sizes:
    Dt, Dj, T, me = 61, 151, 600, 6
    F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
    wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
    eng = snickery_amd.HipSearchEngine(0)
    eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj); eng.set_greedy_layout(me, False, 0)
    U = synthetic_targets(F_unw, T, seed=1) * wt
    Us = [synthetic_targets(F_unw, T - 6 * i, seed=2 + i) * wt for i in range(6)]
    steps = T // me
    ref = None
    for name, mode, hoist, f16 in (('exact scan', 0, 0, 0), ('float32 scan', 1, 0, 0), ('float32 scan, hoisted', 1, 1, 0), ('hoisted, float16 tiles', 1, 1, 2)):
        eng.set_option('greedy_mode', mode); eng.set_option('greedy_hoist', hoist); eng.set_option('greedy_f16', f16)
        p, d = eng.greedy(U, return_distances=True)
        if ref is None:
            ref = (p, d)
        assert np.array_equal(p, ref[0]) and np.array_equal(d, ref[1]), name
        eng.reset_timers()
        x0 = eng.info('greedy_exact_windows')
        t0 = time.time()
        for _ in range(3):
            eng.greedy(U)
        dt = (time.time() - t0) / 3
        tm = eng.timers()['greedy_steps']
        us = tm[0] / tm[1] / steps * 1e3
        print('N=%d %-24s one utterance: %.2f ms (%.1f us/step) %.0f frames/s; algorithmic (Dj+1)4N: %.2f TB/s; exact windows/step %.2f; fallbacks %d' % (
            N, name[:24], dt * 1e3, us, T / dt, N * (Dj + 1) * 4.0 / (us * 1e-6) / 1e12,
            (eng.info('greedy_exact_windows') - x0) / 3.0 / steps, eng.info('greedy_fallbacks')), flush=True)
    bref = None
    for name, mode, hoist, f16 in (('exact scan', 0, 0, 0), ('float32 scan', 2, 0, 0), ('float32 scan, hoisted', 2, 1, 0), ('hoisted, float16 tiles', 2, 1, 2)):
        eng.set_option('greedy_mode', mode); eng.set_option('greedy_hoist', hoist); eng.set_option('greedy_f16', f16)
        r = eng.greedy_batch(Us)
        if bref is None:
            bref = r
        assert all(np.array_equal(a, b) for a, b in zip(r, bref)), name
        t0 = time.time()
        for _ in range(2):
            eng.greedy_batch(Us)
        dt = (time.time() - t0) / 2
        print('N=%d %-24s six utterances: %.1f ms -> %.0f frames/s' % (N, name, dt * 1e3, sum(u.shape[0] for u in Us) / dt), flush=True)
    eng.close()
