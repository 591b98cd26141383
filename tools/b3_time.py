"""B3 (1.5 M units, Dt 61, Dj 151, me 6, one 600-frame utterance): time per step with the target values from the float64
pipe (greedy_hoist_fast 0) and from the bf16 pipe (1); paths compared.
    python tools/b3_time.py [units]"""
import sys, os, time
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
U = synthetic_targets(F_unw, T, seed=1) * wt
steps = T // me
ref = None
for fast in (0, 1, 0, 1):
    eng.set_option('greedy_hoist_fast', fast)
    p, d = eng.greedy(U, return_distances=True)
    if ref is None: ref = (p, d)
    same = np.array_equal(p, ref[0]) and np.array_equal(d, ref[1])
    eng.reset_timers()
    x0, r0 = eng.info('greedy_exact_windows'), eng.info('greedy_second_rounds')
    t0 = time.time()
    for _ in range(3): eng.greedy(U)
    dt = (time.time() - t0) / 3
    tm = eng.timers()['greedy_steps']
    us = tm[0] / tm[1] / steps * 1e3
    print('fast %d: %.2f ms per utterance, %.1f us/step (device), same path %s; exact windows/step %.1f, second rounds/step %.2f, f16 launches %d, bf16 products %d, fallbacks %d'
          % (fast, dt * 1e3, us, same, (eng.info('greedy_exact_windows') - x0) / 3.0 / steps, (eng.info('greedy_second_rounds') - r0) / 3.0 / steps,
             eng.info('greedy_f16_launches'), eng.info('greedy_hoist16_launches'), eng.info('greedy_fallbacks')), 'last launch: speculation used %d, several holders %d of %d steps' % (eng.info('greedy_last_speculated'), eng.info('greedy_last_several_holders'), steps), flush=True)
eng.close()
