"""Developer aid: the resident greedy scan on a small voice, with the status words of the launch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import snk_oracle as o
import snk_oracle_c as oc
import snickery_amd

N, Dt, Dj, me = int(sys.argv[1]) if len(sys.argv) > 1 else 30006, 61, 151, 6
F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, 7)
rng = np.random.RandomState(1)
wt, wj = 0.2 + rng.rand(Dt), 0.05 + 0.2 * rng.rand(Dj)
e = snickery_amd.HipSearchEngine(0)
e.upload_db(F_unw, JC_unw); e.set_weights(wt, wj); e.set_greedy_layout(me, False, 0)
U = o.synthetic_targets(F_unw, 20 * me, seed=3) * wt
op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, False, 0, -1)
for res in (1, 0):
    e.set_option('greedy_resident', res)
    f0 = e.info('greedy_fallbacks')
    p, d = e.greedy(U, return_distances=True)
    print('resident', res, 'ok', p == op, 'fallbacks', e.info('greedy_fallbacks') - f0, 'undecided step', e.info('greedy_last_undecided_step'),
          'watchdog', e.info('greedy_last_watchdog'), 'resident launches', e.info('greedy_resident_launches'), 'exact windows', e.info('greedy_exact_windows'),
          'why', [e.info('greedy_last_why_' + k) for k in ('candidates', 'third', 'min', 'tau')], 'dist0', float(od[0]) ** 2)
print(op[:8]); print(p[:8])
