"""The Viterbi side of one group of a B* batch (16 utterances x 600 rows, K = 100, 302 join columns) on its own, from
candidates computed once: stage times per form of the bounds pass (join_lb_variant) and what pass 4 refines.
    python tools/joinlb_time.py [variant ...] [x0] [x1] [--utts U] [--reps R] [--speechlike]      (x0 / x1: form of the exact sparse costs;
    --speechlike: bench.py's AR(1) voice and held-out utterances instead of SURVEY 8d's walk)"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets, speechlike_voice

args = sys.argv[1:]
U = int(args[args.index('--utts') + 1]) if '--utts' in args else 16
R = int(args[args.index('--reps') + 1]) if '--reps' in args else 5
variants = [int(a) for a in args if a in ('0', '1')] or [1, 0]
N, Dt, Dj, T, K = 1048576, 61, 302, 600, 100
held_out = None
if '--speechlike' in args: F_unw, JC_unw, held_out = speechlike_voice(N, Dt, Dj, seed=0)
else: F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
cands, dists = [], []
for s in range(U):
    c, d = eng.knn((held_out(T, s) if held_out else synthetic_targets(F_unw, T, seed=1 + s)) * wt, K)
    cands.append(c); dists.append(d)
eng.set_option('viterbi_mode', 1)
if '--one-set' in args: eng.set_option('join_lb_one_set', 1)
ref = None
forms = [int(a[1:]) for a in args if a in ('x0', 'x1')] or [1]
for v, form in [(v, f) for v in variants for f in forms]:
    eng.set_option('join_lb_variant', v)
    eng.set_option('join_exact_form', form)
    out = eng.viterbi_batch(cands, dists)
    if ref is None: ref = out
    same = all(np.array_equal(a, b) for a, b in zip(out[0], ref[0])) and np.array_equal(out[1], ref[1])
    eng.reset_timers()
    st0 = [eng.info(x) for x in ('dense_cells', 'dense_steps', 'dense_exact_costs', 'set_overflows')]
    t0 = time.time()
    for _ in range(R): eng.viterbi_batch(cands, dists)
    dt = (time.time() - t0) / R
    st1 = [eng.info(x) for x in ('dense_cells', 'dense_steps', 'dense_exact_costs', 'set_overflows')]
    tm_timed = eng.timers()
    eng.set_option('roofline_counters', 1); eng.reset_timers()          # (one more call, counted: the counters cost time, the timed calls ran without)
    eng.viterbi_batch(cands, dists)
    extra = ' pass-3 exact costs / set members per call [%d, %d]' % (eng.info('sparse_exact_costs'), eng.info('sparse_set_members'))
    eng.set_option('roofline_counters', 0)
    print('join_lb_variant %d join_exact_form %d: %.2f ms per call (with the upload of the candidates) same=%s stages %s refined cells / steps / exact costs / overflows per call %s'
          % (v, form, dt * 1e3, same, {k: round(x[0] / R, 3) for k, x in tm_timed.items() if x[1]},
             [round((b - a) / R) for a, b in zip(st0, st1)]) + extra, flush=True)
eng.close()
