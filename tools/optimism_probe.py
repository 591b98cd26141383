"""Optimistic K-NN thresholds (api_knn.hip) on a voice: list lengths, flagged calls and stage times per rank j of the sample minimum
the thresholds come from (0 = the engine's choice; K = guaranteed).   python tools/optimism_probe.py [--speechlike] [rows]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets, speechlike_voice
args = [a for a in sys.argv[1:] if not a.startswith('--')]
rows = int(args[0]) if args else 9600
N, Dt, Dj, K = 1048576, 61, 302, 100
held_out = None
if '--speechlike' in sys.argv: F_unw, JC_unw, held_out = speechlike_voice(N, Dt, Dj, seed=0)
else: F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
U = np.vstack([(held_out(600, s) if held_out else synthetic_targets(F_unw, 600, seed=1 + s)) * wt for s in range((rows + 599) // 600)])[:rows]
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
eng.set_option('tau_optimism', 0)
ref = eng.knn(U, K)
for _ in range(2): eng.knn(U, K)          # (the filter latch settles)
for j in (K, 0, 24, 33, 48, 64):
    eng.set_option('tau_optimism', 0 if j == K else 1)
    eng.set_option('tau_optimism_rank', 0 if j == K else j)
    f0 = eng.info('tau_optimism_failures')
    c, d = eng.knn(U, K)
    eng.reset_timers()
    t0 = time.time()
    for _ in range(3): c, d = eng.knn(U, K)
    dt = (time.time() - t0) / 3
    print('rank %3d (ran with %3d): %.2f ms/call same=%s list mean %.0f max %.0f flagged calls %d off %d status %d coarse %d  %s' % (
        j, eng.info('tau_optimism_rank'), dt * 1e3, np.array_equal(c, ref[0]) and np.array_equal(d, ref[1]), eng.info('last_list_mean'), eng.info('last_list_max'),
        eng.info('tau_optimism_failures') - f0, eng.info('tau_optimism_off'), eng.info('last_f16_status'), eng.info('filter_coarse'),
        {k: round(v[0] / 3, 3) for k, v in eng.timers().items() if v[1]}), flush=True)
if '--hunt' in sys.argv:
    # which rows are flagged: utterance by utterance (32 of them), then row by row
    eng.set_option('tau_optimism', 1); eng.set_option('tau_optimism_rank', 0)
    for sidx in range(32):
        Us = (held_out(600, sidx) if held_out else synthetic_targets(F_unw, 600, seed=1 + sidx)) * wt
        f0 = eng.info('tau_optimism_failures')
        eng.set_option('tau_optimism', 1)                  # (re-arms the voice)
        eng.knn(Us, K)
        if eng.info('tau_optimism_failures') > f0:
            print('utterance %d is flagged' % sidx, flush=True)
            for r in range(600):
                f1 = eng.info('tau_optimism_failures')
                eng.set_option('tau_optimism', 1)
                c1, d1 = eng.knn(Us[r:r + 1], K)
                if eng.info('tau_optimism_failures') > f1:
                    eng.set_option('tau_optimism', 0)
                    c0, d0 = eng.knn(Us[r:r + 1], K)
                    n_guar = eng.info('last_list_mean')
                    eng.set_option('tau_optimism', 1); eng.set_option('tau_optimism_rank', 33)
                    eng.set_option('prefilter', 1)
                    print('  row %d alone is flagged too: guaranteed list %d entries, d_1 %.4f d_K %.4f, same=%s' % (r, n_guar, d0[0, 0], d0[0, -1], np.array_equal(c0, c1)), flush=True)
                    break
            else:
                print('  no single row of it is flagged alone', flush=True)
            break
eng.close()
