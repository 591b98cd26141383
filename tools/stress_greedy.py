"""Stress of the streamed greedy scan's hand-off (greedy32_kernel: publish -> gather, deciding workgroup, holders' slots): many
launches of every instance, every result compared with the first one; no watchdog stall, no fallback allowed.
    python tools/stress_greedy.py [repetitions = 30]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
Dt, Dj, me = 61, 151, 6
total_steps = 0
t00 = time.time()
for N, T in ((65536, 600), (300000, 300), (1500000, 240)):
    F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
    # a stretch of speech that occurs three times (exact ties across workgroups) and near copies of it (several holders)
    for dst in (N // 3, 2 * N // 3 + 77):
        F_unw[dst:dst + 300] = F_unw[1000:1300]; JC_unw[dst:dst + 301] = JC_unw[1000:1301]
    wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
    eng = snickery_amd.HipSearchEngine(0)
    eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj); eng.set_greedy_layout(me, False, 0)
    eng.set_option('greedy_resident', 0)                      # the streamed kernel at every size
    rng = np.random.RandomState(5)
    utts = [synthetic_targets(F_unw, T, seed=1 + u) * wt for u in range(5)]
    utts.append(F_unw[1100:1100 + T // 2].astype(np.float64) * wt)                                     # exact three-way ties
    utts.append((F_unw[1100:1100 + T // 2].astype(np.float64) + 1e-7 * rng.randn(T // 2, Dt)) * wt)     # near ties
    for f16 in (1, 2):
        eng.set_option('greedy_f16', f16)
        ref1 = [eng.greedy(U) for U in utts]
        ref3 = eng.greedy_batch(utts[:3]); ref6 = eng.greedy_batch(utts[:6]); ref7 = eng.greedy_batch(utts)
        s0, f0 = eng.info('greedy_stalls'), eng.info('greedy_fallbacks')
        for r in range(reps):
            for U, p in zip(utts, ref1):
                assert eng.greedy(U) == p
                total_steps += len(p)
            assert eng.greedy_batch(utts[:3]) == ref3 and eng.greedy_batch(utts[:6]) == ref6 and eng.greedy_batch(utts) == ref7
            total_steps += sum(len(p) for p in ref3) + sum(len(p) for p in ref6) + sum(len(p) for p in ref7)
        assert eng.info('greedy_stalls') == s0 and eng.info('greedy_fallbacks') == f0, (eng.info('greedy_stalls'), eng.info('greedy_fallbacks'))
        print('N=%d greedy_f16 %d: %d repetitions of 7 single searches + batches of 3 / 6 / 7, all equal; stalls 0, fallbacks 0; %d steps so far, %.0f s'
              % (N, f16, reps, total_steps, time.time() - t00), flush=True)
    eng.close()
print('done: %d steps' % total_steps)
