"""Randomised parity sweep of the greedy search against the C oracle (not collected by pytest; run on a GPU box):
random database sizes (a handful of workgroups up to the full grid), layouts (multiepoch, last_frame_as_target, join split),
widths, utterance lengths and start states; every case through the streamed float32 scan alone, with float16 join tiles
forced, as a batch, and through the default mode -- paths and distances must be the oracle's bit for bit.
    python tests/fuzz_greedy.py [cases = 60] [seed = 1]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import snickery_amd
import snk_oracle as o
import snk_oracle_c as oc
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = snickery_amd.HipSearchEngine(0)
t0 = time.time()
done = 0
for c in range(cases):
    N = int(rng.choice([700, 3000, 9000, 20000, 33000, 70000]))
    me = int(rng.randint(1, 8)); lfat = bool(rng.randint(2)) and me > 1; mode = int(rng.randint(2))
    Dt = int(rng.choice([7, 20, 61, 64, 65, 130])); Dj = int(rng.choice([33, 70, 96, 100, 129, 151, 200, 302]))
    if mode == 1 and Dj % 2: Dj += 1           # (join_split_mode 1 halves the join columns)
    F_unw, JC_unw = o.synthetic_db(N + me, Dt, Dj, seed=1000 + c)
    if rng.randint(3) == 0:                        # a stretch that occurs twice: exact ties across tiles and workgroups
        L = min(200, N // 4); dst = int(rng.randint(N // 2, N - L - me))
        F_unw[dst:dst + L] = F_unw[100:100 + L]; JC_unw[dst:dst + L + 1] = JC_unw[100:100 + L + 1]
    wt, wj = 0.2 + rng.rand(Dt), 0.05 + 0.2 * rng.rand(Dj)
    eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj); eng.set_greedy_layout(me, lfat, mode)
    lens = [int(rng.randint(1, 40)) * me + int(rng.randint(me)) for _ in range(int(rng.randint(1, 7)))]
    utts = []
    for i, T in enumerate(lens):
        if rng.randint(4) == 0 and T + 120 < N:     # noise-free targets from inside the database (distance 0, ties with a copy)
            utts.append(F_unw[110:110 + T].astype(np.float64) * wt)
        else:
            utts.append(o.synthetic_targets(F_unw, T, seed=7 + i + c) * wt)
    starts = [int(rng.choice([-1, 0, 17, N // 2])) for _ in utts]
    ref = [oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, lfat, mode, st) for U, st in zip(utts, starts)]
    for gm, f16, res in ((1, 1, 0), (1, 2, 0), (2, 1, 1), (2, 2, 0)):
        eng.set_option('greedy_mode', gm); eng.set_option('greedy_f16', f16); eng.set_option('greedy_resident', res)
        for U, st, (op, od) in zip(utts[:2], starts[:2], ref[:2]):
            p, d = eng.greedy(U, start_state=st, return_distances=True)
            assert p == op and np.array_equal(d, od), ('single', c, N, me, lfat, mode, Dt, Dj, gm, f16)
        ps, ds = eng.greedy_batch(utts, start_states=starts, return_distances=True)
        for p, d, (op, od) in zip(ps, ds, ref):
            assert p == op and np.array_equal(d, od), ('batch', c, N, me, lfat, mode, Dt, Dj, gm, f16)
    done += 1
    if done % 10 == 0:
        print('%d cases equal to the oracle (%.0f s); stalls %d' % (done, time.time() - t0, eng.info('greedy_stalls')), flush=True)
eng.set_option('greedy_mode', 2); eng.set_option('greedy_f16', 1); eng.set_option('greedy_resident', 1)
print('done: %d cases, all paths and distances equal to the C oracle; watchdog stalls %d, fallbacks to the exact scan %d'
      % (done, eng.info('greedy_stalls'), eng.info('greedy_fallbacks')))
eng.close()
