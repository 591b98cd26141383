"""snk_sharded_greedy, what ONE rank of G does per step: this GPU scans the share a rank of G would scan of a database of
`units` x G units (its own database has `units` units, one rank: the all-gather is a 16-byte copy), beside snk_greedy's scans
of the same database.     python tools/shard_greedy_time.py [units]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500000
Dt, Dj, T, me = 61, 151, 600, 6
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt, wj = np.full(Dt, 0.4), np.full(Dj, 0.05)
U = synthetic_targets(F_unw, T, seed=1) * wt
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj); eng.set_greedy_layout(me, False, 0)
ref = eng.greedy(U, return_distances=True)
eng.reset_timers(); eng.greedy(U)
fast = eng.timers()['greedy_steps'][0] / (T // me) * 1e3
eng.set_option('greedy_mode', 0)
eng.greedy(U); eng.reset_timers(); p0 = eng.greedy(U, return_distances=True)
exact = eng.timers()['greedy_steps'][0] / (T // me) * 1e3
eng.comm_init(1, 0, eng.comm_unique_id())
eng.sharded_greedy(U); eng.reset_timers()
t0 = time.time(); p1 = eng.sharded_greedy(U, return_distances=True); wall = (time.time() - t0) / (T // me) * 1e6
sh = eng.timers()['greedy_steps'][0] / (T // me) * 1e3
print('%d units, me %d, %d steps: snk_greedy %.1f us/step (float16 scan with the hoisted target term), %.1f us/step (exact float64 scan, one launch per step); '
      'snk_sharded_greedy on one rank %.1f us/step by the device (%.1f wall): scan + 16-byte copy + pick; same path %s'
      % (N, me, T // me, fast, exact, sh, wall, p1[0] == ref[0] and np.array_equal(p1[1], ref[1]) and p0[0] == ref[0]))
