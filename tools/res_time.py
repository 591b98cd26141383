"""Resident greedy scan against the streamed one-launch scan at B1 (N = 65 536, me = 6, T = 600): device time per step
from the engine's HIP events (includes the utterance's share of the hoisting product) and wall time per utterance."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
Dt, Dj, T, me = 61, 151, int(sys.argv[2]) if len(sys.argv) > 2 else 600, 6
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj); eng.set_greedy_layout(me, False, 0)
U = synthetic_targets(F_unw, T, seed=1) * wt
ref = None
for res in (1, 0, 1):
    eng.set_option('greedy_resident', res)
    p = eng.greedy(U)
    ref = ref or p
    eng.reset_timers()
    t0 = time.time()
    for _ in range(5):
        p = eng.greedy(U)
    dt = (time.time() - t0) / 5
    tm = eng.timers()['greedy_steps']
    steps = T // me
    print('resident %d: same path %s; %.2f ms/utt wall; device %.2f us/step (launch incl. product); fallbacks %d exact windows %d rounds %d' % (
        res, p == ref, dt * 1e3, tm[0] / tm[1] / steps * 1e3, eng.info('greedy_fallbacks'), eng.info('greedy_exact_windows'), eng.info('greedy_second_rounds')))
eng.close()

fn = os.environ.get('SNK_GRES_TRACE')
if fn and os.path.exists(fn):
    raw = np.fromfile(fn, dtype=np.uint64).astype(np.float64)
    t, c = raw[:2048].reshape(256, 8), raw[2048:].reshape(256, 8)
    n = min(256, T // me)
    t, c = t[2:n], c[2:n]            # skip the first steps (cold)
    print('shader clock during the launch: %.0f MHz' % (float(np.sum(c[1:, 0] - c[:-1, 0])) / float(np.sum(t[1:, 0] - t[:-1, 0])) * 100.0))
    names = ['start', 'table', 'scan', 'top3', 'published', 'gathered', 'decided', 'broadcast']
    print('timeline of workgroup 0, us from the start of the step (mean over %d steps; 100 MHz clock):' % len(t))
    for k in range(1, 8):
        print('  %-10s %6.2f' % (names[k], float(np.mean(t[:, k] - t[:, 0])) / 100.0))
    print('  step       %6.2f' % (float(np.mean(t[1:, 0] - t[:-1, 0])) / 100.0))
