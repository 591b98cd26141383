"""One bench leg (B* on the speech-like voice, or --compact: SURVEY 8d's walk) through the batch pipeline, host -> host, three steps in
flight, with the stage timers off / roofline stage only / all (option timers 0 / 2 / 1), in alternating order: does the rate depend
on the timestamp events a timed stage puts on its stream?     python tools/leg_timers_ab.py [--compact] [--opt name=value ...]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
snickery_amd.configure_runtime()
if '--torch' in sys.argv:
    import torch                      # (bench.py imports torch first: its bundled HIP runtime is then the process's)
    torch.cuda.synchronize()
from bench import synthetic_db, synthetic_targets, speechlike_voice
N, Dt, Dj, T, K, U = 1048576, 61, 302, 600, 100, 32
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
if '--compact' in sys.argv:
    F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
    utts = [synthetic_targets(F_unw, T, seed=1 + u) * wt for u in range(U)]
else:
    F_unw, JC_unw, held_out = speechlike_voice(N, Dt, Dj, seed=0)
    utts = [held_out(T, u) * wt for u in range(U)]
eng = snickery_amd.HipSearchEngine(0)
args = sys.argv[1:]
for i, a in enumerate(args):
    if a == '--opt':
        n, v = args[i + 1].split('='); eng.set_option(n, float(v))
if '--after-compact' in sys.argv:
    # what bench.py does before this leg: the same engine has searched SURVEY 8d's walk and its permuted copy
    F0, JC0 = synthetic_db(N, Dt, Dj, seed=0)
    u0 = snickery_amd.QueryBatch([synthetic_targets(F0, T, seed=1 + u) * wt for u in range(U)]).pin()
    eng.upload_db(F0, JC0); eng.set_weights(wt, wj)
    for _ in range(8): eng.knn_viterbi_batch_collect(eng.knn_viterbi_batch_submit(u0, K))
    rng = np.random.RandomState(17); perm = rng.permutation(N)
    eng.upload_db(F0[perm], JC0[np.concatenate([perm, [N]])]); eng.set_weights(wt, wj)
    for _ in range(8): eng.knn_viterbi_batch_collect(eng.knn_viterbi_batch_submit(u0, K))
    print('after the compact and the permuted voice: reordered %d filter_coarse %d onepass %d knn_level %d' % (
        eng.info('reordered'), eng.info('filter_coarse'), eng.info('filter_onepass'), eng.info('knn_level')), flush=True)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
batch = snickery_amd.QueryBatch(utts).pin()
for _ in range(12):
    eng.knn_viterbi_batch_collect(eng.knn_viterbi_batch_submit(batch, K))


def run(steps=20, depth=3):
    pending = []
    t0 = time.perf_counter()
    for _ in range(steps):
        pending.append(eng.knn_viterbi_batch_submit(batch, K))
        if len(pending) >= depth:
            eng.knn_viterbi_batch_collect(pending.pop(0))
    while pending:
        eng.knn_viterbi_batch_collect(pending.pop(0))
    return T * U * steps / (time.perf_counter() - t0)
print('voice: reordered %d filter_coarse %d onepass %d knn_level %d warm %d rank %d off %d latch mode %d' % (
    eng.info('reordered'), eng.info('filter_coarse'), eng.info('filter_onepass'), eng.info('knn_level'), eng.info('viterbi_lb_warm_now'),
    eng.info('tau_optimism_rank'), eng.info('tau_optimism_off'), eng.info('viterbi_latch_mode')), flush=True)
if '--mask-scan' in sys.argv:
    # which stage's timestamp events matter: every stage alone (timers 3 + timers_mask), between reference runs
    names = ['h2d', 'prep', 'minima', 'threshold', 'filter', 'bucket', 'finalize', 'join', 'viterbi_dp', 'd2h', 'greedy_target', 'greedy_steps', 'weights',
             'merge', 'join_lb', 'dp_lb', 'join_sparse', 'dp_sparse', 'ballmin']
    eng.set_option('timers', 1); run(); eng.set_option('timers', 0); print('timers 0: %.0f' % run(), flush=True)
    for i in (0, 1, 2, 3, 4, 5, 6, 9, 14, 15, 16, 17):
        eng.set_option('timers', 3); eng.set_option('timers_mask', float(1 << i)); eng.reset_timers()
        print('only %-12s: %.0f frames/s' % (names[i], run()), flush=True)
    for label, mask in (('main stream', 0x7f), ('side streams', (1 << 14) | (1 << 15) | (1 << 16) | (1 << 17))):
        eng.set_option('timers_mask', float(mask)); eng.reset_timers()
        print('only %-12s: %.0f frames/s' % (label, run()), flush=True)
    eng.set_option('timers', 1); print('timers 1: %.0f' % run(), flush=True)
    eng.set_option('timers', 0); print('timers 0: %.0f' % run(), flush=True)
    eng.close(); sys.exit(0)
for mode in (2, 1, 0, 1, 2, 0, 2, 1):
    eng.set_option('timers', mode)
    eng.reset_timers()
    print('timers %d: %.0f frames/s' % (mode, run()), flush=True)
eng.close()
