"""The Viterbi side of one group (16 utterances x 600 rows, K = 100) on bench.py's speech-like voice (AR(1) target and join
features, held-out utterances), per setting of the sparse path's speed knobs -- pass 2's margin (join_beta), its chunking
(viterbi_lb_chunk / viterbi_lb_warm) -- beside the dense exact kernels: stage times, cells refined in pass 4, results compared.
    python tools/speechlike_sweep.py [--utts U]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import speechlike_voice

args = sys.argv[1:]
U = int(args[args.index('--utts') + 1]) if '--utts' in args else 16
N, Dt, Dj, T, K = 1048576, 61, 302, 600, 100
F_unw, JC_unw, held_out = speechlike_voice(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
cands, dists = [], []
for s in range(U):
    c, d = eng.knn(held_out(T, s) * wt, K)
    cands.append(c); dists.append(d)
print('target distances of a row: first %.3f, K-th %.3f; candidates that are successors of one another: %.3f' % (
    dists[0][:, 0].mean(), dists[0][:, -1].mean(), np.mean(np.diff(np.sort(cands[0], axis=1), axis=1) == 1)), flush=True)
eng.set_option('viterbi_mode', 0)
ref = eng.viterbi_batch(cands, dists)
R = 3
def run(label):
    out = eng.viterbi_batch(cands, dists)
    same = all(np.array_equal(a, b) for a, b in zip(out[0], ref[0])) and np.array_equal(out[1], ref[1])
    eng.reset_timers()
    c0 = [eng.info(x) for x in ('dense_cells', 'dense_steps', 'dense_exact_costs', 'set_overflows')]
    t0 = time.time()
    for _ in range(R): eng.viterbi_batch(cands, dists)
    dt = (time.time() - t0) / R
    c1 = [eng.info(x) for x in ('dense_cells', 'dense_steps', 'dense_exact_costs', 'set_overflows')]
    print('%-44s %.2f ms per call same=%s stages %s refined cells / steps / exact costs / overflows per call %s' % (
        label, dt * 1e3, same, {k: round(x[0] / R, 3) for k, x in eng.timers().items() if x[1]}, [round((b - a) / R) for a, b in zip(c0, c1)]), flush=True)
run('dense exact kernels (viterbi_mode 0)')
eng.set_option('viterbi_mode', 1)
for beta in (5e-4, 2e-3, 8e-3, 3e-2):
    for chunk, warm in ((48, 16), (48, 48), (0, 0)):
        eng.set_option('join_beta', beta); eng.set_option('viterbi_lb_chunk', chunk)
        if chunk: eng.set_option('viterbi_lb_warm', warm)
        run('sparse: join_beta %g chunk %d warm %d' % (beta, chunk, warm))
eng.close()
