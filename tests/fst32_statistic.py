"""How often would OpenFST's float32 tropical weights (the reference's pywrapfst path,
script/fst_functions_wrapped.py:47,201,368,389) pick another unit sequence than float64 accumulation?

The HIP path and its parity oracle accumulate in float64; the reference compiles its lattices with float32 arc
weights.  pywrapfst is not in this image, so the float32 chain is the oracle's restatement of it
(oracle/snk_oracle.py _viterbi_fst32).  This script measures, on the B* shape (T 600, K 100, |DB| 1 M) and on
the golden voice's real speech frames, in how many utterances and frames the two disagree and what the
disagreement costs under the exact objective.  Test infrastructure (imports the oracle); run on a GPU box:

    python tests/fst32_statistic.py [n_utts] > profiles/r02_fst32_statistic.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import snk_oracle as o          # noqa: E402
import snickery_amd             # noqa: E402
from bench import synthetic_db, synthetic_targets   # noqa: E402


def compare(eng, utts, K, n_units):
    out = {'utterances': 0, 'frames': 0, 'utterances_with_another_path': 0, 'frames_with_another_unit': 0,
           'max_relative_cost_excess_of_the_f32_path': 0.0, 'ties_within_f32_resolution': 0}
    for U in utts:
        cand, dist = eng.knn(U, K)
        J = eng.join_costs(cand)                                   # (T-1, K, K) float64, +inf where unusable
        ok = o.valid_mask(cand, n_units)
        p64, c64 = eng.viterbi(cand, dist)
        p32, c32 = o._viterbi_fst32(cand, dist.astype(np.float32), J.astype(np.float32), ok)
        p64, p32 = np.asarray(p64), np.asarray(p32)
        out['utterances'] += 1
        out['frames'] += len(p64)
        diff = int(np.sum(p64 != p32))
        if diff:
            out['utterances_with_another_path'] += 1
            out['frames_with_another_unit'] += diff
            # exact (float64) objective of the float32 choice
            slot = [int(np.nonzero(cand[t] == u)[0][0]) for t, u in enumerate(p32)]
            exact = sum(dist[t, s] for t, s in enumerate(slot)) + sum(J[t, slot[t], slot[t + 1]] for t in range(len(slot) - 1))
            excess = (exact - c64) / c64
            out['max_relative_cost_excess_of_the_f32_path'] = max(out['max_relative_cost_excess_of_the_f32_path'], float(excess))
            if excess < 6e-8 * len(slot):
                out['ties_within_f32_resolution'] += 1
    return out


def main():
    n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    N, Dt, Dj, T, K = 1048576, 61, 302, 600, 100
    F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
    wt, wj = np.full(Dt, 0.4), np.full(Dj, 0.05)
    eng = snickery_amd.HipSearchEngine(0)
    eng.upload_db(F_unw, JC_unw)
    eng.set_weights(wt, wj)
    res = {'what': 'float64 accumulation (HIP path = oracle) against the oracle\'s float32 OpenFST chain, same candidates and join costs',
           'b_star': dict(shape='|DB| 1048576, T 600, K 100, synthetic walk (SURVEY 8d)',
                          **compare(eng, [synthetic_targets(F_unw, T, seed=1 + u) * wt for u in range(n_utts)], K, N))}
    gfile = os.path.join(ROOT, 'tests', 'golden', 'reference_mini.npz')
    if os.path.isfile(gfile):
        g = np.load(gfile, allow_pickle=True)
        dims = {'mag': 60, 'real': 45, 'imag': 45, 'lf0': 1}
        tw, jw = o.apply_jcw(g['target_stream_weights'], g['join_stream_weights'], float(g['join_cost_weight']))
        w_t = np.asarray(o.stream_weight_vector(list(tw), ['mag', 'lf0'], dims), dtype=np.float64)
        w_j = np.asarray(o.stream_weight_vector(list(jw), ['mag', 'real', 'imag', 'lf0'], dims), dtype=np.float64)
        F, JC = g['F_unw'], g['JC_unw']
        eng.upload_db(F, JC)
        eng.set_weights(w_t, w_j)
        rng = np.random.RandomState(0)
        utts = []
        for u in range(n_utts):
            s0 = rng.randint(0, F.shape[0] - 300)
            utts.append((F[s0:s0 + 300].astype(np.float64) + 0.2 * rng.randn(300, F.shape[1])) * w_t)
        res['golden_voice'] = dict(shape='|DB| %d real slt frames, T 300, K 50' % F.shape[0], **compare(eng, utts, 50, F.shape[0]))
    eng.close()
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
