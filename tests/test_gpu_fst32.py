"""Option viterbi_weights 1: the recursion in the reference's own arithmetic -- OpenFST's float32 tropical weights
(script/fst_functions_wrapped.py:47,201: lattice weights are parsed into float32; :368 compose, :389 shortestpath) -- against
the oracle's restatement of that chain (oracle/snk_oracle.py _viterbi_fst32; on the golden voice its path is pinned to the
reference's own lattice text, tests/test_oracle_golden.py).  Path AND the float32 total must be equal, bit for bit; float64
stays the default and is unchanged by a round trip through the option."""
import numpy as np
import pytest

import snk_oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def engine():
    import snickery_amd
    e = snickery_amd.HipSearchEngine(0)
    yield e
    e.set_option('viterbi_weights', 0)
    e.close()


def _fst32(engine, cand, dist):
    """Both homes of the float32 chain: the dense kernels (viterbi_mode 0) and, since round 5, the sparse path (bounds, predecessor
    sets, exact costs for the sets, a proof with the float32 roundings folded in) -- same path, same float32 total."""
    engine.set_option('viterbi_weights', 1)
    try:
        engine.set_option('viterbi_mode', 0)
        dense = engine.viterbi(cand, dist)
        engine.set_option('viterbi_mode', 2)
        before = engine.timers().get('viterbi_sparse', (0, 0))[1]
        sparse = engine.viterbi(cand, dist)
        if cand.shape[0] >= 2 and cand.shape[1] <= 208:
            assert engine.timers().get('viterbi_sparse', (0, 0))[1] > before           # the sparse recursion did run
        assert sparse == dense
        return sparse
    finally:
        engine.set_option('viterbi_weights', 0)
        engine.set_option('viterbi_mode', 2)


def test_fst32_on_the_golden_voice(engine, golden, mini_voice):
    engine.upload_db(mini_voice['F_unw'], mini_voice['JC_unw'])
    engine.set_weights(mini_voice['wt'], mini_voice['wj'])
    cand, dist = golden['join_candidates'], golden['knn_distances']
    opath, ocost = o.viterbi(cand, dist, mini_voice['E'], mini_voice['S'], mode='fst32')
    path, cost = _fst32(engine, cand, dist)
    assert path == opath and cost == ocost and np.float32(cost) == cost
    p64, c64 = engine.viterbi(cand, dist)                         # the default is back, and is the float64 recursion
    o64, oc64 = o.viterbi(cand, dist, mini_voice['E'], mini_voice['S'])
    assert p64 == o64 and c64 == oc64
    assert _fst32(engine, cand[:1], dist[:1]) == ([], float('inf'))           # T < 2
    dead = cand.copy()
    dead[5, :] = -1
    assert _fst32(engine, dead, dist) == ([], float('inf'))


@pytest.mark.parametrize('N,T,K,Dj', [(4000, 60, 50, 151), (2500, 30, 100, 302), (2000, 25, 13, 40), (3000, 20, 200, 151),
                                      (3000, 40, 128, 151), (3000, 300, 64, 40)])
def test_fst32_synthetic(engine, N, T, K, Dj):
    F_unw, JC_unw = o.synthetic_db(N, 61, Dj, K)
    rng = np.random.RandomState(K + 100)
    wt = 0.2 + rng.rand(61)
    wj = 0.05 + 0.2 * rng.rand(Dj)
    F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    U = o.synthetic_targets(F_unw, T, seed=5) * wt
    cand, dist = o.knn_bruteforce(F, U, K)
    cand[2, 1] = -1
    cand[3, 0] = 0
    cand[4, 2] = N - 1
    cand[6, 3] = cand[6, 0]; dist[6, 3] = dist[6, 0]              # a duplicate: exact tie between two slots
    opath, ocost = o.viterbi(cand, dist, E, S, mode='fst32')
    path, cost = _fst32(engine, cand, dist)
    assert path == opath and cost == ocost
    # the batch entry points take the same option
    engine.set_option('viterbi_weights', 1)
    try:
        paths, costs = engine.viterbi_batch([cand, cand[:7], cand[:1]], [dist, dist[:7], dist[:1]])
        p7, c7 = o.viterbi(cand[:7], dist[:7], E, S, mode='fst32')
        assert list(paths[0]) == opath and costs[0] == ocost and list(paths[1]) == p7 and costs[1] == c7 and len(paths[2]) == 0
        pk, ck, candk, distk = engine.knn_viterbi(U, K, return_candidates=True)
        opk, ock = o.viterbi(candk, distk, E, S, mode='fst32')
        assert pk == opk and ck == ock
    finally:
        engine.set_option('viterbi_weights', 0)


def test_fst32_on_a_baseline_sized_utterance(engine):
    """One B*-shaped utterance (T 600, K 100, 302 join columns) on a 300 k-unit walk: near ties everywhere, where the
    float32 chain and the float64 recursion part ways (profiles/r02_fst32_statistic.json) -- the device follows the
    float32 chain exactly."""
    N, Dj, T, K = 300000, 302, 600, 100
    F_unw, JC_unw = o.synthetic_db(N, 61, Dj, seed=14)
    wt, wj = np.full(61, 0.4), np.full(Dj, 0.05)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    U = o.synthetic_targets(F_unw, T, seed=21) * wt
    cand, dist = engine.knn(U, K)
    J = engine.join_costs(cand)            # (T-1, K, K) exact float64 (bit-equal to the oracle's: test_gpu_parity.py)
    ok = o.valid_mask(cand, N)
    opath, ocost = o._viterbi_fst32(cand, dist.astype(np.float32), J.astype(np.float32), ok)
    path, cost = _fst32(engine, cand, dist)
    assert path == opath and cost == ocost
    p64, _ = engine.viterbi(cand, dist)
    print('fst32 vs float64 on this utterance: %d of %d frames differ' % (int(np.sum(np.asarray(p64) != np.asarray(path))), T))


def test_fst32_batches_on_the_sparse_path_at_the_headline_shape(engine):
    """viterbi_weights 1 through snk_knn_viterbi_batch with viterbi_mode 2: a batch of B*-shaped utterances
    (T 600, K 100, 302 join columns) on the sparse path equals the oracle's float32 chain utterance by utterance, the margins of
    the proof do not make pass 4 compute everything exactly (a small fraction of the K x K costs is)."""
    import time
    N, Dj, T, K, U = 400000, 302, 600, 100, 8
    F_unw, JC_unw = o.synthetic_db(N, 61, Dj, seed=15)
    wt, wj = np.full(61, 0.4), np.full(Dj, 0.05)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    utts = [o.synthetic_targets(F_unw, T, seed=30 + u) * wt for u in range(U)]
    engine.set_option('viterbi_latch', 0)
    try:
        def timed():
            engine.knn_viterbi_batch(utts, K)
            t0 = time.perf_counter()
            for _ in range(3):
                res = engine.knn_viterbi_batch(utts, K)
            return (time.perf_counter() - t0) / 3, res
        t64, (p64, c64) = timed()
        engine.set_option('viterbi_weights', 1)
        before = engine.info('dense_exact_costs')
        n_sparse = engine.timers().get('viterbi_sparse', (0, 0))[1]
        t32, (p32, c32) = timed()
        assert engine.timers().get('viterbi_sparse', (0, 0))[1] > n_sparse
        refined = (engine.info('dense_exact_costs') - before) / 4.0
        assert refined <= 0.02 * U * T * K * K, refined
        for u in (0, 3, U - 1):
            cand, dist = engine.knn(utts[u], K)
            J = engine.join_costs(cand)
            opath, ocost = o._viterbi_fst32(cand, dist.astype(np.float32), J.astype(np.float32), o.valid_mask(cand, N))
            assert list(p32[u]) == opath and c32[u] == ocost, u
        # (no speed claim: float32 totals of magnitude ~ 200 tie within an ulp of 1.5e-5 -- as wide as the join costs of temporal
        # neighbours differ --, so the proof fails at most steps and pass 4 scans all predecessors of most columns: measured 5 x the
        # float64 step here.  With the latch on, viterbi_mode 2 tries the dense float32 kernels and keeps the faster path.)
        print('float64 step %.2f ms, float32-weights step on the sparse path %.2f ms; exact costs in refinements per batch %d' % (t64 * 1e3, t32 * 1e3, refined))
    finally:
        engine.set_option('viterbi_weights', 0)
        engine.set_option('viterbi_latch', 1)
