"""GPU parity at BASELINE sizes.  Mid size: full comparison against the C oracle.  Headline
size B* (|DB| = 1 048 576, T = 600, K = 100): spot rows against the C oracle plus size-independent
properties (sortedness, exact-distance recomputation, path-cost identity, optimality bounds)."""
import numpy as np
import pytest

import snk_oracle as o
import snk_oracle_c as oc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def engine():
    import snickery_amd
    e = snickery_amd.HipSearchEngine(0)
    yield e
    e.close()


def test_mid_size_full_parity(engine):
    N, Dt, Dj, T, K = 200000, 61, 302, 64, 100
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=4)
    rng = np.random.RandomState(7)
    wt = 0.2 + rng.rand(Dt)
    wj = 0.02 + 0.1 * rng.rand(Dj)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    F = o.weight(F_unw, wt)
    JCw = o.weight(JC_unw, wj)
    # half the rows follow the database (clustered neighbours), half are random (diffuse)
    U = np.vstack([o.synthetic_targets(F_unw, T // 2, seed=5), F_unw[rng.randint(0, N, T // 2)] + rng.randn(T // 2, Dt)]) * wt
    path, cost, cand, dist = engine.knn_viterbi(U, K, return_candidates=True)
    oc_cand, oc_dist = oc.knn(F, U, K)
    assert np.array_equal(cand, oc_cand)
    assert np.array_equal(dist, oc_dist)
    opath, ocost = oc.viterbi(oc_cand, oc_dist, JCw)
    assert path == opath and cost == ocost


def test_headline_size_properties(engine):
    N, Dt, Dj, T, K = 1048576, 61, 302, 600, 100
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=0)
    wt = np.full(Dt, 0.4)
    wj = np.full(Dj, 0.05)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    U = o.synthetic_targets(F_unw, T, seed=1) * wt
    path, cost, cand, dist = engine.knn_viterbi(U, K, return_candidates=True)
    # K-NN: ascending, unique, in range, distances are the exact canonical distances
    assert np.all(np.diff(dist, axis=1) >= 0)
    assert np.all((cand >= 0) & (cand < N))
    assert all(len(set(r)) == K for r in cand[::37])
    F = o.weight(F_unw, wt)
    for t in (0, 299, 599):
        assert np.array_equal(dist[t], np.sqrt(o.sqdist_rows(F[cand[t]], U[t])))
    # spot rows against the C oracle over the whole database
    rows = [3, 311, 598]
    oc_cand, oc_dist = oc.knn(F, U[rows], K)
    assert np.array_equal(cand[rows], oc_cand) and np.array_equal(dist[rows], oc_dist)
    # no unit outside the list is closer than the K-th (checked on GEMM-form distances, all rows of a block)
    fn = (F * F).sum(1)
    for t in rows:
        d2 = fn - 2.0 * F.dot(U[t]) + U[t].dot(U[t])
        assert (d2 < dist[t, -1] ** 2 * (1 - 1e-9)).sum() <= K
    # Viterbi: path uses one candidate per column, cost identity, optimality bounds
    JCw = o.weight(JC_unw, wj)
    E, S = JCw[1:], JCw[:-1]
    assert len(path) == T
    slots = [int(np.nonzero(cand[t] == path[t])[0][0]) for t in range(T)]
    tsum = np.array([dist[t, s] for t, s in enumerate(slots)])
    assert abs(o.path_cost(path, tsum, E, S) - cost) <= 1e-9 * cost
    ok = o.valid_mask(cand, N)
    first = [int(cand[t][ok[t]][0]) for t in range(T)]
    assert cost <= o.path_cost(first, [dist[t][ok[t]][0] for t in range(T)], E, S) + 1e-9
    # the device DP equals the C oracle's DP on the same trellis (bit-exact cost, same path)
    opath, ocost = oc.viterbi(cand, dist, JCw)
    assert path == opath and cost == ocost
    # batch entry point returns the same thing
    paths, costs = engine.knn_viterbi_batch([U, U[:100]], K)
    assert list(paths[0]) == path and costs[0] == cost
