"""Pass 1 of the sparse Viterbi path (joinfast_kernels.hip) on its own: the float32 values it hands to the recursion must
be LOWER bounds of the exact float64 join costs (make_on_the_fly_join_lattice_BLOCK_DIRECT, synth_halfphone.py:3206-3322;
get_natural_distance_vectorised :2942-2951) in every cell, for both forms of the pass -- bf16 pieces of the weighted float32
copy (join_lb_variant 1, default) and float32 operands weighted per gather (0).  How tight they are only decides how much
pass 4 refines; it is reported, and held to a loose floor so that a regression of the bound shows."""
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
    e.set_option('join_lb_variant', 1)
    e.close()


def _case(engine, N, Dj, T, K, seed, offset=0.0, scale=1.0):
    F_unw, JC_unw = o.synthetic_db(N, 61, Dj, seed)
    JC_unw = (JC_unw * scale + offset).astype(np.float32)
    rng = np.random.RandomState(seed + 7)
    wt = 0.2 + rng.rand(61)
    wj = 0.05 + 0.2 * rng.rand(Dj)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    # half the rows follow the database (time neighbours: tiny join costs), half are random frames
    U = np.vstack([o.synthetic_targets(F_unw, T - T // 2, seed=5),
                   F_unw[rng.randint(0, N, T // 2)] + rng.randn(T // 2, 61)]) * wt
    cand, _ = engine.knn(U, K)
    cand = cand.copy()
    cand[1, 0] = 0                       # unusable units, padding, a duplicate
    cand[2, K - 1] = N - 1
    if K > 2:
        cand[3, 1] = -1
        cand[4, 2] = cand[4, 0]
    return cand


@pytest.mark.parametrize('N,Dj,T,K,offset,scale', [
    (20000, 302, 40, 100, 0.0, 1.0), (20000, 151, 30, 50, 0.0, 1.0), (8000, 302, 12, 200, 0.0, 1.0),
    (6000, 40, 20, 16, 0.0, 1.0), (5000, 7, 20, 5, 0.0, 1.0), (5000, 19, 10, 1, 0.0, 1.0), (9000, 320, 12, 64, 0.0, 1.0),
    (20000, 151, 20, 128, 50.0, 1.0),     # rows far from the origin: the uncentred norms enter the bound of variant 1
    (20000, 302, 20, 33, 0.0, 1e-3), (20000, 302, 20, 96, -3.0, 40.0)])
def test_bounds_are_lower_bounds_of_the_exact_join_costs(engine, N, Dj, T, K, offset, scale):
    cand = _case(engine, N, Dj, T, K, seed=N % 89 + K, offset=offset, scale=scale)
    J = engine.join_costs(cand)
    tight = {}
    for variant in (1, 0):
        engine.set_option('join_lb_variant', variant)
        lo, sc = engine.join_bounds(cand)
        assert lo.shape == J.shape and lo.dtype == np.float32
        fin = np.isfinite(J)
        assert np.array_equal(np.isinf(lo), ~fin), variant           # +inf exactly where a unit is unusable
        l64 = lo.astype(np.float64)
        assert (l64[fin] >= 0).all()
        bad = l64[fin] > J[fin]
        assert not bad.any(), (variant, int(bad.sum()), float((l64[fin] - J[fin]).max()))
        assert np.isfinite(sc).all() and (sc >= 0).all()
        # tightness: share of the finite cells bounded to within 1 % (costs that are not tiny against the step's scale)
        big = fin & (J > 0.05 * sc[:, None, None])
        tight[variant] = float((l64[big] >= 0.99 * J[big]).mean()) if big.any() else 1.0
    engine.set_option('join_lb_variant', 1)
    print('join bounds Dj=%d K=%d offset=%g scale=%g: within 1%% of the cost: variant 1 %.4f, variant 0 %.4f'
          % (Dj, K, offset, scale, tight[1], tight[0]))
    if offset == 0.0:
        assert tight[1] >= 0.9 * tight[0] - 0.02, tight


@pytest.mark.parametrize('K', [129, 160, 200, 208])
def test_quadrants_of_wide_candidate_sets(engine, K):
    """Option join_lb_quadrants: pass 1 of K > 128 as 2 x 2 quadrants (four workgroups of four wavefronts with two accumulator
    sets instead of one of seven with one): still lower bounds, +inf in the same cells, the same scale of the step, at least as
    tight; the search through them returns what the dense recursion returns."""
    N, Dj, T = 9000, 302, 14
    cand = _case(engine, N, Dj, T, K, seed=K)
    J = engine.join_costs(cand)
    fin = np.isfinite(J)
    res = {}
    try:
        for q in (0, 1):
            engine.set_option('join_lb_quadrants', q)
            assert engine.info('join_lb_quadrants') == q
            lo, sc = engine.join_bounds(cand)
            assert np.array_equal(np.isinf(lo), ~fin)
            l64 = lo.astype(np.float64)
            assert (l64[fin] >= 0).all() and not (l64[fin] > J[fin]).any(), q
            res[q] = (l64, sc)
        assert np.array_equal(res[0][1], res[1][1])                 # the step's scale: the maximum over the quadrants
        gap0, gap1 = (J[fin] - res[0][0][fin]).mean(), (J[fin] - res[1][0][fin]).mean()
        assert gap1 <= gap0 * 1.001, (gap0, gap1)                   # two accumulator sets: the smaller error constant
        rng = np.random.RandomState(K)
        td = np.sort(rng.rand(T, K), axis=1)
        engine.set_option('viterbi_mode', 0)
        ref = engine.viterbi(cand, td)
        engine.set_option('viterbi_mode', 1)
        assert engine.viterbi(cand, td) == ref
        assert engine.info('join_bound_violations') == 0
    finally:
        engine.set_option('join_lb_quadrants', 0)
        engine.set_option('viterbi_mode', 2)


def test_natural_successors_and_repeated_calls(engine):
    """b = a + 1 joins at exactly 0.0: the bound of such a cell must be 0; a second call after new weights rebuilds the copy."""
    N, Dj, K = 5000, 151, 20
    F_unw, JC_unw = o.synthetic_db(N, 61, Dj, 3)
    engine.upload_db(F_unw, JC_unw)
    rows = np.stack([np.arange(100, 100 + K), np.arange(101, 101 + K), np.arange(102, 102 + K)]).astype(np.int64)
    for w in (0.1, 3.0):
        engine.set_weights(np.full(61, 0.5), np.full(Dj, w))
        J = engine.join_costs(rows)
        lo, _ = engine.join_bounds(rows)
        assert (np.diagonal(J, axis1=1, axis2=2) == 0.0).all()
        assert (np.diagonal(lo, axis1=1, axis2=2) == 0.0).all()
        assert (lo.astype(np.float64) <= J).all()


def test_tripwire_of_the_join_bounds_counts_what_it_should(engine):
    """Every exact join cost the sparse recursion computes is held against the float32 bound of its cell
    (joinfast_kernels.hip: join_bound_violations, join_bound_min_margin).  Honest bounds: no violation, a positive margin, over
    shapes and both forms of pass 1.  Bounds multiplied by 1.5 (test hook) are no bounds: the tripwire must say so."""
    N, Dj, T, K = 60000, 302, 200, 100
    F_unw, JC_unw = o.synthetic_db(N, 61, Dj, seed=9)
    wt, wj = np.full(61, 0.4), np.full(Dj, 0.05)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    rng = np.random.RandomState(3)
    U = np.vstack([o.synthetic_targets(F_unw, T // 2, seed=5), F_unw[rng.randint(0, N, T // 2)] + rng.randn(T // 2, 61)]) * wt
    engine.set_option('viterbi_mode', 1)
    try:
        for variant in (1, 0):
            engine.set_option('join_lb_variant', variant)
            engine.reset_timers()
            assert engine.info('join_bound_violations') == 0 and engine.info('join_bound_min_margin') == np.inf
            path, cost, cand, dist = engine.knn_viterbi(U, K, return_candidates=True)
            opath, ocost = oc.viterbi(cand, dist, o.weight(JC_unw, wj))
            assert path == opath and cost == ocost
            assert engine.info('join_bound_violations') == 0, (variant, engine.info('join_bound_min_margin'))
            m = engine.info('join_bound_min_margin')
            assert 0.0 < m < np.inf, (variant, m)
        engine.set_option('join_lb_variant', 1)
        engine.set_option('join_lb_test_scale', 1.5)
        engine.reset_timers()
        engine.knn_viterbi(U, K)
        assert engine.info('join_bound_violations') > 0 and engine.info('join_bound_min_margin') < 0.0
    finally:
        engine.set_option('join_lb_test_scale', 1.0)
        engine.set_option('join_lb_variant', 1)
        engine.set_option('viterbi_mode', 2)
        engine.reset_timers()
