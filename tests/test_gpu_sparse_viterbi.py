"""The Viterbi path that keeps the join costs off the float64 vector pipe (joinfast_kernels.hip: f32 matrix
lower bounds, predecessor sets, exact float64 costs for the sets only, verified exact recursion) against
the dense exact path (viterbi_kernels.hip) and the oracle: same path, same cost, bit for bit -- whatever
the margin, including margins so small that every column is recomputed densely."""
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
    e.set_option('viterbi_mode', 2)
    e.close()


def _db(N, Dt, Dj, seed):
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed)
    rng = np.random.RandomState(seed + 100)
    wt = 0.2 + rng.rand(Dt)
    wj = 0.05 + 0.2 * rng.rand(Dj)
    return F_unw, JC_unw, wt, wj


@pytest.mark.parametrize('N,Dj,T,K', [(4000, 151, 40, 12), (20000, 302, 75, 50), (20000, 151, 60, 100),
                                      (30000, 40, 33, 16), (20000, 302, 30, 128), (30000, 151, 25, 200),
                                      (5000, 7, 50, 5), (5000, 19, 20, 1), (9000, 320, 20, 64)])
def test_sparse_equals_dense_and_oracle(engine, N, Dj, T, K):
    F_unw, JC_unw, wt, wj = _db(N, 61, Dj, seed=N % 97 + K)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    JCw = o.weight(JC_unw, wj)
    rng = np.random.RandomState(K)
    # half the rows follow the database (candidates are time neighbours: tiny, nearly equal join costs),
    # half are random frames (candidates from all over the database)
    U = np.vstack([o.synthetic_targets(F_unw, T - T // 2, seed=5), F_unw[rng.randint(0, N, T // 2)] + rng.randn(T // 2, 61)]) * wt
    engine.set_option('viterbi_mode', 0)
    path0, cost0, cand, dist = engine.knn_viterbi(U, K, return_candidates=True)
    opath, ocost = oc.viterbi(cand, dist, JCw)
    assert path0 == opath and cost0 == ocost
    engine.set_option('viterbi_mode', 1)
    for beta in (5e-4, 0.0, 1.0):              # default; nothing but the minimum in the sets; overflowing sets
        engine.set_option('join_beta', beta)
        path1, cost1 = engine.viterbi(cand, dist)
        assert path1 == opath and cost1 == ocost, beta
    engine.set_option('join_beta', 5e-4)
    p2, c2 = engine.knn_viterbi(U, K)
    assert p2 == opath and c2 == ocost
    paths, costs = engine.knn_viterbi_batch([U, U[:T // 2], U[:1], U[3:9]], K)
    assert list(paths[0]) == opath and costs[0] == ocost
    assert len(paths[2]) == 0


@pytest.mark.parametrize('waves,K', [(1, 40), (4, 40), (1, 120)])
def test_chunked_lower_bound_recursion_and_both_exact_recursions(engine, waves, K):
    """Pass 2 in chunks side by side (viterbi_lb_chunk / viterbi_lb_warm: chunk lengths around the shift period of 64,
    chunks of one step, a warm-up of one step, a warm-up longer than the utterance) and pass 4 on one or on four compute
    wavefronts: path and cost equal the oracle's bit for bit; only the number of refined cells may move."""
    N, Dj = 20000, 151                                     # K = 40 and 120: the <= 64 and the <= 128 column instances of pass 2
    F_unw, JC_unw, wt, wj = _db(N, 61, Dj, seed=211)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    JCw = o.weight(JC_unw, wj)
    rng = np.random.RandomState(9)
    T = 200
    U = np.vstack([o.synthetic_targets(F_unw, 120, seed=6), F_unw[rng.randint(0, N, T - 120)] + rng.randn(T - 120, 61)]) * wt
    engine.set_option('viterbi_mode', 0)
    path0, cost0, cand, dist = engine.knn_viterbi(U, K, return_candidates=True)
    opath, ocost = oc.viterbi(cand, dist, JCw)
    assert path0 == opath and cost0 == ocost
    # an unusable unit and a duplicate in the middle of a chunk and on a chunk boundary
    cand = cand.copy(); cand[64, 3] = 0; cand[65, 5] = cand[65, 6]; cand[97, :] = np.roll(cand[97, :], 1)
    opath, ocost = oc.viterbi(cand, dist, JCw)
    engine.set_option('viterbi_mode', 1)
    engine.set_option('viterbi_sparse_waves', waves)
    try:
        for chunk, warm in ((0, 16), (1, 1), (7, 3), (32, 16), (63, 5), (64, 64), (65, 1), (48, 500), (199, 16), (500, 16)):
            engine.set_option('viterbi_lb_chunk', chunk)
            engine.set_option('viterbi_lb_warm', warm)
            path1, cost1 = engine.viterbi(cand, dist)
            assert path1 == opath and cost1 == ocost, (chunk, warm)
            paths, costs = engine.viterbi_batch([cand, cand[:70], cand[:1], cand[60:131]], [dist, dist[:70], dist[:1], dist[60:131]])
            assert list(paths[0]) == opath and costs[0] == ocost, (chunk, warm)
            p70, c70 = oc.viterbi(cand[:70], dist[:70], JCw)
            assert list(paths[1]) == p70 and costs[1] == c70, (chunk, warm)
            pm, cm = oc.viterbi(cand[60:131], dist[60:131], JCw)
            assert list(paths[3]) == pm and costs[3] == cm, (chunk, warm)
    finally:
        engine.set_option('viterbi_lb_chunk', 48)
        engine.set_option('viterbi_lb_warm', 16)
        engine.set_option('viterbi_sparse_waves', 1)


def test_sparse_with_unusable_duplicate_and_padded_candidates(engine):
    """Quinphone-style lists: duplicated ids inside a column (exact ties between predecessors: the lower
    slot must win), -1 padding, units 0 and N-1 (no join state), a column with no usable unit."""
    N, Dj, T, K = 3000, 151, 30, 24
    F_unw, JC_unw, wt, wj = _db(N, 61, Dj, seed=3)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    JCw = o.weight(JC_unw, wj)
    rng = np.random.RandomState(1)
    base = rng.randint(1, N - 1, size=(T, K)).astype(np.int64)
    base[:, 1] = base[:, 0]                    # duplicates
    base[:, 5] = base[:, 2]
    base[4, 7:] = -1
    base[9, 0] = 0
    base[9, 3] = N - 1
    base[12:14, :] = np.arange(100, 100 + K)   # natural successors between rows 12 -> 13 shifted by one
    base[13, :] += 1
    dist = np.sort(rng.rand(T, K), axis=1) + 0.5
    dist[:, 1] = dist[:, 0]
    for mode in (0, 1):
        engine.set_option('viterbi_mode', mode)
        path, cost = engine.viterbi(base, dist)
        opath, ocost = oc.viterbi(base, dist, JCw)
        assert path == opath and cost == ocost
    dead = base.copy()
    dead[20, :] = -1                           # no path at all
    for mode in (0, 1):
        engine.set_option('viterbi_mode', mode)
        path, cost = engine.viterbi(dead, dist)
        assert path == [] and cost == np.inf
    engine.set_option('viterbi_mode', 1)
    # the default (mode 2) picks the sparse path for one utterance too (chunked pass 2, one-wavefront pass 4)
    engine.set_option('viterbi_mode', 2)
    before = engine.timers().get('viterbi_sparse', (0, 0))[1]
    engine.viterbi(base, dist)
    assert engine.timers().get('viterbi_sparse', (0, 0))[1] == before + 1
    engine.set_option('viterbi_mode', 1)


def test_long_utterances_batch_sparse_equals_dense(engine):
    """Long utterances (the float32 lower-bound recursion shifts its values every 64 steps; pass 4's loader
    ring wraps many times; back-pointers beyond the LDS budget) in one batch: sparse == dense, bit for bit."""
    N, Dj, K = 300000, 302, 100
    F_unw, JC_unw = o.synthetic_db(N, 61, Dj, seed=14)
    wt, wj = np.full(61, 0.4), np.full(Dj, 0.05)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    rng = np.random.RandomState(15)
    utts = [o.synthetic_targets(F_unw, T, seed=20 + i) * wt for i, T in enumerate([600, 257, 1900, 64, 65, 129])]
    utts.append((F_unw[rng.randint(0, N, 300)] + 0.5 * rng.randn(300, 61)) * wt)      # candidates from all over the database
    engine.set_option('viterbi_mode', 0)
    p0, c0 = engine.knn_viterbi_batch(utts, K)
    engine.set_option('viterbi_mode', 1)
    for beta in (5e-4, 0.0):
        engine.set_option('join_beta', beta)
        p1, c1 = engine.knn_viterbi_batch(utts, K)
        assert all(np.array_equal(a, b) for a, b in zip(p0, p1)) and np.array_equal(c0, c1), beta
    engine.set_option('join_beta', 5e-4)
    path, cost, cand, dist = engine.knn_viterbi(utts[2], K, return_candidates=True)
    opath, ocost = oc.viterbi(cand, dist, o.weight(JC_unw, wj))
    assert path == opath and cost == ocost and list(p0[2]) == opath


def test_exact_join_costs_are_a_small_fraction(engine):
    """Speed guard: sets + refinements together evaluate a small fraction of the K x K exact join costs."""
    N, Dj, T, K = 200000, 302, 300, 100
    F_unw, JC_unw = o.synthetic_db(N, 61, Dj, seed=4)
    wt, wj = np.full(61, 0.4), np.full(Dj, 0.05)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    U = o.synthetic_targets(F_unw, T, seed=5) * wt
    engine.set_option('viterbi_mode', 1)
    before = engine.info('dense_exact_costs')
    path, cost, cand, dist = engine.knn_viterbi(U, K, return_candidates=True)
    refined = engine.info('dense_exact_costs') - before
    opath, ocost = oc.viterbi(cand, dist, o.weight(JC_unw, wj))
    assert path == opath and cost == ocost
    assert refined <= 0.01 * T * K * K, refined


def test_auto_mode_takes_the_sparse_path_for_batches_at_every_k(engine):
    """viterbi_mode 2: a batch goes through the sparse path also at K = 200 (BASELINE config 4; those instances
    spill registers and are still 1.5x faster than the dense kernels) and equals the dense result."""
    N, Dj = 150000, 302
    F_unw, JC_unw = o.synthetic_db(N, 61, Dj, seed=31)
    wt, wj = np.full(61, 0.4), np.full(Dj, 0.05)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    utts = [o.synthetic_targets(F_unw, T, seed=40 + i) * wt for i, T in enumerate([120, 77, 200])]
    for K in (200, 150, 100):
        engine.set_option('viterbi_mode', 0)
        p0, c0 = engine.knn_viterbi_batch(utts, K)
        engine.set_option('viterbi_mode', 2)
        before = engine.timers().get('viterbi_sparse', (0, 0))[1]
        p2, c2 = engine.knn_viterbi_batch(utts, K)
        assert engine.timers().get('viterbi_sparse', (0, 0))[1] > before, K
        assert all(np.array_equal(a, b) for a, b in zip(p0, p2)) and np.array_equal(c0, c2), K



def test_viterbi_batch_of_given_candidates_equals_the_single_calls(engine):
    """snk_viterbi_batch (label-driven preselection: the caller brings the candidates): ragged utterances with
    padding ids, unusable units, a one-row utterance; equal to snk_viterbi per utterance and to the oracle."""
    N, Dj, K = 60000, 151, 40
    F_unw, JC_unw, wt, wj = _db(N, 61, Dj, seed=71)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    JCw = o.weight(JC_unw, wj)
    rng = np.random.RandomState(72)
    cands, dists = [], []
    for T in (50, 1, 33, 120, 2):
        c = np.sort(rng.randint(1, N - 1, (T, K)), axis=1).astype(np.int64)
        c[rng.rand(T, K) < 0.05] = -1                      # back-off padding
        if T > 10:
            c[7, 3] = 0; c[9, 5] = N - 1                   # first / last unit: never joinable
        d = np.sort(rng.rand(T, K), axis=1) + 0.3
        cands.append(c); dists.append(d)
    engine.set_option('viterbi_mode', 2)
    paths, costs = engine.viterbi_batch(cands, dists)
    for u, (c, d) in enumerate(zip(cands, dists)):
        p1, c1 = engine.viterbi(c, d)
        op, oc_ = oc.viterbi(c, d, JCw)
        assert list(paths[u]) == p1 == op and (costs[u] == c1 == oc_ or len(op) == 0)


def test_viterbi_latch_judges_the_two_exact_paths_and_keeps_the_results(engine):
    """viterbi_mode 2, batches (snk_engine.h: vit).  On join rows with natural successors the bounds prune and no trial of the dense
    kernels is ever started; on join rows in random order (no natural successors: pass 4 refines instead of pruning) the dense
    kernels are tried and whichever path has the shorter batch period is kept -- paths and costs are the dense path's bit for bit
    at every stage, and a later trial can bring the other path back (re-armable)."""
    N, Dj, K = 60000, 151, 60
    F_unw, JC_unw = o.synthetic_db(N, 61, Dj, seed=17)
    wt, wj = np.full(61, 0.4), np.full(Dj, 0.05)
    utts = [o.synthetic_targets(F_unw, T, seed=60 + i) * wt for i, T in enumerate([150, 90, 200, 120])]
    engine.set_option('viterbi_mode', 2)

    def run(JC, n_batches):
        engine.upload_db(F_unw, JC)
        engine.set_weights(wt, wj)
        engine.set_option('viterbi_mode', 0)
        ref = engine.knn_viterbi_batch(utts, K)
        engine.set_option('viterbi_mode', 2)
        engine.set_weights(wt, wj)                    # the latch starts over
        pending = None
        for _ in range(n_batches):
            tk = engine.knn_viterbi_batch_submit(utts, K)
            if pending is not None:
                p, c = engine.knn_viterbi_batch_collect(pending)
                assert all(np.array_equal(a, b) for a, b in zip(p, ref[0])) and np.array_equal(c, ref[1])
            pending = tk
        p, c = engine.knn_viterbi_batch_collect(pending)
        assert all(np.array_equal(a, b) for a, b in zip(p, ref[0])) and np.array_equal(c, ref[1])
        return engine.info('viterbi_latch_trials'), engine.info('viterbi_latch_mode')

    trials, mode = run(JC_unw, 24)
    assert trials == 0 and mode == 0                  # natural successors: the sparse path is never questioned
    perm = np.random.RandomState(5).permutation(N + 1)
    trials, mode = run(JC_unw[perm], 48)
    assert trials >= 1, (trials, mode)                # no natural successors: the dense kernels got their trial (either may win)
    engine.set_option('viterbi_latch', 0)
    trials, mode = run(JC_unw[perm], 12)
    assert trials == 0 and mode == 0
    engine.set_option('viterbi_latch', 1)


@pytest.mark.parametrize('delay', [0, 1, 2, 3, 4, 5])
def test_delayed_viterbi_side_in_every_collect_order(engine, delay):
    """join_bounds_delay (where the Viterbi side of a group starts; the last group of a batch is left to the next submit or to its own
    collect) against delay 0: two batches in flight collected in order and out of order, a batch collected before the next is
    submitted, batches of one group and of several, a batch cut into two groups -- the same paths and costs."""
    N, Dt, Dj, K = 60000, 61, 151, 40
    F_unw, JC_unw, wt, wj = _db(N, Dt, Dj, seed=17)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    engine.set_option('viterbi_mode', 2)
    rng = np.random.RandomState(3)
    big = [o.synthetic_targets(F_unw, T, seed=200 + i) * wt for i, T in enumerate((700, 650, 720, 600, 680, 710, 640, 690, 600, 660))]   # > 6 144 rows: two groups
    small = [o.synthetic_targets(F_unw, T, seed=300 + i) * wt for i, T in enumerate((40, 64, 33))]                                       # one group: all tail
    engine.set_option('batch_rows', 2048)
    several = big[:6]                                                                                                                    # four groups of 2 048 rows
    engine.set_option('batch_rows', 12288)
    engine.set_option('join_bounds_delay', 0)
    want = {}
    for name, batch, rows in (('big', big, 12288), ('small', small, 12288), ('several', several, 2048)):
        engine.set_option('batch_rows', rows)
        want[name] = engine.knn_viterbi_batch(batch, K)
    engine.set_option('join_bounds_delay', delay)

    def same(got, name):
        return all(np.array_equal(a, b) for a, b in zip(got[0], want[name][0])) and np.array_equal(got[1], want[name][1])
    try:
        for name, batch, rows in (('big', big, 12288), ('small', small, 12288), ('several', several, 2048)):
            engine.set_option('batch_rows', rows)
            other = 'small' if name != 'small' else 'big'
            ob = small if name != 'small' else big
            if rows != 12288:
                other, ob = name, batch
            # in order
            ta, tb = engine.knn_viterbi_batch_submit(batch, K), engine.knn_viterbi_batch_submit(ob, K)
            ga, gb = engine.knn_viterbi_batch_collect(ta), engine.knn_viterbi_batch_collect(tb)
            assert same(ga, name) and same(gb, other), (delay, name, 'in order')
            # out of order: the batch submitted last is collected first
            ta, tb = engine.knn_viterbi_batch_submit(batch, K), engine.knn_viterbi_batch_submit(ob, K)
            gb, ga = engine.knn_viterbi_batch_collect(tb), engine.knn_viterbi_batch_collect(ta)
            assert same(ga, name) and same(gb, other), (delay, name, 'out of order')
            # one at a time, then three submits with a collect in between
            assert same(engine.knn_viterbi_batch_collect(engine.knn_viterbi_batch_submit(batch, K)), name)
            ta = engine.knn_viterbi_batch_submit(batch, K)
            tb = engine.knn_viterbi_batch_submit(ob, K)
            ga = engine.knn_viterbi_batch_collect(ta)
            tc = engine.knn_viterbi_batch_submit(batch, K)
            gb, gc = engine.knn_viterbi_batch_collect(tb), engine.knn_viterbi_batch_collect(tc)
            assert same(ga, name) and same(gb, other) and same(gc, name), (delay, name, 'three')
    finally:
        engine.set_option('join_bounds_delay', 1)
        engine.set_option('batch_rows', 12288)
    assert engine.info('join_bound_violations') == 0


def test_wide_candidate_sets_take_one_group_per_batch(engine):
    """K > 128 with option wide_one_group (off by default: measured slower at B4): a batch that fits one K-NN call is ONE group and consecutive batches take the two side
    streams / Viterbi workspaces in turn, so that two batches' recursions run side by side.  Same paths and costs as two groups
    per batch, with one, two and three batches in flight, collected in and out of order."""
    N, Dt, Dj, K = 60000, 61, 151, 160
    F_unw, JC_unw, wt, wj = _db(N, Dt, Dj, seed=29)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    engine.set_option('viterbi_mode', 2)
    batches = [[o.synthetic_targets(F_unw, T, seed=400 + 10 * b + i) * wt for i, T in enumerate(lens)]
               for b, lens in enumerate([(300, 280, 310, 290, 305, 295, 300, 310, 290, 300, 280, 310, 290, 305, 295, 300, 310, 290, 300, 280, 310, 290),
                                         (120, 90), (700, 650, 720, 600, 680, 710, 640, 690, 600, 660)])]
    engine.set_option('wide_one_group', 0)
    want = [engine.knn_viterbi_batch(b, K) for b in batches]
    engine.set_option('wide_one_group', 1)

    def same(got, i):
        return all(np.array_equal(a, b) for a, b in zip(got[0], want[i][0])) and np.array_equal(got[1], want[i][1])
    try:
        for i, b in enumerate(batches):
            assert same(engine.knn_viterbi_batch(b, K), i)
        t = [engine.knn_viterbi_batch_submit(b, K) for b in batches]                     # three in flight: streams 0, 1, 0
        got = [engine.knn_viterbi_batch_collect(x) for x in t]
        assert all(same(g, i) for i, g in enumerate(got))
        t = [engine.knn_viterbi_batch_submit(b, K) for b in batches]
        g2, g0, g1 = engine.knn_viterbi_batch_collect(t[2]), engine.knn_viterbi_batch_collect(t[0]), engine.knn_viterbi_batch_collect(t[1])
        assert same(g0, 0) and same(g1, 1) and same(g2, 2)
        ta = engine.knn_viterbi_batch_submit(batches[0], K)
        tb = engine.knn_viterbi_batch_submit(batches[0], K)                              # the same batch twice: both streams
        assert same(engine.knn_viterbi_batch_collect(ta), 0) and same(engine.knn_viterbi_batch_collect(tb), 0)
    finally:
        engine.set_option('wide_one_group', 0)
    assert engine.info('join_bound_violations') == 0
