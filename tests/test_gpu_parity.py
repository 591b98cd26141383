"""GPU parity tests (run on the MI355X box): the HIP path, called through the C ABI
(ctypes -> libsnkhip.so), against the CPU oracle on the same inputs and against the committed
reference-generated golden vectors.  Bar: bit-exact indices AND bit-exact float64 distances /
costs versus the oracle (same canonical summation order); rtol 1e-12 versus the reference's own
numpy/scipy values (different summation order)."""
import numpy as np
import pytest

import snk_oracle as o

pytestmark = pytest.mark.gpu

DIMS = {'mag': 60, 'real': 45, 'imag': 45, 'lf0': 1}


@pytest.fixture(scope='module')
def engine():
    import snickery_amd
    e = snickery_amd.HipSearchEngine(0)
    yield e
    e.close()


@pytest.fixture()
def mini_engine(engine, mini_voice):
    engine.upload_db(mini_voice['F_unw'], mini_voice['JC_unw'])
    engine.set_weights(mini_voice['wt'], mini_voice['wj'])
    return engine


def synth_setup(N, Dt, Dj, seed=0):
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed)
    rng = np.random.RandomState(seed + 100)
    wt = 0.2 + rng.rand(Dt)
    wj = 0.05 + 0.2 * rng.rand(Dj)
    F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
    return F_unw, JC_unw, wt, wj, F, E, S


def test_library_loaded_is_in_tree():
    import snickery_amd
    assert snickery_amd.library_path().endswith('snickery_amd/libsnkhip.so')
    assert snickery_amd.load_library().snk_abi_version() == 1


def test_mfma_f64_fragment_mapping(engine):
    assert engine.selftest_mfma() == 0.0


def test_knn_golden(mini_engine, golden, mini_voice):
    K = int(golden['knn_K'])
    cand, dist = mini_engine.knn(golden['knn_queries'], K)
    assert cand.dtype == np.int64 and dist.dtype == np.float64
    assert np.array_equal(cand, golden['knn_candidates'])                   # reference's own output
    np.testing.assert_allclose(dist, golden['knn_distances'], rtol=1e-12)   # reference summation order
    oc, od = o.knn_bruteforce(mini_voice['F'], golden['knn_queries'], K)
    assert np.array_equal(cand, oc)
    assert np.array_equal(dist, od)                                          # bit-exact vs oracle


@pytest.mark.parametrize('N,T,K,Dt', [(20000, 100, 50, 61), (5000, 37, 100, 61), (3000, 16, 7, 20),
                                      (9000, 50, 30, 184), (1500, 33, 200, 100), (700, 20, 16, 130),
                                      (40000, 70, 40, 100), (40000, 45, 60, 184), (30000, 40, 25, 250),
                                      (20000, 33, 20, 128)])
def test_knn_synthetic(engine, N, T, K, Dt):
    """Dt <= 63: one 64-column chunk; 100 / 184 / 250: the two-, three- and four-chunk f32 prefilter;
    128: no spare column for the norm -> f64 sweep.  Small N: too few sample groups -> f64 sweep."""
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(N, Dt, 24, seed=N % 97)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    U = o.synthetic_targets(F_unw, T, seed=3) * wt
    before = engine.info('f16_fallbacks')
    cand, dist = engine.knn(U, K)
    oc, od = o.knn_bruteforce(F, U, K)
    assert np.array_equal(cand, oc)
    assert np.array_equal(dist, od)
    if N >= 20000:
        assert engine.info('f16_ready') == (0 if Dt % 64 == 0 else 1)
        assert engine.info('f16_fallbacks') == before          # the prefilter's margin held


def test_knn_ties_and_padding(engine):
    """Exact duplicates (digital silence in real voices): order among equal distances is
    (distance, lower id); K > N pads with -1 / 1e15."""
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(400, 61, 24, seed=5)
    F_unw[100:140] = F_unw[50]                 # 41 identical rows
    F_unw[300] = F_unw[50]
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    F = o.weight(F_unw, wt)
    U = (F_unw[[50, 10, 399]] + 0.001) * wt
    cand, dist = engine.knn(U, 60)
    oc, od = o.knn_bruteforce(F, U, 60)
    assert np.array_equal(cand, oc) and np.array_equal(dist, od)
    assert list(cand[0, :42]) == [50] + list(range(100, 140)) + [300]
    small_F, small_JC = F_unw[:20], JC_unw[:21]
    engine.upload_db(small_F, small_JC)
    engine.set_weights(wt, wj)
    cand, dist = engine.knn(U, 32)
    oc, od = o.knn_bruteforce(o.weight(small_F, wt), U, 32)
    assert np.array_equal(cand, oc) and np.array_equal(dist, od)
    assert np.all(cand[:, 20:] == -1) and np.all(dist[:, 20:] == o.VERY_BIG_WEIGHT_VALUE)


def test_knn_list_overflow_retry(engine):
    """A tiny candidate-list capacity forces the overflow -> re-tighten -> retry path."""
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(30000, 61, 24, seed=11)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    U = o.synthetic_targets(F_unw, 48, seed=8) * wt
    engine.set_option('list_capacity', 192)
    engine.set_option('sample_fraction', 1.0 / 64)
    engine.set_option('precision', 0)          # the f64 sweep's own sampled thresholds
    try:
        cand, dist = engine.knn(U, 20)
        retries = engine.info('last_knn_retries')
    finally:
        engine.set_option('list_capacity', 4096)
        engine.set_option('sample_fraction', 1.0 / 16)
        engine.set_option('precision', 1)
    oc, od = o.knn_bruteforce(F, U, 20)
    assert np.array_equal(cand, oc) and np.array_equal(dist, od)
    assert retries >= 1


def test_knn_by_class(engine):
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(6000, 61, 24, seed=2)
    rng = np.random.RandomState(4)
    cls = rng.randint(0, 12, size=6000).astype(np.int32)
    cls[:5] = 77                                # a class with 5 members only
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    engine.set_unit_classes(cls)
    U = o.synthetic_targets(F_unw, 40, seed=9) * wt
    qc = rng.randint(0, 12, size=40).astype(np.int32)
    qc[7] = 77
    cand, dist = engine.knn_by_class(U, 10, qc)
    oc, od = o.knn_by_class(F, U, 10, cls, qc)
    assert np.array_equal(cand, oc) and np.array_equal(dist, od)
    assert list(cand[7, 5:]) == [-1] * 5


def test_join_costs_golden(mini_engine, golden, mini_voice):
    cand = golden['join_candidates']
    J = mini_engine.join_costs(cand)
    Jo = o.join_cost_dense(mini_voice['E'], mini_voice['S'], cand)
    assert np.array_equal(J, Jo)                                  # bit-exact incl. +inf pattern
    cache = dict(zip(map(tuple, golden['join_cache_keys'].tolist()), golden['join_cache_values']))
    ok = o.valid_mask(cand, mini_voice['F'].shape[0])
    n_nat = 0
    for t in range(cand.shape[0] - 1):
        for a in range(cand.shape[1]):
            for b in range(cand.shape[1]):
                if ok[t, a] and ok[t + 1, b]:
                    ref = cache[(int(cand[t, a]), int(cand[t + 1, b]))]
                    assert abs(J[t, a, b] - ref) <= 1e-12 * ref
                    if cand[t + 1, b] == cand[t, a] + 1:
                        assert J[t, a, b] == 0.0          # natural join: exactly zero
                        n_nat += 1
    assert n_nat > 0


@pytest.mark.parametrize('N,T,K,Dj', [(4000, 60, 50, 151), (2500, 30, 100, 302), (2000, 25, 13, 40),
                                      (3000, 20, 200, 151)])
def test_viterbi_synthetic(engine, N, T, K, Dj):
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(N, 61, Dj, seed=K)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    U = o.synthetic_targets(F_unw, T, seed=5) * wt
    cand, dist = o.knn_bruteforce(F, U, K)
    cand[2, 1] = -1
    cand[3, 0] = 0
    cand[4, 2] = N - 1
    J = engine.join_costs(cand)
    assert np.array_equal(J, o.join_cost_dense(E, S, cand))
    path, cost = engine.viterbi(cand, dist)
    opath, ocost = o.viterbi(cand, dist, E, S)
    assert path == opath
    assert cost == ocost
    p2, c2, cand2, dist2 = engine.knn_viterbi(U, K, return_candidates=True)
    oc, od = o.knn_bruteforce(F, U, K)
    assert np.array_equal(cand2, oc) and np.array_equal(dist2, od)
    op2, oc2 = o.viterbi(oc, od, E, S)
    assert p2 == op2 and c2 == oc2


def test_viterbi_golden_and_edges(mini_engine, golden, mini_voice):
    cand, dist = golden['join_candidates'], golden['knn_distances']
    path, cost = mini_engine.viterbi(cand, dist)
    opath, ocost = o.viterbi(cand, dist, mini_voice['E'], mini_voice['S'])
    assert path == opath and cost == ocost
    assert mini_engine.viterbi(cand[:1], dist[:1]) == ([], float('inf'))      # T < 2
    dead = cand.copy()
    dead[5, :] = -1                                                            # no usable unit
    assert mini_engine.viterbi(dead, dist) == ([], float('inf'))
    assert o.viterbi(dead, dist, mini_voice['E'], mini_voice['S']) == ([], np.inf)


def test_viterbi_batch_equals_single(engine):
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(8000, 61, 151, seed=21)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    utts = [o.synthetic_targets(F_unw, T, seed=s) * wt for s, T in [(1, 40), (2, 75), (3, 16), (4, 51), (5, 33)]]
    paths, costs = engine.knn_viterbi_batch(utts, 40)
    for u, U in enumerate(utts):
        p, c = engine.knn_viterbi(U, 40)
        assert list(paths[u]) == p and costs[u] == c
        oc, od = o.knn_bruteforce(F, U, 40)
        op, ocst = o.viterbi(oc, od, E, S)
        assert p == op and c == ocst


def test_viterbi_batch_groups(engine):
    """Many small K-NN groups (batch_rows far below the batch): group boundaries between and inside
    long utterances, alternating side streams and workspace reuse from the third group on; then one
    group for everything.  Same paths and costs either way, equal to the single-utterance calls."""
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(12000, 61, 151, seed=23)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    lens = [40, 130, 7, 64, 2, 90, 33, 1, 51, 77, 120]
    utts = [o.synthetic_targets(F_unw, T, seed=30 + i) * wt for i, T in enumerate(lens)]
    single = [engine.knn_viterbi(U, 30) for U in utts]
    default_rows = int(engine.info('batch_rows'))
    assert default_rows == 12288
    try:
        for rows in (100, 1, 0, 8192, default_rows):
            engine.set_option('batch_rows', rows)
            paths, costs = engine.knn_viterbi_batch(utts, 30)
            for u in range(len(utts)):
                assert list(paths[u]) == single[u][0] and (costs[u] == single[u][1] or len(single[u][0]) == 0)
    finally:
        engine.set_option('batch_rows', default_rows)
    assert len(paths[7]) == 0 and len(paths[4]) == 2          # T = 1: no path
    assert engine.info('batch_redos') == 0 and engine.info('f16_fallbacks') == 0


def test_viterbi_batch_redo_on_overflow(engine):
    """Batch pipeline with deferred K-NN status: an utterance whose sampled thresholds overflow a
    candidate list is redone with exact thresholds at the end of the batch."""
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(30000, 61, 40, seed=13)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    utts = [o.synthetic_targets(F_unw, T, seed=s) * wt for s, T in [(1, 33), (2, 48), (3, 20)]]
    engine.set_option('list_capacity', 192)
    engine.set_option('sample_fraction', 1.0 / 64)
    engine.set_option('precision', 0)          # the f64 sweep's own sampled thresholds
    try:
        before = engine.info('batch_redos')
        paths, costs = engine.knn_viterbi_batch(utts, 20)
        redos = engine.info('batch_redos') - before
    finally:
        engine.set_option('list_capacity', 4096)
        engine.set_option('sample_fraction', 1.0 / 16)
        engine.set_option('precision', 1)
    assert redos >= 1
    for u, U in enumerate(utts):
        oc, od = o.knn_bruteforce(F, U, 20)
        op, ocst = o.viterbi(oc, od, E, S)
        assert list(paths[u]) == op and costs[u] == ocst


@pytest.mark.parametrize('delay', [0, 3, 4])
def test_viterbi_batch_submit_collect(engine, delay):
    """Batches in flight (submit i+1, i+2 before collect i; three workspaces): results equal the one-call form and the
    oracle, in submission order or not; a fourth submit and a one-call batch are refused while three /
    any are pending; an overflowed group is redone at collect time.  delay 3 / 4: the Viterbi side of a group queued
    behind a point inside the next group's K-NN call, a batch's last group by the next submit or its own collect
    (join_bounds_delay, forced whatever the shape)."""
    import snickery_amd
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(30000, 61, 40, seed=17)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    engine.set_option('batch_rows', 64)                  # several groups per batch
    engine.set_option('join_bounds_delay', delay)
    try:
        batches = [[o.synthetic_targets(F_unw, T, seed=10 * b + i) * wt for i, T in enumerate(lens)]
                   for b, lens in enumerate([(33, 48, 20, 7), (50, 2, 61), (40, 40, 40, 40, 9)])]
        ref = [engine.knn_viterbi_batch(b, 20) for b in batches]
        for u, U in enumerate(batches[0]):
            oc, od = o.knn_bruteforce(F, U, 20)
            op, ocst = o.viterbi(oc, od, E, S)
            assert list(ref[0][0][u]) == op and ref[0][1][u] == ocst
        t0 = engine.knn_viterbi_batch_submit(snickery_amd.QueryBatch(batches[0]).pin(), 20)
        t1 = engine.knn_viterbi_batch_submit(batches[1], 20)
        t9 = engine.knn_viterbi_batch_submit(batches[2], 20)              # three workspaces: three batches in flight
        with pytest.raises(snickery_amd.SnkError):
            engine.knn_viterbi_batch_submit(batches[2], 20)              # ... a fourth is refused
        with pytest.raises(snickery_amd.SnkError):
            engine.knn_viterbi_batch(batches[2], 20)
        got9 = engine.knn_viterbi_batch_collect(t9)
        assert all(np.array_equal(a, b) for a, b in zip(got9[0], ref[2][0])) and np.array_equal(got9[1], ref[2][1])
        with pytest.raises(snickery_amd.SnkError):           # no re-weighting under a batch in flight
            engine.set_weights(wt, wj)
        got1 = engine.knn_viterbi_batch_collect(t1)          # out of order
        t2 = engine.knn_viterbi_batch_submit(batches[2], 20)
        got0 = engine.knn_viterbi_batch_collect(t0)
        got2 = engine.knn_viterbi_batch_collect(t2)
        with pytest.raises(snickery_amd.SnkError):
            engine.knn_viterbi_batch_collect(t2)
        for got, want in zip((got0, got1, got2), ref):
            assert all(np.array_equal(a, b) for a, b in zip(got[0], want[0])) and np.array_equal(got[1], want[1])
        # overflow -> redo while another batch is in flight
        engine.set_option('list_capacity', 192)
        engine.set_option('sample_fraction', 1.0 / 64)
        engine.set_option('precision', 0)
        before = engine.info('batch_redos')
        ta = engine.knn_viterbi_batch_submit(batches[0], 20)
        tb = engine.knn_viterbi_batch_submit(batches[2], 20)
        ga = engine.knn_viterbi_batch_collect(ta)
        gb = engine.knn_viterbi_batch_collect(tb)
        assert engine.info('batch_redos') - before >= 1
        for got, want in zip((ga, gb), (ref[0], ref[2])):
            assert all(np.array_equal(a, b) for a, b in zip(got[0], want[0])) and np.array_equal(got[1], want[1])
    finally:
        engine.set_option('list_capacity', 4096)
        engine.set_option('sample_fraction', 1.0 / 16)
        engine.set_option('precision', 1)
        engine.set_option('batch_rows', 12288)
        engine.set_option('join_bounds_delay', 1)


def test_batch_submit_with_resident_rows(engine):
    """Q == NULL (resident=True): the rows the workspace holds from its previous submit are searched again with the same
    results; rows of another shape, or a workspace that never received rows, are refused."""
    import snickery_amd
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(30000, 61, 40, seed=23)
    eng = snickery_amd.HipSearchEngine()         # a fresh engine: its three workspaces hold no rows yet
    eng.upload_db(F_unw, JC_unw)
    eng.set_weights(wt, wj)
    a = snickery_amd.QueryBatch([o.synthetic_targets(F_unw, T, seed=40 + i) * wt for i, T in enumerate((33, 48, 20, 7))])
    b = snickery_amd.QueryBatch([o.synthetic_targets(F_unw, T, seed=50 + i) * wt for i, T in enumerate((50, 2, 61))])
    with pytest.raises(snickery_amd.SnkError):
        eng.knn_viterbi_batch_submit(a, 20, resident=True)
    want_a = eng.knn_viterbi_batch_collect(eng.knn_viterbi_batch_submit(a, 20))        # workspace 0
    want_b = eng.knn_viterbi_batch_collect(eng.knn_viterbi_batch_submit(b, 20))        # workspace 1 (they take turns)
    with pytest.raises(snickery_amd.SnkError):   # workspace 2 never received rows
        eng.knn_viterbi_batch_submit(a, 20, resident=True)
    assert all(np.array_equal(x, y) for x, y in zip(eng.knn_viterbi_batch_collect(eng.knn_viterbi_batch_submit(a, 20))[0], want_a[0]))      # workspace 2
    ta = eng.knn_viterbi_batch_submit(a, 20, resident=True)                            # workspace 0 again: it holds the rows of `a`
    tb = eng.knn_viterbi_batch_submit(b, 20, resident=True)                            # workspace 1: the rows of `b`
    tc = eng.knn_viterbi_batch_submit(a, 20, resident=True)                            # workspace 2: the rows of `a`
    for got, want in ((eng.knn_viterbi_batch_collect(ta), want_a), (eng.knn_viterbi_batch_collect(tb), want_b), (eng.knn_viterbi_batch_collect(tc), want_a)):
        assert all(np.array_equal(x, y) for x, y in zip(got[0], want[0])) and np.array_equal(got[1], want[1])
    with pytest.raises(snickery_amd.SnkError):   # workspace 0 holds the rows of `a`, not of `b`
        eng.knn_viterbi_batch_submit(b, 20, resident=True)
    for u in range(len(a)):
        U = a.Q[a.offsets[u]:a.offsets[u + 1]]
        oc, od = o.knn_bruteforce(F, U, 20)
        op, ocst = o.viterbi(oc, od, E, S)
        assert list(want_a[0][u]) == op and want_a[1][u] == ocst
    eng.close()


def test_greedy_golden(mini_engine, golden, mini_voice):
    for me in (6, 1):
        mini_engine.set_greedy_layout(me, False, 0)
        for utt in (0, 1):
            U = golden['greedy_me%d_utt%d_unit_features' % (me, utt)]
            path, d = mini_engine.greedy(U, return_distances=True)
            assert np.array_equal(np.array(path), golden['greedy_me%d_utt%d_path' % (me, utt)])
            pr, cr, Fwin = o.greedy_layout(mini_voice['F'], mini_voice['E'], mini_voice['S'], me)
            op, od = o.greedy_search(pr, cr, Fwin, o.greedy_queries(U, me))
            assert path == op and np.array_equal(d, od)
    # natural path known answer (synth_simple.py:909-928)
    mini_engine.set_greedy_layout(1, False, 0)
    start = int(golden['greedy_me1_natural_start'])
    ref = golden['greedy_me1_natural_path']
    path, d = mini_engine.greedy(mini_voice['F'][start:start + len(ref)], start_state=start, return_distances=True)
    assert np.array_equal(np.array(path), ref) and np.all(d == 0.0)


@pytest.mark.parametrize('me,lfat,mode,Dj,Dt', [(6, False, 0, 151, 61), (3, True, 0, 40, 61), (4, False, 1, 80, 61),
                                               (1, False, 1, 302, 61), (16, False, 0, 151, 61),
                                               (5, False, 0, 151, 90),
                                               (5, False, 0, 151, 200),     # target rows too wide for LDS: re-read through L2
                                               (3, False, 1, 64, 250),
                                               (2, True, 0, 33, 7)])
def test_greedy_synthetic(engine, me, lfat, mode, Dj, Dt):
    N = 5000
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(N, Dt, Dj, seed=me)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    engine.set_greedy_layout(me, lfat, mode)
    U = o.synthetic_targets(F_unw, 63, seed=6) * wt
    pr, cr, Fwin = o.greedy_layout(F, E, S, me, lfat, mode)
    Q = o.greedy_queries(U, me, lfat)
    for start in (-1, 17):
        path, d = engine.greedy(U, start_state=start, return_distances=True)
        op, od = o.greedy_search(pr, cr, Fwin, Q, start_state=start)
        assert path == op
        assert np.array_equal(d, od)


def test_path_scores(engine):
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(3000, 61, 151, seed=31)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    U = o.synthetic_targets(F_unw, 30, seed=2) * wt
    path, cost = engine.knn_viterbi(U, 20)
    tsq, jsq = engine.path_scores(U, path, 0, 61, 151)
    assert np.array_equal(tsq, o.target_scores(F, U, path))
    assert np.array_equal(jsq, o.join_scores_viterbi(E, S, path))
    engine.set_greedy_layout(3, False, 0)
    gp = engine.greedy(U)
    pr, cr, Fwin = o.greedy_layout(F, E, S, 3)
    Q = o.greedy_queries(U, 3)
    tsq, jsq = engine.path_scores(U, gp, 1, 61 * 3, 151)
    assert np.array_equal(tsq, o.target_scores(Fwin, Q, gp))
    assert np.array_equal(jsq, o.join_scores_greedy(pr, cr, gp))


def test_error_behaviour(engine):
    import snickery_amd
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(500, 61, 24, seed=1)
    engine.upload_db(F_unw, JC_unw)
    with pytest.raises(snickery_amd.SnkError):
        engine.knn(np.zeros((4, 61)), 5)              # weights not set
    with pytest.raises(snickery_amd.SnkError):
        engine.set_weights(wt[:10], wj)               # wrong length
    engine.set_weights(wt, wj)
    with pytest.raises(snickery_amd.SnkError):
        engine.knn(np.zeros((4, 60)), 5)              # wrong dimension
    with pytest.raises(snickery_amd.SnkError):
        engine.greedy(np.zeros((4, 61)))              # layout not set


def test_knn_mass_duplicates(engine):
    """More database units exactly tied with the K-th neighbour than a candidate list holds (digital
    silence in a real voice): the rows fall through to the exact one-workgroup-per-row selection and
    come back in the reference order -- distance, then lowest unit id."""
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(30000, 61, 24, seed=17)
    F_unw = F_unw.copy()
    F_unw[2000:11500] = F_unw[1234]                # 9500 identical "silence" frames (> the longest list, 8192: the ladder's level 1)
    F_unw[11500:11550] = F_unw[1234] + 1e-4        # and a few near them
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    F = o.weight(F_unw, wt)
    U = np.vstack([(F_unw[1234] + 0.01) * wt, o.synthetic_targets(F_unw, 5, seed=2) * wt, F_unw[1234] * wt])
    before = engine.info('exact_row_fallbacks')
    cand, dist = engine.knn(U, 60)
    oc, od = o.knn_bruteforce(F, U, 60)
    assert np.array_equal(cand, oc) and np.array_equal(dist, od)
    assert engine.info('exact_row_fallbacks') - before >= 2
    assert list(cand[6, :3]) == [1234, 2000, 2001] and dist[6, 0] == 0.0
    # the same through the class-restricted search and the batch entry point
    cls = (np.arange(30000) % 3).astype(np.int32)
    engine.set_unit_classes(cls)
    qc = np.array([0, 1, 2, 0, 1, 2, 1], dtype=np.int32)
    cand, dist = engine.knn_by_class(U, 60, qc)
    oc, od = o.knn_by_class(F, U, 60, cls, qc)
    assert np.array_equal(cand, oc) and np.array_equal(dist, od)
    paths, costs = engine.knn_viterbi_batch([U, U[:3]], 60)
    oc, od = o.knn_bruteforce(F, U, 60)
    op, ocst = o.viterbi(oc, od, E, S)
    assert list(paths[0]) == op and costs[0] == ocst


def test_knn_entry_pool_exhausted_is_never_silent(engine):
    """The filter sweep appends survivors to a global entry pool; when the pool runs out it drops entries
    of ARBITRARY rows.  The first attempt's overflow must lead to a retry, and a retry that overflows
    again to the exact per-row selection of EVERY row -- never to rc 0 with truncated lists."""
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(40000, 61, 24, seed=19)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    U = o.synthetic_targets(F_unw, 40, seed=3) * wt
    oc, od = o.knn_bruteforce(F, U, 30)
    try:
        engine.set_option('pool_chunk_limit', 8)        # far fewer chunks than resident wavefronts
        for precision in (1, 0):
            engine.set_option('precision', precision)
            before, fb = engine.info('pool_overflows'), engine.info('f16_fallbacks')
            cand, dist = engine.knn(U, 30)
            assert np.array_equal(cand, oc) and np.array_equal(dist, od)
            # what is certain: 40 rows x ~1800 survivors never fit 8 chunks of 2048 entries -- the prefilter path hands the
            # call to the exact sweep, and the exact sweep's first attempt (sampled thresholds) is retried with exact
            # thresholds.  Whether that retry overflows as well (then every row goes through the exact per-row selection,
            # pool_overflows + 1) depends on which wavefronts happen to get the eight chunks: the survivors of these clustered
            # targets sit in a handful of slabs.
            if precision == 1:
                assert engine.info('f16_fallbacks') == fb + 1
            assert engine.info('last_knn_retries') == 1
            assert engine.info('pool_overflows') - before in (0, 1)
        paths, costs = engine.knn_viterbi_batch([U, U[:7]], 30)      # deferred status -> redo at collect
        op, ocst = o.viterbi(oc, od, E, S)
        assert list(paths[0]) == op and costs[0] == ocst
    finally:
        engine.set_option('pool_chunk_limit', 0)
        engine.set_option('precision', 1)


def test_column_selection_equals_dropped_columns(engine):
    """snk_set_column_selection (stream truncation): full-width queries and matrices with selected-out
    columns against the oracle on arrays with those columns DROPPED, as the reference does
    (synth_simple.py:982-992) -- K-NN, Viterbi (single and batch), greedy, per-column path scores."""
    N, Dt, Dj, K, T, me = 6000, 61, 40, 15, 36, 3
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(N, Dt, Dj, seed=41)
    tsel = list(range(40)) + [60]                         # mag truncated to 40, lf0 kept
    jsel = list(range(10)) + list(range(20, 33))
    engine.upload_db(F_unw, JC_unw)
    engine.set_column_selection(tsel, jsel)
    with pytest.raises(Exception):
        engine.knn(o.synthetic_targets(F_unw, 4, seed=1) * wt, K)        # weights must be set again
    engine.set_weights(wt, wj)
    try:
        U = o.synthetic_targets(F_unw, T, seed=5) * wt                   # full width, NOT masked by the caller
        Fd, Ed, Sd, Ud = F[:, tsel], E[:, jsel], S[:, jsel], U[:, tsel]
        cand, dist = engine.knn(U, K)
        oc, od = o.knn_bruteforce(Fd, Ud, K)
        assert np.array_equal(cand, oc) and np.array_equal(dist, od)
        path, cost = engine.viterbi(cand, dist)
        op, ocost = o.viterbi(oc, od, Ed, Sd)
        assert path == op and cost == ocost
        paths, costs = engine.knn_viterbi_batch([U[:20], U[20:]], K)
        for u, (a, b) in enumerate([(0, 20), (20, T)]):
            c2, d2 = o.knn_bruteforce(Fd, Ud[a:b], K)
            p2, cost2 = o.viterbi(c2, d2, Ed, Sd)
            assert list(paths[u]) == p2 and costs[u] == cost2
        engine.set_greedy_layout(me, False, 0)
        gp, gd = engine.greedy(U, return_distances=True)
        pr, cr, Fwin = o.greedy_layout(Fd, Ed, Sd, me)
        og, ogd = o.greedy_search(pr, cr, Fwin, o.greedy_queries(Ud, me))
        assert gp == og and np.array_equal(gd, ogd)
        tsq, jsq = engine.path_scores(U, path, 0, Dt, Dj)
        assert not np.any(tsq[:, 40:60]) and not np.any(jsq[:, 10:20]) and not np.any(jsq[:, 33:])
        np.testing.assert_allclose(tsq[:, tsel].sum(), o.target_scores(Fd, Ud, op).sum(), rtol=1e-12)
    finally:
        engine.set_column_selection(None, None)


@pytest.mark.parametrize('me,lfat,mode,Dj,Dt', [(6, False, 0, 151, 61), (3, True, 0, 40, 61), (4, False, 1, 80, 61),
                                               (5, False, 0, 151, 200)])
def test_greedy_batch_equals_single_and_oracle(engine, me, lfat, mode, Dj, Dt):
    """snk_greedy_batch (two utterances share every scan): ragged lengths, an odd number of
    utterances, one too short for a single step, start states -- each path and distance equal to
    snk_greedy's and the oracle's."""
    N = 5000
    F_unw, JC_unw, wt, wj, F, E, S = synth_setup(N, Dt, Dj, seed=20 + me)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    engine.set_greedy_layout(me, lfat, mode)
    lens = [63, 40, me - 1 if me > 1 else 1, 57, 22]
    utts = [o.synthetic_targets(F_unw, T, seed=30 + i) * wt for i, T in enumerate(lens)]
    starts = [-1, 17, -1, 0, 400]
    paths, dists = engine.greedy_batch(utts, start_states=starts, return_distances=True)
    pr, cr, Fwin = o.greedy_layout(F, E, S, me, lfat, mode)
    for u, U in enumerate(utts):
        sp, sd = engine.greedy(U, start_state=starts[u], return_distances=True)
        assert paths[u] == sp and np.array_equal(dists[u], sd)
        if len(U) >= me:
            op, od = o.greedy_search(pr, cr, Fwin, o.greedy_queries(U, me, lfat), start_state=starts[u])
            assert paths[u] == op and np.array_equal(dists[u], od)
        else:
            assert paths[u] == []
