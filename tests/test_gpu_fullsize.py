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


def test_config_b2_shape_k50(engine):
    """BASELINE configs[1] shape (slt_arctic full, magphase-60 targets, K=50, search_epsilon=0
    Viterbi): whole K-NN + Viterbi against the C oracle."""
    N, Dt, Dj, T, K = 300000, 61, 302, 240, 50
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=12)
    wt = np.full(Dt, 0.5)
    wj = np.full(Dj, 0.04)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    F, JCw = o.weight(F_unw, wt), o.weight(JC_unw, wj)
    U = o.synthetic_targets(F_unw, T, seed=13) * wt
    path, cost, cand, dist = engine.knn_viterbi(U, K, return_candidates=True)
    oc_cand, oc_dist = oc.knn(F, U, K)
    assert np.array_equal(cand, oc_cand) and np.array_equal(dist, oc_dist)
    opath, ocost = oc.viterbi(oc_cand, oc_dist, JCw)
    assert path == opath and cost == ocost


def test_config_b5_shape_halfphone_classes(engine):
    """BASELINE configs[4] shape (synth_halfphone: wide linguistic+acoustic target vector,
    monophone-restricted K-NN, K=100): class-restricted lists against the oracle on sampled rows,
    then the Viterbi over the padded lists against the C oracle's DP."""
    N, Dt, Dj, T, K = 400000, 184, 151, 120, 100
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=21)
    rng = np.random.RandomState(22)
    wt = 0.1 + 0.5 * rng.rand(Dt)
    wj = np.full(Dj, 0.05)
    cls = rng.randint(0, 45, size=N).astype(np.int32)
    cls[1000:1060] = 99                          # a rare phone: 60 units < K
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    engine.set_unit_classes(cls)
    F, JCw = o.weight(F_unw, wt), o.weight(JC_unw, wj)
    U = o.synthetic_targets(F_unw, T, seed=23) * wt
    qc = rng.randint(0, 45, size=T).astype(np.int32)
    qc[50] = 99
    cand, dist = engine.knn_by_class(U, K, qc)
    rows = list(range(0, T, 7)) + [50]
    ocand, odist = o.knn_by_class(F, U[rows], K, cls, qc[rows])
    assert np.array_equal(cand[rows], ocand) and np.array_equal(dist[rows], odist)
    assert np.all(cls[cand[cand >= 0].reshape(-1)] == np.repeat(qc, K)[(cand >= 0).reshape(-1)])
    assert list(cand[50, 60:]) == [-1] * 40
    path, cost = engine.viterbi(cand, dist)
    opath, ocost = oc.viterbi(cand, dist, JCw)
    assert path == opath and cost == ocost


def test_config_b4_shape_two_shards_k200(engine):
    """BASELINE configs[3] shape (row-sharded DB, K=200, all-gather of shard-local top-K, one
    Viterbi): two half-database shards on this GPU stand in for two ranks."""
    import torch
    import snickery_amd
    from snickery_amd.dist import shard_bounds
    N, Dt, Dj, T, K = 500000, 61, 302, 96, 200
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=31)
    wt = np.full(Dt, 0.4)
    wj = np.full(Dj, 0.05)
    F, JCw = o.weight(F_unw, wt), o.weight(JC_unw, wj)
    U = o.synthetic_targets(F_unw, T, seed=32) * wt
    dev = torch.device('cuda', 0)
    d2_all = torch.empty(2, T, K, dtype=torch.float64, device=dev)
    id_all = torch.empty(2, T, K, dtype=torch.int64, device=dev)
    shards = []
    for r in range(2):
        lo, hi = shard_bounds(N, 2, r)
        e = snickery_amd.HipSearchEngine(0)
        e.upload_target_only(F_unw[lo:hi])
        e.set_shard(lo, N)
        e.set_weights(wt, None)
        e.knn_local_dev(U, K, d2_all[r].data_ptr(), id_all[r].data_ptr())
        shards.append(e)
    torch.cuda.synchronize()
    cand, dist = shards[0].merge_topk_dev(d2_all.data_ptr(), id_all.data_ptr(), 2, T, K)
    oc_cand, oc_dist = oc.knn(F, U, K)
    assert np.array_equal(cand, oc_cand) and np.array_equal(dist, oc_dist)
    for e in shards:
        e.close()
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    path, cost = engine.viterbi(cand, dist)
    opath, ocost = oc.viterbi(oc_cand, oc_dist, JCw)
    assert path == opath and cost == ocost


def test_sharded_batch_pipeline_on_one_gpu(engine):
    """The two device halves of ShardedSearch.knn_viterbi_batch with the exchange done by hand:
    4 shard engines on this GPU produce their local lists for a ragged batch, the (G, R_own, K)
    blocks an all-to-all would deliver are sliced out, and the owner-side merge + Viterbi must
    reproduce the unsharded batch search exactly."""
    import torch
    import snickery_amd
    from snickery_amd.dist import shard_bounds
    N, Dt, Dj, K, G = 262144, 61, 151, 100, 4
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=41)
    wt = np.full(Dt, 0.4)
    wj = np.full(Dj, 0.07)
    utts = [o.synthetic_targets(F_unw, T, seed=50 + i) * wt for i, T in enumerate([300, 64, 411, 2, 150])]
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    ref_paths, ref_costs = engine.knn_viterbi_batch(utts, K)
    lens = [u.shape[0] for u in utts]
    R = sum(lens)
    dev = torch.device('cuda', 0)
    d2 = torch.empty(G, R, K, dtype=torch.float64, device=dev)
    ids = torch.empty(G, R, K, dtype=torch.int64, device=dev)
    fallbacks = 0
    for r in range(G):
        lo, hi = shard_bounds(N, G, r)
        e = snickery_amd.HipSearchEngine(0)
        e.upload_target_only(F_unw[lo:hi])
        e.set_shard(lo, N)
        e.set_weights(wt, None)
        if r % 2:
            e.set_option('batch_rows', 200)       # row groups that cut through utterances
        e.knn_local_batch_dev(utts, K, d2[r].data_ptr(), ids[r].data_ptr())
        fallbacks += e.info('f16_fallbacks') + e.info('batch_redos')
        e.close()
    torch.cuda.synchronize()
    assert fallbacks == 0                      # shard-sized databases keep the f32 prefilter path
    # "rank" 1 of 2 owners: utterances 3..4; "rank" 0: utterances 0..2
    engine.set_option('batch_rows', 256)           # several recursion groups on the owner as well
    for a, b in [(0, 3), (3, 5)]:
        r0, r1 = sum(lens[:a]), sum(lens[:b])
        d2_own = d2[:, r0:r1].contiguous()
        id_own = ids[:, r0:r1].contiguous()
        torch.cuda.synchronize()                   # the copies ran on torch's stream, the engine has its own
        paths, costs = engine.merge_viterbi_batch_dev(d2_own.data_ptr(), id_own.data_ptr(), G, lens[a:b], K)
        for j, u in enumerate(range(a, b)):
            assert np.array_equal(paths[j], ref_paths[u]) and costs[j] == ref_costs[u]
    engine.set_option('batch_rows', 12288)
    assert len(ref_paths[3]) == 2


def test_very_long_utterance(engine):
    """An utterance longer than the rows of one K-NN call (32768): the search is cut by rows inside
    the library, join costs / recursion run over the whole utterance (back-pointers in global memory)."""
    N, Dt, Dj, T, K = 20000, 61, 40, 33000, 16
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=61)
    wt = np.full(Dt, 0.5)
    wj = np.full(Dj, 0.1)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    F, JCw = o.weight(F_unw, wt), o.weight(JC_unw, wj)
    rng = np.random.RandomState(62)
    U = (F_unw[rng.randint(0, N, T)] + 0.3 * rng.randn(T, Dt)) * wt
    path, cost, cand, dist = engine.knn_viterbi(U, K, return_candidates=True)
    oc_cand, oc_dist = oc.knn(F, U, K)
    assert np.array_equal(cand, oc_cand) and np.array_equal(dist, oc_dist)
    opath, ocost = oc.viterbi(oc_cand, oc_dist, JCw)
    assert path == opath and cost == ocost
    paths, costs = engine.knn_viterbi_batch([U[:100], U, U[:7]], K)
    assert list(paths[1]) == path and costs[1] == cost


@pytest.mark.parametrize('Dt,chunks', [(61, 1), (100, 2), (184, 3), (250, 4)])
def test_f32_prefilter_is_not_slower_than_the_f64_sweep(engine, Dt, chunks):
    """Performance guard for every chunk count of the f32 filter sweep (one kernel variant each): it
    has four times the matrix rate of the float64 sweep, so it must not lose to it.  (A change that
    made the three-chunk variant spill registers once cost 4x at the halfphone width and went
    unnoticed because only the headline shape was timed.)"""
    N, T, K = 300000, 2048, 100
    F_unw, JC_unw = o.synthetic_db(N, Dt, 4, seed=3)
    wt = np.full(Dt, 0.4)
    U = o.synthetic_targets(F_unw, T, seed=4) * wt
    times = {}
    for precision in (1, 0):
        engine.set_option('precision', precision)
        engine.upload_db(F_unw, JC_unw)
        engine.set_weights(wt, np.full(4, 0.1))
        engine.knn(U, K)
        engine.reset_timers()
        for _ in range(2):
            cand, dist = engine.knn(U, K)
        times[precision] = engine.timers()['knn_filter'][0]
        if precision == 1:
            assert engine.info('f16_fallbacks') == 0
            ref = (cand, dist)
        else:
            assert np.array_equal(cand, ref[0]) and np.array_equal(dist, ref[1])
    engine.set_option('precision', 1)
    assert 1.25 * times[1] < times[0], (Dt, times)        # measured ratios: 1.6 (one chunk) to 2.15 (four)


@pytest.mark.parametrize('N,Dt,Dj,T,K,U', [(700000, 61, 302, 300, 50, 4),        # B2
                                           (1048576, 61, 302, 300, 100, 4),      # B*
                                           (1500000, 61, 302, 300, 200, 4),      # B4 (K = 200)
                                           (1300000, 184, 151, 120, 100, 8)])    # B5 (three-point halfphone width)
def test_baseline_shapes_stay_on_the_fast_path(engine, N, Dt, Dj, T, K, U):
    """Every BASELINE shape must run the f32 prefilter without falling back to the f64 sweep and
    without redoing a group (silent, correct, 2x slower: K = 200 once overflowed the candidate
    lists sized for K <= 128 on every call), and agree with the f64 path."""
    from bench import synthetic_db, synthetic_targets
    F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
    wt, wj = np.full(Dt, 0.4), np.full(Dj, 0.05)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    utts = [synthetic_targets(F_unw, T, seed=1 + u) * wt for u in range(U)]
    before = (engine.info('f16_fallbacks'), engine.info('batch_redos'), engine.info('exact_row_fallbacks'))
    engine.reset_timers()                           # (also the tripwire's counters)
    paths, costs = engine.knn_viterbi_batch(utts, K)
    cand, dist = engine.knn(utts[0], K)
    assert (engine.info('f16_fallbacks'), engine.info('batch_redos'), engine.info('exact_row_fallbacks')) == before
    # tripwire of the prefilter's key bound: no row's exact K-th key came within twice the assumed key error of its
    # filter threshold -- a 2x violation of the bf16 accumulation assumption would not have changed these results
    assert engine.info('prefilter_margin_rows') == 0, engine.info('prefilter_min_margin')
    assert engine.info('prefilter_min_margin') >= 2.0
    # ... and of the join bounds of the sparse Viterbi path (joinfast_kernels.hip): every exact join cost the recursion computed was
    # held against the float32 lower bound of its cell -- none came out below it, and the smallest margin is positive
    assert engine.info('join_bound_violations') == 0, engine.info('join_bound_min_margin')
    assert engine.info('join_bound_min_margin') > 0.0
    engine.set_option('precision', 0)
    try:
        engine.set_weights(wt, wj)
        paths0, costs0 = engine.knn_viterbi_batch(utts, K)
        cand0, dist0 = engine.knn(utts[0], K)
    finally:
        engine.set_option('precision', 1)
    assert all(np.array_equal(a, b) for a, b in zip(paths, paths0)) and np.array_equal(costs, costs0)
    assert np.array_equal(cand, cand0) and np.array_equal(dist, dist0)
    # the oracle at the configuration's own N (not only the f64 sweep of the same library): sixteen query rows against
    # the C oracle's brute force over the WHOLE database, and the whole Viterbi of the first utterance against the C
    # oracle's recursion on the device's candidates
    F = o.weight(F_unw, wt)
    rows = np.unique(np.linspace(0, T - 1, 16).astype(np.int64))
    oc_cand, oc_dist = oc.knn(F, utts[0][rows], K)
    assert np.array_equal(cand[rows], oc_cand) and np.array_equal(dist[rows], oc_dist)
    del F
    JCw = o.weight(JC_unw, wj)
    opath, ocost = oc.viterbi(cand, dist, JCw)
    assert [int(v) for v in paths[0]] == opath and costs[0] == ocost


def test_speechlike_voice_at_headline_size(engine):
    """bench.py's speech-like leg at its own size: N = 1 048 576 units whose target AND join features are AR(1)
    walks (consecutive frames a sizeable fraction of the data's spread apart: no compact tiles, no near-contiguous candidates,
    join bounds that prune little), utterances from a HELD-OUT walk.  Sixteen query rows against the C oracle's brute force
    over the whole database, the whole Viterbi of every utterance of a batch against the C oracle's recursion, tripwires clean."""
    from bench import speechlike_voice
    N, Dt, Dj, T, K, U = 1048576, 61, 302, 600, 100, 4
    F_unw, JC_unw, held_out = speechlike_voice(N, Dt, Dj, seed=0)
    wt, wj = np.full(Dt, 0.4), np.full(Dj, 0.05)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    utts = [held_out(T, u) * wt for u in range(U)]
    engine.reset_timers()
    for _ in range(3):                              # the engine judges the voice (filter latch, unit order, Viterbi path) over its first batches
        paths, costs = engine.knn_viterbi_batch(utts, K)
    assert engine.info('exact_row_fallbacks') == 0
    # join costs that hardly differ between candidates: pass 2's chunks need the long warm-up (snk_engine.h lb_warm_eff); the first
    # batch's refinements took the voice there
    assert engine.info('viterbi_lb_warm_now') == 48, engine.info('dense_cells')
    assert engine.info('prefilter_margin_rows') == 0, engine.info('prefilter_min_margin')
    assert engine.info('join_bound_violations') == 0, engine.info('join_bound_min_margin')
    cd = [engine.knn(u, K) for u in utts]
    F = o.weight(F_unw, wt)
    rows = np.unique(np.linspace(0, T - 1, 16).astype(np.int64))
    oc_cand, oc_dist = oc.knn(F, utts[0][rows], K)
    assert np.array_equal(cd[0][0][rows], oc_cand), 'candidates of the spot rows differ from the C oracle'
    assert np.array_equal(cd[0][1][rows], oc_dist), 'distances of the spot rows differ from the C oracle'
    del F
    JCw = o.weight(JC_unw, wj)
    for u in range(U):
        opath, ocost = oc.viterbi(cd[u][0], cd[u][1], JCw)
        assert [int(v) for v in paths[u]] == opath, 'utterance %d: path differs from the C oracle' % u
        assert costs[u] == ocost, ('utterance %d' % u, costs[u], ocost)
    # the dense exact kernels (what the Viterbi latch may pick for such a voice) return the same
    engine.set_option('viterbi_mode', 0)
    try:
        p0, c0 = engine.knn_viterbi_batch(utts, K)
    finally:
        engine.set_option('viterbi_mode', 2)
    assert all(np.array_equal(a, b) for a, b in zip(paths, p0)) and np.array_equal(costs, c0)
