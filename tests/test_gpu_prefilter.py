"""The bf16-split K-NN prefilter (knn16_kernels.hip knn_sweep16b): results stay those of the exact search, and the
error bound the prefilter is trusted to holds with room against float64 keys.  The prefilter only decides which units
reach the exact float64 re-rank; a key off by more than eps could drop a true neighbour, so the bound is the thing
to test, not only the end result."""
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
    e.set_option('prefilter', 1)
    e.close()


def setup(engine, N, Dt, seed, offset=0.0, scale=None):
    F_unw, JC_unw = o.synthetic_db(N, Dt, 24, seed)
    if scale is not None:
        F_unw = (F_unw * scale).astype(np.float32)
    if offset:
        F_unw = (F_unw + np.float32(offset)).astype(np.float32)
    rng = np.random.RandomState(seed + 100)
    wt = 0.2 + rng.rand(Dt)
    wj = 0.05 + 0.2 * rng.rand(24)
    F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    return F_unw, wt, F


@pytest.mark.parametrize('N,T,K,Dt', [(65536, 96, 100, 61), (40000, 64, 50, 45), (30000, 40, 30, 150), (50000, 33, 200, 189),
                                      (33333, 70, 16, 20), (40000, 50, 50, 123), (30000, 35, 100, 100), (20000, 33, 20, 125)])
def test_bf16_prefilter_results_are_the_exact_ones(engine, N, T, K, Dt):
    engine.set_option('prefilter', 1)
    F_unw, wt, F = setup(engine, N, Dt, seed=N % 89)
    assert engine.info('prefilter_bf16_active') == 1
    U = o.synthetic_targets(F_unw, T, seed=5) * wt
    before = engine.info('f16_fallbacks')
    cand, dist = engine.knn(U, K)
    oc, od = o.knn_bruteforce(F, U, K)
    assert np.array_equal(cand, oc)
    assert np.array_equal(dist, od)
    assert engine.info('f16_fallbacks') == before               # margins held: no exact re-sweep was needed
    engine.set_option('prefilter', 0)
    engine.set_weights(wt, np.full(24, 0.1))
    assert engine.info('prefilter_bf16_active') == 0
    c0, d0 = engine.knn(U, K)
    assert np.array_equal(c0, cand) and np.array_equal(d0, dist)


def test_shapes_without_a_bf16_variant_keep_float32_operands(engine):
    engine.set_option('prefilter', 1)
    for Dt in (250, 63, 127, 191):  # four chunks; no three spare columns (one, two, three chunks)
        F_unw, wt, F = setup(engine, 30000, Dt, seed=Dt)
        assert engine.info('prefilter_bf16_active') == 0
        U = o.synthetic_targets(F_unw, 20, seed=5) * wt
        cand, dist = engine.knn(U, 20)
        oc, od = o.knn_bruteforce(F, U, 20)
        assert np.array_equal(cand, oc) and np.array_equal(dist, od)


def slab_minima_f64(F, U, rows):
    n_slabs = (F.shape[0] + rows - 1) // rows
    keys = (F * F).sum(1)[None, :] - 2.0 * (U @ F.T)                 # (T, N) float64
    pad = n_slabs * rows - F.shape[0]
    if pad:
        keys = np.concatenate([keys, np.full((U.shape[0], pad), np.inf)], axis=1)
    return keys.reshape(U.shape[0], n_slabs, rows).min(axis=2)


@pytest.mark.parametrize('prefilter', [1, 0])
@pytest.mark.parametrize('Dt,offset,scale', [(61, 0.0, None), (61, 3.0, None), (150, 0.0, None), (189, -2.0, None),
                                             (61, 0.0, 37.0), (45, 100.0, 0.01), (123, 1.5, None)])
def test_prefilter_keys_stay_inside_their_bound(engine, prefilter, Dt, offset, scale):
    """|key~ - key| <= eps[t] is what the filter's margins assume.  Offsets make ||f|| large against the
    distances (cancellation: the hard case for the split operands); scale moves the exponent range."""
    engine.set_option('prefilter', prefilter)
    N, T = 32768 + 77, 64
    F_unw, wt, F = setup(engine, N, Dt, seed=Dt + 1, offset=offset, scale=scale)
    assert engine.info('prefilter_bf16_active') == prefilter
    U = o.synthetic_targets(F_unw, T, seed=9) * wt
    got, eps, rows = engine.prefilter_minima(U)
    want = slab_minima_f64(F, U, rows)
    assert got.shape == want.shape
    ratio = np.abs(got.astype(np.float64) - want) / eps[:, None]
    # the float32 result itself is rounded once more (2^-24 relative): far inside the bound
    print('prefilter %d Dt %d offset %g: max |key~ - key| / eps = %.4f (c_acc %.3g, rho_lo 2^%.2f, rho_res 2^%.2f)' % (
        prefilter, Dt, offset, ratio.max(), engine.info('prefilter_eps_c'),
        np.log2(max(engine.info('prefilter_rho_lo'), 1e-300)), np.log2(max(engine.info('prefilter_rho_res'), 1e-300))))
    assert ratio.max() <= 0.5, ratio.max()
    # and the exact search is still exact on this data
    cand, dist = engine.knn(U, 40)
    oc, od = o.knn_bruteforce(F, U, 40)
    assert np.array_equal(cand, oc) and np.array_equal(dist, od)


def _bf16_bits(x):
    """float64 array -> bf16 bit patterns (round to nearest even) and their exact values"""
    f = np.asarray(x, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16
    bits = u.astype(np.uint16)
    vals = (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    return bits, vals


def test_bf16_mfma_accumulation_stays_inside_the_assumed_bound(engine):
    """The bound of the bf16-split prefilter assumes that ONE v_mfma_f32_32x32x16_bf16 is off by at most
    prefilter_mfma_unit (2^-20) of the sum of its |products| and |C| (knn16_kernels.hip c_acc).  The unit's internal order
    is not documented, so it is probed: patterns built to hurt an implementation that truncates aligned products or
    rounds after every addition.  (A first version of the bound assumed 2^-22; pattern 0 refuted it at 1.6 x that.)"""
    unit = engine.info('prefilter_mfma_unit')
    assert unit == 2.0 ** -20
    rng = np.random.RandomState(0)
    worst = 0.0
    patterns = []
    # (1) one product of 1 beside fifteen just below half a unit in the last place of float32 (2^-25 + a bit each)
    a = np.full((32, 16), 2.0 ** -12); a[:, 0] = 1.0
    b = np.full((16, 32), 2.0 ** -13 * 1.9375); b[0, :] = 1.0
    patterns.append((a, b, np.zeros((32, 32))))
    # (1b) products just below two units of the cut the first pattern revealed (2^-25 of the largest term), with the
    #      largest term among the products and with C as the largest term
    b1 = np.full((16, 32), 2.0 ** -13 * 1.9921875); b1[0, :] = 1.0
    patterns.append((a, b1, np.zeros((32, 32))))
    a1 = np.full((32, 16), 2.0 ** -12)
    patterns.append((a1, np.full((16, 32), 2.0 ** -13 * 1.9921875), np.ones((32, 32))))
    patterns.append((a1, np.full((16, 32), -2.0 ** -13 * 1.9921875), np.ones((32, 32))))
    # (2) the same beside a large C
    patterns.append((a, b, np.full((32, 32), 1024.0)))
    # (3) cancellation: +x / -x pairs and a small remainder
    a3 = rng.randn(32, 16); b3 = rng.randn(16, 32)
    a3[:, 1::2] = a3[:, 0::2]; b3[1::2, :] = -b3[0::2, :] * (1 + 2.0 ** -7)
    patterns.append((a3, b3, np.zeros((32, 32))))
    # (4) exponents spread over the whole float32-relevant range
    patterns.append((rng.randn(32, 16) * 2.0 ** rng.randint(-20, 20, (32, 16)), rng.randn(16, 32) * 2.0 ** rng.randint(-20, 20, (16, 32)),
                     rng.randn(32, 32) * 2.0 ** rng.randint(-10, 30, (32, 32))))
    # (5) plain random data, C of the size of the sum (the sweep's situation), many draws
    for _ in range(40):
        patterns.append((rng.randn(32, 16) * 3, rng.randn(16, 32) * 3, rng.randn(32, 32) * 40))
    # (6) all products positive and equal: sixteen roundings in a row if the unit adds one by one
    patterns.append((np.full((32, 16), 1.0 + 2.0 ** -7), np.full((16, 32), 1.0 + 2.0 ** -7), np.full((32, 32), 1.0 / 3)))
    for pi, (a, b, c) in enumerate(patterns):
        ab, av = _bf16_bits(a)
        bb, bv = _bf16_bits(b)
        c32 = np.asarray(c, dtype=np.float32)
        got = engine.probe_mfma_bf16(ab, bb, c32).astype(np.float64)
        exact = av @ bv + c32.astype(np.float64)
        mass = np.abs(av) @ np.abs(bv) + np.abs(c32.astype(np.float64))
        ratio = np.abs(got - exact) / (unit * mass + 1e-300)
        print('   pattern %d: ratio %.3f  signed mean error / 2^-24 mass %.3f' % (pi, float(ratio.max()),
              float(np.mean((got - exact) / (2.0 ** -24 * mass + 1e-300)))))
        worst = max(worst, float(ratio.max()))
    print('one bf16 MFMA: max |D - exact| / (2^-20 (sum |products| + |C|)) = %.3f' % worst)
    assert worst <= 0.75, worst


@pytest.mark.parametrize('N,T,K,Dt,offset,scale', [(65536, 600, 100, 61, 0.0, None),        # the headline widths: one chunk, eight tiles resident
                                                   (50000, 333, 50, 100, 0.0, None),         # two chunks
                                                   (40000, 120, 100, 184, 0.0, None),        # three chunks (halfphone width)
                                                   (30000, 97, 30, 61, 6.0, None),           # norms far above the distances: the coarse pass lets most pairs through
                                                   (20000, 33, 200, 45, 50.0, 0.02),
                                                   (1500, 40, 20, 61, 0.0, None),            # fewer tiles than a wavefront keeps resident
                                                   (70001, 1, 16, 61, 0.0, None)])           # one query row
def test_two_pass_filter_selects_what_the_one_pass_filter_selects(engine, N, T, K, Dt, offset, scale):
    """The bf16-split filter as two passes (knn_coarse16b: hi.hi term alone against thr32 + e1 -> tile pairs; knn_refine16b:
    three-term keys of those pairs against thr32) must let through exactly what the one-pass sweep lets through -- the
    candidate lists in front of the exact re-rank have the same lengths row by row -- and the results are the oracle's."""
    engine.set_option('prefilter', 1)
    engine.set_option('prefilter_ball_bound', 0)        # the same thresholds for both filters (stage A' belongs to the two-pass form)
    F_unw, wt, F = setup(engine, N, Dt, seed=N % 97, offset=offset, scale=scale)
    assert engine.info('prefilter_bf16_active') == 1
    U = o.synthetic_targets(F_unw, T, seed=7) * wt
    oc, od = o.knn_bruteforce(F, U, K)
    lists = {}
    for two_pass in (1, 0):
        engine.set_option('prefilter_two_pass', two_pass)
        before = engine.info('f16_fallbacks')
        cand, dist = engine.knn(U, K)
        assert np.array_equal(cand, oc) and np.array_equal(dist, od)
        fell_back = engine.info('f16_fallbacks') != before
        # (norms far above the distances: the keys' error bound is wider than the gaps between neighbours, the lists of
        # either filter may overflow and the exact sweep then serves the call -- same results, checked above)
        assert not fell_back or offset != 0.0
        lists[two_pass] = None if fell_back else (engine.info('last_list_mean'), engine.info('last_list_max'))
    engine.set_option('prefilter_two_pass', 1)
    assert lists[1] == lists[0] or None in (lists[1], lists[0]), lists
    cand, dist = engine.knn(U, K)
    n_tiles = (N + 31) // 32
    pairs, over = engine.info('coarse_pairs'), engine.info('coarse_pair_overflow')
    assert (over == 0 or offset != 0.0) and 0 < pairs <= ((T + 31) // 32) * (n_tiles + 8)
    if offset == 0.0 and N >= 20000 and T >= 32:
        # the point of the coarse pass: most tile pairs never reach the three-term keys (4.5 % at B*, tools/knn_time.py)
        assert pairs < 0.5 * ((T + 31) // 32) * n_tiles, pairs
    # stage A' (the K-th smallest key among the units of the tiles nearest to a row as a second bound): same results, and
    # the lists in front of the exact re-rank do not get longer
    engine.set_option('prefilter_ball_bound', 1)         # (an option: off by default, HISTORY.md 4.1c)
    before = engine.info('f16_fallbacks')
    cand, dist = engine.knn(U, K)
    assert np.array_equal(cand, oc) and np.array_equal(dist, od)
    if engine.info('f16_fallbacks') == before and lists[1] is not None:
        assert engine.info('last_list_mean') <= lists[1][0] + 1e-9 and engine.info('last_list_max') <= lists[1][1]
        print('lists N=%d T=%d K=%d: mean %.0f -> %.0f, max %.0f -> %.0f with the scout bound'
              % (N, T, K, lists[1][0], engine.info('last_list_mean'), lists[1][1], engine.info('last_list_max')))
    engine.set_option('prefilter_ball_bound', 0)


def test_units_in_no_order_go_to_the_one_pass_sweep(engine):
    """A database whose units stand in random order: a tile of 32 holds unrelated frames, so the ball pass lists (nearly) every
    tile pair -> the voice goes to the coarse sweep; that one lists most pairs too -> the voice goes on to the one-pass
    three-term sweep (`filter_onepass`), which lists nothing.  Results are the oracle's at every stage of the descent, and once
    there the calls run without an overflowing list (no fallback to the exact sweep).  (`reorder 0`: the engine's own answer to such
    a voice -- an order of its own, test_a_voice_in_no_order_is_given_one -- is switched off to see the descent.)"""
    engine.set_option('prefilter', 1); engine.set_option('prefilter_two_pass', 1); engine.set_option('prefilter_ball_bound', 0)
    engine.set_option('reorder', 0)
    N, Dt, K, T = 65536, 61, 100, 600
    F0, JC0 = o.synthetic_db(N, Dt, 24, 5)
    perm = np.random.RandomState(3).permutation(N)
    F_unw, JC_unw = F0[perm], JC0[np.concatenate([perm, [N]])]
    rng = np.random.RandomState(105)
    wt, wj = 0.2 + rng.rand(Dt), 0.05 + 0.2 * rng.rand(24)
    F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    assert engine.info('filter_coarse') == 0 and engine.info('filter_onepass') == 0
    U = o.synthetic_targets(F_unw, T, seed=7) * wt
    oc, od = o.knn_bruteforce(F, U, K)
    seen = []
    for call in range(5):
        before = engine.info('f16_fallbacks')
        cand, dist = engine.knn(U, K)
        assert np.array_equal(cand, oc) and np.array_equal(dist, od)
        seen.append((int(engine.info('filter_coarse')), int(engine.info('filter_onepass')), int(engine.info('f16_fallbacks') - before)))
    assert seen[-1][:2] == (1, 1), seen                       # both latches set ...
    assert seen[-1][2] == 0 and seen[-2][2] == 0, seen         # ... and the calls behind them do not fall back
    # new weights: the voice is judged afresh
    engine.set_weights(wt * 1.5, wj)
    assert engine.info('filter_coarse') == 0 and engine.info('filter_onepass') == 0
    engine.set_option('reorder', 1)


def test_batches_on_units_in_no_order_settle_on_the_one_pass_sweep(engine):
    """The same descent (ball pass -> coarse sweep -> one-pass sweep) through the batch entry point, where the pairs a group listed
    are seen at the batch's collect: after a few batches the voice sits on the one-pass sweep and no group is redone any more;
    paths and costs equal the single-utterance calls' at every stage."""
    engine.set_option('prefilter', 1); engine.set_option('prefilter_two_pass', 1); engine.set_option('prefilter_ball_bound', 0)
    engine.set_option('reorder', 0)
    N, Dt, Dj, K = 65536, 61, 24, 50
    F0, JC0 = o.synthetic_db(N, Dt, Dj, 8)
    perm = np.random.RandomState(4).permutation(N)
    F_unw, JC_unw = F0[perm], JC0[np.concatenate([perm, [N]])]
    rng = np.random.RandomState(108)
    wt, wj = 0.2 + rng.rand(Dt), 0.05 + 0.2 * rng.rand(Dj)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    utts = [o.synthetic_targets(F_unw, T, seed=40 + i) * wt for i, T in enumerate([300, 280, 320, 300])]
    single = None
    redos = []
    for call in range(5):
        before = engine.info('batch_redos')
        paths, costs = engine.knn_viterbi_batch(utts, K)
        redos.append(int(engine.info('batch_redos') - before))
        if single is None:
            single = [engine.knn_viterbi(U, K) for U in utts]
        for u in range(len(utts)):
            assert list(paths[u]) == single[u][0] and costs[u] == single[u][1]
    assert engine.info('filter_coarse') == 1 and engine.info('filter_onepass') == 1, redos
    assert redos[-1] == 0 and redos[-2] == 0, redos
    engine.set_option('reorder', 1)


def test_a_voice_in_no_order_is_given_one(engine):
    """The engine's answer to units in no order (kmeans_kernels.hip): when the ball pass lists too many tile pairs the units are
    clustered and the prefilter's operands laid out cluster by cluster -- tiles are compact again, the ball pass serves the voice,
    nothing of the permutation reaches a result (candidates, distances, tie order: the oracle's on the database order).  A voice
    whose own order is already the better one (consecutive frames) keeps it even when a batch of far rows questions it."""
    engine.set_option('prefilter', 1); engine.set_option('prefilter_two_pass', 1); engine.set_option('prefilter_ball_bound', 0)
    engine.set_option('reorder', 1)
    N, Dt, K, T = 262144, 61, 100, 600
    F0, JC0 = o.synthetic_db(N, Dt, 8, 5)
    perm = np.random.RandomState(3).permutation(N)
    F_unw, JC_unw = F0[perm], JC0[np.concatenate([perm, [N]])]
    dup = np.random.RandomState(9).randint(0, N, 50)
    F_unw[dup] = F_unw[(dup + 7777) % N]                       # exact duplicates: ties must still go to the lower unit id
    rng = np.random.RandomState(105)
    wt, wj = 0.2 + rng.rand(Dt), np.full(8, 0.1)
    F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    U = o.synthetic_targets(F0, T, seed=7) * wt                # an utterance follows the speech the units were cut from, whatever their order
    U[:50] = F_unw[dup] * wt                                   # rows that ARE duplicated units
    rows = np.unique(np.concatenate([np.arange(0, 50, 7), np.linspace(50, T - 1, 12).astype(np.int64)]))
    oc_, od_ = oc.knn(F, U[rows], K)
    for call in range(4):
        cand, dist = engine.knn(U, K)
        assert np.array_equal(cand[rows], oc_) and np.array_equal(dist[rows], od_), call
    assert engine.info('reordered') == 1 and engine.info('reorders') == 1
    assert engine.info('reorder_radius_after') < 0.5 * engine.info('reorder_radius_before')
    assert engine.info('filter_coarse') == 0 and engine.info('filter_onepass') == 0       # the ball pass serves the voice again
    # the batch pipeline on the reordered voice, class-restricted search included
    utts = [U[:300], U[300:]]
    paths, costs = engine.knn_viterbi_batch(utts, K)
    for u, Uu in enumerate(utts):
        p1, c1 = engine.knn_viterbi(Uu, K)
        assert list(paths[u]) == p1 and costs[u] == c1
    cls = np.random.RandomState(4).randint(0, 7, N).astype(np.int32)
    engine.set_unit_classes(cls)
    qc = np.random.RandomState(5).randint(0, 7, T).astype(np.int32)
    cc, cd = engine.knn_by_class(U, K, qc)
    occ, ocd = o.knn_by_class(F, U[rows], K, cls, qc[rows])
    assert np.array_equal(cc[rows], occ) and np.array_equal(cd[rows], ocd)
    # new weights: the order stays (any order is valid), the voice is judged afresh and stays on the ball pass
    wt2 = wt * 1.3
    engine.set_weights(wt2, wj)
    U2 = o.synthetic_targets(F0, T, seed=7) * wt2
    U2[:50] = F_unw[dup] * wt2
    cand, dist = engine.knn(U2, K)
    oc2, od2 = oc.knn(o.weight(F_unw, wt2), U2[rows], K)
    assert np.array_equal(cand[rows], oc2) and np.array_equal(dist[rows], od2)
    assert engine.info('reordered') == 1 and engine.info('reorders') == 1 and engine.info('filter_coarse') == 0
    # a voice whose own order is the better one keeps it
    engine.upload_db(F0, JC0)
    engine.set_weights(wt, wj)
    far = (F0[rng.randint(0, N, 320)] + 6.0 * rng.randn(320, Dt)) * wt
    for _ in range(3):
        engine.knn(far, 50)
    assert engine.info('reordered') == 0 and engine.info('reorder_useless') == 1
    assert engine.info('reorder_radius_after') > engine.info('reorder_radius_before')


def test_the_filter_latches_are_rearmable(engine):
    """A voice does not stay on a slower filter for ever (the latches were one-way).  Queries far from every unit make the
    ball pass list most tile pairs -> the voice goes to the coarse sweep (and on, if that lists most pairs too); queries that follow
    the database then list next to nothing, and a counting probe of the pass the voice left (every 16 calls) takes it back, one
    rung at a time.  Results are the oracle's at every stage; `latch_rearm 0` keeps the voice where it is."""
    engine.set_option('prefilter', 1); engine.set_option('prefilter_two_pass', 1); engine.set_option('prefilter_ball_bound', 0)
    N, Dt, K, T = 200000, 61, 50, 320
    F_unw, JC_unw = o.synthetic_db(N, Dt, 8, 11)
    rng = np.random.RandomState(12)
    wt, wj = 0.2 + rng.rand(Dt), np.full(8, 0.1)
    F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    far = (F_unw[rng.randint(0, N, T)] + 6.0 * rng.randn(T, Dt)) * wt          # the K-th neighbour of such a row is far away: wide thresholds
    near = o.synthetic_targets(F_unw, T, seed=13) * wt
    for _ in range(3):
        cand, dist = engine.knn(far, K)
    oc, od = o.knn_bruteforce(F, far[:24], K)
    assert np.array_equal(cand[:24], oc) and np.array_equal(dist[:24], od)
    assert engine.info('filter_coarse') == 1                                   # the ball pass listed too much
    down = (engine.info('filter_coarse'), engine.info('filter_onepass'))
    oc, od = o.knn_bruteforce(F, near[:24], K)
    engine.set_option('latch_rearm', 0)
    for _ in range(40):
        cand, dist = engine.knn(near, K)
    assert (engine.info('filter_coarse'), engine.info('filter_onepass')) == down and engine.info('filter_rearms') == 0
    engine.set_option('latch_rearm', 1)
    states = []
    for _ in range(80):
        cand, dist = engine.knn(near, K)
        assert np.array_equal(cand[:24], oc) and np.array_equal(dist[:24], od)
        states.append((int(engine.info('filter_coarse')), int(engine.info('filter_onepass'))))
    assert states[-1] == (0, 0), states[::8]                                   # back on the ball pass
    assert engine.info('filter_rearms') >= 1
    # ... and down again when the far rows come back
    for _ in range(3):
        engine.knn(far, K)
    assert engine.info('filter_coarse') == 1


def test_optimistic_thresholds_are_proven_row_by_row_or_redone(engine):
    """Optimistic thresholds (api_knn.hip): the filter threshold is the j-th smallest sample minimum with j < K -- an estimate
    that makes the lists a few K long instead of ~18 K.  Never trusted: the re-rank proves every row's list complete (exact
    K-th key + eps under the threshold) or flags the call, which is redone with the guaranteed thresholds.  Here: the default
    rank shortens the lists and changes nothing; a rank of 1 (thresholds far too low) flags nearly every row, the results stay
    the oracle's through the redo -- single calls and batches -- and after a few failures the voice stops trying."""
    engine.set_option('prefilter', 1)
    N, Dt, K, T = 300000, 61, 100, 96
    F_unw, wt, F = setup(engine, N, Dt, seed=41)
    U = o.synthetic_targets(F_unw, T, seed=6) * wt
    ocand, odist = oc.knn(F, U, K)
    engine.set_option('tau_optimism', 0)
    c0, d0 = engine.knn(U, K)
    long_lists = engine.info('last_list_mean')
    assert engine.info('tau_optimism_rank') == 0
    engine.set_option('tau_optimism', 1)
    fails = engine.info('tau_optimism_failures')
    c1, d1 = engine.knn(U, K)
    j = engine.info('tau_optimism_rank')
    assert 1 <= j < K and engine.info('tau_optimism_failures') == fails
    assert engine.info('last_list_mean') < 0.5 * long_lists, (engine.info('last_list_mean'), long_lists)
    for c, d in ((c0, d0), (c1, d1)):
        assert np.array_equal(c, ocand) and np.array_equal(d, odist)
    # thresholds that are far too low: every call is flagged and redone; the third failure turns the voice's optimism off
    engine.set_option('tau_optimism_rank', 1)
    for i in range(3):
        c2, d2 = engine.knn(U, K)
        assert np.array_equal(c2, ocand) and np.array_equal(d2, odist)
        assert engine.info('tau_optimism_failures') == fails + i + 1
    assert engine.info('tau_optimism_off') == 1
    c3, d3 = engine.knn(U, K)
    assert engine.info('tau_optimism_rank') == 0 and engine.info('tau_optimism_failures') == fails + 3
    assert np.array_equal(c3, ocand) and np.array_equal(d3, odist)
    # a batch (deferred status words: the flagged group is redone at the collect), two in flight
    engine.set_option('tau_optimism_rank', 1)                  # (re-arms the voice)
    utts = [o.synthetic_targets(F_unw, t, seed=60 + i) * wt for i, t in enumerate((40, 64, 33))]
    redos = engine.info('batch_redos')
    ta, tb = engine.knn_viterbi_batch_submit(utts, K), engine.knn_viterbi_batch_submit(utts[:2], K)
    pa, ca = engine.knn_viterbi_batch_collect(ta)
    pb, cb = engine.knn_viterbi_batch_collect(tb)
    assert engine.info('batch_redos') > redos
    engine.set_option('tau_optimism_rank', 0)
    engine.set_option('tau_optimism', 0)
    pr, cr = engine.knn_viterbi_batch(utts, K)
    engine.set_option('tau_optimism', 1)
    assert all(np.array_equal(a, b) for a, b in zip(pa, pr)) and np.array_equal(ca, cr)
    assert all(np.array_equal(a, b) for a, b in zip(pb, pr[:2])) and np.array_equal(cb, cr[:2])
    po, co = engine.knn_viterbi_batch(utts, K)                 # the default rank again
    assert all(np.array_equal(a, b) for a, b in zip(po, pr)) and np.array_equal(co, cr)
