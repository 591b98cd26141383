"""The hoisted target term of the greedy search (greedy_hoist_kernels.hip: ||Q[s] - Fwin[i]||^2 of all steps and
windows as one float64 matrix product per utterance; the scan then streams the join columns only).  It only
prefilters: paths and distances must stay the oracle's bit for bit (synth_simple.py:458-503), in every layout, with
large norms (the expansion cancels), after a change of weights, and when the product is not used at all."""
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


def _setup(engine, N, Dt, Dj, seed, me, lfat, mode, offset=0.0):
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed)
    if offset:
        F_unw = (F_unw + np.float32(offset)).astype(np.float32)
    rng = np.random.RandomState(seed + 100)
    wt, wj = 0.2 + rng.rand(Dt), 0.05 + 0.2 * rng.rand(Dj)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    engine.set_greedy_layout(me, lfat, mode)
    return F_unw, JC_unw, wt, wj


@pytest.mark.parametrize('me,lfat,mode,Dj,Dt,offset', [(6, False, 0, 151, 61, 0.0), (3, True, 0, 100, 61, 0.0), (4, False, 1, 302, 61, 0.0),
                                                      (1, False, 0, 70, 61, 0.0), (6, False, 0, 151, 61, 8.0), (5, True, 1, 200, 130, -3.0),
                                                      (2, False, 0, 96, 64, 0.0), (7, False, 0, 151, 65, 0.0)])
def test_hoisted_scan_equals_oracle(engine, me, lfat, mode, Dj, Dt, offset):
    N = 30000 + me
    engine.set_option('greedy_mode', 2)
    engine.set_option('greedy_hoist', 1)
    F_unw, JC_unw, wt, wj = _setup(engine, N, Dt, Dj, seed=3 * me + Dt, me=me, lfat=lfat, mode=mode, offset=offset)
    # 1, 16, 17 and 33 steps (the product works on blocks of 16 steps), an utterance shorter than a window
    lens = [33 * me + (me - 1), 16 * me, 17 * me, me, max(me - 1, 1)]
    utts = [o.synthetic_targets(F_unw, T, seed=6 + i) * wt for i, T in enumerate(lens)]
    starts = [-1, 17, N - me - 3, 0, -1]
    before = engine.info('greedy_hoist_launches')
    for U, st in zip(utts[:3], starts[:3]):           # one utterance per call: the hoisted scan in auto mode
        path, d = engine.greedy(U, start_state=st, return_distances=True)
        op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, lfat, mode, st)
        assert path == op and np.array_equal(d, od)
    assert engine.info('greedy_hoist_launches') == before + 3
    paths, dists = engine.greedy_batch(utts, start_states=starts, return_distances=True)
    for U, st, p, d in zip(utts, starts, paths, dists):
        op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, lfat, mode, st)
        assert p == op and np.array_equal(d, od)
    assert engine.info('greedy_fallbacks') == 0
    # other weights: the window norms of the product follow them
    wt2, wj2 = wt[::-1].copy() * 1.7, wj * 0.3
    engine.set_weights(wt2, wj2)
    U2 = o.synthetic_targets(F_unw, 20 * me, seed=77) * wt2
    path, d = engine.greedy(U2, return_distances=True)
    op, od = oc.greedy_f32(F_unw, JC_unw, wt2, wj2, U2, me, lfat, mode, -1)
    assert path == op and np.array_equal(d, od)
    # the same without the product (switched off; and refused for its size): nothing but the time may change
    for opt, val in (('greedy_hoist', 0), ('greedy_hoist_max_gb', 1e-6)):
        engine.set_option('greedy_hoist', 1); engine.set_option('greedy_hoist_max_gb', 48.0)
        engine.set_option(opt, val)
        n0 = engine.info('greedy_hoist_launches')
        p3, d3 = engine.greedy_batch([U2, U2[:7 * me]], return_distances=True)
        assert engine.info('greedy_hoist_launches') == n0
        assert p3[0] == op and np.array_equal(d3[0], od) and p3[1] == op[:7] and np.array_equal(d3[1], od[:7])
    engine.set_option('greedy_hoist', 1); engine.set_option('greedy_hoist_max_gb', 48.0)


def test_narrow_join_streams_keep_the_unhoisted_scan(engine):
    """Fewer than three 32-column join chunks: the scan computes the target term itself (greedy_hoist_supported)."""
    engine.set_option('greedy_mode', 2)
    F_unw, JC_unw, wt, wj = _setup(engine, 20000, 61, 40, seed=5, me=4, lfat=False, mode=0)
    U = o.synthetic_targets(F_unw, 40, seed=1) * wt
    n0 = engine.info('greedy_hoist_launches')
    paths, dists = engine.greedy_batch([U, U[:20]], return_distances=True)
    op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, 4, False, 0, -1)
    assert paths[0] == op and np.array_equal(dists[0], od) and engine.info('greedy_hoist_launches') == n0


def test_ties_and_near_ties_with_the_hoisted_term(engine):
    """Duplicated speech (exact three-way ties: lowest index by float64 totals), clean targets (distance 0: the
    product's ABSOLUTE bound is all there is) and targets a hair away from the database."""
    N, Dt, Dj, me = 60000, 61, 151, 6
    engine.set_option('greedy_mode', 2)
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, 31)
    for dst in (25000, 47011):
        F_unw[dst:dst + 400] = F_unw[3000:3400]
        JC_unw[dst:dst + 401] = JC_unw[3000:3401]
    rng = np.random.RandomState(131)
    wt, wj = 0.2 + rng.rand(Dt), 0.05 + 0.2 * rng.rand(Dj)
    engine.upload_db(F_unw, JC_unw); engine.set_weights(wt, wj); engine.set_greedy_layout(me, False, 0)
    U = F_unw[3100:3100 + 20 * me].astype(np.float64) * wt
    n0, f0 = engine.info('greedy_hoist_launches'), engine.info('greedy_fallbacks')
    path, d = engine.greedy(U, start_state=3100, return_distances=True)
    assert path == list(range(3100, 3100 + 20 * me, me)) and np.all(d == 0.0)
    Un = (F_unw[3100:3100 + 10 * me].astype(np.float64) + 1e-7 * rng.randn(10 * me, Dt)) * wt
    path, d = engine.greedy(Un, return_distances=True)
    op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, Un, me, False, 0, -1)
    assert path == op and np.array_equal(d, od)
    assert engine.info('greedy_hoist_launches') == n0 + 2
    assert engine.info('greedy_fallbacks') == f0
    # search_epsilon > 0 on clean targets: the float32 minimum may only be taken where the bound is small against it
    for eps in (10.0, 0.05):
        pe, de = engine.greedy(U, start_state=3100, search_epsilon=eps, return_distances=True)
        assert np.all(de == 0.0)                       # nearest distance 0: (1 + eps) 0 leaves no slack at all


def test_six_utterances_share_a_scan(engine):
    """With the product a scan serves up to six utterances (wide table entries); without it, three: a batch of seven
    ragged utterances must come out as one by one in both settings."""
    N, Dt, Dj, me = 40000, 61, 151, 6
    engine.set_option('greedy_mode', 2); engine.set_option('greedy_hoist', 1)
    F_unw, JC_unw, wt, wj = _setup(engine, N, Dt, Dj, seed=11, me=me, lfat=False, mode=0)
    lens = [50 * me, 49 * me + 3, 12 * me, 31 * me, me, 50 * me, 7 * me]
    utts = [o.synthetic_targets(F_unw, T, seed=40 + i) * wt for i, T in enumerate(lens)]
    utts[5] = F_unw[7000:7000 + 50 * me].astype(np.float64) * wt          # clean targets: the natural path, distance 0
    starts = [-1, 5, -1, 39000, 0, 7000, -1]
    want = [oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, False, 0, st) for U, st in zip(utts, starts)]
    for hoist in (1, 0):
        engine.set_option('greedy_hoist', hoist)
        n0 = engine.info('greedy_hoist_launches')
        paths, dists = engine.greedy_batch(utts, start_states=starts, return_distances=True)
        for (op, od), p, d in zip(want, paths, dists):
            assert p == op and np.array_equal(d, od)
        assert engine.info('greedy_hoist_launches') - n0 == (2 if hoist else 0)      # 6 + 1 utterances
    assert paths[5] == list(range(7000, 7000 + 50 * me, me))
    assert engine.info('greedy_fallbacks') == 0
    engine.set_option('greedy_hoist', 1)


def test_watchdog_ends_a_launch_that_cannot_finish(engine):
    """The one-launch scan waits inside the kernel for its other workgroups.  If one of them never arrives (a device
    shared with another spinning launch; here: a test hook), the wait must end by itself -- seconds, not forever --
    and the exact scan, which never waits inside a kernel, finishes the call with the same result."""
    import time
    N, Dt, Dj, me = 30000, 61, 151, 6
    engine.set_option('greedy_mode', 2); engine.set_option('greedy_hoist', 1)
    F_unw, JC_unw, wt, wj = _setup(engine, N, Dt, Dj, seed=21, me=me, lfat=False, mode=0)
    U = o.synthetic_targets(F_unw, 10 * me, seed=3) * wt
    op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, False, 0, -1)
    s0, f0 = engine.info('greedy_stalls'), engine.info('greedy_fallbacks')
    engine.set_option('greedy_test_stall', 1)
    t0 = time.time()
    try:
        path, d = engine.greedy(U, return_distances=True)
    finally:
        engine.set_option('greedy_test_stall', 0)
    assert time.time() - t0 < 30.0
    assert path == op and np.array_equal(d, od)
    assert engine.info('greedy_stalls') == s0 + 1 and engine.info('greedy_fallbacks') == f0 + 1
    path, d = engine.greedy(U, return_distances=True)                       # and the next launch is a normal one
    assert path == op and np.array_equal(d, od) and engine.info('greedy_stalls') == s0 + 1


@pytest.mark.parametrize('me,lfat,mode,Dj,Dt,offset', [(6, False, 0, 151, 61, 0.0), (3, True, 0, 100, 61, 0.0), (4, False, 1, 302, 61, 0.0),
                                                      (1, False, 0, 70, 61, 2.5), (5, True, 1, 200, 130, -3.0), (2, False, 0, 129, 64, 0.0)])
def test_float16_join_tiles_keep_the_results(engine, me, lfat, mode, Dj, Dt, offset):
    """Streamed databases are scanned from a float16 copy of the join tiles (half the bytes; here forced on a small
    one, greedy_f16 2).  The wider bound lets more windows through to the exact decision; the paths and distances
    stay the oracle's."""
    N = 30000 + me
    engine.set_option('greedy_mode', 2); engine.set_option('greedy_hoist', 1); engine.set_option('greedy_f16', 2)
    try:
        F_unw, JC_unw, wt, wj = _setup(engine, N, Dt, Dj, seed=5 * me + Dt, me=me, lfat=lfat, mode=mode, offset=offset)
        lens = [33 * me + (me - 1), 16 * me, 17 * me, me]
        utts = [o.synthetic_targets(F_unw, T, seed=16 + i) * wt for i, T in enumerate(lens)]
        starts = [-1, 17, N - me - 3, 0]
        f0 = engine.info('greedy_fallbacks')
        ref0 = oc.greedy_f32(F_unw, JC_unw, wt, wj, utts[0], me, lfat, mode, starts[0])
        # scans of float16 tiles take the target values from the bf16 matrix pipe (three exact bf16 pieces per operand,
        # hoist_product16_kernel); greedy_hoist_fast 0 keeps the float64 product
        for fast in (0, 1):
            engine.set_option('greedy_hoist_fast', fast)
            n16 = engine.info('greedy_hoist16_launches')
            for U, st in zip(utts[:3], starts[:3]):
                path, d = engine.greedy(U, start_state=st, return_distances=True)
                op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, lfat, mode, st)
                assert path == op and np.array_equal(d, od)
            assert engine.info('greedy_hoist16_launches') == n16 + 3 * fast
        assert engine.info('greedy_f16_delta') > 0.0
        # the decision before the gather (greedy_speculate, default on) against the plain hand-off: the same results either way
        for spec in (0, 1):
            engine.set_option('greedy_speculate', spec)
            path, d = engine.greedy(utts[0], start_state=starts[0], return_distances=True)
            assert path == ref0[0] and np.array_equal(d, ref0[1])
            assert spec == 1 or engine.info('greedy_last_speculated') == 0
        paths, dists = engine.greedy_batch(utts[:3], start_states=starts[:3], return_distances=True)      # three per scan
        for U, st, p, d in zip(utts, starts, paths, dists):
            op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, lfat, mode, st)
            assert p == op and np.array_equal(d, od)
        # other join weights: the bound's norm follows them
        wj2 = wj[::-1].copy() * 3.0
        engine.set_weights(wt, wj2)
        d0 = engine.info('greedy_f16_delta')
        path, d = engine.greedy(utts[1], return_distances=True)
        op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj2, utts[1], me, lfat, mode, -1)
        assert path == op and np.array_equal(d, od) and engine.info('greedy_f16_delta') != d0
        assert engine.info('greedy_fallbacks') == f0
        # duplicated speech: exact ties through the float16 scan
        if me == 6:
            F2, J2 = F_unw.copy(), JC_unw.copy()
            F2[20000:20300] = F2[3000:3300]; J2[20000:20301] = J2[3000:3301]
            engine.upload_db(F2, J2); engine.set_weights(wt, wj); engine.set_greedy_layout(me, lfat, mode)
            U = F2[3100:3100 + 15 * me].astype(np.float64) * wt
            path, d = engine.greedy(U, start_state=3100, return_distances=True)
            assert path == list(range(3100, 3100 + 15 * me, me)) and np.all(d == 0.0)
    finally:
        engine.set_option('greedy_f16', 1); engine.set_option('greedy_hoist_fast', 1); engine.set_option('greedy_speculate', 1)


def test_values_outside_the_float16_range_keep_float32_tiles(engine):
    """A join value beyond 65 504 cannot be a float16: the copy is refused and the scan reads the float32 tiles."""
    N, Dt, Dj, me = 30000, 61, 151, 6
    engine.set_option('greedy_mode', 2); engine.set_option('greedy_hoist', 1); engine.set_option('greedy_f16', 2)
    try:
        F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, 77)
        JC_unw[12345, 7] = 1.0e5
        rng = np.random.RandomState(177)
        wt, wj = 0.2 + rng.rand(Dt), 0.05 + 0.2 * rng.rand(Dj)
        engine.upload_db(F_unw, JC_unw); engine.set_weights(wt, wj); engine.set_greedy_layout(me, False, 0)
        U = o.synthetic_targets(F_unw, 12 * me, seed=5) * wt
        n0 = engine.info('greedy_f16_launches')
        path, d = engine.greedy(U, return_distances=True)
        op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, False, 0, -1)
        assert path == op and np.array_equal(d, od)
        assert engine.info('greedy_f16_launches') == n0
    finally:
        engine.set_option('greedy_f16', 1)


@pytest.mark.parametrize('N,me,lfat,mode,Dj,Dt', [(65536 + 5, 6, False, 0, 151, 61),      # B1: every compute unit holds 256 windows
                                                 (30011, 4, False, 1, 302, 61),            # half-column join (synth_halfphone epoch voices)
                                                 (1078, 6, True, 0, 151, 61),              # the golden mini voice's size: five workgroups
                                                 (777, 1, False, 0, 70, 61)])
def test_resident_scan_equals_oracle_and_streamed_scan(engine, N, me, lfat, mode, Dj, Dt):
    """A database whose windowed join matrix fits the chip's LDS is searched by the resident scan (greedy_res_kernels.hip:
    one utterance per call, every workgroup decides each step for itself from the gathered records).  Same paths and
    distances as the oracle and as the streamed one-launch scan; the counter says which scan ran."""
    engine.set_option('greedy_mode', 2); engine.set_option('greedy_hoist', 1); engine.set_option('greedy_resident', 1)
    F_unw, JC_unw, wt, wj = _setup(engine, N, Dt, Dj, seed=5 * me + 1, me=me, lfat=lfat, mode=mode)
    # a stretch of speech that occurs twice: exact ties between two workgroups (and inside one: the copy starts 40 windows on)
    F_unw[300:300 + 30] = F_unw[100:100 + 30]; JC_unw[300:300 + 31] = JC_unw[100:100 + 31]
    F_unw[140:140 + 20] = F_unw[100:100 + 20]; JC_unw[140:140 + 21] = JC_unw[100:100 + 21]
    engine.upload_db(F_unw, JC_unw); engine.set_weights(wt, wj); engine.set_greedy_layout(me, lfat, mode)
    utts = [(o.synthetic_targets(F_unw, 40 * me + (me - 1), seed=9) * wt, -1),
            (o.synthetic_targets(F_unw, 17 * me, seed=10) * wt, N - me - 3),
            (F_unw[100:100 + 3 * me].astype(np.float64) * wt, 100),            # noise-free targets inside the stretch: exact three-way ties
            (F_unw[100:100 + 3 * me].astype(np.float64) * wt, -1)]
    for U, st in utts:
        r0, f0 = engine.info('greedy_resident_launches'), engine.info('greedy_fallbacks')
        path, d = engine.greedy(U, start_state=st, return_distances=True)
        op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, lfat, mode, st)
        assert path == op and np.array_equal(d, od)
        assert engine.info('greedy_resident_launches') == r0 + 1
        engine.set_option('greedy_resident', 0)
        try:
            path2, d2 = engine.greedy(U, start_state=st, return_distances=True)
        finally:
            engine.set_option('greedy_resident', 1)
        assert path2 == op and np.array_equal(d2, od) and engine.info('greedy_resident_launches') == r0 + 1
    # search_epsilon: the float32 minimum is within (1 + eps) of the nearest window
    U, st = utts[0]
    path, d = engine.greedy(U, start_state=st, search_epsilon=0.05, return_distances=True)
    op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, lfat, mode, st)
    assert len(path) == len(op)
    # (the (1 + eps) contract itself is checked step by step in test_gpu_greedy32.py::test_search_epsilon_contract, which
    # runs through this scan as well: one utterance, 30 000 units)


@pytest.mark.parametrize('resident', [1, 0])
def test_one_launch_hand_off_equals_the_fenced_variant_over_1000_steps(engine, resident):
    """The hand-off between the steps of the one-launch scans is '8-byte agent-scope atomics on both sides, storing
    wavefronts drained before the arrival' -- valid on gfx950 for hipMalloc memory by measurement, not by the memory
    model.  Cross-check: 1 000 steps of a B1-sized voice, under uneven load (a second engine keeps
    the chip busy beside it), with and without agent-scope release / acquire fences around every hand-off, and against
    the C oracle."""
    import snickery_amd
    N, Dt, Dj, me = 65536, 61, 151, 1
    engine.set_option('greedy_mode', 2); engine.set_option('greedy_hoist', 1); engine.set_option('greedy_resident', resident)
    F_unw, JC_unw, wt, wj = _setup(engine, N, Dt, Dj, seed=11, me=me, lfat=False, mode=0)
    U = o.synthetic_targets(F_unw, 1000, seed=12) * wt
    op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, False, 0, -1)
    s0 = engine.info('greedy_stalls')
    try:
        path, d = engine.greedy(U, return_distances=True)
        engine.set_option('greedy_fenced', 1)
        pathf, df = engine.greedy(U, return_distances=True)
    finally:
        engine.set_option('greedy_fenced', 0); engine.set_option('greedy_resident', 1)
    assert path == op and np.array_equal(d, od)
    assert pathf == op and np.array_equal(df, od)
    assert engine.info('greedy_stalls') == s0
