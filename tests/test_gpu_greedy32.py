"""The float32 prefilter scan of greedy_joint_search (greedy32_kernels.hip; synth_simple.py:458-503): forced on
(greedy_mode 1) it must return the oracle's paths and distances bit for bit at search_epsilon = 0 -- whatever
the layout, however many candidates fall inside the float32 margin -- and satisfy the reference's (1 + eps)
contract at search_epsilon > 0 (config/slt_simplified_mini.cfg:92 ships 10.0)."""
import numpy as np
import pytest

import snk_oracle as o
import snk_oracle_c as oc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def engine():
    import snickery_amd
    e = snickery_amd.HipSearchEngine(0)
    e.set_option('greedy_mode', 1)
    yield e
    e.close()


def _setup(N, Dt, Dj, seed):
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed)
    rng = np.random.RandomState(seed + 100)
    return F_unw, JC_unw, 0.2 + rng.rand(Dt), 0.05 + 0.2 * rng.rand(Dj)


@pytest.mark.parametrize('me,lfat,mode,Dj,Dt', [(6, False, 0, 151, 61), (3, True, 0, 40, 61), (4, False, 1, 80, 61),
                                               (1, False, 1, 302, 61), (16, False, 0, 151, 61), (5, False, 0, 151, 200),
                                               (2, True, 0, 33, 7)])
def test_float32_scan_equals_oracle(engine, me, lfat, mode, Dj, Dt):
    N = 20000
    F_unw, JC_unw, wt, wj = _setup(N, Dt, Dj, seed=me)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    engine.set_greedy_layout(me, lfat, mode)
    utts = [o.synthetic_targets(F_unw, T, seed=6 + i) * wt for i, T in enumerate([63, 40, max(me - 1, 1), 57, 22])]
    starts = [-1, 17, -1, 0, 400]
    for U, st in zip(utts[:2], starts[:2]):
        path, d = engine.greedy(U, start_state=st, return_distances=True)
        op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, lfat, mode, st)
        assert path == op and np.array_equal(d, od)
    paths, dists = engine.greedy_batch(utts, start_states=starts, return_distances=True)       # three per scan, ragged
    for U, st, p, d in zip(utts, starts, paths, dists):
        op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, lfat, mode, st)
        assert p == op and np.array_equal(d, od)
    assert engine.info('greedy_fallbacks') == 0


def test_near_ties_second_phase_and_mass_ties_fallback(engine):
    """Clean targets from a stretch of speech that occurs three times: exact three-way ties at every step (the
    lowest index wins: decided by canonical float64 totals).  Then a block of identical frames (digital silence):
    more ties than any lane keeps -- the launch reports the step and the exact scan finishes the utterance."""
    N, Dt, Dj, me = 60000, 61, 151, 6
    F_unw, JC_unw, wt, wj = _setup(N, Dt, Dj, seed=31)
    for dst in (25000, 47011):
        F_unw[dst:dst + 400] = F_unw[3000:3400]
        JC_unw[dst:dst + 401] = JC_unw[3000:3401]
    F_unw[52000:57500] = F_unw[52000]                     # 5 500 tied windows: more than one second-phase round takes
    JC_unw[52000:57501] = JC_unw[52000]
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    engine.set_greedy_layout(me, False, 0)
    U = F_unw[3100:3100 + 20 * me].astype(np.float64) * wt
    before = engine.info('greedy_fallbacks')
    path, d = engine.greedy(U, start_state=3100, return_distances=True)
    assert path == list(range(3100, 3100 + 20 * me, me)) and np.all(d == 0.0)
    op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, False, 0, 3100)
    assert path == op and np.array_equal(d, od)
    assert engine.info('greedy_fallbacks') == before and engine.info('greedy_exact_windows') >= 3 * 20
    Us = F_unw[52100:52100 + 5 * me].astype(np.float64) * wt
    path, d = engine.greedy(Us, return_distances=True)
    op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, Us, me, False, 0, -1)
    assert path == op and np.array_equal(d, od)
    assert engine.info('greedy_fallbacks') == before + 1
    # near ties only (noise far below the float32 margin): many candidates, every lane asked (second phase)
    rng = np.random.RandomState(2)
    Un = (F_unw[3100:3100 + 10 * me].astype(np.float64) + 1e-7 * rng.randn(10 * me, Dt)) * wt
    path, d = engine.greedy(Un, return_distances=True)
    op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, Un, me, False, 0, -1)
    assert path == op and np.array_equal(d, od)


@pytest.mark.parametrize('eps', [10.0, 0.05])
def test_search_epsilon_contract(engine, eps):
    """search_epsilon > 0: every returned window lies within (1 + eps) of the exact nearest distance of its step
    (the joint_tree.query(..., eps) contract, synth_simple.py:488-490) -- checked against the oracle's full
    distance vector of every step, following the GPU's own path."""
    N, Dt, Dj, me = 30000, 61, 151, 6
    F_unw, JC_unw, wt, wj = _setup(N, Dt, Dj, seed=41)
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    engine.set_greedy_layout(me, False, 0)
    U = o.synthetic_targets(F_unw, 20 * me, seed=9) * wt
    path, d = engine.greedy(U, search_epsilon=eps, return_distances=True)
    exact_path, _ = engine.greedy(U, search_epsilon=0.0, return_distances=True)
    assert len(path) == 20
    prev = -1
    for s, (i, di) in enumerate(zip(path, d)):
        # distances of ALL windows at this step, given the path so far (the oracle restarted at the previous pick)
        if prev < 0:
            _, _, d2 = oc.greedy_f32(F_unw, JC_unw, wt, wj, U[s * me:(s + 1) * me], me, False, 0, -1, d2_step=0)
        else:
            # prev_join_rep of the next step = current_join_rep[prev] = unit_end_data[prev + me - 1] = unit_start_data[prev + me]
            _, _, d2 = oc.greedy_f32(F_unw, JC_unw, wt, wj, U[s * me:(s + 1) * me], me, False, 0, prev + me, d2_step=0)
        assert di == np.sqrt(d2[i])                                  # the reported distance is the exact one of the pick
        assert di <= (1.0 + eps) * np.sqrt(d2.min()) * (1 + 1e-12)
        prev = i
    assert path == exact_path or eps > 0                             # (the float32 minimum usually IS the exact one)
