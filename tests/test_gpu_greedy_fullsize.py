"""Greedy search (greedy_joint_search, synth_simple.py:458-503) at the sizes it is quoted on: B1
(N = 65 536: the README demo voice, BASELINE configs[0]) and B3 (N = 1.5 M: IS2018_nick_simplified.cfg,
BASELINE configs[2]), multiepoch 6 and 5, magphase-60 widths.  snk_greedy and snk_greedy_batch against
the C oracle (snko_greedy_f32), paths and distances bit for bit.  At these sizes every wavefront of the
persistent grid walks many 64-window tiles and the arrival tree has all its levels -- which the small
cases of test_gpu_parity.py never reach."""
import numpy as np
import pytest

import snk_oracle as o
import snk_oracle_c as oc

pytestmark = pytest.mark.gpu

DT, DJ = 61, 151


@pytest.fixture(scope='module')
def engine():
    import snickery_amd
    e = snickery_amd.HipSearchEngine(0)
    yield e
    e.close()


def _voice(N, seed, dup=None):
    F_unw, JC_unw = o.synthetic_db(N, DT, DJ, seed=seed)
    if dup is not None:
        # a stretch of speech that occurs twice (a repeated phrase / digital silence): every window inside
        # the stretch ties EXACTLY with its copy -- in another tile, wavefront and workgroup -- and the
        # lower index must win
        src, dst, n = dup
        F_unw[dst:dst + n] = F_unw[src:src + n]
        JC_unw[dst:dst + n + 1] = JC_unw[src:src + n + 1]
    rng = np.random.RandomState(seed + 100)
    wt = 0.2 + rng.rand(DT)
    wj = 0.02 + 0.1 * rng.rand(DJ)
    return F_unw, JC_unw, wt, wj


def _check(engine, F_unw, JC_unw, wt, wj, U, me, start=-1, max_steps=None, lfat=False):
    steps = U.shape[0] // me if max_steps is None else max_steps
    Ucut = U[:steps * me]
    path, d = engine.greedy(Ucut, start_state=start, return_distances=True)
    op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, Ucut, me, lfat, 0, start)
    assert path == op, [(i, a, b) for i, (a, b) in enumerate(zip(path, op)) if a != b][:5]
    assert np.array_equal(d, od)
    return path


def test_b1_full_utterance(engine):
    """B1: N = 65 536, me = 6, T = 600: the whole path (100 steps), with and without a start state."""
    N = 65536
    F_unw, JC_unw, wt, wj = _voice(N, 70, dup=(12000, 40000, 700))
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    engine.set_greedy_layout(6, False, 0)
    U = o.synthetic_targets(F_unw, 600, seed=71) * wt
    _check(engine, F_unw, JC_unw, wt, wj, U, 6)
    _check(engine, F_unw, JC_unw, wt, wj, U, 6, start=31000)
    # targets taken from inside the duplicated stretch, noise-free: exact ties between the stretch and its copy
    Ud = (F_unw[12100:12100 + 120].astype(np.float64)) * wt
    p = _check(engine, F_unw, JC_unw, wt, wj, Ud, 6, start=12100)
    assert p == list(range(12100, 12100 + 120, 6))            # the lower of the two tied windows, every step
    _check(engine, F_unw, JC_unw, wt, wj, Ud, 6)              # free start: whatever wins inside the stretch ties with its copy
    # batch entry point: two utterances per scan, ragged, one with a start state
    utts = [U, Ud, U[:301], U[:5]]
    starts = [-1, 12100, 31000, -1]
    paths, dists = engine.greedy_batch(utts, start_states=starts, return_distances=True)
    for u, st, p, d in zip(utts, starts, paths, dists):
        op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, u, 6, False, 0, st)
        assert p == op and np.array_equal(d, od)
    assert engine.info('greedy_bound_violations') == 0, engine.info('greedy_bound_max_used')      # (resident + streamed scans of this test)
    assert engine.info('greedy_bound_max_used') <= 1.0


@pytest.mark.parametrize('me', [6, 5])
def test_b3_nick_size(engine, me):
    """B3: N = 1.5 M, multiepoch 6 (config file) and 5 (BASELINE.json), magphase-60 widths."""
    N = 1500000
    F_unw, JC_unw, wt, wj = _voice(N, 80 + me, dup=(200000, 1100000, 900))
    engine.upload_db(F_unw, JC_unw)
    engine.set_weights(wt, wj)
    engine.reset_timers()                         # (also the tripwire of the scan's bound)
    engine.set_greedy_layout(me, False, 0)
    U = o.synthetic_targets(F_unw, 600, seed=81) * wt
    n16, nh = engine.info('greedy_f16_launches'), engine.info('greedy_hoist_launches')
    _check(engine, F_unw, JC_unw, wt, wj, U, me, max_steps=24)
    # a database of this size is streamed: hoisted target term + float16 join tiles are what has just been checked
    assert engine.info('greedy_f16_launches') == n16 + 1 and engine.info('greedy_hoist_launches') == nh + 1
    # ... with the target values from the bf16 pipe, and steps decided before the gather by the workgroup whose minimum was the smallest
    # published so far (greedy32_kernels.hip, the step's tail): both paths were part of what was compared with the oracle
    assert engine.info('greedy_hoist16_launches') >= 1 and engine.info('greedy_last_speculated') >= 1
    _check(engine, F_unw, JC_unw, wt, wj, U, me, start=1234567, max_steps=20)
    n = 20 * me
    Ud = (F_unw[200300:200300 + n].astype(np.float64)) * wt
    p = _check(engine, F_unw, JC_unw, wt, wj, Ud, me, start=200300)
    assert p == list(range(200300, 200300 + n, me))
    utts = [U[:12 * me], Ud[:10 * me + 3], U[100:100 + 7 * me]]
    starts = [-1, 200300, 777777]
    paths, dists = engine.greedy_batch(utts, start_states=starts, return_distances=True)
    for u, st, p, d in zip(utts, starts, paths, dists):
        op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, u, me, False, 0, st)
        assert p == op and np.array_equal(d, od)
    # tripwire of the scan's bound (float16 tiles + target values from the bf16 pipe: the probed accumulation property): every exact
    # total weighed in these steps stayed within the bound of the float32 minimum, and windows WERE weighed
    assert engine.info('greedy_bound_violations') == 0, engine.info('greedy_bound_max_used')
    assert 0.0 < engine.info('greedy_bound_max_used') <= 1.0
