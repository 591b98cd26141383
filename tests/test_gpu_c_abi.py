"""The C ABI from C: tests/c_abi/smoke.c (gcc, include/snk.h, libsnkhip.so; no Python, no torch in the
process) against the same calls through the ctypes binding and the oracle."""
import os
import subprocess
import sys
import numpy as np
import pytest

import snk_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _lcg_stream():
    state = 12345
    while True:
        state = (state * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        yield float((state >> 11) & ((1 << 53) - 1)) / float(1 << 52) - 1.0


@pytest.fixture()
def engine():
    import snickery_amd
    e = snickery_amd.HipSearchEngine(0)
    yield e
    e.close()


def test_c_program_matches_binding_and_oracle(tmp_path, engine):
    exe = os.path.join(str(tmp_path), 'smoke')
    lib = os.path.join(ROOT, 'snickery_amd')
    subprocess.check_call(['gcc', '-O1', '-std=c99', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'tests', 'c_abi', 'smoke.c'), '-o', exe,
                           '-L', lib, '-l:libsnkhip.so', '-Wl,-rpath,' + lib])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = dict((l.split(' ', 1)[0], l.split(' ')[1:]) for l in out.stdout.strip().splitlines())

    # the same data in numpy (same generator, same order of draws)
    N, Dt, Dj, K, T, me = 5000, 61, 40, 12, 30, 3
    g = _lcg_stream()
    F = np.empty((N, Dt), dtype=np.float32)
    JC = np.empty((N + 1, Dj), dtype=np.float32)
    for c in range(Dt):
        F[:, c] = np.cumsum([0.1 * next(g) for _ in range(N)]).astype(np.float32)
    for c in range(Dj):
        JC[:, c] = np.cumsum([0.1 * next(g) for _ in range(N + 1)]).astype(np.float32)
    wt, wj = np.full(Dt, 0.4), np.full(Dj, 0.05)
    Q = np.empty((T, Dt))
    for t in range(T):
        for c in range(Dt):
            Q[t, c] = (float(F[1000 + t, c]) + 0.05 * next(g)) * wt[c]
    Fw, E, S = o.weighted_db(F, JC, wt, wj)
    oc, od = o.knn_bruteforce(Fw, Q, K)
    assert [int(v) for v in lines['knn_row0']] == list(oc[0])
    assert float(lines['knn_dist0'][0]) == od[0, 0] and float(lines['knn_dist0'][1]) == od[0, K - 1]
    op, ocost = o.viterbi(oc, od, E, S)
    assert int(lines['viterbi'][0]) == len(op) and float(lines['viterbi'][1]) == ocost
    assert [int(v) for v in lines['viterbi'][2:]] == op
    b = lines['batch']
    p0, c0 = o.viterbi(*o.knn_bruteforce(Fw, Q[:18], K), E, S)
    p1, c1 = o.viterbi(*o.knn_bruteforce(Fw, Q[18:], K), E, S)
    assert [int(b[0]), int(b[1])] == [18, 12] and float(b[2]) == c0 and float(b[3]) == c1
    assert [int(v) for v in b[4:]] == p0 + p1
    pr, cr, Fwin = o.greedy_layout(Fw, E, S, me)
    og, ogd = o.greedy_search(pr, cr, Fwin, o.greedy_queries(Q, me))
    gl = lines['greedy']
    assert int(gl[0]) == len(og) and [int(v) for v in gl[1:-1]] == og and float(gl[-1]) == ogd[0]
    # and through the binding
    engine.upload_db(F, JC)
    engine.set_weights(wt, wj)
    cand, dist = engine.knn(Q, K)
    assert np.array_equal(cand, oc) and np.array_equal(dist, od)
