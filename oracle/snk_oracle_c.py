"""ctypes wrapper of the C oracle (oracle/snk_oracle.c).  Test infrastructure only."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SNK_ORACLE_LIB: another build of the same source (oracle/Makefile `asan`: AddressSanitizer + UBSan, run under LD_PRELOAD=libasan)
_SO = os.environ.get('SNK_ORACLE_LIB') or os.path.join(_HERE, '_build', 'libsnkoracle.so')
_lib = None
_i64p = ctypes.POINTER(ctypes.c_int64)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int)


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            subprocess.check_call(['make', '-C', _HERE])
        _lib = ctypes.CDLL(_SO)
        _lib.snko_viterbi.restype = ctypes.c_int64
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def knn(F, U, K):
    F = np.ascontiguousarray(F, np.float64); U = np.ascontiguousarray(U, np.float64)
    T = U.shape[0]
    cand = np.empty((T, K), np.int64); dist = np.empty((T, K), np.float64)
    lib().snko_knn(_p(F, _f64p), ctypes.c_int64(F.shape[0]), F.shape[1], _p(U, _f64p), ctypes.c_int64(T), K,
                   _p(cand, _i64p), _p(dist, _f64p))
    return cand, dist


def join_dense(JCw, cand):
    JCw = np.ascontiguousarray(JCw, np.float64); cand = np.ascontiguousarray(cand, np.int64)
    T, K = cand.shape
    J = np.empty((T - 1, K, K), np.float64)
    lib().snko_join(_p(JCw, _f64p), ctypes.c_int64(JCw.shape[0] - 1), JCw.shape[1], _p(cand, _i64p),
                    ctypes.c_int64(T), K, _p(J, _f64p))
    return J


def viterbi(cand, tdist, JCw):
    cand = np.ascontiguousarray(cand, np.int64); tdist = np.ascontiguousarray(tdist, np.float64)
    T, K = cand.shape
    J = join_dense(JCw, cand) if T > 1 else np.zeros((0, K, K))
    path = np.empty(T, np.int64); cost = ctypes.c_double()
    n = lib().snko_viterbi(_p(cand, _i64p), _p(tdist, _f64p), _p(J, _f64p), ctypes.c_int64(T), K,
                           ctypes.c_int64(JCw.shape[0] - 1), _p(path, _i64p), ctypes.byref(cost))
    return [int(v) for v in path[:n]], float(cost.value)


def greedy_f32(F_unw, JC_unw, wt, wj, unit_features, multiepoch=1, last_frame_as_target=False, join_split_mode=0,
               start_state=-1, max_steps=None, d2_step=-1):
    """greedy_joint_search straight from the unweighted float32 database (snko_greedy_f32): same
    results as snk_oracle.greedy_search on the weighted float64 copies, without materialising them.
    unit_features: (T, Dt) weighted targets (not yet reshaped).  Returns (path, dists[, d2 of step d2_step])."""
    F_unw = np.ascontiguousarray(F_unw, np.float32); JC_unw = np.ascontiguousarray(JC_unw, np.float32)
    wt = np.ascontiguousarray(wt, np.float64); wj = np.ascontiguousarray(wj, np.float64)
    N, Dt = F_unw.shape
    assert JC_unw.shape[0] == N + 1 and wt.shape == (Dt,) and wj.shape == (JC_unw.shape[1],)
    me = int(multiepoch)
    ep = np.array([0, me - 1] if (last_frame_as_target and me > 1) else list(range(me)), np.int32)
    U = np.asarray(unit_features, np.float64)
    steps = U.shape[0] // me
    if max_steps is not None:
        steps = min(steps, int(max_steps))
    Q = np.ascontiguousarray(U[:steps * me].reshape(steps, me, Dt)[:, ep, :].reshape(steps * len(ep), Dt))
    path = np.empty(steps, np.int64); dists = np.empty(steps, np.float64)
    d2 = np.empty(N - me + 1, np.float64) if d2_step >= 0 else None
    lib().snko_greedy_f32(_p(F_unw, ctypes.POINTER(ctypes.c_float)), ctypes.c_int64(N), Dt, _p(wt, _f64p),
                          _p(JC_unw, ctypes.POINTER(ctypes.c_float)), JC_unw.shape[1], _p(wj, _f64p), me,
                          _p(ep, _i32p), len(ep), int(join_split_mode), _p(Q, _f64p), ctypes.c_int64(steps),
                          ctypes.c_int64(start_state), _p(path, _i64p), _p(dists, _f64p), ctypes.c_int64(d2_step),
                          _p(d2, _f64p) if d2 is not None else None)
    if d2 is not None:
        return [int(v) for v in path], dists, d2
    return [int(v) for v in path], dists
