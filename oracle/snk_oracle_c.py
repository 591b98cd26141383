"""ctypes wrapper of the C oracle (oracle/snk_oracle.c).  Test infrastructure only."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libsnkoracle.so')
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
