"""ctypes binding of libsnkhip.so (include/snk.h).

``HipSearchEngine`` mirrors the search-related methods of the reference's ``Synthesiser``
(script/synth_simple.py, script/synth_halfphone.py) one level below the config/HDF5 front end
in ``snickery_amd.synthesiser``: same argument meaning, numpy in / numpy out.
"""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class SnkError(RuntimeError):
    pass


def library_path():
    # SNK_LIBRARY: another BUILD of the same library (the sanitizer build of `make asan-host`, tests/test_host_asan.py);
    # there is no other implementation to point it at
    return os.environ.get('SNK_LIBRARY') or os.path.join(_HERE, 'libsnkhip.so')


_c_i64p = ctypes.POINTER(ctypes.c_int64)
_c_i32p = ctypes.POINTER(ctypes.c_int32)
_c_f64p = ctypes.POINTER(ctypes.c_double)
_c_f32p = ctypes.POINTER(ctypes.c_float)

# every symbol include/snk.h declares: (restype, argtypes)
_SIGNATURES = {
    'snk_abi_version': (ctypes.c_int, []),
    'snk_last_error': (ctypes.c_char_p, []),
    'snk_device_count': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    'snk_create': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    'snk_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'snk_upload_db': (ctypes.c_int, [ctypes.c_void_p, _c_f32p, ctypes.c_int64, ctypes.c_int,
                                     _c_f32p, ctypes.c_int64, ctypes.c_int]),
    'snk_set_weights': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, ctypes.c_int, _c_f64p, ctypes.c_int]),
    'snk_knn': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                               _c_i64p, _c_f64p]),
    'snk_set_unit_classes': (ctypes.c_int, [ctypes.c_void_p, _c_i32p, ctypes.c_int64]),
    'snk_knn_by_class': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, ctypes.c_int64, ctypes.c_int,
                                        ctypes.c_int, _c_i32p, _c_i64p, _c_f64p]),
    'snk_prefilter_minima': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, ctypes.c_int64, ctypes.c_int, _c_f32p, ctypes.c_int64,
                                            _c_f64p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]),
    'snk_candidate_distances': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, ctypes.c_int64, ctypes.c_int, _c_i64p,
                                               ctypes.c_int, _c_f64p]),
    'snk_join_costs': (ctypes.c_int, [ctypes.c_void_p, _c_i64p, ctypes.c_int64, ctypes.c_int, _c_f64p]),
    'snk_join_bounds': (ctypes.c_int, [ctypes.c_void_p, _c_i64p, ctypes.c_int64, ctypes.c_int, _c_f32p, _c_f32p]),
    'snk_viterbi': (ctypes.c_int, [ctypes.c_void_p, _c_i64p, _c_f64p, ctypes.c_int64, ctypes.c_int,
                                   _c_i64p, _c_i64p, _c_f64p]),
    'snk_viterbi_batch': (ctypes.c_int, [ctypes.c_void_p, _c_i64p, _c_f64p, _c_i64p, ctypes.c_int, ctypes.c_int,
                                         _c_i64p, _c_i64p, _c_f64p]),
    'snk_knn_viterbi': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, ctypes.c_int64, ctypes.c_int,
                                       ctypes.c_int, _c_i64p, _c_f64p, _c_i64p, _c_i64p, _c_f64p]),
    'snk_knn_viterbi_batch': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, _c_i64p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, _c_i64p, _c_i64p, _c_f64p]),
    'snk_greedy_batch': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, _c_i64p, ctypes.c_int, ctypes.c_int, _c_i64p,
                                        ctypes.c_double, _c_i64p, _c_f64p, _c_i64p]),
    'snk_set_column_selection': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                                ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    'snk_host_register': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    'snk_host_unregister': (ctypes.c_int, [ctypes.c_void_p]),
    'snk_knn_viterbi_batch_submit': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, _c_i64p, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    'snk_knn_viterbi_batch_collect': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_i64p, _c_i64p, _c_f64p]),
    'snk_set_greedy_layout': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'snk_greedy': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                  ctypes.c_double, _c_i64p, _c_f64p, _c_i64p]),
    'snk_sharded_greedy': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                          _c_i64p, _c_f64p, ctypes.POINTER(ctypes.c_int64)]),
    'snk_path_scores': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, _c_i64p, ctypes.c_int64, ctypes.c_int,
                                       _c_f64p, _c_f64p]),
    'snk_get_timers': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, ctypes.c_int]),
    'snk_timer_name': (ctypes.c_char_p, [ctypes.c_int]),
    'snk_timer_count': (ctypes.c_int, []),
    'snk_reset_timers': (ctypes.c_int, [ctypes.c_void_p]),
    'snk_set_shard': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]),
    'snk_knn_local_dev': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, ctypes.c_int64, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    'snk_merge_topk_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_int64, ctypes.c_int, _c_i64p, _c_f64p]),
    'snk_knn_local_batch_dev': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, _c_i64p, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    'snk_knn_local_batch_bounds_dev': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, _c_i64p, ctypes.c_int, ctypes.c_int,
                                                      ctypes.c_int, ctypes.c_void_p]),
    'snk_knn_local_batch_bounded_dev': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, _c_i64p, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'snk_merge_viterbi_batch_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                   _c_i64p, ctypes.c_int, ctypes.c_int, _c_i64p, _c_i64p, _c_f64p]),
    'snk_upload_join_only': (ctypes.c_int, [ctypes.c_void_p, _c_f32p, ctypes.c_int64, ctypes.c_int]),
    'snk_upload_frames': (ctypes.c_int, [ctypes.c_void_p, _c_f32p, _c_f64p, ctypes.c_int64, ctypes.c_int]),
    'snk_concat_fragments': (ctypes.c_int, [ctypes.c_void_p, _c_i64p, _c_i64p, _c_i64p, ctypes.c_int64, ctypes.c_int,
                                            ctypes.c_int, _c_f64p, _c_f64p, _c_f64p]),
    'snk_comm_unique_id': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    'snk_comm_init': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'snk_comm_init_transport': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'snk_comm_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'snk_shard_plan': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_i64p, _c_i64p]),
    'snk_upload_global_sample': (ctypes.c_int, [ctypes.c_void_p, _c_f32p, ctypes.c_int64, ctypes.c_int]),
    'snk_sharded_knn_viterbi_batch': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, _c_i64p, ctypes.c_int, ctypes.c_int,
                                                     ctypes.c_int, _c_i64p, _c_i64p, _c_f64p]),
    'snk_sharded_knn_viterbi_batch_submit': (ctypes.c_int, [ctypes.c_void_p, _c_f64p, _c_i64p, ctypes.c_int, ctypes.c_int,
                                                            ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    'snk_sharded_knn_viterbi_batch_collect': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_i64p, _c_i64p, _c_f64p]),
    'snk_copy_to_host': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    'snk_copy_to_device': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    'snk_set_option': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double]),
    'snk_get_info': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, _c_f64p]),
    'snk_probe_mfma_bf16': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint16),
                                           _c_f32p, _c_f32p]),
    'snk_selftest_mfma': (ctypes.c_int, [ctypes.c_void_p, _c_f64p]),
}


def device_count():
    """GPUs visible to the process (snk_device_count)."""
    n = ctypes.c_int(0)
    if load_library().snk_device_count(ctypes.byref(n)) != 0:
        raise SnkError(load_library().snk_last_error().decode())
    return int(n.value)


def configure_runtime(hw_queues=8):
    """OPT-IN process-wide tuning of the ROCm runtime, for callers that own the process (bench.py, a tuning job): an engine works on
    four streams (K-NN, two Viterbi sides, copies) and the runtime maps a process's streams onto four hardware queues by default --
    beside a framework's own streams two of them then share one (measured + 1..3 % on the B* step with eight).  Sets
    GPU_MAX_HW_QUEUES unless the caller's environment already chose a number.  The runtime reads it when it STARTS: call this before
    anything touches the GPU (before `import torch` initialises it, before the first engine).  Importing the package never does this
    by itself.  Returns the value now in the environment."""
    os.environ.setdefault('GPU_MAX_HW_QUEUES', str(int(hw_queues)))
    return os.environ['GPU_MAX_HW_QUEUES']


def load_library():
    """Load libsnkhip.so and bind every symbol of include/snk.h.  Raises SnkError when the
    library has not been built (``python -c 'import __graft_entry__ as g; g.build()'`` or
    ``make``): the product never falls back to a CPU path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.isfile(path):
        raise SnkError('%s not found: build it with `make` (hipcc --offload-arch=gfx950); '
                       'there is no CPU fallback' % path)
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _LIB = lib
    return lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


class QueryBatch(object):
    """The query rows of several utterances as the C ABI takes them: one contiguous float64 matrix
    (rows of all utterances in order) + row offsets.  Callers that search the same tune set again and
    again (balance_stream_weights.py) build it once; every batch entry point accepts it in place of
    the list of per-utterance matrices and then skips the concatenation (75 MB of host copying for
    256 utterances of 600 frames)."""

    def __init__(self, utterances):
        mats = [_f64(u) for u in utterances]
        self.lengths = [int(m.shape[0]) for m in mats]
        self.offsets = np.zeros(len(mats) + 1, dtype=np.int64)
        self.offsets[1:] = np.cumsum(self.lengths)
        self.Q = np.ascontiguousarray(np.vstack(mats))

    def __len__(self):
        return len(self.lengths)

    def pin(self):
        """Page-lock the rows (hipHostRegister): uploads of this batch are then queued on the stream
        instead of making the host wait for it.  Undone when the batch is garbage collected."""
        if not getattr(self, '_pinned', False):
            lib = load_library()
            if lib.snk_host_register(ctypes.c_void_p(self.Q.ctypes.data), self.Q.nbytes) != 0:
                raise SnkError(lib.snk_last_error().decode('utf-8', 'replace'))
            self._pinned = True
        return self

    def __del__(self):
        if getattr(self, '_pinned', False):
            try:
                load_library().snk_host_unregister(ctypes.c_void_p(self.Q.ctypes.data))
            except Exception:
                pass

    def subset(self, lo, hi):
        """Utterances lo..hi-1 as a batch of their own (a view of the same rows)."""
        b = QueryBatch.__new__(QueryBatch)
        b.lengths = self.lengths[lo:hi]
        b.offsets = self.offsets[lo:hi + 1] - self.offsets[lo]
        b.Q = self.Q[self.offsets[lo]:self.offsets[hi]]
        return b


def _as_batch(utterances):
    return utterances if isinstance(utterances, QueryBatch) else QueryBatch(utterances)


class HipSearchEngine(object):
    """One engine = one MI355X + one HIP stream (see include/snk.h)."""

    def __init__(self, device=0):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        self._check(self._lib.snk_create(int(device), ctypes.byref(self._h)))
        self.device = int(device)
        self.n_units = 0
        self.Dt = 0
        self.Dj = 0

    # -- plumbing -----------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise SnkError(self._lib.snk_last_error().decode('utf-8', 'replace'))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.snk_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- database -----------------------------------------------------------
    def upload_db(self, train_unit_features_unweighted, join_contexts_unweighted):
        F = np.ascontiguousarray(train_unit_features_unweighted, dtype=np.float32)
        JC = np.ascontiguousarray(join_contexts_unweighted, dtype=np.float32)
        assert F.ndim == 2 and JC.ndim == 2
        self._check(self._lib.snk_upload_db(self._h, _ptr(F, _c_f32p), F.shape[0], F.shape[1],
                                            _ptr(JC, _c_f32p), JC.shape[0], JC.shape[1]))
        self.n_units, self.Dt = F.shape
        self.Dj = JC.shape[1]

    def upload_target_only(self, train_unit_features_unweighted):
        """Shard of the target features without a join matrix (multi-GPU K-NN ranks)."""
        F = np.ascontiguousarray(train_unit_features_unweighted, dtype=np.float32)
        self._check(self._lib.snk_upload_db(self._h, _ptr(F, _c_f32p), F.shape[0], F.shape[1], None, 0, 0))
        self.n_units, self.Dt = F.shape

    def upload_join_only(self, join_contexts_unweighted):
        JC = np.ascontiguousarray(join_contexts_unweighted, dtype=np.float32)
        self._check(self._lib.snk_upload_join_only(self._h, _ptr(JC, _c_f32p), JC.shape[0], JC.shape[1]))
        self.Dj = JC.shape[1]

    def set_weights(self, target_weight_vector, join_weight_vector):
        """Per-COLUMN weight vectors (what set_target_weights / set_join_weights build from
        the per-stream weights, synth_simple.py:234-274)."""
        wt = _f64(target_weight_vector if target_weight_vector is not None else [])
        wj = _f64(join_weight_vector if join_weight_vector is not None else [])
        self._check(self._lib.snk_set_weights(self._h, _ptr(wt, _c_f64p), wt.size, _ptr(wj, _c_f64p), wj.size))

    def set_unit_classes(self, unit_class):
        uc = np.ascontiguousarray(unit_class, dtype=np.int32)
        self._check(self._lib.snk_set_unit_classes(self._h, _ptr(uc, _c_i32p), uc.size))

    def set_shard(self, global_row_offset, global_n):
        self._check(self._lib.snk_set_shard(self._h, int(global_row_offset), int(global_n)))

    # -- search ---------------------------------------------------------------
    def knn(self, unit_features, n_candidates):
        """preselect_units_acoustic (synth_halfphone.py:1359-1366): (candidates, distances)."""
        Q = _f64(unit_features)
        T, D = Q.shape
        K = int(n_candidates)
        cand = np.empty((T, K), dtype=np.int64)
        dist = np.empty((T, K), dtype=np.float64)
        self._check(self._lib.snk_knn(self._h, _ptr(Q, _c_f64p), T, D, K, _ptr(cand, _c_i64p), _ptr(dist, _c_f64p)))
        return cand, dist

    def knn_by_class(self, unit_features, n_candidates, query_class):
        """preselect_units_monophone_then_acoustic (synth_halfphone.py:1369-1396)."""
        Q = _f64(unit_features)
        T, D = Q.shape
        K = int(n_candidates)
        qc = np.ascontiguousarray(query_class, dtype=np.int32)
        assert qc.size == T
        cand = np.empty((T, K), dtype=np.int64)
        dist = np.empty((T, K), dtype=np.float64)
        self._check(self._lib.snk_knn_by_class(self._h, _ptr(Q, _c_f64p), T, D, K, _ptr(qc, _c_i32p),
                                               _ptr(cand, _c_i64p), _ptr(dist, _c_f64p)))
        return cand, dist

    def prefilter_minima(self, unit_features):
        """Diagnostic (include/snk.h snk_prefilter_minima): (slab_min (T, n_slabs) f32, eps (T), rows_per_slab)."""
        Q = _f64(unit_features)
        T, D = Q.shape
        n_slabs, rows = ctypes.c_int64(0), ctypes.c_int(0)
        self._check(self._lib.snk_prefilter_minima(self._h, _ptr(Q, _c_f64p), T, D, None, 0, None, ctypes.byref(n_slabs),
                                                   ctypes.byref(rows)))
        out = np.empty((T, n_slabs.value), dtype=np.float32)
        eps = np.empty(T, dtype=np.float64)
        self._check(self._lib.snk_prefilter_minima(self._h, _ptr(Q, _c_f64p), T, D, _ptr(out, _c_f32p), out.size,
                                                   _ptr(eps, _c_f64p), ctypes.byref(n_slabs), ctypes.byref(rows)))
        return out, eps, rows.value

    def candidate_distances(self, unit_features, candidates):
        """Distance part of preselect_units_quinphone (synth_halfphone.py:1343-1349)."""
        Q = _f64(unit_features)
        cand = np.ascontiguousarray(candidates, dtype=np.int64)
        T, D = Q.shape
        assert cand.shape[0] == T
        dist = np.empty(cand.shape, dtype=np.float64)
        self._check(self._lib.snk_candidate_distances(self._h, _ptr(Q, _c_f64p), T, D, _ptr(cand, _c_i64p),
                                                      cand.shape[1], _ptr(dist, _c_f64p)))
        return dist

    def join_costs(self, candidates):
        cand = np.ascontiguousarray(candidates, dtype=np.int64)
        T, K = cand.shape
        J = np.empty((T - 1, K, K), dtype=np.float64)
        self._check(self._lib.snk_join_costs(self._h, _ptr(cand, _c_i64p), T, K, _ptr(J, _c_f64p)))
        return J

    def join_bounds(self, candidates):
        """Diagnostic: pass 1 of the sparse Viterbi path, the float32 lower bounds of join_costs(candidates) and the
        per-step scale (snk_join_bounds)."""
        cand = np.ascontiguousarray(candidates, dtype=np.int64)
        T, K = cand.shape
        lo = np.empty((T - 1, K, K), dtype=np.float32)
        scale = np.empty((T - 1,), dtype=np.float32)
        self._check(self._lib.snk_join_bounds(self._h, _ptr(cand, _c_i64p), T, K, _ptr(lo, _c_f32p), _ptr(scale, _c_f32p)))
        return lo, scale

    def viterbi(self, candidates, distances):
        """viterbi_search (synth_halfphone.py:1399-1436): (path list[int], cost)."""
        cand = np.ascontiguousarray(candidates, dtype=np.int64)
        tdist = _f64(distances)
        T, K = cand.shape
        assert tdist.shape == (T, K)
        path = np.empty((T,), dtype=np.int64)
        plen = ctypes.c_int64(0)
        cost = ctypes.c_double(0.0)
        self._check(self._lib.snk_viterbi(self._h, _ptr(cand, _c_i64p), _ptr(tdist, _c_f64p), T, K,
                                          _ptr(path, _c_i64p), ctypes.byref(plen), ctypes.byref(cost)))
        return [int(v) for v in path[:plen.value]], float(cost.value)

    def viterbi_batch(self, candidates, distances):
        """viterbi_search for a list of utterances with given candidates ((T_u, K) int64 / float64 pairs) in one call:
        (list of paths, costs)."""
        cands = [np.ascontiguousarray(c, dtype=np.int64) for c in candidates]
        dists = [_f64(d) for d in distances]
        assert len(cands) == len(dists) and len(cands) > 0
        K = cands[0].shape[1]
        assert all(c.shape == d.shape and c.shape[1] == K for c, d in zip(cands, dists))
        offs = np.zeros(len(cands) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([c.shape[0] for c in cands])
        call = np.ascontiguousarray(np.vstack(cands))
        dall = np.ascontiguousarray(np.vstack(dists))
        total = int(offs[-1])
        path = np.empty((max(total, 1),), dtype=np.int64)
        plen = np.zeros(len(cands), dtype=np.int64)
        cost = np.zeros(len(cands), dtype=np.float64)
        self._check(self._lib.snk_viterbi_batch(self._h, _ptr(call, _c_i64p), _ptr(dall, _c_f64p), _ptr(offs, _c_i64p), len(cands), K,
                                                _ptr(path, _c_i64p), _ptr(plen, _c_i64p), _ptr(cost, _c_f64p)))
        return [path[int(offs[u]):int(offs[u]) + int(plen[u])].copy() for u in range(len(cands))], cost

    def knn_viterbi(self, unit_features, n_candidates, return_candidates=False):
        Q = _f64(unit_features)
        T, D = Q.shape
        K = int(n_candidates)
        path = np.empty((T,), dtype=np.int64)
        plen = ctypes.c_int64(0)
        cost = ctypes.c_double(0.0)
        cand = dist = None
        cp = dp = None
        if return_candidates:
            cand = np.empty((T, K), dtype=np.int64)
            dist = np.empty((T, K), dtype=np.float64)
            cp, dp = _ptr(cand, _c_i64p), _ptr(dist, _c_f64p)
        self._check(self._lib.snk_knn_viterbi(self._h, _ptr(Q, _c_f64p), T, D, K, cp, dp,
                                              _ptr(path, _c_i64p), ctypes.byref(plen), ctypes.byref(cost)))
        out = ([int(v) for v in path[:plen.value]], float(cost.value))
        if return_candidates:
            return out + (cand, dist)
        return out

    def knn_viterbi_batch(self, utterances, n_candidates):
        """Several utterances in one call (stages of consecutive utterances overlap).
        Returns (list of paths (np.int64 arrays), costs array)."""
        b = _as_batch(utterances)
        offs, n = b.offsets, len(b)
        paths = np.empty((int(offs[-1]),), dtype=np.int64)
        plen = np.zeros(n, dtype=np.int64)
        cost = np.zeros(n, dtype=np.float64)
        self._check(self._lib.snk_knn_viterbi_batch(self._h, _ptr(b.Q, _c_f64p), _ptr(offs, _c_i64p), n, b.Q.shape[1],
                                                    int(n_candidates), _ptr(paths, _c_i64p),
                                                    _ptr(plen, _c_i64p), _ptr(cost, _c_f64p)))
        out = [paths[offs[u]:offs[u] + plen[u]].copy() for u in range(n)]
        return out, cost

    def set_column_selection(self, target_columns=None, join_columns=None):
        """Stream truncation (truncate_target_streams / truncate_join_streams): indices of the columns
        that take part, None = all.  Takes effect with the next set_weights; queries keep full width."""
        def arr(cols):
            if cols is None:
                return None, -1
            a = np.ascontiguousarray(cols, dtype=np.int32)
            return a, int(a.size)
        t, nt = arr(target_columns)
        j, nj = arr(join_columns)
        ip = ctypes.POINTER(ctypes.c_int)
        self._check(self._lib.snk_set_column_selection(self._h, t.ctypes.data_as(ip) if t is not None else None, nt,
                                                       j.ctypes.data_as(ip) if j is not None else None, nj))

    def knn_viterbi_batch_submit(self, utterances, n_candidates, resident=False):
        """Queue a batch and return at once; at most three batches may be in flight.  Returns a ticket
        for knn_viterbi_batch_collect.  Submitting the next batch before collecting this one hides
        this one's tail (last recursions, copy to the host) behind the next one's K-NN.
        resident: the rows of this very batch are still on the device from the previous submit on the workspace this
        ticket gets (three workspaces take turns: the first three submits must upload) and are searched again without an
        upload (include/snk.h)."""
        b = _as_batch(utterances)
        ticket = ctypes.c_int(-1)
        self._check(self._lib.snk_knn_viterbi_batch_submit(self._h, None if resident else _ptr(b.Q, _c_f64p), _ptr(b.offsets, _c_i64p), len(b),
                                                           b.Q.shape[1], int(n_candidates), ctypes.byref(ticket)))
        return (ticket.value, b)              # the batch (host rows) stays referenced until collected

    def knn_viterbi_batch_collect(self, ticket):
        """Wait for a submitted batch: (list of paths (np.int64 arrays), costs array)."""
        tid, b = ticket
        offs, n = b.offsets, len(b)
        paths = np.empty((int(offs[-1]),), dtype=np.int64)
        plen = np.zeros(n, dtype=np.int64)
        cost = np.zeros(n, dtype=np.float64)
        self._check(self._lib.snk_knn_viterbi_batch_collect(self._h, int(tid), _ptr(paths, _c_i64p), _ptr(plen, _c_i64p),
                                                            _ptr(cost, _c_f64p)))
        return [paths[offs[u]:offs[u] + plen[u]].copy() for u in range(n)], cost

    def set_greedy_layout(self, multiepoch=1, last_frame_as_target=False, join_split_mode=0):
        """get_tree_for_greedy_search (synth_simple.py:190-229) without building anything."""
        self._check(self._lib.snk_set_greedy_layout(self._h, int(multiepoch), int(bool(last_frame_as_target)),
                                                    int(join_split_mode)))

    def greedy(self, unit_features, start_state=-1, search_epsilon=0.0, return_distances=False):
        """greedy_joint_search (synth_simple.py:458-503): list of window indices."""
        Q = _f64(unit_features)
        T, D = Q.shape
        path = np.empty((max(T, 1),), dtype=np.int64)
        dist = np.empty((max(T, 1),), dtype=np.float64)
        n = ctypes.c_int64(0)
        self._check(self._lib.snk_greedy(self._h, _ptr(Q, _c_f64p), T, D, int(start_state), float(search_epsilon),
                                         _ptr(path, _c_i64p), _ptr(dist, _c_f64p), ctypes.byref(n)))
        p = [int(v) for v in path[:n.value]]
        if return_distances:
            return p, dist[:n.value].copy()
        return p

    def greedy_batch(self, utterances, start_states=None, search_epsilon=0.0, return_distances=False):
        """greedy_joint_search for several utterances in one call; up to six (three without the hoisted target term) share every scan of the
        database.  Returns a list of paths (and a list of distance arrays)."""
        b = _as_batch(utterances)
        n = len(b)
        me_steps = np.zeros(n, dtype=np.int64)
        total_rows = int(b.offsets[-1])
        path = np.empty((max(total_rows, 1),), dtype=np.int64)
        dist = np.empty((max(total_rows, 1),), dtype=np.float64)
        st = None if start_states is None else np.ascontiguousarray(start_states, dtype=np.int64)
        self._check(self._lib.snk_greedy_batch(self._h, _ptr(b.Q, _c_f64p), _ptr(b.offsets, _c_i64p), n, b.Q.shape[1],
                                               _ptr(st, _c_i64p) if st is not None else None, float(search_epsilon),
                                               _ptr(path, _c_i64p), _ptr(dist, _c_f64p), _ptr(me_steps, _c_i64p)))
        outs, douts, pos = [], [], 0
        for u in range(n):
            k = int(me_steps[u])
            outs.append([int(v) for v in path[pos:pos + k]])
            douts.append(dist[pos:pos + k].copy())
            pos += k
        return (outs, douts) if return_distances else outs

    def path_scores(self, unit_features, path, mode, n_target_cols, n_join_cols):
        Q = _f64(unit_features)
        p = np.ascontiguousarray(path, dtype=np.int64)
        L = p.size
        tsq = np.empty((L, n_target_cols), dtype=np.float64)
        jsq = np.empty((max(L - 1, 0), n_join_cols), dtype=np.float64)
        self._check(self._lib.snk_path_scores(self._h, _ptr(Q, _c_f64p), _ptr(p, _c_i64p), L, int(mode),
                                              _ptr(tsq, _c_f64p), _ptr(jsq, _c_f64p)))
        return tsq, jsq

    # -- multi-GPU (device pointers) ------------------------------------------
    def knn_local_dev(self, unit_features, n_candidates, d2_dev_ptr, id_dev_ptr):
        Q = _f64(unit_features)
        T, D = Q.shape
        self._check(self._lib.snk_knn_local_dev(self._h, _ptr(Q, _c_f64p), T, D, int(n_candidates),
                                                ctypes.c_void_p(d2_dev_ptr), ctypes.c_void_p(id_dev_ptr)))

    def merge_topk_dev(self, d2_dev_ptr, id_dev_ptr, n_lists, T, n_candidates):
        K = int(n_candidates)
        cand = np.empty((T, K), dtype=np.int64)
        dist = np.empty((T, K), dtype=np.float64)
        self._check(self._lib.snk_merge_topk_dev(self._h, ctypes.c_void_p(d2_dev_ptr), ctypes.c_void_p(id_dev_ptr),
                                                 int(n_lists), int(T), K, _ptr(cand, _c_i64p), _ptr(dist, _c_f64p)))
        return cand, dist

    def knn_local_batch_dev(self, utterances, n_candidates, d2_dev_ptr, id_dev_ptr):
        """Shard-local top-K of the rows of all utterances, written to caller device buffers
        (R, K) in utterance order; complete when the call returns."""
        b = _as_batch(utterances)
        self._check(self._lib.snk_knn_local_batch_dev(self._h, _ptr(b.Q, _c_f64p), _ptr(b.offsets, _c_i64p), len(b),
                                                      b.Q.shape[1], int(n_candidates),
                                                      ctypes.c_void_p(d2_dev_ptr), ctypes.c_void_p(id_dev_ptr)))

    def knn_local_batch_bounds_dev(self, utterances, n_candidates, bound_dev_ptr):
        """Stage A of the shard-local search only: per row of the batch, an upper bound of the K-th
        nearest key of this shard, written to the caller's device buffer (R,) float64.  The caller
        all-reduces (MIN) the bounds of all shards and hands them to knn_local_batch_bounded_dev."""
        b = _as_batch(utterances)
        self._check(self._lib.snk_knn_local_batch_bounds_dev(self._h, _ptr(b.Q, _c_f64p), _ptr(b.offsets, _c_i64p), len(b),
                                                             b.Q.shape[1], int(n_candidates), ctypes.c_void_p(bound_dev_ptr)))
        return b.offsets, b.Q.shape[1]

    def knn_local_batch_bounded_dev(self, lengths, n_columns, n_candidates, bound_dev_ptr, d2_dev_ptr, id_dev_ptr):
        """Shard-local top-K of the batch of the preceding knn_local_batch_bounds_dev call (its query
        rows are still on the device), filtered against the caller's bounds; a list may hold fewer
        than K entries (id -1 padding)."""
        offs = np.zeros(len(lengths) + 1, dtype=np.int64)
        offs[1:] = np.cumsum(lengths)
        self._check(self._lib.snk_knn_local_batch_bounded_dev(self._h, None, _ptr(offs, _c_i64p), len(lengths),
                                                              int(n_columns), int(n_candidates),
                                                              ctypes.c_void_p(bound_dev_ptr),
                                                              ctypes.c_void_p(d2_dev_ptr), ctypes.c_void_p(id_dev_ptr)))

    def merge_viterbi_batch_dev(self, d2_dev_ptr, id_dev_ptr, n_lists, lengths, n_candidates):
        """Owner-rank half of the sharded search: (G, R, K) gathered lists of the utterances with
        the given row counts -> (list of paths, costs)."""
        offs = np.zeros(len(lengths) + 1, dtype=np.int64)
        offs[1:] = np.cumsum(lengths)
        paths = np.empty((int(offs[-1]),), dtype=np.int64)
        plen = np.zeros(len(lengths), dtype=np.int64)
        cost = np.zeros(len(lengths), dtype=np.float64)
        self._check(self._lib.snk_merge_viterbi_batch_dev(self._h, ctypes.c_void_p(d2_dev_ptr),
                                                          ctypes.c_void_p(id_dev_ptr), int(n_lists),
                                                          _ptr(offs, _c_i64p), len(lengths), int(n_candidates),
                                                          _ptr(paths, _c_i64p), _ptr(plen, _c_i64p), _ptr(cost, _c_f64p)))
        return [paths[offs[u]:offs[u] + plen[u]].copy() for u in range(len(lengths))], cost

    # -- waveform side ----------------------------------------------------------
    # ---- collectives inside the library (include/snk.h: snk_comm_*) ------------------------------
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes from ncclGetUniqueId: rank 0 makes them, every rank passes them to comm_init."""
        lib = load_library()
        buf = ctypes.create_string_buffer(128)
        n = ctypes.c_int(0)
        if lib.snk_comm_unique_id(buf, 128, ctypes.byref(n)):
            raise SnkError(lib.snk_last_error().decode())
        return buf.raw[:n.value]

    def comm_init(self, nranks, rank, unique_id):
        """RCCL communicator of this engine (ncclCommInitRank; librccl is loaded here)."""
        self._check(self._lib.snk_comm_init(self._h, int(nranks), int(rank), ctypes.c_char_p(bytes(unique_id))))

    def comm_init_transport(self, nranks, rank, transport):
        """Caller-provided collectives (a TransportCallbacks object); kept alive with the engine."""
        self._transport = transport
        self._check(self._lib.snk_comm_init_transport(self._h, int(nranks), int(rank), ctypes.byref(transport.struct)))

    def comm_destroy(self):
        self._check(self._lib.snk_comm_destroy(self._h))
        self._transport = None

    def upload_global_sample(self, sample_unweighted):
        """Every s-th unit of the WHOLE database (replicated on every rank); before set_weights."""
        S = np.ascontiguousarray(sample_unweighted, dtype=np.float32)
        self._check(self._lib.snk_upload_global_sample(self._h, _ptr(S, _c_f32p), S.shape[0], S.shape[1]))

    def sharded_greedy(self, unit_features, start_state=-1, return_distances=False):
        """snk_sharded_greedy: greedy_joint_search with every step's scan split over the ranks of the communicator (every rank
        holds the whole database and passes the same arguments); the whole path on every rank."""
        Q = _f64(unit_features)
        T, D = Q.shape
        path = np.empty((max(T, 1),), dtype=np.int64)
        dist = np.empty((max(T, 1),), dtype=np.float64)
        n = ctypes.c_int64(0)
        self._check(self._lib.snk_sharded_greedy(self._h, _ptr(Q, _c_f64p), T, D, int(start_state), _ptr(path, _c_i64p),
                                                 _ptr(dist, _c_f64p), ctypes.byref(n)))
        p = [int(v) for v in path[:n.value]]
        if return_distances:
            return p, dist[:n.value].copy()
        return p

    def sharded_knn_viterbi_batch(self, utterances, n_candidates):
        """snk_sharded_knn_viterbi_batch: every rank passes the same batch; returns the paths and costs of
        ALL utterances on every rank."""
        b = _as_batch(utterances)
        n = len(b)
        total = int(b.offsets[-1])
        path = np.empty((max(total, 1),), dtype=np.int64)
        plen = np.zeros(n, dtype=np.int64)
        cost = np.zeros(n, dtype=np.float64)
        self._check(self._lib.snk_sharded_knn_viterbi_batch(self._h, _ptr(b.Q, _c_f64p), _ptr(b.offsets, _c_i64p), n,
                                                            b.Q.shape[1], int(n_candidates), _ptr(path, _c_i64p),
                                                            _ptr(plen, _c_i64p), _ptr(cost, _c_f64p)))
        return [path[int(b.offsets[u]):int(b.offsets[u]) + int(plen[u])].copy() for u in range(n)], cost

    def sharded_knn_viterbi_batch_submit(self, utterances, n_candidates):
        """Queue a sharded step and return a ticket (at most two in flight; every rank issues the same sequence of
        submits and collects).  Submitting step i + 1 before collecting step i runs step i's Viterbi side beside
        step i + 1's K-NN."""
        b = _as_batch(utterances)
        ticket = ctypes.c_int(-1)
        self._check(self._lib.snk_sharded_knn_viterbi_batch_submit(self._h, _ptr(b.Q, _c_f64p), _ptr(b.offsets, _c_i64p), len(b),
                                                                   b.Q.shape[1], int(n_candidates), ctypes.byref(ticket)))
        return (ticket.value, b)              # the batch (host rows) stays referenced until collected

    def sharded_knn_viterbi_batch_collect(self, ticket):
        tid, b = ticket
        offs, n = b.offsets, len(b)
        path = np.empty((max(int(offs[-1]), 1),), dtype=np.int64)
        plen = np.zeros(n, dtype=np.int64)
        cost = np.zeros(n, dtype=np.float64)
        self._check(self._lib.snk_sharded_knn_viterbi_batch_collect(self._h, int(tid), _ptr(path, _c_i64p), _ptr(plen, _c_i64p),
                                                                    _ptr(cost, _c_f64p)))
        return [path[int(offs[u]):int(offs[u]) + int(plen[u])].copy() for u in range(n)], cost

    def upload_frames(self, spec, fzv):
        """spec (rows, 3*H) float32 = [mag | real | imag], fzv (rows, 2) float64 = [f0_interp, vuv]."""
        spec = np.ascontiguousarray(spec, dtype=np.float32)
        fzv = _f64(fzv)
        assert spec.ndim == 2 and spec.shape[1] % 3 == 0 and fzv.shape == (spec.shape[0], 2)
        self._check(self._lib.snk_upload_frames(self._h, _ptr(spec, _c_f32p), _ptr(fzv, _c_f64p), spec.shape[0],
                                                spec.shape[1] // 3))
        self._frames_H = spec.shape[1] // 3

    def concat_fragments(self, first_row, utt_lo, utt_hi, multiepoch, overlap, in_taper):
        first_row = np.ascontiguousarray(first_row, dtype=np.int64)
        utt_lo = np.ascontiguousarray(utt_lo, dtype=np.int64)
        utt_hi = np.ascontiguousarray(utt_hi, dtype=np.int64)
        taper = _f64(in_taper if overlap > 0 else [0.0])
        n = first_row.size
        spec = np.empty((n * multiepoch, 3 * self._frames_H), dtype=np.float64)
        fz = np.empty((n * multiepoch,), dtype=np.float64)
        self._check(self._lib.snk_concat_fragments(self._h, _ptr(first_row, _c_i64p), _ptr(utt_lo, _c_i64p),
                                                   _ptr(utt_hi, _c_i64p), n, int(multiepoch), int(overlap),
                                                   _ptr(taper, _c_f64p), _ptr(spec, _c_f64p), _ptr(fz, _c_f64p)))
        H = self._frames_H
        return spec[:, :H], spec[:, H:2 * H], spec[:, 2 * H:], fz.reshape((-1, 1))

    # -- introspection ----------------------------------------------------------
    def timers(self):
        """{stage: (total_ms, launches)} measured with HIP events on the engine's stream."""
        n = self._lib.snk_timer_count()
        buf = np.zeros(2 * n, dtype=np.float64)
        self._lib.snk_get_timers(self._h, _ptr(buf, _c_f64p), 2 * n)
        return dict((self._lib.snk_timer_name(i).decode(), (float(buf[i]), int(buf[n + i]))) for i in range(n))

    def reset_timers(self):
        self._check(self._lib.snk_reset_timers(self._h))

    def set_option(self, name, value):
        self._check(self._lib.snk_set_option(self._h, name.encode(), float(value)))

    def info(self, name):
        v = ctypes.c_double(0.0)
        self._check(self._lib.snk_get_info(self._h, name.encode(), ctypes.byref(v)))
        return float(v.value)

    def probe_mfma_bf16(self, A_bits, B_bits, C):
        """One v_mfma_f32_32x32x16_bf16 (snk_probe_mfma_bf16): A (32, 16) / B (16, 32) uint16 bf16 bit patterns,
        C (32, 32) float32 -> D (32, 32) float32."""
        A = np.ascontiguousarray(A_bits, dtype=np.uint16).reshape(32, 16)
        B = np.ascontiguousarray(B_bits, dtype=np.uint16).reshape(16, 32)
        Cm = np.ascontiguousarray(C, dtype=np.float32).reshape(32, 32)
        D = np.empty((32, 32), dtype=np.float32)
        self._check(self._lib.snk_probe_mfma_bf16(self._h, A.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)),
                                                  B.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), _ptr(Cm, _c_f32p), _ptr(D, _c_f32p)))
        return D

    def selftest_mfma(self):
        v = ctypes.c_double(0.0)
        self._check(self._lib.snk_selftest_mfma(self._h, ctypes.byref(v)))
        return float(v.value)


def shard_plan(n_items, nranks, rank):
    """snk_shard_plan: contiguous block [lo, hi) of rank (the library's own split; dist.shard_bounds is its twin)."""
    lib = load_library()
    lo, hi = ctypes.c_int64(0), ctypes.c_int64(0)
    if lib.snk_shard_plan(int(n_items), int(nranks), int(rank), ctypes.byref(lo), ctypes.byref(hi)):
        raise SnkError(lib.snk_last_error().decode())
    return lo.value, hi.value


class _TransportStruct(ctypes.Structure):
    _fields_ = [('ctx', ctypes.c_void_p),
                ('all_reduce_min_f64', ctypes.c_void_p),
                ('all_gather', ctypes.c_void_p),
                ('all_to_all_v', ctypes.c_void_p)]


class TransportCallbacks(object):
    """A snk_transport whose collectives are Python callables working on HOST numpy buffers: the device
    buffers the library hands over are copied to the host (snk_copy_to_host), given to

        all_reduce_min(array f64) -> array          all_gather(bytes array) -> concatenated bytes array
        all_to_all_v(send bytes array, send_off, send_bytes, recv_off, recv_bytes, recv_total) -> bytes array

    and the results copied back.  For functional tests (several ranks on one GPU, gloo underneath): the
    production transport is RCCL on the device buffers (comm_init)."""

    _AR = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)
    _AG = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)
    _AA = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, _c_i64p, _c_i64p, ctypes.c_void_p, _c_i64p,
                           _c_i64p, ctypes.c_void_p)

    def __init__(self, nranks, all_reduce_min, all_gather, all_to_all_v, device_sync):
        lib = load_library()
        G = int(nranks)

        def to_host(ptr, nbytes):
            a = np.empty(int(nbytes), dtype=np.uint8)
            if nbytes and lib.snk_copy_to_host(a.ctypes.data_as(ctypes.c_void_p), ptr, int(nbytes)):
                raise SnkError(lib.snk_last_error().decode())
            return a

        def to_dev(ptr, a):
            a = np.ascontiguousarray(a).view(np.uint8)
            if a.size and lib.snk_copy_to_device(ptr, a.ctypes.data_as(ctypes.c_void_p), a.size):
                raise SnkError(lib.snk_last_error().decode())

        def guard(fn):
            def wrapped(*args):
                try:
                    device_sync()              # the buffers were produced on the engine's stream
                    fn(*args)
                    return 0
                except Exception:              # an exception must not unwind through the C frames
                    import traceback
                    traceback.print_exc()
                    return 1
            return wrapped

        def ar(_ctx, buf, n, _stream):
            out = all_reduce_min(to_host(buf, 8 * n).view(np.float64))
            to_dev(buf, np.asarray(out, dtype=np.float64))

        def ag(_ctx, send, recv, nbytes, _stream):
            to_dev(recv, all_gather(to_host(send, nbytes)))

        def aa(_ctx, send, soff, sbytes, recv, roff, rbytes, _stream):
            so = [int(soff[p]) for p in range(G)]; sb = [int(sbytes[p]) for p in range(G)]
            ro = [int(roff[p]) for p in range(G)]; rb = [int(rbytes[p]) for p in range(G)]
            send_total = max(o + b for o, b in zip(so, sb))
            recv_total = max(o + b for o, b in zip(ro, rb))
            out = all_to_all_v(to_host(send, send_total), so, sb, ro, rb, recv_total)
            # only the received blocks are written: send and receive buffer may be the same array with the blocks
            # between them untouched (the query rows of a sharded step)
            base = recv if isinstance(recv, int) else ctypes.cast(recv, ctypes.c_void_p).value
            for p in range(G):
                if rb[p]:
                    to_dev(ctypes.c_void_p(base + ro[p]), out[ro[p]:ro[p] + rb[p]])

        self._keep = (self._AR(guard(ar)), self._AG(guard(ag)), self._AA(guard(aa)))
        self.struct = _TransportStruct(None, ctypes.cast(self._keep[0], ctypes.c_void_p),
                                       ctypes.cast(self._keep[1], ctypes.c_void_p),
                                       ctypes.cast(self._keep[2], ctypes.c_void_p))
