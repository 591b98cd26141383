"""Drives the HOST side of libsnkhip.so -- argument checks, utterance grouping, the two-batches-in-flight state machine, the
page-locked staging ring and its wrap-around, the refusals in front of a collective -- through the sanitizer build of
`make asan-host` (the library's own translation units compiled for the host with -fsanitize=address,undefined and linked
against tools/fakehip: allocation bookkeeping, no device, kernels never run, every device result reads as zero).
Run by tests/test_host_asan.py in a child process with the sanitizer runtime preloaded; prints HOST-ASAN-OK at the end.
What is checked here is that every call returns (or refuses) as the C ABI documents and that no sanitizer report fires;
results are NOT checked -- there is no device behind the calls."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert os.environ.get('SNK_LIBRARY', '').endswith('libsnkhip_host_asan.so'), 'run through tests/test_host_asan.py'
import snickery_amd                                   # noqa: E402
from snickery_amd import SnkError                     # noqa: E402
from snickery_amd.engine import TransportCallbacks, shard_plan, load_library      # noqa: E402

lib = load_library()
rng = np.random.RandomState(0)
N, Dt, Dj = 6000, 61, 151
F = rng.randn(N, Dt).astype(np.float32)
JC = rng.randn(N + 1, Dj).astype(np.float32)
wt, wj = np.full(Dt, 0.5), np.full(Dj, 0.1)


def refused(fn, *a, **k):
    try:
        fn(*a, **k)
    except (SnkError, AssertionError, ValueError):
        return True
    return False


# ---- creation, uploads, weights: every refusal in front of the first launch ----
assert refused(snickery_amd.HipSearchEngine, 3)                      # no such device
e = snickery_amd.HipSearchEngine(0)
assert refused(e.knn, rng.randn(4, Dt), 5)                           # nothing uploaded
assert refused(e.set_weights, wt, wj)
assert refused(e.upload_db, F, JC[:-1])                              # join rows != N + 1
e.upload_db(F, JC)
assert refused(e.knn, rng.randn(4, Dt), 5)                           # weights not set
assert refused(e.set_weights, wt[:-1], wj)
assert refused(e.set_weights, wt, wj[:-1])
e.set_weights(wt, wj)
assert refused(e.knn, rng.randn(4, Dt + 1), 5)                       # wrong width
assert refused(e.knn, rng.randn(4, Dt), 0)
assert refused(e.knn, rng.randn(4, Dt), 209)
assert refused(e.set_option, 'no_such_option', 1.0)
assert refused(e.set_option, 'viterbi_weights', 2.0)
assert refused(e.set_option, 'join_lb_variant', 3.0)
assert refused(e.info, 'no_such_info')

# ---- single calls of every search entry point (kernels are no-ops: shapes and copies only) ----
U = rng.randn(77, Dt)
cand, dist = e.knn(U, 12)
assert cand.shape == (77, 12)
e.knn_by_class if False else None
e.set_unit_classes(rng.randint(0, 5, N).astype(np.int32))
e.knn_by_class(U, 12, rng.randint(0, 5, 77).astype(np.int32))
c2 = rng.randint(1, N - 1, (30, 12)).astype(np.int64)
d2 = np.sort(rng.rand(30, 12), axis=1)
e.candidate_distances(U[:30], c2)
e.join_costs(c2)
e.join_bounds(c2)
for mode in (0, 1, 2):
    e.set_option('viterbi_mode', mode)
    for variant in (0, 1):
        e.set_option('join_lb_variant', variant)
        e.viterbi(c2, d2)
        e.viterbi_batch([c2, c2[:7], c2[:1]], [d2, d2[:7], d2[:1]])
        e.knn_viterbi(U, 12)
e.set_option('viterbi_weights', 1)
e.viterbi(c2, d2)
e.set_option('viterbi_fst32_slack', 5e-7); assert e.info('viterbi_fst32_slack') == 5e-7
e.knn_viterbi(U, 12)
assert refused(e.set_option, 'viterbi_fst32_slack', 1.0)
e.set_option('viterbi_weights', 0)
# round 5's options and infos: the ladder, the latches, the engine's own order, the tripwires
for name, val in (('reorder', 0), ('reorder', 1), ('reorder_iterations', 2), ('latch_rearm', 0), ('latch_rearm', 1),
                  ('viterbi_latch', 0), ('viterbi_latch', 1), ('viterbi_refine_gate', 0.02), ('join_lb_test_scale', 1.0)):
    e.set_option(name, val)
assert refused(e.set_option, 'viterbi_refine_gate', 2.0)
for name in ('knn_level', 'knn_escalations', 'join_bound_violations', 'join_bound_min_margin', 'greedy_bound_violations',
             'greedy_bound_max_used', 'filter_rearms', 'filter_probe_period', 'viterbi_latch_mode', 'viterbi_latch_switches',
             'reordered', 'reorders', 'reorder_useless', 'reorder_radius_before', 'reorder_radius_after', 'viterbi_refine_gate'):
    e.info(name)
e.set_option('reorder_now', 1)               # the engine's own order at the next K-NN call (no-op kernels: host logic and sizes)
e.knn(U, 12)
e.knn_viterbi_batch([U, U[:9]], 12)
assert refused(e.viterbi, c2, d2[:5])
assert refused(e.join_costs, c2[:1])
assert refused(e.join_bounds, c2[:1])
e.path_scores(U[:30], c2[:, 0], 0, Dt, Dj)
e.prefilter_minima(U)

# ---- batches: one call, three in flight, a fourth refused, out-of-order collects, double collect, state changes under a batch ----
utts = [rng.randn(T, Dt) for T in (40, 3, 75, 1, 22)]
for rows in (12288, 64, 1):                                             # grouping: one group, several, one utterance each
    e.set_option('batch_rows', rows)
    p, c = e.knn_viterbi_batch(utts, 10)
    assert len(p) == 5 and len(c) == 5
e.set_option('batch_rows', 12288)
qb = snickery_amd.QueryBatch(utts).pin()
t0 = e.knn_viterbi_batch_submit(qb, 10)
t1 = e.knn_viterbi_batch_submit(utts[:2], 10)
t9 = e.knn_viterbi_batch_submit(utts[:3], 10)                           # three workspaces: a third batch in flight
assert refused(e.knn_viterbi_batch_submit, utts, 10)                    # three in flight already
e.knn_viterbi_batch_collect(t9)
assert refused(e.knn_viterbi_batch, utts, 10)
assert refused(e.set_weights, wt, wj)                                   # no state change under a batch in flight
assert refused(e.upload_db, F, JC)
assert refused(e.set_column_selection, [0, 1], None)
assert refused(e.join_costs, c2)
e.knn_viterbi_batch_collect(t1)                                         # out of order
assert refused(e.knn_viterbi_batch_submit, qb, 10, resident=True)       # the workspace next in turn holds other rows (t9's: another shape)
t2 = e.knn_viterbi_batch_submit(qb, 10)
e.knn_viterbi_batch_collect(t0)
e.knn_viterbi_batch_collect(t2)
assert refused(e.knn_viterbi_batch_collect, t0)                         # collected already
assert refused(e.knn_viterbi_batch_collect, (7, t0[1]))                 # no such ticket
# where the Viterbi side of a group starts (join_bounds_delay): the last group of a batch is queued by the NEXT submit or by
# its own collect, whichever comes first -- every order of submits and collects, one group and several per batch
for delay in (0, 1, 2, 3, 4, 5):
    e.set_option('join_bounds_delay', delay); assert e.info('join_bounds_delay') == delay
    for rows in (12288, 64):
        e.set_option('batch_rows', rows)
        a = e.knn_viterbi_batch_submit(utts, 10); e.knn_viterbi_batch_collect(a)                  # collect flushes its own tail
        a = e.knn_viterbi_batch_submit(utts, 10); b2 = e.knn_viterbi_batch_submit(utts[:2], 10)     # the second submit flushes the first's
        e.knn_viterbi_batch_collect(b2); e.knn_viterbi_batch_collect(a)                            # out of order
        a = e.knn_viterbi_batch_submit(utts, 10); b2 = e.knn_viterbi_batch_submit(utts[:1], 10)
        e.knn_viterbi_batch_collect(a); c3 = e.knn_viterbi_batch_submit(utts[1:], 10)              # a third behind a pending tail
        e.knn_viterbi_batch_collect(b2); e.knn_viterbi_batch_collect(c3)
assert refused(e.set_option, 'join_bounds_delay', 6)
a = e.knn_viterbi_batch_submit(utts, 10)
assert refused(e.set_option, 'join_bounds_delay', 0)                       # not under a batch in flight
e.knn_viterbi_batch_collect(a)
e.set_option('join_bounds_delay', 1); e.set_option('batch_rows', 12288)
# resident submits: refused until a workspace holds rows of that shape, and again after a new column selection
e2 = snickery_amd.HipSearchEngine(0)
e2.upload_db(F, JC); e2.set_weights(wt, wj)
assert refused(e2.knn_viterbi_batch_submit, qb, 10, resident=True)
for _ in range(3):                                                      # three workspaces take turns: each must have seen the rows once
    e2.knn_viterbi_batch_collect(e2.knn_viterbi_batch_submit(qb, 10))
e2.knn_viterbi_batch_collect(e2.knn_viterbi_batch_submit(qb, 10, resident=True))
e2.set_column_selection(list(range(0, Dt, 2)), None)
e2.set_weights(wt, wj)
assert refused(e2.knn_viterbi_batch_submit, qb, 10, resident=True)      # rows were masked with the old selection
e2.knn_viterbi_batch_collect(e2.knn_viterbi_batch_submit(qb, 10))
# row_offsets[0] != 0 through the raw entry point
Q = np.ascontiguousarray(np.vstack(utts))
offs = np.array([1, 40, 43], dtype=np.int64)
tk = ctypes.c_int(0)
rc = lib.snk_knn_viterbi_batch_submit(e2._h, Q.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), offs.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                      2, Dt, 10, ctypes.byref(tk))
assert rc != 0 and b'row_offsets[0]' in lib.snk_last_error()
e2.close()

# ---- the staging ring: uploads far beyond one 64 MB chunk wrap it several times; pinned and pageable callers ----
big = rng.randn(8192, Dt)                                              # 4 MB per upload
for i in range(40):
    e.knn(big, 4)
pinned = snickery_amd.QueryBatch([big, big[:100]]).pin()
for i in range(6):
    e.knn_viterbi_batch(pinned, 4)

# ---- greedy entry points ----
e.set_greedy_layout(6, False, 0)
for mode in (0, 1, 2):
    e.set_option('greedy_mode', mode)
    e.greedy(rng.randn(60, Dt))
    e.greedy_batch([rng.randn(60, Dt), rng.randn(13, Dt), rng.randn(5, Dt)])
e.set_option('greedy_mode', 2)
assert refused(e.set_greedy_layout, 17, False, 0)                       # multiepoch beyond 16
assert refused(e.greedy, rng.randn(60, Dt + 2))

# ---- row shards: plan, shard-local entry points, the refusals of the sharded step before any collective ----
for n, g in ((10, 3), (7, 8), (1048576, 8), (0, 2)):
    got = [shard_plan(n, g, r) for r in range(g)]
    assert got[0][0] == 0 and got[-1][1] == n and all(a[1] == b[0] for a, b in zip(got, got[1:]))
assert refused(shard_plan, 10, 0, 0) and refused(shard_plan, 10, 2, 2)
assert refused(e.sharded_knn_viterbi_batch, utts, 10)                    # no communicator
calls = []


def _a2a(send, so, sb, ro, rb, rt):
    # rank 0 of two with a silent partner: its own block comes back, the partner's blocks read as zeros
    calls.append('aa')
    out = np.zeros(int(rt), dtype=np.uint8)
    out[ro[0]:ro[0] + rb[0]] = send[so[0]:so[0] + sb[0]]
    return out


def _ag(a):
    calls.append('ag')
    if a.size == 16:                         # the compacted exchange's totals (2 ranks x int64): the partner sends nothing
        return np.concatenate([a, np.zeros(16, dtype=a.dtype)])
    return np.tile(a, 2)


tr = TransportCallbacks(2, lambda a: (calls.append('ar'), a)[1], _ag, _a2a, lambda: None)
e3 = snickery_amd.HipSearchEngine(0)
JC2 = rng.randn(2 * N + 1, Dj).astype(np.float32)
e3.upload_target_only(F); e3.upload_join_only(JC2); e3.set_shard(0, 2 * N); e3.set_weights(wt, wj)
e3.comm_init_transport(2, 0, tr)
assert refused(e3.sharded_knn_viterbi_batch, [rng.randn(4, Dt + 1)], 10) and not calls      # refused BEFORE the first collective
assert refused(e3.sharded_knn_viterbi_batch, utts, 300) and not calls
for opt in ((1, 1, 1), (0, 1, 1), (1, 0, 1), (1, 1, 0)):
    e3.set_option('shard_gather_queries', opt[0]); e3.set_option('shard_refine', opt[1]); e3.set_option('shard_compact', opt[2])
    e3.sharded_knn_viterbi_batch(utts, 10)
e3.set_option('shard_compact', 1)
assert 'ar' in calls and 'aa' in calls and 'ag' in calls
s0 = e3.sharded_knn_viterbi_batch_submit(utts, 10)
s1 = e3.sharded_knn_viterbi_batch_submit(utts[:3], 10)
assert refused(e3.sharded_knn_viterbi_batch_submit, utts, 10)
assert refused(e3.knn_viterbi_batch_submit, utts, 10)                   # a sharded step is in flight
e3.sharded_knn_viterbi_batch_collect(s0)
e3.sharded_knn_viterbi_batch_collect(s1)
assert refused(e3.sharded_greedy, rng.randn(12, Dt))                    # this engine holds a shard
e3.comm_destroy()
assert refused(e3.sharded_knn_viterbi_batch, utts, 10)
e3.close()
# the greedy search with every step's scan split over the ranks: whole database on the rank, one all-gather per step
e4 = snickery_amd.HipSearchEngine(0)
e4.upload_db(F, JC); e4.set_weights(wt, wj)
assert refused(e4.sharded_greedy, rng.randn(12, Dt))                    # no communicator, no layout
e4.comm_init_transport(2, 1, tr)
n_ag = calls.count('ag')
assert refused(e4.sharded_greedy, rng.randn(12, Dt))                    # no layout: refused through the collective verdict (every rank would refuse)
assert calls.count('ag') == n_ag + 1 and 'refused on every rank' in snickery_amd.load_library().snk_last_error().decode()
e4.set_greedy_layout(3, False, 0)
n_ag = calls.count('ag')
e4.sharded_greedy(rng.randn(12, Dt), return_distances=True)
assert calls.count('ag') == n_ag + 1 + 4                                # the collective verdict + one all-gather per step
e4.sharded_greedy(rng.randn(2, Dt))                                     # shorter than one window: no step
e4.close()

# ---- waveform-side gather ----
H = 9
spec = rng.randn(500, 3 * H).astype(np.float32); fzv = rng.rand(500, 2)
e.upload_frames(spec, fzv)
first = np.array([10, 50, 90], dtype=np.int64)
e.concat_fragments(first, np.zeros(3, np.int64), np.full(3, 500, np.int64), 6, 2, np.hanning(4)[:2])
e.timers(); e.reset_timers()
e.close()
assert refused(e.knn, U, 5)                                             # closed handle
print('HOST-ASAN-OK')
