"""One GPU standing in for rank 0 of G in snk_sharded_knn_viterbi_batch (collectives inside the library):
B* database row-sharded G ways, 32 utterances per GPU and step, the replicated 1/16 global sample.  The
transport is a stand-in that answers as the other ranks would (their bounds were computed beforehand by
this GPU, their lists arrive as padding: a row's neighbours sit in one shard on this data), so the DEVICE
stage times of one rank's step are the real ones; the wire time is not measured (one GPU) and is added
from the payload.

    python tools/shard_library_time.py [G] [utts_per_gpu]
"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from snickery_amd.dist import global_sample, shard_bounds
from snickery_amd.engine import TransportCallbacks
from bench import synthetic_db, synthetic_targets

_pos = [a for a in sys.argv[1:] if not a.startswith('--')]
G = int(_pos[0]) if len(_pos) > 0 else 8
UPG = int(_pos[1]) if len(_pos) > 1 else 32
N, Dt, Dj, T, K = 1048576, 61, 302, 600, 100
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt, wj = np.full(Dt, 0.4), np.full(Dj, 0.05)
utts = snickery_amd.QueryBatch([synthetic_targets(F_unw, T, seed=1 + u) * wt for u in range(UPG * G)])
R = int(sum(utts.lengths))
lo, hi = shard_bounds(N, G, 0)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_target_only(F_unw[lo:hi])
eng.upload_join_only(JC_unw)
eng.set_shard(lo, N)
eng.upload_global_sample(global_sample(F_unw, 16))
eng.set_weights(wt, wj)
eng.set_option('shard_gather_queries', 0)     # the stand-in transport has no other ranks to send their rows: upload all of them

state = {'bounds': None, 'capture': None}
COMPACT = '--padded' not in sys.argv
eng.set_option('shard_compact', 1 if COMPACT else 0)


class _Captured(Exception):
    pass


def all_reduce_min(a):
    if state['capture'] is not None:           # capture pass: keep this "rank"'s bounds and abandon the step
        state['capture'] = np.minimum(state['capture'], a)
        raise _Captured()
    if a.size != state['bounds'].size:         # the second bound of a K-NN call: the other shards hold no neighbours of these rows
        return a
    return np.minimum(a, state['bounds'])


def all_gather(a):
    if a.size == 8 * G:                      # the compacted exchange's totals: the other shards hold nothing of the owned rows
        return np.concatenate([a, np.zeros(a.size * (G - 1), dtype=a.dtype)])
    return np.tile(a, G)


def all_to_all_v(send, soff, sbytes, roff, rbytes, recv_total):
    out = np.zeros(recv_total, dtype=np.uint8)
    own = send[soff[0]:soff[0] + sbytes[0]]
    out[roff[0]:roff[0] + rbytes[0]] = own
    if COMPACT:                              # compacted blocks of the other shards: counts of zero, no entries
        return out
    # the other shards' lists of the owned rows: padding (id -1 / +inf) -- recognisable from the dtype of the payload
    is_ids = np.frombuffer(own[:8].tobytes(), dtype=np.int64)[0] < (1 << 40) if own.size >= 8 else True
    pad = np.frombuffer((np.int64(-1) if is_ids else np.float64(np.inf)).tobytes(), dtype=np.uint8)
    for p in range(1, G):
        out[roff[p]:roff[p] + rbytes[p]] = np.tile(pad, rbytes[p] // 8)
    return out


tc = TransportCallbacks(G, all_reduce_min, all_gather, all_to_all_v, torch.cuda.synchronize)
# the bounds the other ranks contribute: this GPU computes them beforehand, standing in for each rank in turn
# (stage A of a rank's own rows against the replicated global sample; the step is abandoned at the all-reduce)
state['capture'] = np.full(R, np.finfo(np.float64).max)
import contextlib, io
for r in range(G):
    eng.comm_init_transport(G, r, tc)
    try:
        with contextlib.redirect_stderr(io.StringIO()):
            eng.sharded_knn_viterbi_batch(utts, K)
    except snickery_amd.SnkError:
        pass
state['bounds'], state['capture'] = state['capture'], None
eng.comm_init_transport(G, 0, tc)
eng.sharded_knn_viterbi_batch(utts, K)                         # warm-up
redo0 = eng.info('batch_redos')
eng.reset_timers()
paths, costs = eng.sharded_knn_viterbi_batch(utts, K)
tm = {k: round(v[0], 2) for k, v in eng.timers().items() if v[1]}
MAIN = ('h2d_queries', 'prepare_queries', 'knn_minima', 'knn_threshold', 'knn_filter', 'knn_bucket', 'knn_finalize', 'merge_topk')
main = sum(tm.get(k, 0) for k in MAIN)                                   # what the main stream carries
side = tm.get('join_lower_bounds', 0) + tm.get('join_exact_sparse', 0) + tm.get('join_costs', 0)   # whole-chip passes of the side streams
own = tm.get('h2d_queries', 0) * (1.0 - 1.0 / G)                          # the stand-in uploads all rows, a rank its own share
print('G=%d rank 0, %d rows per step (%d owned): stages ms %s' % (G, R, R // G, tm))
wire = (G - 1) / G * R * K * 16 / 1e9
print('whole-chip kernels of one rank\'s step of %d frames: main stream %.2f ms (with the upload of the rank\'s own share of the rows: the stand-in '
      'uploads all, %.2f ms) + side streams %.2f ms (bounds and exact costs of the owned rows; here on lists that are mostly padding) = %.2f ms; '
      'exchange payload %.0f MB per rank padded, %.1f MB sent (shard_compact %d)'
      % (R, main - own, main, side, main - own + side, wire * 1e3, eng.info('shard_last_sent_mb'), eng.info('shard_compact')))
print('redone steps: %d, f32 fallbacks: %d' % (eng.info('batch_redos') - redo0, eng.info('f16_fallbacks')))
