#!/usr/bin/env python3
"""SURVEY 8d's B4 "12 M weak-scaling" database on ONE GPU (the scale the reference cites:
script/active_learning_join.py:109): N units x 61 target columns, (N + 1) x 302 join columns -- the join matrix crosses 2^32
elements at N = 12 M, so every 32-bit offset in the host and kernel index arithmetic is exercised.

  python tools/bigdb_time.py [N] [--k 100,200] [--utts 4] [--greedy-steps 10]

Checks (each against the C oracle, test infrastructure):  16 K-NN rows against the brute force over the WHOLE database;
the whole Viterbi of the first utterance against the oracle's recursion on the device's candidates (join rows of the
candidates gathered into a compact matrix: the float64 copy of the whole join matrix would be 29 GB);  the greedy
search's first steps (join_split_mode 1: the 302 join columns as two halves of 151) against snko_greedy_f32;  rows taken
from the LAST units of the database (the high offsets).  Prints stage times and one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def walk_matrix(rows, cols, seed, block=16):
    """cumsum(randn(rows, cols)) / global std as float32 (SURVEY 8d's generator), built by column blocks so that the
    float64 intermediate stays at rows x block."""
    rng = np.random.RandomState(seed)
    out = np.empty((rows, cols), np.float32)
    s1 = s2 = 0.0
    for c0 in range(0, cols, block):
        c1 = min(cols, c0 + block)
        x = np.cumsum(rng.randn(rows, c1 - c0), axis=0)
        s1 += float(x.sum()); s2 += float((x * x).sum())
        out[:, c0:c1] = x
        del x
    n = float(rows) * cols
    std = np.sqrt(s2 / n - (s1 / n) ** 2)
    for c0 in range(0, cols, 64):
        out[:, c0:c0 + 64] *= np.float32(1.0 / std)
    return out


def compact_join(JC_unw, wj, cand):
    """float64 weighted join rows of the candidates only, ids renumbered: rows c and c + 1 of every candidate c, sorted
    (so that the renumbered c + 1 follows the renumbered c), with a dummy row at either end (ids 0 and N - 1 are unusable
    in the reference, synth_halfphone.py:3238-3268: the renumbered ids must stay clear of both)."""
    ids = np.unique(np.concatenate([cand.reshape(-1), cand.reshape(-1) + 1]))
    ids = ids[(ids >= 0) & (ids < JC_unw.shape[0])]
    JCc = np.zeros((ids.size + 2, JC_unw.shape[1]), np.float64)
    JCc[1:-1] = JC_unw[ids].astype(np.float64) * wj
    pos = np.searchsorted(ids, cand) + 1
    return JCc, pos.astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('units', nargs='?', type=int, default=12000000)
    ap.add_argument('--k', default='100,200')
    ap.add_argument('--utts', type=int, default=4)
    ap.add_argument('--frames', type=int, default=600)
    ap.add_argument('--greedy-steps', type=int, default=10)
    ap.add_argument('--no-greedy', action='store_true')
    args = ap.parse_args()
    N, Dt, Dj, T = args.units, 61, 302, args.frames
    import snk_oracle_c as oc
    import snickery_amd

    def meminfo():
        try:
            with open('/proc/meminfo') as f:
                return dict((l.split(':')[0], int(l.split()[1]) // 1024) for l in f if l.split(':')[0] in ('MemTotal', 'MemAvailable'))
        except OSError:
            return {}
    print('host memory (MB):', meminfo(), flush=True)
    t0 = time.time()
    F_unw = walk_matrix(N, Dt, 0)
    JC_unw = walk_matrix(N + 1, Dj, 1)
    print('database built in %.1f s: F %s (%.2f GB), JC %s (%.2f GB, %d elements = 2^%.2f)' % (
        time.time() - t0, F_unw.shape, F_unw.nbytes / 1e9, JC_unw.shape, JC_unw.nbytes / 1e9, JC_unw.size, np.log2(JC_unw.size)), flush=True)
    wt, wj = np.full(Dt, 0.4), np.full(Dj, 0.05)
    rng = np.random.RandomState(5)
    # utterances: random starts, one from the very end of the database (the highest row offsets), one across the 2^32-element
    # boundary of the join matrix (row 2^32 / 302 = 14 221 746 for Dj = 302 -- inside the database from N = 14.3 M; for N = 12 M
    # the padded device copies (pitch 304 / 320 columns) cross it at rows 14.1 M / 13.4 M: not inside either, so the end is the test)
    starts = [int(rng.randint(0, N - T)) for _ in range(args.utts)]
    starts[-1] = N - T - 3
    if args.utts > 2:
        starts[-2] = min(N - T - 1, (1 << 32) // 304 - T // 2) if N > (1 << 32) // 304 else N // 2
    utts = [(F_unw[s:s + T].astype(np.float64) + 0.3 * np.random.RandomState(100 + i).randn(T, Dt)) * wt for i, s in enumerate(starts)]

    eng = snickery_amd.HipSearchEngine(0)
    t0 = time.time()
    eng.upload_db(F_unw, JC_unw)
    t_up = time.time() - t0
    t0 = time.time()
    eng.set_weights(wt, wj)
    t_w = time.time() - t0
    print('upload %.1f s, set_weights %.2f s' % (t_up, t_w), flush=True)
    out = {'units': N, 'target_dim': Dt, 'join_dim': Dj, 'join_elements': int(JC_unw.size), 'upload_s': t_up, 'set_weights_s': t_w, 'legs': []}
    F = None
    ok_all = True
    for K in [int(k) for k in args.k.split(',')]:
        leg = {'K': K}
        before = (eng.info('f16_fallbacks'), eng.info('batch_redos'), eng.info('exact_row_fallbacks'))
        eng.reset_timers()
        paths, costs = eng.knn_viterbi_batch(utts, K)          # first call: builds the float32 join copy
        eng.reset_timers()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            paths, costs = eng.knn_viterbi_batch(utts, K)
        dt = (time.perf_counter() - t0) / reps
        tm = eng.timers()
        leg['ms_per_batch'] = dt * 1e3
        leg['frames_per_s'] = args.utts * T / dt
        leg['stages_ms_per_batch'] = dict((k, v[0] / reps) for k, v in tm.items() if v[1])
        leg['fallbacks'] = [eng.info('f16_fallbacks') - before[0], eng.info('batch_redos') - before[1], eng.info('exact_row_fallbacks') - before[2]]
        leg['prefilter_margin_rows'] = eng.info('prefilter_margin_rows'); leg['prefilter_min_margin'] = eng.info('prefilter_min_margin')
        leg['list_mean'] = eng.info('last_list_mean')
        print('K=%d: %.2f ms per batch of %d x %d frames (%.0f frames/s), fallbacks %s, stages %s' % (
            K, dt * 1e3, args.utts, T, leg['frames_per_s'], leg['fallbacks'], dict((k, round(v, 3)) for k, v in leg['stages_ms_per_batch'].items())), flush=True)
        # parity: the first and the last utterance
        if F is None:
            t0 = time.time()
            F = F_unw.astype(np.float64) * wt
            print('float64 weighted copy for the oracle: %.1f s' % (time.time() - t0), flush=True)
        for ui in (0, args.utts - 1):
            U = utts[ui]
            path, cost, cand, dist = eng.knn_viterbi(U, K, return_candidates=True)
            same_batch = bool(np.array_equal(np.asarray(path), np.asarray(paths[ui])) and cost == costs[ui])
            rows = np.unique(np.linspace(0, T - 1, 16).astype(np.int64))
            t0 = time.time()
            oc_cand, oc_dist = oc.knn(F, U[rows], K)
            t_or = time.time() - t0
            knn_ok = bool(np.array_equal(cand[rows], oc_cand) and np.array_equal(dist[rows], oc_dist))
            JCc, pos = compact_join(JC_unw, wj, cand)
            pos[(cand < 1) | (cand >= N - 1)] = 0          # unusable in the reference (first / last unit): unusable after the renumbering too
            opath, ocost = oc.viterbi(pos, dist, JCc)
            back = dict(zip(pos.reshape(-1).tolist(), cand.reshape(-1).tolist()))
            opath = [back[p] for p in opath]
            vit_ok = bool(list(path) == opath and cost == ocost)
            print('  utterance %d (start %d): 16 K-NN rows == oracle over %d units: %s (oracle %.1f s); Viterbi == oracle: %s; single == batch: %s; max id %d' % (
                ui, starts[ui], N, knn_ok, t_or, vit_ok, same_batch, int(cand.max())), flush=True)
            leg.setdefault('parity', []).append({'utterance': ui, 'start': starts[ui], 'knn_rows_equal_oracle': knn_ok, 'viterbi_equals_oracle': vit_ok,
                                                'single_equals_batch': same_batch})
            ok_all = ok_all and knn_ok and vit_ok and same_batch
        out['legs'].append(leg)
    del F
    if not args.no_greedy:
        me = 6
        eng.set_greedy_layout(me, False, 1)
        for st in (-1, N - 5000):
            s0 = N - 4000 if st >= 0 else starts[0]
            U = (F_unw[s0:s0 + args.greedy_steps * me].astype(np.float64) + 0.3 * np.random.RandomState(7).randn(args.greedy_steps * me, Dt)) * wt
            eng.greedy(U, start_state=st)
            eng.reset_timers()
            t0 = time.perf_counter()
            path, d = eng.greedy(U, start_state=st, return_distances=True)
            dt = time.perf_counter() - t0
            ms, launches = eng.timers()['greedy_steps']
            t0 = time.time()
            op, od = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, False, 1, st)
            g_ok = bool(path == op and np.array_equal(d, od))
            print('greedy (me 6, split mode 1, start %d): %d steps, %.1f us per step on the device, path == oracle: %s (oracle %.1f s); f16 launches %d, path head %s' % (
                st, len(path), ms / max(launches, 1) / max(len(path), 1) * 1e3, g_ok, time.time() - t0, eng.info('greedy_f16_launches'), path[:3]), flush=True)
            out.setdefault('greedy', []).append({'start_state': st, 'steps': len(path), 'us_per_step': ms / max(launches, 1) / max(len(path), 1) * 1e3,
                                                 'equals_oracle': g_ok, 'wall_ms': dt * 1e3})
            ok_all = ok_all and g_ok
    out['all_equal_oracle'] = ok_all
    eng.close()
    print(json.dumps(out))
    return 0 if ok_all else 1


if __name__ == '__main__':
    sys.exit(main())
