"""Randomised parity sweep on the GPU: random shapes / weights / options through K-NN, class-restricted
K-NN, join costs + Viterbi (single and batch) and the greedy search, each compared bit for bit with
the oracle.  Prints one line per case and a summary; exit code 1 on any mismatch.

    python tests/fuzz_parity.py [n_cases] [seed]
    SNK_FUZZ_OPTS=greedy_f16=2 python tests/fuzz_parity.py ...     (engine options for every case)
"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import snickery_amd
import snk_oracle as o


def one_case(rng, idx):
    N = int(rng.choice([97, 300, 1000, 2500, 6000, 15000]))
    Dt = int(rng.choice([1, 7, 20, 61, 64, 65, 90, 128, 184, 250]))
    Dj = int(rng.choice([2, 16, 40, 151, 302]))
    K = int(rng.choice([1, 2, 7, 30, 50, 100, 200]))
    T = int(rng.choice([1, 2, 3, 17, 40]))
    me = int(rng.choice([1, 2, 5, 6, 9, 16]))
    lfat = bool(rng.rand() < 0.3)
    mode = int(rng.rand() < 0.3 and Dj % 2 == 0)
    precision = int(rng.choice([0, 1]))
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=int(rng.randint(1 << 30)))
    if rng.rand() < 0.3:                                  # duplicated stretches: exact ties
        a, b, n = int(rng.randint(N // 2)), int(N // 2 + rng.randint(N // 4)), int(1 + rng.randint(min(50, N // 4)))
        F_unw[b:b + n] = F_unw[a:a + n]
    wt = rng.rand(Dt) * 0.9 + 0.05
    wj = rng.rand(Dj) * 0.3 + 0.01
    if rng.rand() < 0.3:
        wt[rng.rand(Dt) < 0.3] = 0.0                       # truncated streams
    F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
    # stream truncation: the engine keeps full-width arrays and queries, the oracle gets dropped columns
    tsel = jsel = None
    if rng.rand() < 0.25 and mode == 0:
        tsel = np.sort(rng.choice(Dt, size=int(rng.randint(1, Dt + 1)), replace=False))
        jsel = np.sort(rng.choice(Dj, size=int(rng.randint(1, Dj + 1)), replace=False))
        F, E, S = F[:, tsel], E[:, jsel], S[:, jsel]
    drop = (lambda U: U[:, tsel]) if tsel is not None else (lambda U: U)
    desc = 'N=%d Dt=%d Dj=%d K=%d T=%d me=%d lfat=%d mode=%d prec=%d sel=%d' % (N, Dt, Dj, K, T, me, lfat, mode, precision, tsel is not None)
    eng = snickery_amd.HipSearchEngine(0)
    for kv in os.environ.get('SNK_FUZZ_OPTS', '').split(','):      # e.g. SNK_FUZZ_OPTS=greedy_f16=2,greedy_mode=1
        if '=' in kv:
            eng.set_option(kv.split('=')[0], float(kv.split('=')[1]))
    bad = []
    try:
        eng.set_option('precision', precision)
        eng.upload_db(F_unw, JC_unw)
        if tsel is not None:
            eng.set_column_selection(tsel, jsel)
        eng.set_weights(wt, wj)
        U = o.synthetic_targets(F_unw, T, seed=int(rng.randint(1 << 30)), noise=float(rng.choice([0.0, 0.3, 2.0]))) * wt
        Keff = min(K, 208)
        cand, d = eng.knn(U, Keff)
        oc, od = o.knn_bruteforce(F, drop(U), Keff)
        if not (np.array_equal(cand, oc) and np.array_equal(d, od)):
            bad.append('knn')
        ncls = int(rng.choice([1, 3, 45]))
        ucls = rng.randint(ncls, size=N).astype(np.int32)
        qcls = rng.choice(np.unique(ucls), size=T).astype(np.int32)    # (a class without units is an error in the reference)
        eng.set_unit_classes(ucls)
        cc, cd = eng.knn_by_class(U, Keff, qcls)
        occ, ocd = o.knn_by_class(F, drop(U), Keff, ucls, qcls)
        if not (np.array_equal(cc, occ) and np.array_equal(cd, ocd)):
            bad.append('knn_by_class')
        if T >= 1:
            path, cost = eng.viterbi(oc, od)
            op, ocost = o.viterbi(oc, od, E, S)
            if not (list(path) == list(op) and (cost == ocost or (np.isnan(cost) and np.isnan(ocost)) or len(op) == 0)):
                bad.append('viterbi')
            lens = [T, max(1, T // 2), T + 3]
            utts = [o.synthetic_targets(F_unw, t, seed=int(rng.randint(1 << 30))) * wt for t in lens]
            paths, costs = eng.knn_viterbi_batch(utts, Keff)
            for u, Uu in enumerate(utts):
                c2, d2 = o.knn_bruteforce(F, drop(Uu), Keff)
                p2, cost2 = o.viterbi(c2, d2, E, S)
                if not (list(paths[u]) == list(p2) and (len(p2) == 0 or costs[u] == cost2)):
                    bad.append('batch[%d]' % u)
        if N >= me and (mode == 0 or Dj % 2 == 0) and Dt <= 250:
            eng.set_greedy_layout(me, lfat, mode)
            Tg = max(T, me * 3)
            Ug = o.synthetic_targets(F_unw, Tg, seed=int(rng.randint(1 << 30))) * wt
            start = int(rng.choice([-1, 0, (N - me) // 2]))
            gp, gd = eng.greedy(Ug, start_state=start, return_distances=True)
            pr, cr, Fwin = o.greedy_layout(F, E, S, me, lfat, mode)
            og, ogd = o.greedy_search(pr, cr, Fwin, o.greedy_queries(drop(Ug), me, lfat), start_state=start)
            if not (list(gp) == list(og) and np.array_equal(gd, ogd)):
                bad.append('greedy')
            # two utterances per scan
            Ug2 = o.synthetic_targets(F_unw, max(me, Tg - 5), seed=int(rng.randint(1 << 30))) * wt
            bp, bd = eng.greedy_batch([Ug, Ug2, Ug[:Tg // 2 + 1]], start_states=[start, -1, start], return_distances=True)
            og2, ogd2 = o.greedy_search(pr, cr, Fwin, o.greedy_queries(drop(Ug2), me, lfat), start_state=-1)
            s3, d3 = eng.greedy(Ug[:Tg // 2 + 1], start_state=start, return_distances=True)
            if not (bp[0] == list(og) and np.array_equal(bd[0], ogd) and bp[1] == list(og2) and np.array_equal(bd[1], ogd2)
                    and bp[2] == s3 and np.array_equal(bd[2], d3)):
                bad.append('greedy_batch')
    except Exception as e:                                   # an engine error is a failure too
        bad.append('EXC %s: %s' % (type(e).__name__, str(e)[:200]))
    finally:
        eng.close()
    print('%4d %s : %s' % (idx, desc, 'ok' if not bad else 'MISMATCH ' + ','.join(bad)), flush=True)
    return not bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.RandomState(seed)
    t0 = time.time()
    ok = sum(one_case(rng, i) for i in range(n))
    print('%d / %d cases ok in %.0f s' % (ok, n, time.time() - t0))
    sys.exit(0 if ok == n else 1)


if __name__ == '__main__':
    main()
