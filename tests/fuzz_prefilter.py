"""Randomised parity sweep of the K-NN at sizes where the matrix prefilter is engaged (bf16-split one to three
chunks, float32 operands elsewhere): walks, clouds with offsets, duplicated stretches, zero weights, scaled data;
candidates and distances against the C oracle, bit for bit.  Test infrastructure; run on a GPU box:

    python tests/fuzz_prefilter.py [n_cases] [seed]
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
import snk_oracle_c as oc


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    eng = snickery_amd.HipSearchEngine(0)
    bad = 0
    t0 = time.time()
    per_db = int(os.environ.get('SNK_FUZZ_PER_DB', '8'))     # cases per database: new weights (= new operands, bounds, balls), K, rows, noise
    i = 0
    margin_rows, min_margin = 0.0, float('inf')
    budget = float(os.environ.get('SNK_FUZZ_SECONDS', '0'))     # > 0: stop starting new databases after that many seconds
    while i < n_cases and not (budget > 0 and time.time() - t0 > budget):
        N = int(rng.choice([40000, 100000, 300000]))
        Dt = int(rng.choice([20, 45, 61, 100, 123, 125, 150, 184, 189, 200]))
        kind = int(rng.randint(4))
        if kind == 0:
            F_unw, _ = o.synthetic_db(N, Dt, 8, seed=int(rng.randint(1 << 30)))
        else:
            F_unw = rng.randn(N, Dt).astype(np.float32)
            if kind == 2:
                F_unw += np.float32(rng.choice([3.0, 30.0, -7.0]))
            if kind == 3:
                F_unw *= np.float32(rng.choice([1e-3, 50.0]))
        if rng.rand() < 0.4:                               # duplicated stretch: exact ties
            a, b, n = int(rng.randint(N // 2)), int(N // 2 + rng.randint(N // 4)), int(1 + rng.randint(300))
            F_unw[b:b + n] = F_unw[a:a + n]
        JC_unw = rng.randn(N + 1, 8).astype(np.float32)
        eng.upload_db(F_unw, JC_unw)
        sd = float(F_unw[:20000].std())
        for _ in range(min(per_db, n_cases - i)):
            K = int(rng.choice([1, 16, 50, 100, 200]))
            T = int(rng.choice([33, 100, 257]))
            wt = rng.rand(Dt) * 0.9 + 0.05
            if rng.rand() < 0.3:
                wt[rng.rand(Dt) < 0.3] = 0.0
            eng.set_option('prefilter', int(rng.choice([1, 1, 2, 0])))
            eng.set_option('prefilter_balls', int(rng.choice([1, 1, 0])))
            eng.set_weights(wt, np.full(8, 0.1))
            s = int(rng.randint(0, N - T))
            U = (F_unw[s:s + T].astype(np.float64) + float(rng.choice([0.0, 0.05, 0.3])) * sd * rng.randn(T, Dt)) * wt
            F = o.weight(F_unw, wt)
            before = eng.info('f16_fallbacks')
            eng.reset_timers()                             # the tripwire counters are since the last reset
            cand, dist = eng.knn(U, K)
            ocand, odist = oc.knn(F, U, K)
            ok = np.array_equal(cand, ocand) and np.array_equal(dist, odist)
            bad += not ok
            bf = int(eng.info('prefilter_bf16_active'))
            mr, mm = (eng.info('prefilter_margin_rows'), eng.info('prefilter_min_margin')) if bf else (0.0, float('inf'))
            margin_rows += mr
            min_margin = min(min_margin, mm)
            print('%4d N=%d Dt=%d K=%d T=%d kind=%d bf16=%d coarse=%d fallbacks=%d margin_rows=%d min_margin=%.2f : %s' % (
                i, N, Dt, K, T, kind, bf, int(eng.info('filter_coarse')), int(eng.info('f16_fallbacks') - before), int(mr), mm,
                'ok' if ok else 'MISMATCH'), flush=True)
            i += 1
    print('tripwire over all bf16 cases: prefilter_margin_rows %d, smallest margin %.2f' % (int(margin_rows), min_margin))
    print('%d / %d cases ok in %.0f s' % (i - bad, i, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
