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
    for i in range(n_cases):
        N = int(rng.choice([40000, 100000, 300000]))
        Dt = int(rng.choice([20, 45, 61, 100, 123, 125, 150, 184, 189, 200]))
        K = int(rng.choice([1, 16, 50, 100, 200]))
        T = int(rng.choice([33, 100, 257]))
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
        wt = rng.rand(Dt) * 0.9 + 0.05
        if rng.rand() < 0.3:
            wt[rng.rand(Dt) < 0.3] = 0.0
        eng.set_option('prefilter', int(rng.choice([1, 1, 2, 0])))
        eng.upload_db(F_unw, JC_unw)
        eng.set_weights(wt, np.full(8, 0.1))
        s = int(rng.randint(0, N - T))
        U = (F_unw[s:s + T].astype(np.float64) + float(rng.choice([0.0, 0.05, 0.3])) * F_unw.std() * rng.randn(T, Dt)) * wt
        F = o.weight(F_unw, wt)
        before = eng.info('f16_fallbacks')
        cand, dist = eng.knn(U, K)
        ocand, odist = oc.knn(F, U, K)
        ok = np.array_equal(cand, ocand) and np.array_equal(dist, odist)
        bad += not ok
        print('%3d N=%d Dt=%d K=%d T=%d kind=%d bf16=%d fallbacks=%d : %s' % (
            i, N, Dt, K, T, kind, int(eng.info('prefilter_bf16_active')), int(eng.info('f16_fallbacks') - before),
            'ok' if ok else 'MISMATCH'), flush=True)
    print('%d / %d cases ok in %.0f s' % (n_cases - bad, n_cases, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
