"""The minima sweep (knn_sweep16b<minima>, the kernel of stage A) alone over the WHOLE B* database for `rows` query rows
(snk_prefilter_minima): time per launch and what the bf16 pipe issues against its dense peak.
    python tools/minima_time.py [rows]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets, BF16_MFMA_PEAK_TFLOPS
N, Dt, Dj = 1048576, 61, 8
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 600
F, JC = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
U = np.vstack([synthetic_targets(F, 600, seed=1 + s) * wt for s in range((rows + 599) // 600)])[:rows]
eng = snickery_amd.HipSearchEngine(0)
eng.set_option('reorder', 0)
eng.upload_db(F, JC); eng.set_weights(wt, wj)
eng.prefilter_minima(U)
eng.reset_timers()
for _ in range(5): eng.prefilter_minima(U)
tm = eng.timers()
ms = tm['knn_minima'][0] / tm['knn_minima'][1]
rp = (rows + 31) // 32 * 32
print('minima sweep: %d rows x %d units: %.3f ms per launch, issued %.0f TFLOP/s = %.3f of the bf16 peak' % (
    rows, N, ms, 3 * 2.0 * rp * N * 64 / (ms * 1e-3) / 1e12, 3 * 2.0 * rp * N * 64 / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS))
