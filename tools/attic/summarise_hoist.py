"""profiles/<tag>_greedy.md from the passes of tools/prof_hoist.sh (gpurun_out/ph)."""
import csv, glob, collections, sys, os
src = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/ph'
tag = sys.argv[2] if len(sys.argv) > 2 else 'r02_h'
N, Dt, Dj, me, steps = 1500000, 61, 151, 6, 100
KS, KP = 'greedy32_kernel<false, true, 1, true>', 'hoist_product_kernel'       # one utterance, float16 join tiles


def durations(d, ker):
    out = []
    for f in glob.glob(os.path.join(src, d, '*', '*_kernel_trace.csv')):
        for r in csv.DictReader(open(f)):
            if ker in r['Kernel_Name']:
                out.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return out


def counters(ker):
    ctr = {}
    for f in sorted(glob.glob(os.path.join(src, '*', '*', '*_counter_collection.csv'))):
        acc = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            if ker in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
        for k in acc:
            ctr[k] = acc[k] / cnt[k]
    return ctr


ds, dp = durations('stats', KS), durations('stats', KP)
ds1, dp1 = durations('stats_b1', 'greedy32_kernel<false, true, 3, false>'), durations('stats_b1', KP)
cs, cp = counters(KS), counters(KP)
scan, prod = sum(ds) / len(ds), sum(dp) / len(dp)
alg = N * (Dj + 1) * 4.0
streamed = N * ((Dj + 7) // 8 * 16 + 4.0)              # 19 16-byte columns of eight halves + one float32 target value per window
fetch_s = cs['FETCH_SIZE'] * 1024 * 2 / steps          # KB reported, x2: gfx950 correction (MI355X_MICROARCH.md, HBM section)
write_p = cp.get('WRITE_SIZE', 0.0) * 1024
fetch_p = cp['FETCH_SIZE'] * 1024 * 2
K = me * ((Dt + 63) // 64) * 64
flop = 2.0 * ((steps + 15) // 16 * 16) * ((N - me + 1 + 127) // 128 * 128) * K
lines = ['# Round 2 -- greedy search with the hoisted target term and float16 join tiles at N = 1.5 M units, Dt = 61, Dj = 151, multiepoch 6 (B3 shape)', '',
         'Command: `bash tools/prof_hoist.sh` (rocprofv3 --kernel-trace --stats, then separate --pmc passes, over tools/prof_hoist.py: 600-frame utterances = 100 steps each, one persistent launch per utterance).', '',
         '* `greedy32_kernel<false, true, 1, true>` (the scan over float16 join tiles, all 100 steps in one launch, exact decisions included): %d launches, average **%.2f ms** = %.1f us per step' % (len(ds), scan / 1e3, scan / steps),
         '* `hoist_product_kernel` (float64 matrix pipe, W = 100 x 1.5 M target terms): %d launches, average **%.2f ms** = %.1f us per step of the utterance; %.3g FLOP (padded) -> %.1f TFLOP/s = %.0f %% of the 78.6 TFLOP/s float64 matrix peak' % (
             len(dp), prod / 1e3, prod / steps, flop, flop / (prod * 1e-6) / 1e12, 100 * flop / (prod * 1e-6) / 78.6e12),
         '* per step, scan + product: **%.1f us**; algorithmic bytes (Dj + 1) x 4 x N = %.0f MB -> %.2f TB/s = **%.0f %% of the 8 TB/s HBM peak** (the scan alone: %.0f %%; on the %.0f MB it requests, 19 16-byte columns of eight float16 join values + one target value per window: %.0f %%)' % (
             (scan + prod) / steps, alg / 1e6, alg / ((scan + prod) / steps * 1e-6) / 1e12, 100 * alg / ((scan + prod) / steps * 1e-6) / 8e12,
             100 * alg / (scan / steps * 1e-6) / 8e12, streamed / 1e6, 100 * streamed / (scan / steps * 1e-6) / 8e12),
         '* HBM traffic of the scan per step: FETCH_SIZE x2 (gfx950 correction) = %.0f MB = %.2fx the algorithmic bytes, %.2fx the requested ones; WRITE_SIZE %.2f MB per step' % (
             fetch_s / 1e6, fetch_s / alg, fetch_s / streamed, cs.get('WRITE_SIZE', 0.0) * 1024 / steps / 1e6),
         '* HBM traffic of the product per launch: fetch %.0f MB (the database once: %.0f MB), write %.0f MB (W: %.0f MB)' % (
             fetch_p / 1e6, N * 64 * 4 / 1e6, write_p / 1e6, steps * ((N + 127) // 128 * 128) * 4 / 1e6),
         '* scan: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = %.2f, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.2f (s_waitcnt); vector memory reads %.3g per step' % (
             cs['SQ_ACTIVE_INST_VALU'] / cs['SQ_WAVE_CYCLES'], cs['SQ_WAIT_INST_ANY'] / cs['SQ_WAVE_CYCLES'], cs['SQ_INSTS_VMEM_RD'] / steps),
         '* product: SQ_VALU_MFMA_BUSY_CYCLES %.3g over 1 024 SIMDs = %.2f M cycles each = %.2f ms at 2.4 GHz (%.2f ms at the 1.85 GHz the chip holds under matrix load) of the launch\'s %.2f ms: the kernel is bound by the float64 matrix pipe' % (
             cp['SQ_VALU_MFMA_BUSY_CYCLES'], cp['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / 1e6, cp['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / 2.4e9 * 1e3,
             cp['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / 1.85e9 * 1e3, prod / 1e3),
         '* the same at N = 65 536 (B1 shape): scan %.2f ms = %.1f us per step, product %.3f ms' % (
             sum(ds1) / len(ds1) / 1e3, sum(ds1) / len(ds1) / steps, sum(dp1) / len(dp1) / 1e3), '',
         'raw per-launch averages, scan: ' + ', '.join('%s=%.4g' % kv for kv in sorted(cs.items())), '',
         'raw per-launch averages, product: ' + ', '.join('%s=%.4g' % kv for kv in sorted(cp.items()))]
open(os.path.join('profiles', tag + '_greedy.md'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
