"""profiles/<tag>_greedy.md from the passes of tools/prof_greedy.sh (gpurun_out/pg)."""
import csv, glob, collections, sys, os
src = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pg'
tag = sys.argv[2] if len(sys.argv) > 2 else 'r01_f'
KER = 'greedy_step_kernel'
N, Dt, Dj, me = 1500000, 61, 151, 6
ctr = {}
for f in sorted(glob.glob(os.path.join(src, '*', '*', '*_counter_collection.csv'))):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if KER in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
    for k in acc:
        ctr[k] = acc[k] / cnt[k]
dur = []
for f in glob.glob(os.path.join(src, 'stats', '*', '*_kernel_trace.csv')):
    for r in csv.DictReader(open(f)):
        if KER in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
avg = sum(dur) / len(dur)
alg = N * (Dj + Dt) * 4.0
fetch = ctr['FETCH_SIZE'] * 1024 * 2          # KB reported, x2: gfx950 correction (MI355X_MICROARCH.md, HBM section)
write = ctr.get('WRITE_SIZE', 0.0) * 1024
wave = ctr['SQ_WAVE_CYCLES']
lines = ['# Round 1 -- greedy_step_kernel<lds> at N = 1.5 M units, Dt = 61, Dj = 151, multiepoch 6 (B3 shape)', '',
         'Command: `bash tools/prof_greedy.sh` (rocprofv3 --kernel-trace --stats, then separate --pmc passes, over tools/prof_greedy.py: one 600-frame utterance = 100 steps).', '',
         '* kernel-trace: %d launches, average **%.1f us** per step (under the tracer; tools/greedy_time.py with HIP events and no tracer: 260 us = 4.9 TB/s = 61 %%)' % (len(dur), avg),
         '* algorithmic bytes per step (Dj + Dt) x 4 x N = %.0f MB -> %.2f TB/s = **%.0f %% of the 8 TB/s HBM peak**' % (alg / 1e6, alg / (avg * 1e-6) / 1e12, 100 * alg / (avg * 1e-6) / 8e12),
         '* HBM traffic per step: FETCH_SIZE %.0f KB reported -> x2 (gfx950 correction) = %.0f MB; WRITE_SIZE %.1f MB; => %.2fx the algorithmic bytes' % (ctr['FETCH_SIZE'], fetch / 1e6, write / 1e6, (fetch + write) / alg),
         '* vector ALU: SQ_INSTS_VALU %.3g wavefront instructions per step (%.1f per column and window); SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = %.2f of the wavefront cycles; two wavefronts per SIMD -> VALU busy ~%.0f %%' % (
             ctr['SQ_INSTS_VALU'], ctr['SQ_INSTS_VALU'] * 64 / (N * (Dj + me * Dt)), ctr['SQ_ACTIVE_INST_VALU'] / wave, 200 * ctr['SQ_ACTIVE_INST_VALU'] / wave),
         '* LDS: SQ_INSTS_LDS %.3g, SQ_LDS_BANK_CONFLICT %.0f, SQ_LDS_IDX_ACTIVE %.3g (of GRBM_GUI_ACTIVE %.3g per XCD x 256 CUs: ~%.0f %% busy)' % (
             ctr['SQ_INSTS_LDS'], ctr['SQ_LDS_BANK_CONFLICT'], ctr['SQ_LDS_IDX_ACTIVE'], ctr['GRBM_GUI_ACTIVE'], 100 * ctr['SQ_LDS_IDX_ACTIVE'] / (ctr['GRBM_GUI_ACTIVE'] / 8 * 256)),
         '* waiting: SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.2f (s_waitcnt), SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.2f; SQ_WAIT_INST_LDS share %.3f' % (
             ctr['SQ_WAIT_INST_ANY'] / wave, ctr['SQ_WAIT_ANY'] / wave, ctr['SQ_WAIT_INST_LDS'] / wave),
         '* SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = %.2f; SALU instructions %.3g; vector memory reads %.3g per step' % (
             ctr['SQ_ACTIVE_INST_ANY'] / wave, ctr['SQ_INSTS_SALU'], ctr['SQ_INSTS_VMEM_RD']), '',
         'raw per-launch averages: ' + ', '.join('%s=%.4g' % kv for kv in sorted(ctr.items()))]
open(os.path.join('profiles', tag + '_greedy.md'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
