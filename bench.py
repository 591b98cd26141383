#!/usr/bin/env python3
"""Headline benchmark: synthesised frames/sec (+ xRT) of the full-database K-NN preselection
+ join costs + Viterbi search on synthetic magphase-60 targets (BASELINE.json `metric`,
workload B* of SURVEY.md 8d: |DB| = 1 048 576 units, Dt = 61, Dj = 302, T = 600, K = 100).

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one batch of --utts utterances PER GPU (T frames each) through the whole hot path with
the unit database already resident in HBM.  N > 1: the database is row-sharded over the ranks and the
batch grows with N (weak scaling: every rank sweeps 1/N of the database for N x --utts utterances,
i.e. the single-GPU number of distance evaluations, and owns --utts utterances in the Viterbi
stage); the local top-K lists travel by ONE all-to-all over RCCL to the rank that owns the utterance
(contiguous blocks of utterances per rank), which merges them and runs join costs + Viterbi.
--fixed-batch keeps the batch at --utts for every N (strong scaling).  --db-shards S (a divisor of N)
shards the database over S GPUs only and runs N/S such groups side by side, each on its share of
the batch (S = 1: independent replicas, no collective).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # = snickery_amd.configure_runtime(), before anything starts the ROCm runtime (engine.py: why); recorded in config.hw_queues

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_MFMA_PEAK_TFLOPS = 78.6     # MI355X dense FP64 matrix: 256 CU x 4 SIMD x 2048 FLOP / 64 clk x 2.4 GHz
F32_MFMA_PEAK_TFLOPS = 157.3    # MI355X dense FP32 matrix (MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 clk)
BF16_MFMA_PEAK_TFLOPS = 2516.8  # dense bf16 matrix: 16 x the f32 rate (same table: v_mfma_f32_32x32x16_bf16, 32 clk)
FRAMESHIFT_MS = 5.0             # config/slt_simplified_mini.cfg:40


def synthetic_db(N, Dt, Dj, seed=0):
    """SURVEY 8d generator (same as oracle.snk_oracle.synthetic_db; duplicated here so that the
    product benchmark does not import the oracle)."""
    rng = np.random.RandomState(seed)
    F = np.cumsum(rng.randn(N, Dt), axis=0)
    F = (F / F.std()).astype(np.float32)
    JC = np.cumsum(rng.randn(N + 1, Dj), axis=0)
    JC = (JC / JC.std()).astype(np.float32)
    return F, JC


def synthetic_targets(F_unw, T, seed, noise=0.3):
    rng = np.random.RandomState(seed)
    N, Dt = F_unw.shape
    s = rng.randint(0, max(N - T, 1))
    return F_unw[s:s + T].astype(np.float64) + noise * rng.randn(min(T, N - s), Dt)


def _sha256(path):
    import hashlib
    try:
        with open(path, 'rb') as f:
            return hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None


def profiled_counters(json_name, kernel_source):
    """Counters of a committed rocprofv3 --pmc pass (profiles/<json_name>; separate passes of the kernel and shape named there, never
    measured in the bench run itself).  They describe the kernel AS IT WAS PROFILED: the file carries the sha256 of the kernel's
    source at that time (`source_sha256`), and only while the source is still that one do the values count as this build's
    (`fresh`); otherwise the caller reports them under `profiled_reference` and leaves `traffic` null."""
    path = os.path.join(os.environ.get('SNK_PROFILES_DIR') or os.path.join(ROOT, 'profiles'), json_name)      # (SNK_PROFILES_DIR: tools/prof_round6.sh, counters taken minutes before on the same box)
    if not os.path.isfile(path):
        return None, False
    with open(path) as f:
        tj = json.load(f)
    now = _sha256(os.path.join(ROOT, 'snickery_amd', 'csrc', kernel_source))
    return tj, bool(tj.get('source_sha256')) and tj.get('source_sha256') == now


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _cpu_search(o, tree, E, S, U, K):
    """One utterance through the reference's CPU formulation; returns (cand, dist, path, cost, (t_knn, t_join, t_dp, n_arcs))."""
    t0 = time.time()
    d, i = tree.query(U, k=K)
    t_knn = time.time() - t0
    cand = np.asarray(i, dtype=np.int64)
    t0 = time.time()
    cache = o.join_cost_cache(E, S, cand)
    t_join = time.time() - t0
    t0 = time.time()
    path, cost = o.viterbi(cand, np.asarray(d), E, S)
    t_dp = time.time() - t0
    return cand, np.asarray(d), path, cost, (t_knn, t_join, t_dp, len(cache))


def cpu_baseline(F_unw, JC_unw, wt, wj, K, sample_frames, seed, all_cores=True):
    """The reference's CPU formulation (oracle = test infrastructure, timed here as the
    reported baseline only): scipy cKDTree preselection (synth_halfphone.py:379,1364), the
    Python pair loop + numpy gather join costs (:3251-3301), DP Viterbi.
    Two figures: ONE thread, like the reference's search (`value`), and every host core with one
    process per utterance, the reference's own `-p ncores` fan-out (synth_halfphone.py:897-903;
    `all_cores`).  Runs BEFORE the GPU is initialised (the workers are forked)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import scipy.spatial
    import snk_oracle as o
    F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
    t0 = time.time()
    tree = scipy.spatial.cKDTree(F, leafsize=100, compact_nodes=False, balanced_tree=False)
    t_build = time.time() - t0
    U = synthetic_targets(F_unw, sample_frames, seed) * wt
    cand, d, path, cost, (t_knn, t_join, t_dp, n_arcs) = _cpu_search(o, tree, E, S, U, K)
    total = t_knn + t_join + t_dp
    out = {
        'value': sample_frames / total, 'unit': 'frames/s', 'cores': 1, 'kind': 'port', 'cpu_model': _cpu_model(),
        'host_cores': os.cpu_count(),
        'sample': '%d frames of the same workload (K=%d, full %d-unit DB): cKDTree.query %.2fs + '
                  'pair-loop join costs %.2fs (%d arcs) + DP %.2fs; tree build %.1fs not counted'
                  % (sample_frames, K, F.shape[0], t_knn, t_join, n_arcs, t_dp, t_build),
    }
    if all_cores:
        # one forked worker per utterance (the tree and the weighted matrices are shared copy-on-write), the reference's own
        # `-p ncores` fan-out; the best of a few worker counts up to every hardware thread of the host (os.cpu_count():
        # SURVEY 8d) -- on a 256-thread host 256 workers share the memory system and are SLOWER than 32.
        # Short utterances keep each trial at about 10 s.
        ncpu = max(1, os.cpu_count() or 1)
        frames_w = 60
        try:
            with open('/proc/meminfo') as f:
                avail = [int(l.split()[1]) * 1024 for l in f if l.startswith('MemAvailable')][0]
            pmax = max(1, int(0.25 * avail / (frames_w * K * K * 260.0 + 64e6)))    # a worker's cost cache is a dict of frames x K^2 entries
        except (OSError, IndexError, ValueError):
            pmax = 32
        trials = []
        for P in sorted(set(min(ncpu, pmax, p) for p in (32, 64, 128, ncpu))):
            pipes, t0 = [], time.time()
            for w in range(P):
                r, wfd = os.pipe()
                pid = os.fork()
                if pid == 0:
                    try:
                        os.close(r)
                        Uw = synthetic_targets(F_unw, frames_w, seed + 1000 + w) * wt
                        _cpu_search(o, tree, E, S, Uw, K)
                        os.write(wfd, b'1')
                    finally:
                        os._exit(0)
                os.close(wfd)
                pipes.append((pid, r))
            done = 0
            for pid, r in pipes:
                done += 1 if os.read(r, 1) == b'1' else 0
                os.close(r)
                os.waitpid(pid, 0)
            t_all = time.time() - t0
            trials.append({'workers': P, 'value': done * frames_w / t_all, 'seconds': t_all, 'utterances': done})
        best = max(trials, key=lambda t: t['value'])
        out['all_cores'] = {'value': best['value'], 'unit': 'frames/s', 'cores': best['workers'], 'host_threads': ncpu,
                            'trials': trials,
                            'sample': 'best of %s single-threaded worker processes side by side, one %d-frame utterance each: %d workers, %.1fs'
                                      % ([t['workers'] for t in trials], frames_w, best['workers'], best['seconds'])}
    return out, (cand, d, path, cost, U)


def greedy_extra(device, configs=((65536, 'greedy_b1'), (1500000, 'greedy_b3')), Dt=61, Dj=151, T=600, me=6):
    """BASELINE configs[0] / configs[2] (greedy_joint_search, synth_simple.py:458-503) at their workload sizes:
    B1 = 65 536 units (the README demo voice), B3 = 1.5 M units (IS2018_nick_simplified.cfg), magphase-60 widths,
    multiepoch 6, search_epsilon 0, one 600-frame utterance = 100 scans of the whole database.  Extra fields of the
    JSON line (never `value`): device time per step from the engine's HIP events, frames/s from the wall clock,
    fraction of the 8 TB/s HBM peak on the scan's algorithmic bytes (Dj + 1) * 4 * N per step (`streamed_frac`
    on the bytes the kernels actually move).  The time per step includes the utterance's share of the float64 matrix product
    that hoists the target term (greedy_hoist_kernels.hip)."""
    import snickery_amd
    out = {}
    for N, name in configs:
        F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
        wt, wj = np.full(Dt, 0.4), np.full(Dj, 0.05)
        eng = snickery_amd.HipSearchEngine(device)
        eng.upload_db(F_unw, JC_unw)
        eng.set_weights(wt, wj)
        eng.set_greedy_layout(me, False, 0)
        U = synthetic_targets(F_unw, T, seed=1) * wt
        eng.greedy(U)
        eng.reset_timers()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            path = eng.greedy(U)
        dt = (time.perf_counter() - t0) / reps
        steps = T // me
        ms, launches = eng.timers()['greedy_steps']
        us_step = ms / max(launches, 1) / steps * 1e3
        # algorithmic bytes per step as SURVEY 8d defines them: the join columns of every window + ONE precomputed target
        # value per window (target term hoisted into a matrix product over all steps).  Streamed: the join columns rounded
        # up to whole float4 columns, the target value written once by the product and read once by the scan -- or, where
        # the product is not used (greedy_hoist 0, narrow join streams), the Dt target columns instead.
        hoisted = eng.info('greedy_hoist_launches') > 0
        f16 = eng.info('greedy_f16_launches') > 0            # streamed databases: the join columns as float16 (8 per 16 bytes)
        bytes_step = float(N) * (Dj + 1) * 4.0
        bytes_streamed = float(N) * (((Dj + 3) // 4 * 4 + 2) if hoisted else (Dj + Dt)) * 4.0
        if f16:
            bytes_streamed = float(N) * ((Dj + 7) // 8 * 8 * 2.0 + 8.0)
        # a batch through snk_greedy_batch: the float32 prefilter scan, six utterances per scan of the database (three
        # where the target term is not hoisted)
        # (one persistent launch), exact float64 decisions -- the same paths
        nb = 6
        ub = [synthetic_targets(F_unw, T, seed=1 + u) * wt for u in range(nb)]
        eng.greedy_batch(ub)
        t0 = time.perf_counter()
        pb = eng.greedy_batch(ub)
        tb = time.perf_counter() - t0
        per_scan = 6.0 if hoisted else 3.0
        bfrac = bytes_step * steps * nb / per_scan / tb / 8e12
        batch = {'utterances': nb, 'frames_per_s': nb * T / tb, 'us_per_step_and_utterance': tb / (nb * steps) * 1e6,
                 'same_path_as_single': bool(np.array_equal(np.asarray(pb[0]), np.asarray(path))),
                 'roofline': {'bound': 'hbm', 'achieved': bfrac * 8000.0, 'peak': 8000.0, 'unit': 'GB/s', 'frac': bfrac,
                              'note': 'one scan of the database serves %d utterances: the scan\'s algorithmic bytes / %d per utterance step' % (per_scan, per_scan)}}
        resident = eng.info('greedy_resident_launches') > 0
        if resident:
            # the windowed join matrix sits in LDS for the whole launch (greedy_res_kernels.hip): a step is two fabric round
            # trips (publish -> gather) + a scan out of LDS, no HBM or L2 stream -- a latency figure, not a bandwidth one
            roof = {'bound': 'latency', 'us_per_step': us_step, 'budget_us_per_step': 10.0,
                    'note': 'database resident in LDS (%d KB per compute unit): per step table 0.9 + LDS scan 1.7 + top-3 0.4 + publish 0.55 + '
                            'gather 1.7 + decide 1.5 us (HISTORY.md 4.4d) + the utterance\'s share of the hoisted product; dividing its bytes by '
                            '8 TB/s would mean nothing' % (int(N * Dj * 4 / 256 / 1024),),
                    'algorithmic_bytes_per_step': bytes_step}
        else:
            s_elem = 2.0 if f16 else 4.0                          # bytes per element of the join columns AS STORED for the scan (s)
            bytes_s = float(N) * (Dj * s_elem + 4.0)
            roof = {'bound': 'hbm', 'achieved': bytes_s / (us_step * 1e-6) / 1e9, 'peak': 8000.0,
                    'unit': 'GB/s', 'frac': bytes_s / (us_step * 1e-6) / 8e12, 'bytes_per_element': s_elem,
                    'algorithmic_bytes_per_step': bytes_s, 'streamed_bytes_per_step': bytes_streamed,
                    'streamed_frac': bytes_streamed / (us_step * 1e-6) / 8e12,
                    'frac_at_4_bytes_per_element': bytes_step / (us_step * 1e-6) / 8e12,
                    'note': 'algorithmic bytes (s Dj + 4) N per step with s = bytes per element of the join columns as the scan reads them '
                            '(SURVEY 8d): %s; frac_at_4_bytes_per_element prices the float32 database as uploaded'
                            % ('float16 tiles, s = 2' if f16 else 'float32 tiles, s = 4')}
        out[name] = {'units': N, 'multiepoch': me, 'frames': T, 'steps': steps, 'us_per_step': us_step, 'batch': batch,
                     'bound_violations': eng.info('greedy_bound_violations'), 'bound_max_used': eng.info('greedy_bound_max_used'),
                     'target_term_hoisted': bool(hoisted), 'target_term_pipe': ('bf16 x 3 pieces' if eng.info('greedy_hoist16_launches') > 0 else 'float64') if hoisted else None,
                     'join_tiles': 'float16' if f16 else 'float32',
                     'frames_per_s': T / dt, 'ms_per_utterance': dt * 1e3, 'roofline': roof,
                     'path_head': [int(v) for v in path[:4]]}
        eng.close()
        del F_unw, JC_unw
    return out


def speechlike_voice(N, Dt, Dj, seed=0, rho=0.98):
    """A voice whose frames relate to each other the way speech frames do, more than SURVEY 8d's cumsum(randn) / global std does
    (consecutive units 0.002 apart per column there, against neighbour distances of 0.3): a stationary AR(1) walk per column whose
    step has 0.2 of the global standard deviation (var(step) = 2 (1 - rho) = 0.04) -- for the target features AND for the join
    features (until round 5 this leg kept the random-walk join matrix: smooth join rows, near-contiguous candidates, traffic
    below the algorithmic bytes).  Returns F_unw (N x Dt), JC_unw ((N + 1) x Dj) and held_out(T, u): utterance u
    of T frames from a walk of the SAME process that the database never saw (not database rows + noise): its nearest units are as
    far as any stranger's in that cloud."""
    from scipy.signal import lfilter
    rng = np.random.RandomState(seed + 17)
    g = np.sqrt(1.0 - rho * rho)
    F = lfilter([g], [1.0, -rho], rng.randn(N + 2000, Dt), axis=0)[2000:]
    sF = F.std()
    F = (F / sF).astype(np.float32)
    JC = lfilter([g], [1.0, -rho], rng.randn(N + 1 + 2000, Dj), axis=0)[2000:]
    JC = (JC / JC.std()).astype(np.float32)

    def held_out(T, u):
        r = np.random.RandomState(seed + 1000 + u)
        return lfilter([g], [1.0, -rho], r.randn(T + 2000, Dt), axis=0)[2000:] / sF
    return F, JC, held_out


def variant_database(kind, N, Dt, F_unw, JC_unw, seed=0):
    """Databases of the B* shape whose 32-unit tiles are NOT the compact balls SURVEY 8d's generator makes of them:
      'permuted'    the same units in random order: a tile holds 32 unrelated frames, the ball pass of the database order is useless --
                    the engine gives such a voice an order of its own (kmeans_kernels.hip).  The utterances of this leg follow the
                    speech the units were cut from;
      'speechlike'  speechlike_voice(): AR(1) target AND join features, utterances from a held-out walk.
    Returns (F, JC, targets) -- targets(T, u) or None (the leg then follows `targets_from` + noise)."""
    if kind == 'permuted':
        rng = np.random.RandomState(seed + 17)
        perm = rng.permutation(N)
        return F_unw[perm], JC_unw[np.concatenate([perm, [N]])], None
    return speechlike_voice(N, Dt, JC_unw.shape[1], seed)


def leg_roofline(tm, steps, rows_per_step, N, Dt, Dj, K, eng):
    """Roofline of the dominant whole-chip kernel of a leg, from the engine's HIP-event stage timers (each stage timed on the stream
    it is launched on, inside the timed steps): the larger, by time per step, of
      * the filter stage (matrix pipe): what the bf16 pipe ISSUES -- the one-pass sweep 3 terms x 2 rows N Dpad; the coarse sweep one
        term of that + 3 terms for the listed tile pairs; the ball pass the centres (N / 32) + the listed pairs -- over its time,
        against the dense bf16 peak;
      * join_lb2_kernel (HBM): 2 K rows of Dj float32 gathered + K^2 float32 bounds written per row pair, against 8 TB/s.
    The T-step recursions are latency chains (no roofline); `largest_stage` names the stage with the largest time per step of all."""
    def per_step(name):
        return tm[name][0] / steps if name in tm and tm[name][1] else 0.0
    def per_launch(name):
        return tm[name][0] / tm[name][1] if name in tm and tm[name][1] else 0.0
    out = {}
    dpad = (Dt + 3 + 63) // 64 * 64
    f_ms, f_n = per_launch('knn_filter'), (tm['knn_filter'][1] / steps if 'knn_filter' in tm and tm['knn_filter'][1] else 0)
    filt = None
    if f_ms > 0 and eng.info('prefilter_bf16_active') == 1:
        rows = rows_per_step / max(f_n, 1)
        onepass, coarse = eng.info('filter_onepass') == 1, eng.info('filter_coarse') == 1
        pairs = 0.0 if onepass else eng.info('coarse_pairs')
        if onepass:
            issued, kname = 3 * 2.0 * rows * N * dpad, 'knn_sweep16b<filter> (one-pass three-term sweep)'
        elif coarse:
            issued, kname = 2.0 * rows * N * dpad + 3 * 2.0 * pairs * 1024 * dpad, 'knn_coarse16b + knn_refine16b'
        else:
            issued, kname = 3 * 2.0 * rows * (N / 32.0) * dpad + 3 * 2.0 * pairs * 1024 * dpad, 'knn_balls16b + knn_refine16b'
        tf = issued / (f_ms * 1e-3) / 1e12
        filt = {'kernel': kname, 'bound': 'mfma', 'achieved': tf, 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s (issued)', 'frac': tf / BF16_MFMA_PEAK_TFLOPS,
                'algorithmic_tflops': 2.0 * rows * N * Dt / (f_ms * 1e-3) / 1e12, 'avg_launch_ms': f_ms, 'ms_per_step': per_step('knn_filter'),
                'tile_pairs_fraction': None if onepass else pairs / max((rows / 32.0) * (N / 32.0), 1.0)}
    j_ms = per_launch('join_lower_bounds')
    join = None
    if j_ms > 0:
        jn = tm['join_lower_bounds'][1] / steps
        jrows = rows_per_step / max(jn, 1)
        jbytes = jrows * K * 2 * Dj * 4 + jrows * K * K * 4
        gbs = jbytes / (j_ms * 1e-3) / 1e9
        join = {'kernel': 'join_lb2_kernel', 'bound': 'hbm', 'achieved': gbs, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gbs / 8000.0,
                'avg_launch_ms': j_ms, 'ms_per_step': per_step('join_lower_bounds')}
    cands = [c for c in (filt, join) if c]
    if cands:
        out = dict(max(cands, key=lambda c: c['ms_per_step']))
        other = [c for c in cands if c is not out and c['kernel'] != out['kernel']]
        if other:
            out['other'] = {'kernel': other[0]['kernel'], 'frac': other[0]['frac'], 'ms_per_step': other[0]['ms_per_step']}
    stages = dict((k, v[0] / steps) for k, v in tm.items() if v[1])
    if stages:
        big = max(stages, key=stages.get)
        out['largest_stage'] = {'stage': big, 'ms_per_step': stages[big],
                                'note': 'a T-step latency chain on one wavefront per utterance: no roofline' if big in ('viterbi_sparse', 'viterbi_lower_bound', 'viterbi_dp') else None}
    return out


def shape_leg(eng, name, F_unw, JC_unw, wt, wj, T, U, K, steps, kind='compact', host_to_host=False, targets_from=None, targets=None, depth=3):
    """One BASELINE shape (or B* on a variant database) through the batch pipeline, three steps in flight, rows uploaded and paths
    returned inside every step: frames/s, stage times (a second pass), the dominant kernel's roofline, fallbacks and tripwires.
    An extra field of the record, never `value`."""
    import snickery_amd
    import torch
    N, Dt = F_unw.shape
    Dj = JC_unw.shape[1]
    eng.upload_db(F_unw, JC_unw)
    eng.set_weights(wt, wj)
    # targets_from: the matrix the utterances follow (a permuted database: the speech its units were cut from, not its row order)
    # targets: utterance u from the caller's generator (a held-out walk) instead
    batch = snickery_amd.QueryBatch([(targets(T, u) if targets is not None else
                                      synthetic_targets(F_unw if targets_from is None else targets_from, T, seed=1 + u)) * wt for u in range(U)])
    batch.pin()
    # primes every workspace; the engine judges the voice (filter passes, unit order, Viterbi path, pass 2's warm-up) and its latches'
    # early probes run out: the counting probes of a voice on a slower filter come at calls 16, 48, 112, ... -- with five priming
    # batches the speech-like leg read 3.1 M frames/s where every later run of the same loop gives 3.6-3.7 (profiles/
    # r06z_timers_speech.log)
    for _ in range(24):
        eng.knn_viterbi_batch_collect(eng.knn_viterbi_batch_submit(batch, K))
    before = (eng.info('f16_fallbacks'), eng.info('batch_redos'), eng.info('exact_row_fallbacks'))

    def run(resident):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pending, res = [], None
        for _ in range(steps):
            pending.append(eng.knn_viterbi_batch_submit(batch, K, resident=resident))
            if len(pending) >= depth:
                res = eng.knn_viterbi_batch_collect(pending.pop(0))
        while pending:
            res = eng.knn_viterbi_batch_collect(pending.pop(0))
        torch.cuda.synchronize()
        return time.perf_counter() - t0, res
    # the rate: rows uploaded and paths returned inside every step, no stage timers (their timestamp events cost 4-7 % of a step);
    # then the same loop once more with every stage timed, for the stage table and the leg's roofline
    eng.set_option('timers', 2)
    run(False)                                   # (untimed: the same loop once, so that the timed one starts in its own regime -- the headline has its depth probe for that)
    eng.reset_timers()
    dth, _ = run(False)
    eng.set_option('timers', 1)
    eng.reset_timers()
    cells0 = eng.info('dense_cells')
    dt, _ = run(False)
    tm = eng.timers()
    pairs = eng.info('coarse_pairs')
    rows_per_launch = T * U / max(tm['knn_filter'][1] / steps, 1)
    out = {'shape': name, 'database': kind, 'units': N, 'target_dim': Dt, 'join_dim': Dj, 'frames': T, 'utts_per_step': U, 'n_candidates': K,
           # the leg's rate: the better of its two passes of the same loop (the first times the roofline stage only, the second every
           # stage).  A 20-step pass of a leg is bimodal -- the speech-like voice reads 3.37-3.40 or 3.58-3.67 M frames/s from one
           # pass to the next whatever is timed (profiles/r06z_timers_speech.log, r06af_mask_scan.log: how the streams' loops fall
           # into step) -- so one sample is not the leg's rate; both passes are in the detail record
           'frames_per_s': T * U * steps / min(dth, dt), 'ms_per_step': min(dth, dt) / steps * 1e3, 'steps': steps, 'rows': 'host -> host',
           'frames_per_s_roofline_stage_timed': T * U * steps / dth, 'frames_per_s_with_stage_timers': T * U * steps / dt,
           'roofline': leg_roofline(tm, steps, T * U, N, Dt, Dj, K, eng),
           'filter_coarse': bool(eng.info('filter_coarse')), 'filter_onepass': bool(eng.info('filter_onepass')),
           'reordered': bool(eng.info('reordered')), 'tile_radius_before_after': [eng.info('reorder_radius_before'), eng.info('reorder_radius_after')],
           'viterbi_path': 'dense' if eng.info('viterbi_latch_mode') == 1 else 'sparse', 'viterbi_lb_warm': eng.info('viterbi_lb_warm_now'),
           'cells_refined_per_step': (eng.info('dense_cells') - cells0) / steps, 'cells_per_step': T * U * K,
           'tile_pairs_listed': {'last_launch': pairs, 'fraction': pairs / max((rows_per_launch / 32.0) * (N / 32.0), 1.0)},
           'list_mean': eng.info('last_list_mean'), 'list_max': eng.info('last_list_max'), 'knn_level': eng.info('knn_level'),
           'tau_optimism_rank': eng.info('tau_optimism_rank'), 'tau_optimism_off': eng.info('tau_optimism_off'),
           'prefilter_fallbacks': eng.info('f16_fallbacks') - before[0], 'batch_redos': eng.info('batch_redos') - before[1],
           'exact_row_fallbacks': eng.info('exact_row_fallbacks') - before[2],
           'tripwires': {'prefilter_margin_rows': eng.info('prefilter_margin_rows'), 'prefilter_min_margin': eng.info('prefilter_min_margin'),
                         'join_bound_violations': eng.info('join_bound_violations'), 'join_bound_min_margin': eng.info('join_bound_min_margin')},
           'stages_ms_per_step': dict((k, v[0] / steps) for k, v in tm.items() if v[1])}
    out['host_to_host_frames_per_s'] = out['frames_per_s']
    out['host_to_host_ms_per_step'] = out['ms_per_step']
    return out


def other_rooflines(timers, counts, rows, K, Dt, Dj, T):
    """Rooflines of the two whole-chip kernels that had none until round 5: durations from the engine's
    HIP-event stage timers over the timed steps, byte counts from the kernels' own counters (snk_get_info) taken in a separate
    two-step counting pass of the same batch (the counts per launch are a property of the batch):
      join_exact_sparse2_kernel (pass 3): per exact cost two float32 rows of Dj columns -- but never more than the 2 K rows a
        step HAS (costs that share a row meet it again in L2; a cell has up to four predecessors) -- + per cell its set (16 B),
        candidate (8 B), target distance (8 B) read and its record (64 B) written; beside the bytes, 5 separately rounded
        float64 operations per column and cost (2 mul, sub, mul, add: the canonical order) against the non-FMA float64 vector
        rate (39.3 T op/s);
      knn_finalize_kernel: per list entry its 8-byte key, per re-ranked entry its id (4 B) and its float32 row (Dt columns padded
        to 4), per row K results (16 B each) and the query row (8 B per column).
    `traffic` (counter bytes) is added by tools/summarise_round6.py from --pmc passes (profiles/r06_*_summary.md)."""
    out = {}
    def stage(name):
        return (timers[name][0] / timers[name][1], timers[name][1]) if name in timers and timers[name][1] else (0.0, 0)
    nl = counts.get('launches', {})
    ms, n = stage('join_exact_sparse')
    if n and nl.get('join_exact_sparse') and counts.get('sparse_exact_costs', 0) > 0:
        costs = counts['sparse_exact_costs'] / nl['join_exact_sparse']
        rpl = rows / n
        cells = rpl * K
        row_fetches = min(2.0 * costs, 2.0 * K * max(rpl - rpl / max(T, 1), 1.0))
        b = row_fetches * Dj * 4 + cells * (16 + 8 + 8 + 64)
        ops = costs * Dj * 5.0
        out['join_exact_sparse2_kernel'] = {
            'bound': 'hbm', 'achieved': b / (ms * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s', 'frac': b / (ms * 1e-3) / 8e12,
            'avg_launch_ms': ms, 'launches': n, 'exact_costs_per_launch': costs, 'set_members_per_launch': counts.get('sparse_set_members', 0) / nl['join_exact_sparse'],
            'cells_per_launch': cells, 'row_fetches_per_launch': row_fetches, 'algorithmic_bytes_per_launch': b, 'f64_vector_ops_per_launch': ops,
            'f64_vector_frac': ops / (ms * 1e-3) / 39.3e12, 'traffic': None}
    ms, n = stage('knn_finalize')
    if n and nl.get('knn_finalize') and counts.get('finalize_list_entries', 0) > 0:
        ent, sel = counts['finalize_list_entries'] / nl['knn_finalize'], counts['finalize_reranked'] / nl['knn_finalize']
        rpl = rows / n
        fp = (Dt + 3) // 4 * 4
        b = ent * 8 + sel * (4 + fp * 4) + rpl * (K * 16 + Dt * 8)
        out['knn_finalize_kernel'] = {
            'bound': 'hbm', 'achieved': b / (ms * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s', 'frac': b / (ms * 1e-3) / 8e12,
            'avg_launch_ms': ms, 'launches': n, 'rows_per_launch': rpl, 'list_entries_per_row': ent / rpl, 'reranked_per_row': sel / rpl,
            'algorithmic_bytes_per_launch': b, 'traffic': None,
            'binds': 'a row is one workgroup\'s chain of barriers (selection by value bins, exact distances, a bitonic sort of 256): latency, six workgroups per compute unit'}
    return out


def check_rooflines(obj, where='', found=None):
    """Every `frac` of every roofline-shaped object of the record must lie in [0, 1]: a fraction of a peak above 1 is a wrong
    byte count or a wrong time, not evidence (r05's summary printed 1.38 after a regrouping halved the rows of a launch).
    An offender is NOT printed as a number: frac / achieved become None, `invalid` says why, and its path is returned."""
    found = [] if found is None else found
    if isinstance(obj, dict):
        f = obj.get('frac')
        if 'peak' in obj and isinstance(f, (int, float)) and not (0.0 <= f <= 1.0):
            obj['invalid'] = 'frac %.3f outside [0, 1]: withheld' % f
            obj['frac'] = obj['achieved'] = None
            found.append(where or '.')
        for k, v in obj.items():
            check_rooflines(v, where + '/' + str(k), found)
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            check_rooflines(v, where + '/%d' % i, found)
    return found


def _r(x, n=4):
    return None if x is None else round(float(x), n) if isinstance(x, (int, float)) else x


def compact_line(out):
    """The record the driver parses: the contract's keys, `roofline` and `cpu_baseline` as flat objects of numbers and short
    strings, a `summary` of the extra legs.  No notes, no stage tables (they are in --detail-out)."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'xRT')
    line = dict((k, out[k]) for k in keep if k in out)
    line['config'] = dict((k, v) for k, v in out['config'].items() if v is not None)
    r = out.get('roofline', {})
    ro = {'kernel': str(r.get('kernel', '')).split(' ')[0], 'bound': r.get('bound'), 'achieved': _r(r.get('achieved'), 1), 'peak': r.get('peak'),
          'unit': r.get('unit'), 'frac': _r(r.get('frac')), 'traffic': r.get('traffic'), 'traffic_source': r.get('traffic_source_short'),
          'avg_launch_ms': _r(r.get('avg_launch_ms')), 'launches': r.get('launches'), 'rows_per_launch': r.get('rows_per_launch'),
          'algorithmic_bytes_per_launch': r.get('algorithmic_bytes_per_launch'), 'timed_by': 'HIP events on the kernel\'s stream, inside the timed region'}
    for k in ('issue_frac', 'issue_frac_source', 'binds', 'invalid'):
        if r.get(k) is not None:
            ro[k] = r[k] if isinstance(r[k], str) else _r(r[k])
    if isinstance(r.get('alone'), dict) and 'avg_launch_ms' in r['alone']:
        ro['alone'] = {'avg_launch_ms': _r(r['alone']['avg_launch_ms']), 'frac': _r(r['alone'].get('frac')), 'rows_per_launch': r['alone'].get('rows_per_launch')}
    line['roofline'] = ro
    fs = out.get('filter_stage')
    if isinstance(fs, dict) and fs is not r and 'join_lb' in ro['kernel']:
        line['filter_stage'] = {'kernel': str(fs.get('kernel', '')).split(' (')[0], 'bound': 'mfma', 'issued_frac': _r(fs.get('issued', {}).get('frac')),
                                'avg_launch_ms': _r(fs.get('avg_launch_ms')), 'traffic_ratio': _r(fs.get('traffic_ratio'), 2),
                                'mfma_busy': fs.get('mfma_busy') if not isinstance(fs.get('mfma_busy'), dict) else
                                dict((k, _r(v, 3)) for k, v in fs['mfma_busy'].items())}
    c = out.get('cpu_baseline')
    if isinstance(c, dict):
        line['cpu_baseline'] = {'value': _r(c['value'], 2), 'unit': c['unit'], 'cores': c['cores'], 'kind': c['kind'], 'sample': c['sample'][:200],
                                'cpu_model': c.get('cpu_model'), 'host_cores': c.get('host_cores'),
                                'gpu_matches_cpu_path': c.get('gpu_matches_cpu_path'), 'gpu_matches_cpu_candidates': c.get('gpu_matches_cpu_candidates')}
        if 'all_cores' in c:
            line['cpu_baseline']['all_cores'] = {'value': _r(c['all_cores']['value'], 1), 'cores': c['all_cores']['cores']}
    for k in ('replicas', 'note'):
        if k in out:
            line[k] = out[k] if isinstance(out[k], str) else dict((a, _r(b, 3)) for a, b in out[k].items() if a != 'note')
    if 'host_ms_per_step' in out:
        line['host_ms_per_step'] = dict((k, _r(v, 3)) for k, v in out['host_ms_per_step'].items())
    if 'knn_lists' in out:
        line['knn_lists'] = dict((k, _r(v, 1)) for k, v in out['knn_lists'].items())
    line['stages_ms_per_step'] = dict((k, _r(v, 3)) for k, v in out.get('stages_ms_per_step', {}).items())
    line['roofline_check'] = out.get('roofline_check')
    line['summary'] = out.get('summary')
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--units', type=int, default=1048576)
    ap.add_argument('--frames', type=int, default=600)
    ap.add_argument('--candidates', type=int, default=100)
    ap.add_argument('--utts', type=int, default=32, help='utterances per step (batch)')
    ap.add_argument('--target-dim', type=int, default=61)
    ap.add_argument('--join-dim', type=int, default=302)
    ap.add_argument('--cpu-sample-frames', type=int, default=600,
                    help='frames of the CPU baseline sample (600 = one utterance of the workload, ~15 s on one core)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-greedy', action='store_true', help='skip the greedy configs (extra fields greedy_b1 / greedy_b3)')
    ap.add_argument('--no-cpu-all-cores', action='store_true', help='skip the all-cores leg of the CPU baseline')
    ap.add_argument('--no-variants', action='store_true', help='skip the B* legs on the non-compact databases (extra field noncompact)')
    ap.add_argument('--no-shapes', action='store_true', help='skip the legs on the other BASELINE shapes B2 / B4 / B5 (extra field shapes)')
    ap.add_argument('--viterbi-mode', type=int, default=2, choices=(0, 1, 2),
                    help='2: the engine default (batches: f32 matrix lower bounds + verified sparse exact recursion); '
                         '1: force that path; 0: dense exact float64 join costs')
    ap.add_argument('--opt', action='append', default=[], metavar='NAME=VALUE', help='engine option (snk_set_option), for experiments')
    ap.add_argument('--join-beta', type=float, default=None, help='margin of the predecessor sets (speed only)')
    ap.add_argument('--in-flight', type=int, default=0, choices=(0, 1, 2, 3),
                    help='N = 1: steps in flight; 2 submits step i+1 before collecting step i, so the tail of a step '
                         '(the per-utterance recursions of its last group, the copy of the results) runs beside the K-NN of the '
                         'next one -- how a tuning loop over a tune set drives the engine; 3 (the library has three workspaces) '
                         'submits step i+2 as well, so that the K-NN stream has work while the host waits for step i (a slow host; on a '
                         'fast one the deeper queue costs more than it gives); 0 (default): the better of 2 and 3, tried for ten steps '
                         'each after the warm-up, untimed; every step completes inside the timed region.  1: strictly one step at a '
                         'time (reported as extra field one_in_flight otherwise)')
    ap.add_argument('--resident-rows', action='store_true',
                    help='N = 1, experiments: `value` = the rate with the query rows left in HBM by two untimed priming submits '
                         '(default: `value` is the host -> host rate of SURVEY 8d -- the query rows are uploaded from and the paths '
                         'returned to host memory inside every timed step, as every caller in the package does; the resident-rows rate '
                         'is then the extra field `resident_rows`)')
    ap.add_argument('--upload-every-step', action='store_true', help='(the default since round 6; kept so that old command lines still run)')
    ap.add_argument('--detail-out', default=os.path.join('gpurun_out', 'bench_detail.json'), metavar='PATH',
                    help='where the full record goes (stage tables, every leg, the notes); the LAST stdout line is the compact '
                         'record (< 6 KB) the driver parses.  "" = do not write it')
    ap.add_argument('--exchange', choices=('library', 'torch'), default='library',
                    help="N > 1, sharded database: collectives inside libsnkhip.so (snk_comm_init: RCCL on the engine's stream; "
                         "default) or torch.distributed collectives between the device-pointer entry points (snickery_amd/dist.py)")
    ap.add_argument('--no-replicas-extra', action='store_true', help='N > 1: skip the extra pass that times the GPUs as independent replicas')
    ap.add_argument('--fixed-batch', action='store_true', help='N > 1: keep the batch at --utts (strong scaling)')
    ap.add_argument('--db-shards', type=int, default=0,
                    help='N > 1: shard the database over this many GPUs (a divisor of N; default N) and replicate '
                         'those shard groups N / db-shards times, each group taking its share of the utterances; '
                         '1 = independent replicas, no collective on the data path')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    N, Dt, Dj, T, K, U = args.units, args.target_dim, args.join_dim, args.frames, args.candidates, args.utts
    if not args.fixed_batch:
        U *= world                  # weak scaling: --utts utterances per GPU and step

    import torch
    import snickery_amd
    # SNK_BENCH_SHARE_GPU=1: functional test of the multi-rank path on a ONE-GPU box -- all ranks use
    # cuda:0 and the collectives run on gloo through host memory (RCCL refuses two ranks on one
    # device).  Never a measurement.
    share_gpu = os.environ.get('SNK_BENCH_SHARE_GPU') == '1'
    if share_gpu:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        if share_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
    wt = np.full(Dt, 0.8 * 0.5)                 # target_stream_weights * (1 - join_cost_weight)
    wj = np.full(Dj, 0.2 * 0.25)                # join_stream_weights * join_cost_weight
    utts = [synthetic_targets(F_unw, T, seed=1 + u) * wt for u in range(U)]
    frames_per_step = sum(u.shape[0] for u in utts)
    # the batch as the C ABI takes it (one contiguous matrix + row offsets), built once: a tuning loop
    # searches the same tune set every iteration.  On one GPU the timed steps search the rows where the two priming
    # submits left them, in HBM (--upload-every-step: uploaded in every step, the `with_upload` figure).
    batch = snickery_amd.QueryBatch(utts)

    cpu_ref = None
    if world == 1 and not args.no_cpu_baseline:
        # before the GPU is initialised: the all-cores figure forks its workers
        cpu_ref = cpu_baseline(F_unw, JC_unw, wt, wj, K, args.cpu_sample_frames, seed=1, all_cores=not args.no_cpu_all_cores)

    eng = snickery_amd.HipSearchEngine(local_rank)       # raises without libsnkhip.so / gfx950
    eng.set_option('viterbi_mode', args.viterbi_mode)
    if args.join_beta is not None:
        eng.set_option('join_beta', args.join_beta)
    for kv in args.opt:
        name, value = kv.split('=')
        eng.set_option(name, float(value))
    if world == 1:
        eng.upload_db(F_unw, JC_unw)
        eng.set_weights(wt, wj)
        n_local = N

        def step():
            return eng.knn_viterbi_batch(batch, K)
        pipeline = None
    else:
        from snickery_amd.dist import HipShardEngine, ShardedSearch, shard_bounds
        # ranks [q*S, (q+1)*S) form shard group q: the database is row-sharded over the S ranks of a
        # group, the groups are replicas of each other and split the batch (S = world: one group)
        S = args.db_shards if args.db_shards > 0 else world
        if world % S:
            raise SystemExit('--db-shards %d does not divide the number of GPUs %d' % (S, world))
        n_groups, my_group, sub_rank = world // S, rank // S, rank % S
        group = None
        if n_groups > 1 and S > 1:
            groups = [dist.new_group(ranks=list(range(q * S, (q + 1) * S))) for q in range(n_groups)]
            group = groups[my_group]
        u_lo, u_hi = shard_bounds(U, n_groups, my_group)
        my_utts = batch.subset(u_lo, u_hi)
        my_frames = int(sum(my_utts.lengths))
        lo, hi = shard_bounds(N, S, sub_rank)
        n_local = hi - lo
        if S == 1:
            eng.upload_db(F_unw, JC_unw)
            eng.set_weights(wt, wj)

            def step():
                return eng.knn_viterbi_batch(my_utts, K)
            # independent replicas: each rank keeps two of its own steps in flight
            pipeline = (lambda: eng.knn_viterbi_batch_submit(my_utts, K), eng.knn_viterbi_batch_collect)
        else:
            eng.upload_target_only(F_unw[lo:hi])
            eng.upload_join_only(JC_unw)
            eng.set_shard(lo, N)
            search = None
            if args.exchange == 'library':
                # collectives inside libsnkhip.so (RCCL on the engine's stream, no host synchronisation between
                # bounds, exchange and merge); every rank bounds its own share of the rows against a
                # replicated 1/16 sample of the whole database
                from snickery_amd.dist import LibraryShardedSearch, global_sample
                eng.upload_global_sample(global_sample(F_unw, 16))
                eng.set_weights(wt, wj)
                # the library's own RCCL communicator has only ever been opened with one rank on the one-GPU test boxes:
                # if it cannot be opened here on ANY rank, every rank takes the exchange through torch.distributed
                # (the same HIP kernels, the caller's communicator) and the JSON line says so
                err = ''
                try:
                    search = LibraryShardedSearch(eng, rank=sub_rank, world_size=S, group=group,
                                                  transport='gloo' if share_gpu and os.environ.get('SNK_BENCH_FORCE_RCCL') != '1' else 'rccl')
                except Exception as e:        # noqa: BLE001
                    err = str(e)
                flag = torch.tensor([0 if search is None else 1], dtype=torch.int32, device='cpu' if share_gpu else 'cuda')
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    sys.stderr.write('rank %d: library communicator not available (%s): exchange through torch.distributed\n' % (rank, err or 'another rank failed'))
                    if search is not None:
                        eng.comm_destroy()
                    args.exchange = 'torch (library communicator could not be opened)'
                    search = None
            if search is None:
                if args.exchange == 'torch':
                    eng.set_weights(wt, wj)
                search = ShardedSearch(HipShardEngine(eng, torch.device('cuda', local_rank)), rank=sub_rank, world_size=S,
                                       group=group)

            def step():
                return search.knn_viterbi_batch(my_utts, K)
            # collectives in the library: two sharded steps in flight (every rank issues the same sequence of submits and
            # collects); the torch-side exchange synchronises the host between its stages and stays one step at a time
            pipeline = (lambda: search.submit(my_utts, K), search.collect) if hasattr(search, 'submit') else None

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        paths, costs = step()
    # What is resident when the timed region starts: the unit DATABASE (weighted, operands built).  The query rows are the
    # step's input and cross the boundary the way every caller of the package hands them over -- host memory (page-locked) ->
    # HBM inside the step, paths back to host memory inside the step: SURVEY 8d's wall time, VERDICT r5 item 1.
    # --resident-rows: the rows where two untimed priming submits left them (Q == NULL, include/snk.h), for experiments.
    resident = world == 1 and args.in_flight != 1 and args.resident_rows
    if world == 1 and args.in_flight != 1:
        batch.pin()
        for _ in range(3):                  # every workspace of the pipeline primed (and holds the rows for the resident-rows pass)
            paths, costs = eng.knn_viterbi_batch_collect(eng.knn_viterbi_batch_submit(batch, K))

    host_ms = {'submit': 0.0, 'collect': 0.0, 'n': 0}

    def pipelined(steps, res, depth=None):
        # `depth` steps in flight: a step is collected when `depth` are pending (every step completes before this returns)
        depth = depth or (args.in_flight if args.in_flight >= 2 else 2)
        pending, out = [], None
        for _ in range(steps):
            h0 = time.perf_counter()
            pending.append(eng.knn_viterbi_batch_submit(batch, K, resident=res))
            h1 = time.perf_counter()
            if len(pending) >= depth:
                out = eng.knn_viterbi_batch_collect(pending.pop(0))
            host_ms['submit'] += (h1 - h0) * 1e3; host_ms['collect'] += (time.perf_counter() - h1) * 1e3; host_ms['n'] += 1
        while pending:
            out = eng.knn_viterbi_batch_collect(pending.pop(0))
        return out
    depth_probe = None
    if world == 1 and args.in_flight == 0:
        # how many steps to keep in flight is the caller's knob, and the better value depends on the HOST (two: the next submit
        # follows the wait for the step before the last -- a slow host lets the K-NN stream run dry; three: never dry, but the
        # deeper queue makes every submit slower): both tried here, untimed, the better one runs the timed region
        eng.set_option('timers', 2)
        depth_probe = {}
        for d in (2, 3, 2, 3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            pipelined(10, resident, depth=d)
            torch.cuda.synchronize()
            depth_probe[d] = max(depth_probe.get(d, 0.0), frames_per_step * 10 / (time.perf_counter() - t1))
        args.in_flight = 3 if depth_probe[3] > depth_probe[2] else 2
        host_ms.update({'submit': 0.0, 'collect': 0.0, 'n': 0})
    elif world > 1 and args.in_flight == 0:
        args.in_flight = 2
    if world == 1:
        # inside the timed region only the roofline kernel's stage is timed (HIP events on its stream); the stage table comes from
        # a second, untimed pass of the same loop with every stage timed: a timed stage is two timestamp events on its stream,
        # ~60 per step, 4-7 % of it (profiles/r06h_ab.log)
        eng.set_option('timers', 2)
    eng.reset_timers()
    sync()
    t0 = time.perf_counter()
    if world == 1 and args.in_flight >= 2:
        # steps in flight: step i+1 is submitted before step i is collected, so the tail of a step
        # (its last recursions, the copy of the results, the host-side hand-over) runs beside the K-NN
        # of the next one -- what a tuning loop over a tune set does.  All K steps complete inside
        # the timed region.
        paths, costs = pipelined(args.steps, resident)
    elif world > 1 and args.in_flight >= 2 and pipeline is not None:
        my_utts.pin()
        submit, collect = pipeline
        pending = None
        for _ in range(args.steps):
            ticket = submit()
            if pending is not None:
                paths, costs = collect(pending)
            pending = ticket
        paths, costs = collect(pending)
    else:
        for _ in range(args.steps):
            paths, costs = step()
    sync()
    elapsed = time.perf_counter() - t0
    host_main = dict(host_ms)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if share_gpu else 'cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    timers = eng.timers()
    staged = None
    if world == 1:
        # the stage pass: the same loop, every stage timed (an extra: its rate is `with_stage_timers`)
        eng.set_option('timers', 1)
        eng.reset_timers()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if args.in_flight >= 2:
            pipelined(args.steps, resident)
        else:
            for _ in range(args.steps):
                step()
        torch.cuda.synchronize()
        es = time.perf_counter() - t1
        staged = {'value': frames_per_step * args.steps / es, 'unit': 'frames/s', 'ms_per_step': es / args.steps * 1e3}
        timers_roof = timers
        timers = eng.timers()
        if 'join_lower_bounds' in timers_roof:
            timers['join_lower_bounds'] = timers_roof['join_lower_bounds']      # (the roofline kernel: as timed inside the timed region)
        eng.set_option('timers', 2)                                             # (the extra passes below run like the timed region)
    # counters the rooflines of the two kernels without one until round 5 are priced on, read before any
    # other pass adds to them: exact costs pass 3 took from the rows, list entries the re-rank read / gave exact distances
    roof_counts = {}
    knn_main = dict((k, eng.info(k)) for k in ('last_list_mean', 'last_list_max', 'tau_optimism_rank', 'tau_optimism_failures', 'tau_optimism_off', 'batch_redos', 'f16_fallbacks')) if world == 1 else {}
    starved = (eng.info('submits_starved'), eng.info('submits_pipelined'), eng.info('submits_starved_recent')) if world == 1 else None
    trip_main = dict((k, eng.info(k)) for k in ('prefilter_margin_rows', 'prefilter_min_margin', 'join_bound_violations', 'join_bound_min_margin')) if world == 1 else {}
    # N > 1, database sharded: the same GPUs as independent replicas (every GPU the whole database -- B* needs 3.5 GB of
    # 288 -- and its own 32 utterances, no collective), timed the same way: an extra field, never `value`.  Sharding is for
    # databases beyond one GPU's memory; for one that fits, this is what the exchange costs.
    replicas = None
    if world > 1 and S > 1 and not args.fixed_batch and not args.no_replicas_extra:
        t_rep, err = float('inf'), ''
        try:
            eng_r = snickery_amd.HipSearchEngine(local_rank)
            eng_r.set_option('viterbi_mode', args.viterbi_mode)
            eng_r.upload_db(F_unw, JC_unw)
            eng_r.set_weights(wt, wj)
            mine = batch.subset(*shard_bounds(U, world, rank))
            mine.pin()
            eng_r.knn_viterbi_batch(mine, K)
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            pending = None
            for _ in range(args.steps):
                tk = eng_r.knn_viterbi_batch_submit(mine, K)
                if pending is not None:
                    eng_r.knn_viterbi_batch_collect(pending)
                pending = tk
            eng_r.knn_viterbi_batch_collect(pending)
            torch.cuda.synchronize()
            t_rep = time.perf_counter() - t1
            eng_r.close()
        except Exception as e:        # noqa: BLE001 -- every rank still reaches the reduction below
            err = str(e)
        tt = torch.tensor([t_rep], dtype=torch.float64, device='cpu' if share_gpu else 'cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if np.isfinite(float(tt.item())):
            replicas = {'value': frames_per_step * args.steps / float(tt.item()), 'unit': 'frames/s',
                        'ms_per_step': float(tt.item()) / args.steps * 1e3,
                        'note': 'the same GPUs as %d independent replicas (whole database on every GPU, no collective), two steps in flight' % world}
        elif rank == 0:
            sys.stderr.write('replicas extra skipped: %s\n' % (err or 'a rank failed'))
    # the same pipeline in the OTHER input mode (an extra field, never `value`): rows resident in HBM, or uploaded every step
    other_mode = None
    if world == 1 and args.in_flight >= 2:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pu, cu = pipelined(args.steps, not resident)
        torch.cuda.synchronize()
        eu = time.perf_counter() - t1
        other_mode = {'value': frames_per_step * args.steps / eu, 'unit': 'frames/s', 'ms_per_step': eu / args.steps * 1e3,
                      'same_results': bool(all(np.array_equal(a, b) for a, b in zip(pu, paths)) and np.array_equal(cu, costs)),
                      'note': ('the query rows (%.1f MB per step) cross PCIe inside every timed step, from page-locked host memory' % (batch.Q.nbytes / 1e6)) if resident else
                              'the query rows searched where two untimed priming submits left them in HBM (Q == NULL): a mode no caller of the package uses'}
    depth2 = None
    if world == 1 and args.in_flight >= 2:
        # the same loop at the OTHER depth (two steps in flight: the depth of rounds 2-5), an extra field
        od = 2 if args.in_flight == 3 else 3
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pipelined(args.steps, resident, depth=od)
        torch.cuda.synchronize()
        e2d = time.perf_counter() - t1
        depth2 = {'steps_in_flight': od, 'value': frames_per_step * args.steps / e2d, 'unit': 'frames/s', 'ms_per_step': e2d / args.steps * 1e3}
    if world == 1:
        # the counting pass (untimed, two steps): the kernels add up what their rooflines are priced on only while option
        # roofline_counters is on -- thousands of workgroups adding to one address cost 0.3-0.4 ms per launch (profiles/r06_a)
        eng.set_option('roofline_counters', 1)
        eng.set_option('timers', 1)
        eng.reset_timers()
        for _ in range(2):
            eng.knn_viterbi_batch(batch, K)
        tmc = eng.timers()
        roof_counts = dict((k, eng.info(k)) for k in ('sparse_exact_costs', 'sparse_set_members', 'finalize_list_entries', 'finalize_reranked'))
        roof_counts['launches'] = {'join_exact_sparse': tmc.get('join_exact_sparse', (0, 0))[1], 'knn_finalize': tmc.get('knn_finalize', (0, 0))[1]}
        eng.set_option('roofline_counters', 0)
        eng.reset_timers()
    # a second, separately timed pass in the OTHER submission mode (an extra field, never `value`)
    two_in_flight = one_in_flight = None
    if world == 1 and args.in_flight == 1:
        batch.pin()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pending = None
        for _ in range(args.steps):
            ticket = eng.knn_viterbi_batch_submit(batch, K)
            if pending is not None:
                eng.knn_viterbi_batch_collect(pending)
            pending = ticket
        p2, c2 = eng.knn_viterbi_batch_collect(pending)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t1
        two_in_flight = {'value': frames_per_step * args.steps / e2, 'unit': 'frames/s', 'ms_per_step': e2 / args.steps * 1e3,
                         'same_results': bool(all(np.array_equal(a, b) for a, b in zip(p2, paths)) and np.array_equal(c2, costs)),
                         'note': 'step i+1 submitted before step i is collected (snk_knn_viterbi_batch_submit / _collect)'}
    elif world == 1:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            p2, c2 = step()
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t1
        one_in_flight = {'value': frames_per_step * args.steps / e2, 'unit': 'frames/s', 'ms_per_step': e2 / args.steps * 1e3,
                         'same_results': bool(all(np.array_equal(a, b) for a, b in zip(p2, paths)) and np.array_equal(c2, costs)),
                         'note': 'strictly one step at a time (snk_knn_viterbi_batch): the last group\'s per-utterance recursions are exposed'}
    if rank == 0:
        total_frames = frames_per_step * args.steps
        value = total_frames / elapsed
        ms, launches = timers['knn_filter']
        avg_ms = ms / max(launches, 1)
        # algorithmic flops: 2 * rows * N_local * Dt for the rows this rank swept; the
        # batch entry points group utterances, so one filter launch covers rows_per_launch rows
        rows_swept = (frames_per_step if world == 1 else my_frames) * args.steps
        rows_per_launch = rows_swept / max(launches, 1)
        flops = 2.0 * rows_per_launch * n_local * Dt
        # HBM bytes per launch from the committed PMC passes (separate --pmc runs, FETCH_SIZE
        # doubled as MI355X_MICROARCH.md prescribes for gfx950); only valid for the profiled shape
        # which sweep did the filtering: the f32 prefilter (default; results are made exact by the
        # float64 re-rank) or, when it had to fall back / was switched off, the f64 sweep
        f32_mode = eng.info('precision') == 1 and eng.info('f16_ready') == 1 and eng.info('f16_fallbacks') == 0
        bf16_mode = f32_mode and eng.info('prefilter_bf16_active') == 1
        peak = BF16_MFMA_PEAK_TFLOPS if bf16_mode else F32_MFMA_PEAK_TFLOPS if f32_mode else F64_MFMA_PEAK_TFLOPS
        terms = 4 if eng.info('prefilter') == 2 else 3
        two_pass = bf16_mode and eng.info('prefilter_two_pass') == 1
        coarse_now = two_pass and eng.info('filter_coarse') == 1
        kname = (('knn_coarse16b + knn_refine16b' if coarse_now else 'knn_balls16b + knn_refine16b') +
                 ' (the filter stage, v_mfma_f32_32x32x16_bf16: %s lists the (32 units x 32 rows) tile pairs that can hold a key under '
                 'the row thresholds, the refine pass takes the %d-term bf16-split keys of those pairs only; exact f64 re-rank in '
                 'knn_finalize)' % ('a one-term hi.hi sweep' if coarse_now else 'a bound from the tiles\' centres and radii', terms)
                 if two_pass else
                 'knn_sweep16b<filter> (v_mfma_f32_32x32x16_bf16 prefilter, every operand split into two bf16 pieces: '
                 '%d MFMA terms per product; exact f64 re-rank in knn_finalize)' % terms if bf16_mode else
                 'knn_sweep16<filter> (v_mfma_f32_32x32x2_f32 prefilter; exact f64 re-rank in knn_finalize)'
                 if f32_mode else 'knn_sweep<filter> (v_mfma_f64_16x16x4_f64)')
        traffic = None
        traffic_source = None
        achieved = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        out = {
            'metric': 'synthesised frames/sec, full-DB K=%d K-NN preselection + Viterbi' % K,
            'value': value, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong' if args.fixed_batch else 'weak',
            'vs_baseline': None, 'dtype': 'f64' if not f32_mode else 'f64 (bf16-split matrix prefilter + exact f64 re-rank)' if bf16_mode else 'f64 (f32 matrix prefilter + exact f64 re-rank)', 'data': 'synthetic',
            'xRT': (total_frames * FRAMESHIFT_MS / 1e3) / elapsed,
            'config': {'workload': 'B* synthetic magphase-60 (SURVEY 8d): |DB|=%d units, Dt=%d, Dj=%d, '
                                   'T=%d frames x %d utterances per step, K=%d, search_epsilon=0'
                                   % (N, Dt, Dj, T, U, K),
                       'units': N, 'target_dim': Dt, 'join_dim': Dj, 'frames': T, 'utts_per_step': U, 'utts_per_gpu': U // world,
                       'n_candidates': K, 'steps_in_flight': args.in_flight if world == 1 else (min(args.in_flight, 2) if pipeline is not None else 1),
                       'sharding': 'none' if world == 1 else (
                           '%d independent replicas' % world if S == 1 else
                           'db-rows/%d + all-to-all of local top-K' % S + (' x %d replica groups' % n_groups if n_groups > 1 else '')),
                       'exchange': None if world == 1 or S == 1 else args.exchange,
                       'hw_queues': os.environ.get('GPU_MAX_HW_QUEUES')},
            'value_includes_query_upload': not resident,
            'filter_stage': {'bound': 'mfma', 'kernel': kname,
                             'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                             'frac': achieved / peak, 'traffic': traffic, 'traffic_source': traffic_source,
                             'avg_launch_ms': avg_ms, 'launches': launches,
                             'rows_per_launch': rows_per_launch, 'flops_per_launch': flops,
                             'note': 'NOT the dominant kernel (see `roofline`).  achieved / frac price the ALGORITHMIC 2 N rows Dt flops -- the '
                                     'distance evaluations the reference\'s tree query stands for -- of a stage that skips most tile pairs; '
                                     'what the matrix pipe executes is `issued`, what it is busy `mfma_busy`'},
            'stages_ms_per_step': dict((k, v[0] / args.steps) for k, v in timers.items() if v[1]),
            'stage_launches_per_step': dict((k, v[1] / args.steps) for k, v in timers.items() if v[1]),
        }
        out['viterbi'] = {'mode': 'f32 matrix lower bounds + verified sparse exact recursion' if args.viterbi_mode else 'dense exact float64 join costs',
                          'cells_refined': eng.info('dense_cells'), 'steps_with_refinement': eng.info('dense_steps'),
                          'exact_costs_in_refinement': eng.info('dense_exact_costs')}
        for key, tname in (('us_per_step_exact_recursion', 'viterbi_sparse'), ('us_per_step_bounds_recursion', 'viterbi_lower_bound')):
            if tname in timers and timers[tname][1]:
                # latency chains: launch duration / the longest utterance's steps (the utterances of a launch run side by side)
                out['viterbi'][key] = timers[tname][0] / timers[tname][1] / max(T - 1, 1) * 1e3
        variant1 = eng.info('join_lb_variant') == 1
        if args.viterbi_mode and 'join_lower_bounds' in timers and timers['join_lower_bounds'][1]:
            # `roofline`: the whole-chip kernel with the largest total time in the round's rocprof stats
            # (profiles/r04_*_kernel_stats.csv) -- pass 1 of the sparse Viterbi path, lower bounds of all K x K join costs of
            # consecutive candidate rows.  It gathers the 2 K candidate rows of a step (2 (T-1) K Dj s bytes per
            # utterance, s = 4) and writes K x K bounds; its matrix work (3 bf16 MFMA terms) is a twentieth of the pipe's
            # rate at that byte rate: HBM-bound.  (The T-step recursions viterbi_sparse1_kernel / viterbi_lb_kernel are
            # latency chains on one or a few wavefronts per utterance: `viterbi.us_per_step_*`, no roofline claim.)
            jms, jl = timers['join_lower_bounds']
            jrows = rows_swept / max(jl, 1)
            jbytes = jrows * K * 2 * Dj * 4 + jrows * K * K * 4        # gathered E and S rows (each read once per row pair) + the bounds written
            javg = jms / max(jl, 1)
            jfl = 2.0 * jrows * K * K * Dj
            jtraffic, jsrc, jsrc_short, jref, jissue, jissue_src = None, None, None, None, None, None
            if variant1 and world == 1 and N == 1048576 and Dj == 302 and K == 100:
                for jname in ('r06_traffic_joinlb2.json', 'r05_traffic_joinlb2.json', 'r04_traffic_joinlb2.json'):
                    tj, fresh = profiled_counters(jname, 'joinlb2_kernels.hip')
                    # a counter file counts only when its rows per launch were READ FROM THE KERNEL TRACE of the pass that took the
                    # counters (grid.x of join_lb2_kernel = row pairs), never assumed (r05's file assumed 9 599 after a regrouping had
                    # made it 4 800: traffic wrong by 2 x)
                    if tj is None or not tj.get('rows_from_trace') or not tj.get('rows_per_launch'):
                        continue
                    per_row = tj['hbm_bytes_per_launch'] / tj['rows_per_launch']
                    if fresh:
                        jtraffic = per_row * jrows
                        jsrc = ('profiles/%s (separate --pmc passes of this kernel alone on %d row pairs per launch -- grid size read from the kernel '
                                'trace of the same pass; the kernel source unchanged since; scaled to the %d rows of this run\'s launches)' % (jname, tj['rows_per_launch'], jrows))
                        jsrc_short = 'profiles/' + jname
                        if tj.get('issue_frac') is not None:
                            jissue, jissue_src = tj['issue_frac'], 'profiles/%s: (4 SQ_INSTS_VALU + 8 MFMA) / SIMD cycles, the kernel alone' % jname
                    else:
                        jref = {'file': 'profiles/' + jname, 'hbm_bytes_per_launch': per_row * jrows,
                                'note': 'counters of an EARLIER build of this kernel (its source has changed since the pass): not this run\'s traffic'}
                    break
            out['roofline'] = {
                'bound': 'hbm',
                'kernel': ('join_lb2_kernel (joinlb2_kernels.hip: gather of the weighted float32 join rows of 2 K candidates per step, bf16-split '
                           'K x K x Dj product on v_mfma_f32_32x32x16_bf16, proven lower bounds written as float32)' if variant1 else
                           'join_lb_kernel (joinfast_kernels.hip: v_mfma_f32_16x16x4_f32, rows weighted in float64 per gather)'),
                'achieved': jbytes / (javg * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s', 'frac': jbytes / (javg * 1e-3) / 1e9 / 8000.0,
                'traffic': jtraffic, 'traffic_source': jsrc, 'traffic_source_short': jsrc_short, 'profiled_reference': jref,
                'issue_frac': jissue, 'issue_frac_source': jissue_src,
                'binds': 'vector-issue port (issue_frac), not HBM' if jissue is not None and jissue > 0.6 else None,
                'avg_launch_ms': javg, 'launches': jl, 'rows_per_launch': jrows, 'algorithmic_bytes_per_launch': jbytes,
                'flops_per_launch': jfl, 'mfma_tflops': jfl / (javg * 1e-3) / 1e12,
                'rule': 'the whole-chip kernel with the largest total time under rocprofv3 (profiles/r06_*_kernel_stats.csv; the largest of all is a '
                        'latency chain on 16 workgroups: `largest_total_time_kernel`); timed here with '
                        'HIP events on the stream it is launched on, inside the timed region, while the K-NN of the next group shares the chip',
                'note': 'algorithmic bytes per SURVEY 8d: every candidate row gathered once per row pair (2 K rows of Dj float32) + K^2 float32 bounds; '
                        'rows that consecutive steps share are served from L2, so `traffic` lies below them',
                'what_bounds_it': 'the vector-issue port, not HBM (DESIGN.md 4.3, ablations on this chip: 0.50 ms per launch as built, 0.34 with '
                                  'every gather replaced by a constant, 0.45 without its MFMAs; unchanged by an XCD-aware row order): 93 vector '
                                  'instructions + 12 MFMAs per k-block and wavefront -- `frac` is the share of a ceiling the kernel does not touch'}
            if 'viterbi_sparse' in timers and timers['viterbi_sparse'][1]:
                # the kernel with the largest TOTAL time in the stats is not a whole-chip one: said here, with its own figures
                vms, vl = timers['viterbi_sparse']
                out['roofline']['largest_total_time_kernel'] = {
                    'kernel': 'viterbi_sparse1_kernel (the exact recursion over T steps)', 'avg_launch_ms': vms / max(vl, 1), 'launches': vl,
                    'workgroups_per_launch': U // max(vl // max(args.steps, 1), 1) if vl else None, 'compute_units': 256,
                    'us_per_recursion_step': out['viterbi'].get('us_per_step_exact_recursion'),
                    'why_no_roofline': 'a chain of T dependent steps on ONE wavefront per utterance (16 workgroups per launch on 256 compute units): its time is '
                                       'T x the latency of a step (K candidates: minima over the kept predecessors, refinement of the step on the spot), it moves '
                                       '0.1 GB per launch and issues no matrix work; it runs beside the next group\'s K-NN and does not gate the step -- the '
                                       'whole-chip kernels do (sum of their stand-alone times 4.1 of the 4.6 ms step, DESIGN.md 4.3)'}
        else:
            out['roofline'] = dict(out['filter_stage'])
        if host_main['n']:
            # the host's side of a step: time inside submit (uploads and launches queued) and inside collect (mostly waiting)
            out['host_ms_per_step'] = {'submit': host_main['submit'] / host_main['n'], 'collect_wait': host_main['collect'] / host_main['n']}
            if starved:
                # submits (since the engine was created) that found everything of the batch before already run: the K-NN stream was idle
                out['host_ms_per_step'].update({'submits_that_found_the_stream_idle': starved[0], 'of_pipelined_submits': starved[1], 'recent_share': starved[2]})
        if knn_main:
            # the K-NN lists of the timed steps: their mean length per row, the rank of the sample minimum the thresholds came from
            # (0: the guaranteed K-th; j < K: optimistic, every row proven by the re-rank), groups redone with guaranteed thresholds
            out['knn_lists'] = knn_main
        if world == 1 and roof_counts:
            out['other_rooflines'] = other_rooflines(timers, roof_counts, rows_swept, K, Dt, Dj, T)
        if bf16_mode:
            # what the matrix pipe executes for those algorithmic flops: 64-column tiles, 4 bf16 terms per product
            dpad = (Dt + 3 + 63) // 64 * 64
            issued = terms * 2.0 * rows_per_launch * n_local * dpad
            if two_pass:
                # first pass: one product per (tile of 32 units, row) -- the centres -- or the hi.hi term of every unit;
                # second pass: the listed tile pairs (the most recent launch's count)
                pairs = eng.info('coarse_pairs')
                first = (2.0 * rows_per_launch * n_local * dpad) if coarse_now else (terms * 2.0 * rows_per_launch * (n_local / 32.0) * dpad)
                issued = first + terms * 2.0 * pairs * 32 * 32 * dpad
                out['filter_stage']['tile_pairs_listed'] = {'last_launch': pairs, 'of': (rows_per_launch / 32.0) * (n_local / 32.0),
                                                        'fraction': pairs / max((rows_per_launch / 32.0) * (n_local / 32.0), 1.0)}
            out['filter_stage']['issued'] = {'tflops': issued / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0,
                                         'frac': issued / (avg_ms * 1e-3) / 1e12 / peak if avg_ms > 0 else 0.0,
                                         'note': ('what the matrix pipe executes: the first pass plus %d bf16 MFMA terms (hi.hi + hi.lo + lo.hi%s) for the listed '
                                                  'tile pairs, on Dt padded to 64 columns; frac above prices the algorithmic 2 N rows Dt flops -- the '
                                                  'distance evaluations the reference\'s tree query stands for -- against the bf16 peak, so skipping '
                                                  'tiles raises it' if two_pass else
                                                  'float32-grade keys on the bf16 pipe cost %d bf16 MFMA terms per product (hi.hi + hi.lo + '
                                                  'lo.hi%s) on Dt padded to 64 columns; frac above prices only the algorithmic '
                                                  '2 N rows Dt flops against the bf16 peak') % (terms, ' + lo.lo' if terms == 4 else '')}
        if world == 1 and N == 1048576 and Dt == 61 and two_pass:
            for cname in ('r06_filter_counters.json', 'r05_filter_counters.json', 'r04_filter_counters.json'):
                cj, fresh = profiled_counters(cname, 'knn16_kernels.hip')
                if cj is None or int(cj.get('rows_per_launch', 0)) != int(round(rows_per_launch)):
                    continue
                vals = {'traffic': cj['hbm_bytes_per_launch'], 'traffic_ratio': cj['traffic_ratio'], 'mfma_busy': cj['mfma_busy']}
                if fresh:
                    out['filter_stage'].update(vals)
                    out['filter_stage']['traffic_source'] = 'profiles/%s (separate --pmc passes of these kernels and this shape, their source unchanged since)' % cname
                else:
                    vals.update({'file': 'profiles/' + cname, 'note': 'counters of an EARLIER build of these kernels: not this run\'s'})
                    out['filter_stage']['profiled_reference'] = vals
                break
        if world == 1 and bf16_mode:
            # tripwire of the bf16-split prefilter's key bound (untimed, after the timed region): the same step with the
            # float32-operand prefilter, whose bound is the analytical one of an f32 FMA chain, must select the same units;
            # and how close the exact K-th keys of the timed steps came to the filter thresholds (include/snk.h)
            margin_rows, min_margin = trip_main.get('prefilter_margin_rows', eng.info('prefilter_margin_rows')), trip_main.get('prefilter_min_margin', eng.info('prefilter_min_margin'))
            pre_was = eng.info('prefilter')
            eng.set_option('prefilter', 0)
            eng.set_weights(wt, wj)
            p0, c0 = eng.knn_viterbi_batch(batch, K)
            eng.set_option('prefilter', pre_was)
            eng.set_weights(wt, wj)
            out['prefilter_tripwire'] = {
                'gpu_matches_f32_prefilter': bool(all(np.array_equal(a, b) for a, b in zip(p0, paths)) and np.array_equal(c0, costs)),
                'prefilter_margin_rows': margin_rows, 'prefilter_min_margin': min_margin,
                'note': 'margin = (filter threshold - exact K-th key) / assumed key error eps, over every row of the timed steps: '
                        'the factor by which the true key errors could exceed eps before a row could lose a neighbour; '
                        'prefilter_margin_rows counts the rows under 2 (include/snk.h, DESIGN.md 4.1a, 6.1)'}
        if other_mode is not None:
            out['with_upload' if resident else 'resident_rows'] = other_mode
        if depth2 is not None:
            out['other_depth'] = depth2
        if depth_probe is not None:
            out['depth_probe_frames_per_s'] = dict((str(k), v) for k, v in depth_probe.items())
        if staged is not None:
            out['with_stage_timers'] = staged
        out['config']['inputs'] = ('database resident in HBM; query rows resident too (--resident-rows)' if resident else
                                   'database resident in HBM; query rows host -> HBM and paths HBM -> host inside every timed step')
        if two_in_flight is not None:
            out['two_in_flight'] = two_in_flight
        if one_in_flight is not None:
            out['one_in_flight'] = one_in_flight
        if replicas is not None:
            out['replicas'] = replicas
        if share_gpu:
            out['note'] = 'FUNCTIONAL TEST: all ranks share cuda:0, collectives on gloo through host memory; not a measurement'
        if cpu_ref is not None:
            base, ref = cpu_ref
            out['cpu_baseline'] = base
            # the same sample through the HIP path must select the same units
            gp, gc, gcand, gdist = eng.knn_viterbi(ref[4], K, return_candidates=True)
            out['cpu_baseline']['gpu_matches_cpu_path'] = bool(gp == ref[2])
            out['cpu_baseline']['gpu_matches_cpu_candidates'] = bool(np.array_equal(gcand, ref[0]))
        # the three tripwires of the probed MFMA accumulation bound, over the timed B* steps (before any other voice is uploaded)
        tripwires = dict(trip_main) if trip_main else {'prefilter_margin_rows': eng.info('prefilter_margin_rows'), 'prefilter_min_margin': eng.info('prefilter_min_margin'),
                                                        'join_bound_violations': eng.info('join_bound_violations'), 'join_bound_min_margin': eng.info('join_bound_min_margin')}
        if 'prefilter_tripwire' in out:
            tripwires['gpu_matches_f32_prefilter'] = out['prefilter_tripwire']['gpu_matches_f32_prefilter']
        if world == 1 and args.viterbi_mode and isinstance(out.get('roofline'), dict) and 'join_lb2' in str(out['roofline'].get('kernel', '')):
            # the roofline kernel ALONE, live: the Viterbi side of one group (half the batch) from resident candidates, nothing beside
            # it -- `frac` above is its duration inside the pipeline, where it shares the chip with the next group's filter passes
            # (since round 5 it starts behind that group's thresholds instead of starving its stage A: DESIGN.md 4.3)
            try:
                grp = utts[:max(1, U // 2)]
                cd = [eng.knn(u, K) for u in grp]
                eng.set_option('viterbi_mode', 1)
                eng.viterbi_batch([c for c, _ in cd], [d for _, d in cd])
                eng.reset_timers()
                for _ in range(3): eng.viterbi_batch([c for c, _ in cd], [d for _, d in cd])
                ta = eng.timers().get('join_lower_bounds')
                eng.set_option('viterbi_mode', args.viterbi_mode)
                if ta and ta[1]:
                    a_ms = ta[0] / ta[1]
                    rows_a = sum(len(u) for u in grp)
                    a_bytes = out['roofline']['algorithmic_bytes_per_launch'] * rows_a / max(out['roofline']['rows_per_launch'], 1.0)
                    out['roofline']['alone'] = {'avg_launch_ms': a_ms, 'launches': ta[1], 'rows_per_launch': rows_a,
                                                'achieved': a_bytes / (a_ms * 1e-3) / 1e9, 'frac': a_bytes / (a_ms * 1e-3) / 1e9 / 8000.0,
                                                'note': 'the same kernel on the same rows with nothing beside it, HIP events, this run'}
            except Exception as ex:          # (an extra: never the reason a bench line is missing)
                out['roofline']['alone'] = {'error': str(ex)[:200]}
        leg_steps = max(20, args.steps)                 # (a timed region's first and last steps run without neighbours: ~2 steps' worth per region -- ten-step legs read 8 % low)
        if world == 1 and not args.no_variants and N >= 65536:
            # the same workload on databases whose tiles are not compact balls: what the fallbacks of the ball
            # pass cost, each leg with the roofline of its dominant kernel
            legs = []
            for kind in ('permuted', 'speechlike'):
                Fv, JCv, tg = variant_database(kind, N, Dt, F_unw, JC_unw)
                legs.append(shape_leg(eng, 'B*', Fv, JCv, wt, wj, T, U, K, leg_steps, kind=kind, host_to_host=True,
                                      targets_from=F_unw if kind == 'permuted' else None, targets=tg, depth=max(args.in_flight, 2)))
                del Fv, JCv
            out['noncompact'] = legs
        if world == 1 and not args.no_shapes:
            # SURVEY 8d's other Viterbi shapes, one GPU: B2 (slt full, K 50), B4 (Nick, K 200), B5 (halfphone width, Dt 184) -- the
            # latter also with its units in random order (halfphone databases are where consecutive units are least alike)
            del F_unw, JC_unw
            shapes = []
            for sname, sN, sDt, sDj, sT, sU, sK, kinds in (('B2', 700000, 61, 302, 600, 32, 50, ('compact',)),
                                                            ('B4 (1 GPU)', 1500000, 61, 302, 600, 32, 200, ('compact',)),
                                                            ('B5', 1300000, 184, 151, 120, 64, 100, ('compact', 'permuted'))):
                Fs, JCs = synthetic_db(sN, sDt, sDj, seed=0)
                wts, wjs = np.full(sDt, 0.4), np.full(sDj, 0.05)
                for kind in kinds:
                    Fk, JCk = (Fs, JCs) if kind == 'compact' else variant_database(kind, sN, sDt, Fs, JCs)[:2]
                    shapes.append(shape_leg(eng, sname, Fk, JCk, wts, wjs, sT, sU, sK, leg_steps, kind=kind, host_to_host=True,
                                            targets_from=Fs if kind == 'permuted' else None, depth=max(args.in_flight, 2)))
                del Fs, JCs
            out['shapes'] = shapes
        if world == 1 and not args.no_greedy:
            eng.close()                     # free the last database before the greedy voices are built
            eng = None
            out['extra'] = greedy_extra(local_rank)
            for g in ('greedy_b1', 'greedy_b3'):
                tripwires[g + '_bound_violations'] = out['extra'][g].get('bound_violations')
                tripwires[g + '_bound_max_used'] = out['extra'][g].get('bound_max_used')
        # ---- the compact record: the LAST stdout line, what the driver parses (r05's 22.9 KB line did not parse).
        # Everything else -- stage tables, every leg in full, the notes -- goes to --detail-out. ----
        def brief(leg):
            r = leg.get('roofline', {})
            return {'frames_per_s': round(leg.get('host_to_host_frames_per_s', leg['frames_per_s'])), 'ms_per_step': round(leg.get('host_to_host_ms_per_step', leg['ms_per_step']), 3),
                    'rows': 'host -> host' if 'host_to_host_frames_per_s' in leg else 'resident', 'kernel': str(r.get('kernel'))[:40],
                    'bound': r.get('bound'), 'frac': None if r.get('frac') is None else round(r['frac'], 3),
                    'redos': leg['batch_redos'] + leg['prefilter_fallbacks']}
        summary = {'tripwires': tripwires}
        if other_mode is not None:
            summary['resident_rows_frames_per_s' if not resident else 'host_to_host_frames_per_s'] = round(other_mode['value'])
        if depth2 is not None:
            summary['%d_steps_in_flight_frames_per_s' % depth2['steps_in_flight']] = round(depth2['value'])
        for leg in out.get('noncompact', []):
            summary['B* ' + leg['database']] = brief(leg)
        for leg in out.get('shapes', []):
            summary[leg['shape'] + ('' if leg['database'] == 'compact' else ' ' + leg['database'])] = brief(leg)
        if 'extra' in out:
            summary['greedy_us_per_step'] = {'B1': round(out['extra']['greedy_b1']['us_per_step'], 2), 'B3': round(out['extra']['greedy_b3']['us_per_step'], 2),
                                             'B3_frac_hbm': round(out['extra']['greedy_b3']['roofline'].get('frac', 0.0), 3)}
        if 'other_rooflines' in out:
            summary['other_rooflines_frac_hbm'] = dict((k, _r(v['frac'], 3)) for k, v in out['other_rooflines'].items())
        out['summary'] = summary
        bad = check_rooflines(out)
        out['roofline_check'] = 'every frac within [0, 1]' if not bad else {'invalid': bad}
        line = compact_line(out)
        if args.detail_out:
            try:
                dpath = args.detail_out if os.path.isabs(args.detail_out) else os.path.join(ROOT, args.detail_out)
                os.makedirs(os.path.dirname(dpath) or '.', exist_ok=True)
                with open(dpath, 'w') as f:
                    json.dump(out, f, indent=1)
                line['detail'] = args.detail_out
            except OSError as ex:
                line['detail'] = 'not written: %s' % str(ex)[:80]
        text = json.dumps(line)
        assert len(text) < 6144 and '\n' not in text, len(text)
        sys.stdout.flush()
        print(text, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if eng is not None:
        eng.close()


if __name__ == '__main__':
    main()
