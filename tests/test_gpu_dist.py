"""The sharded search path (snickery_amd/dist.py) with the real HIP engine: two ranks share the one
GPU of the test box (gloo collectives through host memory; RCCL refuses two ranks on one device).
Everything but the RCCL transport itself is the production path: shard upload, shard-local top-K
into device buffers, exchange, merge, join costs, Viterbi on the owner, result gather."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import snk_oracle as o
    import snickery_amd
    from snickery_amd.dist import HipShardEngine, ShardedSearch, shard_bounds
    N, Dt, Dj, K = 30001, 61, 40, 25
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=5)
    wt = np.full(Dt, 0.4)
    wj = np.full(Dj, 0.05)
    F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
    lo, hi = shard_bounds(N, world, rank)
    torch.cuda.set_device(0)
    eng = snickery_amd.HipSearchEngine(0)
    eng.upload_target_only(F_unw[lo:hi])
    eng.upload_join_only(JC_unw)
    eng.set_shard(lo, N)
    eng.set_weights(wt, wj)
    search = ShardedSearch(HipShardEngine(eng, torch.device('cuda', 0)))
    utts = [o.synthetic_targets(F_unw, T, seed=s) * wt for s, T in [(1, 40), (2, 25), (3, 33)]]
    ok = True
    cand, d = search.knn(utts[0], K)
    oc, od = o.knn_bruteforce(F, utts[0], K)
    ok = ok and np.array_equal(cand, oc) and np.array_equal(d, od)
    for batch in (utts, utts[1:2]):                      # second batch: rank 1 owns nothing
        paths, costs = search.knn_viterbi_batch(batch, K)
        for u, U in enumerate(batch):
            c, dd = o.knn_bruteforce(F, U, K)
            p, cost = o.viterbi(c, dd, E, S)
            ok = ok and list(paths[u]) == p and costs[u] == cost
    out[rank] = bool(ok)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_search_two_ranks_one_gpu():
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] is True and out[1] is True


def _lib_worker(rank, world, port, out, use_sample):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import snk_oracle as o
    import snickery_amd
    from snickery_amd.dist import HipShardEngine, LibraryShardedSearch, ShardedSearch, global_sample, shard_bounds
    from snickery_amd.engine import shard_plan
    N, Dt, Dj, K = 60001, 61, 40, 25
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=5)
    wt = np.full(Dt, 0.4)
    wj = np.full(Dj, 0.05)
    F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
    lo, hi = shard_bounds(N, world, rank)
    ok = (lo, hi) == shard_plan(N, world, rank)
    torch.cuda.set_device(0)
    eng = snickery_amd.HipSearchEngine(0)
    eng.upload_target_only(F_unw[lo:hi])
    eng.upload_join_only(JC_unw)
    eng.set_shard(lo, N)
    if use_sample:
        eng.upload_global_sample(global_sample(F_unw, 4))
    eng.set_weights(wt, wj)
    lib_search = LibraryShardedSearch(eng, transport='gloo')
    utts = [o.synthetic_targets(F_unw, T, seed=s) * wt for s, T in [(1, 40), (2, 25), (3, 33), (4, 2), (5, 61)]]
    for batch in (utts, utts[1:2]):                      # second batch: rank 1 owns nothing
        paths, costs = lib_search.knn_viterbi_batch(batch, K)
        ok = ok and len(paths) == len(batch)
        for u, U in enumerate(batch):
            c, dd = o.knn_bruteforce(F, U, K)
            p, cost = o.viterbi(c, dd, E, S)
            ok = ok and list(paths[u]) == p and costs[u] == cost
    # arguments every rank can judge alone are refused BEFORE the first collective is queued: nobody waits for a rank
    # that left, and the next step runs as if nothing had happened
    for bad_K in (500, 0):
        try:
            lib_search.knn_viterbi_batch(utts, bad_K)
            ok = False
        except snickery_amd.SnkError:
            pass
    try:
        lib_search.knn_viterbi_batch([u[:, :30] for u in utts], K)           # wrong number of columns
        ok = False
    except snickery_amd.SnkError:
        pass
    paths, costs = lib_search.knn_viterbi_batch(utts[:2], K)
    for u, U in enumerate(utts[:2]):
        c, dd = o.knn_bruteforce(F, U, K)
        p, cost = o.viterbi(c, dd, E, S)
        ok = ok and list(paths[u]) == p and costs[u] == cost
    # the torch-side path (dist.py collectives) gives the same answer
    eng.comm_destroy()
    tpaths, tcosts = ShardedSearch(HipShardEngine(eng, torch.device('cuda', 0))).knn_viterbi_batch(utts, K)
    lpaths, lcosts = LibraryShardedSearch(eng, transport='gloo').knn_viterbi_batch(utts, K)
    ok = ok and all(np.array_equal(a, b) for a, b in zip(tpaths, lpaths)) and np.array_equal(tcosts, lcosts)
    # two steps in flight: step i + 1 is submitted before step i is collected (its Viterbi side then runs beside the
    # K-NN of step i + 1); different batches in flight, every rank issues the same sequence
    lib2 = LibraryShardedSearch(eng, transport='gloo')
    batches = [utts, utts[2:], utts[:3], utts]
    want = [lib2.knn_viterbi_batch(b, K) for b in batches]
    got, pending = [], None
    for b in batches:
        tk = lib2.submit(b, K)
        if pending is not None:
            got.append(lib2.collect(pending))
        pending = tk
    got.append(lib2.collect(pending))
    for (gp, gc), (wp, wc) in zip(got, want):
        ok = ok and all(np.array_equal(a, b) for a, b in zip(gp, wp)) and np.array_equal(gc, wc)
    try:                                                  # a third step while two are in flight is refused
        t1, t2 = lib2.submit(utts, K), lib2.submit(utts, K)
        try:
            lib2.submit(utts, K)
            ok = False
        except snickery_amd.SnkError:
            pass
        lib2.collect(t1); lib2.collect(t2)
    except snickery_amd.SnkError:
        ok = False
    # a list overflow on the fast path of ONE rank makes ALL ranks redo the step together (exact sweep)
    if rank == 0:
        eng.set_option('list_capacity', 64)
        eng.set_option('sample_fraction', 1.0 / 64)
    before = eng.info('batch_redos')
    paths, costs = LibraryShardedSearch(eng, transport='gloo').knn_viterbi_batch(utts, K)
    ok = ok and all(np.array_equal(a, b) for a, b in zip(paths, lpaths)) and np.array_equal(costs, lcosts)
    redone = eng.info('batch_redos') - before
    flags = [None] * world
    dist.all_gather_object(flags, int(redone))
    ok = ok and len(set(flags)) == 1                     # every rank took the same decision
    # ... also with two steps in flight: the unsafe step is redone at its collect, the other one is untouched
    lib3 = LibraryShardedSearch(eng, transport='gloo')
    before = eng.info('batch_redos')
    ta, tb = lib3.submit(utts, K), lib3.submit(utts[:3], K)
    pa, ca = lib3.collect(ta)
    pb, cb = lib3.collect(tb)
    ok = ok and all(np.array_equal(a, b) for a, b in zip(pa, lpaths)) and np.array_equal(ca, lcosts)
    ok = ok and all(np.array_equal(a, b) for a, b in zip(pb, lpaths[:3])) and np.array_equal(cb, lcosts[:3])
    flags2 = [None] * world
    dist.all_gather_object(flags2, int(eng.info('batch_redos') - before))
    ok = ok and len(set(flags2)) == 1 and (flags[0] == 0 or flags2[0] >= 1)
    out[rank] = bool(ok)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('use_sample', [False, True])
def test_library_collectives_two_ranks_one_gpu(use_sample):
    """snk_sharded_knn_viterbi_batch (collectives inside libsnkhip.so) over the caller-provided transport
    (gloo through host memory): two ranks, shards of one GPU, ragged batch, a rank owning nothing, with and
    without the replicated global sample; equal to the oracle and to the torch-side path of dist.py."""
    port = 27500 + (os.getpid() % 2000) + int(use_sample)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_lib_worker, args=(2, port, out, use_sample), nprocs=2, join=True)
    assert out[0] is True and out[1] is True


def test_library_collectives_three_ranks_one_gpu():
    """The same with three ranks (uneven shards of 60 001 units, 5 utterances over 3 owners: blocks of 2 / 2 / 1, and a
    one-utterance batch that leaves two ranks without any): more of the offset arithmetic of the exchanges."""
    port = 21500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_lib_worker, args=(3, port, out, True), nprocs=3, join=True)
    assert out[0] is True and out[1] is True and out[2] is True


def test_library_collectives_eight_ranks_one_gpu():
    """G = 8, the size of the node the sharded search is for and never ran on: eight ranks on this one GPU over
    the gloo transport -- the 8-peer compacted all-to-all-v, the merge tree of depth 3, uneven shards (60 001 units over 8 ranks),
    5 utterances over 8 owners (three own none), the collective redo, two steps in flight; equal to the oracle on every rank."""
    port = 19500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_lib_worker, args=(8, port, out, True), nprocs=8, join=True)
    assert all(out[r] is True for r in range(8))


def _greedy_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import snk_oracle as o
    import snickery_amd
    from snickery_amd.dist import gloo_transport
    torch.cuda.set_device(0)
    ok = True
    # (units, Dt, Dj, multiepoch, last_frame_as_target, join_split_mode): several tiles per rank, a last tile that is not full,
    # fewer tiles than ranks (130 units: three tiles of 64 windows at me = 1 -- two at me = 6 -- for three ranks)
    for N, Dt, Dj, me, lfat, mode in ((5000, 61, 151, 6, False, 0), (3001, 61, 40, 1, False, 1), (130, 20, 16, 6, True, 0), (4097, 90, 33, 3, False, 0)):
        F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=me + Dj)
        wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
        F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
        # a few exact duplicates of units in different ranks' shares: ties of the distance across ranks (lowest window wins)
        if N >= 3000:
            F_unw[N - 200:N - 190] = F_unw[100:110]; JC_unw[N - 200:N - 189] = JC_unw[100:111]
            F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
        eng = snickery_amd.HipSearchEngine(0)
        eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
        eng.set_greedy_layout(me, lfat, mode)
        eng.comm_init_transport(world, rank, gloo_transport(world))
        pr, cr, Fwin = o.greedy_layout(F, E, S, me, lfat, mode)
        for T, start in ((63, -1), (40, 17), (me, -1), (max(me - 1, 1), -1)):
            U = o.synthetic_targets(F_unw, T, seed=9 + T) * wt
            if N >= 3000 and T == 63: U[:me * 3] = (F_unw[100:100 + me * 3] * wt)          # the duplicated units' own rows
            path, d = eng.sharded_greedy(U, start_state=start, return_distances=True)
            op, od = o.greedy_search(pr, cr, Fwin, o.greedy_queries(U, me, lfat), start_state=start)
            sp, sd = eng.greedy(U, start_state=start, return_distances=True)
            ok = ok and path == op and np.array_equal(d, od) and path == sp and np.array_equal(d, sd)
        eng.close()
    # ---- a rank-local failure must refuse the call on EVERY rank, before any step's collective: the last rank has
    # no greedy layout; then the ranks disagree about start_state; then a good call still works on the same communicator ----
    N, Dt, Dj, me = 3000, 61, 40, 3
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=77)
    wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
    eng = snickery_amd.HipSearchEngine(0)
    eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
    if rank != world - 1:
        eng.set_greedy_layout(me, False, 0)
    eng.comm_init_transport(world, rank, gloo_transport(world))
    U = o.synthetic_targets(F_unw, 30, seed=5) * wt
    try:
        eng.sharded_greedy(U)
        ok = False
    except snickery_amd.SnkError as e:
        ok = ok and ('refused on every rank' in str(e))
    eng.set_greedy_layout(me, False, 0)
    try:
        eng.sharded_greedy(U, start_state=5 if rank == 0 else 6)
        ok = False
    except snickery_amd.SnkError as e:
        ok = ok and ('disagree' in str(e))
    F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
    pr, cr, Fwin = o.greedy_layout(F, E, S, me, False, 0)
    ok = ok and eng.sharded_greedy(U) == o.greedy_search(pr, cr, Fwin, o.greedy_queries(U, me, False))[0]
    eng.close()
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_greedy_ranks_on_one_gpu(world):
    """snk_sharded_greedy: every step's scan split over the ranks (each scans the windows of its tiles), one all-gather of the
    ranks' winners per step; the path and the distances equal the oracle's and snk_greedy's bit for bit."""
    port = 29500 + ((os.getpid() + 7 * world) % 2000)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_greedy_worker, args=(world, port, out), nprocs=world, join=True)
    assert all(out[r] is True for r in range(world))


def test_library_rccl_communicator_single_rank():
    """The RCCL transport itself, as far as one GPU allows: ncclCommInitRank with one rank, then the
    sharded entry point (its collectives degenerate to copies) against the unsharded search."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import snk_oracle as o
    import snickery_amd
    N, Dt, Dj, K = 40000, 61, 40, 20
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=8)
    wt, wj = np.full(Dt, 0.4), np.full(Dj, 0.05)
    eng = snickery_amd.HipSearchEngine(0)
    eng.upload_db(F_unw, JC_unw)
    eng.set_shard(0, N)
    eng.set_weights(wt, wj)
    utts = [o.synthetic_targets(F_unw, T, seed=s) * wt for s, T in [(1, 40), (2, 25)]]
    ref_paths, ref_costs = eng.knn_viterbi_batch(utts, K)
    uid = eng.comm_unique_id()
    assert len(uid) == 128
    eng.comm_init(1, 0, uid)
    paths, costs = eng.sharded_knn_viterbi_batch(utts, K)
    assert all(np.array_equal(a, b) for a, b in zip(paths, ref_paths)) and np.array_equal(costs, ref_costs)
    eng.comm_destroy()
    with pytest.raises(snickery_amd.SnkError):
        eng.sharded_knn_viterbi_batch(utts, K)           # no communicator
    eng.close()


def _child_report(r):
    """What a failed child run said: the lines that are not the launcher's boilerplate (the runtime's own last words --
    'Memory access fault ...', 'terminate called ...', a Python traceback of bench.py -- come long before the launcher's
    summary), then the tail."""
    lines = r.stderr.splitlines()
    keep = [l for l in lines if not l.startswith(('E1', 'W1', 'I1')) and 'torch/distributed' not in l and 'amdgpu.ids' not in l]
    return 'returncode %s\n--- stderr without launcher lines (first 80) ---\n%s\n--- stderr tail ---\n%s' % (
        r.returncode, '\n'.join(keep[:80]), r.stderr[-1500:])


def _run_bench_child(cmd, env):
    """bench.py under torch.distributed.run with the ranks SHARING the one GPU of the test box (never a production
    layout).  A child that dies -- by a signal or otherwise -- FAILS the test, with what it said (_child_report) in the
    message.  (Round 3 re-ran a child killed by a signal once; that hid exactly the class of failure these tests exist to
    show.  SNK_TEST_RETRY_SIGNALLED_CHILD=1 restores the retry for triage loops; the driver does not set it, and a retried
    run is still reported on stderr.)"""
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if (r.returncode != 0 and os.environ.get('SNK_TEST_RETRY_SIGNALLED_CHILD') == '1'
            and any(t in r.stderr for t in ('Signal 6', 'SIGABRT', 'Signal 11', 'SIGSEGV', 'Memory access fault'))):
        os.write(2, ('[snk-test] multi-rank child died with a signal; its report, then ONE retry (triage mode):\n%s\n' % _child_report(r)).encode())
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    return r


@pytest.mark.parametrize('nproc,shards,sharding,extra', [
    (2, 0, 'db-rows/2 + all-to-all of local top-K', []),
    (4, 2, 'db-rows/2 + all-to-all of local top-K x 2 replica groups', []),
    (2, 1, '2 independent replicas', []),
    # G = 8 (no 8-GPU box has ever been available): 8-peer all-to-all-v, the depth-3 merge tree, shards of
    # uneven size (40 003 units over 8 ranks), 40 utterances over 8 owners ...
    (8, 0, 'db-rows/8 + all-to-all of local top-K', ['--units', '40003']),
    # ... and 5 utterances over 8 owners: three ranks own none (they still filter every row against their shard)
    (8, 0, 'db-rows/8 + all-to-all of local top-K', ['--units', '40003', '--fixed-batch']),
    (8, 4, 'db-rows/4 + all-to-all of local top-K x 2 replica groups', [])])
def test_bench_multi_rank_one_gpu(nproc, shards, sharding, extra):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run), ranks sharing the GPU over the gloo transport:
    database sharded over all ranks (default), shard groups x replica groups, independent replicas -- at 2, 4 and 8 ranks.
    The last stdout line is the compact record (< 6 KB)."""
    env = dict(os.environ, SNK_BENCH_SHARE_GPU='1')
    port = 23500 + (os.getpid() % 2000) + nproc * 3 + shards + 7 * len(extra)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
           '--gpus', str(nproc), '--steps', '2', '--warmup', '1', '--units', '40000', '--frames', '60', '--utts', '5',
           '--candidates', '20', '--no-cpu-baseline', '--db-shards', str(shards), '--detail-out', ''] + extra
    r = _run_bench_child(cmd, env)
    assert r.returncode == 0, _child_report(r)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and r.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 6144
    js = json.loads(lines[0])
    fixed = '--fixed-batch' in extra
    assert js['n_gpus'] == nproc and js['value'] > 0 and js['scaling'] == ('strong' if fixed else 'weak')
    assert js['config']['utts_per_step'] == (5 if fixed else 5 * nproc) and js['config']['sharding'] == sharding
    assert js['config']['units'] == int(extra[1]) if extra else 40000
    assert 'roofline' in js and 'FUNCTIONAL TEST' in js['note']


def test_bench_falls_back_to_the_callers_communicator_when_rccl_cannot_open():
    """Two ranks on ONE device: RCCL refuses the communicator (duplicate GPU).  Every rank must then agree to run the
    exchange through torch.distributed instead of one rank dying and the other waiting -- the situation the driver's
    multi-GPU run would be in if the in-library communicator failed for a reason the one-GPU boxes cannot show."""
    env = dict(os.environ, SNK_BENCH_SHARE_GPU='1', SNK_BENCH_FORCE_RCCL='1')
    port = 25500 + (os.getpid() % 2000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
           '--gpus', '2', '--steps', '2', '--warmup', '1', '--units', '40000', '--frames', '60', '--utts', '5',
           '--candidates', '20', '--no-cpu-baseline', '--detail-out', '']
    r = _run_bench_child(cmd, env)
    assert r.returncode == 0, _child_report(r)
    js = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert js['value'] > 0 and 'could not be opened' in js['config']['exchange']
    assert 'library communicator not available' in r.stderr
