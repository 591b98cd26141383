"""
CPU ORACLE (test infrastructure only) -- numpy restatement of the Snickery
unit-selection search path.

This file is the CHECKER.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product
(``snickery_amd``) never imports anything under ``oracle/`` and has no CPU
fallback.

Every function cites the reference file:line it restates (paths relative to the
reference checkout, i.e. ``script/...``).

Parity status
-------------
* weighting / greedy layout / greedy search / acoustic, monophone-restricted and
  quinphone K-NN / join costs / per-stream path scores: PINNED against outputs
  of the (lib2to3-converted) reference itself, captured by
  ``tools/make_golden.py`` into ``tests/golden/*.npz``.
* Viterbi (``viterbi_search``): the reference delegates the arithmetic to
  OpenFST 1.5.4 (``pywrapfst``: compose + shortestpath, float32 tropical
  weights, ``fst_functions_wrapped.py:368,389``) which is not vendored and not
  installable here.  The two lattice BUILDERS are pinned: their arc text is
  recorded from the reference's own functions (a recording stand-in for
  ``openfst.Compiler``) and ``fst_arc_lists`` below emits the same arcs.  The DP
  restates the published shortest-path semantics and equals (i) an independent
  product search of T o J over that recorded text
  (``fst_shortest_path_bruteforce``) and (ii) exhaustive path enumeration on
  tiny trellises.  Index parity versus the OpenFST binary's own compose /
  shortestpath arithmetic is "parity unpinned".

Canonical floating-point order
------------------------------
All squared Euclidean distances are accumulated in float64, column by column
(c = 0..D-1), as ``acc = acc + (a_c - b_c) * (a_c - b_c)`` with separately
rounded subtract, multiply and add (no FMA contraction).  The C oracle
(``snk_oracle.c``, built with -ffp-contract=off) and the HIP kernels use exactly
this order, so distances are bit-identical across the three; against the
reference's own numpy/scipy values (pairwise / tree-internal summation order)
the stated tolerance is rtol 1e-12.

Tie rule: wherever the reference's result depends on an implementation-defined
order among EXACTLY equal distances (cKDTree traversal, OpenFST state
discovery) the oracle and the product use: smaller distance first, then
smaller unit id (K-NN, greedy); lowest predecessor slot k', then lowest final
slot k (Viterbi).
"""
import numpy as np

VERY_BIG_WEIGHT_VALUE = 1000000000000000.0   # const.py:3
TARGET_REP_WIDTHS = {'onepoint': 1, 'twopoint': 2, 'threepoint': 3,
                     'epoch': 1, 'sample': 1}  # const.py:17


# --------------------------------------------------------------------------
# a2: weighting  (synth_simple.py:234-274, synth_halfphone.py:682-737,
#                 speech_manip.py:209-213)
# --------------------------------------------------------------------------
def stream_weight_vector(weights, stream_list, datadims, repetitions=1,
                         double_for_epoch_join=False, duration_weight=None):
    """Per-stream weights -> per-column weight vector.

    synth_simple.py:238-243 / :260-267 ; synth_halfphone.py:686-695 (epoch DB
    from train_halfphone doubles the join vector) and :716-728 (target
    repetitions, optional duration column).
    """
    assert len(weights) == len(stream_list)
    vec = []
    for i, stream in enumerate(stream_list):
        vec.extend([weights[i]] * datadims[stream])
    if double_for_epoch_join:
        vec = vec + vec
    vec = vec * repetitions
    if duration_weight is not None:
        vec.append(duration_weight)
    return np.array(vec, dtype=np.float64)


def weight(speech, weight_vec):
    """speech_manip.py:209-213 -- broadcast multiply; f32 data * f64 weights -> f64."""
    weight_vec = np.array(weight_vec).reshape((1, -1))
    return speech * weight_vec


def apply_jcw(target_stream_weights, join_stream_weights, join_cost_weight):
    """APPLY_JCW_ON_TOP: synth_simple.py:129-131."""
    tw = np.array(target_stream_weights) * (1.0 - join_cost_weight)
    jw = np.array(join_stream_weights) * join_cost_weight
    return tw, jw


def weighted_db(F_unw, JC_unw, wt, wj):
    """set_target_weights / set_join_weights -> (F, E, S), all float64.

    E = unit_end_data = JCw[1:], S = unit_start_data = JCw[:-1]
    (synth_simple.py:247-251).
    """
    F = weight(F_unw, wt).astype(np.float64)
    JCw = weight(JC_unw, wj).astype(np.float64)
    return F, JCw[1:, :], JCw[:-1, :]


# --------------------------------------------------------------------------
# a3: greedy layout (synth_simple.py:190-225 ; synth_halfphone.py:539-596)
# --------------------------------------------------------------------------
def greedy_layout(F, E, S, multiepoch=1, last_frame_as_target=False,
                  join_split_mode=0):
    """Returns (prev_join_rep S', current_join_rep E', Fwin).

    join_split_mode 0: synth_simple (S' = unit_start_data, E' = unit_end_data,
    :194-195).  join_split_mode 1: synth_halfphone epoch DB, S'/E' are the
    first/second half of the unit_start_data columns (:552-553; py2 ``n/2``).
    """
    if join_split_mode == 0:
        prev_rep, cur_rep = S, E
    else:
        n = S.shape[1]
        prev_rep, cur_rep = S[:, :n // 2], S[:, n // 2:]
    me = int(multiepoch)
    Fwin = F
    if me > 1:
        overlap = me - 1
        m, n = F.shape
        # segment_axis(F, me, overlap=me-1, axis=0).reshape(m-overlap, n*me)
        Fwin = np.hstack([F[e:m - overlap + e, :] for e in range(me)])
        if last_frame_as_target:
            Fwin = np.hstack([Fwin[:, :n], Fwin[:, -n:]])
        cur_rep = cur_rep[overlap:, :]
        prev_rep = prev_rep[:-overlap, :]
    return prev_rep, cur_rep, Fwin


def greedy_queries(unit_features, multiepoch=1, last_frame_as_target=False):
    """synth_simple.py:473-483: segment_axis(overlap=0, end='cut') + reshape."""
    me = int(multiepoch)
    U = np.asarray(unit_features, dtype=np.float64)
    if me > 1:
        m, n = U.shape
        steps = m // me               # py2 integer division, tail frames dropped
        U = U[:steps * me, :].reshape(steps, n * me)
        if last_frame_as_target:
            U = np.hstack([U[:, :n], U[:, -n:]])
    return U


def sqdist_rows(A, b):
    """Canonical-order squared distances of every row of A to vector b."""
    acc = np.zeros(A.shape[0], dtype=np.float64)
    for c in range(A.shape[1]):
        d = A[:, c] - b[c]
        acc = acc + d * d
    return acc


# --------------------------------------------------------------------------
# a6: greedy_joint_search (synth_simple.py:458-503 == synth_halfphone.py:1900-1945)
# --------------------------------------------------------------------------
def greedy_search(prev_rep, cur_rep, Fwin, Q, start_state=-1):
    """Exact (search_epsilon=0) greedy joint search.

    d2(i) = ||prev - S'[i]||^2 + ||Q[s] - Fwin[i]||^2  (the squared norm of the
    concatenated vector the reference's joint tree holds, :224 + :488-490),
    i* = argmin (lowest index on exact ties), prev <- E'[i*] (:501).
    Returns (path, dists) -- dists are the Euclidean (sqrt) values the tree
    query would return (:490).
    """
    n_join = cur_rep.shape[1]
    if start_state < 0:
        prev = np.zeros((n_join,), dtype=np.float64)
    else:
        prev = prev_rep[start_state, :]
    path, dists = [], []
    for q in Q:
        d2 = sqdist_rows(prev_rep, prev) + sqdist_rows(Fwin, q)
        ix = int(np.argmin(d2))       # first occurrence == lowest id
        path.append(ix)
        dists.append(float(np.sqrt(d2[ix])))
        prev = cur_rep[ix, :]
    return path, np.array(dists)


def greedy_search_ckdtree(prev_rep, cur_rep, Fwin, Q, start_state=-1, eps=0.0):
    """The reference's own formulation: scipy cKDTree over hstack[S', Fwin]
    (synth_simple.py:224-229, :487-501).  Used as the timed CPU baseline and to
    cross-check greedy_search at eps=0."""
    import scipy.spatial
    combined = np.hstack([prev_rep, Fwin])
    tree = scipy.spatial.cKDTree(combined, leafsize=100, balanced_tree=False)
    n_join = cur_rep.shape[1]
    prev = np.zeros((n_join,)) if start_state < 0 else prev_rep[start_state, :]
    path, dists = [], []
    for q in Q:
        both = np.concatenate([prev, q]).reshape((1, -1))
        d, i = tree.query(both, k=1, eps=eps)
        ix = int(np.asarray(i).flatten()[0])
        path.append(ix)
        dists.append(float(np.asarray(d).flatten()[0]))
        prev = cur_rep[ix, :]
    return path, np.array(dists)


# --------------------------------------------------------------------------
# a7/a8: preselection (synth_halfphone.py:1359-1396)
# --------------------------------------------------------------------------
def knn_bruteforce(F, U, K, chunk=8192):
    """Exact K-NN of every row of U in F: what ``cKDTree(F).query(U, k=K)``
    returns (synth_halfphone.py:1364): candidates (T,K) int64 ascending by
    Euclidean distance, distances (T,K) float64 (not squared).
    Ordering: (distance, unit id).  K > N pads with id -1 / VERY_BIG (the
    convention of preselect_units_monophone_then_acoustic, :1378-1379).
    """
    F = np.asarray(F, dtype=np.float64)
    U = np.asarray(U, dtype=np.float64)
    T, N = U.shape[0], F.shape[0]
    cand = np.full((T, K), -1, dtype=np.int64)
    dist = np.full((T, K), VERY_BIG_WEIGHT_VALUE, dtype=np.float64)
    for t in range(T):
        d2 = np.empty(N, dtype=np.float64)
        for s in range(0, N, chunk):
            d2[s:s + chunk] = sqdist_rows(F[s:s + chunk], U[t])
        k = min(K, N)
        if k < N:
            part = np.argpartition(d2, k - 1)[:k]
            kth = d2[part].max()
            sel = np.nonzero(d2 <= kth)[0]          # include all ties at the cut
        else:
            sel = np.arange(N)
        order = np.lexsort((sel, d2[sel]))[:k]      # by distance, then id
        ids = sel[order]
        cand[t, :k] = ids
        dist[t, :k] = np.sqrt(d2[ids])
    return cand, dist


def knn_ckdtree(F, U, K, workers=1):
    """The reference's formulation (synth_halfphone.py:379,1364)."""
    import scipy.spatial
    tree = scipy.spatial.cKDTree(F, leafsize=100, compact_nodes=False,
                                 balanced_tree=False)
    d, i = tree.query(U, k=K, workers=workers)
    return np.asarray(i, dtype=np.int64).reshape(U.shape[0], K), \
        np.asarray(d).reshape(U.shape[0], K)


def knn_by_class(F, U, K, unit_class, query_class):
    """preselect_units_monophone_then_acoustic (synth_halfphone.py:1369-1396):
    K-NN restricted to DB units of the query's class; local ids mapped back to
    global ids; short classes padded with -1 / VERY_BIG_WEIGHT_VALUE."""
    T = U.shape[0]
    cand = np.full((T, K), -1, dtype=np.int64)
    dist = np.full((T, K), VERY_BIG_WEIGHT_VALUE, dtype=np.float64)
    unit_class = np.asarray(unit_class)
    for t in range(T):
        members = np.nonzero(unit_class == query_class[t])[0]
        assert members.size > 0, 'unseen class %s' % (query_class[t],)
        c, d = knn_bruteforce(F[members], U[t:t + 1], min(K, members.size))
        cand[t, :c.shape[1]] = members[c[0]]
        dist[t, :c.shape[1]] = d[0]
    return cand, dist


# --------------------------------------------------------------------------
# a9: quinphone preselection (synth_halfphone.py:1305-1354, label_manip.py:16-32)
# --------------------------------------------------------------------------
def break_quinphone(quinphone, delimiter='/'):
    """label_manip.py:16-32: (mono, diphone, triphone, quinphone); diphone direction by _L/_R."""
    q = quinphone.split(delimiter)
    assert len(q) == 5
    mono = q[2]
    tri = delimiter.join(q[1:4])
    if mono.endswith('_L'):
        di = delimiter.join(q[1:3])
    elif mono.endswith('_R'):
        di = delimiter.join(q[2:4])
    else:
        raise SystemExit('efvaedvsdv')
    return (mono, di, tri, quinphone)


def build_unit_index(train_unit_names):
    """synth_halfphone.py:281-292."""
    unit_index = {}
    for i, quinphone in enumerate(train_unit_names):
        for form in break_quinphone(quinphone):
            unit_index.setdefault(form, []).append(i)
    return unit_index


def candidate_distances(F, U, cand):
    """dists of synth_halfphone.py:1343-1349: row -1 indexes the LAST unit (numpy semantics)."""
    cand = np.asarray(cand, dtype=np.int64)
    out = np.empty(cand.shape, dtype=np.float64)
    for t in range(cand.shape[0]):
        out[t] = np.sqrt(sqdist_rows(F[cand[t]], U[t]))
    return out


def preselect_units_quinphone(unit_index, F, U, unit_names, K):
    """quinphone, then triphone, diphone, monophone matches in database order, no
    de-duplication, stop at K; nothing found -> [1]; pad with -1."""
    candidates = []
    for quinphone in unit_names:
        cur = []
        mono, di, tri, quin = break_quinphone(quinphone)
        for form in [quin, tri, di, mono]:
            for unit in unit_index.get(form, []):
                cur.append(unit)
                if len(cur) == K:
                    break
            if len(cur) == K:
                break
        if len(cur) == 0:
            cur = [1]
        cur += [-1] * (K - len(cur))
        candidates.append(cur)
    candidates = np.array(candidates, dtype=np.int64)
    return candidates, candidate_distances(F, U, candidates)


# --------------------------------------------------------------------------
# a10: join cost (synth_halfphone.py:2942-2951 inside :3206-3322)
# --------------------------------------------------------------------------
def valid_mask(cand, n_units):
    """Units usable in the join lattice: id != -1 and mini=1 <= id < maxi=N-1
    (synth_halfphone.py:3238-3240,3262-3268)."""
    cand = np.asarray(cand)
    return (cand >= 1) & (cand < n_units - 1)


def join_cost_pairs(E, S, first, second):
    """get_natural_distance_vectorised(first, second, order=1):
    ||E[first] - S[second]||_2 in canonical order."""
    first = np.asarray(first, dtype=np.int64)
    second = np.asarray(second, dtype=np.int64)
    acc = np.zeros(first.shape[0], dtype=np.float64)
    for c in range(E.shape[1]):
        d = E[first, c] - S[second, c]
        acc = acc + d * d
    return np.sqrt(acc)


def join_cost_cache(E, S, cand):
    """The reference's de-duplicated {(first, second): cost} dict
    (synth_halfphone.py:3251-3301)."""
    n_units = E.shape[0]
    ok = valid_mask(cand, n_units)
    seen = {}
    first_list, second_list = [], []
    T = cand.shape[0]
    for t in range(T - 1):
        a = cand[t][ok[t]]
        b = cand[t + 1][ok[t + 1]]
        for f in a:
            for s in b:
                key = (int(f), int(s))
                if key in seen:
                    continue
                seen[key] = True
                first_list.append(key[0])
                second_list.append(key[1])
    d = join_cost_pairs(E, S, first_list, second_list)
    return dict(((f, s), w) for f, s, w in zip(first_list, second_list, d))


def join_cost_dense(E, S, cand):
    """(T-1, K, K) tensor: J[t, a, b] = c(cand[t,a], cand[t+1,b]); +inf where
    either unit is not usable."""
    T, K = cand.shape
    n_units = E.shape[0]
    ok = valid_mask(cand, n_units)
    J = np.full((max(T - 1, 0), K, K), np.inf, dtype=np.float64)
    safe = np.where(ok, cand, 1)
    for t in range(T - 1):
        fa = np.repeat(safe[t], K)
        sb = np.tile(safe[t + 1], K)
        d = join_cost_pairs(E, S, fa, sb).reshape(K, K)
        m = ok[t][:, None] & ok[t + 1][None, :]
        J[t] = np.where(m, d, np.inf)
    return J


# --------------------------------------------------------------------------
# a11-a14: Viterbi == shortest path through T o J
#   (fst_functions_wrapped.py:28-58,172-217,285-408 ; synth_halfphone.py:1399-1436)
# --------------------------------------------------------------------------
def viterbi(cand, tdist, E, S, mode='f64'):
    """delta_0[k] = tdist[0,k];
    delta_t[k] = tdist[t,k] + min_{k' valid}(delta_{t-1}[k'] + c(cand[t-1,k'], cand[t,k]))
    answer = back-trace from argmin_{k valid} delta_{T-1}[k]  (SURVEY 9.2).

    mode 'f64'   : float64 arithmetic (parity target of the HIP path).
    mode 'fst32' : emulates OpenFST's float32 tropical weights: arc weights are
                   rounded to f32 when the text FSTs are compiled, composed arc
                   weight = f32(tdist + c), path weight accumulated in f32.
    Returns (path_unit_ids list[int], cost float).  No valid path -> ([], inf)
    (T < 2 is a documented edge case: the reference's J has no states then).
    """
    cand = np.asarray(cand, dtype=np.int64)
    tdist = np.asarray(tdist, dtype=np.float64)
    T, K = cand.shape
    n_units = E.shape[0]
    if T < 2:
        return [], np.inf
    ok = valid_mask(cand, n_units)
    J = join_cost_dense(E, S, cand)
    if mode == 'fst32':
        return _viterbi_fst32(cand, tdist.astype(np.float32), J.astype(np.float32), ok)
    inf = np.inf
    delta = np.where(ok[0], tdist[0], inf)
    back = np.zeros((T, K), dtype=np.int64)
    for t in range(1, T):
        tot = delta[:, None] + J[t - 1]
        tot = np.where(ok[t - 1][:, None], tot, inf)
        bp = np.argmin(tot, axis=0)            # first minimum == lowest k'
        best = tot[bp, np.arange(K)]
        back[t] = bp
        delta = np.where(ok[t], tdist[t] + best, inf)
    k = int(np.argmin(delta))                  # lowest final slot on ties
    if not np.isfinite(delta[k]):
        return [], np.inf
    cost = float(delta[k])
    slots = [k]
    for t in range(T - 1, 0, -1):
        k = int(back[t, k])
        slots.append(k)
    slots.reverse()
    return [int(cand[t, s]) for t, s in enumerate(slots)], cost


def _viterbi_fst32(cand, tdist32, J32, ok):
    """float32 chain as OpenFST would accumulate it over T o J."""
    f = np.float32
    T, K = cand.shape
    inf = f(np.inf)
    acc = np.where(ok[0], f(0.0), inf).astype(f)       # free epsilon entry (:195-196)
    back = np.zeros((T, K), dtype=np.int64)
    for t in range(1, T):
        arcw = (tdist32[t - 1][:, None] + J32[t - 1]).astype(f)
        tot = (acc[:, None] + arcw).astype(f)
        tot = np.where(ok[t - 1][:, None], tot, inf)
        bp = np.argmin(tot, axis=0)
        back[t] = bp
        acc = np.where(ok[t], tot[bp, np.arange(K)], inf).astype(f)
    final = np.where(ok[T - 1], (acc + tdist32[T - 1]).astype(f), inf)
    k = int(np.argmin(final))
    if not np.isfinite(final[k]):
        return [], np.inf
    cost = float(final[k])
    slots = [k]
    for t in range(T - 1, 0, -1):
        k = int(back[t, k])
        slots.append(k)
    slots.reverse()
    return [int(cand[t, s]) for t, s in enumerate(slots)], cost


def path_cost(cand_path, tdist_path, E, S):
    """Objective value of a unit-id path: sum tdist + sum join (f64)."""
    p = np.asarray(cand_path, dtype=np.int64)
    j = join_cost_pairs(E, S, p[:-1], p[1:]) if len(p) > 1 else np.zeros(0)
    return float(np.sum(tdist_path) + np.sum(j))


def viterbi_enumerate(cand, tdist, E, S):
    """Exhaustive minimum over all K^T slot sequences (tiny cases only)."""
    import itertools
    cand = np.asarray(cand, dtype=np.int64)
    T, K = cand.shape
    ok = valid_mask(cand, E.shape[0])
    best, best_cost = None, np.inf
    for slots in itertools.product(range(K), repeat=T):
        if not all(ok[t, s] for t, s in enumerate(slots)):
            continue
        ids = [cand[t, s] for t, s in enumerate(slots)]
        c = sum(tdist[t, s] for t, s in enumerate(slots))
        c += float(np.sum(join_cost_pairs(E, S, ids[:-1], ids[1:])))
        if c < best_cost:
            best, best_cost = [int(i) for i in ids], c
    return (best or []), best_cost


def fst_arc_lists(cand, tdist, cost_cache):
    """The exact arc lists the reference hands to the OpenFST compiler.

    T: make_target_sausage_lattice (fst_functions_wrapped.py:28-58)
    J: cost_cache_to_compiled_fst   (fst_functions_wrapped.py:172-217)
    Returned as python tuples (src, dst, ilabel, olabel, weight).
    """
    T_arcs = []
    frames, cands = np.shape(tdist)
    start = 0
    end = 0
    for i in range(frames):
        end = start + 1
        for j in range(cands):
            ix = int(cand[i, j])
            if ix == -1:
                continue
            T_arcs.append((start, end, ix + 1, ix + 1, float(tdist[i, j])))
        start = end
    T_final = end
    frames_set = sorted(set([k for pair in cost_cache.keys() for k in pair]))
    frame2state = dict(zip(frames_set, range(1, len(frames_set) + 1)))
    J_arcs = []
    for fr in frames_set:
        J_arcs.append((0, frame2state[fr], 0, 0, 0.0))
    for (fro, to), w in cost_cache.items():
        J_arcs.append((frame2state[fro], frame2state[to], fro + 1, fro + 1, float(w)))
    sink = len(frames_set) + 1
    for fr in frames_set:
        J_arcs.append((frame2state[fr], sink, fr + 1, fr + 1, 0.0))
    return (T_arcs, T_final), (J_arcs, sink)


def fst_shortest_path_bruteforce(T_fst, J_fst):
    """Independent tropical-semiring product search over T o J (epsilon on J's
    input side), Dijkstra on (t_state, j_state).  Returns (unit ids, cost)."""
    import heapq
    (T_arcs, T_final), (J_arcs, J_final) = T_fst, J_fst
    t_out, j_out, j_eps = {}, {}, {}
    for a in T_arcs:
        t_out.setdefault(a[0], []).append(a)
    for a in J_arcs:
        if a[2] == 0:
            j_eps.setdefault(a[0], []).append(a)
        else:
            j_out.setdefault((a[0], a[2]), []).append(a)
    start = (0, 0)
    heap = [(0.0, 0, start, None, None)]
    done = {}
    cnt = 1
    while heap:
        cost, _, st, par, lab = heapq.heappop(heap)
        if st in done:
            continue
        done[st] = (par, lab, cost)
        if st == (T_final, J_final):
            out, cur = [], st
            while done[cur][0] is not None:
                if done[cur][1]:
                    out.append(done[cur][1] - 1)
                cur = done[cur][0]
            return out[::-1], cost
        ts, js = st
        for a in j_eps.get(js, []):
            heapq.heappush(heap, (cost + a[4], cnt, (ts, a[1]), st, 0)); cnt += 1
        for ta in t_out.get(ts, []):
            for ja in j_out.get((js, ta[3]), []):
                heapq.heappush(heap, (cost + ta[4] + ja[4], cnt, (ta[1], ja[1]), st, ta[3]))
                cnt += 1
    return [], np.inf


# --------------------------------------------------------------------------
# a15: per-stream scores (synth_halfphone.py:1964-1981, 2977-3008)
# --------------------------------------------------------------------------
def aggregate_squared_errors_by_stream(sq_errs, stream_list, datadims, repetitions=1):
    """synth_halfphone.py:2977-3008 (epoch / one-repetition case): per-stream
    row sums of squared errors -> (T, nstreams)."""
    out = []
    start = 0
    for _ in range(repetitions):
        cols = []
        for stream in stream_list:
            w = datadims[stream]
            cols.append(sq_errs[:, start:start + w].sum(axis=1))
            start += w
        out.append(np.vstack(cols).T)
    return sum(out)


def target_scores(F, U, path):
    """get_target_scores_per_stream core: (F[path] - U)**2 (:1964-1969)."""
    return (F[np.asarray(path, dtype=np.int64), :] - U[:len(path)]) ** 2


def join_scores_viterbi(E, S, path):
    """(E[p[:-1]] - S[p[1:]])**2 (synth_halfphone.py:1976-1979)."""
    p = np.asarray(path, dtype=np.int64)
    return (E[p[:-1], :] - S[p[1:], :]) ** 2


def join_scores_greedy(prev_rep, cur_rep, path):
    """(prev_join_rep[p[1:]] - current_join_rep[p[:-1]])**2 (:1973-1975)."""
    p = np.asarray(path, dtype=np.int64)
    return (prev_rep[p[1:], :] - cur_rep[p[:-1], :]) ** 2


# --------------------------------------------------------------------------
# synthetic workload generator shared by tests and bench
# --------------------------------------------------------------------------
def synthetic_db(N, Dt, Dj, seed=0):
    rng = np.random.RandomState(seed)
    F = np.cumsum(rng.randn(N, Dt), axis=0)
    F = (F / F.std()).astype(np.float32)
    JC = np.cumsum(rng.randn(N + 1, Dj), axis=0)
    JC = (JC / JC.std()).astype(np.float32)
    return F, JC


def synthetic_targets(F_unw, T, seed=1, noise=0.3):
    rng = np.random.RandomState(seed)
    N, Dt = F_unw.shape
    s = rng.randint(0, max(N - T, 1))
    return F_unw[s:s + T].astype(np.float64) + noise * rng.randn(min(T, N - s), Dt)
