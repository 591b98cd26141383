"""Pins the CPU oracle (oracle/snk_oracle.py) against outputs of the reference
itself (tests/golden/reference_mini.npz, produced by tools/make_golden.py from
the lib2to3-converted reference running in the build container)."""
import itertools
import numpy as np
import snk_oracle as o


def test_weight_vectors_and_weighted_rows(golden, mini_voice):
    assert np.array_equal(mini_voice['wt'], golden['target_weight_vector'])
    assert np.array_equal(mini_voice['wj'], golden['join_weight_vector'])
    F, E, S = mini_voice['F'], mini_voice['E'], mini_voice['S']
    assert F.dtype == np.float64 and E.dtype == np.float64
    assert np.array_equal(F[[0, 7, -1]], golden['train_unit_features_w_rows'])
    assert np.array_equal(E[[0, 5, -1]], golden['unit_end_data_rows'])
    assert np.array_equal(S[[0, 5, -1]], golden['unit_start_data_rows'])


def test_greedy_layout_shape(golden, mini_voice):
    pr, cr, Fwin = o.greedy_layout(mini_voice['F'], mini_voice['E'], mini_voice['S'], 6)
    assert tuple(golden['greedy_me6_tree_shape']) == (Fwin.shape[0], pr.shape[1] + Fwin.shape[1])
    assert bool(golden['greedy_me6_tree_dtype_is_f64'])
    pr, cr, Fwin = o.greedy_layout(mini_voice['F'], mini_voice['E'], mini_voice['S'], 1)
    assert tuple(golden['greedy_me1_tree_shape']) == (Fwin.shape[0], pr.shape[1] + Fwin.shape[1])


def test_greedy_paths_match_reference(golden, mini_voice):
    for me in (6, 1):
        pr, cr, Fwin = o.greedy_layout(mini_voice['F'], mini_voice['E'], mini_voice['S'], me)
        for utt in (0, 1):
            U = golden['greedy_me%d_utt%d_unit_features' % (me, utt)]
            Q = o.greedy_queries(U, me)
            path, _ = o.greedy_search(pr, cr, Fwin, Q)
            ref = golden['greedy_me%d_utt%d_path' % (me, utt)]
            assert len(path) == U.shape[0] // me
            assert np.array_equal(np.array(path), ref), (me, utt)


def test_greedy_natural_path_known_answer(golden, mini_voice):
    """resynth_training_chunk's assert (synth_simple.py:909-928): consecutive
    training rows as targets, start_state=start => consecutive ids."""
    pr, cr, Fwin = o.greedy_layout(mini_voice['F'], mini_voice['E'], mini_voice['S'], 1)
    start = int(golden['greedy_me1_natural_start'])
    ref = golden['greedy_me1_natural_path']
    assert np.array_equal(ref, np.arange(start, start + len(ref)))
    path, d = o.greedy_search(pr, cr, Fwin, mini_voice['F'][start:start + len(ref)], start_state=start)
    assert np.array_equal(np.array(path), ref)
    assert np.all(d == 0.0)


def test_greedy_ckdtree_formulation_agrees(golden, mini_voice):
    pr, cr, Fwin = o.greedy_layout(mini_voice['F'], mini_voice['E'], mini_voice['S'], 6)
    Q = o.greedy_queries(golden['greedy_me6_utt1_unit_features'], 6)
    p1, d1 = o.greedy_search(pr, cr, Fwin, Q)
    p2, d2 = o.greedy_search_ckdtree(pr, cr, Fwin, Q)
    assert p1 == p2
    np.testing.assert_allclose(d1, d2, rtol=1e-12)


def test_knn_matches_reference(golden, mini_voice):
    K = int(golden['knn_K'])
    cand, dist = o.knn_bruteforce(mini_voice['F'], golden['knn_queries'], K)
    assert np.array_equal(cand, golden['knn_candidates'])
    np.testing.assert_allclose(dist, golden['knn_distances'], rtol=1e-12)
    assert cand.dtype == np.int64 and dist.dtype == np.float64


def test_join_cache_matches_reference(golden, mini_voice):
    cand = golden['join_candidates']
    cache = o.join_cost_cache(mini_voice['E'], mini_voice['S'], cand)
    keys = np.array(sorted(cache.keys()), dtype=np.int64)
    assert np.array_equal(keys, golden['join_cache_keys'])
    vals = np.array([cache[tuple(k)] for k in keys])
    np.testing.assert_allclose(vals, golden['join_cache_values'], rtol=1e-12, atol=0)
    # naturally adjacent units join at exactly zero cost, in the reference too
    nat = [i for i, (a, b) in enumerate(keys) if b == a + 1]
    assert len(nat) > 0
    assert np.all(golden['join_cache_values'][nat] == 0.0)
    assert np.all(vals[nat] == 0.0)
    # units 0, N-1 and -1 never appear (synth_halfphone.py:3238-3268)
    N = mini_voice['F'].shape[0]
    assert keys.min() >= 1 and keys.max() < N - 1


def test_join_dense_consistent_with_cache(golden, mini_voice):
    cand = golden['join_candidates']
    J = o.join_cost_dense(mini_voice['E'], mini_voice['S'], cand)
    cache = dict(zip(map(tuple, golden['join_cache_keys']), golden['join_cache_values']))
    ok = o.valid_mask(cand, mini_voice['F'].shape[0])
    n = 0
    for t in range(cand.shape[0] - 1):
        for a in range(cand.shape[1]):
            for b in range(cand.shape[1]):
                if ok[t, a] and ok[t + 1, b]:
                    ref = cache[(int(cand[t, a]), int(cand[t + 1, b]))]
                    assert abs(J[t, a, b] - ref) <= 1e-12 * max(ref, 1e-300)
                    n += 1
                else:
                    assert np.isinf(J[t, a, b])
    assert n > 1000


def test_viterbi_equals_fst_product_search_on_reference_arcs(golden, mini_voice):
    """T and J arc lists exactly as fst_functions_wrapped.py:28-58,172-217 emit
    them, built from the reference-captured candidates/distances/cost_cache;
    independent tropical product search == DP."""
    E, S = mini_voice['E'], mini_voice['S']
    cand_all, dist_all = golden['join_candidates'], golden['knn_distances']
    cache_all = dict(zip(map(tuple, golden['join_cache_keys'].tolist()), golden['join_cache_values']))
    for t0, T, K in [(0, 3, 3), (2, 4, 3), (3, 5, 2), (4, 3, 4), (10, 6, 2)]:
        cand = cand_all[t0:t0 + T, :K]
        dist = dist_all[t0:t0 + T, :K]
        cache = {}
        ok = o.valid_mask(cand, E.shape[0])
        for t in range(T - 1):
            for a, b in itertools.product(range(K), range(K)):
                if ok[t, a] and ok[t + 1, b]:
                    key = (int(cand[t, a]), int(cand[t + 1, b]))
                    cache[key] = cache_all[key]
        p_fst, c_fst = o.fst_shortest_path_bruteforce(*o.fst_arc_lists(cand, dist, cache))
        p_dp, c_dp = o.viterbi(cand, dist, E, S)
        p_en, c_en = o.viterbi_enumerate(cand, dist, E, S)
        assert p_dp == p_en == p_fst, (t0, T, K)
        assert abs(c_dp - c_en) <= 1e-12 * c_en and abs(c_dp - c_fst) <= 1e-9 * c_en


def test_viterbi_invariants_full_fixture(golden, mini_voice):
    E, S = mini_voice['E'], mini_voice['S']
    cand, dist = golden['join_candidates'], golden['knn_distances']
    path, cost = o.viterbi(cand, dist, E, S)
    N = E.shape[0]
    assert len(path) == cand.shape[0]
    assert all(1 <= u < N - 1 for u in path)
    tpath = [dist[t, list(cand[t]).index(u)] for t, u in enumerate(path)]
    assert abs(o.path_cost(path, tpath, E, S) - cost) <= 1e-12 * cost
    # never worse than the best-per-column path
    ok = o.valid_mask(cand, N)
    greedy = [int(cand[t][ok[t]][0]) for t in range(cand.shape[0])]
    gt = [dist[t][ok[t]][0] for t in range(cand.shape[0])]
    assert cost <= o.path_cost(greedy, gt, E, S) + 1e-12
    p32, c32 = o.viterbi(cand, dist, E, S, mode='fst32')
    assert abs(c32 - cost) <= 1e-5 * cost
    assert o.viterbi(cand[:1], dist[:1], E, S) == ([], np.inf)      # T<2 edge case


def test_knn_by_class_padding(mini_voice):
    F = mini_voice['F'][:300]
    cls = np.arange(300) % 7
    cls[:3] = 99                      # a class with only 3 members
    U = F[[10, 0, 20]] + 0.01
    cand, dist = o.knn_by_class(F, U, 5, cls, [cls[10], 99, cls[20]])
    assert np.all(cls[cand[0]] == cls[10]) and cand[0, 0] == 10
    assert list(cand[1, 3:]) == [-1, -1] and np.all(dist[1, 3:] == o.VERY_BIG_WEIGHT_VALUE)
    assert set(cand[1, :3]) == {0, 1, 2}


def test_quinphone_preselection_matches_reference(golden, mini_voice):
    names = [n.decode() for n in golden['quin_unit_names']]
    qnames = [n.decode() for n in golden['quin_query_names']]
    index = o.build_unit_index(names)
    cand, dist = o.preselect_units_quinphone(index, mini_voice['F'], golden['quin_queries'], qnames, 9)
    assert np.array_equal(cand, golden['quin_candidates'])
    np.testing.assert_allclose(dist, golden['quin_distances'], rtol=1e-12)
    assert list(golden['quin_candidates'][2]) == [1] + [-1] * 8          # unseen label -> naive back-off


def test_knn_by_class_matches_reference_monophone_preselection(golden, mini_voice):
    """oracle.knn_by_class against the REFERENCE's preselect_units_monophone_then_acoustic output
    (tests/golden/reference_preselect.npz, generated by tools/make_golden.py)."""
    import os
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_preselect.npz'))
    K = int(ref['mono_n_candidates'])
    names = [n.decode() for n in golden['quin_unit_names']]
    qnames = [n.decode() for n in ref['mono_query_names']]
    monos = sorted(set(n.split('/')[2] for n in names))
    ucls = np.array([monos.index(n.split('/')[2]) for n in names], dtype=np.int32)
    qcls = np.array([monos.index(n.split('/')[2]) for n in qnames], dtype=np.int32)
    cand, dist = o.knn_by_class(mini_voice['F'], ref['mono_queries'], K, ucls, qcls)
    assert np.array_equal(cand, ref['mono_candidates'])
    np.testing.assert_allclose(dist, ref['mono_distances'], rtol=1e-12)


def _parse_fst_text(text):
    """AT&T text as the reference prints it into openfst.Compiler(): arcs 'src dst ilabel olabel [w]',
    last line = the final state."""
    arcs, final = [], None
    for line in text.splitlines():
        f = line.split()
        if len(f) == 1:
            final = int(f[0])
        else:
            arcs.append((int(f[0]), int(f[1]), int(f[2]), int(f[3]), float(f[4]) if len(f) > 4 else 0.0))
    return arcs, final


def test_viterbi_on_the_lattices_the_reference_itself_emits(golden, mini_voice):
    """The arc text of the target sausage and of the join lattice, recorded from the reference's OWN
    make_target_sausage_lattice / cost_cache_to_compiled_fst (a recording stand-in for
    openfst.Compiler, tools/make_golden.py) for the 40-frame, K = 12 fixture with its padding,
    first/last-unit and natural-join edits:
      * the oracle's restatement of the two builders emits the same arcs;
      * the tropical shortest path of T o J over that text (independent product search) is the path
        and cost of the oracle's DP.
    What stays unpinned is OpenFST's own compose / shortestpath arithmetic (float32 weights)."""
    import os
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_preselect.npz'))
    T_arcs, T_final = _parse_fst_text(bytes(ref['fst_target_text']).decode())
    J_arcs, J_final = _parse_fst_text(bytes(ref['fst_join_text']).decode())
    E, S = mini_voice['E'], mini_voice['S']
    cand, dist = golden['join_candidates'], golden['knn_distances']
    cache = dict(zip(map(tuple, golden['join_cache_keys'].tolist()), golden['join_cache_values']))
    (oT, oTf), (oJ, oJf) = o.fst_arc_lists(cand, dist, cache)
    assert oTf == T_final and oJf == J_final
    assert oT == T_arcs                                   # same arcs in the same order ('%s' of a float round-trips)
    assert sorted(oJ) == sorted(J_arcs)                   # the cache's iteration order is not part of the semantics
    p_fst, c_fst = o.fst_shortest_path_bruteforce((T_arcs, T_final), (J_arcs, J_final))
    p_dp, c_dp = o.viterbi(cand, dist, E, S)
    assert p_fst == p_dp and len(p_dp) == cand.shape[0]
    assert abs(c_fst - c_dp) <= 1e-12 * c_dp
    # float32 weights as OpenFST stores them: the best path of this fixture does not change
    p32, c32 = o.viterbi(cand, dist, E, S, mode='fst32')
    assert p32 == p_dp and abs(c32 - c_dp) <= 1e-5 * c_dp
