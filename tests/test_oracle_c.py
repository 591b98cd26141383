"""The C restatement of the oracle (oracle/snk_oracle.c) must agree bit for bit with the numpy
oracle that is pinned against the reference's golden vectors."""
import numpy as np
import snk_oracle as o
import snk_oracle_c as oc


def test_c_oracle_matches_numpy_oracle_and_golden(golden, mini_voice):
    K = int(golden['knn_K'])
    cand, dist = oc.knn(mini_voice['F'], golden['knn_queries'], K)
    assert np.array_equal(cand, golden['knn_candidates'])
    ncand, ndist = o.knn_bruteforce(mini_voice['F'], golden['knn_queries'], K)
    assert np.array_equal(cand, ncand) and np.array_equal(dist, ndist)
    JCw = o.weight(mini_voice['JC_unw'], mini_voice['wj'])
    jc = golden['join_candidates']
    assert np.array_equal(oc.join_dense(JCw, jc), o.join_cost_dense(mini_voice['E'], mini_voice['S'], jc))
    assert oc.viterbi(jc, golden['knn_distances'], JCw) == o.viterbi(jc, golden['knn_distances'], mini_voice['E'], mini_voice['S'])


def test_c_oracle_padding_and_ties():
    F_unw, JC_unw = o.synthetic_db(300, 20, 12, seed=9)
    F_unw[40:60] = F_unw[7]
    wt = np.full(20, 0.3)
    F = o.weight(F_unw, wt)
    U = (F_unw[[7, 100]] + 0.01) * wt
    c1, d1 = oc.knn(F, U, 25)
    c2, d2 = o.knn_bruteforce(F, U, 25)
    assert np.array_equal(c1, c2) and np.array_equal(d1, d2)
    c1, d1 = oc.knn(F[:10], U, 16)
    c2, d2 = o.knn_bruteforce(F[:10], U, 16)
    assert np.array_equal(c1, c2) and np.array_equal(d1, d2)


def test_c_greedy_from_unweighted_f32_matches_numpy_oracle(golden, mini_voice):
    """snko_greedy_f32 (what the GPU tests use at N = 65 536 / 1.5 M) against the numpy oracle that is
    pinned to the reference's own greedy paths -- every layout, start states, the golden voice."""
    import itertools
    F_unw, JC_unw = o.synthetic_db(900, 13, 10, seed=3)
    rng = np.random.RandomState(4)
    wt, wj = 0.2 + rng.rand(13), 0.05 + rng.rand(10)
    F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
    U = o.synthetic_targets(F_unw, 50, seed=5) * wt
    for me, lfat, split, start in itertools.product([1, 3, 6], [False, True], [0, 1], [-1, 17]):
        pr, cr, Fwin = o.greedy_layout(F, E, S, me, lfat, split)
        p0, d0 = o.greedy_search(pr, cr, Fwin, o.greedy_queries(U, me, lfat), start)
        p1, d1 = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, me, lfat, split, start)
        assert p0 == p1 and np.array_equal(d0, d1), (me, lfat, split, start)
    # the reference's own path on the golden voice (multiepoch 6)
    for me in (6, 1):
        for utt in (0, 1):
            p, _ = oc.greedy_f32(mini_voice['F_unw'], mini_voice['JC_unw'], mini_voice['wt'], mini_voice['wj'],
                                 golden['greedy_me%d_utt%d_unit_features' % (me, utt)], me)
            assert p == [int(v) for v in golden['greedy_me%d_utt%d_path' % (me, utt)]]
    # squared distances of one step (used by the approximate-search contract tests)
    p2, d2, all_d2 = oc.greedy_f32(F_unw, JC_unw, wt, wj, U, 6, False, 0, -1, d2_step=2)
    assert all_d2.argmin() == p2[2] and np.sqrt(all_d2.min()) == d2[2]
