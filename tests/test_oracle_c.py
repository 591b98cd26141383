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
