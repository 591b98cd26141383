"""snk_merge_topk_dev (the owner's merge of the shards' lists, knn_kernels.hip merge_topk_path_kernel) against a sort of the
concatenated lists: sorted lists as the shards write them (ties of the distance across shards, padding, any number of shards),
and lists that are NOT sorted (the entry point does not promise it: such a row is sorted by its wavefront)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reference(d2, ids, K):
    G, T, _ = d2.shape
    cand = np.full((T, K), -1, dtype=np.int64)
    dist = np.full((T, K), 1e15)
    for t in range(T):
        k = np.where(ids[:, t, :].ravel() >= 0, d2[:, t, :].ravel(), np.inf)
        i = ids[:, t, :].ravel()
        order = np.lexsort((i, k))[:K]
        ok = np.isfinite(k[order])
        cand[t, ok] = i[order][ok]
        dist[t, ok] = np.sqrt(k[order][ok])
    return cand, dist


@pytest.mark.parametrize('G,K,T', [(2, 7, 50), (3, 100, 40), (5, 64, 33), (8, 100, 300), (8, 208, 20), (16, 100, 10)])
def test_merge_of_the_shards_lists(mini_voice, G, K, T):
    import torch
    import snickery_amd
    rng = np.random.RandomState(G * 1000 + K)
    d2 = np.empty((G, T, K)); ids = np.empty((G, T, K), dtype=np.int64)
    for t in range(T):
        # unit ids are unique over the shards; distances from a small set of values so that shards tie
        perm = rng.permutation(G * K * 3)[:G * K].reshape(G, K)
        for g in range(G):
            n_valid = K if rng.rand() < 0.5 else rng.randint(0, K + 1)
            dd = np.round(rng.rand(K) * 20) / 4.0 if t % 3 else rng.rand(K)
            ii = perm[g] + 100000 * g
            order = np.lexsort((ii, dd))
            dd, ii = dd[order], ii[order]
            dd[n_valid:] = 123.0           # what a padded entry carries is never read
            ii[n_valid:] = -1
            d2[g, t], ids[g, t] = dd, ii
    dev = torch.device('cuda', 0)
    e = snickery_amd.HipSearchEngine(0)
    e.upload_db(mini_voice['F_unw'], mini_voice['JC_unw']); e.set_weights(mini_voice['wt'], mini_voice['wj'])
    for shuffled in (False, True):
        if shuffled:                        # lists in no order: rows 0, 2, 4 ...
            for t in range(0, T, 2):
                for g in range(G):
                    p = rng.permutation(K)
                    d2[g, t], ids[g, t] = d2[g, t][p], ids[g, t][p]
        td2 = torch.from_numpy(d2).to(dev); tid = torch.from_numpy(ids).to(dev)
        torch.cuda.synchronize()
        cand, dist = e.merge_topk_dev(td2.data_ptr(), tid.data_ptr(), G, T, K)
        rc, rd = _reference(d2, ids, K)
        assert np.array_equal(cand, rc), (G, K, shuffled)
        assert np.array_equal(dist, rd), (G, K, shuffled)
    e.close()
