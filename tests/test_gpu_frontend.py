"""GPU tests of the drop-in Synthesiser front end: driven through config file + unit database +
stream files exactly like the reference's synth_simple.py / synth_halfphone.py."""
import os

import numpy as np
import pytest

import snk_oracle as o
from voice_fixture import build_voice

pytestmark = pytest.mark.gpu


def test_synth_simple_flavour_reproduces_reference_paths(tmp_path, golden, mini_voice):
    from snickery_amd.synthesiser import Synthesiser
    for me in (6, 1):
        cfgfile, config = build_voice(tmp_path, golden, greedy=True, multiepoch=me)
        synth = Synthesiser(cfgfile, verbose=False)
        assert synth.flavour == 'simple'
        assert synth.get_sentence_set('test') == ['arctic_b0001']
        paths = synth.synth_from_config()
        assert np.array_equal(np.array(paths['arctic_b0001']), golden['greedy_me%d_utt0_path' % me])
        p1 = synth.greedy_joint_search(golden['greedy_me%d_utt1_unit_features' % me])
        assert np.array_equal(np.array(p1), golden['greedy_me%d_utt1_path' % me])
        synth.close()


def test_reconfigure_settings_and_scores(tmp_path, golden, mini_voice):
    from snickery_amd.synthesiser import Synthesiser
    cfgfile, config = build_voice(tmp_path, golden, greedy=True, multiepoch=3)
    synth = Synthesiser(cfgfile, verbose=False)
    new = dict(config)
    new.update(join_stream_weights=[0.4, 0.3, 0.2, 0.1], target_stream_weights=[0.3, 0.7], join_cost_weight=0.35,
               search_epsilon=0.0, multiepoch=4, magphase_use_target_f0=True, magphase_overlap=0,
               truncate_target_streams=[40, -1], truncate_join_streams=[30, -1, 20, 1])
    desc = synth.reconfigure_settings(new)
    assert 'join_cost_weight: 0.2 -> 0.35' in desc and 'multiepoch: 3 -> 4' in desc
    assert synth.reconfigure_settings(new) == ''
    # oracle with the reference's own semantics: weight, then DROP the truncated columns
    dims = {'lf0': 1, 'mag': 60, 'real': 45, 'imag': 45}
    tw, jw = o.apply_jcw(new['target_stream_weights'], new['join_stream_weights'], new['join_cost_weight'])
    wt = o.stream_weight_vector(list(tw), ['mag', 'lf0'], dims)
    wj = o.stream_weight_vector(list(jw), ['mag', 'real', 'imag', 'lf0'], dims)
    F, E, S = o.weighted_db(golden['F_unw'], golden['JC_unw'], wt, wj)
    tsel = list(range(40)) + [60]
    jsel = list(range(30)) + list(range(60, 105)) + list(range(105, 125)) + [150]
    F, E, S = F[:, tsel], E[:, jsel], S[:, jsel]
    pr, cr, Fwin = o.greedy_layout(F, E, S, 4)
    synth.mode_of_operation = 'stream_weight_balancing'
    tscores, jscores = synth.synth_utt('arctic_b0001', synth_type='test')
    synth.mode_of_operation = 'normal'
    path = synth.synth_utt('arctic_b0001', synth_type='test')
    U = synth.prepare_targets('arctic_b0001')[:, tsel]
    op, _ = o.greedy_search(pr, cr, Fwin, o.greedy_queries(U, 4))
    assert path == op
    # per-stream scores (what balance_stream_weights.py consumes)
    ot = o.target_scores(Fwin, o.greedy_queries(U, 4), op)
    oj = o.join_scores_greedy(pr, cr, op)
    assert tscores.shape == (len(op), 2) and jscores.shape == (len(op) - 1, 4)
    np.testing.assert_allclose(tscores.sum(), ot.sum(), rtol=1e-12)
    np.testing.assert_allclose(jscores.sum(), oj.sum(), rtol=1e-12)
    synth.close()


def test_halfphone_flavour_viterbi(tmp_path, golden, mini_voice):
    """greedy_search=False: preselect + Viterbi on a train_simple database (the combination
    BASELINE config 2 describes; parity against the oracle)."""
    from snickery_amd.synthesiser import Synthesiser
    cfgfile, config = build_voice(tmp_path, golden, greedy=False, multiepoch=1, n_candidates=12)
    synth = Synthesiser(cfgfile, verbose=False)
    assert synth.flavour == 'halfphone'
    U = synth.prepare_targets('arctic_b0001')
    assert np.array_equal(U, golden['greedy_me6_utt0_unit_features'][1:-1])     # speech[1:-1] (:1525)
    cand, dist = synth.preselect_units_acoustic(golden['knn_queries'])
    assert np.array_equal(cand, golden['knn_candidates'])
    path = synth.viterbi_search(golden['join_candidates'], golden['knn_distances'])
    opath, ocost = o.viterbi(golden['join_candidates'], golden['knn_distances'], mini_voice['E'], mini_voice['S'])
    assert path == opath and synth.last_path_cost == ocost
    full = synth.synth_utt('arctic_b0001', synth_type='test')
    oc, od = o.knn_bruteforce(mini_voice['F'], U, 12)
    op, _ = o.viterbi(oc, od, mini_voice['E'], mini_voice['S'])
    assert full == op
    synth.close()


def test_quinphone_preselection(tmp_path, golden, mini_voice):
    """preselect_units_quinphone: label back-off on the host, candidate distances on the GPU;
    candidates identical to the reference's own output, distances bit-exact vs the oracle."""
    from snickery_amd.synthesiser import Synthesiser
    cfgfile, config = build_voice(tmp_path, golden, greedy=False, multiepoch=1, n_candidates=9)
    synth = Synthesiser(cfgfile, verbose=False)
    synth.train_unit_names = golden['quin_unit_names']
    qnames = [n.decode() for n in golden['quin_query_names']]
    cand, dist = synth.preselect_units_quinphone(golden['quin_queries'], qnames)
    assert np.array_equal(cand, golden['quin_candidates'])
    np.testing.assert_allclose(dist, golden['quin_distances'], rtol=1e-12)
    assert np.array_equal(dist, o.candidate_distances(mini_voice['F'], golden['quin_queries'], cand))
    synth.close()


def test_shard_engine_device_pointer_api(golden, mini_voice):
    """The multi-GPU exchange path on one GPU: shard-local top-K written straight into torch CUDA
    tensors (what RCCL all-gathers), then merged on device.  Two half-database 'shards' on the same
    GPU stand in for two ranks; the merged result must equal the unsharded search."""
    import torch
    import snickery_amd
    from snickery_amd.dist import HipShardEngine, ShardedSearch, shard_bounds
    F_unw, JC_unw = mini_voice['F_unw'], mini_voice['JC_unw']
    N = F_unw.shape[0]
    U = golden['knn_queries']
    K = 12
    dev = torch.device('cuda', 0)
    d2_all = torch.empty(2, U.shape[0], K, dtype=torch.float64, device=dev)
    id_all = torch.empty(2, U.shape[0], K, dtype=torch.int64, device=dev)
    engines = []
    for r in range(2):
        lo, hi = shard_bounds(N, 2, r)
        e = snickery_amd.HipSearchEngine(0)
        e.upload_target_only(F_unw[lo:hi])
        e.upload_join_only(JC_unw)
        e.set_shard(lo, N)
        e.set_weights(mini_voice['wt'], mini_voice['wj'])
        e.knn_local_dev(U, K, d2_all[r].data_ptr(), id_all[r].data_ptr())
        engines.append(e)
    torch.cuda.synchronize()
    cand, dist = engines[0].merge_topk_dev(d2_all.data_ptr(), id_all.data_ptr(), 2, U.shape[0], K)
    assert np.array_equal(cand, golden['knn_candidates'])
    oc, od = o.knn_bruteforce(mini_voice['F'], U, K)
    assert np.array_equal(cand, oc) and np.array_equal(dist, od)
    # Viterbi on a "rank" that holds only a target shard but the full join matrix
    path, cost = engines[1].viterbi(golden['join_candidates'], golden['knn_distances'])
    assert (path, cost) == o.viterbi(golden['join_candidates'], golden['knn_distances'], mini_voice['E'], mini_voice['S'])
    # world_size 1 through the ShardedSearch front end
    search = ShardedSearch(HipShardEngine(engines[0], dev), rank=0, world_size=1)
    engines[0].upload_db(F_unw, JC_unw)
    engines[0].set_shard(0, N)
    engines[0].set_weights(mini_voice['wt'], mini_voice['wj'])
    c1, d1 = search.knn(U, K)
    assert np.array_equal(c1, oc) and np.array_equal(d1, od)
    for e in engines:
        e.close()


@pytest.mark.parametrize('method', ['monophone_then_acoustic', 'quinphone'])
def test_halfphone_label_driven_synth_utt(tmp_path, golden, method):
    """synth_halfphone.py:1478-1625 end to end for a label-driven voice: twopoint halfphone targets
    + normalised duration from the state-aligned label (host, pinned to the reference in
    test_hostprep.py), label-aware preselection and Viterbi on the GPU; against the oracle."""
    import re
    from snickery_amd import hostprep as hp
    from snickery_amd.synthesiser import Synthesiser
    from voice_fixture import build_halfphone_voice
    K = 10
    cfgfile, config, db = build_halfphone_voice(tmp_path, golden, method, n_candidates=K)
    synth = Synthesiser(cfgfile, verbose=False)
    assert synth.target_weight_vector.size == 2 * 61 + 1 and synth.target_weight_vector[-1] == 0.3
    U, names = synth.prepare_targets('arctic_b0001', 'test', return_names=True)
    # the same preparation spelled out with the host functions
    dirs = hp.locate_stream_directories(config['test_data_dirs'], config['stream_list_target'])
    speech = hp.standardise(hp.compose_speech(dirs, 'arctic_b0001', config['stream_list_target'],
                                              config['datadims_target']), golden['mean_target'], golden['std_target'])
    labs = hp.suppress_weird_festival_pauses(hp.read_label(os.path.join(config['test_lab_dir'], 'arctic_b0001.lab'),
                                                          re.compile(config['quinphone_regex'])))
    enames, feats, timings = hp.get_halfphone_stats(speech, labs, 'twopoint')
    nd = hp.get_norm_durations(enames, timings, synth.duration_stats) * 1.1
    expect = hp.weight(np.hstack([feats, nd]), synth.target_weight_vector)
    assert list(names) == list(enames) and np.array_equal(U, expect) and U.shape == (22, 123)
    F = o.weight(db['train_unit_features'], synth.target_weight_vector)
    JCw = o.weight(db['join_contexts'], synth.join_weight_vector)
    unit_names = [n.decode() for n in db['train_unit_names']]
    if method == 'monophone_then_acoustic':
        monos = [n.split('/')[2] for n in unit_names]
        ids = dict((m, i) for i, m in enumerate(sorted(set(monos))))
        ocand, odist = o.knn_by_class(F, U, K, np.array([ids[m] for m in monos]),
                                      np.array([ids[n.split('/')[2]] for n in names]))
        cand, dist = synth.preselect_units_monophone_then_acoustic(U, names)
    else:
        ocand, odist = o.preselect_units_quinphone(o.build_unit_index(unit_names), F, U, list(names), K)
        cand, dist = synth.preselect_units_quinphone(U, names)
        assert any('pau' in n for n in names)            # unseen phone -> the reference's naive back-off
    assert np.array_equal(cand, ocand) and np.array_equal(dist, odist)
    opath, ocost = o.viterbi(ocand, odist, JCw[1:], JCw[:-1])
    assert synth.synth_utt('arctic_b0001', synth_type='test') == opath
    assert synth.last_path_cost == ocost
    # a list of utterances: label-driven preselection per utterance, ONE Viterbi call for all (snk_viterbi_batch)
    before = synth.engine.timers().get('viterbi_sparse', (0, 0))[1]
    bulk = synth.synth_utts_bulk(['arctic_b0001', 'arctic_b0001', 'arctic_b0001'], synth_type='test')
    assert bulk == [opath, opath, opath]
    assert synth.engine.timers().get('viterbi_sparse', (0, 0))[1] > before
    # a tuning loop: other stream weights, the same utterances.  The label-dependent preparation (targets before
    # weighting, names, quinphone candidate ids) is reused; the result is that of a search started from the files.
    n_streams = len(synth.stream_list_target)
    synth.set_target_weights(list(0.3 + 0.4 * np.arange(1, n_streams + 1) / n_streams))
    synth.set_join_weights([0.7] * len(synth.stream_list_join))
    assert len(synth._bulk_cache) == 1
    warm = synth.synth_utts_bulk(['arctic_b0001', 'arctic_b0001'], synth_type='test')
    cold = synth.synth_utt('arctic_b0001', synth_type='test')
    assert warm == [cold, cold]
    U2, names2 = synth.prepare_targets('arctic_b0001', 'test', return_names=True)
    F2 = o.weight(db['train_unit_features'], synth.target_weight_vector)
    JC2 = o.weight(db['join_contexts'], synth.join_weight_vector)
    if method == 'quinphone':
        c2, d2 = o.preselect_units_quinphone(o.build_unit_index(unit_names), F2, U2, list(names2), K)
    else:
        c2, d2 = o.knn_by_class(F2, U2, K, np.array([ids[m] for m in monos]), np.array([ids[n.split('/')[2]] for n in names2]))
    assert cold == o.viterbi(c2, d2, JC2[1:], JC2[:-1])[0]
    synth.close()


@pytest.mark.parametrize('greedy,me', [(False, 1), (True, 3)])
def test_bulk_synthesis_and_stream_weight_balancing(tmp_path, golden, mini_voice, greedy, me):
    """balance_stream_weights.py's consumer pattern on the GPU path: the whole tune set through the
    batch entry point (K-NN + Viterbi batch, or two greedy searches per scan on a greedy voice -- what
    the reference's loop tunes) equals utterance-by-utterance synth_utt, and the balancing loop's first
    measurement is exactly the per-stream contributions of those paths."""
    from snickery_amd.synthesiser import Synthesiser
    from snickery_amd.balance_stream_weights import balance_stream_weights, mean_nonzero_contributions
    extra = '''
join_cost_weight = 1.0
tune_data_dirs = test_data_dirs
tune_patterns = ['arctic_b']
n_tune_utts = 5
'''
    cfgfile, config = build_voice(tmp_path, golden, greedy=greedy, multiepoch=me, n_candidates=12, extra_config=extra)
    for stream in ('mag', 'lf0'):                       # a second, shorter tune utterance
        golden['test0_raw_' + stream][10:97].astype(np.float32).tofile(
            os.path.join(config['data'], 'low', stream, 'arctic_b0002.' + stream))
    synth = Synthesiser(cfgfile, verbose=False)
    names = synth.get_sentence_set('tune')
    assert names == ['arctic_b0001', 'arctic_b0002']
    paths = synth.synth_utts_bulk(names, synth_type='tune')
    assert paths == [synth.synth_utt(n, synth_type='tune') for n in names]
    synth.mode_of_operation = 'stream_weight_balancing'
    bulk = synth.synth_utts_bulk(names, synth_type='tune')
    single = [synth.synth_utt(n, synth_type='tune') for n in names]
    for (bt, bj), (st, sj) in zip(bulk, single):
        assert np.array_equal(bt, st) and np.array_equal(bj, sj)
        assert bt.shape[1] == 2 and bj.shape[1] == 4
    res = balance_stream_weights(synth, max_epochs=3, report=lambda *_: None)
    synth.set_join_weights(np.ones(4))
    synth.set_target_weights(np.ones(2))
    ones = [synth.synth_utt(n, synth_type='tune') for n in names]
    first = mean_nonzero_contributions(np.vstack([j for t, j in ones]), np.vstack([t for t, j in ones]))
    assert np.array_equal(res['contribs'][0], first)
    assert len(res['losses']) == 3 and res['best_weights'].shape == (6,) and np.all(res['best_weights'] >= 0)
    synth.close()


def test_join_knn(tmp_path, golden, mini_voice):
    """initialise_join_table_with_knn's search (active_learning_join.py:198-204): unit_start_data rows
    against unit_end_data, on the engine; bit-exact against the oracle's brute force."""
    from snickery_amd.synthesiser import Synthesiser
    cfgfile, config = build_voice(tmp_path, golden, greedy=False, multiepoch=1, n_candidates=12)
    synth = Synthesiser(cfgfile, verbose=False)
    E, S = mini_voice['E'], mini_voice['S']
    idx, dist = synth.join_knn(6, first=100, last=260)
    oi, od = o.knn_bruteforce(E, S[100:260], 6)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    # the natural successor joins at exactly 0 (E[i] == S[i+1]): unit i+1's start row finds unit i first
    assert np.all(idx[:, 0] == np.arange(100, 260) - 1) and np.all(dist[:, 0] == 0.0)
    synth.close()


@pytest.mark.parametrize('overlap', [2, 0, 4])
def test_concatenate_magphase_matches_reference(tmp_path, golden, overlap):
    """The step after the search: fragments of the selected units cross-faded and overlap-added on the
    GPU equal, bit for bit, the matrices the REFERENCE's concatenateMagPhaseEpoch_sep_files handed to
    its vocoder for the same path (utterance starts, windows past an utterance end, a repeated unit)."""
    from snickery_amd.synthesiser import Synthesiser
    from voice_fixture import write_full_spectra
    H = 17
    extra = "\nfull_magphase_dir = data + '/high'\nmagphase_overlap = 2\n"
    names = [n.decode() for n in golden['concat_utt_names']]
    override = dict(filenames=golden['concat_filenames'], unit_index_within_sentence_dset=golden['concat_unit_index'])
    cfgfile, config = build_voice(tmp_path, golden, greedy=True, multiepoch=6, extra_config=extra, db_override=override)
    write_full_spectra(config['full_magphase_dir'], list(zip(names, golden['concat_utt_frames'].tolist())), H, seed=77)
    synth = Synthesiser(cfgfile, verbose=False)
    synth.load_full_magphase(fft_half_len=H)
    path = golden['concat_path' if overlap else 'concat_path_no_overlap']
    mag, real, imag, fz = synth.concatenate_magphase([int(p) for p in path], overlap=overlap)
    assert mag.shape == golden['concat_ov%d_mag' % overlap].shape == (len(path) * 6, H)
    assert np.array_equal(mag, golden['concat_ov%d_mag' % overlap])
    assert np.array_equal(real, golden['concat_ov%d_real' % overlap])
    assert np.array_equal(imag, golden['concat_ov%d_imag' % overlap])
    assert np.array_equal(fz, golden['concat_ov%d_fz' % overlap])
    if overlap == 0:                              # like the reference, no silent short fragments
        with pytest.raises(Exception):
            synth.concatenate_magphase([int(p) for p in golden['concat_path']], overlap=0)
    synth.close()


def test_writer_to_search_round_trip_epoch_database(tmp_path):
    """Producer -> consumer: a pitch-synchronous epoch database written by snickery_amd.train_halfphone
    (2 x Dj join rows, byte-identical to the reference's file, tests/test_hostprep.py) is loaded by the
    drop-in Synthesiser and searched (K-NN + join costs + Viterbi) on the GPU; the path equals the
    oracle's on the arrays of that database, with the doubled join weight vector of
    synth_halfphone.py:693-695."""
    import sys
    from snickery_amd import hostprep as hp, train_halfphone
    from snickery_amd.synthesiser import Synthesiser
    import voice_fixture
    data = os.path.join(str(tmp_path), 'corpus')
    voice_fixture.write_halfphone_corpus(data)
    cfgfile = voice_fixture.halfphone_corpus_config(os.path.join(str(tmp_path), 'hp.cfg'),
                                                    os.path.join(str(tmp_path), 'work'), data, 'epoch', False)
    with open(cfgfile, 'a') as f:
        f.write('''
test_data_dirs = join_datadirs
n_test_utts = 1
weight_target_data = True
weight_join_data = True
target_stream_weights = [0.1, 1.0]
join_stream_weights = [0.25, 0.25, 0.25, 0.25]
join_cost_weight = 0.2
greedy_search = False
search_epsilon = 0.0
multiepoch = 1
n_candidates = 8
preselection_method = 'acoustic'
join_cost_type = 'natural2'
get_selection_info = False
hold_waves_in_memory = False
preload_all_magphase_utts = False
''')
    config = hp.load_config(cfgfile)
    train_halfphone.main_work(config, report=lambda *_: None)
    synth = Synthesiser(cfgfile, verbose=False)
    assert synth.flavour == 'halfphone' and synth._double_join
    db = hp.load_database(train_halfphone.get_data_dump_name(config))
    dims = config['datadims']
    tw, jw = o.apply_jcw(config['target_stream_weights'], config['join_stream_weights'], config['join_cost_weight'])
    wt = o.stream_weight_vector(list(tw), config['stream_list_target'], dims)
    wj1 = o.stream_weight_vector(list(jw), config['stream_list_join'], dims)
    wj = np.concatenate([wj1, wj1])
    F, E, S = o.weighted_db(db['train_unit_features'], db['join_contexts'], wt, wj)
    U = synth.prepare_targets('arctic_b0001')
    path = synth.synth_utt('arctic_b0001', synth_type='test')
    oc, od = o.knn_bruteforce(F, U, 8)
    op, ocost = o.viterbi(oc, od, E, S)
    assert path == op and synth.last_path_cost == ocost
    # the all-pairs join K-NN (initialise_join_table_with_knn, active_learning_join.py:184-212) on this voice's doubled
    # [j_t, j_t+1] join rows: 2 x 151 = 302 columns -- the blocked bf16-split product of knn_wide16b + the exact re-rank
    # (the matrix path: asserted below); bit-exact against the oracle's brute force, natural successors first at distance 0
    assert S.shape[1] == 302
    n = min(S.shape[0], 300)
    idx, dist = synth.join_knn(7, first=1, last=n)
    oi, od2 = o.knn_bruteforce(E, S[1:n], 7)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od2)
    assert np.all(idx[:, 0] == np.arange(1, n) - 1) and np.all(dist[:, 0] == 0.0)
    assert synth._join_engine.info('wide_launches') >= 1 and synth._join_engine.info('exact_row_fallbacks') == 0
    synth.close()


def test_wide_rows_take_the_matrix_path():
    """K-NN on rows of 257 .. 512 columns (the doubled join rows of an epoch voice as a database: initialise_join_table_with_knn,
    active_learning_join.py:184-212): the blocked bf16-split product (knn_wide16b) + exact float64 re-rank serves the plain
    search -- asserted through the wide_launches counter --, the canonical-distance selection the class-restricted one and
    `precision 0`; K = 200, exact duplicates (ties: the lower id first), the oracle's candidates and distances bit for bit."""
    import snickery_amd
    N, Dt, K = 5000, 302, 200
    F_unw, JC_unw = o.synthetic_db(N, Dt, 8, seed=91)
    F_unw[4000:4010] = F_unw[50:60]                      # exact duplicates: ties, the lower id first
    rng = np.random.RandomState(92)
    wt = 0.1 + rng.rand(Dt)
    e = snickery_amd.HipSearchEngine(0)
    e.upload_db(F_unw, JC_unw)
    e.set_weights(wt, np.full(8, 0.1))
    assert e.info('wide_ready') == 0                      # built at the first call that can take the path, not by snk_set_weights
    F = o.weight(F_unw, wt)
    U = np.vstack([o.synthetic_targets(F_unw, 40, seed=93), F_unw[50:60].astype(np.float64)]) * wt
    before = e.info('wide_launches')
    cand, dist = e.knn(U, K)
    assert e.info('wide_ready') == 1
    assert e.info('wide_launches') == before + 1 and e.info('f16_fallbacks') == 0
    oc_, od_ = o.knn_bruteforce(F, U, K)
    assert np.array_equal(cand, oc_) and np.array_equal(dist, od_)
    assert list(cand[40:, 0]) == list(range(50, 60)) and list(cand[40:, 1]) == list(range(4000, 4010))
    e.set_option('precision', 0)                         # the exact selection (a workgroup per query row): same results
    c0, d0 = e.knn(U, K)
    assert e.info('wide_launches') == before + 1 and np.array_equal(c0, oc_) and np.array_equal(d0, od_)
    e.set_option('precision', 1)
    cls = rng.randint(0, 7, size=N).astype(np.int32)
    qc = rng.randint(0, 7, size=U.shape[0]).astype(np.int32)
    e.set_unit_classes(cls)
    c2, d2 = e.knn_by_class(U, 30, qc)
    oc2, od2 = o.knn_by_class(F, U, 30, cls, qc)
    assert np.array_equal(c2, oc2) and np.array_equal(d2, od2)
    assert e.info('f16_ready') == 0
    e.close()


@pytest.mark.parametrize('N,Dt,T,K', [(60000, 302, 700, 100), (20000, 257, 33, 7), (40000, 509, 100, 50), (9000, 317, 64, 1)])
def test_wide_rows_against_the_c_oracle(N, Dt, T, K):
    """The blocked product at database sizes where stage A works on a sample (stride > 1), at the widths' edges (257: five
    k-blocks of a 320-column pad; 317 and 509: the last widths with three spare columns in their pads) and K = 1."""
    import snickery_amd
    import snk_oracle_c as oc
    F_unw, JC_unw = o.synthetic_db(N, Dt, 8, seed=N % 71)
    rng = np.random.RandomState(5)
    wt = 0.1 + rng.rand(Dt)
    e = snickery_amd.HipSearchEngine(0)
    e.upload_db(F_unw, JC_unw)
    e.set_weights(wt, np.full(8, 0.1))
    F = o.weight(F_unw, wt)
    U = np.vstack([o.synthetic_targets(F_unw, T - T // 3, seed=3), F_unw[rng.randint(0, N, T // 3)] + 0.2 * rng.randn(T // 3, Dt)]) * wt
    cand, dist = e.knn(U, K)
    assert e.info('wide_ready') == 1
    assert e.info('wide_launches') == 1 and e.info('f16_fallbacks') == 0
    oc_, od_ = oc.knn(F, U, K)
    assert np.array_equal(cand, oc_) and np.array_equal(dist, od_)
    e.close()


def test_monophone_then_acoustic_matches_reference_output(tmp_path, golden, mini_voice):
    """preselect_units_monophone_then_acoustic against the REFERENCE's own output for the same
    database, labels and queries (per-phone cKDTrees + index converters, synth_halfphone.py:385-402,
    1369-1396; tests/golden/reference_preselect.npz): identical candidates, distances to 1e-12 and
    bit-exact against the oracle."""
    from snickery_amd.synthesiser import Synthesiser
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_preselect.npz'))
    K = int(ref['mono_n_candidates'])
    cfgfile, config = build_voice(tmp_path, golden, greedy=False, multiepoch=1, n_candidates=K)
    synth = Synthesiser(cfgfile, verbose=False)
    synth.train_unit_names = golden['quin_unit_names']
    qnames = [n.decode() for n in ref['mono_query_names']]
    cand, dist = synth.preselect_units_monophone_then_acoustic(ref['mono_queries'], qnames)
    assert np.array_equal(cand, ref['mono_candidates'])
    np.testing.assert_allclose(dist, ref['mono_distances'], rtol=1e-12)
    names = [n.decode() for n in golden['quin_unit_names']]
    monos = sorted(set(n.split('/')[2] for n in names))
    ucls = np.array([monos.index(n.split('/')[2]) for n in names], dtype=np.int32)
    qcls = np.array([monos.index(n.split('/')[2]) for n in qnames], dtype=np.int32)
    oc, od = o.knn_by_class(mini_voice['F'], ref['mono_queries'], K, ucls, qcls)
    assert np.array_equal(cand, oc) and np.array_equal(dist, od)
    synth.close()


def test_per_stream_scores_match_reference_output(tmp_path, golden):
    """get_target_scores_per_stream / get_join_scores_per_stream (Viterbi and greedy forms) against
    the REFERENCE's own values for the same database, weights, queries and path
    (synth_halfphone.py:1964-1981, 2977-3008; tests/golden/reference_preselect.npz)."""
    from snickery_amd.synthesiser import Synthesiser
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_preselect.npz'))
    path = [int(v) for v in ref['scores_path']]
    feats = golden['knn_queries']
    cfgfile, config = build_voice(tmp_path, golden, greedy=False, multiepoch=1, n_candidates=12)
    synth = Synthesiser(cfgfile, verbose=False)
    np.testing.assert_allclose(synth.get_target_scores_per_stream(feats, path), ref['scores_target'], rtol=1e-12)
    np.testing.assert_allclose(synth.get_join_scores_per_stream(path), ref['scores_join_viterbi'], rtol=1e-12)
    synth.close()
    cfgfile, config = build_voice(tmp_path, golden, greedy=True, multiepoch=1)
    synth = Synthesiser(cfgfile, verbose=False)
    np.testing.assert_allclose(synth.get_target_scores_per_stream(feats, path), ref['scores_target'], rtol=1e-12)
    np.testing.assert_allclose(synth.get_join_scores_per_stream(path), ref['scores_join_greedy'], rtol=1e-12)
    synth.close()


def test_reconfigure_settings_matches_reference_end_to_end(tmp_path, golden):
    """The weight-tuning entry point against the REFERENCE's own run (tests/golden/reference_reconf.npz:
    a fresh synth_simple.Synthesiser reconfigured with new stream weights, join cost weight,
    multiepoch 6 -> 4 and truncated streams, then synth_utt): same description text, same prepared
    target features, same greedy path."""
    import json
    from snickery_amd.synthesiser import Synthesiser
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_reconf.npz'))
    extra = "truncate_target_streams = [-1, -1]\ntruncate_join_streams = [-1, -1, -1, -1]\nmagphase_overlap = 2\nmagphase_use_target_f0 = True\n"
    cfgfile, config = build_voice(tmp_path, golden, greedy=True, multiepoch=6, extra_config=extra,
                                  db_override=dict(filenames=ref['reconf_filenames'],
                                                   unit_index_within_sentence_dset=ref['reconf_unit_index']))
    synth = Synthesiser(cfgfile, verbose=False)
    new = dict(config)
    new.update(json.loads(bytes(ref['reconf_settings_json']).decode()))
    assert synth.reconfigure_settings(new) == bytes(ref['reconf_description']).decode()
    U = synth.prepare_targets('arctic_b0001')
    # the reference DROPS the truncated columns; here they stay in place with weight 0 (adds exactly +0.0)
    tsel = list(range(40)) + [60]
    np.testing.assert_allclose(U[:, tsel], ref['reconf_unit_features'], rtol=1e-12, atol=1e-300)
    assert not np.any(U[:, 40:60])
    path = synth.synth_utt('arctic_b0001', synth_type='test')
    assert np.array_equal(np.array(path), ref['reconf_path'])
    # the .trace.txt lines (get_path_information_epoch); the reference ran under Python 3 here, where its
    # HDF5 file names are bytes and print as b'...': Python 2 prints the bare name
    want = [l.replace("b'", '').replace("'", '') for l in bytes(ref['reconf_trace_lines']).decode().splitlines()]
    assert synth.get_path_information_epoch(U, path) == want
    synth.close()


def test_bulk_greedy_equals_one_by_one(tmp_path, golden):
    """synth_utts_bulk on a greedy voice (snk_greedy_batch: two utterances per scan of the database)
    returns what synth_utt returns utterance by utterance -- the reference's own path for the test
    sentence -- in normal and in stream-weight-balancing mode."""
    from snickery_amd.synthesiser import Synthesiser
    cfgfile, config = build_voice(tmp_path, golden, greedy=True, multiepoch=6)
    synth = Synthesiser(cfgfile, verbose=False)
    names = ['arctic_b0001'] * 3
    bulk = synth.synth_utts_bulk(names, synth_type='test')
    assert all(np.array_equal(np.array(p), golden['greedy_me6_utt0_path']) for p in bulk)
    synth.mode_of_operation = 'stream_weight_balancing'
    one = synth.synth_utt('arctic_b0001', synth_type='test')
    for t, j in synth.synth_utts_bulk(names, synth_type='test'):
        assert np.array_equal(t, one[0]) and np.array_equal(j, one[1])
    synth.close()


def test_synth_from_config_replicas_equal_the_serial_loop(tmp_path, golden):
    """synth_from_config(ncores=2): the reference's multiprocessing.Pool over the sentences
    (synth_halfphone.py:897-903) as two replica processes with their own engines -- here both on GPU 0 --
    after the weights were changed at run time (the replicas must see them)."""
    from snickery_amd.synthesiser import Synthesiser
    extra = '''
tune_data_dirs = test_data_dirs
tune_patterns = ['arctic_b']
n_tune_utts = 5
'''
    cfgfile, config = build_voice(tmp_path, golden, greedy=False, multiepoch=1, n_candidates=12, extra_config=extra)
    for i, (lo, hi) in enumerate(((10, 97), (30, 140), (0, 60))):
        for stream in ('mag', 'lf0'):
            golden['test0_raw_' + stream][lo:hi].astype(np.float32).tofile(
                os.path.join(config['data'], 'low', stream, 'arctic_b%04d.%s' % (i + 2, stream)))
    synth = Synthesiser(cfgfile, verbose=False)
    synth.set_target_weights([0.0, 3.0])          # only lf0 counts: other paths than with the configured weights
    names = synth.get_sentence_set('tune')
    assert len(names) == 4
    serial = synth.synth_from_config(synth_type='tune')
    replicas = synth.synth_from_config(synth_type='tune', ncores=2, devices=[0, 0])
    assert list(replicas) == names
    for n in names:
        assert np.array_equal(np.asarray(serial[n]), np.asarray(replicas[n]))
    default_weights = Synthesiser(cfgfile, verbose=False)
    assert any(not np.array_equal(np.asarray(default_weights.synth_utt(n, synth_type='tune')), np.asarray(serial[n])) for n in names)
    default_weights.close()
    synth.close()
