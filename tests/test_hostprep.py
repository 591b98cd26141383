"""CPU tests of the host-side mirror of the reference interface (no GPU, no oracle needed):
target preparation and file naming against values captured from the reference itself."""
import os
import sys
import pytest
import numpy as np

from voice_fixture import build_voice


def test_target_preparation_matches_reference(tmp_path, golden):
    from snickery_amd import hostprep as hp
    cfgfile, config = build_voice(tmp_path, golden)
    dirs = hp.locate_stream_directories(config['test_data_dirs'], config['stream_list_target'])
    unnorm = hp.compose_speech(dirs, 'arctic_b0001', config['stream_list_target'], config['datadims_target'])
    assert unnorm.shape == (golden['test0_raw_mag'].shape[0], 61)
    assert np.any(unnorm[:, -1] == hp.SPECIAL_UV_VALUE)                  # unvoiced frames marked
    speech = hp.standardise(unnorm, golden['mean_target'], golden['std_target'])
    feats = hp.weight(speech, golden['target_weight_vector'])
    # the reference's own synth_utt produced exactly this matrix (tools/make_golden.py)
    assert np.array_equal(feats, golden['greedy_me6_utt0_unit_features'])
    assert feats.dtype == np.float64
    uv = unnorm[:, -1] == hp.SPECIAL_UV_VALUE
    assert np.allclose(speech[uv, -1], golden['std_target'][0, -1] * -20.0)


def test_missing_stream_returns_sentinel(tmp_path, golden):
    from snickery_amd import hostprep as hp
    cfgfile, config = build_voice(tmp_path, golden)
    dirs = hp.locate_stream_directories(config['test_data_dirs'], config['stream_list_target'])
    out = hp.compose_speech(dirs, 'does_not_exist', config['stream_list_target'], config['datadims_target'])
    assert out.shape == (1, 1)                                            # data_manipulation.py:25-27


def test_file_naming_matches_reference(tmp_path, golden):
    from snickery_amd import hostprep as hp
    cfgfile, config = build_voice(tmp_path, golden)
    assert os.path.basename(hp.get_data_dump_name(config)) == str(golden['db_basename'])
    name = hp.make_synthesis_condition_name(config)
    assert name.startswith('greedy-yes_target-0.1-1.0_join-0.25-0.25-0.25-0.25_scale-0.2_presel-acoustic')
    assert name.endswith('multiepoch-6')


def test_weight_vectors_and_truncation(golden):
    from snickery_amd import hostprep as hp
    dims = {'lf0': 1, 'mag': 60, 'real': 45, 'imag': 45}
    tw = np.array([0.1, 1.0]) * (1.0 - 0.2)
    vec = np.array(hp.stream_weight_vector(list(tw), ['mag', 'lf0'], dims))
    assert np.array_equal(vec, golden['target_weight_vector'])
    sel = hp.get_selection_vector(['mag', 'real', 'imag', 'lf0'], dims, [30, -1, 0, 1])
    assert sel == list(range(0, 30)) + list(range(60, 105)) + [150]


def test_database_loader_sidecar(tmp_path, golden):
    from snickery_amd import hostprep as hp
    cfgfile, config = build_voice(tmp_path, golden)
    db = hp.load_database(hp.get_data_dump_name(config))
    assert db['train_unit_features'].shape == golden['F_unw'].shape
    assert db['std_target'].shape == (1, 61) and db['mean_target'].shape == (61,)   # the (1,D) quirk
    assert db['join_contexts'].shape[0] == db['train_unit_features'].shape[0] + 1


def test_cabi_exports_every_declared_symbol():
    """The C-ABI library loads without a GPU and exports everything include/snk.h declares."""
    import re
    import snickery_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, 'include', 'snk.h')).read()
    declared = set(re.findall(r'\b(snk_[a-z0-9_]+)\s*\(', header))
    declared.discard('snk_engine')
    lib = snickery_amd.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert lib.snk_abi_version() == 1
    assert lib.snk_timer_count() >= 10 and lib.snk_timer_name(0).decode() != ''


def test_halfphone_label_driven_targets_match_reference(tmp_path, golden):
    """read_label / get_halfphone_stats / get_norm_durations / silence helpers against what the
    reference's own functions returned for the same synthetic state-aligned label
    (tools/make_golden.py, halfphone_* keys)."""
    import re
    from snickery_amd import hostprep as hp
    labfile = os.path.join(str(tmp_path), 'utt.lab')
    with open(labfile, 'w') as f:
        f.write(golden['halfphone_label_text'].item().decode())
    labs = hp.read_label(labfile, re.compile(golden['halfphone_regex'].item().decode()))
    assert np.array_equal(np.array([t for t, _ in labs]), golden['halfphone_label_times'])
    assert [q for _, q in labs] == [[x.decode() for x in row] for row in golden['halfphone_label_fields']]
    speech = golden['halfphone_speech']
    for rep in ('onepoint', 'twopoint', 'threepoint'):
        names, feats, timings = hp.get_halfphone_stats(speech, labs, representation_type=rep)
        assert np.array_equal(feats, golden['halfphone_features_' + rep])
        assert list(names) == [n.decode() for n in golden['halfphone_names']]
        assert np.array_equal(np.array(timings), golden['halfphone_timings'])
    stats = dict(zip([m.decode() for m in golden['halfphone_duration_monophones']],
                     [tuple(r) for r in golden['halfphone_duration_stats']]))
    nd = hp.get_norm_durations(names, timings, stats)
    assert nd.shape == (len(names), 1) and np.array_equal(nd, golden['halfphone_norm_durations'])
    supp = hp.suppress_weird_festival_pauses(labs)
    assert [q for _, q in supp] == [[x.decode() for x in row] for row in golden['halfphone_suppressed_fields']]
    assert any('pau' in q for _, q in supp) and not any('B_150' in q for _, q in supp)
    lo, hi = golden['halfphone_trimmed_range']
    assert np.array_equal(hp.reinsert_terminal_silence(speech[lo:hi], labs), golden['halfphone_reinserted_silence'])
    # error behaviour of the reference: state count not a multiple of five, unknown representation
    import pytest
    with pytest.raises(AssertionError):
        hp.get_halfphone_stats(speech, labs[:-1])
    with pytest.raises(ValueError):
        hp.get_halfphone_stats(speech, labs, representation_type='fourpoint')


def test_balance_stream_weights_matches_reference_loop(golden):
    """snickery_amd.balance_stream_weights against the trajectory the REFERENCE's own script printed
    when driven by the same deterministic stand-in Synthesiser (tools/make_golden.py, bsw_* keys)."""
    from snickery_amd.balance_stream_weights import balance_stream_weights, mean_nonzero_contributions
    from bsw_stub import StubSynthesiser
    log = []
    res = balance_stream_weights(StubSynthesiser(), report=log.append)
    assert np.allclose(res['losses'], golden['bsw_losses'], rtol=1e-12, atol=0)
    traj = np.array(res['weight_history'])
    assert traj.shape == golden['bsw_weight_trajectory'].shape
    assert np.allclose(traj, golden['bsw_weight_trajectory'], rtol=0, atol=5.1e-7)       # printed with %f
    assert np.array_equal(res['join_stream_weights'], golden['bsw_join_stream_weights'])
    assert np.array_equal(res['target_stream_weights'], golden['bsw_target_stream_weights'])
    assert any('loss approaching 0' in str(l) for l in log)
    # columns without any positive entry contribute 0.0 (no division by zero)
    m = mean_nonzero_contributions(np.array([[0.0, 2.0], [0.0, 4.0]]), np.array([[1.0], [0.0]]))
    assert list(m) == [0.0, 3.0, 1.0]


def test_database_writer_reproduces_reference_database(tmp_path, golden):
    """snickery_amd.train_simple on the regenerated synthetic voice (same seeded generator that
    tools/make_golden.py fed to the REFERENCE's train_simple.main_work): every array of the unit
    database is bit-identical to what the reference wrote, names / dtypes / shapes included."""
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from make_golden import write_voice
    from voice_fixture import CFG
    from snickery_amd import hostprep as hp, train_simple
    data = os.path.join(str(tmp_path), 'voice')
    write_voice(data, np.random.RandomState(20240))
    cfgfile = os.path.join(str(tmp_path), 'voice.cfg')
    with open(cfgfile, 'w') as f:
        f.write(CFG % dict(workdir=os.path.join(str(tmp_path), 'work'), data=data, greedy='True', multiepoch=6,
                           n_candidates=12))
    config = hp.load_config(cfgfile)
    dbfile = train_simple.main_work(config, overwrite_existing_data=False, report=lambda *_: None)
    assert os.path.basename(dbfile) == str(golden['db_basename'])
    db = hp.load_database(dbfile)
    assert sorted(db.keys()) == sorted(k.decode() if isinstance(k, bytes) else str(k) for k in golden['hdf5_keys'])
    for key, gkey in [('train_unit_features', 'F_unw'), ('join_contexts', 'JC_unw'), ('mean_target', 'mean_target'),
                      ('std_target', 'std_target'), ('mean_join', 'mean_join'), ('std_join', 'std_join')]:
        assert db[key].dtype == np.float32 and db[key].shape == golden[gkey].shape
        assert np.array_equal(db[key], golden[gkey]), key
    n = db['train_unit_features'].shape[0]
    assert db['join_contexts'].shape[0] == n + 1 and db['std_target'].shape == (1, 61)
    assert db['train_unit_names'].dtype == np.dtype('S50') and set(db['train_unit_names']) == {b'_'}
    assert db['unit_index_within_sentence_dset'].dtype == np.int32 and db['unit_index_within_sentence_dset'][0] == 0
    assert not any(b'arctic_b' in fn for fn in db['filenames'])          # test material is held out
    # existing data is protected unless overwriting is asked for (train_simple.py:33-37)
    import pytest
    with pytest.raises(SystemExit):
        train_simple.main_work(config, overwrite_existing_data=False, report=lambda *_: None)
    train_simple.main_work(config, overwrite_existing_data=True, report=lambda *_: None)


@pytest.mark.parametrize('tag,rep,duration', [('epoch', 'epoch', False), ('twopoint', 'twopoint', True),
                                              ('threepoint', 'threepoint', False)])
def test_halfphone_database_writer_reproduces_reference_database(tmp_path, tag, rep, duration):
    """snickery_amd.train_halfphone on the regenerated pitch-synchronous corpus (the same seeded
    generator that tools/make_golden.py fed to the REFERENCE's train_halfphone.main_work): every
    dataset of the database has the reference's name, shape, dtype and bytes (sha256) -- epoch voice
    (pairs of join frames, pitch-mark triples), halfphone voices with and without the duration
    target, including the reference's placement of the final join row."""
    import hashlib
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import voice_fixture
    from snickery_amd import hostprep as hp, train_halfphone
    ref = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_trainhp.npz'), allow_pickle=True)
    data = os.path.join(str(tmp_path), 'corpus')
    voice_fixture.write_halfphone_corpus(data)
    cfgfile = voice_fixture.halfphone_corpus_config(os.path.join(str(tmp_path), 'hp.cfg'), os.path.join(str(tmp_path), 'work'),
                                                    data, rep, duration)
    config = hp.load_config(cfgfile)
    dbfile = train_halfphone.main_work(config, report=lambda *_: None)
    assert os.path.basename(dbfile) == str(ref[tag + '_db_basename'])
    db = hp.load_database(dbfile)
    assert sorted(db.keys()) == sorted(k.decode() for k in ref[tag + '_keys'])
    for key, arr in db.items():
        arr = np.asarray(arr)
        assert list(arr.shape) == list(ref['%s_%s_shape' % (tag, key)]), key
        assert arr.dtype.str == str(ref['%s_%s_dtype' % (tag, key)]), key
        if arr.ndim == 2:
            assert np.array_equal(arr[[0, arr.shape[0] // 2, -1], :8], ref['%s_%s_rows' % (tag, key)]), key
        assert hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest() == str(ref['%s_%s_sha256' % (tag, key)]), key
    with pytest.raises(SystemExit):                       # existing data is protected (train_halfphone.py:78-82)
        train_halfphone.main_work(config, report=lambda *_: None)


@pytest.mark.parametrize('tag,rep,duration', [('twopoint', 'twopoint', True), ('threepoint', 'threepoint', False)])
def test_halfphone_writer_dump_join_data_reproduces_reference_files(tmp_path, tag, rep, duration):
    """dump_join_data (train_halfphone.py:277-282, get_join_data_AL :1207-1242): the second file the reference writes for
    active_learning_join.py -- name, datasets, shapes, dtypes and bytes (sha256) equal to what the reference itself wrote
    on the same corpus (tools/make_golden_joindata.py), and the voice file beside it unchanged by the option."""
    import hashlib
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import voice_fixture
    from snickery_amd import hostprep as hp, train_halfphone
    ref = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_joindata.npz'), allow_pickle=True)
    data = os.path.join(str(tmp_path), 'corpus')
    voice_fixture.write_halfphone_corpus(data)
    cfgfile = voice_fixture.halfphone_corpus_config(os.path.join(str(tmp_path), 'hp.cfg'), os.path.join(str(tmp_path), 'work'),
                                                    data, rep, duration)
    config = hp.load_config(cfgfile)
    config['dump_join_data'] = True
    config['join_cost_halfwidth'] = int(ref[tag + '_halfwidth'])
    dbfile = train_halfphone.main_work(config, report=lambda *_: None)
    for kind, fname in (('join', train_halfphone.get_data_dump_name(config, joindata=True)), ('voice', dbfile)):
        assert os.path.basename(fname) == str(ref['%s_%s_basename' % (tag, kind)])
        db = hp.load_database(fname)
        assert sorted(db.keys()) == sorted(k.decode() for k in ref['%s_%s_keys' % (tag, kind)])
        for key, arr in db.items():
            arr = np.asarray(arr)
            assert list(arr.shape) == list(ref['%s_%s_%s_shape' % (tag, kind, key)]), key
            assert arr.dtype.str == str(ref['%s_%s_%s_dtype' % (tag, kind, key)]), key
            if kind == 'join':
                assert np.array_equal(arr[[0, arr.shape[0] // 2, -1], :8], ref['%s_%s_%s_rows' % (tag, kind, key)]), key
            assert hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest() == str(ref['%s_%s_%s_sha256' % (tag, kind, key)]), (kind, key)
    config['target_representation'] = 'epoch'
    with pytest.raises(NotImplementedError):
        train_halfphone.build_database(config, report=lambda *_: None)


def test_writers_store_full_magphase_reproduces_reference_voices(tmp_path):
    """store_full_magphase (train_simple.py:145-149,260-299; train_halfphone.py:269-273,504-543): both writers on corpora
    with `<stream>_full` analysis files -- every dataset of the voice, the four mp_* arrays included, has the name, shape,
    dtype and bytes (sha256) of what the reference itself wrote (tools/make_golden_fullmag.py); files that do not hold
    one row per unit are refused."""
    import hashlib
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import voice_fixture
    from make_golden import write_voice
    from snickery_amd import hostprep as hp, train_halfphone, train_simple
    ref = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_fullmag.npz'), allow_pickle=True)

    def check(tag, dbfile):
        assert os.path.basename(dbfile) == str(ref[tag + '_basename'])
        db = hp.load_database(dbfile)
        assert sorted(db.keys()) == sorted(k.decode() for k in ref[tag + '_keys'])
        assert {'mp_mag', 'mp_imag', 'mp_real', 'mp_fz'} <= set(db.keys())
        for key, arr in db.items():
            arr = np.asarray(arr)
            assert list(arr.shape) == list(ref['%s_%s_shape' % (tag, key)]), key
            assert arr.dtype.str == str(ref['%s_%s_dtype' % (tag, key)]), key
            assert hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest() == str(ref['%s_%s_sha256' % (tag, key)]), key

    data = os.path.join(str(tmp_path), 'voice')
    write_voice(data, np.random.RandomState(20240))
    names = sorted(f[:-4] for f in os.listdir(os.path.join(data, 'low', 'mag')))
    voice_fixture.write_full_magphase_for_writers(os.path.join(data, 'high'), os.path.join(data, 'low'), names, 2)
    cfgfile = os.path.join(str(tmp_path), 'voice.cfg')
    with open(cfgfile, 'w') as f:
        f.write(voice_fixture.CFG % dict(workdir=os.path.join(str(tmp_path), 'work_simple'), data=data, greedy='True', multiepoch=6,
                                         n_candidates=12))
        f.write("store_full_magphase = True\nfull_magphase_dir = data + '/high/'\n")
    config = hp.load_config(cfgfile)
    dbfile = train_simple.main_work(config, report=lambda *_: None)
    check('simple', dbfile)
    # the reader's side (synth_simple.py:100-104, concatenateMagPhaseEpoch :655-674): frames of the selected units
    db = hp.load_database(dbfile)
    first = names[0]
    full = dict((e, hp.get_speech(os.path.join(data, 'high', e + '_full', first + '.' + e), 1 if e == 'f0' else 513)) for e in ('mag', 'imag', 'real', 'f0'))
    path = [5, 0, 17, 5]                                     # units of the first utterance: unit k <-> row k + 1 of its files
    mag, real, imag, fz = hp.gather_stored_magphase(db['mp_mag'], db['mp_imag'], db['mp_real'], db['mp_fz'], path)
    for got, e in ((mag, 'mag'), (real, 'real'), (imag, 'imag'), (fz, 'f0')):
        assert np.array_equal(got, full[e][[k + 1 for k in path], :]), e
    assert hp.gather_stored_magphase(db['mp_mag'], db['mp_imag'], db['mp_real'], db['mp_fz'], path, fzero=np.ones((4, 1)))[3].sum() == 4.0

    data = os.path.join(str(tmp_path), 'corpus')
    names = voice_fixture.write_halfphone_corpus(data)
    voice_fixture.write_full_magphase_for_writers(os.path.join(data, 'high'), os.path.join(data, 'low'), names, 0)
    cfgfile = voice_fixture.halfphone_corpus_config(os.path.join(str(tmp_path), 'hp.cfg'), os.path.join(str(tmp_path), 'work_hp'),
                                                    data, 'epoch', False)
    config = hp.load_config(cfgfile)
    config['store_full_magphase'] = True
    config['full_magphase_dir'] = data + '/high/'
    check('hpepoch', train_halfphone.main_work(config, report=lambda *_: None))
    config['target_representation'] = 'twopoint'           # a halfphone voice has fewer units than analysis frames
    with pytest.raises(ValueError):
        train_halfphone.build_database(config, report=lambda *_: None)


def test_hdf5_voice_without_h5py(tmp_path):
    """The reference keeps its voice in HDF5 (train_simple.py:95-149, read back at synth_simple.py:72-106).
    This interpreter has no h5py: the file is read and written through libhdf5's C API
    (snickery_amd.hdf5_io).  The database writers above therefore produced REAL HDF5 files -- checked here
    by their signature -- whose datasets come back with the reference's bytes; and a file written the
    reference's way (h5py: resizable, chunked, fixed-length strings) reads the same."""
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import voice_fixture
    from snickery_amd import hdf5_io, hostprep as hp, train_halfphone
    if not hdf5_io.available():
        pytest.skip('no libhdf5 in this image')
    data = os.path.join(str(tmp_path), 'corpus')
    voice_fixture.write_halfphone_corpus(data)
    cfgfile = voice_fixture.halfphone_corpus_config(os.path.join(str(tmp_path), 'hp.cfg'), os.path.join(str(tmp_path), 'work'),
                                                    data, 'epoch', False)
    dbfile = train_halfphone.main_work(hp.load_config(cfgfile), report=lambda *_: None)
    assert os.path.isfile(dbfile) and not os.path.isfile(dbfile + '.npz')         # the HDF5 itself, no sidecar
    with open(dbfile, 'rb') as f:
        assert f.read(8) == b'\x89HDF\r\n\x1a\n'
    db = hp.load_database(dbfile)
    again = str(tmp_path / 'again.hdf5')
    hdf5_io.write_datasets(again, db)
    back = hdf5_io.read_datasets(again)
    assert sorted(back) == sorted(db)
    for k in db:
        assert back[k].dtype == db[k].dtype and np.array_equal(back[k], db[k]), k
    # a file written the reference's way (h5py, maxshape=(None, d)), when an interpreter with h5py is around
    conda = '/opt/conda/bin/python3.9'
    if os.path.exists(conda):
        other = str(tmp_path / 'ref_style.hdf5')
        code = ("import h5py, numpy as np\n"
                "f = h5py.File(%r, 'w')\n"
                "d = f.create_dataset('train_unit_features', (7, 5), maxshape=(None, 5), dtype='f'); d[:, :] = np.arange(35).reshape(7, 5)\n"
                "n = f.create_dataset('train_unit_names', (7,), maxshape=(None,), dtype='|S50'); n[:] = np.array(['a/b_%%d' %% i for i in range(7)]).astype('S50')\n"
                "c = f.create_dataset('cutpoints', (7, 3), maxshape=(None, 3), dtype='i'); c[:, :] = np.arange(21).reshape(7, 3)\n"
                "f.close()\n" % other)
        if subprocess.run([conda, '-c', code]).returncode == 0:
            got = hp.load_database(other)
            assert np.array_equal(got['train_unit_features'], np.arange(35, dtype=np.float32).reshape(7, 5))
            assert got['train_unit_names'].dtype == np.dtype('S50') and got['train_unit_names'][3] == b'a/b_3'
            assert got['cutpoints'].dtype == np.int32 and got['cutpoints'][6, 2] == 20
