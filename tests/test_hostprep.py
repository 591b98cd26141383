"""CPU tests of the host-side mirror of the reference interface (no GPU, no oracle needed):
target preparation and file naming against values captured from the reference itself."""
import os
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
