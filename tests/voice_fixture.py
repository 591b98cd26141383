"""Builds a temporary Snickery voice directory (config + unit DB sidecar + target stream files)
from the committed golden fixture, so that the front end can be driven exactly like the reference."""
import os
import numpy as np

CFG = '''
workdir = %(workdir)r
data = %(data)r
join_datadirs = [data + '/low/']
target_datadirs = join_datadirs
test_data_dirs = join_datadirs
test_patterns = ['arctic_b']
n_train_utts = 100
datadims = {'lf0':1, 'mag': 60, 'real': 45, 'imag': 45}
stream_list_join = ['mag', 'real', 'imag', 'lf0']
datadims_join = datadims
stream_list_target = ['mag', 'lf0']
datadims_target = datadims
frameshift_ms = 5
sample_rate = 16000
weight_target_data = True
weight_join_data = True
target_stream_weights = [0.1, 1.0]
join_stream_weights = [1.0 / float(len(stream_list_join))]  *  len(stream_list_join)
join_cost_weight = 0.2
target_representation = 'epoch'
n_test_utts = 3
greedy_search = %(greedy)s
search_epsilon = 0.0
multiepoch = %(multiepoch)d
last_frame_as_target = False
get_selection_info = False
hold_waves_in_memory = False
preload_all_magphase_utts = False
n_candidates = %(n_candidates)d
preselection_method = 'acoustic'
join_cost_type = 'pitch_sync'
'''


def build_voice(tmpdir, golden, greedy=True, multiepoch=6, n_candidates=12, extra_config='', db_override=None):
    from snickery_amd import hostprep as hp
    workdir = os.path.join(str(tmpdir), 'work')
    data = os.path.join(str(tmpdir), 'voice')
    os.makedirs(os.path.join(workdir, 'data_dumps'), exist_ok=True)
    for stream in ('mag', 'lf0', 'real', 'imag'):
        os.makedirs(os.path.join(data, 'low', stream), exist_ok=True)
    for stream in ('mag', 'lf0'):
        golden['test0_raw_' + stream].astype(np.float32).tofile(
            os.path.join(data, 'low', stream, 'arctic_b0001.' + stream))
    cfgfile = os.path.join(str(tmpdir), 'voice_%s_me%d.cfg' % ('greedy' if greedy else 'viterbi', multiepoch))
    with open(cfgfile, 'w') as f:
        f.write(CFG % dict(workdir=workdir, data=data, greedy=str(bool(greedy)), multiepoch=multiepoch,
                           n_candidates=n_candidates))
        f.write(extra_config)
    config = hp.load_config(cfgfile)
    N = golden['F_unw'].shape[0]
    db = dict(train_unit_features=golden['F_unw'], join_contexts=golden['JC_unw'],
              mean_target=golden['mean_target'], std_target=golden['std_target'],
              mean_join=golden['mean_join'], std_join=golden['std_join'],
              train_unit_names=np.array(['_'] * N).astype('S50'),
              filenames=np.array(['arctic_a%04d' % (1 + i // 150) for i in range(N)]).astype('S50'),
              unit_index_within_sentence_dset=(np.arange(N) % 150).astype(np.int32))
    db.update(db_override or {})
    np.savez(hp.get_data_dump_name(config) + '.npz', **db)
    return cfgfile, config


PHONES = ['a', 'b', 'k', 's', '#']


def build_halfphone_voice(tmpdir, golden, preselection_method, n_candidates=10, seed=5):
    """A label-driven halfphone voice (twopoint targets + normalised duration, quinphone unit
    names) assembled from the golden frame database, plus a state-aligned label for the test
    utterance.  Returns (cfgfile, config, db arrays)."""
    rng = np.random.RandomState(seed)
    F, JC = golden['F_unw'], golden['JC_unw']
    n_units = (F.shape[0] - 1) // 2
    feats = np.hstack([F[0:2 * n_units:2], F[1:2 * n_units:2], rng.randn(n_units, 1).astype(np.float32)])
    names = []
    for i in range(n_units):
        ctx = [PHONES[rng.randint(len(PHONES))] for _ in range(5)]
        ctx[2] += '_L' if i % 2 == 0 else '_R'
        names.append('/'.join(ctx))
    monos = sorted(set(n.split('/')[2] for n in names))[:-1]          # one phone left to oov_stats
    override = dict(train_unit_features=feats.astype(np.float32), join_contexts=JC[:n_units + 1],
                    train_unit_names=np.array(names).astype('S50'),
                    filenames=np.array(['arctic_a0001'] * n_units).astype('S50'),
                    unit_index_within_sentence_dset=np.arange(n_units).astype(np.int32),
                    duration_monophones=np.array(monos).astype('S10'),
                    duration_stats=np.column_stack([4.0 + rng.rand(len(monos)) * 4, 1.0 + rng.rand(len(monos))]))
    labdir = os.path.join(str(tmpdir), 'lab')
    os.makedirs(labdir, exist_ok=True)
    extra = '''
target_representation = 'twopoint'
add_duration_as_target = True
duration_target_weight = 0.3
target_duration_stretch_factor = 1.1
quinphone_regex = r'([^~]+)~([^-]+)-([^\\+]+)\\+([^\\=]+)\\=([^:]+)'
lab_extension = 'lab'
test_lab_dir = %r
preselection_method = %r
suppress_weird_festival_pauses = True
''' % (labdir, preselection_method)
    cfgfile, config = build_voice(tmpdir, golden, greedy=False, multiepoch=1, n_candidates=n_candidates,
                                  extra_config=extra, db_override=override)
    n_frames = golden['test0_raw_mag'].shape[0]
    seq = ['xx', 'xx', '#'] + [PHONES[rng.randint(4)] for _ in range(9)] + ['#', 'xx', 'xx']
    if preselection_method == 'quinphone':
        seq[6] = 'B_150'                              # becomes 'pau', a phone the database lacks
    lines, now = [], 0
    n_states = 5 * (len(seq) - 4)
    durs = rng.multinomial(n_frames + 2 - n_states, np.ones(n_states) / n_states) + 1   # label 2 frames longer
    for i in range(2, len(seq) - 2):
        for state in range(2, 7):
            dur = int(durs[(i - 2) * 5 + state - 2]) * 50000
            lines.append('%d %d %s~%s-%s+%s=%s:/x[%d]' % (now, now + dur, seq[i - 2], seq[i - 1], seq[i],
                                                       seq[i + 1], seq[i + 2], state))
            now += dur
    with open(os.path.join(labdir, 'arctic_b0001.lab'), 'w') as f:
        f.write('\n'.join(lines) + '\n')
    return cfgfile, config, override


def write_full_spectra(root, frames_per_utt, H, seed=77):
    """Synthetic 'high/' analysis data (full_magphase_dir): per utterance mag / real / imag (frames, H)
    and f0 (frames, 1) with unvoiced stretches (f0 = 0), float32 files in the reference's layout.
    Shared by tools/make_golden.py (input of the reference's concatenation) and the tests."""
    rng = np.random.RandomState(seed)
    for stream in ('mag', 'real', 'imag', 'f0'):
        os.makedirs(os.path.join(root, stream), exist_ok=True)
    for name, n in frames_per_utt:
        for stream in ('mag', 'real', 'imag'):
            (rng.rand(n, H) * 2.0 + (0.5 if stream == 'mag' else -1.0)).astype(np.float32).tofile(
                os.path.join(root, stream, name + '.' + stream))
        f0 = 120.0 + 30.0 * np.sin(np.arange(n) / 9.0) + rng.rand(n)
        pos = 0
        while pos < n:
            seg = int(rng.randint(4, 25))
            if rng.rand() < 0.35:
                f0[pos:pos + seg] = 0.0
            pos += seg
        f0.astype(np.float32).reshape(-1, 1).tofile(os.path.join(root, 'f0', name + '.f0'))
