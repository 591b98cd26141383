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


def build_voice(tmpdir, golden, greedy=True, multiepoch=6, n_candidates=12):
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
    config = hp.load_config(cfgfile)
    N = golden['F_unw'].shape[0]
    np.savez(hp.get_data_dump_name(config) + '.npz',
             train_unit_features=golden['F_unw'], join_contexts=golden['JC_unw'],
             mean_target=golden['mean_target'], std_target=golden['std_target'],
             mean_join=golden['mean_join'], std_join=golden['std_join'],
             train_unit_names=np.array(['_'] * N).astype('S50'),
             filenames=np.array(['arctic_a%04d' % (1 + i // 150) for i in range(N)]).astype('S50'),
             unit_index_within_sentence_dset=(np.arange(N) % 150).astype(np.int32))
    return cfgfile, config
