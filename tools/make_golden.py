#!/opt/conda/bin/python3.9
"""
Golden-vector generator: runs the REAL reference (CSTR-Edinburgh/snickery,
mounted read-only at /root/reference) in this container and records inputs and
outputs of its search path as small .npz fixtures under tests/golden/.

  /opt/conda/bin/python3.9 tools/make_golden.py        # needs h5py (conda python)

What it does (SURVEY.md section 10 recipe):
  1. copies the reference's Python-2 modules to a TEMP dir and converts that
     copy with lib2to3 (nothing converted is ever written into this repo);
  2. applies the three py3 patches (integer division, bytes unit names);
  3. puts empty stub modules for magphase / libaudio / pywrapfst / pylab /
     smoothing.* / StashableKDTree first on sys.path (imported at module level
     by the reference but not needed by the search path);
  4. writes a tiny synthetic magphase-60 voice (raw float32 stream files);
  5. runs the converted train_simple.main_work  -> real HDF5 unit DB;
  6. runs the converted synth_simple.Synthesiser(cfg).synth_utt(...) with
     greedy_joint_search wrapped to record (unit_features, path) and to stop
     before waveform generation;
  7. injects the same arrays into the converted synth_halfphone.Synthesiser to
     record preselect_units_acoustic() and the join cost_cache built by
     make_on_the_fly_join_lattice_BLOCK_DIRECT() (the OpenFST compile step is
     replaced by a capture of its input dict -- OpenFST is not available).

Only data (inputs + expected outputs) is stored.  The fixtures are what
tests/test_oracle_golden.py pins the oracle against.
"""
import os
import sys
import shutil
import subprocess
import tempfile
import types
import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '..', 'tests', 'golden')

MODULES = ['synth_simple', 'synth_halfphone', 'fst_functions_wrapped', 'speech_manip',
           'data_manipulation', 'segmentaxis', 'const', 'matrix_operations',
           'file_naming', 'util', 'train_simple', 'train_halfphone', 'label_manip',
           'resample', 'resample_labels', 'mulaw2', 'varying_filter', 'data_fudging',
           'balance_stream_weights']

DIMS = {'mag': 60, 'real': 45, 'imag': 45, 'lf0': 1}


def convert_reference(tmp):
    src = os.path.join(tmp, 'conv')
    os.makedirs(src)
    for m in MODULES:
        shutil.copy(os.path.join(REF, 'script', m + '.py'), os.path.join(src, m + '.py'))
    files = [os.path.join(src, m + '.py') for m in MODULES]
    subprocess.check_call([sys.executable, '-m', 'lib2to3', '-w', '-n'] + files,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    def patch(fname, old, new, count=1):
        p = os.path.join(src, fname)
        s = open(p).read()
        assert s.count(old) >= count, (fname, old, s.count(old))
        s = s.replace(old, new)
        open(p, 'w').write(s)

    patch('synth_simple.py', 'm/multiepoch', 'm//multiepoch')
    patch('synth_halfphone.py', 'm/multiepoch', 'm//multiepoch')
    patch('synth_halfphone.py', ':n/2]', ':n//2]')
    patch('synth_halfphone.py', 'n/2:]', 'n//2:]')
    patch('synth_simple.py', 'extra_frames=overlap/2', 'extra_frames=overlap//2')
    patch('synth_simple.py', 'taper = overlap / 2', 'taper = overlap // 2')
    patch('train_simple.py', 'unit_names = np.array(unit_names)',
          "unit_names = np.array(unit_names).astype('S50')")
    patch('train_simple.py', 'filenames = [base] * m',
          "filenames = np.array([base] * m).astype('S50')")

    patch('train_halfphone.py', 'int(start) / 50000', 'int(start) // 50000')
    patch('train_halfphone.py', '(int(end) / 50000)', '(int(end) // 50000)')
    patch('train_halfphone.py', 'nphones = len(labels) / 5', 'nphones = len(labels) // 5')
    # main_work under py3 + h5py: integer unit count, byte strings for the S50 datasets
    patch('train_halfphone.py', 'n_halfphones = (n_states / 5) * 2', 'n_halfphones = (n_states // 5) * 2')
    patch('train_halfphone.py', 'phones_dset[start:start+m] = unit_names',
          "phones_dset[start:start+m] = np.array(unit_names).astype('S50')")
    patch('train_halfphone.py', 'filenames_dset[start:start+m] = filenames',
          "filenames_dset[start:start+m] = np.array(filenames).astype('S50')")
    patch('train_halfphone.py', 'f["duration_monophones"][:] = duration_monophones',
          'f["duration_monophones"][:] = duration_monophones.astype(\'S50\')')

    stubs = os.path.join(tmp, 'stubs')
    os.makedirs(os.path.join(stubs, 'smoothing'))
    for name in ['magphase', 'libaudio', 'pylab', 'StashableKDTree']:
        open(os.path.join(stubs, name + '.py'), 'w').write('')
    # pywrapfst (OpenFST) is not available.  The stand-in records the arc text that the reference's
    # lattice builders print into openfst.Compiler(); compile() returns an inert object.
    open(os.path.join(stubs, 'pywrapfst.py'), 'w').write('''
CAPTURED = []
class _Inert(object):
    def arcsort(self, **kw): return self
class Compiler(object):
    def __init__(self): self.text = []
    def write(self, s): self.text.append(s)
    def compile(self):
        CAPTURED.append(''.join(self.text))
        return _Inert()
''')
    for name in ['__init__', 'fft_feats', 'libwavgen', 'libaudio']:
        open(os.path.join(stubs, 'smoothing', name + '.py'), 'w').write('')
    sys.path.insert(0, src)
    sys.path.insert(0, stubs)
    return src


def write_voice(root, rng):
    """Synthetic magphase-60 stream files with speech-like continuity and ~30% UV."""
    names = ['arctic_a%04d' % i for i in range(1, 9)] + ['arctic_b0001', 'arctic_b0002']
    for stream in DIMS:
        os.makedirs(os.path.join(root, 'low', stream))
    for name in names:
        n = int(rng.randint(110, 190))
        base = np.cumsum(rng.randn(n, 8), axis=0) * 0.3
        for stream, dim in DIMS.items():
            proj = rng.randn(8, dim) if stream != 'lf0' else rng.randn(8, 1) * 0.05
            data = base.dot(proj) + 0.1 * rng.randn(n, dim)
            if stream == 'lf0':
                data = 5.0 + data
                uv = np.zeros(n, dtype=bool)
                pos = 0
                while pos < n:
                    seg = int(rng.randint(5, 30))
                    if rng.rand() < 0.3:
                        uv[pos:pos + seg] = True
                    pos += seg
                data[uv, 0] = 0.0
            data.astype(np.float32).tofile(os.path.join(root, 'low', stream, name + '.' + stream))
    return names


def write_config(path, workdir, data, multiepoch):
    ref_cfg = open(os.path.join(REF, 'config', 'slt_simplified_mini.cfg')).read()
    # the reference's own README demo config, re-pointed at the synthetic voice
    cfg = ref_cfg.replace("workdir = \n", "workdir = %r\n" % workdir)
    assert 'workdir = %r' % workdir in cfg
    cfg += "\n\n## ---- overrides appended by tools/make_golden.py ----\n"
    cfg += "data = %r\n" % data
    cfg += "join_datadirs = [data + '/low/']\n"
    cfg += "target_datadirs = join_datadirs\n"
    cfg += "test_data_dirs = join_datadirs\n"
    cfg += "test_patterns = ['arctic_b']\n"
    cfg += "n_train_utts = 100\n"
    cfg += "search_epsilon = 0.0\n"
    cfg += "multiepoch = %d\n" % multiepoch
    open(path, 'w').write(cfg)


class StopAfterSearch(Exception):
    pass


def main():
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix='snk_golden_')
    try:
        convert_reference(tmp)
        rng = np.random.RandomState(20240)
        data = os.path.join(tmp, 'voice')
        names = write_voice(data, rng)
        workdir = os.path.join(tmp, 'work')
        os.makedirs(workdir)

        import h5py
        import train_simple
        import synth_simple

        fixtures = {}
        for me in (6, 1):
            cfgfile = os.path.join(tmp, 'mini_me%d.cfg' % me)
            write_config(cfgfile, workdir, data, me)
            config = {}
            exec(compile(open(cfgfile).read(), cfgfile, 'exec'), config)
            del config['__builtins__']
            if me == 6:
                train_simple.main_work(config, overwrite_existing_data=True)
            synth = synth_simple.Synthesiser(cfgfile)

            captured = []
            orig = synth.greedy_joint_search

            def wrapped(unit_features, start_state=-1, holdout=[]):
                path = orig(unit_features, start_state=start_state, holdout=holdout)
                captured.append((np.array(unit_features), np.array(path)))
                raise StopAfterSearch()

            synth.greedy_joint_search = wrapped
            for base in synth.get_sentence_set('test'):
                try:
                    synth.synth_utt(base, synth_type='test')
                except StopAfterSearch:
                    pass
            assert len(captured) == 2, len(captured)
            tree_data = np.array(synth.joint_tree.data)
            fixtures['greedy_me%d_tree_shape' % me] = np.array(tree_data.shape)
            fixtures['greedy_me%d_tree_dtype_is_f64' % me] = np.array(tree_data.dtype == np.float64)
            for i, (uf, path) in enumerate(captured):
                fixtures['greedy_me%d_utt%d_unit_features' % (me, i)] = uf
                fixtures['greedy_me%d_utt%d_path' % (me, i)] = path.astype(np.int64)
            # natural-path known answer (resynth_training_chunk, synth_simple.py:909-916)
            if me == 1:
                start = 37
                feats = synth.train_unit_features[start:start + 25, :]
                synth.greedy_joint_search = orig
                natural = orig(feats, start_state=start)
                fixtures['greedy_me1_natural_start'] = np.array(start)
                fixtures['greedy_me1_natural_path'] = np.array(natural, dtype=np.int64)
            if me == 6:
                # DB arrays exactly as the reference loaded them from its own HDF5
                fixtures['F_unw'] = synth.train_unit_features_unweighted
                fixtures['JC_unw'] = synth.join_contexts_unweighted
                fixtures['mean_target'] = synth.mean_vec_target
                fixtures['std_target'] = synth.std_vec_target
                fixtures['mean_join'] = synth.mean_vec_join
                fixtures['std_join'] = synth.std_vec_join
                fixtures['target_weight_vector'] = np.array(synth.target_weight_vector)
                fixtures['unit_end_data_rows'] = np.array(synth.unit_end_data[[0, 5, -1], :])
                fixtures['unit_start_data_rows'] = np.array(synth.unit_start_data[[0, 5, -1], :])
                fixtures['train_unit_features_w_rows'] = np.array(
                    synth_simple.weight(synth.train_unit_features_unweighted[[0, 7, -1]],
                                        synth.target_weight_vector))
                fixtures['target_stream_weights'] = np.array(config['target_stream_weights'])
                fixtures['join_stream_weights'] = np.array(config['join_stream_weights'])
                fixtures['join_cost_weight'] = np.array(config['join_cost_weight'])
                # raw target streams of the first test utterance (host-prep fixture)
                base = synth.get_sentence_set('test')[0]
                for stream in ('mag', 'lf0'):
                    fixtures['test0_raw_' + stream] = np.fromfile(
                        os.path.join(data, 'low', stream, base + '.' + stream),
                        dtype=np.float32).reshape(-1, DIMS[stream])
                dbfile = synth_simple.get_data_dump_name(config)
                fixtures['db_basename'] = np.array(os.path.basename(dbfile))
                with h5py.File(dbfile, 'r') as f:
                    fixtures['hdf5_keys'] = np.array(sorted(f.keys()))
                    fixtures['hdf5_std_target_shape'] = np.array(f['std_target'].shape)
                    fixtures['hdf5_mean_target_shape'] = np.array(f['mean_target'].shape)

        # ---- reconfigure_settings (synth_simple.py:776-830): the weight-tuning entry point.  A fresh
        # Synthesiser on the multiepoch-6 voice is reconfigured (stream weights, join cost weight,
        # multiepoch 4, truncated streams) and searched again: description text, the target features it
        # prepares under the new settings and the greedy path ----
        reconf = {}
        cfgfile = os.path.join(tmp, 'mini_reconf.cfg')
        write_config(cfgfile, workdir, data, 6)
        with open(cfgfile, 'a') as f:
            f.write("truncate_target_streams = [-1, -1]\ntruncate_join_streams = [-1, -1, -1, -1]\n")
        synth_r = synth_simple.Synthesiser(cfgfile)
        new_settings = dict(join_stream_weights=[0.4, 0.3, 0.2, 0.1], target_stream_weights=[0.3, 0.7],
                            join_cost_weight=0.35, search_epsilon=0.0, multiepoch=4, magphase_use_target_f0=True,
                            magphase_overlap=2, truncate_target_streams=[40, -1], truncate_join_streams=[30, -1, 20, 1])
        desc = synth_r.reconfigure_settings(new_settings)
        captured_r = []
        orig_r = synth_r.greedy_joint_search

        def wrapped_r(unit_features, start_state=-1, holdout=[]):
            path = orig_r(unit_features, start_state=start_state, holdout=holdout)
            captured_r.append((np.array(unit_features), np.array(path)))
            raise StopAfterSearch()

        synth_r.greedy_joint_search = wrapped_r
        try:
            synth_r.synth_utt(synth_r.get_sentence_set('test')[0], synth_type='test')
        except StopAfterSearch:
            pass
        reconf['reconf_description'] = np.array(desc.encode())
        reconf['reconf_settings_json'] = np.array(__import__('json').dumps(new_settings).encode())
        reconf['reconf_unit_features'] = captured_r[0][0]
        reconf['reconf_path'] = captured_r[0][1].astype(np.int64)
        reconf['reconf_tree_shape'] = np.array(np.array(synth_r.joint_tree.data).shape)
        reconf['reconf_trace_lines'] = np.array('\n'.join(synth_r.get_path_information_epoch(captured_r[0][0], captured_r[0][1])).encode())
        reconf['reconf_filenames'] = np.array(synth_r.train_filenames).astype('S50')
        reconf['reconf_unit_index'] = np.array(synth_r.unit_index_within_sentence).astype(np.int32)
        np.savez_compressed(os.path.join(OUT, 'reference_reconf.npz'), **reconf)

        # ---- preselect + join lattice from converted synth_halfphone ----
        import synth_halfphone
        import scipy.spatial
        sh = synth_halfphone.Synthesiser.__new__(synth_halfphone.Synthesiser)
        K = 12
        sh.config = {'n_candidates': K, 'target_representation': 'epoch',
                     'join_cost_type': 'pitch_sync'}
        sh.verbose = False
        F_unw = fixtures['F_unw']
        JC_unw = fixtures['JC_unw']
        wt = fixtures['target_weight_vector']
        jw = np.array(config['join_stream_weights']) * config['join_cost_weight']
        wj = np.concatenate([[w] * DIMS[s] for w, s in zip(jw, config['stream_list_join'])])
        sh.train_unit_features = synth_halfphone.weight(F_unw, wt)
        jcw = synth_halfphone.weight(JC_unw, wj)
        sh.unit_end_data = jcw[1:, :]
        sh.unit_start_data = jcw[:-1, :]
        sh.tree = scipy.spatial.cKDTree(sh.train_unit_features, leafsize=100,
                                        compact_nodes=False, balanced_tree=False)  # synth_halfphone.py:379
        uf = fixtures['greedy_me6_utt0_unit_features'][:40]
        cand, dist = sh.preselect_units_acoustic(uf)
        cand = np.array(cand)
        # exercise the reference's exclusions: unit 0, unit N-1 and padding -1
        cand_edit = cand.copy()
        cand_edit[3, 2] = 0
        cand_edit[4, 1] = F_unw.shape[0] - 1
        cand_edit[5, 0] = -1
        cand_edit[6, 4] = cand_edit[5, 3] + 1      # a natural join (cost exactly 0)
        captured_cache = []
        synth_halfphone.cost_cache_to_compiled_fst = \
            lambda cost_cache, join_cost_weight=1.0: captured_cache.append(dict(cost_cache))
        sh.make_on_the_fly_join_lattice_BLOCK_DIRECT(cand_edit)
        cache = captured_cache[0]
        keys = np.array(sorted(cache.keys()), dtype=np.int64)
        vals = np.array([cache[tuple(k)] for k in keys], dtype=np.float64)
        fixtures['knn_K'] = np.array(K)
        fixtures['knn_queries'] = uf
        fixtures['knn_candidates'] = cand.astype(np.int64)
        fixtures['knn_distances'] = np.array(dist, dtype=np.float64)
        fixtures['join_candidates'] = cand_edit.astype(np.int64)
        fixtures['join_cache_keys'] = keys
        fixtures['join_cache_values'] = vals
        fixtures['join_weight_vector'] = wj

        # ---- quinphone preselection (synth_halfphone.py:1305-1354) with synthetic labels ----
        from label_manip import break_quinphone
        prng = np.random.RandomState(77)
        phones = ['a', 'b', 'k', 's', 'sil']
        N = F_unw.shape[0]
        names = []
        for i in range(N):
            ctx = [phones[prng.randint(len(phones))] for _ in range(5)]
            ctx[2] = ctx[2] + ('_L' if i % 2 == 0 else '_R')
            names.append('/'.join(ctx))
        sh.train_unit_names = np.array(names)
        sh.unit_index = {}
        for (i, quinphone) in enumerate(sh.train_unit_names):          # synth_halfphone.py:281-292
            mono, diphone, triphone, quinphone = break_quinphone(quinphone)
            for form in [mono, diphone, triphone, quinphone]:
                if form not in sh.unit_index:
                    sh.unit_index[form] = []
                sh.unit_index[form].append(i)
        qnames = [names[5], names[40], 'zz/zz/zz_L/zz/zz', 'a/b/k_R/s/a', names[300]]
        qfeats = fixtures['greedy_me6_utt0_unit_features'][:len(qnames)]
        sh.config['n_candidates'] = 9
        qc, qd = sh.preselect_units_quinphone(qfeats, qnames)
        fixtures['quin_unit_names'] = np.array(names).astype('S40')
        fixtures['quin_query_names'] = np.array(qnames).astype('S40')
        fixtures['quin_queries'] = qfeats
        fixtures['quin_candidates'] = np.array(qc, dtype=np.int64)
        fixtures['quin_distances'] = np.array(qd, dtype=np.float64)

        # ---- monophone-then-acoustic preselection (synth_halfphone.py:1369-1396): the per-phone trees and
        # index converters exactly as Synthesiser.__init__ builds them (:385-402), same labels ----
        import const
        sh.number_of_units = N
        sh.phonetrees = {}
        sh.phonetrees_index_converters = {}
        monophones = np.array([q.split(const.label_delimiter)[2] for q in sh.train_unit_names])
        for phone in dict(zip(monophones, monophones)):
            train = sh.train_unit_features[monophones == phone, :]
            sh.phonetrees[phone] = scipy.spatial.cKDTree(train, leafsize=10, compact_nodes=False, balanced_tree=False)
            sh.phonetrees_index_converters[phone] = np.arange(sh.number_of_units)[monophones == phone]
        mnames = [names[i] for i in (5, 40, 300, 7, 512, 1000, 3)]
        mfeats = fixtures['greedy_me6_utt0_unit_features'][10:10 + len(mnames)]
        mc, md = sh.preselect_units_monophone_then_acoustic(mfeats, mnames)
        presel = {'mono_query_names': np.array(mnames).astype('S40'), 'mono_queries': mfeats,
                  'mono_candidates': np.array(mc, dtype=np.int64), 'mono_distances': np.array(md, dtype=np.float64),
                  'mono_n_candidates': np.array(sh.config['n_candidates'])}
        # ---- the arc text of the two lattices as the reference's own builders print it into
        # openfst.Compiler() (fst_functions_wrapped.py:28-58, 172-217), for the candidates / distances /
        # cost cache recorded above ----
        import fst_functions_wrapped
        import pywrapfst
        del pywrapfst.CAPTURED[:]
        fst_functions_wrapped.make_target_sausage_lattice(fixtures['knn_distances'], fixtures['join_candidates'])
        fst_functions_wrapped.cost_cache_to_compiled_fst(cache)
        assert len(pywrapfst.CAPTURED) == 2
        presel['fst_target_text'] = np.array(pywrapfst.CAPTURED[0].encode())
        presel['fst_join_text'] = np.array(pywrapfst.CAPTURED[1].encode())

        # ---- per-stream scores along a path (synth_halfphone.py:1964-1981, 2977-3008), Viterbi and
        # greedy forms, on the reference's own weighted arrays ----
        sh.stream_list_target = config['stream_list_target']
        sh.datadims_target = config['datadims_target']
        sh.stream_list_join = config['stream_list_join']
        sh.datadims_join = config['datadims_join']
        spath = [int(v) for v in fixtures['knn_candidates'][:, 0]]
        sfeats = fixtures['knn_queries']
        presel['scores_path'] = np.array(spath, dtype=np.int64)
        presel['scores_target'] = np.array(sh.get_target_scores_per_stream(sfeats, spath))
        sh.config['greedy_search'] = False
        presel['scores_join_viterbi'] = np.array(sh.get_join_scores_per_stream(spath))
        sh.config['greedy_search'] = True
        sh.prev_join_rep = sh.unit_start_data
        sh.current_join_rep = sh.unit_end_data
        presel['scores_join_greedy'] = np.array(sh.get_join_scores_per_stream(spath))
        sh.config['greedy_search'] = False
        np.savez_compressed(os.path.join(OUT, 'reference_preselect.npz'), **presel)

        # ---- label-driven halfphone targets (synth_halfphone.py:1527-1549) on a synthetic
        # state-aligned label: read_label / get_halfphone_stats / get_norm_durations ----
        import re
        import train_halfphone
        regex_text = r'([^~]+)~([^-]+)-([^\+]+)\+([^\=]+)\=([^:]+)'
        regex = re.compile(regex_text)
        lrng = np.random.RandomState(99)
        seq = ['xx', 'xx', '#', 'h', 'e', 'B_150', 'l', 'ou', '#', 'xx', 'xx']
        lines, now = [], 0
        for i in range(2, len(seq) - 2):
            for state in range(2, 7):
                dur = int(lrng.randint(1, 9)) * 50000 + (20000 if state == 4 else 0)   # odd times exercise flooring
                lab = '%s~%s-%s+%s=%s:/A:1_2/B:3[%d]' % (seq[i - 2], seq[i - 1], seq[i], seq[i + 1], seq[i + 2], state)
                lines.append('%d %d %s' % (now, now + dur, lab))
                now += dur
        label_text = '\n'.join(lines) + '\n'
        labfile = os.path.join(tmp, 'utt.lab')
        open(labfile, 'w').write(label_text)
        labs = train_halfphone.read_label(labfile, regex)
        label_frames = labs[-1][0][1]
        hp_speech = lrng.randn(label_frames - 3, 7)          # 3 frames short: end clamping
        fixtures['halfphone_regex'] = np.array(regex_text.encode())
        fixtures['halfphone_label_text'] = np.array(label_text.encode())
        fixtures['halfphone_label_times'] = np.array([t for (t, q) in labs], dtype=np.int64)
        fixtures['halfphone_label_fields'] = np.array([q for (t, q) in labs]).astype('S8')
        fixtures['halfphone_speech'] = hp_speech
        for rep in ['onepoint', 'twopoint', 'threepoint']:
            hnames, hfeats, htimes = train_halfphone.get_halfphone_stats(hp_speech, labs, representation_type=rep)
            fixtures['halfphone_features_' + rep] = np.array(hfeats)
        htimes = list(htimes)
        fixtures['halfphone_names'] = np.array(hnames).astype('S40')
        fixtures['halfphone_timings'] = np.array(htimes, dtype=np.int64)
        dstats = {'h_L': (4.0, 1.5), 'h_R': (6.5, 2.0), 'e_L': (3.0, 0.5), '#_L': (10.0, 4.0), 'ou_R': (7.25, 3.0)}
        fixtures['halfphone_duration_monophones'] = np.array(sorted(dstats)).astype('S8')
        fixtures['halfphone_duration_stats'] = np.array([dstats[k] for k in sorted(dstats)], dtype=np.float64)
        fixtures['halfphone_norm_durations'] = train_halfphone.get_norm_durations(hnames, htimes, dstats)
        supp = synth_halfphone.suppress_weird_festival_pauses(labs)
        fixtures['halfphone_suppressed_fields'] = np.array([q for (t, q) in supp]).astype('S8')
        trimmed = hp_speech[labs[4][0][1]:labs[-5][0][0]]    # as if the terminal silences had been trimmed
        fixtures['halfphone_reinserted_silence'] = train_halfphone.reinsert_terminal_silence(trimmed, labs)
        fixtures['halfphone_trimmed_range'] = np.array([labs[4][0][1], labs[-5][0][0]], dtype=np.int64)

        # ---- waveform-side concatenation (synth_simple.py:538-747): retrieve_magphase_frag +
        # concatenateMagPhaseEpoch_sep_files up to the vocoder call, on synthetic analysis data with a
        # 17-bin spectrum (module constant patched at run time) ----
        sys.path.insert(0, os.path.join(HERE, '..', 'tests'))
        import voice_fixture
        H = 17
        synth_simple.FFTHALFLEN = H
        cfgfile6 = os.path.join(tmp, 'mini_me6.cfg')
        synth6 = synth_simple.Synthesiser(cfgfile6)
        # py3: the HDF5 byte strings must be str for the reference's path joins
        synth6.train_filenames = np.array([f.decode() if isinstance(f, bytes) else str(f) for f in synth6.train_filenames])
        utt_frames = []
        for base in sorted(set(fn.decode() if isinstance(fn, bytes) else str(fn) for fn in synth6.train_filenames)):
            n = int(np.sum(np.array([f.decode() if isinstance(f, bytes) else str(f) for f in synth6.train_filenames]) == base))
            utt_frames.append((base, n))
        voice_fixture.write_full_spectra(os.path.join(data, 'high'), utt_frames, H, seed=77)
        synth6.config['full_magphase_dir'] = os.path.join(data, 'high')
        captured_syn = []
        synth_simple.magphase.synthesis_from_lossless = lambda mag, real, imag, fz, sr: captured_syn.append(
            (np.array(mag), np.array(real), np.array(imag), np.array(fz))) or np.zeros(4)
        synth_simple.la.write_audio_file = lambda *a, **k: None
        n_units = len(synth6.train_filenames)
        uix = np.array(synth6.unit_index_within_sentence)
        starts = np.nonzero(uix == 0)[0]
        # a path with an utterance start (zero padding in front), windows that run past an utterance end
        # (zero padding behind), ordinary windows and an immediate repeat
        cpath = [int(starts[1]), int(starts[1]) + 6, int(starts[2]) - 3, int(starts[2]) - 6, 40, 46, 46, 300, int(starts[3]) - 1,
                 int(fixtures['greedy_me6_utt0_path'][5])]
        # (without overlap the reference asserts on windows that run past an utterance end)
        safe_path = [int(starts[1]), int(starts[1]) + 6, 40, 46, 46, 300, int(fixtures['greedy_me6_utt0_path'][5])]
        fixtures['concat_path_no_overlap'] = np.array(safe_path, dtype=np.int64)
        for ov in (2, 0, 4):
            synth6.concatenateMagPhaseEpoch_sep_files(cpath if ov else safe_path, os.path.join(tmp, 'x.wav'), overlap=ov)
            m_, r_, i_, f_ = captured_syn[-1]
            fixtures['concat_ov%d_mag' % ov] = m_
            fixtures['concat_ov%d_real' % ov] = r_
            fixtures['concat_ov%d_imag' % ov] = i_
            fixtures['concat_ov%d_fz' % ov] = f_
        fixtures['concat_path'] = np.array(cpath, dtype=np.int64)
        fixtures['concat_utt_frames'] = np.array([n for (b, n) in utt_frames], dtype=np.int64)
        fixtures['concat_utt_names'] = np.array([b for (b, n) in utt_frames]).astype('S50')
        fixtures['concat_filenames'] = np.array(synth6.train_filenames).astype('S50')
        fixtures['concat_unit_index'] = uix.astype(np.int32)

        # ---- the reference's balance_stream_weights.py (host logic only) driven by a deterministic
        # stand-in Synthesiser (tests/bsw_stub.py): recorded weight trajectory and result ----
        import io
        import runpy
        import contextlib
        sys.path.insert(0, os.path.join(HERE, '..', 'tests'))
        import bsw_stub
        real_synth_class = synth_halfphone.Synthesiser
        synth_halfphone.Synthesiser = bsw_stub.StubSynthesiser
        old_argv = sys.argv
        sys.argv = ['balance_stream_weights.py', '-c', 'unused.cfg']
        out = io.StringIO()
        try:
            with contextlib.redirect_stdout(out):
                runpy.run_path(os.path.join(tmp, 'conv', 'balance_stream_weights.py'), run_name='__main__')
        finally:
            sys.argv = old_argv
            synth_halfphone.Synthesiser = real_synth_class
        lines = out.getvalue().splitlines()
        traj = [[float(x) for x in l.split(':', 1)[1].split()] for l in lines if l.startswith('     weights:')]
        loss = [float(l.split('loss')[1].split('=')[0]) for l in lines if l.startswith('=== iteration')]
        jw = [l for l in lines if l.startswith('join_stream_weights = ')][0]
        tw = [l for l in lines if l.startswith('target_stream_weights = ')][0]
        fixtures['bsw_weight_trajectory'] = np.array(traj)            # printed with %f
        fixtures['bsw_losses'] = np.array(loss)
        fixtures['bsw_join_stream_weights'] = np.array(eval(jw.split('=', 1)[1]))
        fixtures['bsw_target_stream_weights'] = np.array(eval(tw.split('=', 1)[1]))

        # ---- database writer of train_halfphone.py (main_work, :63-628) on a small pitch-synchronous
        # corpus (tests/voice_fixture.write_halfphone_corpus): the reference's own HDF5 files, recorded
        # as (shape, dtype, sha256) per dataset plus a few rows, in a separate fixture file ----
        import hashlib
        hp_data = os.path.join(tmp, 'hp_corpus')
        voice_fixture.write_halfphone_corpus(hp_data)
        trainhp = {}
        for tag, rep, duration in (('epoch', 'epoch', False), ('twopoint', 'twopoint', True), ('threepoint', 'threepoint', False)):
            hp_work = os.path.join(tmp, 'hp_work_' + tag)
            os.makedirs(hp_work)
            hp_cfg = voice_fixture.halfphone_corpus_config(os.path.join(tmp, 'hp_%s.cfg' % tag), hp_work, hp_data, rep, duration)
            config = {}
            exec(compile(open(hp_cfg).read(), hp_cfg, 'exec'), config)
            del config['__builtins__']
            with contextlib.redirect_stdout(io.StringIO()):
                train_halfphone.main_work(config, overwrite_existing_data=True)
            dbfile = train_halfphone.get_data_dump_name(config)
            trainhp[tag + '_db_basename'] = np.array(os.path.basename(dbfile))
            with h5py.File(dbfile, 'r') as f:
                trainhp[tag + '_keys'] = np.array(sorted(f.keys())).astype('S40')
                for key in f.keys():
                    arr = f[key][...]
                    trainhp['%s_%s_shape' % (tag, key)] = np.array(arr.shape, dtype=np.int64)
                    trainhp['%s_%s_dtype' % (tag, key)] = np.array(arr.dtype.str)
                    trainhp['%s_%s_sha256' % (tag, key)] = np.array(hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest())
                    if arr.ndim == 2:
                        trainhp['%s_%s_rows' % (tag, key)] = arr[[0, arr.shape[0] // 2, -1], :8]
        np.savez_compressed(os.path.join(OUT, 'reference_trainhp.npz'), **trainhp)
        print('wrote tests/golden/reference_trainhp.npz (%d bytes)' % os.path.getsize(os.path.join(OUT, 'reference_trainhp.npz')))

        np.savez_compressed(os.path.join(OUT, 'reference_mini.npz'), **fixtures)
        sz = os.path.getsize(os.path.join(OUT, 'reference_mini.npz'))
        print('wrote tests/golden/reference_mini.npz (%d bytes), N=%d' % (sz, F_unw.shape[0]))
        for k in sorted(fixtures):
            print('  ', k, np.shape(fixtures[k]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
