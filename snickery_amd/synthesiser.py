"""Drop-in ``Synthesiser`` for the SEARCH path of Snickery's script/synth_simple.py and
script/synth_halfphone.py: same config files, same unit database, same method names,
arguments and return values -- the search itself runs on the MI355X through libsnkhip.so.

Scope: config load, database load, stream weighting/truncation, multiepoch
greedy search, acoustic / class-restricted preselection, join lattice + Viterbi, per-stream
scores, sentence selection, ``.trace``-style path output.  Waveform generation (magphase
concatenation, after the search) is outside the path and is NOT performed: ``synth_utt``
returns the selected unit path.

flavour='simple'    mirrors synth_simple.Synthesiser  (greedy only, synth_simple.py:48-185)
flavour='halfphone' mirrors synth_halfphone.Synthesiser with target_representation == 'epoch'
                    (greedy or preselect + Viterbi, synth_halfphone.py:151-415).
"""
import glob
import re
import os
import timeit

import numpy as np

from . import hostprep as hp
from .engine import HipSearchEngine

APPLY_JCW_ON_TOP = True          # synth_simple.py:43, synth_halfphone.py (same flag)
LABEL_DELIMITER = '/'            # const.py:6
VERY_BIG_WEIGHT_VALUE = 1000000000000000.0


class Synthesiser(object):

    def __init__(self, config_file, holdout_percent=0.0, flavour=None, device=0, verbose=True):
        self.mode_of_operation = 'normal'
        self.verbose = verbose
        self.config = hp.load_config(config_file)
        self.config_file = config_file
        if flavour is None:
            flavour = 'simple' if self.config.get('greedy_search', False) else 'halfphone'
        assert flavour in ('simple', 'halfphone')
        self.flavour = flavour

        self.stream_list_target = self.config['stream_list_target']
        self.stream_list_join = self.config['stream_list_join']
        self.datadims_target = self.config['datadims_target']
        self.datadims_join = self.config['datadims_join']
        self.target_representation = 'epoch' if flavour == 'simple' else self.config['target_representation']

        db = hp.load_database(hp.get_data_dump_name(self.config))
        self.train_unit_features_unweighted = np.asarray(db['train_unit_features'])
        self.train_unit_names = db['train_unit_names']
        self.train_filenames = db['filenames']
        self.mean_vec_target = db['mean_target']
        self.std_vec_target = db['std_target']
        self.mean_vec_join = db['mean_join']
        self.std_vec_join = db['std_join']
        self.join_contexts_unweighted = np.asarray(db['join_contexts'])
        self.unit_index_within_sentence = db.get('unit_index_within_sentence_dset')
        self.train_cutpoints = db.get('cutpoints')      # absent from train_simple DBs
        if self.config.get('store_full_magphase', False):  # synth_simple.py:100-104 = synth_halfphone.py:215-219
            self.mp_mag, self.mp_imag, self.mp_real, self.mp_fz = db['mp_mag'], db['mp_imag'], db['mp_real'], db['mp_fz']
        self.number_of_units = self.train_unit_features_unweighted.shape[0]
        if self.config.get('add_duration_as_target', False):
            # synth_halfphone.py:206-211
            monophones = [m.decode() if isinstance(m, bytes) else str(m) for m in db['duration_monophones']]
            stats = np.asarray(db['duration_stats'])
            self.duration_stats = dict(zip(monophones, zip(stats[:, 0].tolist(), stats[:, 1].tolist())))
        if self.target_representation != 'epoch':
            self.quinphone_regex = re.compile(self.config['quinphone_regex'])     # synth_halfphone.py:281-282

        self.holdout_percent = holdout_percent
        self.holdout_samples = 0
        if holdout_percent > 0.0:
            hs = int(self.number_of_units * (holdout_percent / 100.0))
            self.train_unit_features_unweighted_dev = self.train_unit_features_unweighted[-hs:, :]
            self.train_unit_features_unweighted = self.train_unit_features_unweighted[:-hs, :]
            self.train_unit_names_dev = self.train_unit_names[-hs:]
            self.train_unit_names = self.train_unit_names[:-hs]
            # the join rows of the held-out units are dropped with them (synth_simple.py:253-255)
            self.join_contexts_unweighted = self.join_contexts_unweighted[:-hs, :]
            self.number_of_units -= hs
            self.holdout_samples = hs

        # join layout: synth_halfphone doubles the join weight vector for an epoch database from
        # train_halfphone.py whose rows are [j_t, j_{t+1}] (synth_halfphone.py:693-695)
        dj = sum(self.datadims_join[s] for s in self.stream_list_join)
        self._double_join = (self.join_contexts_unweighted.shape[1] == 2 * dj)
        if not self._double_join and self.join_contexts_unweighted.shape[1] != dj:
            raise ValueError('join_contexts has %d columns, streams give %d' % (
                self.join_contexts_unweighted.shape[1], dj))

        self.engine = HipSearchEngine(device)      # raises: no CPU fallback
        self.engine.upload_db(self.train_unit_features_unweighted, self.join_contexts_unweighted)

        self._target_stream_w = None
        self._join_stream_w = None
        self._target_trunc = None
        self._join_trunc = None
        if APPLY_JCW_ON_TOP:
            self.set_target_weights(np.array(self.config['target_stream_weights']) * (1.0 - self.config['join_cost_weight']), _apply=False)
            self.set_join_weights(np.array(self.config['join_stream_weights']) * self.config['join_cost_weight'], _apply=False)
        else:
            self.set_target_weights(self.config['target_stream_weights'], _apply=False)
            self.set_join_weights(self.config['join_stream_weights'], _apply=False)
        if flavour == 'simple':
            if 'truncate_target_streams' in self.config:
                self.truncate_target_streams(self.config['truncate_target_streams'], _apply=False)
            if 'truncate_join_streams' in self.config:
                self.truncate_join_streams(self.config['truncate_join_streams'], _apply=False)
        self._apply_weights()

        self.first_silent_unit = 0
        if self.target_representation == 'epoch':
            self.config['preselection_method'] = 'acoustic'
        if flavour == 'simple':
            assert self.config['greedy_search'] is True
        if self.config.get('greedy_search', False):
            if self.config.get('multiple_search_trees', 1) > 1:
                raise SystemExit('multiple_search_trees not implemented yet -- try adjusting search_epsilon instead to speed up search')
            self.get_tree_for_greedy_search()
        elif self.config.get('preselection_method') == 'monophone_then_acoustic':
            self._setup_monophone_classes()

        self.test_data_target_dirs = hp.locate_stream_directories(self.config['test_data_dirs'], self.stream_list_target)
        if self.config.get('tune_data_dirs', ''):
            self.tune_data_target_dirs = hp.locate_stream_directories(self.config['tune_data_dirs'], self.stream_list_target)

    # ------------------------------------------------------------------ weights
    def set_join_weights(self, weights, _apply=True):
        """synth_simple.py:234-255 / synth_halfphone.py:682-707."""
        assert len(weights) == len(self.stream_list_join)
        vec = hp.stream_weight_vector(list(weights), self.stream_list_join, self.datadims_join)
        if self._double_join:
            vec = vec + vec
        self.join_weight_vector = np.array(vec, dtype=np.float64)
        if _apply:
            self._apply_weights()

    def set_target_weights(self, weights, _apply=True):
        """synth_simple.py:257-274 / synth_halfphone.py:713-737."""
        assert len(weights) == len(self.stream_list_target), (weights, self.stream_list_target)
        vec = hp.stream_weight_vector(list(weights), self.stream_list_target, self.datadims_target)
        vec = vec * hp.TARGET_REP_WIDTHS[self.target_representation]
        if self.config.get('add_duration_as_target', False):
            vec.append(self.config.get('duration_target_weight', 0.0))
        self.target_weight_vector = np.array(vec, dtype=np.float64)
        if _apply:
            self._apply_weights()

    def truncate_join_streams(self, truncation_values, _apply=True):
        """synth_simple.py:982-985.  Dropping columns == zero weight + zero query column: both
        add exactly +0.0 to every squared distance, in the same column order."""
        self._join_trunc = hp.get_selection_vector(self.stream_list_join, self.datadims_join, truncation_values)
        if _apply:
            self._apply_weights()

    def truncate_target_streams(self, truncation_values, _apply=True):
        """synth_simple.py:987-992."""
        self._target_trunc = hp.get_selection_vector(self.stream_list_target, self.datadims_target, truncation_values)
        self.target_truncation_vector = self._target_trunc
        if _apply:
            self._apply_weights()

    def _apply_weights(self):
        wt = self.target_weight_vector.copy()
        if self._target_trunc is not None:
            keep = np.zeros(wt.size, dtype=bool)
            keep[self._target_trunc] = True
            wt[~keep] = 0.0
        wj = self.join_weight_vector.copy()
        if self._join_trunc is not None:
            keep = np.zeros(wj.size, dtype=bool)
            n1 = len(hp.stream_weight_vector([1] * len(self.stream_list_join), self.stream_list_join, self.datadims_join))
            for off in ([0, n1] if self._double_join else [0]):
                keep[np.array(self._join_trunc) + off] = True
            wj[~keep] = 0.0
        self._device_wt, self._device_wj = wt, wj
        self.engine.set_weights(wt, wj)      # O(N*D) on device, no index rebuild

    def _mask_query(self, unit_features):
        if self._target_trunc is None:
            return unit_features
        q = np.zeros_like(unit_features)
        q[:, self._target_trunc] = unit_features[:, self._target_trunc]
        return q

    # ------------------------------------------------------------------ search
    def get_tree_for_greedy_search(self):
        """synth_simple.py:190-229 / synth_halfphone.py:539-609: nothing is built -- the engine
        addresses the windowed database in place."""
        self.engine.set_greedy_layout(self.config.get('multiepoch', 1),
                                      self.config.get('last_frame_as_target', False),
                                      1 if (self.flavour == 'halfphone' and self._double_join) else 0)

    def greedy_joint_search(self, unit_features, start_state=-1, holdout=[]):
        """synth_simple.py:458-503."""
        assert self.config['target_representation'] == 'epoch'
        t = self.start_clock('Greedy search')
        unit_features = np.asarray(unit_features, dtype=np.float64)
        if self._target_trunc is not None and unit_features.shape[1] == len(self._target_trunc):
            full = np.zeros((unit_features.shape[0], self.target_weight_vector.size))
            full[:, self._target_trunc] = unit_features
            unit_features = full
        path = self.engine.greedy(unit_features, start_state=start_state,
                                  search_epsilon=self.config.get('search_epsilon', 0.0))
        self.stop_clock(t)
        return path

    def preselect_units_acoustic(self, unit_features):
        """synth_halfphone.py:1359-1366."""
        t = self.start_clock('Acoustic select units ')
        candidates, distances = self.engine.knn(np.asarray(unit_features, dtype=np.float64), self.config['n_candidates'])
        self.stop_clock(t)
        return (candidates, distances)

    def _setup_monophone_classes(self):
        """Per-phone trees of synth_halfphone.py:385-402 become one class id per unit."""
        names = [n.decode() if isinstance(n, bytes) else str(n) for n in self.train_unit_names]
        monophones = [n.split(LABEL_DELIMITER)[2] for n in names]
        self.monophone_ids = dict((m, i) for i, m in enumerate(sorted(set(monophones))))
        self.engine.set_unit_classes(np.array([self.monophone_ids[m] for m in monophones], dtype=np.int32))

    def preselect_units_monophone_then_acoustic(self, unit_features, unit_names):
        """synth_halfphone.py:1369-1396."""
        t = self.start_clock('Preselect units ')
        if not hasattr(self, 'monophone_ids'):
            self._setup_monophone_classes()
        monophones = [q.split(LABEL_DELIMITER)[2] for q in unit_names]
        assert len(monophones) == unit_features.shape[0], (len(monophones), unit_features.shape[0])
        for phone in monophones:
            assert phone in self.monophone_ids, 'unseen monophone %s' % (phone)
        qc = np.array([self.monophone_ids[m] for m in monophones], dtype=np.int32)
        out = self.engine.knn_by_class(np.asarray(unit_features, dtype=np.float64), self.config['n_candidates'], qc)
        self.stop_clock(t)
        return out

    @staticmethod
    def break_quinphone(quinphone):
        """label_manip.py:16-32."""
        q = quinphone.split(LABEL_DELIMITER)
        assert len(q) == 5
        mono = q[2]
        tri = LABEL_DELIMITER.join(q[1:4])
        if mono.endswith('_L'):
            di = LABEL_DELIMITER.join(q[1:3])
        elif mono.endswith('_R'):
            di = LABEL_DELIMITER.join(q[2:4])
        else:
            raise SystemExit('efvaedvsdv')
        return (mono, di, tri, quinphone)

    def _setup_unit_index(self):
        """synth_halfphone.py:281-292: label -> unit ids, for every back-off level."""
        self.unit_index = {}
        for i, name in enumerate(self.train_unit_names):
            name = name.decode() if isinstance(name, bytes) else str(name)
            for form in self.break_quinphone(name):
                self.unit_index.setdefault(form, []).append(i)

    def preselect_units_quinphone(self, unit_features, unit_names, candidates=None):
        """synth_halfphone.py:1305-1354: label back-off lookup on the host, distances on the GPU.  `candidates`: the
        ids of an earlier lookup for the same names (they depend on the labels and n_candidates only, not on the stream
        weights -- synth_utts_bulk keeps them across the iterations of a tuning loop)."""
        if candidates is not None:
            t = self.start_clock('Compute target distances...')
            distances = self.engine.candidate_distances(np.asarray(unit_features, dtype=np.float64), candidates)
            self.stop_clock(t)
            return (candidates, distances)
        t = self.start_clock('Preselect units ')
        if not hasattr(self, 'unit_index'):
            self._setup_unit_index()
        K = self.config['n_candidates']
        candidates = []
        for quinphone in unit_names:
            cur = []
            mono, di, tri, quin = self.break_quinphone(quinphone)
            for form in [quin, tri, di, mono]:
                for unit in self.unit_index.get(form, []):
                    cur.append(unit)
                    if len(cur) == K:
                        break
                if len(cur) == K:
                    break
            if len(cur) == 0:
                self.report('Warning: no cands in training data to match %s! Use v naive backoff to silence...' % (quinphone))
                cur = [1]
            cur += [-1] * (K - len(cur))
            candidates.append(cur)
        candidates = np.array(candidates, dtype=np.int64)
        self.stop_clock(t)
        t = self.start_clock('Compute target distances...')
        distances = self.engine.candidate_distances(np.asarray(unit_features, dtype=np.float64), candidates)
        self.stop_clock(t)
        return (candidates, distances)

    def load_full_magphase(self, fft_half_len=hp.FFTHALFLEN):
        """preload_all_magphase_utts (synth_simple.py:525-535): the full-resolution analysis frames of
        every database utterance, resident on the device (1 M frames: 6.2 GB of 288)."""
        names = []
        for fn in self.train_filenames:
            fn = fn.decode() if isinstance(fn, bytes) else str(fn)
            if not names or names[-1] != fn:
                if fn in names:
                    raise ValueError('utterance %s is not contiguous in the database' % fn)
                names.append(fn)
        spec, fzv, self._frame_spans = hp.load_full_spectra(self.config['full_magphase_dir'], names, fft_half_len)
        self.engine.upload_frames(spec, fzv)

    def concatenate_magphase_stored(self, path, fzero=None):
        """concatenateMagPhaseEpoch (synth_simple.py:655-674; the `store_full_magphase` branch of synth_utt, :439-441):
        (mag, real, imag, fz) of the selected units from the frames stored with the voice."""
        return hp.gather_stored_magphase(self.mp_mag, self.mp_imag, self.mp_real, self.mp_fz, path, fzero)

    def concatenate_magphase(self, path, overlap=None, fzero=None):
        """retrieve_magphase_frag + the overlap-add of concatenateMagPhaseEpoch_sep_files
        (synth_simple.py:538-747) up to the vocoder call: returns (mag, real, imag, fz), what the
        reference hands to magphase.synthesis_from_lossless."""
        if overlap is None:
            overlap = self.config.get('magphase_overlap', 0)
        assert overlap % 2 == 0, 'frame overlap should be even number'
        if not hasattr(self, '_frame_spans'):
            self.load_full_magphase()
        multiepoch = self.config.get('multiepoch', 1)
        first, lo, hi = [], [], []
        for index in path:
            fn = self.train_filenames[index]
            fn = fn.decode() if isinstance(fn, bytes) else str(fn)
            a, b = self._frame_spans[fn]
            first.append(a + int(self.unit_index_within_sentence[index]))
            lo.append(a)
            hi.append(b)
        mag, real, imag, fz = self.engine.concat_fragments(first, lo, hi, multiepoch, overlap, hp.in_taper(overlap))
        if fzero is not None and np.size(fzero) > 0:
            fz = fzero
        return mag, real, imag, fz

    def join_knn(self, k, first=0, last=None):
        """Nearest `unit_end_data` rows of `unit_start_data[first:last]`: the K-NN of
        initialise_join_table_with_knn (active_learning_join.py:184-212: sklearn KDTree over
        unit_end_data, queried with unit_start_data) on the GPU engine.  The join matrix takes the
        place of the target database in a second engine; rows are queried in slices.
        Returns (indices (n, k) int64, distances (n, k) float64)."""
        if self.join_contexts_unweighted.shape[1] > 512:
            # up to 256 columns the matrix sweeps serve the search; the doubled [j_t, j_t+1] join rows of an epoch voice
            # from train_halfphone (2 x 151 columns) go through the engine's canonical-distance selection (api_knn.hip
            # knn_device: a workgroup per query row) -- exact, slower; beyond 512 columns the engine refuses
            raise ValueError('join_knn: join vectors of %d columns exceed the 512 columns the K-NN engine supports'
                             % self.join_contexts_unweighted.shape[1])
        if getattr(self, '_join_engine', None) is None:
            self._join_engine = HipSearchEngine(self.engine.device)
            self._join_engine.upload_target_only(self.join_contexts_unweighted[1:, :])
            self._join_engine_weights = None
        if self._join_engine_weights is None or not np.array_equal(self._join_engine_weights, self._device_wj):
            self._join_engine.set_weights(self._device_wj, None)
            self._join_engine_weights = self._device_wj.copy()
        S = self.join_contexts_unweighted[:-1, :][first:last]
        out_i, out_d = [], []
        for r0 in range(0, S.shape[0], 8192):
            q = hp.weight(S[r0:r0 + 8192].astype(np.float64), self._device_wj)
            i, d = self._join_engine.knn(q, k)
            out_i.append(i)
            out_d.append(d)
        return np.vstack(out_i), np.vstack(out_d)

    def viterbi_search(self, candidates, distances):
        """synth_halfphone.py:1399-1436."""
        t = self.start_clock('Compose and find shortest path')
        best_path, cost = self.engine.viterbi(candidates, distances)
        self.stop_clock(t)
        self.last_path_cost = cost
        self.report('got shortest path:')
        self.report(best_path)
        return best_path

    # ------------------------------------------------------------------ drivers
    def get_sentence_set(self, set_name):
        """synth_simple.py:287-339."""
        assert set_name in ['test', 'tune']
        first_stream = self.stream_list_target[0]
        if set_name == 'test':
            data_dirs = self.test_data_target_dirs[first_stream]
            name_patterns = self.config.get('test_patterns', [])
            limit = self.config['n_test_utts']
        else:
            data_dirs = self.tune_data_target_dirs[first_stream]
            name_patterns = self.config.get('tune_patterns', [])
            limit = self.config['n_tune_utts']
        flist = sorted(glob.glob(data_dirs + '/*.' + first_stream))
        flist = [os.path.split(f)[-1].rsplit('.', 1)[0] for f in flist]
        if name_patterns:
            selected = []
            for fname in flist:
                for pattern in name_patterns:
                    if pattern in fname and fname not in selected:
                        selected.append(fname)
            flist = selected
        train_names = set(n.decode() if isinstance(n, bytes) else str(n) for n in np.unique(self.train_filenames))
        flist = [n for n in flist if n not in train_names]
        if limit > 0:
            flist = flist[:limit]
        return flist

    def synth_from_config(self, inspect_join_weights_only=False, synth_type='test', outdir='', ncores=1, devices=None):
        """synth_simple.py:279-284 / synth_halfphone.py:891-908: returns {utterance: path}.

        ncores > 1 is the reference's ``multiprocessing.Pool(processes=ncores)`` over the sentences
        (synth_halfphone.py:897-903), here as utterance-level REPLICAS: `ncores` worker processes, each with
        its own engine and its own copy of the voice on ``devices[r % len(devices)]`` (default: every visible
        GPU in turn), worker r taking sentences r, r + ncores, ...  No exchange between them -- a voice is a
        few GB of the 288 GB of one GPU; the row-sharded database (snickery_amd/dist.py) is the other way to
        use several GPUs, for one utterance stream.

        inspect_join_weights_only: accepted and without effect, as in the reference -- synth_simple.py never reads it
        (:279-284) and synth_halfphone.py only mentions it in the unreachable ``junk`` method (:911-935)."""
        if inspect_join_weights_only:
            import warnings
            warnings.warn('inspect_join_weights_only has no effect (neither in the reference: synth_halfphone.py:911)')
        flist = self.get_sentence_set(synth_type)
        if ncores <= 1 or len(flist) <= 1:
            return dict((f, self.synth_utt(f, synth_type=synth_type, outdir=outdir)) for f in flist)
        import multiprocessing
        from . import engine as _engine
        if devices is None:
            devices = list(range(max(_engine.device_count(), 1)))
        ncores = min(int(ncores), len(flist))
        state = self._runtime_state()
        # replicas that share a device use the greedy scan that launches per step: the one-launch scan waits inside the
        # kernel for all of its workgroups, and two such launches on one GPU can keep each other from starting
        shared = ncores > len(set(devices))
        jobs = [(self.config_file, self.flavour, devices[r % len(devices)], state, flist[r::ncores], synth_type, outdir, shared)
                for r in range(ncores)]
        # 'spawn': a forked child inherits an initialised HIP runtime it cannot use
        pool = multiprocessing.get_context('spawn').Pool(processes=ncores)
        try:
            shares = pool.map(_replica_worker, jobs)
        finally:
            pool.close()
            pool.join()
        done = dict(pair for share in shares for pair in share)
        return dict((f, done[f]) for f in flist)

    def _runtime_state(self):
        """What set_*_weights / truncate_* / reconfigure_settings changed since the config file was read."""
        return {'config': dict(self.config), 'target_weight_vector': self.target_weight_vector.copy(),
                'join_weight_vector': self.join_weight_vector.copy(), 'target_trunc': self._target_trunc,
                'join_trunc': self._join_trunc, 'mode_of_operation': self.mode_of_operation}

    def _restore_runtime_state(self, state):
        self.config = dict(state['config'])
        self.target_weight_vector = state['target_weight_vector']
        self.join_weight_vector = state['join_weight_vector']
        self._target_trunc, self._join_trunc = state['target_trunc'], state['join_trunc']
        if self._target_trunc is not None:
            self.target_truncation_vector = self._target_trunc
        self.mode_of_operation = state['mode_of_operation']
        self._apply_weights()

    def _weight_targets(self, unit_features):
        """The tail of prepare_targets: stream weights and truncated columns (what changes between the iterations of
        a tuning loop; everything before it depends on the files only)."""
        if self.flavour == 'simple' or self.config['weight_target_data']:
            unit_features = hp.weight(unit_features, self.target_weight_vector)
        return self._mask_query(unit_features)

    def prepare_targets(self, base, synth_type='test', return_names=False, _weighted=True):
        """Target preparation of synth_utt (synth_simple.py:370-396, synth_halfphone.py:1478-1552):
        frame-level targets for the epoch representation, one row per halfphone (taken at the
        state-alignment points of the utterance's label) otherwise."""
        if synth_type not in ('test', 'tune'):
            raise SystemExit('Unknown synth_type  9489384')
        data_dirs = self.test_data_target_dirs if synth_type == 'test' else self.tune_data_target_dirs
        unit_names = None
        unnorm_speech = hp.compose_speech(data_dirs, base, self.stream_list_target, self.config['datadims_target'])
        if self.config.get('standardise_target_data', True):
            speech = hp.standardise(unnorm_speech, self.mean_vec_target, self.std_vec_target)
        else:
            speech = unnorm_speech
        if self.flavour == 'simple':
            unit_features = speech[1:-1, :] if self.config.get('REPLICATE_IS2018_EXP', False) else speech
        else:
            if self.target_representation == 'epoch':
                unit_features = speech[1:-1, :]
            else:
                lab_dir = self.config.get('%s_lab_dir' % synth_type, '')
                labfile = os.path.join(lab_dir, base + '.' + self.config['lab_extension'])
                self.report('reading %s' % (labfile))
                labs = hp.read_label(labfile, self.quinphone_regex)
                if self.config.get('untrim_silence_target_speech', False):
                    speech = hp.reinsert_terminal_silence(speech, labs)
                if self.config.get('suppress_weird_festival_pauses', False):
                    labs = hp.suppress_weird_festival_pauses(labs)
                unit_names, unit_features, unit_timings = hp.get_halfphone_stats(
                    speech, labs, representation_type=self.target_representation)
                if self.config.get('add_duration_as_target', False):
                    norm_durations = hp.get_norm_durations(unit_names, unit_timings, self.duration_stats)
                    norm_durations *= self.config.get('target_duration_stretch_factor', 1.0)
                    unit_features = np.hstack([unit_features, norm_durations])
        if _weighted:
            unit_features = self._weight_targets(unit_features)
        return (unit_features, unit_names) if return_names else unit_features

    def synth_utts_bulk(self, fnames, synth_type='test'):
        """synth_utt over a list of utterances (the reference's commented-out synth_utts_bulk,
        synth_halfphone.py:1060-1180, and the list comprehension of balance_stream_weights.py:92).
        With acoustic preselection + Viterbi the whole list goes through ONE call of the batch entry
        point (grouped K-NN, one join launch and one recursion launch per group); other
        configurations loop.  Returns what synth_utt returns, per utterance."""
        fnames = list(fnames)
        plain = (self.mode_of_operation in ('normal', 'stream_weight_balancing')
                 and not self.config.get('get_selection_info', False))
        if plain and fnames and self.config.get('greedy_search', False):
            # greedy voices (what balance_stream_weights.py tunes): two utterances share every scan of
            # the database (snk_greedy_batch)
            assert self.config.get('target_representation') == 'epoch'
            t = self.start_clock('Get speech (bulk)')
            feats = [self.prepare_targets(f, synth_type) for f in fnames]
            self.stop_clock(t)
            t = self.start_clock('Batched greedy search')
            paths = self.engine.greedy_batch(feats, search_epsilon=self.config.get('search_epsilon', 0.0))
            self.stop_clock(t)
            if self.mode_of_operation == 'stream_weight_balancing':
                return [(self.get_target_scores_per_stream(U, p), self.get_join_scores_per_stream(p))
                        for U, p in zip(feats, paths)]
            return paths
        method = self.config.get('preselection_method')
        if (plain and len(fnames) > 1 and not self.config.get('greedy_search', False)
                and method in ('quinphone', 'monophone_then_acoustic')):
            # label-driven preselection (the reference's own configs use 'quinphone'): candidates per utterance as in
            # synth_utt, then ONE call for the Viterbi search of all of them (join bounds + sparse exact recursion,
            # snk_viterbi_batch) instead of a dense join + recursion per utterance
            t = self.start_clock('Get speech + preselection (bulk)')
            feats, cands, dists = [], [], []
            cache = self.__dict__.setdefault('_bulk_cache', {})
            # everything but the stream weights that the unweighted targets, the unit names and the candidate ids depend
            # on: a changed setting (reconfigure_settings, a caller editing self.config) must not meet a stale entry
            state = repr(sorted((k, repr(v)) for k, v in self.config.items()
                                if not k.endswith('_stream_weights') and k != 'join_cost_weight'))
            if len(cache) > 4096:                              # bounded: a tune set is tens of sentences
                cache.clear()
            for f in fnames:
                # what depends on the files and labels only is kept across calls (a tuning loop searches the same tune
                # set again and again with other weights): unweighted targets, unit names, quinphone candidate ids
                key = (synth_type, f, method, self.config['n_candidates'], state)
                ent = cache.get(key)
                if ent is None:
                    raw, names = self.prepare_targets(f, synth_type, return_names=True, _weighted=False)
                    ent = cache[key] = {'raw': raw, 'names': names, 'cand': None}
                U, names = self._weight_targets(ent['raw']), ent['names']
                if method == 'quinphone':
                    c, d = self.preselect_units_quinphone(U, names, candidates=ent['cand'])
                    ent['cand'] = c
                else:
                    c, d = self.preselect_units_monophone_then_acoustic(U, names)
                feats.append(U); cands.append(c); dists.append(d)
            self.stop_clock(t)
            if len(set(c.shape[1] for c in cands)) == 1:
                t = self.start_clock('Batched Viterbi')
                paths, costs = self.engine.viterbi_batch(cands, dists)
                self.stop_clock(t)
                paths = [[int(u) for u in p] for p in paths]
                if self.mode_of_operation == 'stream_weight_balancing':
                    return [(self.get_target_scores_per_stream(U, p), self.get_join_scores_per_stream(p))
                            for U, p in zip(feats, paths)]
                return paths
        batched = (plain and not self.config.get('greedy_search', False) and method == 'acoustic')
        if not batched or not fnames:
            return [self.synth_utt(f, synth_type=synth_type) for f in fnames]
        t = self.start_clock('Get speech (bulk)')
        feats = [self.prepare_targets(f, synth_type) for f in fnames]
        self.stop_clock(t)
        t = self.start_clock('Batched preselection + Viterbi')
        paths, costs = self.engine.knn_viterbi_batch(feats, self.config['n_candidates'])
        self.stop_clock(t)
        paths = [[int(u) for u in p] for p in paths]
        if self.mode_of_operation == 'stream_weight_balancing':
            return [(self.get_target_scores_per_stream(U, p), self.get_join_scores_per_stream(p))
                    for U, p in zip(feats, paths)]
        return paths

    def synth_utt(self, base, synth_type='tune', outstem='', outdir=''):
        """Search part of synth_utt (synth_simple.py:342-456 / synth_halfphone.py:1478-1696).
        Returns the unit path, or (tscores, jscores) in 'stream_weight_balancing' mode."""
        t = self.start_clock('Get speech ')
        unit_features, unit_names = self.prepare_targets(base, synth_type, return_names=True)
        self.stop_clock(t)
        if self.config.get('greedy_search', False):
            assert self.config.get('target_representation') == 'epoch'
            best_path = self.greedy_joint_search(unit_features)
        else:
            method = self.config['preselection_method']
            if method == 'acoustic':
                candidates, distances = self.preselect_units_acoustic(unit_features)
            elif method == 'quinphone':
                candidates, distances = self.preselect_units_quinphone(unit_features, unit_names)
            elif method == 'monophone_then_acoustic':
                candidates, distances = self.preselect_units_monophone_then_acoustic(unit_features, unit_names)
            else:
                raise SystemExit('preselection_method unknown')
            if self.mode_of_operation == 'find_join_candidates':
                return candidates
            best_path = self.viterbi_search(candidates, distances)
        if self.mode_of_operation == 'stream_weight_balancing':
            return (self.get_target_scores_per_stream(unit_features, best_path),
                    self.get_join_scores_per_stream(best_path))
        if self.config.get('get_selection_info', False) and (outdir or outstem):
            stem = outstem or os.path.join(outdir, base)
            with open(stem + '.trace.txt', 'w') as f:
                for line in self.get_path_information_epoch(unit_features, best_path):
                    f.write(line + '\n')
        return best_path

    # ------------------------------------------------------------------ scores / info
    def _aggregate(self, sq_errs, stream_list, datadims, reps=1):
        """aggregate_squared_errors_by_stream (synth_halfphone.py:2977-3008)."""
        out, start = [], 0
        width = sum(datadims[s] for s in stream_list)
        for stream in stream_list:
            w = datadims[stream]
            acc = 0.0
            for r in range(reps):
                acc = acc + sq_errs[:, r * width + start: r * width + start + w].sum(axis=1)
            out.append(acc)
            start += w
        return np.vstack(out).T

    def get_target_scores_per_stream(self, target_features, best_path):
        me = self.config.get('multiepoch', 1) if self.config.get('greedy_search', False) else 1
        nep = 2 if (me > 1 and self.config.get('last_frame_as_target', False)) else me
        mode = 1 if self.config.get('greedy_search', False) else 0
        tsq, _ = self.engine.path_scores(np.asarray(target_features, dtype=np.float64)[:len(best_path) * me],
                                         best_path, mode, nep * self.target_weight_vector.size,
                                         self._join_score_cols(mode))
        return self._aggregate(tsq, self.stream_list_target, self.datadims_target, reps=nep)

    def _join_score_cols(self, mode):
        n = self.join_weight_vector.size
        return n // 2 if (mode == 1 and self.flavour == 'halfphone' and self._double_join) else n

    def get_join_scores_per_stream(self, best_path):
        me = self.config.get('multiepoch', 1) if self.config.get('greedy_search', False) else 1
        mode = 1 if self.config.get('greedy_search', False) else 0
        nep = 2 if (me > 1 and self.config.get('last_frame_as_target', False)) else me
        dummy = np.zeros((len(best_path) * me, self.target_weight_vector.size))
        _, jsq = self.engine.path_scores(dummy, best_path, mode, nep * self.target_weight_vector.size,
                                         self._join_score_cols(mode))
        reps = 2 if (self._double_join and mode == 0) else 1
        return self._aggregate(jsq, self.stream_list_join, self.datadims_join, reps=reps)

    def get_path_information_epoch(self, target_features, best_path):
        """synth_simple.py:859-883 == synth_halfphone.py:2142-2153: one '<filename> <start> <end>' line per
        selected unit (what goes into the .trace.txt artefact); start = index of the unit within its
        sentence, end = start + multiepoch."""
        multiepoch = self.config.get('multiepoch', 1)
        lines = []
        for p in best_path:
            fn = self.train_filenames[p]
            fn = fn.decode() if isinstance(fn, bytes) else str(fn)
            start = int(self.unit_index_within_sentence[p]) if self.unit_index_within_sentence is not None else int(p)
            lines.append('%s %s %s' % (fn, start, start + multiepoch))
        return lines

    # ------------------------------------------------------------------ reconfiguration
    def reconfigure_settings(self, changed_config_values):
        """synth_simple.py:776-830: returns a description of what changed ('' if nothing).
        Re-weighting is a device-side O(N*D) pass; no tree is rebuilt."""
        assert self.config['target_representation'] == 'epoch'
        assert self.config['greedy_search']
        for key in ['join_stream_weights', 'target_stream_weights', 'join_cost_weight', 'search_epsilon',
                    'multiepoch', 'magphase_use_target_f0', 'magphase_overlap',
                    'truncate_target_streams', 'truncate_join_streams']:
            assert key in changed_config_values, key
        rebuild = False
        description = ''
        for item in ['join_cost_weight', 'join_stream_weights', 'target_stream_weights', 'multiepoch',
                     'truncate_target_streams', 'truncate_join_streams']:
            if self.config.get(item) != changed_config_values[item]:
                description += '%s: %s -> %s\n' % (item, self.config.get(item), changed_config_values[item])
                self.config[item] = changed_config_values[item]
                rebuild = True
        for item, default in [('search_epsilon', 1.0), ('magphase_use_target_f0', True), ('magphase_overlap', 0)]:
            if self.config.get(item, default) != changed_config_values[item]:
                description += '%s: %s -> %s\n' % (item, self.config.get(item, default), changed_config_values[item])
                self.config[item] = changed_config_values[item]
        self.__dict__.pop('_bulk_cache', None)             # targets / candidates kept by the bulk path follow the settings
        if rebuild:
            self.set_join_weights(np.array(self.config['join_stream_weights']) * self.config['join_cost_weight'], _apply=False)
            self.set_target_weights(np.array(self.config['target_stream_weights']) * (1.0 - self.config['join_cost_weight']), _apply=False)
            self.truncate_target_streams(self.config['truncate_target_streams'], _apply=False)
            self.truncate_join_streams(self.config['truncate_join_streams'], _apply=False)
            self._apply_weights()
            self.get_tree_for_greedy_search()
        return description

    def reconfigure_from_config_file(self):
        """synth_simple.py:834-852."""
        return self.reconfigure_settings(hp.load_config(self.config_file))

    # ------------------------------------------------------------------ clock / report
    def report(self, msg):
        if self.verbose:
            print(msg)

    def start_clock(self, comment):
        return (timeit.default_timer(), comment)

    def stop_clock(self, start):
        start_time, comment = start
        if self.verbose:
            print('%s--> took %.2f seconds' % ((comment + '... ').ljust(45), timeit.default_timer() - start_time))

    def close(self):
        self.engine.close()
        if getattr(self, '_join_engine', None) is not None:
            self._join_engine.close()
            self._join_engine = None


def _replica_worker(job):
    """One replica of synth_from_config(ncores > 1): its own Synthesiser on its own device, its share of the sentences."""
    config_file, flavour, device, state, fnames, synth_type, outdir, shared = job
    synth = Synthesiser(config_file, flavour=flavour, device=device, verbose=False)
    try:
        synth._restore_runtime_state(state)
        if shared:
            synth.engine.set_option('greedy_mode', 0)
        return [(f, synth.synth_utt(f, synth_type=synth_type, outdir=outdir)) for f in fnames]
    finally:
        synth.close()
