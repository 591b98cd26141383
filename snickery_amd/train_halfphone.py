#!/usr/bin/env python3
"""Unit-database writer for halfphone voices and for pitch-synchronous epoch voices: Python-3
counterpart of the reference's ``script/train_halfphone.py`` (main_work, :63-628) -- the writer of
the databases ``synth_halfphone.py`` reads (K-NN preselection + Viterbi).  Same utterance
selection, statistics, arrays, names, dtypes and shapes as the reference, including

* epoch voices: ``train_unit_features = t_speech[1:-1]``, ``join_contexts`` rows ``[j_t, j_{t+1}]``
  (2 x Dj columns, :446), ``cutpoints`` = triples of consecutive pitch marks in samples (:442);
* halfphone voices: start/middle/end frame features per halfphone (+ the normalised duration),
  cut points at the pitch marks nearest to the unit boundaries, one pitch-synchronous join frame
  per boundary; and the reference's placement of the database's final join row -- it is written at
  row ``m`` (the unit count of the LAST utterance, :552) and the last row of ``join_contexts`` stays
  zero.  Readers depend on the file, so the file is reproduced as it is.

``dump_join_data`` (halfphone voices; :277-282, :484, :529-531, :620-627, ``get_join_data_AL`` :1207-1242) writes the
second file ``<condition>.joindata.hdf5`` with ``start_join_feats`` / ``end_join_feats``: ``join_cost_halfwidth``
pitch-synchronous join frames after each unit's first and before each unit's last cut point -- the data
``active_learning_join.py`` trains a join cost on -- including the reference's index arithmetic at the utterance edges.

``store_full_magphase`` (:269-273, :504-543) adds ``mp_mag`` / ``mp_imag`` / ``mp_real`` / ``mp_fz``, one full-resolution
analysis frame per unit (epoch voices: the files must hold one row per pitch mark).

Not covered: ``target_representation = 'sample'`` (waveform-sample voices).

    python -m snickery_amd.train_halfphone -c voice.cfg [-X]
"""
import os
import sys
from argparse import ArgumentParser

import numpy as np

from . import hostprep as hp
from .train_simple import select_utterances


def get_data_dump_name(config, joindata=False):
    """train_halfphone.py:878-904 (the default of target_representation differs from file_naming.py)."""
    name = '%s_utts_jstreams-%s_tstreams-%s_rep-%s' % (
        config['n_train_utts'], '-'.join(config['stream_list_join']), '-'.join(config['stream_list_target']),
        config.get('target_representation', 'twopoint'))
    return os.path.join(config['workdir'], 'data_dumps', name + ('.joindata.hdf5' if joindata else '.hdf5'))


def segment_rows(a, length):
    """segment_axis(a, length, overlap=length-1, axis=0) of the reference (segmentaxis.py):
    all windows of `length` consecutive rows."""
    n = a.shape[0] - length + 1
    return np.stack([a[i:i + n] for i in range(length)], axis=1)


def get_join_data_AL(speech, pm_indices, halfwidth):
    """train_halfphone.py:1207-1242: per unit the `halfwidth` join frames from its first cut point on and the
    `halfwidth` frames ending halfwidth + 1 ... 2 rows before its last one, flattened.  The reference pads the
    utterance by repeating its last (first) row when the last unit's window (the first unit's) would leave it and
    then indexes with the shifted cut points as they are -- a negative index wraps around, as numpy does; kept."""
    starts = pm_indices[:, 0]
    ends = pm_indices[:, 1].copy()
    start_speech = speech
    if starts[-1] + halfwidth > ends[-1]:
        difference = int(starts[-1] + halfwidth - ends[-1])
        start_speech = np.vstack([speech] + difference * [speech[-1, :].reshape((1, -1))])
    start_contexts = segment_rows(start_speech, halfwidth)[starts, :, :].reshape((len(starts), -1))
    end_speech = speech
    if ends[0] - (halfwidth + 1) < 0:
        difference = int((ends[0] - (halfwidth + 1)) * -1)
        end_speech = np.vstack(difference * [speech[0, :].reshape((1, -1))] + [speech])
    ends = ends - (halfwidth + 1)
    end_contexts = segment_rows(end_speech, halfwidth)[ends, :, :].reshape((len(ends), -1))
    return start_contexts, end_contexts


def build_database(config, report=print, joindata=None):
    """The arrays of the database as a dict (train_halfphone.py:63-600).  joindata: a dict that receives
    start_join_feats / end_join_feats when the config asks for dump_join_data."""
    rep = config['target_representation']
    if rep == 'sample':
        raise NotImplementedError('sample voices are not covered')
    store_mp = bool(config.get('store_full_magphase', False))
    mp = ([], [], [], [])
    dump_join = bool(config.get('dump_join_data', False))
    if dump_join and rep == 'epoch':
        raise NotImplementedError('dump_join_data needs unit cut points: halfphone voices only '
                                  '(the reference reads an undefined name here for epoch voices, train_halfphone.py:485)')
    epoch = rep == 'epoch'
    stream_list_target, datadims_target = config['stream_list_target'], config['datadims_target']
    stream_list_join, datadims_join = config['stream_list_join'], config['datadims_join']
    target_stream_dirs = hp.locate_stream_directories(config['target_datadirs'], stream_list_target)
    join_stream_dirs = hp.locate_stream_directories(config['join_datadirs'], stream_list_join)
    if 'test_patterns' not in config:
        raise KeyError('test_patterns')                  # the reference reads it unconditionally (:122)
    flist = select_utterances(config, target_stream_dirs)

    mean_vec_target, std_vec_target = hp.get_mean_std(target_stream_dirs, stream_list_target, datadims_target, flist)
    mean_vec_join, std_vec_join = hp.get_mean_std(join_stream_dirs, stream_list_join, datadims_join, flist)
    add_duration = bool(config.get('add_duration_as_target', False))
    sample_rate = config.get('sample_rate', 48000)

    # first pass (:197-236): which utterances, duration statistics per halfphone class
    duration_stats = {}
    if epoch:
        probe_stream, probe_dir = sorted(target_stream_dirs.items())[0]
        flist = [base for base in flist if os.path.exists(os.path.join(probe_dir, base + '.' + probe_stream))]
    else:
        duration_data = {}
        for base in flist:
            label = hp.read_label(os.path.join(config['label_datadir'], base + '.' + config['lab_extension']),
                                  config['quinphone_regex'])
            assert len(label) % 5 == 0
            if add_duration:
                for name, dur in hp.get_halfphone_lengths(label):
                    duration_data.setdefault(name, []).append(dur)
        for name, vals in duration_data.items():
            vals = np.array(vals)
            duration_stats[name] = (vals.mean(), max(vals.std(), 0.001))        # variance floor

    features, contexts, names, filenames, indices, cuts = [], [], [], [], [], []
    start_feats, end_feats = [], []
    last_context, last_m = None, 0
    for base in flist:
        pm_file = os.path.join(config['pm_datadir'], base + '.pm')
        if not os.path.isfile(pm_file):
            report('Warning: no pm -- skip!')
            continue
        pms_seconds = hp.read_pm(pm_file)
        if pms_seconds.shape == (1, 1):
            report('Warning: trouble reading pm file -- skip!')
            continue
        t_speech = hp.compose_speech(target_stream_dirs, base, stream_list_target, datadims_target)
        if t_speech.size == 1:
            continue
        t_speech = hp.standardise(t_speech, mean_vec_target, std_vec_target)
        j_speech = hp.compose_speech(join_stream_dirs, base, stream_list_join, datadims_join)
        if j_speech.size == 1:
            continue
        if config.get('standardise_join_data', True):
            j_speech = hp.standardise(j_speech, mean_vec_join, std_vec_join)
        if j_speech.shape[0] != len(pms_seconds):
            report('Warning: number of rows in join cost features not same as number of pitchmarks: '
                   'these features should be pitch synchronous. Skipping utterance!')
            continue
        if epoch:
            unit_features = t_speech[1:-1, :]
            pms_samples = np.array(pms_seconds * sample_rate, dtype=int)
            cutpoints = segment_rows(pms_samples, 3)
            n_j = j_speech.shape[1]
            context_data = segment_rows(j_speech, 2).reshape((j_speech.shape[0] - 1, 2 * n_j))
            unit_names = np.array(['_'] * (t_speech.shape[0] - 2))
        else:
            labs = hp.read_label(os.path.join(config['label_datadir'], base + '.' + config['lab_extension']),
                                 config['quinphone_regex'])
            if config.get('untrim_silence_target_speech', False):
                t_speech = hp.reinsert_terminal_silence(t_speech, labs)
            t_speech = hp.pad_speech_to_length(t_speech, labs)
            if t_speech.size == 1:
                report('Skip utterance')
                continue
            unit_names, unit_features, timings = hp.get_halfphone_stats(t_speech, labs, rep)
            if add_duration:
                unit_features = np.hstack([unit_features, hp.get_norm_durations(unit_names, timings, duration_stats)])
            cutpoints, cutpoint_indices = hp.get_cutpoints(timings, pms_seconds, sample_rate)
            context_data = hp.get_contexts_for_pitch_synchronous_joincost(j_speech, cutpoint_indices)
            if dump_join:
                sj, ej = get_join_data_AL(j_speech, cutpoint_indices, config['join_cost_halfwidth'])
                start_feats.append(sj)
                end_feats.append(ej)
        m = unit_features.shape[0]
        assert context_data.shape[0] == m + 1, (context_data.shape[0], m)
        if store_mp:                                        # :504-517, :537-543 (one analysis frame per unit: epoch voices)
            for acc, part in zip(mp, hp.full_magphase_rows(config, base, m)):
                acc.append(part)
        features.append(unit_features)
        contexts.append(context_data[:-1, :])
        names.extend(list(unit_names))
        filenames.extend([base] * len(cutpoints))
        indices.append(np.arange(m))
        cuts.append(cutpoints)
        last_context, last_m = context_data[-1, :], m
    if not features:
        raise RuntimeError('no utterance could be added to the database')

    n_units = sum(f.shape[0] for f in features)
    join_contexts = np.zeros((n_units + 1, contexts[0].shape[1]), dtype=np.float32)
    join_contexts[:n_units] = np.vstack(contexts)
    if not epoch:
        join_contexts[last_m, :] = last_context          # as the reference places it (:552; see the module text)
    db = {
        'train_unit_features': np.vstack(features).astype(np.float32),
        'train_unit_names': np.array(names).astype('S50'),
        'filenames': np.array(filenames).astype('S50'),
        'unit_index_within_sentence_dset': np.concatenate(indices).astype(np.int32),
        'cutpoints': np.vstack(cuts).astype(np.int32),
        'join_contexts': join_contexts,
        'mean_target': np.asarray(mean_vec_target, dtype=np.float32),
        'std_target': np.asarray(std_vec_target, dtype=np.float32),
        'mean_join': np.asarray(mean_vec_join, dtype=np.float32),
        'std_join': np.asarray(std_vec_join, dtype=np.float32),
    }
    if store_mp:
        for key, acc in zip(('mp_mag', 'mp_imag', 'mp_real', 'mp_fz'), mp):
            db[key] = np.vstack(acc).astype(np.float32)
    if dump_join and joindata is not None:
        joindata['start_join_feats'] = np.vstack(start_feats).astype(np.float32)
        joindata['end_join_feats'] = np.vstack(end_feats).astype(np.float32)
    if add_duration:
        keys = sorted(duration_stats)
        db['duration_monophones'] = np.array(keys).astype('S50')
        db['duration_stats'] = np.array([duration_stats[k] for k in keys], dtype=np.float32).reshape((len(keys), 2))
    return db


def main_work(config, overwrite_existing_data=False, report=print):
    """train_halfphone.py:63-628.  Returns the path of the database."""
    database_fname = get_data_dump_name(config)
    present = [p for p in (database_fname, database_fname + '.npz') if os.path.isfile(p)]
    if present:
        if not overwrite_existing_data:
            sys.exit('Data already exists at %s -- run with -X to overwrite it' % (present[0]))
        for p in present:
            os.remove(p)
    os.makedirs(os.path.dirname(database_fname), exist_ok=True)
    joindata = {}
    db = build_database(config, report=report, joindata=joindata)
    written = hp.store_database(database_fname, db)
    report('Stored training data for %s units to %s' % (db['train_unit_features'].shape[0], written))
    if joindata:                                              # train_halfphone.py:277-282, :620-627
        join_fname = get_data_dump_name(config, joindata=True)
        for p in (join_fname, join_fname + '.npz'):
            if os.path.isfile(p):
                os.remove(p)
        report('Storing data for learning join cost: %s' % hp.store_database(join_fname, joindata))
    return database_fname


def main(argv=None):
    a = ArgumentParser()
    a.add_argument('-c', dest='config_fname', required=True)
    a.add_argument('-X', dest='overwrite_existing_data', action='store_true',
                   help='clear any previous training data first')
    opts = a.parse_args(argv)
    main_work(hp.load_config(opts.config_fname), overwrite_existing_data=opts.overwrite_existing_data)


if __name__ == '__main__':
    main()
