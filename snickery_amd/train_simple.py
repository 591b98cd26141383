#!/usr/bin/env python3
"""Unit-database writer for epoch voices: Python-3 counterpart of the reference's
``script/train_simple.py`` (main_work, :27-335), producing the database the search engine consumes
(SURVEY a1): same utterance selection, same statistics, same arrays, names, dtypes and shapes --
including the (1, D) shape of the std vectors and the extra first row of ``join_contexts`` (the first
utterance's first frame doubles as initial history, :215-217).

``store_full_magphase`` adds ``mp_mag`` / ``mp_imag`` / ``mp_real`` / ``mp_fz``: one full-resolution analysis frame per unit
(:145-149, :260-299), the arrays ``synth_simple.py:100-104`` reads back.

The arrays are written to the HDF5 file itself, as the reference does (h5py where installed, else
libhdf5 through ctypes: ``snickery_amd.hdf5_io``); only an image with neither gets the ``.npz`` sidecar.

    python -m snickery_amd.train_simple -c voice.cfg [-X]
"""
import glob
import os
import sys
from argparse import ArgumentParser

import numpy as np

from . import hostprep as hp


def select_utterances(config, target_stream_dirs):
    """train_simple.py:52-83: files of the first target stream, limited by count or name pattern,
    minus test material, optionally intersected with a list file."""
    first_stream = config['stream_list_target'][0]
    utt_list = sorted(glob.glob(target_stream_dirs[first_stream] + '/*.' + first_stream))
    flist = [os.path.split(fname)[-1].replace('.' + first_stream, '') for fname in utt_list]
    n_train_utts = config.get('n_train_utts', 0)            # 0: all sentences
    if isinstance(n_train_utts, int):
        if n_train_utts == 0 or n_train_utts > len(flist):
            n_train_utts = len(flist)
        flist = flist[:n_train_utts]
    elif isinstance(n_train_utts, str):
        flist = [name for name in flist if n_train_utts in name]
    if 'test_patterns' in config:
        flist = [name for name in flist if not any(pattern in name for pattern in config['test_patterns'])]
    if 'train_list' in config:
        assert os.path.isfile(config['train_list']), 'File %s does not exist' % (config['train_list'])
        with open(config['train_list']) as f:
            keep = set(line.strip() for line in f if line.strip())
        flist = [name for name in flist if name in keep]
    assert len(flist) > 0
    return flist


def build_database(config, report=print):
    """The arrays of the database as a dict (train_simple.py:85-318)."""
    assert config['target_representation'] == 'epoch'
    stream_list_target, datadims_target = config['stream_list_target'], config['datadims_target']
    stream_list_join, datadims_join = config['stream_list_join'], config['datadims_join']
    target_stream_dirs = hp.locate_stream_directories(config['target_datadirs'], stream_list_target)
    join_stream_dirs = hp.locate_stream_directories(config['join_datadirs'], stream_list_join)
    flist = select_utterances(config, target_stream_dirs)

    mean_vec_target, std_vec_target = hp.get_mean_std(target_stream_dirs, stream_list_target, datadims_target, flist)
    mean_vec_join, std_vec_join = hp.get_mean_std(join_stream_dirs, stream_list_join, datadims_join, flist)

    # utterances without a file in the (alphabetically) first target stream are dropped (:121-131)
    probe_stream, probe_dir = sorted(target_stream_dirs.items())[0]
    flist = [base for base in flist if os.path.exists(os.path.join(probe_dir, base + '.' + probe_stream))]

    replicate = config.get('REPLICATE_IS2018_EXP', False)
    features, contexts, names, filenames, indices = [], [], [], [], []
    store_mp = bool(config.get('store_full_magphase', False))
    mp = ([], [], [], [])
    first_base = flist[0] if flist else None
    for base in flist:
        t_speech = hp.compose_speech(target_stream_dirs, base, stream_list_target, datadims_target)
        if t_speech.size == 1:                              # a stream file is missing
            continue
        t_speech = hp.standardise(t_speech, mean_vec_target, std_vec_target)
        j_speech = hp.compose_speech(join_stream_dirs, base, stream_list_join, datadims_join)
        if j_speech.size == 1:
            continue
        j_speech = hp.standardise(j_speech, mean_vec_join, std_vec_join)
        if j_speech.shape[0] != t_speech.shape[0]:
            report('Warning: number of rows in target cost features not same as number in join cost features:')
            report(' Skipping utterance!')
            continue
        first_sentence_in_corpus = base == first_base
        if replicate:
            unit_features = t_speech[1:-1, :]
            context_data = j_speech[:-1, :] if first_sentence_in_corpus else j_speech[1:-1, :]
        else:
            unit_features = t_speech
            # join_contexts carries one extra leading row of history: the first frame, assumed silent
            context_data = np.vstack([j_speech[0, :].reshape((1, -1)), j_speech]) if first_sentence_in_corpus else j_speech
        m = unit_features.shape[0]
        if store_mp:                                        # :260-275, :292-299
            for acc, part in zip(mp, hp.full_magphase_rows(config, base, m)):
                acc.append(part)
        features.append(unit_features)
        contexts.append(context_data)
        names.extend(['_'] * m)
        filenames.extend([base] * m)
        indices.append(np.arange(m))
    if not features:
        raise RuntimeError('no utterance could be added to the database')
    if flist and filenames[0] != first_base:
        # the reference writes rows start+1.. for every utterance but the first of the LIST: with the
        # first utterance skipped, row 0 of join_contexts would stay unset there; refuse rather than guess
        raise RuntimeError('the first training utterance (%s) could not be used' % first_base)
    db = {
        'train_unit_features': np.vstack(features).astype(np.float32),
        'train_unit_names': np.array(names).astype('S50'),
        'filenames': np.array(filenames).astype('S50'),
        'unit_index_within_sentence_dset': np.concatenate(indices).astype(np.int32),
        'join_contexts': np.vstack(contexts).astype(np.float32),
        'mean_target': np.asarray(mean_vec_target, dtype=np.float32),
        'std_target': np.asarray(std_vec_target, dtype=np.float32),
        'mean_join': np.asarray(mean_vec_join, dtype=np.float32),
        'std_join': np.asarray(std_vec_join, dtype=np.float32),
    }
    if store_mp:
        for key, acc in zip(('mp_mag', 'mp_imag', 'mp_real', 'mp_fz'), mp):
            db[key] = np.vstack(acc).astype(np.float32)
    return db


def main_work(config, overwrite_existing_data=False, report=print):
    """train_simple.py:27-335.  Returns the path of the database."""
    database_fname = hp.get_data_dump_name(config)
    present = [p for p in (database_fname, database_fname + '.npz') if os.path.isfile(p)]
    if present:
        if not overwrite_existing_data:
            sys.exit('Data already exists at %s -- run with -X to overwrite it' % (present[0]))
        for p in present:
            os.remove(p)
    os.makedirs(os.path.dirname(database_fname), exist_ok=True)
    db = build_database(config, report=report)
    written = hp.store_database(database_fname, db)
    report('Stored training data for %s units to %s' % (db['train_unit_features'].shape[0], written))
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
