#!/opt/conda/bin/python3.9
"""Golden data for `store_full_magphase` of the two database writers: runs the REAL reference's train_simple.main_work
and train_halfphone.main_work (converted to Python 3 in a temp dir by tools/make_golden.convert_reference; nothing of it
is written into this repo) with store_full_magphase = True on the seeded fixture corpora and records (name, shape, dtype,
sha256) of every dataset of the voices they write into tests/golden/reference_fullmag.npz.

  /opt/conda/bin/python3.9 tools/make_golden_fullmag.py        # needs h5py (conda python)
"""
import contextlib
import hashlib
import io
import os
import shutil
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..', 'tests'))
OUT = os.path.join(HERE, '..', 'tests', 'golden')


def record(rec, tag, fname):
    import h5py
    rec[tag + '_basename'] = np.array(os.path.basename(fname))
    with h5py.File(fname, 'r') as f:
        rec[tag + '_keys'] = np.array(sorted(f.keys())).astype('S40')
        for key in f.keys():
            arr = f[key][...]
            rec['%s_%s_shape' % (tag, key)] = np.array(arr.shape, dtype=np.int64)
            rec['%s_%s_dtype' % (tag, key)] = np.array(arr.dtype.str)
            rec['%s_%s_sha256' % (tag, key)] = np.array(hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest())


def main():
    import make_golden
    import voice_fixture
    tmp = tempfile.mkdtemp(prefix='snk_fullmag_')
    try:
        make_golden.convert_reference(tmp)
        import train_simple
        import train_halfphone
        import file_naming
        rec = {}
        # ---- train_simple: the epoch voice of tests/voice_fixture.CFG (every frame a unit: files of frames + 2 rows) ----
        data = os.path.join(tmp, 'voice')
        make_golden.write_voice(data, np.random.RandomState(20240))
        names = sorted(f[:-4] for f in os.listdir(os.path.join(data, 'low', 'mag')))
        voice_fixture.write_full_magphase_for_writers(os.path.join(data, 'high'), os.path.join(data, 'low'), names, 2)
        cfg = os.path.join(tmp, 'voice.cfg')
        with open(cfg, 'w') as f:
            f.write(voice_fixture.CFG % dict(workdir=os.path.join(tmp, 'work_simple'), data=data, greedy='True', multiepoch=6, n_candidates=12))
            f.write("store_full_magphase = True\nfull_magphase_dir = data + '/high/'\n")
        config = {}
        exec(compile(open(cfg).read(), cfg, 'exec'), config)
        del config['__builtins__']
        with contextlib.redirect_stdout(io.StringIO()):
            train_simple.main_work(config, overwrite_existing_data=True)
        record(rec, 'simple', file_naming.get_data_dump_name(config))
        # ---- train_halfphone, epoch voice: one unit per inner pitch mark (files of frames rows) ----
        data = os.path.join(tmp, 'hp_corpus')
        names = voice_fixture.write_halfphone_corpus(data)
        voice_fixture.write_full_magphase_for_writers(os.path.join(data, 'high'), os.path.join(data, 'low'), names, 0)
        work = os.path.join(tmp, 'work_hp')
        os.makedirs(work)
        cfg = voice_fixture.halfphone_corpus_config(os.path.join(tmp, 'hp.cfg'), work, data, 'epoch', False)
        config = {}
        exec(compile(open(cfg).read(), cfg, 'exec'), config)
        del config['__builtins__']
        config['store_full_magphase'] = True
        config['full_magphase_dir'] = data + '/high/'
        with contextlib.redirect_stdout(io.StringIO()):
            train_halfphone.main_work(config, overwrite_existing_data=True)
        record(rec, 'hpepoch', train_halfphone.get_data_dump_name(config))
        np.savez_compressed(os.path.join(OUT, 'reference_fullmag.npz'), **rec)
        print('wrote tests/golden/reference_fullmag.npz (%d bytes)' % os.path.getsize(os.path.join(OUT, 'reference_fullmag.npz')))
        for k in sorted(rec):
            if k.endswith('_shape') and '_mp_' in k:
                print('  ', k, rec[k])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
