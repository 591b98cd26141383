#!/opt/conda/bin/python3.9
"""Golden data for the `dump_join_data` output of the halfphone database writer: runs the REAL reference's
train_halfphone.main_work (converted to Python 3 in a temp dir by tools/make_golden.convert_reference, nothing of it
is written into this repo) with dump_join_data = True on the seeded corpus of tests/voice_fixture.py and records
(name, shape, dtype, sha256, a few rows) of every dataset of the `.joindata.hdf5` file it writes, and of the voice
file written beside it, into tests/golden/reference_joindata.npz.

  /opt/conda/bin/python3.9 tools/make_golden_joindata.py        # needs h5py (conda python)
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


def main():
    import h5py
    import make_golden
    import voice_fixture
    tmp = tempfile.mkdtemp(prefix='snk_joindata_')
    try:
        make_golden.convert_reference(tmp)
        import train_halfphone
        data = os.path.join(tmp, 'hp_corpus')
        voice_fixture.write_halfphone_corpus(data)
        rec = {}
        for tag, rep, duration, halfwidth in (('twopoint', 'twopoint', True, 3), ('threepoint', 'threepoint', False, 2)):
            work = os.path.join(tmp, 'work_' + tag)
            os.makedirs(work)
            cfg = voice_fixture.halfphone_corpus_config(os.path.join(tmp, 'hp_%s.cfg' % tag), work, data, rep, duration)
            config = {}
            exec(compile(open(cfg).read(), cfg, 'exec'), config)
            del config['__builtins__']
            config['dump_join_data'] = True
            config['join_cost_halfwidth'] = halfwidth
            with contextlib.redirect_stdout(io.StringIO()):
                train_halfphone.main_work(config, overwrite_existing_data=True)
            rec[tag + '_halfwidth'] = np.array(halfwidth)
            for kind, fname in (('join', train_halfphone.get_data_dump_name(config, joindata=True)),
                                ('voice', train_halfphone.get_data_dump_name(config))):
                rec['%s_%s_basename' % (tag, kind)] = np.array(os.path.basename(fname))
                with h5py.File(fname, 'r') as f:
                    rec['%s_%s_keys' % (tag, kind)] = np.array(sorted(f.keys())).astype('S40')
                    for key in f.keys():
                        arr = f[key][...]
                        rec['%s_%s_%s_shape' % (tag, kind, key)] = np.array(arr.shape, dtype=np.int64)
                        rec['%s_%s_%s_dtype' % (tag, kind, key)] = np.array(arr.dtype.str)
                        rec['%s_%s_%s_sha256' % (tag, kind, key)] = np.array(hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest())
                        if kind == 'join':
                            rec['%s_%s_%s_rows' % (tag, kind, key)] = arr[[0, arr.shape[0] // 2, -1], :8]
        np.savez_compressed(os.path.join(OUT, 'reference_joindata.npz'), **rec)
        print('wrote tests/golden/reference_joindata.npz (%d bytes)' % os.path.getsize(os.path.join(OUT, 'reference_joindata.npz')))
        for k in sorted(rec):
            if k.endswith('_shape'):
                print('  ', k, rec[k])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
