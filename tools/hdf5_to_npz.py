#!/usr/bin/env python
"""Convert a Snickery unit database (.hdf5 from train_simple.py / train_halfphone.py) to the
.npz sidecar that snickery_amd.hostprep.load_database reads when h5py is unavailable.
Run with an interpreter that has h5py, e.g. /opt/conda/bin/python3.9."""
import sys
import h5py
import numpy as np

src = sys.argv[1]
out = {}
with h5py.File(src, 'r') as f:
    for k in f.keys():
        out[k] = f[k][...]
np.savez(src + '.npz', **out)
print('wrote', src + '.npz', sorted(out.keys()))
