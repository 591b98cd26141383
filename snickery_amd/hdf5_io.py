"""HDF5 unit databases without h5py: a thin ctypes binding of libhdf5's C API.

The reference loads its voice with ``h5py.File(datafile)`` (script/synth_simple.py:72-106,
script/synth_halfphone.py:184-225) and writes it with ``f.create_dataset(..., maxshape=...)``
(script/train_simple.py:95-149).  The default interpreter of the target image has no h5py, but the
HDF5 C library itself is there (``/opt/conda/lib/libhdf5.so``); these few calls are all the voice
format needs: flat files of float32 / float64 / int32 / int64 / fixed-length string datasets.

    read_datasets(path, names=None) -> {name: ndarray}
    write_datasets(path, {name: ndarray}, resizable=True)
"""
import ctypes
import ctypes.util
import os

import numpy as np

_hid = ctypes.c_int64
_H5F_ACC_RDONLY, _H5F_ACC_TRUNC = 0, 2
_H5P_DEFAULT, _H5S_ALL = 0, 0
_H5T_INTEGER, _H5T_FLOAT, _H5T_STRING = 0, 1, 3
_H5S_UNLIMITED = ctypes.c_uint64(-1).value
_lib = None


class Hdf5Error(RuntimeError):
    pass


def _candidates():
    env = os.environ.get('SNK_LIBHDF5')
    if env:
        yield env
    found = ctypes.util.find_library('hdf5')
    if found:
        yield found
    for p in ('/opt/conda/lib/libhdf5.so', '/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so',
              '/usr/lib/x86_64-linux-gnu/libhdf5_serial.so', 'libhdf5.so'):
        yield p


def library():
    """The loaded libhdf5 (raises Hdf5Error when the image has none)."""
    global _lib
    if _lib is not None:
        return _lib
    err = None
    for cand in _candidates():
        try:
            lib = ctypes.CDLL(cand)
        except OSError as e:
            err = e
            continue
        try:
            lib.H5open()
        except AttributeError as e:                # a library of that name without the HDF5 entry points
            err = e
            continue
        bound = True
        for name, res, args in (
                ('H5Fopen', _hid, [ctypes.c_char_p, ctypes.c_uint, _hid]),
                ('H5Fcreate', _hid, [ctypes.c_char_p, ctypes.c_uint, _hid, _hid]),
                ('H5Fclose', ctypes.c_int, [_hid]),
                ('H5Lexists', ctypes.c_int, [_hid, ctypes.c_char_p, _hid]),
                ('H5Dopen2', _hid, [_hid, ctypes.c_char_p, _hid]),
                ('H5Dclose', ctypes.c_int, [_hid]),
                ('H5Dget_space', _hid, [_hid]),
                ('H5Dget_type', _hid, [_hid]),
                ('H5Dread', ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
                ('H5Dwrite', ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
                ('H5Dcreate2', _hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid, _hid]),
                ('H5Sclose', ctypes.c_int, [_hid]),
                ('H5Sget_simple_extent_ndims', ctypes.c_int, [_hid]),
                ('H5Sget_simple_extent_dims', ctypes.c_int, [_hid, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
                ('H5Screate_simple', _hid, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
                ('H5Tclose', ctypes.c_int, [_hid]),
                ('H5Tcopy', _hid, [_hid]),
                ('H5Tget_class', ctypes.c_int, [_hid]),
                ('H5Tget_size', ctypes.c_size_t, [_hid]),
                ('H5Tget_sign', ctypes.c_int, [_hid]),
                ('H5Tset_size', ctypes.c_int, [_hid, ctypes.c_size_t]),
                ('H5Tset_strpad', ctypes.c_int, [_hid, ctypes.c_int]),
                ('H5Tis_variable_str', ctypes.c_int, [_hid]),
                ('H5Pcreate', _hid, [_hid]),
                ('H5Pset_chunk', ctypes.c_int, [_hid, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
                ('H5Pclose', ctypes.c_int, [_hid]),
                ('H5Literate', ctypes.c_int, [_hid, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p, ctypes.c_void_p]),
                ('H5Eset_auto2', ctypes.c_int, [_hid, ctypes.c_void_p, ctypes.c_void_p])):
            fn = getattr(lib, name, None)
            if fn is None and name == 'H5Literate':
                # libhdf5 >= 1.12: H5Literate is a macro over the versioned symbols (H5Literate1 keeps this signature)
                fn = getattr(lib, 'H5Literate1', None)
                if fn is not None:
                    setattr(lib, 'H5Literate', fn)
            if fn is None:
                err = Hdf5Error('%s has no symbol %s' % (cand, name))
                bound = False
                break
            fn.restype, fn.argtypes = res, args
        if not bound:
            continue
        lib.H5Eset_auto2(0, None, None)            # errors come back as return codes, not as stderr dumps
        _lib = lib
        return lib
    raise Hdf5Error('no libhdf5 found (set SNK_LIBHDF5 to its path): %s' % (err,))


def available():
    try:
        library()
        return True
    except Hdf5Error:
        return False


def _native(lib, name):
    return _hid.in_dll(lib, name).value


def _mem_type(lib, dtype):
    """(hid of the memory type, needs_close) for a numpy dtype."""
    dtype = np.dtype(dtype)
    table = {np.dtype(np.float32): 'H5T_NATIVE_FLOAT_g', np.dtype(np.float64): 'H5T_NATIVE_DOUBLE_g',
             np.dtype(np.int32): 'H5T_NATIVE_INT32_g', np.dtype(np.int64): 'H5T_NATIVE_INT64_g',
             np.dtype(np.uint8): 'H5T_NATIVE_UINT8_g', np.dtype(np.int8): 'H5T_NATIVE_INT8_g',
             np.dtype(np.int16): 'H5T_NATIVE_INT16_g', np.dtype(np.uint16): 'H5T_NATIVE_UINT16_g',
             np.dtype(np.uint32): 'H5T_NATIVE_UINT32_g', np.dtype(np.uint64): 'H5T_NATIVE_UINT64_g'}
    if dtype in table:
        return _native(lib, table[dtype]), False
    if dtype.kind == 'S':
        t = lib.H5Tcopy(_native(lib, 'H5T_C_S1_g'))
        lib.H5Tset_size(t, max(dtype.itemsize, 1))
        lib.H5Tset_strpad(t, 1)                    # H5T_STR_NULLPAD: what h5py writes for numpy 'S50'
        return t, True
    raise Hdf5Error('unsupported dtype %s' % (dtype,))


def _names(lib, fid):
    """Names of the links of the root group."""
    out = []
    CB = ctypes.CFUNCTYPE(ctypes.c_int, _hid, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p)

    def visit(_g, name, _info, _data):
        out.append(name.decode())
        return 0
    cb = CB(visit)
    idx = ctypes.c_uint64(0)
    if lib.H5Literate(fid, 0, 0, ctypes.byref(idx), ctypes.cast(cb, ctypes.c_void_p), None) < 0:
        raise Hdf5Error('H5Literate failed')
    return out


def read_datasets(path, names=None):
    """Datasets of the root group of `path` as numpy arrays (all of them, or those of `names` that exist)."""
    lib = library()
    fid = lib.H5Fopen(os.fsencode(path), _H5F_ACC_RDONLY, _H5P_DEFAULT)
    if fid < 0:
        raise Hdf5Error('cannot open %s as HDF5' % (path,))
    out = {}
    try:
        for name in (names if names is not None else _names(lib, fid)):
            if lib.H5Lexists(fid, name.encode(), _H5P_DEFAULT) <= 0:
                continue
            did = lib.H5Dopen2(fid, name.encode(), _H5P_DEFAULT)
            if did < 0:
                continue                           # a group, not a dataset
            sid, tid = lib.H5Dget_space(did), lib.H5Dget_type(did)
            mem, close_mem = -1, False
            try:
                if sid < 0 or tid < 0:
                    raise Hdf5Error('%s: cannot query space / type of dataset %s' % (path, name))
                nd = lib.H5Sget_simple_extent_ndims(sid)
                dims = (ctypes.c_uint64 * max(nd, 1))()
                if nd > 0:
                    lib.H5Sget_simple_extent_dims(sid, dims, None)
                shape = tuple(int(dims[i]) for i in range(nd))
                cls, size = lib.H5Tget_class(tid), int(lib.H5Tget_size(tid))
                if cls == _H5T_FLOAT and size in (4, 8):
                    dtype = np.dtype(np.float32 if size == 4 else np.float64)
                    mem, _ = _mem_type(lib, dtype)
                elif cls == _H5T_INTEGER and size in (1, 2, 4, 8):
                    signed = lib.H5Tget_sign(tid) != 0
                    dtype = np.dtype('%s%d' % ('i' if signed else 'u', size))
                    mem, _ = _mem_type(lib, dtype)
                elif cls == _H5T_STRING and lib.H5Tis_variable_str(tid) == 0:
                    dtype = np.dtype('S%d' % size)
                    mem, close_mem = lib.H5Tcopy(tid), True
                    if mem < 0:
                        close_mem = False
                        raise Hdf5Error('%s: cannot copy the string type of dataset %s' % (path, name))
                else:
                    raise Hdf5Error('%s: dataset %s has an unsupported type (class %d, %d bytes)' % (path, name, cls, size))
                arr = np.empty(shape, dtype=dtype)
                if arr.size and lib.H5Dread(did, mem, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, arr.ctypes.data_as(ctypes.c_void_p)) < 0:
                    raise Hdf5Error('%s: reading dataset %s failed' % (path, name))
                out[name] = arr
            finally:
                if close_mem:
                    lib.H5Tclose(mem)
                if tid >= 0:
                    lib.H5Tclose(tid)
                if sid >= 0:
                    lib.H5Sclose(sid)
                lib.H5Dclose(did)
    finally:
        lib.H5Fclose(fid)
    return out


def write_datasets(path, arrays, resizable=True):
    """Write {name: ndarray} as root-level datasets.  resizable: chunked with unlimited first
    dimension, like the reference's ``create_dataset(..., maxshape=(None, d))`` voices."""
    lib = library()
    fid = lib.H5Fcreate(os.fsencode(path), _H5F_ACC_TRUNC, _H5P_DEFAULT, _H5P_DEFAULT)
    if fid < 0:
        raise Hdf5Error('cannot create %s' % (path,))
    try:
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            if a.dtype.kind == 'U':
                a = a.astype('S')
            mem, close_mem = _mem_type(lib, a.dtype)
            nd = a.ndim
            dims = (ctypes.c_uint64 * max(nd, 1))(*a.shape)
            chunked = resizable and nd >= 1 and a.size > 0
            maxd = (ctypes.c_uint64 * max(nd, 1))(*((_H5S_UNLIMITED,) + tuple(a.shape[1:]))) if chunked else None
            sid = lib.H5Screate_simple(nd, dims, maxd)
            dcpl, did = _H5P_DEFAULT, -1
            try:
                if sid < 0:
                    raise Hdf5Error('%s: cannot create the data space of %s' % (path, name))
                if chunked:
                    dcpl = lib.H5Pcreate(_native(lib, 'H5P_CLS_DATASET_CREATE_ID_g'))
                    if dcpl < 0:
                        chunked = False
                        raise Hdf5Error('%s: cannot create a property list for %s' % (path, name))
                    row = int(np.prod(a.shape[1:])) * a.dtype.itemsize if nd > 1 else a.dtype.itemsize
                    rows = max(1, min(a.shape[0], (1 << 20) // max(row, 1)))
                    chunk = (ctypes.c_uint64 * nd)(*((rows,) + tuple(a.shape[1:])))
                    if lib.H5Pset_chunk(dcpl, nd, chunk) < 0:
                        raise Hdf5Error('%s: cannot set the chunk shape of %s' % (path, name))
                did = lib.H5Dcreate2(fid, name.encode(), mem, sid, _H5P_DEFAULT, dcpl, _H5P_DEFAULT)
                if did < 0:
                    raise Hdf5Error('%s: cannot create dataset %s' % (path, name))
                if a.size and lib.H5Dwrite(did, mem, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, a.ctypes.data_as(ctypes.c_void_p)) < 0:
                    raise Hdf5Error('%s: writing dataset %s failed' % (path, name))
            finally:
                if did >= 0:
                    lib.H5Dclose(did)
                if chunked and dcpl >= 0:
                    lib.H5Pclose(dcpl)
                if sid >= 0:
                    lib.H5Sclose(sid)
                if close_mem:
                    lib.H5Tclose(mem)
    finally:
        lib.H5Fclose(fid)
