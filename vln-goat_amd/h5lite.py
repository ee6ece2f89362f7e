"""Minimal HDF5 access through the HDF5 C library (libhdf5, via ctypes) with the h5py surface the reference's feature readers use:

    with h5lite.File(path, 'r') as f:
        for key in f.keys():                      # P/data/dataset.py:811-818 (view features), :838-861 (REVERIE / SOON object features)
            ft = f[key][...]                      # whole dataset -> numpy
            for name, value in f[key].attrs.items(): ...

The reference reads its pre-extracted CLIP / ViT view features and object features with h5py; this image ships no h5py, but it does ship
the C library itself (/opt/conda/lib/libhdf5.so.103, HDF5 1.10.6).  The format work — superblock, B-trees, heaps, chunking, filters, type
conversion — is libhdf5's; this module only marshals handles and buffers.  `features.FeatureStore.from_hdf5` and
`rollout.ObjectStore.from_hdf5` prefer h5py when it is importable and fall back to this module.

Supported: datasets of integer / float element types of any rank (read whole, converted by the library to the native type of the same
class and size), attributes of those types, fixed- and variable-length string attributes / datasets (-> numpy object arrays of str, as
h5py returns them).  A minimal writer (`File(path, 'w').create_dataset(name, data)`, `.attrs[name] = value`) exists for test fixtures
and for converting stores.  Locating the library: $GOAT_LIBHDF5, then the loader's search path, then the conda tree of this image."""
import ctypes
import ctypes.util
import os

import numpy as np

hid_t = ctypes.c_int64           # HDF5 >= 1.10
hsize_t = ctypes.c_uint64
herr_t = ctypes.c_int

_LIB = [None]
_CANDIDATES = ('/opt/conda/lib/libhdf5.so.103', '/opt/conda/lib/libhdf5.so', '/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so',
               '/usr/lib/x86_64-linux-gnu/libhdf5_serial.so.103', '/usr/lib/x86_64-linux-gnu/libhdf5_serial.so')

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT = 0
H5S_ALL = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5_INDEX_NAME, H5_ITER_INC = 0, 0
H5T_VARIABLE = ctypes.c_size_t(-1).value
H5S_SCALAR = 0


class Hdf5Error(RuntimeError):
    pass


def available():
    try:
        _lib()
        return True
    except (OSError, Hdf5Error):
        return False


def _lib():
    if _LIB[0] is not None:
        return _LIB[0]
    names = [os.environ.get('GOAT_LIBHDF5'), ctypes.util.find_library('hdf5'), ctypes.util.find_library('hdf5_serial')] + list(_CANDIDATES)
    err = None
    for n in names:
        if not n:
            continue
        try:
            h = ctypes.CDLL(n)
        except OSError as e:
            err = e
            continue
        _declare(h)
        if h.H5open() < 0:
            raise Hdf5Error('H5open failed in %s' % n)
        maj, mi, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
        h.H5get_libversion(ctypes.byref(maj), ctypes.byref(mi), ctypes.byref(rel))
        if (maj.value, mi.value) < (1, 10):
            raise Hdf5Error('%s is HDF5 %d.%d: this binding assumes the 64-bit handles of HDF5 >= 1.10' % (n, maj.value, mi.value))
        h.H5Eset_auto2(0, None, None)             # errors come back as negative return values: they are raised here, not printed by the library
        _LIB[0] = h
        return h
    raise OSError('libhdf5 not found (set GOAT_LIBHDF5 to its path): %s' % err)


_ITER_CB = ctypes.CFUNCTYPE(herr_t, hid_t, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p)
_ATTR_CB = ctypes.CFUNCTYPE(herr_t, hid_t, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p)


def _declare(h):
    def f(name, res, *args):
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, list(args)
    vp, cp = ctypes.c_void_p, ctypes.c_char_p
    f('H5open', herr_t)
    f('H5get_libversion', herr_t, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint))
    f('H5Eset_auto2', herr_t, hid_t, vp, vp)
    f('H5Fopen', hid_t, cp, ctypes.c_uint, hid_t)
    f('H5Fcreate', hid_t, cp, ctypes.c_uint, hid_t, hid_t)
    f('H5Fclose', herr_t, hid_t)
    f('H5Literate', herr_t, hid_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(hsize_t), _ITER_CB, vp)
    f('H5Lexists', ctypes.c_int, hid_t, cp, hid_t)
    f('H5Oopen', hid_t, hid_t, cp, hid_t)
    f('H5Oclose', herr_t, hid_t)
    f('H5Iget_type', ctypes.c_int, hid_t)
    f('H5Dopen2', hid_t, hid_t, cp, hid_t)
    f('H5Dclose', herr_t, hid_t)
    f('H5Dget_space', hid_t, hid_t)
    f('H5Dget_type', hid_t, hid_t)
    f('H5Dread', herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, vp)
    f('H5Dcreate2', hid_t, hid_t, cp, hid_t, hid_t, hid_t, hid_t, hid_t)
    f('H5Dwrite', herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, vp)
    f('H5Dvlen_reclaim', herr_t, hid_t, hid_t, hid_t, vp)
    f('H5Sget_simple_extent_ndims', ctypes.c_int, hid_t)
    f('H5Sget_simple_extent_dims', ctypes.c_int, hid_t, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t))
    f('H5Sget_simple_extent_npoints', ctypes.c_int64, hid_t)
    f('H5Screate_simple', hid_t, ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t))
    f('H5Screate', hid_t, ctypes.c_int)
    f('H5Sclose', herr_t, hid_t)
    f('H5Tget_class', ctypes.c_int, hid_t)
    f('H5Tget_size', ctypes.c_size_t, hid_t)
    f('H5Tget_sign', ctypes.c_int, hid_t)
    f('H5Tis_variable_str', ctypes.c_int, hid_t)
    f('H5Tcopy', hid_t, hid_t)
    f('H5Tset_size', herr_t, hid_t, ctypes.c_size_t)
    f('H5Tclose', herr_t, hid_t)
    f('H5Tset_fields', herr_t, hid_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t)
    f('H5Tset_ebias', herr_t, hid_t, ctypes.c_size_t)
    f('H5Aiterate2', herr_t, hid_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(hsize_t), _ATTR_CB, vp)
    f('H5Aopen', hid_t, hid_t, cp, hid_t)
    f('H5Aclose', herr_t, hid_t)
    f('H5Aget_space', hid_t, hid_t)
    f('H5Aget_type', hid_t, hid_t)
    f('H5Aread', herr_t, hid_t, hid_t, vp)
    f('H5Acreate2', hid_t, hid_t, cp, hid_t, hid_t, hid_t, hid_t)
    f('H5Awrite', herr_t, hid_t, hid_t, vp)


def _native(name):
    """a predefined native type handle (a global of the library, valid after H5open)"""
    return hid_t.in_dll(_lib(), name + '_g').value


_NATIVE_OF = {('f', 4): 'H5T_NATIVE_FLOAT', ('f', 8): 'H5T_NATIVE_DOUBLE',
              ('i', 1): 'H5T_NATIVE_INT8', ('i', 2): 'H5T_NATIVE_INT16', ('i', 4): 'H5T_NATIVE_INT32', ('i', 8): 'H5T_NATIVE_INT64',
              ('u', 1): 'H5T_NATIVE_UINT8', ('u', 2): 'H5T_NATIVE_UINT16', ('u', 4): 'H5T_NATIVE_UINT32', ('u', 8): 'H5T_NATIVE_UINT64'}


def _check(rc, what):
    if rc < 0:
        raise Hdf5Error('libhdf5: %s failed (%d)' % (what, rc))
    return rc


def _shape(space):
    h = _lib()
    nd = _check(h.H5Sget_simple_extent_ndims(space), 'H5Sget_simple_extent_ndims')
    if nd == 0:
        return ()
    dims = (hsize_t * nd)()
    _check(h.H5Sget_simple_extent_dims(space, dims, None), 'H5Sget_simple_extent_dims')
    return tuple(int(d) for d in dims)


def _read(obj, ftype, space, reader, what):
    """the whole object (dataset / attribute) -> numpy, by the element class of its file type"""
    h = _lib()
    shape = _shape(space)
    n = int(np.prod(shape)) if shape else 1
    cls = h.H5Tget_class(ftype)
    size = int(h.H5Tget_size(ftype))
    if cls in (H5T_INTEGER, H5T_FLOAT):
        if cls == H5T_FLOAT:
            kind = 'f'
            if size == 2:                        # half precision on disk: read as float32 (the library converts)
                size = 4
        else:
            kind = 'i' if h.H5Tget_sign(ftype) == 1 else 'u'
        key = (kind, size)
        if key not in _NATIVE_OF:
            raise Hdf5Error('%s: %d-byte %s elements are not supported' % (what, size, kind))
        out = np.empty(shape, dtype=np.dtype('%s%d' % (kind, size)))
        if n:
            _check(reader(_native(_NATIVE_OF[key]), out.ctypes.data_as(ctypes.c_void_p)), what)
        return out
    if cls == H5T_STRING:
        mem = _check(h.H5Tcopy(_native('H5T_C_S1')), 'H5Tcopy')
        try:
            if h.H5Tis_variable_str(ftype) > 0:
                _check(h.H5Tset_size(mem, H5T_VARIABLE), 'H5Tset_size')
                buf = (ctypes.c_char_p * max(n, 1))()
                if n:
                    _check(reader(mem, ctypes.cast(buf, ctypes.c_void_p)), what)
                vals = [(buf[i] or b'').decode('utf-8', 'replace') for i in range(n)]
                if n:
                    h.H5Dvlen_reclaim(mem, space, H5P_DEFAULT, ctypes.cast(buf, ctypes.c_void_p))
            else:
                _check(h.H5Tset_size(mem, size), 'H5Tset_size')
                raw = ctypes.create_string_buffer(max(n, 1) * size)
                if n:
                    _check(reader(mem, ctypes.cast(raw, ctypes.c_void_p)), what)
                vals = [raw.raw[i * size:(i + 1) * size].split(b'\0', 1)[0].decode('utf-8', 'replace') for i in range(n)]
        finally:
            h.H5Tclose(mem)
        if not shape:
            return vals[0]
        out = np.empty(n, dtype=object)
        out[:] = vals
        return out.reshape(shape)
    raise Hdf5Error('%s: HDF5 type class %d is not supported (integer, float and string elements are)' % (what, cls))


def _np_to_file(arr):
    """numpy value -> (contiguous array, native type handle to close or None, mem type) for a write"""
    a = np.ascontiguousarray(arr)
    if a.dtype == np.float16:                    # IEEE half, defined field by field as h5py does (HDF5 has no predefined half type)
        h = _lib()
        t = _check(h.H5Tcopy(_native('H5T_IEEE_F32LE')), 'H5Tcopy')
        _check(h.H5Tset_fields(t, 15, 10, 5, 0, 10), 'H5Tset_fields')
        _check(h.H5Tset_size(t, 2), 'H5Tset_size')
        _check(h.H5Tset_ebias(t, 15), 'H5Tset_ebias')
        return a, t, t
    if a.dtype.kind in 'fiu':
        key = (a.dtype.kind, a.dtype.itemsize)
        if key not in _NATIVE_OF:
            raise Hdf5Error('cannot write dtype %s' % a.dtype)
        return a, None, _native(_NATIVE_OF[key])
    raise Hdf5Error('cannot write dtype %s' % a.dtype)


def _space_for(shape):
    h = _lib()
    if len(shape) == 0:
        return _check(h.H5Screate(H5S_SCALAR), 'H5Screate')
    dims = (hsize_t * len(shape))(*shape)
    return _check(h.H5Screate_simple(len(shape), dims, None), 'H5Screate_simple')


class _Attrs:
    def __init__(self, owner):
        self.owner = owner

    def keys(self):
        names = []

        def cb(loc, name, info, data):
            names.append(name.decode())
            return 0
        idx = hsize_t(0)
        _check(_lib().H5Aiterate2(self.owner.id, H5_INDEX_NAME, H5_ITER_INC, ctypes.byref(idx), _ATTR_CB(cb), None), 'H5Aiterate2')
        return names

    def __iter__(self):
        return iter(self.keys())

    def __contains__(self, name):
        return name in self.keys()

    def __getitem__(self, name):
        h = _lib()
        a = h.H5Aopen(self.owner.id, name.encode(), H5P_DEFAULT)
        if a < 0:
            raise KeyError(name)
        try:
            t, s = h.H5Aget_type(a), h.H5Aget_space(a)
            try:
                return _read(a, t, s, lambda mem, buf: h.H5Aread(a, mem, buf), 'H5Aread(%s)' % name)
            finally:
                h.H5Tclose(t)
                h.H5Sclose(s)
        finally:
            h.H5Aclose(a)

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def __setitem__(self, name, value):
        h = _lib()
        if isinstance(value, (list, tuple)) and value and all(isinstance(v, str) for v in value) or \
                (isinstance(value, np.ndarray) and value.dtype.kind in 'OUS'):
            vals = [str(v).encode() for v in np.asarray(value, dtype=object).reshape(-1)]
            mem = _check(h.H5Tcopy(_native('H5T_C_S1')), 'H5Tcopy')
            _check(h.H5Tset_size(mem, H5T_VARIABLE), 'H5Tset_size')
            space = _space_for((len(vals),))
            buf = (ctypes.c_char_p * len(vals))(*vals)
            a = _check(h.H5Acreate2(self.owner.id, name.encode(), mem, space, H5P_DEFAULT, H5P_DEFAULT), 'H5Acreate2')
            try:
                _check(h.H5Awrite(a, mem, ctypes.cast(buf, ctypes.c_void_p)), 'H5Awrite')
            finally:
                h.H5Aclose(a)
                h.H5Sclose(space)
                h.H5Tclose(mem)
            return
        arr, tmp, mem = _np_to_file(np.asarray(value))
        space = _space_for(arr.shape)
        a = _check(h.H5Acreate2(self.owner.id, name.encode(), mem, space, H5P_DEFAULT, H5P_DEFAULT), 'H5Acreate2')
        try:
            _check(h.H5Awrite(a, mem, arr.ctypes.data_as(ctypes.c_void_p)), 'H5Awrite')
        finally:
            h.H5Aclose(a)
            h.H5Sclose(space)
            if tmp is not None:
                h.H5Tclose(tmp)


class Dataset:
    def __init__(self, did, name):
        self.id, self.name = did, name
        self.attrs = _Attrs(self)
        h = _lib()
        s = h.H5Dget_space(did)
        try:
            self.shape = _shape(s)
        finally:
            h.H5Sclose(s)

    def __getitem__(self, key):
        h = _lib()
        t, s = h.H5Dget_type(self.id), h.H5Dget_space(self.id)
        try:
            full = _read(self.id, t, s, lambda mem, buf: h.H5Dread(self.id, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf), 'H5Dread(%s)' % self.name)
        finally:
            h.H5Tclose(t)
            h.H5Sclose(s)
        if key is Ellipsis or key == () or (isinstance(key, slice) and key == slice(None)):
            return full
        return full[key]               # (whole-dataset read, then numpy indexing: the reference reads `f[key][...]`)

    def close(self):
        if self.id is not None:
            _lib().H5Dclose(self.id)
            self.id = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 (interpreter shutdown)
            pass


class File:
    """h5py.File(path, mode) for mode 'r' (and 'w': fixtures / conversion) over the root group."""

    def __init__(self, path, mode='r'):
        h = _lib()
        self.path, self.mode = str(path), mode
        if mode == 'r':
            if not os.path.exists(self.path):
                raise FileNotFoundError(self.path)
            self.id = h.H5Fopen(self.path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == 'w':
            self.id = h.H5Fcreate(self.path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r' or 'w'")
        if self.id < 0:
            raise Hdf5Error('cannot open %s as an HDF5 file (mode %s)' % (self.path, mode))
        self.attrs = _Attrs(self)
        self._open = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        for d in self._open:
            d.close()
        self._open = []
        if self.id is not None and self.id >= 0:
            _lib().H5Fclose(self.id)
        self.id = None

    def keys(self):
        names = []

        def cb(group, name, info, data):
            names.append(name.decode())
            return 0
        idx = hsize_t(0)
        _check(_lib().H5Literate(self.id, H5_INDEX_NAME, H5_ITER_INC, ctypes.byref(idx), _ITER_CB(cb), None), 'H5Literate')
        return names

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def __contains__(self, name):
        return _lib().H5Lexists(self.id, str(name).encode(), H5P_DEFAULT) > 0

    def __getitem__(self, name):
        did = _lib().H5Dopen2(self.id, str(name).encode(), H5P_DEFAULT)
        if did < 0:
            raise KeyError(name)
        d = Dataset(did, str(name))
        self._open.append(d)
        return d

    def create_dataset(self, name, data=None, dtype=None):
        h = _lib()
        arr = np.asarray(data, dtype=dtype)
        arr, tmp, mem = _np_to_file(arr)
        space = _space_for(arr.shape)
        did = _check(h.H5Dcreate2(self.id, str(name).encode(), mem, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 'H5Dcreate2(%s)' % name)
        try:
            if arr.size:
                _check(h.H5Dwrite(did, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(ctypes.c_void_p)), 'H5Dwrite(%s)' % name)
        finally:
            h.H5Sclose(space)
            if tmp is not None:
                h.H5Tclose(tmp)
        d = Dataset(did, str(name))
        self._open.append(d)
        return d


def open_file(path, mode='r'):
    """h5py.File when h5py is importable, else this module's File over libhdf5; raises ImportError when neither is there."""
    try:
        import h5py
        return h5py.File(path, mode)
    except ImportError:
        pass
    if not available():
        raise ImportError('reading %s needs h5py or the HDF5 C library (libhdf5.so; set GOAT_LIBHDF5 to its path): neither was found' % path)
    return File(path, mode)
