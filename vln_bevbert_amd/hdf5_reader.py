"""Read-only access to the reference's HDF5 feature files through libhdf5's C API (ctypes), for environments without h5py.

The reference keeps one dataset per viewpoint in the root group of three files (precompute_features/grid_mp3d_clip.py:168-180
``create_dataset(key, data=..., dtype='float16', compression='gzip')``; grid_depth.py:122-131 ``create_dataset(key,
data=depth_item)``; grid_sem.py:146-155 ``dtype='uint8', compression='gzip'``) and reads them back with
``h5py.File(path, 'r')[key][...]`` (map_nav_src/utils/data.py:22-27, pretrain_src/data/dataset.py:110-118).  ``open_file``
returns an object with that much of h5py's interface -- ``keys()``, ``in``, ``[key][...]``, ``[key].shape / .dtype`` -- backed by
h5py when it is importable and by libhdf5 otherwise.  Elements are read in the file's own datatype (no conversion inside the
library: a dataset stored as IEEE half / single / double or as an integer comes back as the numpy dtype of the same layout),
chunking and the gzip filter are the library's business.

Only ``feature_cache.convert_hdf5`` uses this: the one-off conversion into the sharded cache.  Nothing on the training or
rollout path opens an HDF5 file."""
import ctypes
import ctypes.util
import glob
import os

import numpy as np

_H5F_ACC_RDONLY, _H5P_DEFAULT, _H5S_ALL = 0, 0, 0
_H5_INDEX_NAME, _H5_ITER_INC = 0, 0
_H5T_INTEGER, _H5T_FLOAT = 0, 1
_H5T_ORDER_LE = 0
_H5T_SGN_NONE = 0
_hid = ctypes.c_int64                     # hid_t since HDF5 1.10


class Hdf5Error(RuntimeError):
    pass


def _find_library():
    cand = [os.environ.get("BEVBERT_LIBHDF5"), ctypes.util.find_library("hdf5")]
    for root in ("/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", "/usr/lib/x86_64-linux-gnu/hdf5/serial", "/usr/local/lib"):
        cand += sorted(glob.glob(os.path.join(root, "libhdf5.so*")) + glob.glob(os.path.join(root, "libhdf5_serial.so*")))
    errors = []
    for c in cand:
        if not c:
            continue
        try:
            return ctypes.CDLL(c), c
        except OSError as e:
            errors.append(f"{c}: {e}")
    raise Hdf5Error("no h5py and no loadable libhdf5 (set BEVBERT_LIBHDF5 to the shared library): " + "; ".join(errors[:3]))


_LIB = None


def _lib():
    """libhdf5 with the prototypes of the dozen calls used here; checks the library is a 1.10+ (64-bit hid_t)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    L, path = _find_library()
    P, I, U, SZ = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_size_t
    protos = {
        "H5open": (I, []), "H5get_libversion": (I, [P, P, P]),
        "H5Fopen": (_hid, [ctypes.c_char_p, U, _hid]), "H5Fclose": (I, [_hid]),
        "H5Literate": (I, [_hid, I, I, P, P, P]),
        "H5Lexists": (I, [_hid, ctypes.c_char_p, _hid]),
        "H5Dopen2": (_hid, [_hid, ctypes.c_char_p, _hid]), "H5Dclose": (I, [_hid]),
        "H5Dget_space": (_hid, [_hid]), "H5Dget_type": (_hid, [_hid]),
        "H5Dread": (I, [_hid, _hid, _hid, _hid, _hid, P]),
        "H5Sget_simple_extent_ndims": (I, [_hid]), "H5Sget_simple_extent_dims": (I, [_hid, P, P]), "H5Sclose": (I, [_hid]),
        "H5Tget_class": (I, [_hid]), "H5Tget_size": (SZ, [_hid]), "H5Tget_sign": (I, [_hid]), "H5Tget_order": (I, [_hid]),
        "H5Tclose": (I, [_hid]),
        "H5Eset_auto2": (I, [_hid, P, P]),
    }
    if not hasattr(L, "H5Literate"):        # HDF5 >= 1.12 exports the versioned names only; the callback ignores the info struct
        for alt in ("H5Literate2", "H5Literate1"):
            if hasattr(L, alt):
                L.H5Literate = getattr(L, alt)
                break
    for name, (res, args) in protos.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    if L.H5open() < 0:
        raise Hdf5Error(f"{path}: H5open failed")
    v = (ctypes.c_uint * 3)()
    L.H5get_libversion(ctypes.byref(v, 0), ctypes.byref(v, 4), ctypes.byref(v, 8))
    if (v[0], v[1]) < (1, 10):
        raise Hdf5Error(f"{path} is HDF5 {v[0]}.{v[1]}.{v[2]}: 1.10 or newer is needed (64-bit identifiers)")
    L.H5Eset_auto2(0, None, None)          # errors are reported through return codes (raised below), not printed by the library
    L.version = tuple(v)
    L.path = path
    _LIB = L
    return L


_ITER_CB = ctypes.CFUNCTYPE(ctypes.c_int, _hid, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p)


class _Dataset:
    """One dataset of the root group: ``shape``, ``dtype`` and ``[...]`` (the whole array) like h5py's."""
    _id = -1

    def __init__(self, f, name):
        L = self._L = f._L
        self.name = name
        self._id = L.H5Dopen2(f._id, name.encode(), _H5P_DEFAULT)
        if self._id < 0:
            raise KeyError(f"{name!r} is not a dataset of {f.path}")
        sp, tp = L.H5Dget_space(self._id), L.H5Dget_type(self._id)
        try:
            nd = L.H5Sget_simple_extent_ndims(sp)
            if nd < 0:
                raise Hdf5Error(f"{name}: no simple dataspace")
            dims = (ctypes.c_uint64 * max(nd, 1))()
            L.H5Sget_simple_extent_dims(sp, dims, None)
            self.shape = tuple(int(d) for d in dims[:nd])
            cls, size = L.H5Tget_class(tp), int(L.H5Tget_size(tp))
            if L.H5Tget_order(tp) != _H5T_ORDER_LE and size > 1:
                raise Hdf5Error(f"{name}: big-endian elements are not supported")
            if cls == _H5T_FLOAT and size in (2, 4, 8):
                self.dtype = np.dtype(f"<f{size}")
            elif cls == _H5T_INTEGER and size in (1, 2, 4, 8):
                self.dtype = np.dtype(("<u" if L.H5Tget_sign(tp) == _H5T_SGN_NONE else "<i") + str(size))
            else:
                raise Hdf5Error(f"{name}: datatype class {cls} of {size} bytes is not one the feature files use")
        finally:
            L.H5Sclose(sp)
            L.H5Tclose(tp)

    def __getitem__(self, idx):
        if idx is not Ellipsis and idx != () and idx != slice(None):
            return self[...][idx]
        L = self._L
        out = np.empty(self.shape, dtype=self.dtype)
        tp = L.H5Dget_type(self._id)       # memory type = file type: the library copies (and inflates), it does not convert
        try:
            if out.size and L.H5Dread(self._id, tp, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, out.ctypes.data_as(ctypes.c_void_p)) < 0:
                raise Hdf5Error(f"{self.name}: H5Dread failed (a filter this libhdf5 was built without?)")
        finally:
            L.H5Tclose(tp)
        return out

    def close(self):
        if self._id >= 0:
            self._L.H5Dclose(self._id)
            self._id = -1

    __del__ = close


class _File:
    """``h5py.File(path, 'r')`` as far as the feature readers use it; a context manager."""
    _id = -1

    def __init__(self, path):
        self._L = _lib()
        self.path = path
        if not os.path.isfile(path):
            raise FileNotFoundError(path)
        self._id = self._L.H5Fopen(os.fsencode(path), _H5F_ACC_RDONLY, _H5P_DEFAULT)
        if self._id < 0:
            raise Hdf5Error(f"{path}: not an HDF5 file this libhdf5 ({self._L.path}) can open")

    def keys(self):
        names = []

        def visit(_group, name, _info, _data):
            names.append(name.decode())
            return 0
        idx = ctypes.c_uint64(0)
        cb = _ITER_CB(visit)
        if self._L.H5Literate(self._id, _H5_INDEX_NAME, _H5_ITER_INC, ctypes.byref(idx), cb, None) < 0:
            raise Hdf5Error(f"{self.path}: cannot list the root group")
        return names                           # increasing by name, the order h5py iterates a group in

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def __contains__(self, name):
        return self._L.H5Lexists(self._id, name.encode(), _H5P_DEFAULT) > 0

    def __getitem__(self, name):
        return _Dataset(self, name)

    def close(self):
        if self._id >= 0:
            self._L.H5Fclose(self._id)
            self._id = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    __del__ = close


def backend():
    """'h5py', 'libhdf5 <version> (<path>)' or None: what ``open_file`` will use."""
    try:
        import h5py  # noqa: F401
        return "h5py"
    except ImportError:
        pass
    try:
        L = _lib()
        return "libhdf5 %d.%d.%d (%s)" % (L.version + (L.path,))
    except Hdf5Error:
        return None


def open_file(path, prefer=None):
    """Open an HDF5 file for reading.  ``prefer='libhdf5'`` skips h5py (the tests compare the two)."""
    if prefer != "libhdf5":
        try:
            import h5py
            return h5py.File(path, "r")
        except ImportError:
            pass
    return _File(path)
