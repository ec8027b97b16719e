"""HDF5 volumes without h5py: a thin ctypes binding to the HDF5 C library (libhdf5 1.10+) for the two things the reference does
with h5py - read one dataset into a numpy array (`utilities/base_data_utils.py:179-214`) and write one chunked, gzip-compressed
dataset (`:351-356`).  This image ships libhdf5 (conda's) but no h5py for the torch interpreter; where h5py IS importable it is
used instead (`base_data_utils._h5_backend`).  Host-side I/O either side of the accelerated path - no GPU involved.

Only fixed-size numeric datasets (the volumes and label / probability outputs of the reference): int / uint 8-64, float 16 / 32 /
64, little endian.  float16 uses the same custom 2-byte IEEE type h5py registers (5 exponent bits, bias 15)."""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os
from pathlib import Path

import numpy as np

_hid = C.c_int64
_hsize = C.c_uint64
_lib = None
_H5P_DEFAULT = 0
_H5S_ALL = 0
_H5F_ACC_RDONLY, _H5F_ACC_TRUNC = 0, 2
_H5T_INTEGER, _H5T_FLOAT = 0, 1
_H5T_SGN_NONE = 0
_H5D_CHUNKED = 2


class Hdf5Error(RuntimeError):
    pass


def _candidates():
    env = os.environ.get("VOLSEG_LIBHDF5")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    yield from ("libhdf5.so", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so")


def library():
    """The loaded libhdf5 (raises Hdf5Error when there is none)."""
    global _lib
    if _lib is not None:
        return _lib
    last = None
    for name in _candidates():
        try:
            lib = C.CDLL(name)
        except OSError as e:
            last = e
            continue
        if lib.H5open() < 0:
            continue
        sig = {
            "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]), "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]), "H5Fclose": (C.c_int, [_hid]),
            "H5Screate_simple": (_hid, [C.c_int, C.POINTER(_hsize), C.POINTER(_hsize)]), "H5Sclose": (C.c_int, [_hid]),
            "H5Sget_simple_extent_ndims": (C.c_int, [_hid]), "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(_hsize), C.POINTER(_hsize)]),
            "H5Pcreate": (_hid, [_hid]), "H5Pclose": (C.c_int, [_hid]), "H5Pset_chunk": (C.c_int, [_hid, C.c_int, C.POINTER(_hsize)]),
            "H5Pset_deflate": (C.c_int, [_hid, C.c_uint]), "H5Pset_create_intermediate_group": (C.c_int, [_hid, C.c_uint]),
            "H5Pget_layout": (C.c_int, [_hid]), "H5Pget_chunk": (C.c_int, [_hid, C.c_int, C.POINTER(_hsize)]),
            "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]), "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]),
            "H5Dclose": (C.c_int, [_hid]), "H5Dget_space": (_hid, [_hid]), "H5Dget_type": (_hid, [_hid]), "H5Dget_create_plist": (_hid, [_hid]),
            "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]), "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
            "H5Tcopy": (_hid, [_hid]), "H5Tclose": (C.c_int, [_hid]), "H5Tget_class": (C.c_int, [_hid]), "H5Tget_size": (C.c_size_t, [_hid]),
            "H5Tget_sign": (C.c_int, [_hid]), "H5Tset_fields": (C.c_int, [_hid, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]),
            "H5Tset_size": (C.c_int, [_hid, C.c_size_t]), "H5Tset_ebias": (C.c_int, [_hid, C.c_size_t]),
            "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]), "H5Eset_auto2": (C.c_int, [_hid, C.c_void_p, C.c_void_p]),
        }
        for fn, (res, args) in sig.items():
            f = getattr(lib, fn)
            f.restype, f.argtypes = res, args
        lib.H5Eset_auto2(0, None, None)         # errors come back as return codes; no stack dump on stderr
        _lib = lib
        return lib
    raise Hdf5Error(f"no usable HDF5 C library found (set VOLSEG_LIBHDF5 to libhdf5.so): {last}")


def available() -> bool:
    try:
        library()
        return True
    except Hdf5Error:
        return False


def _g(name: str) -> int:
    return _hid.in_dll(library(), name).value


_NATIVE = {"uint8": "H5T_NATIVE_UINT8_g", "int8": "H5T_NATIVE_INT8_g", "uint16": "H5T_NATIVE_UINT16_g", "int16": "H5T_NATIVE_INT16_g",
           "uint32": "H5T_NATIVE_UINT32_g", "int32": "H5T_NATIVE_INT32_g", "uint64": "H5T_NATIVE_UINT64_g", "int64": "H5T_NATIVE_INT64_g",
           "float32": "H5T_NATIVE_FLOAT_g", "float64": "H5T_NATIVE_DOUBLE_g"}


def _type_for(dtype: np.dtype):
    """(hid, owned): the HDF5 type of a numpy dtype; float16 is built the way h5py builds it."""
    lib = library()
    name = np.dtype(dtype).name
    if name == "bool":
        name = "uint8"
    if name in _NATIVE:
        return _g(_NATIVE[name]), False
    if name == "float16":
        t = lib.H5Tcopy(_g("H5T_IEEE_F32LE_g"))
        if t < 0 or lib.H5Tset_fields(t, 15, 10, 5, 0, 10) < 0 or lib.H5Tset_size(t, 2) < 0 or lib.H5Tset_ebias(t, 15) < 0:
            raise Hdf5Error("could not build the 16-bit float type")
        return t, True
    raise Hdf5Error(f"unsupported dtype {dtype} (fixed-size integer and float volumes only)")


def _dtype_of(tid: int) -> np.dtype:
    lib = library()
    cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
    if cls == _H5T_INTEGER and size in (1, 2, 4, 8):
        return np.dtype(("u" if lib.H5Tget_sign(tid) == _H5T_SGN_NONE else "i") + str(size))
    if cls == _H5T_FLOAT and size in (2, 4, 8):
        return np.dtype("f" + str(size))
    raise Hdf5Error(f"unsupported HDF5 datatype (class {cls}, {size} bytes): fixed-size integer and float volumes only")


def exists(path, internal_path: str) -> bool:
    """Whether `internal_path` names a link in the file (every component checked: H5Lexists fails on a missing parent)."""
    lib = library()
    f = lib.H5Fopen(str(path).encode(), _H5F_ACC_RDONLY, _H5P_DEFAULT)
    if f < 0:
        raise Hdf5Error(f"cannot open {path}")
    try:
        parts = [p for p in internal_path.split("/") if p]
        for i in range(len(parts)):
            if lib.H5Lexists(f, ("/" + "/".join(parts[:i + 1])).encode(), _H5P_DEFAULT) <= 0:
                return False
        return True
    finally:
        lib.H5Fclose(f)


def read_dataset(path, internal_path: str = "/data"):
    """-> (numpy array, chunk shape or None), like `h5py.File(path)[internal_path][()]` and `.chunks`."""
    lib = library()
    f = lib.H5Fopen(str(path).encode(), _H5F_ACC_RDONLY, _H5P_DEFAULT)
    if f < 0:
        raise Hdf5Error(f"cannot open {path}")
    d = s = t = pl = mt = -1
    owned = False
    try:
        d = lib.H5Dopen2(f, internal_path.encode(), _H5P_DEFAULT)
        if d < 0:
            raise KeyError(f"{internal_path!r} not found in {path}")
        s, t = lib.H5Dget_space(d), lib.H5Dget_type(d)
        nd = lib.H5Sget_simple_extent_ndims(s)
        dims = (_hsize * max(nd, 1))()
        if nd > 0:
            lib.H5Sget_simple_extent_dims(s, dims, None)
        dtype = _dtype_of(t)
        out = np.empty(tuple(int(dims[i]) for i in range(nd)), dtype=dtype)
        mt, owned = _type_for(dtype)
        if out.size and lib.H5Dread(d, mt, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)) < 0:
            raise Hdf5Error(f"reading {internal_path!r} from {path} failed")
        chunks = None
        pl = lib.H5Dget_create_plist(d)
        if pl >= 0 and lib.H5Pget_layout(pl) == _H5D_CHUNKED and nd > 0:
            cd = (_hsize * nd)()
            if lib.H5Pget_chunk(pl, nd, cd) >= 0:
                chunks = tuple(int(c) for c in cd)
        return out, chunks
    finally:
        if owned and mt >= 0: lib.H5Tclose(mt)
        if pl >= 0: lib.H5Pclose(pl)
        if t >= 0: lib.H5Tclose(t)
        if s >= 0: lib.H5Sclose(s)
        if d >= 0: lib.H5Dclose(d)
        lib.H5Fclose(f)


def guess_chunk(shape, itemsize: int):
    """A chunk shape of 16 KiB - 1 MiB for `chunks=True` (h5py picks one by a similar halving rule; the exact shape is not part of
    any result): halve the largest axis until the chunk is <= 512 KiB."""
    chunk = [max(1, int(n)) for n in shape]
    while np.prod(chunk, dtype=np.int64) * itemsize > 512 * 1024:
        i = int(np.argmax(chunk))
        if chunk[i] == 1:
            break
        chunk[i] = (chunk[i] + 1) // 2
    return tuple(chunk)


def write_dataset(path, internal_path: str, data: np.ndarray, chunks=True, compression: str | None = "gzip", level: int = 4) -> None:
    """`h5py.File(path, "w").create_dataset(internal_path, data=data, chunks=chunks, compression=compression)`: a new file with
    one dataset; parent groups are created; gzip needs chunking (as in HDF5 itself)."""
    lib = library()
    data = np.ascontiguousarray(data)
    if data.dtype == np.bool_:
        data = data.view(np.uint8)
    f = lib.H5Fcreate(str(path).encode(), _H5F_ACC_TRUNC, _H5P_DEFAULT, _H5P_DEFAULT)
    if f < 0:
        raise Hdf5Error(f"cannot create {path}")
    s = d = dcpl = lcpl = t = -1
    owned = False
    try:
        nd = data.ndim
        dims = (_hsize * max(nd, 1))(*data.shape)
        s = lib.H5Screate_simple(nd, dims, None)
        t, owned = _type_for(data.dtype)
        dcpl = lib.H5Pcreate(_g("H5P_CLS_DATASET_CREATE_ID_g"))
        lcpl = lib.H5Pcreate(_g("H5P_CLS_LINK_CREATE_ID_g"))
        lib.H5Pset_create_intermediate_group(lcpl, 1)
        want_chunks = bool(chunks) or bool(compression)
        if want_chunks and nd > 0 and data.size > 0:
            shape = guess_chunk(data.shape, data.itemsize) if chunks is True or not chunks else tuple(int(c) for c in chunks)
            if lib.H5Pset_chunk(dcpl, nd, (_hsize * nd)(*shape)) < 0:
                raise Hdf5Error(f"bad chunk shape {shape}")
            if compression:
                if compression != "gzip":
                    raise Hdf5Error(f"compression {compression!r}: only gzip is built")
                if lib.H5Pset_deflate(dcpl, int(level)) < 0:
                    raise Hdf5Error("the deflate filter is not available in this HDF5 build")
        d = lib.H5Dcreate2(f, internal_path.encode(), t, s, lcpl, dcpl, _H5P_DEFAULT)
        if d < 0:
            raise Hdf5Error(f"cannot create dataset {internal_path!r} in {path}")
        if data.size and lib.H5Dwrite(d, t, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, data.ctypes.data_as(C.c_void_p)) < 0:
            raise Hdf5Error(f"writing {internal_path!r} to {path} failed")
    finally:
        if d >= 0: lib.H5Dclose(d)
        if owned and t >= 0: lib.H5Tclose(t)
        if lcpl >= 0: lib.H5Pclose(lcpl)
        if dcpl >= 0: lib.H5Pclose(dcpl)
        if s >= 0: lib.H5Sclose(s)
        if lib.H5Fclose(f) < 0:
            raise Hdf5Error(f"closing {path} failed")


def is_hdf5(path) -> bool:
    try:
        with open(Path(path), "rb") as fh:
            return fh.read(8) == b"\x89HDF\r\n\x1a\n"
    except OSError:
        return False
