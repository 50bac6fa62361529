"""On-disk format of the reference's result files.

The reference stores ``OptimizationHistory`` and the convergence histories with JLD2
(src/ipopt_optimal_control.jl:74-104, :223-241; src/Tests/test_convergence.jl:76-81).  A JLD2 file IS an HDF5
file: a 512-byte user block that names the format, then HDF5 structures of the 1.8 generation (superblock 2, version-2
object headers, link messages).  This module writes and reads such files through the HDF5 C library itself
(``libhdf5``, bound with ctypes -- there is no h5py in the image), in the encoding JLD2 gives the types these files
hold:

    Int64 / Float64 scalars          scalar dataspace, H5T_STD_I64LE / H5T_IEEE_F64LE
    String                           scalar dataspace, variable-length UTF-8 string
    Array{Float64,N}, Array{Int64,N} simple dataspace with the dimensions REVERSED (Julia is column-major, HDF5
                                     row-major; the bytes are the Julia array's), contiguous layout
    Array{ComplexF64,N}              the same with the compound {re: F64, im: F64}
    Vector{Vector{Float64}}, Vector{Array{Float64,4}}
                                     dataset of object references (H5T_STD_REF_OBJ), one per element; the elements
                                     are datasets of their own (linked under ``_refs/``: libhdf5 only makes
                                     references to linked objects)
    nested dict                      HDF5 group

What is NOT reproduced: the committed datatypes with ``julia_type`` attributes that JLD2 adds for Julia structs
(``Setup/schrodinger_prob``, ``Setup/controls``) and for ``Dict`` values.  Plain datasets need none of it: JLD2 maps
the HDF5 types above back to the Julia types named on the left, which is what ``read_optimization_history`` consumes.
STATUS: HDF5-valid and self-round-tripping; a JLD2 round trip is UNVERIFIED -- nothing here could be checked against a
Julia session (none in the image), and the reference holds no ``.jld2`` file to read.  Known points where JLD2 may
differ from this reading of its format: a bare ``{re, im}`` compound without JLD2's committed datatype loads as a
NamedTuple / reconstructed type rather than ``ComplexF64`` (only ``Setup/target`` is complex; the nine
``OptimizationHistory`` keys are real); an untyped reference dataset loads as ``Vector{Any}`` (which the
``OptimizationHistory`` constructor converts); groups with more than 8 links use dense (fractal-heap) link storage under
the 1.8 format bounds, which needs a JLD2 recent enough to read it (the nine keys sit in the root group: 9-10 links).
INTEGRATION.md section 3a lists the HDF5 objects of the nine keys.  The files are checked by reading them back
(tests/test_host.py) and, where the HDF5 tools are installed, with ``h5dump``.

The library is looked up as ``$QGD_HDF5_LIB``, then ``libhdf5.so`` on the loader path, then ``/opt/conda/lib``;
without it every entry point raises ``RuntimeError`` -- there is no silent fallback to another format.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os

import numpy as np

_JLD2_HEADER = b"HDF5-based Julia Data Format, version 0.1.0\x00 (Julia 1.9.0 64-bit LE)\x00"
_USERBLOCK = 512
_H5F_ACC_RDONLY, _H5F_ACC_TRUNC = 0, 2
_H5S_SCALAR = 0
_H5T_COMPOUND, _H5T_INTEGER, _H5T_FLOAT, _H5T_STRING, _H5T_REFERENCE = 6, 0, 1, 3, 7
_H5T_CSET_UTF8 = 1
_H5T_VARIABLE = C.c_size_t(-1).value
_H5R_OBJECT = 0
_H5F_LIBVER_V18 = 1
_H5I_GROUP, _H5I_DATASET = 2, 5
_H5_INDEX_NAME, _H5_ITER_INC = 0, 0
hid_t = C.c_int64

_lib = None


class _GInfo(C.Structure):
    """H5G_info_t"""
    _fields_ = [("storage_type", C.c_int), ("nlinks", C.c_uint64), ("max_corder", C.c_int64), ("mounted", C.c_int)]


def _load():
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.environ.get("QGD_HDF5_LIB"), ctypes.util.find_library("hdf5"), "/opt/conda/lib/libhdf5.so"]
    err = None
    for cand in cands:
        if not cand:
            continue
        try:
            L = C.CDLL(cand)
            break
        except OSError as exc:
            err = exc
    else:
        raise RuntimeError(f"libhdf5 not found (set QGD_HDF5_LIB): JLD2/HDF5 files cannot be read or written ({err})")
    sig = {
        "H5open": (C.c_int, []), "H5Eset_auto2": (C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
        "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (C.c_int, [hid_t]),
        "H5Pset_userblock": (C.c_int, [hid_t, C.c_uint64]), "H5Pset_libver_bounds": (C.c_int, [hid_t, C.c_int, C.c_int]),
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fclose": (C.c_int, [hid_t]),
        "H5Screate": (hid_t, [C.c_int]), "H5Screate_simple": (hid_t, [C.c_int, C.c_void_p, C.c_void_p]),
        "H5Sclose": (C.c_int, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Dclose": (C.c_int, [hid_t]),
        "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dread": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dget_type": (hid_t, [hid_t]), "H5Dget_space": (hid_t, [hid_t]),
        "H5Dvlen_reclaim": (C.c_int, [hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Tcopy": (hid_t, [hid_t]), "H5Tcreate": (hid_t, [C.c_int, C.c_size_t]), "H5Tclose": (C.c_int, [hid_t]),
        "H5Tinsert": (C.c_int, [hid_t, C.c_char_p, C.c_size_t, hid_t]), "H5Tset_size": (C.c_int, [hid_t, C.c_size_t]),
        "H5Tset_cset": (C.c_int, [hid_t, C.c_int]), "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]),
        "H5Tis_variable_str": (C.c_int, [hid_t]), "H5Tget_nmembers": (C.c_int, [hid_t]),
        "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gclose": (C.c_int, [hid_t]),
        "H5Gget_info": (C.c_int, [hid_t, C.c_void_p]),
        "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_char_p, C.c_size_t, hid_t]),
        "H5Oopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Oclose": (C.c_int, [hid_t]), "H5Iget_type": (C.c_int, [hid_t]),
        "H5Rcreate": (C.c_int, [C.c_void_p, hid_t, C.c_char_p, C.c_int, hid_t]),
        "H5Rdereference2": (hid_t, [hid_t, hid_t, C.c_int, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    if L.H5open() < 0:
        raise RuntimeError("H5open failed")
    L.H5Eset_auto2(0, None, None)           # errors are reported through return values below, not printed
    for g in ("H5T_IEEE_F64LE", "H5T_STD_I64LE", "H5T_NATIVE_DOUBLE", "H5T_NATIVE_INT64", "H5T_C_S1", "H5T_STD_REF_OBJ",
              "H5P_CLS_FILE_CREATE_ID", "H5P_CLS_FILE_ACCESS_ID"):
        setattr(L, "g_" + g, hid_t.in_dll(L, g + "_g").value)
    _lib = L
    return L


def available():
    try:
        _load()
        return True
    except RuntimeError:
        return False


def _chk(v, what):
    if v < 0:
        raise IOError(f"HDF5: {what} failed")
    return v


class _Writer:
    def __init__(self, filename):
        L = self.L = _load()
        fcpl = _chk(L.H5Pcreate(L.g_H5P_CLS_FILE_CREATE_ID), "H5Pcreate")
        fapl = _chk(L.H5Pcreate(L.g_H5P_CLS_FILE_ACCESS_ID), "H5Pcreate")
        _chk(L.H5Pset_userblock(fcpl, _USERBLOCK), "H5Pset_userblock")
        _chk(L.H5Pset_libver_bounds(fapl, _H5F_LIBVER_V18, _H5F_LIBVER_V18), "H5Pset_libver_bounds")
        self.filename = os.fspath(filename)
        self.f = _chk(L.H5Fcreate(self.filename.encode(), _H5F_ACC_TRUNC, fcpl, fapl), f"H5Fcreate({self.filename})")
        L.H5Pclose(fcpl); L.H5Pclose(fapl)
        self.cplx = _chk(L.H5Tcreate(_H5T_COMPOUND, 16), "H5Tcreate")
        L.H5Tinsert(self.cplx, b"re", 0, L.g_H5T_IEEE_F64LE); L.H5Tinsert(self.cplx, b"im", 8, L.g_H5T_IEEE_F64LE)
        self.vstr = _chk(L.H5Tcopy(L.g_H5T_C_S1), "H5Tcopy")
        L.H5Tset_size(self.vstr, _H5T_VARIABLE); L.H5Tset_cset(self.vstr, _H5T_CSET_UTF8)
        self.nref = 0

    def close(self):
        L = self.L
        L.H5Tclose(self.cplx); L.H5Tclose(self.vstr)
        _chk(L.H5Fclose(self.f), "H5Fclose")
        with open(self.filename, "r+b") as fh:        # the user block: what makes the HDF5 file a JLD2 file
            fh.write(_JLD2_HEADER.ljust(_USERBLOCK, b"\x00"))

    def _dataset(self, loc, name, ftype, dims, mtype, buf):
        L = self.L
        if dims is None:
            space = _chk(L.H5Screate(_H5S_SCALAR), "H5Screate")
        else:
            arr = (C.c_uint64 * len(dims))(*dims)
            space = _chk(L.H5Screate_simple(len(dims), arr, None), "H5Screate_simple")
        d = _chk(L.H5Dcreate2(loc, name.encode(), ftype, space, 0, 0, 0), f"H5Dcreate2({name})")
        if dims is None or all(dims):
            _chk(L.H5Dwrite(d, mtype, 0, 0, 0, buf), f"H5Dwrite({name})")
        L.H5Dclose(d); L.H5Sclose(space)

    def _array(self, loc, name, a):
        L = self.L
        a = np.asarray(a)
        if a.dtype == np.bool_ or np.issubdtype(a.dtype, np.integer):
            a, ft, mt = a.astype(np.int64), L.g_H5T_STD_I64LE, L.g_H5T_NATIVE_INT64
        elif np.issubdtype(a.dtype, np.complexfloating):
            a, ft, mt = a.astype(np.complex128), self.cplx, self.cplx
        elif np.issubdtype(a.dtype, np.floating):
            a, ft, mt = a.astype(np.float64), L.g_H5T_IEEE_F64LE, L.g_H5T_NATIVE_DOUBLE
        else:
            raise TypeError(f"{name}: dtype {a.dtype} has no JLD2 encoding here")
        if a.ndim == 0:
            buf = np.ascontiguousarray(a)
            return self._dataset(loc, name, ft, None, mt, buf.ctypes.data_as(C.c_void_p))
        # Julia array of shape a.shape in column-major order == C-ordered array of the reversed shape
        buf = np.ascontiguousarray(a.T)
        self._dataset(loc, name, ft, tuple(reversed(a.shape)), mt, buf.ctypes.data_as(C.c_void_p))

    def _refs(self, loc, name, items):
        L = self.L
        if not L.H5Lexists(self.f, b"_refs", 0) > 0:
            L.H5Gclose(_chk(L.H5Gcreate2(self.f, b"_refs", 0, 0, 0), "H5Gcreate2(_refs)"))
        refs = (C.c_uint64 * max(len(items), 1))()
        for i, it in enumerate(items):
            self.nref += 1
            path = f"_refs/{self.nref:08d}"
            self.put(self.f, path, it)
            _chk(L.H5Rcreate(C.byref(refs, 8 * i), self.f, path.encode(), _H5R_OBJECT, -1), "H5Rcreate")
        self._dataset(loc, name, L.g_H5T_STD_REF_OBJ, (len(items),), L.g_H5T_STD_REF_OBJ, refs)

    def put(self, loc, name, v):
        L = self.L
        if isinstance(v, dict):
            g = _chk(L.H5Gcreate2(loc, name.encode(), 0, 0, 0), f"H5Gcreate2({name})")
            for k, x in v.items():
                self.put(g, str(k), x)
            L.H5Gclose(g)
        elif isinstance(v, str):
            p = C.c_char_p(v.encode())
            self._dataset(loc, name, self.vstr, None, self.vstr, C.byref(p))
        elif isinstance(v, (list, tuple)) and (len(v) == 0 or any(np.ndim(x) > 0 for x in v)):
            # a Julia Vector of arrays: one reference per element (an empty list too: Vector{Vector{Float64}}())
            if len(v) == 0:
                self._array(loc, name, np.zeros(0))
            else:
                self._refs(loc, name, [np.asarray(x) for x in v])
        elif v is None:
            raise TypeError(f"{name}: None (Julia `missing`/`nothing`) has no plain HDF5 encoding; leave the key out")
        else:
            self._array(loc, name, v)


def save(filename, data):
    """JLD2.save(filename, dict): every key a top-level entry (nested dicts become groups)."""
    w = _Writer(filename)
    try:
        for k, v in data.items():
            w.put(w.f, str(k), v)
    finally:
        w.close()


class _Reader:
    def __init__(self, filename):
        L = self.L = _load()
        self.f = _chk(L.H5Fopen(os.fspath(filename).encode(), _H5F_ACC_RDONLY, 0), f"H5Fopen({filename})")

    def close(self):
        self.L.H5Fclose(self.f)

    def _read_dataset(self, d):
        L = self.L
        t, s = L.H5Dget_type(d), L.H5Dget_space(d)
        try:
            nd = L.H5Sget_simple_extent_ndims(s)
            dims = (C.c_uint64 * max(nd, 1))()
            if nd > 0:
                L.H5Sget_simple_extent_dims(s, dims, None)
            shape = tuple(int(x) for x in dims[:nd])
            cls, size = L.H5Tget_class(t), L.H5Tget_size(t)
            count = int(np.prod(shape)) if nd else 1
            if cls == _H5T_FLOAT:
                out = np.zeros(shape, dtype=np.float64)
                if count:
                    _chk(L.H5Dread(d, L.g_H5T_NATIVE_DOUBLE, 0, 0, 0, out.ctypes.data_as(C.c_void_p)), "H5Dread")
            elif cls == _H5T_INTEGER:
                out = np.zeros(shape, dtype=np.int64)
                if count:
                    _chk(L.H5Dread(d, L.g_H5T_NATIVE_INT64, 0, 0, 0, out.ctypes.data_as(C.c_void_p)), "H5Dread")
            elif cls == _H5T_COMPOUND and size == 16 and L.H5Tget_nmembers(t) == 2:
                out = np.zeros(shape, dtype=np.complex128)
                if count:
                    _chk(L.H5Dread(d, t, 0, 0, 0, out.ctypes.data_as(C.c_void_p)), "H5Dread")
            elif cls == _H5T_STRING:
                if L.H5Tis_variable_str(t) > 0:
                    ptrs = (C.c_char_p * count)()
                    _chk(L.H5Dread(d, t, 0, 0, 0, ptrs), "H5Dread")
                    vals = [(p or b"").decode() for p in ptrs]
                    L.H5Dvlen_reclaim(t, s, 0, ptrs)
                else:
                    raw = C.create_string_buffer(size * count)
                    _chk(L.H5Dread(d, t, 0, 0, 0, raw), "H5Dread")
                    vals = [raw.raw[i * size:(i + 1) * size].split(b"\x00")[0].decode() for i in range(count)]
                return vals[0] if nd == 0 else vals
            elif cls == _H5T_REFERENCE:
                refs = (C.c_uint64 * max(count, 1))()
                if count:
                    _chk(L.H5Dread(d, L.g_H5T_STD_REF_OBJ, 0, 0, 0, refs), "H5Dread")
                items = []
                for i in range(count):
                    o = _chk(L.H5Rdereference2(self.f, 0, _H5R_OBJECT, C.byref(refs, 8 * i)), "H5Rdereference2")
                    items.append(self._read_object(o))
                    L.H5Oclose(o)
                return items
            else:
                return None              # a committed Julia struct or another type this reader has no numpy form for
            if nd == 0:
                return out[()].item()
            return out.reshape(shape).T   # back to the Julia shape (a view: Fortran-ordered)
        finally:
            L.H5Tclose(t); L.H5Sclose(s)

    def _read_object(self, o):
        L = self.L
        kind = L.H5Iget_type(o)
        if kind == _H5I_DATASET:
            return self._read_dataset(o)
        if kind == _H5I_GROUP:
            info = _GInfo()                       # (H5Gget_info: the 1.8 API; H5Gget_num_objs is absent from builds without the 1.6 API)
            L.H5Gget_info(o, C.byref(info))
            out = {}
            for i in range(info.nlinks):
                ln = L.H5Lget_name_by_idx(o, b".", _H5_INDEX_NAME, _H5_ITER_INC, i, None, 0, 0)
                buf = C.create_string_buffer(ln + 1)
                L.H5Lget_name_by_idx(o, b".", _H5_INDEX_NAME, _H5_ITER_INC, i, buf, ln + 1, 0)
                name = buf.value.decode()
                if name in ("_refs", "_types"):          # storage of referenced objects / JLD2's committed datatypes
                    continue
                child = L.H5Oopen(o, buf.value, 0)
                if child < 0:
                    continue
                try:
                    out[name] = self._read_object(child)
                finally:
                    L.H5Oclose(child)
            return out
        return None


def load(filename):
    """JLD2.load(filename) for the encodings listed in the module docstring: a dict (groups are nested dicts); entries
    of types without a numpy form (committed Julia structs) come back as None."""
    r = _Reader(filename)
    try:
        root = r.L.H5Oopen(r.f, b"/", 0)
        try:
            return r._read_object(root)
        finally:
            r.L.H5Oclose(root)
    finally:
        r.close()


def is_jld2_name(filename):
    return str(filename).endswith((".jld2", ".h5", ".hdf5"))
