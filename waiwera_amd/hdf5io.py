"""Waiwera's HDF5 output / restart layout through the HDF5 C library (ctypes; there is no h5py in
the image): what `flow_simulation_output` writes (src/flow_simulation.F90:2580-2994, hdf5io.F90)
and what `setup_initial` reads back (src/initial.F90:421-677).

Layout: /time (nt, 1); /cell_index (ncells, 1) int32, natural cell index -> row of the cell
datasets; /cell_fields/<field> (nt, ncells) for fluid and tracer fields, (ncells[, dim]) for the
geometry fields.  Datasets written here have fixed dimensions (the run is over when the file is
written); files with unlimited time dimensions, as the reference writes them, are read the same."""
import ctypes as C
import os

import numpy as np

_LIB = None
hid = C.c_int64
hsize = C.c_uint64


class Hdf5Unavailable(RuntimeError):
    pass


def _lib():
    global _LIB
    if _LIB is None:
        last = None
        for name in (os.environ.get("WAIWERA_HDF5_LIB"), "libhdf5.so", "/opt/conda/lib/libhdf5.so", "libhdf5_serial.so"):
            if not name:
                continue
            try:
                _LIB = C.CDLL(name)
                break
            except OSError as e:
                last = e
        if _LIB is None:
            raise Hdf5Unavailable("HDF5 C library not found (%s); set WAIWERA_HDF5_LIB" % last)
        L = _LIB
        L.H5open()
        for fn, res, args in (
                ("H5Fcreate", hid, [C.c_char_p, C.c_uint, hid, hid]), ("H5Fopen", hid, [C.c_char_p, C.c_uint, hid]),
                ("H5Fclose", C.c_int, [hid]), ("H5Gcreate2", hid, [hid, C.c_char_p, hid, hid, hid]),
                ("H5Gclose", C.c_int, [hid]), ("H5Screate_simple", hid, [C.c_int, C.POINTER(hsize), C.POINTER(hsize)]),
                ("H5Sclose", C.c_int, [hid]), ("H5Dcreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]),
                ("H5Dopen2", hid, [hid, C.c_char_p, hid]), ("H5Dclose", C.c_int, [hid]),
                ("H5Dwrite", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
                ("H5Dread", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]), ("H5Dget_space", hid, [hid]),
                ("H5Dget_type", hid, [hid]), ("H5Tget_class", C.c_int, [hid]), ("H5Tclose", C.c_int, [hid]),
                ("H5Sget_simple_extent_ndims", C.c_int, [hid]),
                ("H5Sget_simple_extent_dims", C.c_int, [hid, C.POINTER(hsize), C.POINTER(hsize)]),
                ("H5Lexists", C.c_int, [hid, C.c_char_p, hid])):
            f = getattr(L, fn)
            f.restype, f.argtypes = res, args
        L.NATIVE_DOUBLE = hid.in_dll(L, "H5T_NATIVE_DOUBLE_g").value
        L.NATIVE_INT = hid.in_dll(L, "H5T_NATIVE_INT_g").value
    return _LIB


def read_dataset(path, name):
    """a dataset of an HDF5 file as a numpy array (float64, or int32 for integer datasets)"""
    L = _lib()
    f = L.H5Fopen(path.encode(), 0, 0)
    if f < 0:
        raise IOError("cannot open " + path)
    try:
        d = L.H5Dopen2(f, name.encode(), 0)
        if d < 0:
            raise KeyError(name)
        sp = L.H5Dget_space(d)
        nd = L.H5Sget_simple_extent_ndims(sp)
        dims = (hsize * max(nd, 1))()
        L.H5Sget_simple_extent_dims(sp, dims, None)
        shape = tuple(int(v) for v in dims[:nd])
        t = L.H5Dget_type(d)
        is_int = L.H5Tget_class(t) == 0   # H5T_INTEGER
        out = np.zeros(shape, dtype=np.int32 if is_int else np.float64)
        rc = L.H5Dread(d, L.NATIVE_INT if is_int else L.NATIVE_DOUBLE, 0, 0, 0, out.ctypes.data)
        L.H5Tclose(t); L.H5Sclose(sp); L.H5Dclose(d)
        if rc < 0:
            raise IOError("cannot read " + name)
        return out
    finally:
        L.H5Fclose(f)


def has_dataset(path, name):
    L = _lib()
    f = L.H5Fopen(path.encode(), 0, 0)
    if f < 0:
        raise IOError("cannot open " + path)
    try:
        parts = name.strip("/").split("/")
        cur = ""
        for p in parts:
            cur += "/" + p
            if L.H5Lexists(f, cur.encode(), 0) <= 0:
                return False
        return True
    finally:
        L.H5Fclose(f)


def write_file(path, datasets):
    """datasets: {"/group/name": array}; groups are created as needed (one level, like the reference)"""
    L = _lib()
    f = L.H5Fcreate(path.encode(), 2, 0, 0)   # H5F_ACC_TRUNC
    if f < 0:
        raise IOError("cannot create " + path)
    groups = {}
    try:
        for name, arr in datasets.items():
            parts = name.strip("/").split("/")
            loc = f
            if len(parts) == 2:
                if parts[0] not in groups:
                    groups[parts[0]] = L.H5Gcreate2(f, parts[0].encode(), 0, 0, 0)
                loc = groups[parts[0]]
            a = np.ascontiguousarray(arr)
            is_int = a.dtype.kind in "iu"
            a = a.astype(np.int32 if is_int else np.float64)
            dims = (hsize * max(a.ndim, 1))(*a.shape)
            sp = L.H5Screate_simple(a.ndim, dims, None)
            ty = L.NATIVE_INT if is_int else L.NATIVE_DOUBLE
            d = L.H5Dcreate2(loc, parts[-1].encode(), ty, sp, 0, 0, 0)
            if d < 0 or L.H5Dwrite(d, ty, 0, 0, 0, a.ctypes.data) < 0:
                raise IOError("cannot write " + name)
            L.H5Dclose(d); L.H5Sclose(sp)
    finally:
        for g in groups.values():
            L.H5Gclose(g)
        L.H5Fclose(f)


def read_state(path, index=-1):
    """cell fields of a Waiwera output file at time index `index` (negative: from the end,
    src/initial.F90:483-489), in natural cell order: {field: (ncells,)} plus "time"."""
    t = read_dataset(path, "/time").ravel()
    k = index if index >= 0 else t.size + index
    ci = read_dataset(path, "/cell_index").ravel()
    out = {"time": float(t[k])}
    for name in ("fluid_pressure", "fluid_temperature", "fluid_vapour_saturation", "fluid_region",
                 "fluid_CO2_partial_pressure", "fluid_air_partial_pressure", "fluid_liquid_salt_mass_fraction", "fluid_solid_saturation"):
        if has_dataset(path, "/cell_fields/" + name):
            out[name] = read_dataset(path, "/cell_fields/" + name)[k][ci]
    return out
