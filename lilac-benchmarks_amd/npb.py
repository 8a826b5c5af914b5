"""ctypes binding of lib/libnpbcg.so (host/npb_cg.h): the NPB CG matrix generator and
benchmark loop, our counterpart of the reference's Fortran caller (NPB3.3.1/CG/cg.f)."""
import ctypes

import numpy as np

from .build import lib_path

NPB_MAX_ITER = 128


class Class(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char), ("na", ctypes.c_int), ("nonzer", ctypes.c_int), ("niter", ctypes.c_int),
                ("shift", ctypes.c_double), ("rcond", ctypes.c_double), ("zeta_verify", ctypes.c_double)]


class CMatrix(ctypes.Structure):
    _fields_ = [("cls", Class), ("nnz", ctypes.c_int64), ("rowstr", ctypes.POINTER(ctypes.c_int)),
                ("colidx", ctypes.POINTER(ctypes.c_int)), ("a", ctypes.POINTER(ctypes.c_double)),
                ("t_gen", ctypes.c_double), ("row_begin", ctypes.c_int), ("row_end", ctypes.c_int)]


class Result(ctypes.Structure):
    _fields_ = [("cls", ctypes.c_char), ("na", ctypes.c_int), ("niter", ctypes.c_int), ("nnz", ctypes.c_int64),
                ("spmv_calls", ctypes.c_int64), ("zeta", ctypes.c_double), ("rnorm", ctypes.c_double),
                ("zeta_hist", ctypes.c_double * NPB_MAX_ITER), ("rnorm_hist", ctypes.c_double * NPB_MAX_ITER),
                ("err", ctypes.c_double), ("verified", ctypes.c_int), ("t_init", ctypes.c_double),
                ("t_bench", ctypes.c_double), ("t_spmv", ctypes.c_double), ("mops", ctypes.c_double)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if not k.endswith("_hist")}
        d["cls"] = self.cls.decode()
        d["zeta_hist"] = list(self.zeta_hist[: self.niter])
        d["rnorm_hist"] = list(self.rnorm_hist[: self.niter])
        return d


class DeviceFns(ctypes.Structure):
    """struct npb_cg_device_fns (host/npb_cg.h): addresses of the backend's opt-in CG entry points."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("matrix_create", "matrix_destroy", "cg_create", "cg_destroy",
                                               "cg_fill_x", "cg_conj_grad", "cg_norms", "cg_x_from_z",
                                               "partition_rows", "cg_create_multi")]

    @classmethod
    def from_backend(cls, be):
        names = {"matrix_create": "spmv_hip_matrix_create", "matrix_destroy": "spmv_hip_matrix_destroy",
                 "cg_create": "spmv_hip_cg_create", "cg_destroy": "spmv_hip_cg_destroy",
                 "cg_fill_x": "spmv_hip_cg_fill_x", "cg_conj_grad": "spmv_hip_cg_conj_grad",
                 "cg_norms": "spmv_hip_cg_norms", "cg_x_from_z": "spmv_hip_cg_x_from_z",
                 "partition_rows": "spmv_hip_partition_rows", "cg_create_multi": "spmv_hip_cg_create_multi"}
        f = cls()
        for field, sym in names.items():
            setattr(f, field, ctypes.cast(getattr(be.lib, sym), ctypes.c_void_p).value)
        return f


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = L = ctypes.CDLL(lib_path("libnpbcg.so"))
        L.npb_makea.argtypes = [ctypes.c_char, ctypes.POINTER(CMatrix)]
        L.npb_makea_part.argtypes = [ctypes.c_char, ctypes.c_int, ctypes.c_int, ctypes.POINTER(CMatrix)]
        L.npb_makea_rows.argtypes = [ctypes.c_char, ctypes.c_int, ctypes.c_int, ctypes.POINTER(CMatrix)]
        L.npb_matrix_free.argtypes = [ctypes.POINTER(CMatrix)]
        L.npb_matrix_free.restype = None
        L.npb_cg_run.argtypes = [ctypes.POINTER(CMatrix), ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                 ctypes.POINTER(Result)]
        L.npb_class_lookup.argtypes = [ctypes.c_char, ctypes.POINTER(Class)]
        L.npb_cg_run_device.argtypes = [ctypes.POINTER(CMatrix), ctypes.POINTER(DeviceFns), ctypes.c_int,
                                        ctypes.c_int, ctypes.POINTER(Result)]
        L.npb_cg_run_device_parts.argtypes = [ctypes.POINTER(CMatrix), ctypes.POINTER(DeviceFns), ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, ctypes.POINTER(Result)]
    return _lib


class NpbMatrix:
    """A generated class matrix; arrays are numpy views of the C allocation (1-based CSR)."""

    def __init__(self, cls, part=None, nparts=None, rows=None):
        """Whole class matrix, or (part, nparts): one block of the multi-GPU row partition, or
        rows=(begin, end): an explicit 0-based row range.  Blocks have a local rowstr and global colidx."""
        self.c = CMatrix()
        if nparts:
            rc = lib().npb_makea_part(cls.encode()[:1], int(part), int(nparts), ctypes.byref(self.c))
        elif rows is not None:
            rc = lib().npb_makea_rows(cls.encode()[:1], int(rows[0]), int(rows[1]), ctypes.byref(self.c))
        else:
            rc = lib().npb_makea(cls.encode()[:1], ctypes.byref(self.c))
        if rc != 0:
            raise RuntimeError("npb_makea(%s) failed with %d" % (cls, rc))
        self.cls = cls
        self.na = self.c.cls.na
        self.row_begin, self.row_end = self.c.row_begin, self.c.row_end
        self.rows = self.row_end - self.row_begin
        self.nnz = self.c.nnz
        self.niter = self.c.cls.niter
        self.nonzer = self.c.cls.nonzer
        self.zeta_verify = self.c.cls.zeta_verify
        self.t_gen = self.c.t_gen
        self.rowstr = np.ctypeslib.as_array(self.c.rowstr, (self.rows + 1,))
        self.colidx = np.ctypeslib.as_array(self.c.colidx, (self.nnz,))
        self.a = np.ctypeslib.as_array(self.c.a, (self.nnz,))

    def run_cg(self, harness_fn, niter=0, verbose=False):
        """harness_fn: address of a spmv_harness_-compatible function (ctypes.c_void_p or int)."""
        r = Result()
        rc = lib().npb_cg_run(ctypes.byref(self.c), harness_fn, niter, 1 if verbose else 0, ctypes.byref(r))
        if rc != 0:
            raise RuntimeError("npb_cg_run failed with %d" % rc)
        return r.as_dict()

    def run_cg_device(self, backend, niter=0, verbose=False):
        """The benchmark with conj_grad on the GPU (opt-in spmv_hip_cg_* entry points)."""
        fns = DeviceFns.from_backend(backend)
        r = Result()
        rc = lib().npb_cg_run_device(ctypes.byref(self.c), ctypes.byref(fns), niter, 1 if verbose else 0,
                                     ctypes.byref(r))
        if rc != 0:
            raise RuntimeError("npb_cg_run_device failed with %d: %s" % (rc, backend.last_error()))
        return r.as_dict()

    def run_cg_device_parts(self, backend, nparts, ndevices=None, niter=0, verbose=False):
        """The benchmark with conj_grad on the GPUs, the matrix in `nparts` row blocks (block k on device k % ndevices):
        spmv_hip_cg_create_multi."""
        fns = DeviceFns.from_backend(backend)
        r = Result()
        nd = ndevices or max(1, min(nparts, backend.device_count()))
        rc = lib().npb_cg_run_device_parts(ctypes.byref(self.c), ctypes.byref(fns), nparts, nd, niter, 1 if verbose else 0,
                                           ctypes.byref(r))
        if rc != 0:
            raise RuntimeError("npb_cg_run_device_parts failed with %d: %s" % (rc, backend.last_error()))
        d = r.as_dict()
        d["row_partitions"], d["devices_used"] = nparts, nd
        return d

    def total_mop(self, niter=None):
        niter = niter or self.niter
        nz = self.nonzer * (self.nonzer + 1)
        return 2.0 * niter * self.na * (3.0 + nz + 25.0 * (5.0 + nz) + 3.0) / 1e6

    def close(self):
        if self.c.rowstr:
            self.rowstr = self.colidx = self.a = None
            lib().npb_matrix_free(ctypes.byref(self.c))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
