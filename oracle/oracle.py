"""ctypes loader for the parity checker: liboracle-spmv.so (our CPU restatement) and, when it
has been built in the authoring container, _ref/native.so (the reference's own native.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py; never by the product."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle-spmv.so")
REF_SO = os.path.join(HERE, "_ref", "native.so")


def build():
    proc = subprocess.run(["make", "-C", HERE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + proc.stdout)
    return proc.stdout


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class _Harness:
    def __init__(self, path, dname, fname):
        self.path = path
        self.lib = ctypes.CDLL(path)
        self._d = getattr(self.lib, dname)
        self._f = getattr(self.lib, fname)
        for fn in (self._d, self._f):
            fn.restype = ctypes.c_void_p
            fn.argtypes = [ctypes.c_void_p] * 6

    def spmv(self, a, x, rowstr, colidx, rows, out=None):
        """y = A*x with the sequential reference loop; dtype of `a` selects double / float."""
        fn = self._d if a.dtype == np.float64 else self._f
        assert x.dtype == a.dtype and rowstr.dtype == np.int32 and colidx.dtype == np.int32
        y = np.empty(rows, a.dtype) if out is None else out
        n = ctypes.c_int(int(rows))
        fn(_ptr(y), _ptr(a), _ptr(x), _ptr(rowstr), _ptr(colidx), ctypes.addressof(n))
        return y

    def harness_fn(self, dtype=np.float64):
        fn = self._d if np.dtype(dtype) == np.float64 else self._f
        return ctypes.cast(fn, ctypes.c_void_p)


class Oracle(_Harness):
    """Our restatement (oracle/spmv_oracle.c)."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        super().__init__(ORACLE_SO, "spmv_harness_", "f_spmv_harness_")
        L = self.lib
        L.oracle_spmv_mt.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int]
        L.oracle_spmv_mt.restype = ctypes.c_int
        L.oracle_spmv_absum.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int]
        L.oracle_spmv_absum.restype = None
        L.oracle_f_spmv_absum.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int]
        L.oracle_f_spmv_absum.restype = None
        L.oracle_fnv1a64.restype = ctypes.c_uint64
        L.oracle_fnv1a64.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.npb_oracle_makea.restype = ctypes.c_long
        L.npb_oracle_makea.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_long,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]

    def absum(self, a, x, rowstr, colidx, rows):
        """sum_j |a_j x_j| per row in float64: the scale of the reordering error bound."""
        out = np.empty(rows, np.float64)
        fn = self.lib.oracle_spmv_absum if a.dtype == np.float64 else self.lib.oracle_f_spmv_absum
        fn(_ptr(out), _ptr(a), _ptr(x), _ptr(rowstr), _ptr(colidx), int(rows))
        return out

    def fnv1a64(self, arr):
        arr = np.ascontiguousarray(arr)
        return int(self.lib.oracle_fnv1a64(_ptr(arr), arr.nbytes))

    def npb_makea(self, na, nonzer, shift, rcond=0.1):
        nz = na * (nonzer + 1) ** 2
        a = np.zeros(nz)
        col = np.zeros(nz, np.int32)
        rs = np.zeros(na + 1, np.int32)
        nnz = self.lib.npb_oracle_makea(na, nonzer, rcond, shift, nz, _ptr(a), _ptr(col), _ptr(rs))
        if nnz < 0:
            raise RuntimeError("npb_oracle_makea failed: %d" % nnz)
        return rs, col[:nnz].copy(), a[:nnz].copy()


    def spmv_mt(self, a, x, rowstr, colidx, rows, nthreads, out=None):
        """The same loop on `nthreads` host threads, rows split by nonzeros: bit-identical to spmv() (each row is
        still summed left to right by one thread).  FP64 only; context for bench.py."""
        assert a.dtype == np.float64 and x.dtype == np.float64
        y = np.empty(rows, np.float64) if out is None else out
        used = self.lib.oracle_spmv_mt(_ptr(y), _ptr(a), _ptr(x), _ptr(rowstr), _ptr(colidx), int(rows), int(nthreads))
        return y, used


class Reference(_Harness):
    """The reference's own native.c/native-impl.c, compiled by oracle/Makefile into _ref/."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        super().__init__(REF_SO, "spmv_harness_", "f_spmv_harness_")


def have_reference():
    return os.path.exists(REF_SO)
