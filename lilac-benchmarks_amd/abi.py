"""ctypes binding of libhip-spmv.so: the two ABI symbols of include/spmv_harness.h and the
extended API of include/spmv_hip.h.  Mirrors the C declarations one to one; no arithmetic here."""
import ctypes
import os

import numpy as np

from .build import lib_path

PLAN_AUTO, PLAN_ORDERED, PLAN_STREAM, PLAN_VECTOR, PLAN_BINNED, PLAN_TILED, PLAN_STRIPED, PLAN_BLOCKED, PLAN_HUB = 0, 1, 2, 3, 4, 5, 6, 7, 8
PLAN_NAMES = {0: "auto", 1: "ordered", 2: "stream", 3: "vector", 4: "binned", 5: "tiled", 6: "striped", 7: "blocked", 8: "hub"}

c_int_p = ctypes.POINTER(ctypes.c_int)


class Info(ctypes.Structure):
    """struct spmv_hip_info (include/spmv_hip.h)"""
    _fields_ = [
        ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("nnz", ctypes.c_int64),
        ("elem_size", ctypes.c_int32), ("device", ctypes.c_int32), ("plan", ctypes.c_int32),
        ("plan_param", ctypes.c_int32), ("min_row_len", ctypes.c_int32), ("max_row_len", ctypes.c_int32),
        ("mean_row_len", ctypes.c_double), ("n_blocks", ctypes.c_int32), ("n_long_rows", ctypes.c_int32),
        ("n_launches", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("algorithmic_bytes", ctypes.c_int64), ("device_bytes", ctypes.c_int64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class Stats(ctypes.Structure):
    """struct spmv_hip_stats (include/spmv_hip.h)"""
    _fields_ = [
        ("calls", ctypes.c_int64), ("uploads", ctypes.c_int64), ("invalidations", ctypes.c_int64),
        ("t_init", ctypes.c_double), ("t_upload", ctypes.c_double), ("t_x_h2d", ctypes.c_double),
        ("t_kernel", ctypes.c_double), ("t_y_d2h", ctypes.c_double), ("t_total", ctypes.c_double),
        ("host_calls", ctypes.c_int64), ("t_host", ctypes.c_double), ("value_refreshes", ctypes.c_int64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def _ptr(arr):
    return arr.ctypes.data_as(ctypes.c_void_p)


class Backend:
    """One loaded libhip-spmv.so.  Raises if the library is missing: there is no fallback."""

    def __init__(self, path=None):
        path = path or os.environ.get("SPMV_HIP_LIB") or lib_path()
        if not os.path.exists(path):
            raise FileNotFoundError(
                "%s not found: build it first (python -c 'import __graft_entry__ as g; g.build()')" % path)
        self.path = path
        self.lib = L = ctypes.CDLL(path)
        for name in ("spmv_harness_", "f_spmv_harness_"):
            fn = getattr(L, name)
            fn.restype = ctypes.c_void_p
            fn.argtypes = [ctypes.c_void_p] * 6
        L.spmv_hip_device_count.restype = ctypes.c_int
        L.spmv_hip_device_free_bytes.restype = ctypes.c_int64
        L.spmv_hip_device_free_bytes.argtypes = [ctypes.c_int]
        L.spmv_hip_version.restype = ctypes.c_char_p
        L.spmv_hip_last_error.restype = ctypes.c_char_p
        L.spmv_hip_matrix_create.restype = ctypes.c_void_p
        L.spmv_hip_matrix_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.spmv_hip_matrix_destroy.argtypes = [ctypes.c_void_p]
        L.spmv_hip_matrix_destroy.restype = None
        L.spmv_hip_matrix_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(Info)]
        L.spmv_hip_matrix_set_plan.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.spmv_hip_matrix_refresh_values.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.spmv_hip_spmv_device.argtypes = [ctypes.c_void_p] * 4
        L.spmv_hip_time_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        L.spmv_hip_time_multi.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(ctypes.c_float)]
        L.spmv_hip_time_call.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        L.spmv_hip_host_alloc.restype = ctypes.c_void_p
        L.spmv_hip_host_alloc.argtypes = [ctypes.c_size_t]
        L.spmv_hip_host_free.argtypes = [ctypes.c_void_p]
        L.spmv_hip_host_free.restype = None
        L.spmv_hip_host_register.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.spmv_hip_host_unregister.argtypes = [ctypes.c_void_p]
        L.spmv_hip_host_unregister.restype = None
        L.spmv_hip_partition_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.spmv_hip_partition_rows.restype = None
        L.spmv_hip_malloc.restype = ctypes.c_void_p
        L.spmv_hip_malloc.argtypes = [ctypes.c_int, ctypes.c_size_t]
        L.spmv_hip_free.argtypes = [ctypes.c_int, ctypes.c_void_p]
        L.spmv_hip_free.restype = None
        L.spmv_hip_memcpy_h2d.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.spmv_hip_memcpy_d2h.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.spmv_hip_memset.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        L.spmv_hip_synchronize.argtypes = [ctypes.c_int]
        L.spmv_hip_set_tuning.argtypes = [ctypes.c_int, ctypes.c_int]
        L.spmv_hip_set_tuning.restype = None
        L.spmv_hip_get_stats.argtypes = [ctypes.POINTER(Stats)]
        L.spmv_hip_get_stats.restype = None
        L.spmv_hip_reset_stats.restype = None
        L.spmv_hip_release.restype = None
        L.spmv_hip_abi_matrix_info.argtypes = [ctypes.POINTER(Info)]
        L.spmv_hip_cg_create.restype = ctypes.c_void_p
        L.spmv_hip_cg_create.argtypes = [ctypes.c_void_p]
        L.spmv_hip_cg_destroy.argtypes = [ctypes.c_void_p]
        L.spmv_hip_cg_destroy.restype = None
        L.spmv_hip_cg_set_x.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.spmv_hip_cg_fill_x.argtypes = [ctypes.c_void_p, ctypes.c_double]
        L.spmv_hip_cg_get.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.spmv_hip_cg_conj_grad.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        L.spmv_hip_cg_norms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_double)]
        L.spmv_hip_cg_x_from_z.argtypes = [ctypes.c_void_p, ctypes.c_double]

    # ---- the drop-in ABI ---------------------------------------------------------------
    def harness(self, ov, a, iv, rowstr, colidx, rows):
        """spmv_harness_ / f_spmv_harness_ with numpy arrays, by reference, exactly like a caller."""
        if a.dtype == np.float64:
            fn = self.lib.spmv_harness_
        elif a.dtype == np.float32:
            fn = self.lib.f_spmv_harness_
        else:
            raise TypeError("a must be float64 or float32")
        assert ov.dtype == a.dtype and iv.dtype == a.dtype
        assert rowstr.dtype == np.int32 and colidx.dtype == np.int32
        n = ctypes.c_int(int(rows))
        fn(_ptr(ov), _ptr(a), _ptr(iv), _ptr(rowstr), _ptr(colidx), ctypes.addressof(n))
        return ov

    def harness_fn(self, dtype=np.float64):
        """Raw function address, for C drivers that take a spmv_harness_fn."""
        fn = self.lib.spmv_harness_ if np.dtype(dtype) == np.float64 else self.lib.f_spmv_harness_
        return ctypes.cast(fn, ctypes.c_void_p)

    # ---- extended API -------------------------------------------------------------------
    def device_count(self):
        return int(self.lib.spmv_hip_device_count())

    def device_free_bytes(self, device=0):
        return int(self.lib.spmv_hip_device_free_bytes(device))

    def version(self):
        return self.lib.spmv_hip_version().decode()

    def last_error(self):
        return self.lib.spmv_hip_last_error().decode()

    def partition_rows(self, rowstr, rows, parts):
        bounds = np.zeros(parts + 1, np.int32)
        self.lib.spmv_hip_partition_rows(_ptr(rowstr), int(rows), int(parts), _ptr(bounds))
        return bounds

    def stats(self):
        s = Stats()
        self.lib.spmv_hip_get_stats(ctypes.byref(s))
        return s.as_dict()

    def reset_stats(self):
        self.lib.spmv_hip_reset_stats()

    def release(self):
        self.lib.spmv_hip_release()

    def abi_matrix_info(self):
        inf = Info()
        if self.lib.spmv_hip_abi_matrix_info(ctypes.byref(inf)) != 0:
            return None
        return inf.as_dict()

    def set_tuning(self, key, value):
        self.lib.spmv_hip_set_tuning(int(key), int(value))

    # device memory
    def malloc(self, nbytes, device=0):
        p = self.lib.spmv_hip_malloc(device, nbytes)
        if not p:
            raise MemoryError(self.last_error())
        return p

    def free(self, p, device=0):
        self.lib.spmv_hip_free(device, p)

    def h2d(self, dst, arr, device=0):
        arr = np.ascontiguousarray(arr)
        if self.lib.spmv_hip_memcpy_h2d(device, dst, _ptr(arr), arr.nbytes) != 0:
            raise RuntimeError(self.last_error())

    def d2h(self, arr, src, device=0):
        assert arr.flags["C_CONTIGUOUS"]
        if self.lib.spmv_hip_memcpy_d2h(device, _ptr(arr), src, arr.nbytes) != 0:
            raise RuntimeError(self.last_error())
        return arr

    def synchronize(self, device=0):
        if self.lib.spmv_hip_synchronize(device) != 0:
            raise RuntimeError(self.last_error())

    def host_register(self, arr):
        """Pins a host array in place (hipHostRegister) so that copies from / to it are DMA transfers; False when the
        runtime declines (the array then still works as pageable memory)."""
        return self.lib.spmv_hip_host_register(_ptr(arr), arr.nbytes) == 0

    def host_unregister(self, arr):
        self.lib.spmv_hip_host_unregister(_ptr(arr))

    def time_multi(self, matrices, x_devs, y_devs, reps):
        """spmv_hip_time_multi: reps products of each device-resident matrix, rep by rep; ms elapsed on each one's stream."""
        n = len(matrices)
        hs = (ctypes.c_void_p * n)(*[m.h for m in matrices])
        xs = (ctypes.c_void_p * n)(*x_devs)
        ys = (ctypes.c_void_p * n)(*y_devs)
        ms = (ctypes.c_float * n)()
        if self.lib.spmv_hip_time_multi(hs, xs, ys, n, int(reps), ms) != 0:
            raise RuntimeError(self.last_error())
        return [float(v) for v in ms]

    def matrix(self, a, rowstr, colidx, rows, device=0, row_begin=0, row_end=None, cols_hint=0):
        return Matrix(self, a, rowstr, colidx, rows, device, row_begin, row_end, cols_hint)


class Matrix:
    """spmv_hip_matrix: a device-resident CSR row block with its execution plan."""

    def __init__(self, be, a, rowstr, colidx, rows, device=0, row_begin=0, row_end=None, cols_hint=0):
        assert a.dtype in (np.float64, np.float32)
        assert rowstr.dtype == np.int32 and colidx.dtype == np.int32
        self.be = be
        self.device = device
        self.dtype = a.dtype
        row_end = rows if row_end is None else row_end
        self.h = be.lib.spmv_hip_matrix_create(_ptr(a), _ptr(rowstr), _ptr(colidx), int(rows), a.dtype.itemsize,
                                               device, row_begin, row_end, cols_hint)
        if not self.h:
            raise RuntimeError("spmv_hip_matrix_create: " + be.last_error())
        self._bufs = []

    def info(self):
        inf = Info()
        self.be.lib.spmv_hip_matrix_info(self.h, ctypes.byref(inf))
        return inf.as_dict()

    def set_plan(self, plan, param=0):
        if self.be.lib.spmv_hip_matrix_set_plan(self.h, plan, param) != 0:
            raise RuntimeError(self.be.last_error())

    def refresh_values(self, a, colidx=None):
        """New values for the same pattern (spmv_hip_matrix_refresh_values); `a` as given at creation."""
        assert a.dtype == self.dtype
        if self.be.lib.spmv_hip_matrix_refresh_values(self.h, _ptr(a), _ptr(colidx) if colidx is not None else None) != 0:
            raise RuntimeError(self.be.last_error())

    def spmv_device(self, x_dev, y_dev, stream=None):
        if self.be.lib.spmv_hip_spmv_device(self.h, x_dev, y_dev, stream) != 0:
            raise RuntimeError(self.be.last_error())

    def time_device(self, x_dev, y_dev, reps, stream=None):
        ms = ctypes.c_float(0)
        if self.be.lib.spmv_hip_time_device(self.h, x_dev, y_dev, reps, stream, ctypes.byref(ms)) != 0:
            raise RuntimeError(self.be.last_error())
        return float(ms.value)

    def time_call(self, x_host, y_host_slice, x_dev, y_dev, reps, stream=None):
        """reps x {x_host -> x_dev, y = A x, y_dev -> y_host_slice} on one stream between two HIP events (ms)."""
        ms = ctypes.c_float(0)
        if self.be.lib.spmv_hip_time_call(self.h, _ptr(x_host), _ptr(y_host_slice), x_dev, y_dev, reps, stream,
                                          ctypes.byref(ms)) != 0:
            raise RuntimeError(self.be.last_error())
        return float(ms.value)

    def spmv_host(self, x):
        """Convenience for tests: uploads x, runs the device SpMV, downloads y."""
        inf = self.info()
        be, dev = self.be, self.device
        x = np.ascontiguousarray(x, dtype=self.dtype)
        assert x.size >= inf["cols"]
        dx = be.malloc(max(x.nbytes, 16), dev)
        dy = be.malloc(max(inf["rows"] * self.dtype.itemsize, 16), dev)
        try:
            be.h2d(dx, x, dev)
            self.spmv_device(dx, dy)
            y = np.empty(inf["rows"], self.dtype)
            be.d2h(y, dy, dev)
        finally:
            be.free(dx, dev)
            be.free(dy, dev)
        return y

    def close(self):
        if self.h:
            self.be.lib.spmv_hip_matrix_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
