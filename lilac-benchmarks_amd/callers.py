"""ctypes binding of lib/libspmvcallers.so (host/callers.h): counterparts of the reference's
SparseBench / bfs / pagerank / parboil callers and of their on-disk formats."""
import ctypes

import numpy as np

from .build import lib_path

c_int_p = ctypes.POINTER(ctypes.c_int)


class CsrD(ctypes.Structure):
    _fields_ = [("rows", ctypes.c_int), ("cols", ctypes.c_int), ("nnz", ctypes.c_int64),
                ("rowstr", c_int_p), ("colidx", c_int_p), ("a", ctypes.POINTER(ctypes.c_double))]

    def arrays(self):
        return (np.ctypeslib.as_array(self.rowstr, (self.rows + 1,)),
                np.ctypeslib.as_array(self.colidx, (max(self.nnz, 1),))[: self.nnz],
                np.ctypeslib.as_array(self.a, (max(self.nnz, 1),))[: self.nnz])


class CsrF(ctypes.Structure):
    _fields_ = [("rows", ctypes.c_int), ("cols", ctypes.c_int), ("nnz", ctypes.c_int64),
                ("rowstr", c_int_p), ("colidx", c_int_p), ("a", ctypes.POINTER(ctypes.c_float))]

    def arrays(self):
        return (np.ctypeslib.as_array(self.rowstr, (self.rows + 1,)),
                np.ctypeslib.as_array(self.colidx, (max(self.nnz, 1),))[: self.nnz],
                np.ctypeslib.as_array(self.a, (max(self.nnz, 1),))[: self.nnz])


class PsNet(ctypes.Structure):
    """struct ps_net (host/callers.h): PATHSAMPLE's per-minimum neighbour lists."""
    _fields_ = [("nmin", ctypes.c_int), ("maxcol", ctypes.c_int), ("ncol", c_int_p), ("nval", c_int_p),
                ("dmat", ctypes.POINTER(ctypes.c_double))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = L = ctypes.CDLL(lib_path("libspmvcallers.so"))
        P = ctypes.POINTER
        L.sb_read_crs.argtypes = [ctypes.c_char_p, P(CsrD)]
        L.sb_write_crs.argtypes = [ctypes.c_char_p, P(CsrD), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        L.sb_generate.argtypes = [ctypes.c_int, ctypes.c_uint64, P(CsrD)]
        L.sb_bicg.argtypes = [P(CsrD), ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
                              ctypes.c_void_p, P(ctypes.c_int64), P(ctypes.c_double)]
        L.bfs_read_pairs.argtypes = [ctypes.c_char_p, P(CsrD)]
        L.bfs_run.argtypes = [P(CsrD), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.pagerank_prepare.argtypes = [P(CsrD), ctypes.c_double]
        L.pagerank_prepare.restype = None
        L.pagerank_run.argtypes = [P(CsrD), ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
        L.pagerank_run.restype = ctypes.c_double
        L.pb_coo_to_csr.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, P(CsrF)]
        L.pb_read_vec.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int]
        L.pb_compare.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.pb_run.argtypes = [P(CsrF), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.pb_run.restype = None
        L.mm_read_csr.argtypes = [ctypes.c_char_p, P(CsrD)]
        L.graph_rmat.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64, P(CsrD)]
        L.graph_grid2d.argtypes = [ctypes.c_int, ctypes.c_int, P(CsrD)]
        L.ps_generate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64, P(PsNet)]
        L.ps_free.argtypes = [P(PsNet)]
        L.ps_free.restype = None
        L.ps_committor.argtypes = [P(PsNet), ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, P(ctypes.c_double)]
        L.ps_committor.restype = ctypes.c_int64
        L.csr_d_free.argtypes = [P(CsrD)]
        L.csr_d_free.restype = None
        L.csr_f_free.argtypes = [P(CsrF)]
        L.csr_f_free.restype = None
    return _lib


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with %d" % (what, rc))


def sb_generate(size, seed=170):
    m = CsrD()
    _chk(lib().sb_generate(size, seed, ctypes.byref(m)), "sb_generate")
    return m


def sb_read(path):
    m = CsrD()
    _chk(lib().sb_read_crs(path.encode(), ctypes.byref(m)), "sb_read_crs")
    return m


def sb_write(path, m):
    _chk(lib().sb_write_crs(path.encode(), ctypes.byref(m), None, None, 0), "sb_write_crs")


def sb_bicg(m, harness_fn, maxit=100, rtol=1e-6):
    hist = np.zeros(maxit + 1)
    x = np.zeros(m.rows + 2)
    calls = ctypes.c_int64(0)
    tm = ctypes.c_double(0)
    its = lib().sb_bicg(ctypes.byref(m), harness_fn, maxit, rtol, hist.ctypes.data, x.ctypes.data,
                        ctypes.byref(calls), ctypes.byref(tm))
    return dict(its=its, hist=hist[: abs(its)], x=x[: m.rows], calls=calls.value, t_mult=tm.value)


def bfs_read(path):
    m = CsrD()
    _chk(lib().bfs_read_pairs(path.encode(), ctypes.byref(m)), "bfs_read_pairs")
    return m


def bfs_run(m, harness_fn, source):
    dist = np.zeros(m.rows + 2)
    levels = lib().bfs_run(ctypes.byref(m), harness_fn, source, dist.ctypes.data)
    return levels, dist[: m.rows]


def graph_rmat(scale, edge_factor=8, seed=20):
    m = CsrD()
    _chk(lib().graph_rmat(scale, edge_factor, seed, ctypes.byref(m)), "graph_rmat")
    return m


def graph_grid2d(nx, ny):
    m = CsrD()
    _chk(lib().graph_grid2d(nx, ny, ctypes.byref(m)), "graph_grid2d")
    return m


def pagerank(m, harness_fn, iters, d=0.85, prepare=True):
    if prepare:
        lib().pagerank_prepare(ctypes.byref(m), d)
    x = np.full(m.rows + 2, 1.0 / m.rows)
    err = lib().pagerank_run(ctypes.byref(m), harness_fn, d, iters, x.ctypes.data)
    return x[: m.rows], err


def ps_generate(nmin, maxcol=12, seed=38):
    n = PsNet()
    _chk(lib().ps_generate(nmin, maxcol, seed, ctypes.byref(n)), "ps_generate")
    return n


def ps_committor(net, harness_fn, npfold, variant):
    """variant 0: PFOLD (Pfold.f90:219-223, the `a` array rewritten before every call); 1: NGT (NGT.f:999-1002)."""
    gp = np.zeros(net.nmin)
    t = ctypes.c_double(0)
    nnz = lib().ps_committor(ctypes.byref(net), harness_fn, npfold, variant, gp.ctypes.data, ctypes.byref(t))
    if nnz < 0:
        raise RuntimeError("ps_committor failed with %d" % nnz)
    return dict(gpfold=gp, nnz=int(nnz), t_calls=t.value)


def ps_free(net):
    lib().ps_free(ctypes.byref(net))


def mm_read(path):
    """Matrix Market coordinate (real / integer / pattern, general / symmetric) -> CsrD."""
    m = CsrD()
    _chk(lib().mm_read_csr(path.encode(), ctypes.byref(m)), "mm_read_csr(%s)" % path)
    return m


def pb_load(mtx_path, binary=False):
    """parboil's converter as its main.c calls it (mirrored = 1); `binary` = the file has no value column
    (the converter's own flag for pattern matrices, convert_dataset.c:93-100)."""
    m = CsrF()
    _chk(lib().pb_coo_to_csr(mtx_path.encode(), 1, 1 if binary else 0, ctypes.byref(m)), "pb_coo_to_csr")
    return m


def pb_run(m, f_harness_fn, vec_path, reps=50):
    x = np.zeros(m.cols + 2, np.float32)
    _chk(lib().pb_read_vec(vec_path.encode(), x.ctypes.data, m.rows), "pb_read_vec")
    y = np.zeros(m.rows + 2, np.float32)
    lib().pb_run(ctypes.byref(m), f_harness_fn, x.ctypes.data, y.ctypes.data, reps)
    return y[: m.rows]


def pb_compare(ref, cmp):
    ref = np.ascontiguousarray(ref, np.float32)
    cmp = np.ascontiguousarray(cmp, np.float32)
    return lib().pb_compare(ref.ctypes.data, cmp.ctypes.data, ref.size)


def free(m):
    (lib().csr_f_free if isinstance(m, CsrF) else lib().csr_d_free)(ctypes.byref(m))
