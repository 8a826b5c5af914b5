"""The summation order of the HUB plan (csrc/spmv_hub.hip), restated in numpy -- addition by addition -- on top of the plan its
host side makes (csrc/spmv_hub_plan.c).

The entry stream (CSR order, values where they lie) is cut into chunks of CH = 1024 entries, whatever the rows.  A wave
takes a chunk in sub-steps of 64 lanes x E entries (E = 2 doubles / 4 floats: one 16-byte load), lane l holding entries
E l .. E l + E - 1 of the sub-step:

    p[i]   = fl(value * x[column])                (hub or not: the LDS table and the table in memory hold the same x; pads: 0 * 0)
    tailv  = sum of the lane's products after its last row end         (t = t + p[i]; reset to 0 at a row end)
    (sv, sf) = segmented inclusive scan of (tailv, lane has a row end) over the wave, the six DPP steps of spmv_wave.h
    sv     = sf ? sv : sv + carry                  (carry: the open sum of the previous sub-step; 0 at the chunk's start)
    cin    = sv of the lane below (lane 0: carry);  carry = sv of lane 63
    run    = cin;  for i: run = run + p[i];  at a row end: y[row] = run, run = 0
    after the chunk's last sub-step: carry_out[chunk] = carry

Row ends are numbered rank0[chunk], rank0[chunk] + 1, ... in entry order; nzrow[rank] is the row.  A row whose entries lie
in chunks cf .. cl (cf < cl) is finished by k_hub_fix: lane l of a wave adds carry_out[cf + l], [cf + l + 64], ... in that
order, group_sum<64>'s fixed tree (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror, row_bcast:15 into
rows 1 and 3, row_bcast:31 into rows 2 and 3; lane 63 holds the sum), then y[row] = t + y[row].  Empty rows get 0.

Nothing is atomic and no product is contracted into an add, so the GPU's result is a function of the plan alone and can
be compared with this BIT FOR BIT (tests/test_gpu_parity.py).  Test infrastructure only.
"""
import ctypes

import numpy as np

from striped_order import _segscan

LANES = np.arange(64)


class HubPlan(ctypes.Structure):
    _fields_ = [("K", ctypes.c_int), ("K2", ctypes.c_int), ("covered", ctypes.c_int64), ("covered2", ctypes.c_int64), ("cols_used", ctypes.c_int),
                ("CH", ctypes.c_int), ("n_chunks", ctypes.c_int),
                ("n_nzrows", ctypes.c_int), ("n_empty", ctypes.c_int), ("n_fix", ctypes.c_int), ("identity", ctypes.c_int),
                ("hubcol", ctypes.POINTER(ctypes.c_int)), ("slot_of", ctypes.POINTER(ctypes.c_int)),
                ("rank0", ctypes.POINTER(ctypes.c_int)), ("nzrow", ctypes.POINTER(ctypes.c_int)),
                ("erow", ctypes.POINTER(ctypes.c_int)), ("fix", ctypes.POINTER(ctypes.c_int))]


def lds_slots(lib, dtype):
    lib.spmv_hub_lds_slots.argtypes = [ctypes.c_int]
    return lib.spmv_hub_lds_slots(np.dtype(dtype).itemsize)


def plan(lib, rs, ci, cols, kmax, ch=1024, min_count=2, kmax2=0):
    """The plan of csrc/spmv_hub_plan.c as numpy arrays (the column counts are the device's job: np.bincount here)."""
    g = HubPlan()
    cnt = np.ascontiguousarray(np.bincount(ci - 1, minlength=cols).astype(np.int32))
    lib.spmv_hub_select.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(HubPlan)]
    assert lib.spmv_hub_select(cnt.ctypes.data, cols, kmax, kmax2, min_count, ctypes.byref(g)) == 0
    rsc = np.ascontiguousarray(rs, dtype=np.int32)
    lib.spmv_hub_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(HubPlan)]
    assert lib.spmv_hub_rows(rsc.ctypes.data, int(rsc[0]), rsc.size - 1, ch, ctypes.byref(g)) == 0
    arr = lambda p, n: np.ctypeslib.as_array(p, (n,)).astype(np.int64).copy() if n else np.zeros(0, np.int64)  # noqa: E731
    both = arr(g.hubcol, g.K + g.K2)
    out = dict(K=g.K, K2=g.K2, covered=int(g.covered), covered2=int(g.covered2), cols_used=g.cols_used, CH=g.CH, n_chunks=g.n_chunks, identity=g.identity,
               hubcol=both[:g.K], hubcol2=both[g.K:],
               slot_of=arr(g.slot_of, cols), rank0=arr(g.rank0, g.n_chunks + 1), nzrow=arr(g.nzrow, g.n_nzrows),
               erow=arr(g.erow, g.n_empty), fix=arr(g.fix, 3 * g.n_fix).reshape(-1, 3))
    assert out["identity"] == int(g.n_empty == 0)
    lib.spmv_hub_plan_free.argtypes = [ctypes.POINTER(HubPlan)]
    lib.spmv_hub_plan_free(ctypes.byref(g))
    return out


def _group_sum64(v):
    """spmv_wave.h group_sum<T, 64> over the last axis (64 lanes); lane 63 of the result holds the sum"""
    v = v + v[..., LANES ^ 1]
    v = v + v[..., LANES ^ 2]
    v = v + v[..., (LANES & ~7) | (7 - (LANES & 7))]
    v = v + v[..., (LANES & ~15) | (15 - (LANES & 15))]
    add = np.zeros_like(v)
    add[..., 16:32] = v[..., 15:16]
    add[..., 48:64] = v[..., 47:48]
    v = v + add
    add = np.zeros_like(v)
    add[..., 32:64] = v[..., 31:32]
    return v + add


def product(p, rs, ci, a, x):
    """y = A x in the HUB kernels' own order of additions, on top of plan `p`."""
    dtype = a.dtype
    zero = dtype.type(0)
    rows, nnz = rs.size - 1, int(ci.size)
    CH, nch = p["CH"], p["n_chunks"]
    E = 16 // dtype.itemsize
    sub = 64 * E
    # the stream, padded behind its end as the device copy is (value 0, the table's zero slot)
    val = np.zeros(nch * CH, dtype)
    val[:nnz] = a
    xv = np.zeros(nch * CH, dtype)
    xv[:nnz] = np.asarray(x, dtype)[ci - 1]
    end = np.zeros(nch * CH, bool)
    lens = np.diff(rs)
    end[(rs[1:][lens > 0].astype(np.int64) - rs[0] - 1)] = True
    prod = (val * xv).reshape(nch, CH // sub, 64, E)
    flag = end.reshape(nch, CH // sub, 64, E)
    y = np.zeros(rows, dtype)
    carry = np.zeros(nch, dtype)
    sums = np.zeros((nch, CH), dtype)  # the sum emitted at every flagged entry
    for k in range(CH // sub):
        pk, fk = prod[:, k], flag[:, k]
        tail = np.zeros((nch, 64), dtype)
        for i in range(E):
            tail = np.where(fk[:, :, i], zero, tail + pk[:, :, i])
        sv, sf = _segscan(tail, fk.any(axis=2))
        sv = np.where(sf, sv, sv + carry[:, None])
        run = np.concatenate([carry[:, None], sv[:, :63]], axis=1)
        carry = sv[:, 63].copy()
        for i in range(E):
            run = run + pk[:, :, i]
            sums[:, k * sub + np.arange(64) * E + i] = run
            run = np.where(fk[:, :, i], zero, run)
    f2 = end.reshape(nch, CH)
    cs = np.cumsum(f2, axis=1)
    rank = p["rank0"][:nch, None] + cs - 1
    assert np.array_equal(p["rank0"][1:], p["rank0"][:-1] + cs[:, -1]) if nch else True
    y[p["nzrow"][rank[f2]]] = sums[f2]
    for row, cf, cl in p["fix"]:
        lanes = np.zeros(64, dtype)
        for c in range(cf, cl):
            lanes[(c - cf) % 64] = lanes[(c - cf) % 64] + carry[c]  # after the last sub-step: what the chunk leaves open
        t = _group_sum64(lanes)[63]
        y[row] = t + y[row]
    y[p["erow"]] = zero
    return y

