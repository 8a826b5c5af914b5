"""The BLOCKED plan's host side (csrc/spmv_blocked_plan.c) without a GPU: geometry, where every (stripe, bin) cell lies in
the two streams, phase 1's workgroups, phase 2's pieces and cut bins -- walked for their invariants, and the whole
two-phase product restated in numpy on top of them (the device builder's placement rule -- an entry's rank inside its cell is
its CSR position among the cell's entries -- and the two kernels' data flow) and held against the oracle.  The GPU tests
(tests/test_gpu_parity.py, test_blocked_plan) check the kernels; this file checks the plan they execute."""
import ctypes

import numpy as np
import pytest

import matgen
from rowtol import row_tolerance

WIN = 256


class BlockedPlan(ctypes.Structure):
    _fields_ = [("S", ctypes.c_int), ("wcols", ctypes.c_int), ("W", ctypes.c_int), ("RB", ctypes.c_int), ("B", ctypes.c_int),
                ("n_wg", ctypes.c_int), ("n_pieces", ctypes.c_int), ("n_slots", ctypes.c_int), ("n_join", ctypes.c_int),
                ("n1", ctypes.c_int64), ("n2", ctypes.c_int64),
                ("c1q", ctypes.POINTER(ctypes.c_uint32)), ("c2q", ctypes.POINTER(ctypes.c_uint32)),
                ("wg", ctypes.POINTER(ctypes.c_int)), ("pw", ctypes.POINTER(ctypes.c_int)), ("pb", ctypes.POINTER(ctypes.c_int)),
                ("ps", ctypes.POINTER(ctypes.c_int)), ("jn", ctypes.POINTER(ctypes.c_int))]


def geometry(lib, rows, cols, elem, bin_rows=0):
    g = BlockedPlan()
    lib.spmv_blocked_geometry.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(BlockedPlan)]
    rc = lib.spmv_blocked_geometry(rows, cols, elem, bin_rows, ctypes.byref(g))
    return rc, g


def build(lib, rs, ci, elem, bin_rows=0, auto=0):
    """-> (rc, plan as numpy arrays, padded cell sizes, cell of every entry)"""
    rows = rs.size - 1
    rc, g = geometry(lib, rows, int(ci.max()), elem, bin_rows)
    assert rc == 0
    rowid = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rs))
    cell = ((ci.astype(np.int64) - 1) // g.wcols) * g.B + rowid // g.RB
    cnt = np.bincount(cell, minlength=g.S * g.B).astype(np.int32)
    lib.spmv_blocked_plan_build.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(BlockedPlan)]
    rc = lib.spmv_blocked_plan_build(cnt.ctypes.data, int(ci.size), auto, ctypes.byref(g))
    if rc:
        return rc, None, None, None
    nc = g.S * g.B
    arr = lambda p, n, dt: np.ctypeslib.as_array(p, (n,)).astype(dt).copy() if n else np.zeros(0, dt)  # noqa: E731
    plan = dict(S=g.S, wcols=g.wcols, W=g.W, RB=g.RB, B=g.B, n_wg=g.n_wg, n_pieces=g.n_pieces, n_slots=g.n_slots, n_join=g.n_join,
                n1=g.n1, n2=g.n2, c1q=arr(g.c1q, nc + 1, np.int64), c2q=arr(g.c2q, nc, np.int64),
                wg=arr(g.wg, 3 * g.n_wg, np.int64).reshape(-1, 3), pw=arr(g.pw, g.n_pieces + 1, np.int64),
                pb=arr(g.pb, g.n_pieces, np.int64), ps=arr(g.ps, g.n_pieces, np.int64), jn=arr(g.jn, 3 * g.n_join, np.int64).reshape(-1, 3))
    lib.spmv_blocked_plan_free.argtypes = [ctypes.POINTER(BlockedPlan)]
    lib.spmv_blocked_plan_free(ctypes.byref(g))
    return 0, plan, cnt.astype(np.int64), cell


def check_invariants(p, padded, raw_cnt):
    S, B = p["S"], p["B"]
    assert np.all(padded % 4 == 0) and np.all(padded >= raw_cnt) and np.all(padded - raw_cnt < 4)
    assert p["n1"] % WIN == 0 and p["n2"] % WIN == 0
    # phase 1: cells in (stripe, bin) order, back to back inside a stripe, a stripe starting on a window
    c1 = p["c1q"] * 4
    assert c1[-1] == p["n1"]
    pad2 = padded.reshape(S, B)
    for s in range(S):
        assert c1[s * B] % WIN == 0
        assert np.array_equal(c1[s * B + 1:(s + 1) * B], c1[s * B] + np.cumsum(pad2[s])[:-1])
        end = c1[s * B] + pad2[s].sum()
        nxt = c1[(s + 1) * B]
        assert end <= nxt < end + WIN and nxt % WIN == 0
    # phase 1's workgroups: the windows of every stripe, each exactly once, in order
    wg = p["wg"]
    covered = np.zeros(p["n1"] // WIN, np.int64)
    for s, a, e in wg:
        assert c1[s * B] // WIN <= a < e <= c1[(s + 1) * B] // WIN
        covered[a:e] += 1
    assert np.all(covered == 1)
    # phase 2: a bin's cells by stripe; pieces tile the stream; a piece holds whole cells of ONE bin; cut bins have slots
    c2 = p["c2q"].reshape(S, B) * 4
    pw, pb, ps = p["pw"] * WIN, p["pb"], p["ps"]
    assert pw[0] == 0 and pw[-1] == p["n2"] and np.all(np.diff(pw) >= 0)
    assert np.all(np.diff(pb) >= 0) and set(pb.tolist()) == set(range(B))
    for b in range(B):
        mine = np.nonzero(pb == b)[0]
        assert np.array_equal(mine, np.arange(mine[0], mine[-1] + 1))
        lo, hi = pw[mine[0]], pw[mine[-1] + 1]
        starts = c2[:, b]
        ends = starts + pad2[:, b]
        assert np.all(starts >= lo) and np.all(ends <= hi) and np.all(starts[1:] >= ends[:-1])
        for s in np.nonzero(pad2[:, b])[0]:  # no cell straddles a piece boundary
            i = np.searchsorted(pw, starts[s], side="right") - 1
            assert pb[i] == b and ends[s] <= pw[i + 1]
        if mine.size == 1:
            assert ps[mine[0]] == -1
        else:
            assert np.array_equal(ps[mine], ps[mine[0]] + np.arange(mine.size)) and ps[mine[0]] >= 0
    jn = p["jn"]
    cut = [b for b in range(B) if np.count_nonzero(pb == b) > 1 or ps[np.nonzero(pb == b)[0][0]] >= 0]
    assert jn[:, 0].tolist() == cut
    for b, s0, n in jn:
        mine = np.nonzero(pb == b)[0]
        assert ps[mine[0]] == s0 and n == mine.size
    assert p["n_slots"] == int(jn[:, 2].sum()) if jn.size else p["n_slots"] == 0


def restated_product(p, padded, cell, rs, ci, a, x):
    """y = A x the way the plan lays it out: the device builder's placement (k_bb_scatter, k_bb_dest), k_blk_products,
    k_blk_bins, k_blk_join -- every product rounded before it is added, sums in double (any order inside a piece)."""
    S, B, RB, W, wcols = p["S"], p["B"], p["RB"], p["W"], p["wcols"]
    rows, nnz = rs.size - 1, ci.size
    n1, n2 = int(p["n1"]), int(p["n2"])
    # an entry's rank inside its cell: its CSR position among the cell's entries
    order = np.argsort(cell, kind="stable")
    first = np.concatenate(([0], np.cumsum(np.bincount(cell, minlength=S * B))))[:-1]
    rank = np.empty(nnz, np.int64)
    rank[order] = np.arange(nnz) - first[cell[order]]
    p1 = p["c1q"][cell] * 4 + rank
    p2 = p["c2q"][cell] * 4 + rank
    rowid = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rs))
    stripe = (ci.astype(np.int64) - 1) // wcols
    val1 = np.zeros(n1, a.dtype)
    col1 = np.full(n1, W - 1, np.int64)  # pads: the zero column
    lrow = np.zeros(n2, np.int64)
    val1[p1] = a
    col1[p1] = ci.astype(np.int64) - 1 - stripe * wcols
    lrow[p2] = rowid - (rowid // RB) * RB
    assert np.unique(p1).size == nnz and np.unique(p2).size == nnz  # nobody shares a place
    dest4 = np.full(n1 // 4, n2 // 4, np.int64)  # the spare group
    for cid in np.nonzero(padded)[0]:
        q = np.arange(padded[cid] // 4)
        dest4[p["c1q"][cid] + q] = p["c2q"][cid] + q
    # phase 1: every workgroup its stripe of x (zero beyond it), products to their destination
    prod = np.zeros(n2 + 4, a.dtype)
    for s, w0, w1 in p["wg"]:
        xs = np.zeros(W, a.dtype)
        c0 = s * wcols
        nx = min(wcols, x.size - c0)
        xs[:nx] = x[c0:c0 + nx]
        e = np.arange(w0 * WIN, w1 * WIN)
        pr = (val1[e] * xs[col1[e]]).astype(a.dtype)
        prod[dest4[e // 4] * 4 + e % 4] = pr
    assert np.all(prod[n2:] == 0)
    # phase 2: a wave per piece into its slice (doubles); uncut bins write y, cut ones their slot; then the join
    y = np.full(rows, np.nan, a.dtype)
    part = np.zeros((max(p["n_slots"], 1), RB))
    for i in range(p["n_pieces"]):
        e = np.arange(p["pw"][i] * WIN, p["pw"][i + 1] * WIN)
        sl = np.bincount(lrow[e], weights=prod[e].astype(np.float64), minlength=RB)
        b = p["pb"][i]
        r0, r1 = b * RB, min(rows, (b + 1) * RB)
        if p["ps"][i] < 0:
            y[r0:r1] = sl[: r1 - r0].astype(a.dtype)
        else:
            part[p["ps"][i]] = sl
    for b, s0, n in p["jn"]:
        r0, r1 = b * RB, min(rows, (b + 1) * RB)
        acc = part[s0].copy()
        for k in range(1, n):
            acc = acc + part[s0 + k]
        y[r0:r1] = acc[: r1 - r0].astype(a.dtype)
    assert not np.any(np.isnan(y))
    return y


CASES = {
    "sparsebench_12": lambda dt: matgen.sparsebench_like(5, 12, dt),
    "wide_short_rows": lambda dt: matgen.random_csr(6, 3001, 200_000, 5, dt, empty_frac=0.1, spread=0.6),
    "unsorted_repeats": lambda dt: matgen.random_csr(7, 1500, 90_000, 12, dt, sort_cols=False, distinct=False),
    "one_long_row": lambda dt: matgen.csr_from_lengths(np.random.default_rng(8), [3] * 700 + [30000] + [2] * 900, 150_000, dt),
    "few_rows": lambda dt: matgen.random_csr(9, 37, 70_000, 300, dt, spread=0.2),
    "narrow": lambda dt: matgen.random_csr(10, 2000, 300, 7, dt),
}


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("bin_rows", [64, 256, 1024])
def test_blocked_plan_invariants_and_restated_product(planlib, oracle, dtype, bin_rows):
    rng = np.random.default_rng(3)
    saw_cut = False
    for name, gen in CASES.items():
        rs, ci, a = gen(dtype)
        rows = rs.size - 1
        rowid = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rs))
        rc, p, padded, cell = build(planlib, rs, ci, a.itemsize, bin_rows)
        assert rc == 0, name
        raw = np.bincount(cell, minlength=p["S"] * p["B"])
        check_invariants(p, padded, raw)
        saw_cut |= p["n_join"] > 0
        x = rng.standard_normal(int(ci.max())).astype(dtype)
        y = restated_product(p, padded, cell, rs, ci, a, x)
        y_ref = oracle.spmv(a, x, rs, ci, rows)
        tol = row_tolerance(np.diff(rs), oracle.absum(a, x, rs, ci, rows), 2.0 ** (-24 if dtype == np.float32 else -53))
        assert np.all(np.abs(y.astype(np.float64) - y_ref) <= tol), name
        del rowid
    assert saw_cut  # the long row's bin is cut at every bin size tried here


def test_blocked_geometry_of_the_benchmark_shapes(planlib):
    """SparseBench 170^3: 256 stripes of 19 192 columns (one workgroup per CU), 1920 bins of 2560 rows (one round of 2048
    waves); beyond 5.2 M columns of doubles a second round of stripes; floats hold 32 764 columns per stripe; too many
    columns or cells are refused."""
    n = 170 ** 3
    rc, g = geometry(planlib, n, n, 8)
    assert rc == 0 and (g.S, g.wcols, g.RB, g.B, g.W) == (256, 19192, 2560, 1920, 20480)
    rc, g = geometry(planlib, 130 ** 3, 130 ** 3, 8)
    assert rc == 0 and (g.S, g.RB) == (256, 1280) and g.B <= 2048
    rc, g = geometry(planlib, 6_000_000, 6_000_000, 8)
    assert rc == 0 and g.S == 512 and g.wcols * g.S >= 6_000_000 and g.wcols <= 20478 and g.B <= 4096
    rc, g = geometry(planlib, n, n, 4)
    assert rc == 0 and g.W == 32768 and g.S == 256 and g.wcols % 4 == 0
    rc, g = geometry(planlib, 1000, 5000, 8)
    assert rc == 0 and g.wcols == 64 and g.S == 79 and g.RB == 1024 and g.B == 1
    assert geometry(planlib, 3000, 45_000_000, 8)[0] == -2      # more than 2048 stripes
    assert geometry(planlib, 40_000_000, 40_000_000, 8)[0] == -2  # more than 2^24 cells
    assert geometry(planlib, 0, 10, 8)[0] == -2 and geometry(planlib, 10, 10, 2)[0] == -2


def test_blocked_builder_declines_what_does_not_suit(planlib):
    """auto_choice: scattered columns are kept; columns local to the rows (few cells occupied) and a crowded cell (a dense
    corner: runs of one row's products, a piece that cannot be cut) are declined; by name (auto_choice 0) all are built."""
    rows = 300_000
    rng = np.random.default_rng(4)
    lens = np.full(rows, 5, np.int64)
    rs = np.concatenate(([1], 1 + np.cumsum(lens))).astype(np.int32)
    rowid = np.repeat(np.arange(rows, dtype=np.int64), lens)
    scattered = rng.integers(1, rows + 1, rowid.size).astype(np.int32)
    k = np.arange(rowid.size, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
    banded = ((rowid + 3 * k) % rows + 1).astype(np.int32)
    assert build(planlib, rs, scattered, 8, auto=1)[0] == 0
    assert build(planlib, rs, banded, 8, auto=1)[0] == -2
    assert build(planlib, rs, banded, 8, auto=0)[0] == 0
    corner = scattered.copy()
    hot = rowid < 500  # the first 500 rows keep to the first 1000 columns: one cell holds 2500 entries ...
    corner[hot] = rng.integers(1, 1001, int(hot.sum()))
    rc, p, padded, cell = build(planlib, rs, corner, 8, auto=1)
    assert rc == 0  # ... which is nothing beside a bin of 5000: kept
    lens2 = lens.copy()
    lens2[:64] = 4000  # 64 rows of 4000 entries inside one stripe: a cell of 256 000 where the mean bin has 6 000
    rs2 = np.concatenate(([1], 1 + np.cumsum(lens2))).astype(np.int32)
    rowid2 = np.repeat(np.arange(rows, dtype=np.int64), lens2)
    crowded = rng.integers(1, rows + 1, rowid2.size).astype(np.int32)
    crowded[rowid2 < 64] = rng.integers(1, 1001, int((rowid2 < 64).sum()))
    assert build(planlib, rs2, crowded, 8, auto=1)[0] == -2
    rc, p, padded, cell = build(planlib, rs2, crowded, 8, auto=0)
    assert rc == 0 and p["n_join"] >= 1
    # uneven throughout: a fifth of the bins hold three times the rest -- every one of them would be cut
    lens3 = lens.copy()
    lens3[: rows // 5] = 15
    rs3 = np.concatenate(([1], 1 + np.cumsum(lens3))).astype(np.int32)
    uneven = rng.integers(1, rows + 1, int(lens3.sum())).astype(np.int32)
    assert build(planlib, rs3, uneven, 8, auto=1)[0] == -2
    rc, p, padded, cell = build(planlib, rs3, uneven, 8, auto=0)
    assert rc == 0 and p["n_join"] * 20 > p["B"]
