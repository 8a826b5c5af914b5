"""Host-side planning (csrc/spmv_plan.c): row blocks, long-row segments, bins, GPU partition.
Pure C, exercised through ctypes without a GPU."""
import ctypes
import os

import numpy as np
import pytest

import matgen

CAP, ROWCAP, SEG = 2044, 512, 2048


class StreamPlan(ctypes.Structure):
    _fields_ = [("nblk", ctypes.c_int), ("blk", ctypes.POINTER(ctypes.c_int)), ("nseg", ctypes.c_int),
                ("seg", ctypes.POINTER(ctypes.c_int)), ("nlrow", ctypes.c_int), ("lrow", ctypes.POINTER(ctypes.c_int)),
                ("nslots", ctypes.c_int), ("n_long_rows", ctypes.c_int)]


class BinPlan(ctypes.Structure):
    _fields_ = [("bin_off", ctypes.c_int * 5), ("bin_rows", ctypes.POINTER(ctypes.c_int))]


def build_stream(lib, rowstr, r0, r1):
    sp = StreamPlan()
    lib.spmv_plan_stream_build.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.POINTER(StreamPlan)]
    assert lib.spmv_plan_stream_build(rowstr.ctypes.data, r0, r1, CAP, ROWCAP, SEG, ctypes.byref(sp)) == 0
    blk = np.ctypeslib.as_array(sp.blk, (2 * (sp.nblk + 1),)).reshape(-1, 2).copy()
    seg = np.ctypeslib.as_array(sp.seg, (4 * sp.nseg,)).reshape(-1, 4).copy() if sp.nseg else np.zeros((0, 4), np.int32)
    lrow = np.ctypeslib.as_array(sp.lrow, (3 * sp.nlrow,)).reshape(-1, 3).copy() if sp.nlrow else np.zeros((0, 3), np.int32)
    res = dict(nblk=sp.nblk, blk=blk, seg=seg, lrow=lrow, nslots=sp.nslots, n_long=sp.n_long_rows)
    lib.spmv_plan_stream_free.argtypes = [ctypes.POINTER(StreamPlan)]
    lib.spmv_plan_stream_free(ctypes.byref(sp))
    return res


def check_stream_plan(rowstr, r0, r1, p):
    lens = np.diff(rowstr)[r0:r1]
    off = (rowstr[r0:r1 + 1] - rowstr[r0]).astype(np.int64)
    blk = p["blk"]
    rows = r1 - r0
    assert blk[-1, 0] == rows and blk[-1, 1] == off[-1]
    if rows:
        assert blk[0, 0] == 0 and blk[0, 1] == 0
    assert np.all(np.diff(blk[:, 0]) > 0) if rows else True  # every block has >= 1 row
    covered_long = 0
    for b in range(p["nblk"]):
        ra, rb = blk[b, 0], blk[b + 1, 0]
        assert blk[b, 1] == off[ra] and blk[b + 1, 1] == off[rb]
        n = off[rb] - off[ra]
        if n > CAP:  # long row: alone in its block
            assert rb - ra == 1
            covered_long += 1
        else:
            assert rb - ra <= ROWCAP
    assert covered_long == p["n_long"] == int((lens > CAP).sum())
    # segments tile every long row exactly, in order
    long_rows = np.nonzero(lens > CAP)[0]
    seg = p["seg"]
    k = 0
    slots_seen = []
    for r in long_rows:
        pos = off[r]
        nseg = 0
        while k < len(seg) and seg[k, 0] == r:
            assert seg[k, 1] == pos and seg[k, 2] - seg[k, 1] <= SEG and seg[k, 2] > seg[k, 1]
            pos = seg[k, 2]
            slots_seen.append(seg[k, 3])
            k += 1
            nseg += 1
        assert pos == off[r + 1] and nseg == -(-lens[r] // SEG)
    assert k == len(seg)
    multi = [s for s in slots_seen if s >= 0]
    assert multi == list(range(p["nslots"]))
    for row, s0, s1 in p["lrow"]:
        assert s1 - s0 == -(-lens[row] // SEG) >= 2


@pytest.fixture(scope="module")
def lib(planlib):
    return planlib


def test_stream_plan_on_edge_cases(lib):
    for name, (rs, ci, a, cols) in matgen.edge_cases().items():
        rows = rs.size - 1
        p = build_stream(lib, rs, 0, rows)
        check_stream_plan(rs, 0, rows, p)


def test_stream_plan_on_sub_ranges_and_skewed(lib):
    rs, ci, a = matgen.rmat_graph(3, 12, 8)
    rows = rs.size - 1
    for r0, r1 in ((0, rows), (0, 1), (17, 900), (rows - 5, rows), (100, 100)):
        check_stream_plan(rs, r0, r1, build_stream(lib, rs, r0, r1))
    rs, ci, a = matgen.npb_like(5, 3000, 240)
    p = build_stream(lib, rs, 0, 3000)
    check_stream_plan(rs, 0, 3000, p)
    fill = np.diff(p["blk"][:, 1])
    assert fill.mean() > 0.85 * CAP  # blocks of NPB-like rows are well filled


def test_row_cap_with_empty_rows(lib):
    rs = np.ones(5001, np.int32)  # 5000 empty rows
    rs[-1] = 1
    p = build_stream(lib, rs, 0, 5000)
    check_stream_plan(rs, 0, 5000, p)
    assert p["nblk"] == -(-5000 // ROWCAP)


def test_bins(lib):
    rs, ci, a, cols = matgen.edge_cases()["rows_1_to_600"]
    bp = BinPlan()
    lib.spmv_plan_bins_build.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(BinPlan)]
    assert lib.spmv_plan_bins_build(rs.ctypes.data, 0, 600, CAP, ctypes.byref(bp)) == 0
    off = list(bp.bin_off)
    assert off[0] == 0 and off[-1] == 600
    rows = np.ctypeslib.as_array(bp.bin_rows, (600,)).copy()
    assert sorted(rows) == list(range(600))
    lens = np.diff(rs)
    limits = [(0, 2), (3, 16), (17, 128), (129, 1 << 30)]
    for b, (lo, hi) in enumerate(limits):
        r = rows[off[b]:off[b + 1]]
        assert np.all(np.diff(r) > 0)  # stable
        assert np.all((lens[r] >= lo) & (lens[r] <= hi))
    lib.spmv_plan_bins_free.argtypes = [ctypes.POINTER(BinPlan)]
    lib.spmv_plan_bins_free(ctypes.byref(bp))


@pytest.mark.parametrize("parts", [1, 2, 3, 4, 8])
def test_partition_balances_nnz(pkg, parts):
    be = pkg.Backend()
    rs, ci, a = matgen.npb_like(6, 5000, 100)
    b = be.partition_rows(rs, 5000, parts)
    assert b[0] == 0 and b[-1] == 5000 and np.all(np.diff(b) >= 0)
    nnz = rs[b[1:]] - rs[b[:-1]]
    assert nnz.sum() == rs[-1] - 1
    assert nnz.max() - nnz.min() <= 2 * np.diff(rs).max()


def test_partition_degenerate(pkg):
    be = pkg.Backend()
    rs = np.array([1, 1, 1, 1, 1], np.int32)  # 4 empty rows
    assert list(be.partition_rows(rs, 4, 2)) == [0, 2, 4]
    rs = np.array([1, 10001], np.int32)  # one heavy row
    b = be.partition_rows(rs, 1, 4)
    assert b[0] == 0 and b[-1] == 1 and np.all(np.diff(b) >= 0)
    assert list(be.partition_rows(np.array([1], np.int32), 0, 3)) == [0, 0, 0, 0]


class TiledHost(ctypes.Structure):
    _fields_ = [("elem", ctypes.c_int), ("W", ctypes.c_int), ("S", ctypes.c_int), ("WPB", ctypes.c_int),
                ("RW", ctypes.c_int), ("B", ctypes.c_int), ("n_groups", ctypes.c_int), ("n_packed", ctypes.c_int64),
                ("nonempty_chunks", ctypes.c_int64), ("val", ctypes.c_void_p), ("col", ctypes.POINTER(ctypes.c_uint16)),
                ("wptr", ctypes.POINTER(ctypes.c_int32)), ("seg", ctypes.POINTER(ctypes.c_uint16))]


def tiled_emulate(lib, rs, ci, a, x, target_blocks=512):
    """Runs csrc/spmv_tiled.c and then does on the CPU exactly what k_tiled does with the layout:
    per row, stripe by stripe, products rounded then added front to back."""
    rows = rs.size - 1
    cols = int(ci.max()) if ci.size else 0
    t = TiledHost()
    lib.spmv_tiled_build.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(TiledHost)]
    rc = lib.spmv_tiled_build(a.ctypes.data, rs.ctypes.data, ci.ctypes.data, 0, rows, cols, a.dtype.itemsize,
                              target_blocks, ctypes.byref(t))
    if rc != 0:
        return rc, None
    nch = t.n_groups * t.S
    wptr = np.ctypeslib.as_array(t.wptr, (nch + 1,))
    seg = np.ctypeslib.as_array(t.seg, (nch * (t.RW + 1),)).reshape(nch, t.RW + 1)
    col = np.ctypeslib.as_array(t.col, (t.n_packed + 1,))
    val = np.ctypeslib.as_array(ctypes.cast(t.val, ctypes.POINTER(ctypes.c_double if a.dtype == np.float64 else ctypes.c_float)),
                                (t.n_packed + 1,))
    assert np.all(wptr % 8 == 0) and np.all(np.diff(wptr) >= 0)
    assert seg[:, -1].sum() == rs[-1] - rs[0]
    y = np.zeros(rows, a.dtype)
    for g in range(t.n_groups):
        for r in range(t.RW):
            row = g * t.RW + r
            if row >= rows:
                break
            acc = a.dtype.type(0)
            for s in range(t.S):
                ch = g * t.S + s
                q0 = int(wptr[ch])
                for k in range(int(seg[ch, r]), int(seg[ch, r + 1])):
                    p = val[q0 + k] * x[s * t.W + int(col[q0 + k])]
                    acc = acc + p
            y[row] = acc
    geo = dict(W=t.W, S=t.S, RW=t.RW, B=t.B, WPB=t.WPB, n_packed=t.n_packed)
    lib.spmv_tiled_free.argtypes = [ctypes.POINTER(TiledHost)]
    lib.spmv_tiled_free(ctypes.byref(t))
    return 0, (y, geo)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_tiled_layout_reproduces_the_reference_order(lib, oracle, dtype):
    """Sorted columns (every NPB matrix): stripe-by-stripe order == left-to-right order, so the TILED
    arithmetic is bit-identical to native-impl.c; unsorted columns: equal within the reordering bound."""
    rng = np.random.default_rng(3)
    rs, ci, a = matgen.random_csr(8, 700, 40000 if dtype == np.float64 else 70000, 9, dtype, empty_frac=0.1)
    x = rng.standard_normal(int(ci.max())).astype(dtype)
    rc, (y, geo) = tiled_emulate(lib, rs, ci, a, x, target_blocks=4)
    assert rc == 0 and geo["RW"] == -(-700 // (geo["WPB"] * 4)) and geo["W"] * a.dtype.itemsize == 32768
    assert geo["S"] == -(-int(ci.max()) // geo["W"])
    y_ref = oracle.spmv(a, x, rs, ci, 700)
    assert np.array_equal(y.view(np.uint32 if dtype == np.float32 else np.uint64),
                          y_ref.view(np.uint32 if dtype == np.float32 else np.uint64))
    rs, ci, a = matgen.random_csr(9, 300, 30000, 12, dtype, sort_cols=False, distinct=False)
    x = rng.standard_normal(int(ci.max())).astype(dtype)
    rc, (y, geo) = tiled_emulate(lib, rs, ci, a, x, target_blocks=2)
    y_ref = oracle.spmv(a, x, rs, ci, 300)
    eps = 2.0 ** -53 if dtype == np.float64 else 2.0 ** -24
    assert np.all(np.abs(y.astype(np.float64) - y_ref) <= (np.diff(rs) + 2) * eps * oracle.absum(a, x, rs, ci, 300))


def test_tiled_rejects_unrepresentable_chunks(lib):
    rs, ci, a, cols = matgen.edge_cases()["long_rows_mixed"]
    big = np.arange(1, 70001, dtype=np.int32)  # one row with 70000 entries inside one 4096-column stripe
    rs2 = np.array([1, 70001], np.int32)
    ci2 = ((big - 1) % 4000 + 1).astype(np.int32)
    rc, _ = tiled_emulate(lib, rs2, ci2, np.ones(70000), np.ones(4000))
    assert rc == -2


class StripedHost(ctypes.Structure):
    _fields_ = [("elem", ctypes.c_int), ("W", ctypes.c_int), ("S", ctypes.c_int), ("WPB", ctypes.c_int),
                ("EPL", ctypes.c_int), ("NB", ctypes.c_int), ("NG", ctypes.c_int), ("GM", ctypes.c_int),
                ("n_wg", ctypes.c_int), ("n_ranges", ctypes.c_int), ("cap", ctypes.c_int), ("skewed", ctypes.c_int),
                ("n_packed", ctypes.c_int64), ("n_dummies", ctypes.c_int64), ("max_tile", ctypes.c_int64),
                ("val", ctypes.c_void_p), ("col", ctypes.POINTER(ctypes.c_uint16)),
                ("tptr", ctypes.POINTER(ctypes.c_int32)), ("rrow", ctypes.POINTER(ctypes.c_int32)),
                ("wgd", ctypes.POINTER(ctypes.c_int32))]


class StripedOpts(ctypes.Structure):
    _fields_ = [("xbytes", ctypes.c_int), ("wpb", ctypes.c_int), ("target_wgs", ctypes.c_int),
                ("wave_rows", ctypes.c_int), ("groups", ctypes.c_int), ("xcd_bias", ctypes.c_int), ("stripes", ctypes.c_int)]


def striped_build(lib, rs, ci, a, xbytes=65536, wpb=8, wgs=6, wave_rows=128, groups=0):
    rows = rs.size - 1
    cols = int(ci.max()) if ci.size else 0
    t = StripedHost()
    o = StripedOpts(xbytes, wpb, wgs, wave_rows, groups, 0, 0)
    lib.spmv_striped_build.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.POINTER(StripedOpts), ctypes.POINTER(StripedHost)]
    rc = lib.spmv_striped_build(a.ctypes.data, rs.ctypes.data, ci.ctypes.data, 0, rows, cols, a.dtype.itemsize,
                                ctypes.byref(o), ctypes.byref(t))
    return rc, t


def striped_emulate(lib, rs, ci, a, x, xbytes=65536, wpb=8, wgs=6, wave_rows=128, groups=0):
    """Runs csrc/spmv_striped.c, checks the invariants k_striped relies on, and evaluates the layout on the CPU
    the way the kernel walks it: workgroup (stripe group, row block of that group), wave, stripe after stripe of the
    group; rows are recovered by counting row-end flags, row sums accumulate across the stripes of a group, the
    groups' partial sums are added in group order."""
    rows = rs.size - 1
    rc, t = striped_build(lib, rs, ci, a, xbytes, wpb, wgs, wave_rows, groups)
    if rc != 0:
        return rc, None
    nr = t.n_ranges
    assert nr == t.n_wg * t.WPB and t.GM == -(-t.S // t.NG) and 1 <= t.NG <= t.S
    tptr = np.ctypeslib.as_array(t.tptr, (nr * (t.GM + 1),)).astype(np.int64).reshape(nr, t.GM + 1)
    rrow = np.ctypeslib.as_array(t.rrow, (nr * 2,)).astype(np.int64).reshape(nr, 2)
    wgd = np.ctypeslib.as_array(t.wgd, (t.n_wg * 4,)).astype(np.int64).reshape(t.n_wg, 4)
    col = np.ctypeslib.as_array(t.col, (t.n_packed + 1,)).astype(np.int64)
    ctype = ctypes.c_double if a.dtype == np.float64 else ctypes.c_float
    val = np.ctypeslib.as_array(ctypes.cast(t.val, ctypes.POINTER(ctype)), (t.n_packed + 1,))
    sub = 64 * t.EPL
    assert np.all(tptr % sub == 0) and np.all(np.diff(tptr.ravel()) >= 0) and tptr.ravel()[-1] == t.n_packed
    assert np.all(rrow[:, 1] >= rrow[:, 0]) and np.all(rrow[:, 1] - rrow[:, 0] <= wave_rows)
    # every group tiles the rows completely, in order, with its own cuts; workgroups are listed group-major
    assert np.all(np.diff(wgd[:, 2]) >= 0) and wgd[0, 2] == 0 and wgd[-1, 2] == t.NG - 1
    for g in range(t.NG):
        qs = np.nonzero(wgd[:, 2] == g)[0]
        assert qs.size >= 1 and np.array_equal(wgd[qs, 3], np.arange(qs.size)) and qs.size <= t.NB
        assert np.all(wgd[qs, 0] == g * t.S // t.NG) and np.all(wgd[qs, 1] == (g + 1) * t.S // t.NG)
        rr = rrow[qs[0] * t.WPB:(qs[-1] + 1) * t.WPB]
        assert rr[0, 0] == 0 and rr[-1, 1] == rows and np.array_equal(rr[1:, 0], rr[:-1, 1])
        if rows >= rr.shape[0]:
            assert np.all(rr[:, 1] > rr[:, 0])  # no idle wave while rows last
    # undo the lane interleave of the values inside each sub-window
    G = 16 // a.dtype.itemsize
    e = np.arange(t.n_packed, dtype=np.int64)
    rr_ = e % sub
    phys = e // sub * sub + (rr_ % t.EPL) // G * (64 * G) + (rr_ // t.EPL) * G + (rr_ % t.EPL) % G
    val = val[:t.n_packed][phys]
    flags = col[:t.n_packed] >> 15
    assert flags.sum() == t.S * rows  # exactly one row end per (stripe, row)
    zero_col = min(xbytes // a.dtype.itemsize, 32768) - 1
    assert t.cap - 1 == zero_col and t.W <= zero_col
    partial = np.zeros((t.NG, rows), a.dtype)
    n_fill = 0
    for q in range(t.n_wg):
        s0, s1, g = int(wgd[q, 0]), int(wgd[q, 1]), int(wgd[q, 2])
        for w in range(t.WPB):
            u = q * t.WPB + w
            nrows = int(rrow[u, 1] - rrow[u, 0])
            acc = np.zeros(nrows, a.dtype)
            for si in range(t.GM + 1):
                if s0 + si >= s1:
                    assert tptr[u, si] == tptr[u, min(s1 - s0, t.GM)]  # past the group's last stripe: the end
                    continue
                s = s0 + si
                row = 0
                run = a.dtype.type(0)
                for p in range(int(tptr[u, si]), int(tptr[u, si + 1])):
                    c = int(col[p])
                    lc = c & 0x7fff
                    if lc >= t.W:  # a filler or a pad: times the zeroed last element of the LDS buffer, never x
                        assert lc == zero_col and val[p] == 0
                        n_fill += c >> 15
                        xv = a.dtype.type(0)
                    else:
                        assert s * t.W + lc < x.size
                        xv = x[s * t.W + lc]
                    run = run + val[p] * xv
                    if c >> 15:
                        acc[row] = acc[row] + run
                        row += 1
                        run = a.dtype.type(0)
                assert row == nrows, (q, w, si, row, nrows)
            partial[g, rrow[u, 0]:rrow[u, 1]] = acc
    assert n_fill == t.n_dummies
    y = partial[0].copy()
    for g in range(1, t.NG):
        y = y + partial[g]
    tiles = np.diff(tptr, axis=1)
    geo = dict(W=t.W, S=t.S, NB=t.NB, NG=t.NG, GM=t.GM, WPB=t.WPB, n_wg=t.n_wg, n_packed=t.n_packed,
               n_dummies=t.n_dummies, max_tile=t.max_tile, skewed=t.skewed, tiles=tiles, wgd=wgd)
    lib.spmv_striped_free.argtypes = [ctypes.POINTER(StripedHost)]
    lib.spmv_striped_free(ctypes.byref(t))
    return 0, (y, geo)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_striped_layout(lib, oracle, dtype):
    rng = np.random.default_rng(5)
    eps = 2.0 ** -53 if dtype == np.float64 else 2.0 ** -24
    cases = [matgen.random_csr(18, 900, 30000, 40, dtype, empty_frac=0.05, sort_cols=False, distinct=False),
             matgen.npb_like(19, 700, 120, dtype),
             matgen.random_csr(20, 50, 20000, 300, dtype)]
    for rs, ci, a in cases:
        rows = rs.size - 1
        x = rng.standard_normal(int(ci.max())).astype(dtype)
        y_ref = oracle.spmv(a, x, rs, ci, rows)
        absum = oracle.absum(a, x, rs, ci, rows)
        seen = set()
        for groups in (0, 1, 2, 1000):  # chosen; one group (a workgroup walks every stripe, no partials); two; one per stripe
            rc, res = striped_emulate(lib, rs, ci, a, x, groups=groups)
            assert rc == 0
            y, geo = res
            seen.add((geo["NG"], geo["NB"]))
            unit = 16 // a.dtype.itemsize
            assert geo["S"] == -(-int(ci.max()) // geo["W"]) and (geo["W"] + unit) * a.dtype.itemsize <= 65536
            assert geo["W"] % unit == 0
            assert geo["W"] - -(-int(ci.max()) // geo["S"]) < unit  # equal stripes, rounded up to 16 bytes of x
            if groups == 0:
                assert geo["n_wg"] <= max(6, geo["NG"] * -(-rows // (8 * 128)))
            if groups == 1:
                assert geo["NG"] == 1
            tol = (np.diff(rs) + geo["S"] + 2) * eps * absum
            assert np.all(np.abs(y.astype(np.float64) - y_ref) <= tol)
        assert len(seen) >= 2 or geo["S"] == 1, seen


def test_striped_wave_ranges_respect_the_accumulator_limit(lib):
    """Rows of very unequal weight: the equal-weight cut must still keep every wave range within wave_rows rows
    (the kernel's row accumulators are registers: lane l, register k <-> row 64 k + l) and leave no range empty."""
    lens = np.concatenate([np.full(3000, 1), np.full(40, 900), np.full(2000, 2)]).astype(np.int64)
    rs, ci, a = matgen.csr_from_lengths(np.random.default_rng(3), lens, 2000)
    x = np.random.default_rng(4).standard_normal(2000)
    rc, res = striped_emulate(lib, rs, ci, a, x, wgs=4, wave_rows=256)
    assert rc == 0
    y, geo = res
    ref = np.array([np.dot(a[rs[r] - 1:rs[r + 1] - 1], x[ci[rs[r] - 1:rs[r + 1] - 1] - 1]) for r in range(lens.size)])
    assert np.allclose(y, ref, rtol=1e-12, atol=1e-12)


def test_striped_rejects_hypersparse(lib):
    rs, ci, a = matgen.sparsebench_like(3, 30)  # 27000 rows, ~5 per row over 27000 columns: 4 stripes, mostly dummies
    rc, _ = striped_emulate(lib, rs, ci, a, np.ones(27000))
    assert rc == -2


_LAYOUT_HASH = r"""
import sys, ctypes, hashlib, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import matgen
from conftest import build_planlib
from test_plan import StripedHost
lib = ctypes.CDLL(build_planlib())
rs, ci, a = matgen.random_csr(77, 9000, 40000, 60)
from test_plan import striped_build
rc, t = striped_build(lib, rs, ci, a, 65536, 8, 24, 128, 0)
assert rc == 0
h = hashlib.sha256()
h.update(np.ctypeslib.as_array(t.col, (t.n_packed,)).tobytes())
h.update(np.ctypeslib.as_array(ctypes.cast(t.val, ctypes.POINTER(ctypes.c_double)), (t.n_packed,)).tobytes())
h.update(np.ctypeslib.as_array(t.tptr, (t.n_ranges * (t.GM + 1),)).tobytes())
h.update(np.ctypeslib.as_array(t.rrow, (t.n_ranges * 2,)).tobytes())
h.update(np.ctypeslib.as_array(t.wgd, (t.n_wg * 4,)).tobytes())
print(h.hexdigest(), t.n_packed, t.n_dummies)
"""


def test_striped_layout_does_not_depend_on_the_thread_count():
    """The conversion runs its passes on host threads (SPMV_HIP_BUILD_THREADS); the layout, and with it
    the order of every addition in k_striped, must be the same for any number of them."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _LAYOUT_HASH % (root, os.path.join(root, "tests"))
    outs = []
    for nt in ("1", "3", "8"):
        env = dict(os.environ, SPMV_HIP_BUILD_THREADS=nt)
        r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout.strip())
    assert outs[0] == outs[1] == outs[2] and len(outs[0].split()[0]) == 64


def test_striped_groups_balance_their_workgroups(lib):
    """Stripes that do not divide into the groups: groups of 3 stripes and of 2.  Each group cuts the rows by ITS
    entries and gets row blocks in proportion to them, so the longest workgroup is within a few per cent of the
    average (with the same cuts and the same block count for every group the larger groups ran half as long again),
    and within a group every wave's stream is as long as its neighbours'."""
    rng = np.random.default_rng(8)
    rows, cols = 6000, 8190 * 12
    rs, ci, a = matgen.random_csr(8, rows, cols, 60, spread=0.3)
    x = rng.standard_normal(int(ci.max()))
    rc, res = striped_emulate(lib, rs, ci, a, x, xbytes=65536, wpb=8, wgs=40, wave_rows=256, groups=5)
    assert rc == 0
    y, geo = res
    assert geo["S"] % geo["NG"] != 0  # uneven groups, or the test tests nothing
    per_wave = geo["tiles"].sum(axis=1).reshape(geo["n_wg"], geo["WPB"])  # entries streamed per wave
    per_wg = per_wave.max(axis=1)
    assert per_wg.max() <= 1.12 * per_wg.mean(), (per_wg.max(), per_wg.mean())
    assert 36 <= geo["n_wg"] <= 40


def test_striped_geometry_prefers_stripes_that_divide_into_the_groups(lib):
    """The class D row block of round 3 in small: the fewest stripes that fit the buffer (23) do not divide into any
    admissible group count, the accumulators (wave_rows) hold the smaller groups at a minimum of row blocks, and the
    workgroups of the larger groups would stream a good deal more than the others.  The builder then takes a few more,
    slightly narrower stripes (24 = 8 groups x 3) so that all workgroups are alike."""
    rng = np.random.default_rng(11)
    rows, cols = 6144, 8190 * 23 - 500
    rs, ci, a = matgen.random_csr(11, rows, cols, 400, spread=0.2)
    x = rng.standard_normal(int(ci.max()))
    rc, res = striped_emulate(lib, rs, ci, a, x, xbytes=65536, wpb=8, wgs=40, wave_rows=256, groups=0)
    assert rc == 0
    y, geo = res
    assert geo["S"] > 23 and geo["S"] % geo["NG"] == 0, geo  # equal groups
    assert 36 <= geo["n_wg"] <= 40
    per_wave = geo["tiles"].sum(axis=1).reshape(geo["n_wg"], geo["WPB"])
    per_wg = per_wave.sum(axis=1)
    assert per_wg.max() <= 1.04 * per_wg.mean(), (per_wg.max(), per_wg.mean())
    # and the product is the product
    prod = a * x[ci - 1]
    want = np.add.reduceat(prod, rs[:-1] - 1)
    assert np.allclose(y, want, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_the_exact_order_restatement_agrees_with_the_oracle(lib, oracle, dtype):
    """tests/striped_order.py restates k_striped's order of additions in numpy (the GPU test compares the kernel's bits with
    it).  Here, without a GPU: it is a correct product -- within the derived STRIPED bar of the oracle -- on shapes that
    exercise one and several stripes and groups, rows that span sub-windows, empty segments, crowded sub-windows, and it
    differs from the plain left-to-right sum in the last bits (or the restatement would be of the wrong order)."""
    from rowtol import STRIPED, row_tolerance
    from striped_order import striped_exact
    rng = np.random.default_rng(12)
    eps = 2.0 ** -53 if dtype == np.float64 else 2.0 ** -24
    cases = {"npb_like": matgen.npb_like(41, 3000, 200, dtype),
             "several_stripes_and_groups": matgen.random_csr(42, 5000, 90000, 70, dtype, spread=0.4),
             "one_entry_rows": matgen.csr_from_lengths(rng, [1] * 3000, 16000, dtype),
             "long_rows": matgen.csr_from_lengths(rng, [3000] * 40, 20000, dtype)}
    differs = 0
    for name, (rs, ci, a) in cases.items():
        rows = rs.size - 1
        x = rng.standard_normal(int(ci.max())).astype(dtype)
        y = striped_exact(lib, rs, ci, a, x)
        assert y is not None, name
        y_ref = oracle.spmv(a, x, rs, ci, rows)
        tol = row_tolerance(np.diff(rs), oracle.absum(a, x, rs, ci, rows), eps, STRIPED, 128)
        assert np.all(np.abs(y.astype(np.float64) - y_ref.astype(np.float64)) <= tol), name
        differs += int(np.count_nonzero(y != y_ref))
        if name == "one_entry_rows":
            assert np.array_equal(y, y_ref)  # one product per row: nothing to reorder
    assert differs > 0
