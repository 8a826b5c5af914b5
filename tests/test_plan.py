"""Host-side planning (csrc/spmv_plan.c): row blocks, long-row segments, bins, GPU partition.
Pure C, exercised through ctypes without a GPU."""
import ctypes

import numpy as np
import pytest

import matgen

CAP, ROWCAP, SEG = 2044, 512, 16384


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
def lib(pkg):
    return pkg.Backend().lib


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
