"""Property tests (hypothesis) for the host-side layout builders: for ANY small CSR shape -- empty rows,
one column, duplicate and unsorted columns, a row longer than everything else, rows that start anywhere
in rowstr -- each builder either declines (-2) or produces a layout whose CPU emulation (the same walk
the kernel does, tests/test_plan.py) reproduces y = A x within the reordering bound.  No GPU needed."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import matgen
from test_plan import striped_emulate, tiled_emulate
import test_blocked_layout as blk
import hub_order
from rowtol import HUB, row_tolerance


@pytest.fixture(scope="module")
def lib(planlib):
    return planlib


@st.composite
def csr_shapes(draw):
    rows = draw(st.integers(1, 400))
    cols = draw(st.sampled_from([1, 2, 17, 300, 5000, 9000, 20000]))
    kind = draw(st.sampled_from(["uniform", "skewed", "mostly_empty", "one_long"]))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        lens = rng.integers(0, min(cols, 40) + 1, rows)
    elif kind == "skewed":
        lens = np.minimum((rng.pareto(1.2, rows) * 3).astype(np.int64), min(cols, 3000))
    elif kind == "mostly_empty":
        lens = np.where(rng.random(rows) < 0.8, 0, rng.integers(1, min(cols, 10) + 1, rows))
    else:
        lens = rng.integers(0, min(cols, 6) + 1, rows)
        lens[rng.integers(0, rows)] = min(cols, 2500)
    distinct = draw(st.booleans())
    sort_cols = draw(st.booleans())
    if distinct:
        lens = np.minimum(lens, cols)
    rs, ci, a = matgen.csr_from_lengths(rng, [int(v) for v in lens], cols, np.float64, sort_cols=sort_cols,
                                        distinct=distinct)
    return rs, ci, a, seed


def reference(rs, ci, a, x):
    """native-impl.c:1-12 in numpy: left to right per row; plus sum |a x| for the tolerance."""
    rows = rs.size - 1
    y = np.zeros(rows)
    ab = np.zeros(rows)
    for r in range(rows):
        lo, hi = rs[r] - 1, rs[r + 1] - 1
        acc = 0.0
        for j in range(lo, hi):
            acc += a[j] * x[ci[j] - 1]
        y[r] = acc
        ab[r] = np.abs(a[lo:hi] * x[ci[lo:hi] - 1]).sum()
    return y, ab


COMMON = dict(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large,
                                                                   HealthCheck.function_scoped_fixture])


def check(rc_res, rs, ci, a, x, extra_terms):
    rc, res = rc_res
    if rc == -2:
        return
    assert rc == 0
    y = res[0]
    y_ref, ab = reference(rs, ci, a, x)
    tol = (np.diff(rs) + extra_terms + 4) * 2.0 ** -53 * ab
    assert np.all(np.abs(y - y_ref) <= tol + 1e-300)


@settings(**COMMON)
@given(csr_shapes())
def test_striped_layout_on_any_shape(lib, shape):
    rs, ci, a, seed = shape
    if ci.size == 0:
        return
    x = np.random.default_rng(seed + 1).standard_normal(int(ci.max()))
    check(striped_emulate(lib, rs, ci, a, x), rs, ci, a, x, 8)


@settings(**COMMON)
@given(csr_shapes())
def test_tiled_layout_on_any_shape(lib, shape):
    rs, ci, a, seed = shape
    if ci.size == 0:
        return
    x = np.random.default_rng(seed + 3).standard_normal(int(ci.max()))
    check(tiled_emulate(lib, rs, ci, a, x, target_blocks=7), rs, ci, a, x, 8)


@settings(**COMMON)
@given(csr_shapes(), st.sampled_from([64, 128, 1024]))
def test_blocked_plan_on_any_shape(lib, shape, bin_rows):
    """csrc/spmv_blocked_plan.c: for any shape and bin size the plan keeps its invariants (cells disjoint and in order, every
    window of phase 1 covered once, no cell across a piece boundary, cut bins with consecutive slots) and the two-phase
    product restated on it reproduces y = A x within the any-order bound."""
    rs, ci, a, seed = shape
    if ci.size == 0:
        return
    rc, p, padded, cell = blk.build(lib, rs, ci, 8, bin_rows)
    assert rc == 0
    blk.check_invariants(p, padded, np.bincount(cell, minlength=p["S"] * p["B"]))
    x = np.random.default_rng(seed + 5).standard_normal(int(ci.max()))
    y = blk.restated_product(p, padded, cell, rs, ci, a, x)
    y_ref, ab = reference(rs, ci, a, x)
    tol = (np.diff(rs) + 12) * 2.0 ** -53 * ab
    assert np.all(np.abs(y - y_ref) <= tol + 1e-300)


@settings(**COMMON)
@given(csr_shapes(), st.sampled_from([0, 1, 5, 64, 16352]), st.sampled_from([0, 3, 4000]), st.sampled_from([np.float64, np.float32]))
def test_hub_plan_on_any_shape(lib, shape, kmax, kmax2, dtype):
    """csrc/spmv_hub_plan.c + tests/hub_order.py: for any shape, any sizes of the two table tiers and both precisions the plan keeps
    its invariants -- hub slots a bijection onto the chosen columns, the tiers disjoint, chunk ranks monotone and ending at
    the number of rows that have entries, every row that crosses a chunk boundary listed once with its chunks -- and the
    product restated in the kernel's own order of additions reproduces y = A x within HUB's derived bar."""
    rs, ci, a, seed = shape
    if ci.size == 0:
        return
    a = a.astype(dtype)
    cols = int(ci.max())
    p = hub_order.plan(lib, rs, ci, cols, kmax, kmax2=kmax2)
    cnt = np.bincount(ci - 1, minlength=cols)
    K, K2 = p["K"], p["K2"]
    assert K <= kmax and K2 <= kmax2 and not np.intersect1d(p["hubcol"], p["hubcol2"]).size
    assert np.array_equal(p["slot_of"][p["hubcol"]], np.arange(K)) and np.array_equal(p["slot_of"][p["hubcol2"]], K + np.arange(K2))
    assert int(np.count_nonzero(p["slot_of"] >= 0)) == K + K2
    assert p["covered"] == int(cnt[p["hubcol"]].sum()) and p["covered2"] == int(cnt[p["hubcol2"]].sum())
    lens = np.diff(rs)
    assert np.all(np.diff(p["rank0"]) >= 0) and p["rank0"][-1] == int(np.count_nonzero(lens)) == p["nzrow"].size
    assert np.array_equal(np.sort(np.concatenate([p["nzrow"], p["erow"]])), np.arange(rs.size - 1))
    assert np.all(p["fix"][:, 1] < p["fix"][:, 2]) and np.unique(p["fix"][:, 0]).size == p["fix"].shape[0]
    x = np.random.default_rng(seed + 7).standard_normal(cols).astype(dtype)
    y = hub_order.product(p, rs, ci, a, x)
    a64, x64 = a.astype(np.float64), x.astype(np.float64)
    y_ref = np.zeros(rs.size - 1)
    ab = np.zeros(rs.size - 1)
    for r in range(rs.size - 1):  # the reference's order, in the matrix's own precision
        lo, hi = rs[r] - 1, rs[r + 1] - 1
        acc = dtype(0)
        for j in range(lo, hi):
            acc = dtype(acc + dtype(a[j] * x[ci[j] - 1]))
        y_ref[r] = acc
        ab[r] = np.abs(a64[lo:hi] * x64[ci[lo:hi] - 1]).sum()
    tol = row_tolerance(lens, ab, 2.0 ** -53 if dtype == np.float64 else 2.0 ** -24, HUB)
    assert np.all(np.abs(y.astype(np.float64) - y_ref) <= tol + 1e-300)
    assert np.all(y[lens == 0] == 0)
