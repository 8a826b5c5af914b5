"""The HUB plan's host side (csrc/spmv_hub_plan.c) without a GPU: which columns become hubs, how the entry stream is cut into
chunks, which row ends a chunk starts with, which rows cross chunk boundaries, which rows are empty -- walked for their
invariants, and the whole product restated in numpy on top of them, sub-step by sub-step in the kernel's own order of
additions (lane-local sums, the six steps of the segmented wave scan, the carry between sub-steps, the closing lane's running
sum, the carries of a row that crosses chunks added in chunk order by a 64-lane strided sum and a fixed tree, then the piece)
and held against the oracle.  tests/hub_order.py is that restatement; the GPU tests (tests/test_gpu_parity.py) compare the
kernel's BITS with it.  This file checks the plan the kernels execute."""
import numpy as np
import pytest

import hub_order
import matgen
from rowtol import HUB, row_tolerance

CASES = {
    "rmat_12": lambda dt: matgen.rmat_graph(5, 12, 8, dt, values="normal"),
    "rmat_14_hub_rows": lambda dt: matgen.rmat_graph(6, 14, 16, dt, values="normal"),
    "many_empty": lambda dt: matgen.edge_cases(dt)["many_empty"][:3],
    "long_rows_mixed": lambda dt: matgen.edge_cases(dt)["long_rows_mixed"][:3],
    "rows_1_to_600": lambda dt: matgen.edge_cases(dt)["rows_1_to_600"][:3],
    "one_col": lambda dt: matgen.edge_cases(dt)["one_col"][:3],
    "all_empty_but_last": lambda dt: matgen.edge_cases(dt)["all_empty_but_last"][:3],
    "tall": lambda dt: matgen.edge_cases(dt)["tall"][:3],
    "npb_like": lambda dt: matgen.npb_like(8, 900, 60, dt),
    "chunk_exact": lambda dt: matgen.csr_from_lengths(np.random.default_rng(3), [1024, 512, 512, 1, 1023, 2048, 1], 3000, dt),
}


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name", sorted(CASES))
def test_hub_plan_and_its_restatement(planlib, oracle, name, dtype):
    rs, ci, a = CASES[name](dtype)
    rows, nnz = rs.size - 1, int(ci.size)
    cols = int(ci.max())
    x = np.random.default_rng(11).standard_normal(cols).astype(dtype)
    for kmax, kmax2 in ((hub_order.lds_slots(planlib, dtype), 0), (7, 0), (0, 0), (7, 20), (5, 1 << 20), (0, 9)):
        p = hub_order.plan(planlib, rs, ci, cols, kmax, kmax2=kmax2)
        # ---- the hubs: the kmax most referenced columns, none referenced once, ascending, slot_of their inverse
        cnt = np.bincount(ci - 1, minlength=cols)
        K = p["K"]
        assert K <= kmax and np.all(np.diff(p["hubcol"]) > 0)
        assert np.array_equal(p["slot_of"][p["hubcol"]], np.arange(K))
        assert np.array_equal(np.nonzero((p["slot_of"] >= 0) & (p["slot_of"] < K))[0], p["hubcol"])
        if K:
            assert cnt[p["hubcol"]].min() >= 2
            rest = np.delete(cnt, p["hubcol"])
            if kmax2:
                pass  # with a second tier the left-over columns are judged below
            elif rest.size and K == kmax:
                assert rest.max() <= cnt[p["hubcol"]].min()  # nobody left out is referenced more than a hub
            elif rest.size:
                assert rest.max() < 2  # room left: every column referenced twice or more is in
        assert p["covered"] == int(cnt[p["hubcol"]].sum())
        assert p["cols_used"] == int(np.count_nonzero(cnt))
        # ---- tier 1 (the dense table in memory): the kmax2 most referenced of the REST, slots behind tier 0's
        K2, h2 = p["K2"], p["hubcol2"]
        assert K2 <= kmax2 and np.all(np.diff(h2) > 0) and not np.intersect1d(h2, p["hubcol"]).size
        assert np.array_equal(p["slot_of"][h2], K + np.arange(K2)) and int(np.count_nonzero(p["slot_of"] >= 0)) == K + K2
        assert p["covered2"] == int(cnt[h2].sum())
        if K2:
            assert cnt[h2].min() >= 2
            rest2 = np.delete(cnt, np.concatenate([p["hubcol"], h2]))
            if rest2.size:
                assert rest2.max() <= cnt[h2].min() if K2 == kmax2 else rest2.max() < 2
            if K:
                assert cnt[h2].max() <= cnt[p["hubcol"]].min()  # nobody in tier 1 is referenced more than a tier-0 hub
        # ---- the rows
        CH = p["CH"]
        lens = np.diff(rs)
        assert p["n_chunks"] == -(-nnz // CH)
        assert np.array_equal(p["nzrow"], np.nonzero(lens)[0]) and np.array_equal(p["erow"], np.nonzero(lens == 0)[0])
        ends = rs[1:][lens > 0].astype(np.int64) - rs[0] - 1  # 0-based position of every row's last entry
        want_rank0 = np.searchsorted(ends, np.arange(p["n_chunks"] + 1, dtype=np.int64) * CH, side="left")
        assert np.array_equal(p["rank0"], want_rank0)
        first = rs[:-1][lens > 0].astype(np.int64) - rs[0]
        crossing = np.nonzero(ends // CH > first // CH)[0]
        assert np.array_equal(p["fix"][:, 0], p["nzrow"][crossing])
        assert np.array_equal(p["fix"][:, 1], first[crossing] // CH) and np.array_equal(p["fix"][:, 2], ends[crossing] // CH)
        # ---- the product, in the kernel's order, on top of the plan
        y = hub_order.product(p, rs, ci, a, x)
        y_ref = oracle.spmv(a, x, rs, ci, rows)
        tol = row_tolerance(lens, oracle.absum(a, x, rs, ci, rows), 2.0 ** -53 if dtype == np.float64 else 2.0 ** -24, HUB)
        assert np.all(np.abs(y.astype(np.float64) - y_ref.astype(np.float64)) <= tol), (name, kmax)
        assert np.all(y[lens == 0] == 0)


def test_integer_products_are_exact_in_the_hub_order(planlib, oracle):
    """bfs (bfs/bfs.cc:36-68): frontier values are small integers in FP64, every partial sum is exact, so the restatement --
    like the kernel -- must agree with the oracle bit for bit whatever the order."""
    rs, ci, a = matgen.rmat_graph(9, 13, 8, np.float64, values="ones")
    cols = int(ci.max())
    x = (np.random.default_rng(2).random(cols) < 0.1).astype(np.float64)
    p = hub_order.plan(planlib, rs, ci, cols, 64)
    y = hub_order.product(p, rs, ci, a, x)
    assert np.array_equal(y, oracle.spmv(a, x, rs, ci, rs.size - 1))
