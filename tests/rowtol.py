"""The per-row parity bar of the plans that do not add in the reference's order -- derived, not tuned.

native-impl.c:1-25 computes, per row, s_ref = fl(...fl(fl(p_1 + p_2) + p_3)... + p_n) with p_j = fl(a_j x_j), n = the row's
length, u = 2^-53 (FP64) or 2^-24 (FP32) the unit roundoff.  Every rounding is a relative perturbation (1 + d), |d| <= u,
of a partial sum whose magnitude is at most S = sum_j |a_j x_j|, and a product passes through at most `depth` additions on
its way into the result, so to first order

    |s - sum_j a_j x_j|  <=  (depth + 1) u S          (the + 1: the product's own rounding; an FMA folds it into the add)

for ANY summation scheme whose longest chain of additions is `depth`; adding an exact zero (a filler, a pad, a stripe
the row has no entry in) is exact and does not count.  The reference's chain is n - 1 long, so

    |y_kernel - y_ref|  <=  (n + D + 1) u S,          D = the longest chain of additions of the kernel's scheme.

D per plan (csrc/spmv_kernels.hip):
  ORDERED       the reference's own order, no FMA: bit-identical, no bar.
  TILED         stored order, product rounded first: bit-identical when a row's columns are sorted, else D = n - 1.
  STREAM        products parked in LDS, a lane group of G = 1..64 lanes adds every G-th one sequentially, then log2 G
  VECTOR        tree steps: D = ceil(n / G) - 1 + log2 G <= n + 5 whatever G the block picked; VECTOR and the long-row
  BINNED        kernels (k_long: 2048-entry segments, a fixed tree per segment, the segments' sums added lane-strided and
                by a tree) stay below the same figure.  Bar: (2 n + 6) u S.
  STRIPED       per 256-entry sub-window: up to 4 lane-local adds, 6 steps of the segmented wave scan, 1 carry from the
                previous sub-window, up to 4 more adds of the lane that closes the row: 15; a (row, stripe) segment that
                spans k sub-windows carries k - 1 times more; then one add per further stripe of the group the row has
                entries in and one per further group -- at most min(n, stripes) each (exact zeros do not count):
                D = 15 + ceil(n / 256) + 2 min(n, stripes).  Bar: (n + 16 + ceil(n/256) + 2 min(n, stripes)) u S.
  BLOCKED       a row's products are added to its sum in LDS in stream order (with RUNS: a run is summed first, <= 4 lane-local
                adds and the scan's 6 steps), a cut bin's slots in slot order: D <= n + 10: the CSR bar covers it.
  HUB           (csrc/spmv_hub.hip) per sub-step of 64 lanes x E entries (E = 2 doubles / 4 floats): up to E lane-local adds,
                6 steps of the segmented wave scan, 1 carry from the previous sub-step, up to E more adds of the lane that
                closes the row: 2 E + 7 <= 15; a row that spans k sub-steps carries k - 1 times more; a row that crosses
                chunks of 1024 entries adds its chunks' carries lane-strided (ceil(chunks / 64) adds), a 6-step tree, and
                the piece in its last chunk: D = 22 + ceil(n / 128) + ceil(n / 65536).
                Bar: (n + 23 + ceil(n/128) + ceil(n/65536)) u S.
Second-order terms ((n u)^2 S) are covered by the factor 1.01.  When the plan is not known (the ABI's automatic choice)
the larger of the two bars applies.  Round 2 used (n + 24) u S, a constant chosen to cover the shapes tested; for the NPB
shapes the derived STRIPED bar is within a few per cent of it (class C: n = 241, 10 stripes: 278 against 265).
"""
import numpy as np

STRIPED, CSR, HUB = "striped", "csr", "hub"


def row_tolerance(lens, absum, eps, plan=None, stripes=128):
    """lens: nonzeros per row; absum: sum_j |a_j x_j| per row (float64); eps: the unit roundoff of the dtype;
    plan: "striped", "csr" (STREAM / VECTOR / BINNED / TILED on unsorted rows) or None (either: the larger bar);
    stripes: column stripes of the STRIPED layout (ceil(cols / plan_param)); the default, 128, is the most the
    automatic choice accepts (spmv_plan_choose)."""
    n = np.asarray(lens, dtype=np.float64)
    csr = 2.0 * n + 6.0
    striped = n + 16.0 + np.ceil(n / 256.0) + 2.0 * np.minimum(n, float(stripes))
    if plan == HUB:
        k = n + 23.0 + np.ceil(n / 128.0) + np.ceil(n / 65536.0)
    elif plan == STRIPED:
        k = striped
    elif plan == CSR:
        k = csr
    else:
        k = np.maximum(csr, striped)
    return 1.01 * k * eps * np.asarray(absum, dtype=np.float64)
