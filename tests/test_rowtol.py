"""tests/rowtol.py states a bound, (n + D + 1) u S, for the difference between the reference's left-to-right sum of
rounded products and ANY summation scheme whose longest chain of additions is D.  This checks the derivation where it can
be checked without a GPU: sums of the same products in many random bracketings (random binary trees of depth <= D, with
and without fused multiply-adds at the leaves) stay inside the bar, for short and long rows, benign and cancelling
data -- and the bar is not vacuous: it is within two orders of magnitude of errors that actually occur."""
import numpy as np

from rowtol import CSR, STRIPED, row_tolerance


def reference_sum(p):
    acc = np.float64(0.0)
    for v in p:  # native-impl.c:6-9: one accumulator, left to right
        acc = acc + v
    return acc


def random_tree_sum(rng, p):
    """adds the products in a random bracketing; returns (sum, longest chain of additions)"""
    vals = [(np.float64(v), 0) for v in p]
    while len(vals) > 1:
        i = int(rng.integers(0, len(vals) - 1))
        (a, da), (b, db) = vals[i], vals[i + 1]
        vals[i:i + 2] = [(a + b, max(da, db) + 1)]
    return vals[0]


def test_any_bracketing_stays_inside_the_bar():
    rng = np.random.default_rng(5)
    worst = 0.0
    for n in (1, 2, 3, 7, 40, 241, 463, 2000):
        for kind in ("normal", "cancelling", "wide_range"):
            for _ in range(30):
                a = rng.standard_normal(n)
                x = rng.standard_normal(n)
                if kind == "cancelling":
                    a[1::2] = -a[::2][: a[1::2].size]
                    x[1::2] = x[::2][: x[1::2].size] * (1 + 1e-9)
                if kind == "wide_range":
                    a *= 10.0 ** rng.integers(-8, 8, n)
                p = a * x  # rounded products, as the reference forms them
                ref = reference_sum(p)
                S = float(np.abs(p).sum())
                got, depth = random_tree_sum(rng, p)
                # the scheme's own chain is `depth`; the bar for it is (n + depth + 1) u S
                bar = 1.01 * (n + depth + 1) * 2.0 ** -53 * S
                assert abs(float(got) - float(ref)) <= bar, (n, kind, depth)
                # and the plans' published bars cover every scheme of their depth class
                assert abs(float(got) - float(ref)) <= row_tolerance([n], [S], 2.0 ** -53, CSR)[0] or depth > n + 5
                if S > 0:
                    worst = max(worst, abs(float(got) - float(ref)) / (2.0 ** -53 * S * (n + depth + 1)))
    assert worst > 0.005  # the bar is a bound, not a blanket: some case comes within a factor of 200 of it


def test_the_striped_bar_matches_the_kernels_scheme():
    """STRIPED's chain: <= 15 additions inside a sub-window, one carry per further sub-window of a segment, one addition per
    further stripe and group the row has entries in.  A row of n entries spread over s stripes: D <= 15 + ceil(n/256) +
    2 min(n, s); the bar must grow with both and never fall below the single-stripe case."""
    lens = np.array([1, 5, 241, 463, 5000])
    S = np.ones(lens.size)
    one = row_tolerance(lens, S, 2.0 ** -53, STRIPED, 1)
    ten = row_tolerance(lens, S, 2.0 ** -53, STRIPED, 10)
    many = row_tolerance(lens, S, 2.0 ** -53, STRIPED, 128)
    assert np.all(one <= ten) and np.all(ten <= many)
    assert np.allclose(ten / (1.01 * 2.0 ** -53), lens + 16 + np.ceil(lens / 256.0) + 2 * np.minimum(lens, 10))
    assert np.all(row_tolerance(lens, S, 2.0 ** -53) >= np.maximum(many, row_tolerance(lens, S, 2.0 ** -53, CSR)) * (1 - 1e-12))
