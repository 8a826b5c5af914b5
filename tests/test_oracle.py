"""Pins the oracle (oracle/spmv_oracle.c) before anything is checked against it:
 - the reference's own known-answer test (libspmv/test.cpp:44-54), exact ==;
 - the golden vectors produced by the reference's native.so (tests/golden/make_golden.py), bit for bit;
 - live, in the authoring container, against oracle/_ref/native.so on fresh seeded matrices."""
import os

import numpy as np
import pytest

import matgen

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "spmv_golden.npz")


def bits(a):
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN, allow_pickle=False)


def golden_case(g, name):
    return tuple(g["%s__%s" % (name, k)] for k in ("rowstr", "colidx", "a", "x", "y"))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_oracle_passes_reference_kat(oracle, dtype):
    rs, ci, a, x, y = matgen.kat_test_cpp(dtype)
    got = oracle.spmv(a, x, rs, ci, 12)
    assert np.array_equal(got, y)  # std::equal in test.cpp:53


def test_oracle_matches_every_golden_vector(oracle, golden):
    names = [str(n) for n in golden["names"]]
    assert len(names) >= 20
    for name in names:
        rs, ci, a, x, y = golden_case(golden, name)
        got = oracle.spmv(a, x, rs, ci, rs.size - 1)
        assert np.array_equal(bits(got), bits(y)), name


def test_golden_has_survey_pins(golden):
    # SURVEY.md 8(a) a7, measured on the reference binary: class S first-call y[0] and zeta history
    assert golden["npb_S_first_call__y"][0] == -2.1763393886130595
    z = golden["npb_S__zeta_hist"]
    assert abs(z[0] - 9.9986441579140) < 1e-12
    assert abs(z[-1] - 8.5971775078648) < 1e-12


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_bit_identical_to_reference_build(oracle, reference, dtype, seed):
    rng = np.random.default_rng(seed)
    rs, ci, a = matgen.random_csr(seed, 2000, 1777, 25, dtype, empty_frac=0.1, sort_cols=bool(seed % 2),
                                  distinct=bool(seed % 2))
    x = rng.standard_normal(1777).astype(dtype)
    y0 = reference.spmv(a, x, rs, ci, 2000)
    y1 = oracle.spmv(a, x, rs, ci, 2000)
    assert np.array_equal(bits(y0), bits(y1))


def test_oracle_bit_identical_to_reference_on_edge_cases(oracle, reference):
    rng = np.random.default_rng(9)
    for dtype in (np.float64, np.float32):
        for name, (rs, ci, a, cols) in matgen.edge_cases(dtype).items():
            x = rng.standard_normal(cols).astype(dtype)
            rows = rs.size - 1
            assert np.array_equal(bits(reference.spmv(a, x, rs, ci, rows)), bits(oracle.spmv(a, x, rs, ci, rows))), name


def test_oracle_writes_zero_for_empty_rows_and_overwrites(oracle):
    rs, ci, a, cols = matgen.edge_cases()["all_empty_but_last"]
    y = np.full(100, 123.0)
    oracle.spmv(a, np.ones(cols), rs, ci, 100, out=y)
    assert np.all(y[:99] == 0.0) and y[99] == a.sum()


def test_absum_bound_helper(oracle):
    rs, ci, a = matgen.random_csr(4, 100, 100, 10)
    x = np.random.default_rng(4).standard_normal(100)
    ab = oracle.absum(a, x, rs, ci, 100)
    y = oracle.spmv(a, x, rs, ci, 100)
    assert np.all(np.abs(y) <= ab * (1 + 1e-12) + 1e-300)


def test_threaded_oracle_is_bit_identical_to_the_sequential_one(oracle):
    """oracle/spmv_oracle_mt.c (bench.py's all-cores context line): rows split over threads, each row still summed
    left to right by one thread."""
    rs, ci, a = matgen.random_csr(91, 7000, 5000, 25, empty_frac=0.1, sort_cols=False, distinct=False)
    x = np.random.default_rng(91).standard_normal(5000)
    y0 = oracle.spmv(a, x, rs, ci, 7000)
    for nt in (1, 2, 7, 16, 300):
        y, used = oracle.spmv_mt(a, x, rs, ci, 7000, nt)
        assert 1 <= used <= 256 and np.array_equal(y.view(np.uint64), y0.view(np.uint64))
    y, used = oracle.spmv_mt(a, x, rs, ci, 3, 8)  # fewer rows than threads
    assert used == 3 and np.array_equal(y[:3].view(np.uint64), y0[:3].view(np.uint64))
