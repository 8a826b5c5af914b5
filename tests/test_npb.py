"""Our NPB CG counterpart (lilac-benchmarks_amd/host/npb_cg.c) against the reference's pins:
the verification table of NPB3.3.1/CG/cg.f:122-163 (epsilon 1e-10, cg.f:363-368), the class
sizes measured on the reference binary (SURVEY.md 8(a) a3 / BASELINE.md section 2) and the
literal restatement of makea/sparse in oracle/npb_makea_oracle.c."""
import os

import numpy as np
import pytest

SIZES = {  # class: (na, nnz, min, max) -- BASELINE.md section 2 / SURVEY.md 8(a) a3
    "S": (1400, 78148, 8, 127),
    "W": (7000, 508402, 9, 168),
    "A": (14000, 1853104, 12, 294),
}
PARAMS = {"S": (1400, 7, 10.0), "W": (7000, 8, 12.0), "A": (14000, 11, 20.0)}


@pytest.mark.parametrize("cls", ["S", "W", "A"])
def test_fast_generator_is_bit_identical_to_literal_makea(pkg, oracle, cls):
    m = pkg.npb.NpbMatrix(cls)
    na, nnz, lo, hi = SIZES[cls]
    assert (m.na, m.nnz) == (na, nnz)
    lens = np.diff(m.rowstr)
    assert (lens.min(), lens.max()) == (lo, hi)
    rs, ci, a = oracle.npb_makea(*PARAMS[cls])
    assert np.array_equal(rs, m.rowstr)
    assert np.array_equal(ci, m.colidx)
    assert np.array_equal(a.view(np.uint64), m.a.view(np.uint64))
    # structural facts the kernels rely on: sorted distinct columns, no empty rows, cols in [1,na]
    assert m.rowstr[0] == 1 and m.colidx.min() >= 1 and m.colidx.max() <= na
    starts = m.rowstr[:-1] - 1
    d = np.diff(m.colidx)
    d[starts[1:] - 1] = 1
    assert np.all(d > 0)


def test_class_c_structure(pkg):
    m = pkg.npb.NpbMatrix("C")  # ~3 s
    assert (m.na, m.nnz) == (150000, 36121058)
    lens = np.diff(m.rowstr)
    assert (lens.min(), lens.max()) == (31, 525)
    assert m.colidx.max() <= 150000 and m.colidx.min() >= 1
    assert abs(m.total_mop() - 143347.5) < 1e-6  # total Mop of class C


@pytest.mark.parametrize("cls", ["S", "W"])
def test_cg_with_oracle_meets_npb_verification(pkg, oracle, cls):
    m = pkg.npb.NpbMatrix(cls)
    res = m.run_cg(oracle.harness_fn())
    assert res["spmv_calls"] == 26 * (1 + m.niter) == 416
    assert res["verified"] == 1 and res["err"] <= 1e-10
    assert abs(res["zeta"] - m.zeta_verify) / m.zeta_verify <= 1e-10


def test_cg_zeta_history_matches_golden_from_reference_spmv(pkg, oracle):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "spmv_golden.npz"))
    m = pkg.npb.NpbMatrix("S")
    res = m.run_cg(oracle.harness_fn())
    assert np.array_equal(np.array(res["zeta_hist"]), g["npb_S__zeta_hist"])  # same SpMV bits -> same CG bits
    assert np.array_equal(np.array(res["rnorm_hist"]), g["npb_S__rnorm_hist"])
    # and the generated matrix is the one the golden file holds
    assert np.array_equal(m.a.view(np.uint64), g["npb_S_first_call__a"].view(np.uint64))
    assert np.array_equal(m.colidx, g["npb_S_first_call__colidx"])


def test_class_e_is_rejected(pkg):
    with pytest.raises(RuntimeError):
        pkg.npb.NpbMatrix("E")  # 6.56e9 nonzeros do not fit the ABI's int


@pytest.mark.parametrize("parts", [2, 3])
def test_row_blocks_are_slices_of_the_whole_matrix(pkg, parts):
    """npb_makea_part generates only one block of the multi-GPU row partition;
    it must hold exactly the rows of the whole matrix, bit for bit, with a local rowstr."""
    w = pkg.npb.NpbMatrix("A")
    blocks = [pkg.npb.NpbMatrix("A", part=k, nparts=parts) for k in range(parts)]
    assert blocks[0].row_begin == 0 and blocks[-1].row_end == w.na
    nnz = [b.nnz for b in blocks]
    assert sum(nnz) == w.nnz and max(nnz) - min(nnz) < 0.01 * w.nnz
    for k, b in enumerate(blocks):
        if k:
            assert b.row_begin == blocks[k - 1].row_end
        s, e = w.rowstr[b.row_begin] - 1, w.rowstr[b.row_end] - 1
        assert np.array_equal(b.colidx, w.colidx[s:e])
        assert np.array_equal(b.a.view(np.uint64), w.a[s:e].view(np.uint64))
        assert np.array_equal(b.rowstr, w.rowstr[b.row_begin:b.row_end + 1] - w.rowstr[b.row_begin] + 1)


@pytest.mark.parametrize("cls", ["S", "W", "A"])
def test_generated_matrices_match_the_committed_digests(pkg, cls):
    """tests/golden/npb_digests.json: our own SHA-256 of rowstr / colidx / a (the values pass through libm's pow(),
    cg.f:819, so a toolchain change could move bits).  SURVEY's FNV checksums of the Fortran arrays are not used: they
    could not be reproduced with byte- or word-wise FNV-1a / FNV-1."""
    import hashlib
    import json
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "npb_digests.json")))["classes"][cls]
    m = pkg.npb.NpbMatrix(cls)
    assert (m.na, m.nnz) == (want["na"], want["nnz"])
    for name in ("rowstr", "colidx", "a"):
        got = hashlib.sha256(np.ascontiguousarray(getattr(m, name)).tobytes()).hexdigest()
        assert got == want[name], (cls, name)
