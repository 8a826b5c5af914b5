"""Seeded synthetic CSR inputs for the parity tests (1-based rowstr/colidx, as the ABI wants).

Shapes follow SURVEY.md 8(d): NPB-like uniform rows, SparseBench-like short rows with uniformly
random columns (big_gen.py:60-83 semantics: row count max(1, trunc(N(5,4))), values abs(N(0,2))),
R-MAT power-law graphs, 2-D grid "road" graphs, plus the edge cases the reference's own
known-answer test exercises (libspmv/test.cpp:44-54: empty rows, a one-element last row, a
column equal to the number of rows)."""
import numpy as np


def csr_from_lengths(rng, lengths, cols, dtype=np.float64, sort_cols=True, distinct=True, values="normal"):
    lengths = np.asarray(lengths, dtype=np.int64)
    rows = lengths.size
    rowstr = np.empty(rows + 1, np.int32)
    rowstr[0] = 1
    np.cumsum(lengths, out=rowstr[1:])
    rowstr[1:] += 1
    nnz = int(lengths.sum())
    colidx = np.empty(nnz, np.int32)
    pos = 0
    for n in lengths:
        n = int(n)
        if n == 0:
            continue
        if distinct and n <= cols:
            if n * 4 < cols:
                c = np.unique(rng.integers(1, cols + 1, size=2 * n + 8))
                while c.size < n:
                    c = np.unique(np.concatenate([c, rng.integers(1, cols + 1, size=2 * n + 8)]))
                c = rng.permutation(c)[:n]
            else:
                c = rng.permutation(cols)[:n] + 1
        else:
            c = rng.integers(1, cols + 1, size=n)
        if sort_cols:
            c = np.sort(c)
        colidx[pos:pos + n] = c
        pos += n
    if values == "normal":
        a = rng.standard_normal(nnz)
    elif values == "ones":
        a = np.ones(nnz)
    elif values == "absnormal":
        a = np.abs(rng.normal(0.0, 2.0, nnz))
    else:
        raise ValueError(values)
    return rowstr, colidx, a.astype(dtype)


def random_csr(seed, rows, cols, mean_len, dtype=np.float64, empty_frac=0.0, spread=0.5, **kw):
    rng = np.random.default_rng(seed)
    lengths = np.maximum(0, np.rint(rng.normal(mean_len, spread * mean_len, rows))).astype(np.int64)
    lengths = np.minimum(lengths, cols)
    if empty_frac > 0:
        lengths[rng.random(rows) < empty_frac] = 0
    return csr_from_lengths(rng, lengths, cols, dtype, **kw)


def npb_like(seed, rows, mean_len=240, dtype=np.float64):
    """rows of ~mean_len sorted distinct columns, like NPB CG."""
    rng = np.random.default_rng(seed)
    lengths = np.clip(np.rint(rng.normal(mean_len, mean_len / 4, rows)), 8, min(rows, 4 * mean_len)).astype(np.int64)
    return csr_from_lengths(rng, lengths, rows, dtype)


def sparsebench_like(seed, side, dtype=np.float64):
    """n = side^3 rows, max(1, trunc(N(5,4))) distinct uniformly random sorted columns per row,
    values abs(N(0,2)) -- the shape big_gen.py:60-83 writes (own seeded PRNG; the reference
    generator is unseeded)."""
    rng = np.random.default_rng(seed)
    n = side ** 3
    lengths = np.maximum(1, np.trunc(rng.normal(5.0, 4.0, n))).astype(np.int64)
    lengths = np.minimum(lengths, n)
    return csr_from_lengths(rng, lengths, n, dtype, values="absnormal")


def rmat_graph(seed, scale, edge_factor=8, dtype=np.float64, values="ones"):
    """R-MAT (a,b,c,d = .57,.19,.19,.05) adjacency in CSR with duplicate edges merged."""
    rng = np.random.default_rng(seed)
    n = 1 << scale
    m = n * edge_factor
    src = np.zeros(m, np.int64)
    dst = np.zeros(m, np.int64)
    for bit in range(scale):
        r = rng.random(m)
        dst_bit = ((r >= 0.57) & (r < 0.76)) | (r >= 0.95)
        src_bit = r >= 0.76
        src |= src_bit.astype(np.int64) << bit
        dst |= dst_bit.astype(np.int64) << bit
    key = np.unique(src * n + dst)
    src, dst = key // n, key % n
    lengths = np.bincount(src, minlength=n)
    rowstr = np.empty(n + 1, np.int32)
    rowstr[0] = 1
    rowstr[1:] = np.cumsum(lengths) + 1
    colidx = (dst + 1).astype(np.int32)
    a = np.ones(colidx.size, dtype) if values == "ones" else rng.standard_normal(colidx.size).astype(dtype)
    return rowstr, colidx, a


def grid2d_graph(nx, ny, dtype=np.float64):
    """4-neighbour grid ("road"-like: degree <= 4), values 1.0."""
    idx = np.arange(nx * ny).reshape(ny, nx)
    pairs = []
    pairs.append((idx[:, :-1].ravel(), idx[:, 1:].ravel()))
    pairs.append((idx[:, 1:].ravel(), idx[:, :-1].ravel()))
    pairs.append((idx[:-1, :].ravel(), idx[1:, :].ravel()))
    pairs.append((idx[1:, :].ravel(), idx[:-1, :].ravel()))
    src = np.concatenate([p[0] for p in pairs])
    dst = np.concatenate([p[1] for p in pairs])
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    n = nx * ny
    rowstr = np.empty(n + 1, np.int32)
    rowstr[0] = 1
    rowstr[1:] = np.cumsum(np.bincount(src, minlength=n)) + 1
    return rowstr, (dst + 1).astype(np.int32), np.ones(dst.size, dtype)


def kat_test_cpp(dtype=np.float64):
    """The reference's known-answer test, data from /root/reference/libspmv/test.cpp:44-54."""
    a = np.array([5.0, 8.0, 3.0, 6.0], dtype)
    rowstr = np.array([1, 2, 2, 3, 3, 3, 4, 4, 4, 4, 4, 4, 5], np.int32)
    colidx = np.array([1, 4, 2, 12], np.int32)
    x = np.array([1.0] * 6 + [2.0] * 6, dtype)
    y = np.array([5.0, 0.0, 8.0, 0.0, 0.0, 3.0, 0.0, 0.0, 0.0, 0.0, 0.0, 12.0], dtype)
    return rowstr, colidx, a, x, y


def edge_cases(dtype=np.float64):
    """name -> (rowstr, colidx, a, cols): the ragged / degenerate shapes of SURVEY section 7 step 1."""
    rng = np.random.default_rng(7)
    out = {}
    out["one_row_one_col"] = csr_from_lengths(rng, [1], 1, dtype) + (1,)
    out["one_row_long"] = csr_from_lengths(rng, [5000], 6000, dtype) + (6000,)
    out["one_col"] = csr_from_lengths(rng, [1] * 300, 1, dtype) + (1,)
    out["all_empty_but_last"] = csr_from_lengths(rng, [0] * 99 + [3], 50, dtype) + (50,)
    out["empty_first_rows"] = csr_from_lengths(rng, [0] * 70 + [4] * 70, 33, dtype) + (33,)
    out["wide"] = csr_from_lengths(rng, rng.integers(0, 9, 40), 100000, dtype) + (100000,)
    out["tall"] = csr_from_lengths(rng, rng.integers(0, 4, 20000), 7, dtype, distinct=False) + (7,)
    out["dup_unsorted"] = csr_from_lengths(rng, rng.integers(1, 40, 500), 64, dtype, sort_cols=False, distinct=False) + (64,)
    out["block_edge_2044"] = csr_from_lengths(rng, [2044, 2045, 2043, 1, 2048, 2049, 4096], 5000, dtype) + (5000,)
    out["long_rows_mixed"] = csr_from_lengths(rng, [3, 40000, 0, 17000, 5, 16384, 16385, 2], 50000, dtype) + (50000,)
    out["many_empty"] = csr_from_lengths(rng, np.where(rng.random(5000) < 0.9, 0, 6), 999, dtype) + (999,)
    out["rows_1_to_600"] = csr_from_lengths(rng, np.arange(1, 601), 700, dtype) + (700,)
    return out
