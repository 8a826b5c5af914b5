#!/usr/bin/env python3
"""tests/golden/make_sb_fixture.py -- writes tests/golden/sparsebench/crsmat010u with the REFERENCE's own generator.

Container-only: imports /root/reference/SparseBench/big_gen.py (stdlib-only), seeds Python's `random` (the
generator is unseeded, big_gen.py:3,64,74,80) and calls its write_random(filename, 10): the size-10 matrix of
SparseBench/run_all's series (n = 10^3 rows).  Only the data file travels; nothing of the reference's source is kept.

What the file pins (tests/test_callers.py): the text format of big_gen.py:52-57 as the Fortran reader takes it
(SRC/reference/gen_crs.f:757-789), including Quirk N1 of SURVEY 8(a) a8: random_crs appends the diagonal AFTER row_ptr
was built (big_gen.py:64-68 vs 74-76), so the file holds about nnz + n entry lines while header and row_ptr describe
nnz -- the reader consumes the first nnz and ignores the rest.
"""
import os
import random
import sys

REF = "/root/reference/SparseBench"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    if not os.path.isdir(REF):
        raise SystemExit("the reference checkout is not here: the committed fixture stays as it is")
    sys.path.insert(0, REF)
    import big_gen
    random.seed(10)
    out = os.path.join(HERE, "sparsebench", "crsmat010u")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    big_gen.write_random(out, 10)
    lines = open(out).read().splitlines()
    n, nnz = int(lines[0][:12]), int(lines[0][12:24])
    print("wrote %s: n=%d header nnz=%d, %d entry lines (%d surplus)" % (out, n, nnz, len(lines) - 1 - (n + 1),
                                                                        len(lines) - 1 - (n + 1) - nnz))


if __name__ == "__main__":
    main()
