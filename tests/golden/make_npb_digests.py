#!/usr/bin/env python3
"""tests/golden/make_npb_digests.py -- rewrites tests/golden/npb_digests.json: SHA-256 of rowstr / colidx / a of NPB
classes S, W, A as our generator (lilac-benchmarks_amd/host/npb_cg.c) produces them.  Run it only when the generator is
changed on purpose; tests/test_npb.py::test_generated_matrices_match_the_committed_digests then guards against
accidental changes (compiler, libm)."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pkgload  # noqa: E402


def digests(pkg, cls):
    m = pkg.npb.NpbMatrix(cls)
    h = lambda v: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()  # noqa: E731
    return {"na": int(m.na), "nnz": int(m.nnz), "rowstr": h(m.rowstr), "colidx": h(m.colidx), "a": h(m.a)}


if __name__ == "__main__":
    pkg = pkgload.load_package()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "npb_digests.json")
    doc = json.load(open(path))
    doc["classes"] = {c: digests(pkg, c) for c in "SWA"}
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(doc["classes"], indent=1))
