#!/usr/bin/env python3
"""Per-kernel resources of the gfx950 code objects inside a HIP shared library: VGPRs, AGPRs, SGPRs, spilled registers,
scratch bytes, LDS bytes -- from the code objects' own metadata notes (clang offload bundles in .hip_fatbin, read with
llvm-readelf).  No GPU needed.
    python tools/kernel_resources.py lilac-benchmarks_amd/lib/libhip-spmv.so [name fragment]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    data = open(path, "rb").read()
    for m in re.finditer(MAGIC, data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + len(MAGIC))
        off = base + len(MAGIC) + 8
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24:off + 24 + tl].decode()
            off += 24 + tl
            if "gfx" in triple and size:
                yield triple, data[base + o:base + o + size]


def kernels(path):
    """[{name, symbol, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds}] over every gfx code object of the library"""
    import yaml
    keys = {"vgpr_count": "vgpr", "agpr_count": "agpr", "sgpr_count": "sgpr", "vgpr_spill_count": "vgpr_spill",
            "sgpr_spill_count": "sgpr_spill", "private_segment_fixed_size": "scratch", "group_segment_fixed_size": "lds",
            "name": "name", "symbol": "symbol"}
    out = []
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], stdout=subprocess.PIPE, text=True, check=True).stdout
        m = re.search(r"^\s*---\n(.*?)^\s*\.\.\.", txt, re.S | re.M)
        if not m:
            continue
        meta = yaml.safe_load(m.group(1))
        for k in meta.get("amdhsa.kernels", []):
            out.append({keys[a[1:]]: v for a, v in k.items() if a[1:] in keys})
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True)
    return p.stdout.splitlines() if p.returncode == 0 else names


def main():
    ks = kernels(sys.argv[1])
    frag = sys.argv[2] if len(sys.argv) > 2 else ""
    names = demangle([k.get("symbol", k.get("name", "?")).replace(".kd", "") for k in ks])
    print("%5s %5s %5s %6s %6s %8s %7s  kernel" % ("vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds"))
    for k, n in sorted(zip(ks, names), key=lambda t: t[1]):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*", "", n)
        if frag in n:
            print("%5d %5d %5d %6d %6d %8d %7d  %s" % (k.get("vgpr", -1), k.get("agpr", 0), k.get("sgpr", -1), k.get("vgpr_spill", 0),
                                                      k.get("sgpr_spill", 0), k.get("scratch", 0), k.get("lds", 0), n))


if __name__ == "__main__":
    main()
