"""Builds the native artefacts in-tree with make (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(PKG_DIR, "lib")


def lib_path(name="libhip-spmv.so"):
    return os.path.join(LIB_DIR, name)


def _make(directory, *targets):
    cmd = ["make", "-C", directory, *targets]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("%s failed:\n%s" % (" ".join(cmd), proc.stdout))
    return proc.stdout


def build_all(verbose=False):
    """Compile libhip-spmv.so (HIP, gfx950) and the host-side drivers."""
    out = _make(os.path.join(PKG_DIR, "csrc"), "all", "debug")
    out += _make(os.path.join(PKG_DIR, "host"))
    if verbose:
        print(out)
    for f in ("libhip-spmv.so", "libhip-spmv-debug.so", "libnpbcg.so", "npbcg", "libspmvcallers.so", "spmvdrv"):
        if not os.path.exists(lib_path(f)):
            raise RuntimeError("build did not produce " + lib_path(f))
    return LIB_DIR
