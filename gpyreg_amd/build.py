"""Builds libgpcore.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "gpcore.hip")
OUT = os.path.join(HERE, "lib", "libgpcore.so")


def _newest_source_mtime():
    d = os.path.join(HERE, "csrc")
    files = [os.path.join(d, f) for f in os.listdir(d)]
    files.append(os.path.join(os.path.dirname(HERE), "include", "gpcore.h"))
    return max(os.path.getmtime(f) for f in files)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile gpyreg_amd/csrc/*.hip -> gpyreg_amd/lib/libgpcore.so.  Returns the path."""
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest_source_mtime():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(OUT):  # GPU box without a compiler: use the shipped build
            return OUT
        raise RuntimeError("hipcc not found and libgpcore.so is not built")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
