"""Builds libgpcore.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

`python -m gpyreg_amd.build` builds the product library, `python -m gpyreg_amd.build --experiments` the experiments
build lib/libgpcore_exp.so (-DGPC_EXPERIMENTS: the product plus the schedules that were measured and rejected --
dataflow graph, independent pipelines, rectangular tiles, right-looking panels; tests/ and tools/ opt into it through
GPYREG_AMD_LIB=<path>, the product never loads it)."""

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "gpcore.hip")
OUT = os.path.join(HERE, "lib", "libgpcore.so")
OUT_EXPERIMENTS = os.path.join(HERE, "lib", "libgpcore_exp.so")


def _newest_source_mtime():
    d = os.path.join(HERE, "csrc")
    files = [os.path.join(d, f) for f in os.listdir(d)]
    files.append(os.path.join(os.path.dirname(HERE), "include", "gpcore.h"))
    return max(os.path.getmtime(f) for f in files)


def build(force: bool = False, verbose: bool = False, experiments: bool = False) -> str:
    """Compile gpyreg_amd/csrc/*.hip -> gpyreg_amd/lib/libgpcore.so (or libgpcore_exp.so).  Returns the path."""
    out = OUT_EXPERIMENTS if experiments else OUT
    if not force and os.path.exists(out) and os.path.getmtime(out) >= _newest_source_mtime():
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(out):  # GPU box without a compiler: use the shipped build
            return out
        raise RuntimeError("hipcc not found and %s is not built" % os.path.basename(out))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", out, SRC]
    if experiments:
        cmd.insert(1, "-DGPC_EXPERIMENTS")
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True, experiments="--experiments" in sys.argv[1:]))
