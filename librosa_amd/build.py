"""In-tree build of the gfx950 library: ``python -m librosa_amd.build``.

One hipcc invocation; the result ``librosa_amd/_liblibrosa_amd.so`` is git-ignored but travels to
the GPU box with the repo snapshot.  hipcc cross-compiles for gfx950 without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_liblibrosa_amd.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(os.path.dirname(HERE), "include", "librosa_amd.h")]


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force=False, verbose=True):
    if not force and not is_stale():
        return OUT
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", OUT, os.path.join(CSRC, "lra_api.hip"),
           f"-L{ROCM}/lib", "-lrocfft", f"-Wl,-rpath,{ROCM}/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
