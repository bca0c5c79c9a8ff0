"""In-tree build of the gfx950 library: ``python -m librosa_amd.build``.

The library holds several hundred instances of the fused kernels; hipcc compiles one translation unit on
one core, so the instances are split into groups (``csrc/lra_fused.h``) that are compiled side by side:
``lra_inst.hip`` once per group plus ``lra_api.hip`` (host API, general rocFFT path), then one link.  The
result ``librosa_amd/_liblibrosa_amd.so`` is git-ignored but travels to the GPU box with the repo snapshot.
hipcc cross-compiles for gfx950 without a GPU.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "_liblibrosa_amd.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".hip"))] + [os.path.join(os.path.dirname(HERE), "include", "librosa_amd.h")]


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(s) > t for s in sources())


def n_groups():
    text = open(os.path.join(CSRC, "lra_fused.h")).read()
    return int(re.search(r"#define LRA_INST_NUM_GROUPS (\d+)", text).group(1))


def build(force=False, verbose=True, extra_flags=(), out=OUT):
    if not force and out == OUT and not is_stale():
        return out
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    os.makedirs(OBJ, exist_ok=True)
    tag = "_" + str(abs(hash(tuple(extra_flags))) % 10**8) if extra_flags else ""
    jobs = [([hipcc, *FLAGS, *extra_flags, "-c", os.path.join(CSRC, "lra_api.hip"), "-o", os.path.join(OBJ, f"api{tag}.o")])]
    for g in range(n_groups()):
        jobs.append([hipcc, *FLAGS, *extra_flags, f"-DLRA_INST_GROUP={g}", "-c", os.path.join(CSRC, "lra_inst.hip"), "-o", os.path.join(OBJ, f"inst{tag}_{g}.o")])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout[-4000:])
        return cmd[-1]

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
        objs = list(pool.map(run, jobs))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, f"-L{ROCM}/lib", "-lrocfft", f"-Wl,-rpath,{ROCM}/lib"]
    run(link)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
