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
import time
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


def _deps_fresh(obj, cmd_line):
    """True when `obj` is newer than everything its depfile (hipcc -MD) lists and was built by the same command line."""
    dep, stamp = obj + ".d", obj + ".cmd"
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(stamp)):
        return False
    if open(stamp).read() != cmd_line:
        return False
    t = os.path.getmtime(obj)
    text = open(dep).read().replace("\\\n", " ")
    files = text.split(":", 1)[1].split() if ":" in text else []
    return bool(files) and all(os.path.exists(f) and os.path.getmtime(f) <= t for f in files)


def build(force=False, verbose=True, extra_flags=(), out=OUT):
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    os.makedirs(OBJ, exist_ok=True)
    tag = "_" + str(abs(hash(tuple(extra_flags))) % 10**8) if extra_flags else ""
    jobs = [([hipcc, *FLAGS, *extra_flags, "-c", os.path.join(CSRC, "lra_api.hip"), "-o", os.path.join(OBJ, f"api{tag}.o")])]
    jobs.append([hipcc, *FLAGS, *extra_flags, "-c", os.path.join(CSRC, "lra_mixed_inst.hip"), "-o", os.path.join(OBJ, f"mixed{tag}.o")])
    for g in range(n_groups()):
        jobs.append([hipcc, *FLAGS, *extra_flags, f"-DLRA_INST_GROUP={g}", "-c", os.path.join(CSRC, "lra_inst.hip"), "-o", os.path.join(OBJ, f"inst{tag}_{g}.o")])

    def run(cmd):
        obj = cmd[-1]
        line = " ".join(cmd)
        compiling = "-c" in cmd
        if compiling:
            # one object per translation unit, rebuilt only when one of the files it actually includes changed (depfile) -- a change
            # to lra_api.hip or to a header only it includes costs one compile, not the eleven instance groups
            if not force and _deps_fresh(obj, line):
                return obj
            cmd = [*cmd, "-MD", "-MF", obj + ".d"]
        if verbose:
            print(line, flush=True)
        t0 = time.time()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + line + "\n" + r.stdout[-4000:])
        if compiling:
            open(obj + ".cmd", "w").write(line)
            os.utime(obj, (t0, t0))  # stamped with the compile's START: a source edited while it ran is newer than the object
        return obj

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
        objs = list(pool.map(run, jobs))
    # (freshness is judged per object from its depfile: a library linked while a source was being edited is older than that edit's object)
    if force or not os.path.exists(out) or any(os.path.getmtime(o) > os.path.getmtime(out) for o in objs):
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, f"-L{ROCM}/lib", "-lrocfft", f"-Wl,-rpath,{ROCM}/lib"]
        run(link)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
