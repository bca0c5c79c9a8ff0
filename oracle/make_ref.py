"""Packs the UNMODIFIED reference package into ``oracle/_ref/librosa_ref.zip`` so that the reference itself -- not only the
NumPy restatement in ``stft_oracle.py`` -- can be timed (and used as the checker) on the GPU box, where ``/root/reference``
does not exist.

TEST / MEASUREMENT INFRASTRUCTURE ONLY.  ``oracle/_ref/`` is git-ignored (nothing of the reference enters the history) but is
not gpurun-ignored: like the built ``.so`` files it travels with the repository snapshot.  The archive is a build product of
``__graft_entry__.build()`` in the container that has ``/root/reference``; ``oracle/ref_shim.py`` unpacks it into a temporary
directory when neither ``LIBROSA_REFERENCE_ROOT`` nor ``/root/reference`` is available, and ``bench.py``'s ``cpu_baseline`` leg
then reports ``"kind": "reference"`` (librosa/core/spectrum.py:57-391, librosa/feature/spectral.py:2022-2161 timed as they are).
Nothing under ``librosa_amd/`` may touch it (tests/test_abi.py::test_product_never_imports_the_oracle).

    python oracle/make_ref.py            # /root/reference -> oracle/_ref/librosa_ref.zip (+ MANIFEST.json with sha256 per file)
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("LIBROSA_REFERENCE_ROOT", "/root/reference")
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "librosa_ref.zip")
KEEP = (".py", ".pyi")
EXTRA = ("util/example_data/registry.txt", "util/example_data/index.json", "core/intervals.msgpack")  # data files read at import time (util/files.py, core/intervals.py)


def wanted():
    root = os.path.join(SRC, "librosa")
    for dirpath, dirnames, files in os.walk(root):
        dirnames[:] = sorted(d for d in dirnames if d != "__pycache__")
        for f in sorted(files):
            rel = os.path.relpath(os.path.join(dirpath, f), SRC)
            if f.endswith(KEEP) or rel.replace(os.sep, "/").split("librosa/", 1)[-1] in EXTRA:
                yield rel


def make(force=False):
    """Returns the archive's path, or None when there is no reference tree to pack."""
    if not os.path.isfile(os.path.join(SRC, "librosa", "__init__.py")):
        return OUT if os.path.isfile(OUT) else None
    files = list(wanted())
    newest = max(os.path.getmtime(os.path.join(SRC, f)) for f in files)
    if not force and os.path.isfile(OUT) and os.path.getmtime(OUT) >= newest:
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    manifest = {}
    with zipfile.ZipFile(OUT, "w", zipfile.ZIP_DEFLATED) as z:
        for rel in files:
            data = open(os.path.join(SRC, rel), "rb").read()
            manifest[rel.replace(os.sep, "/")] = hashlib.sha256(data).hexdigest()
            z.writestr(zipfile.ZipInfo(rel.replace(os.sep, "/"), date_time=(2020, 1, 1, 0, 0, 0)), data, zipfile.ZIP_DEFLATED)
    json.dump({"source": SRC, "files": manifest}, open(os.path.join(OUT_DIR, "MANIFEST.json"), "w"), indent=0, sort_keys=True)
    return OUT


if __name__ == "__main__":
    p = make(force="--force" in sys.argv)
    print(p if p else f"no reference tree under {SRC}: nothing packed")
