"""The C-ABI library loads and exports every symbol include/librosa_amd.h declares (CPU: no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "librosa_amd.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lra_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = _declared_symbols()
    for must in ("lra_ctx_create", "lra_stft_plan_create", "lra_stft_exec", "lra_spectrogram_exec", "lra_mel_plan_create", "lra_melspectrogram_exec",
                 "lra_mel_apply_exec", "lra_istft_plan_create", "lra_istft_exec", "lra_transpose", "lra_last_error"):
        assert must in names
    assert len(names) >= 30


def test_library_exports_every_declared_symbol():
    from librosa_amd import _native

    assert os.path.exists(_native.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_native.LIB_PATH)
    missing = [n for n in _declared_symbols() if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"
    # and the Python binding covers the whole header
    unbound = [n for n in _declared_symbols() if n not in _native.SIGNATURES]
    assert not unbound, f"declared in the header but not bound in _native.SIGNATURES: {unbound}"
    extra = [n for n in _native.SIGNATURES if n not in _declared_symbols()]
    assert not extra, f"bound but not declared in the header: {extra}"


def test_no_device_means_loud_failure():
    """Without a GPU every compute entry point must fail loudly (no CPU fallback)."""
    import numpy as np

    import librosa_amd as L

    if L.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(L.NativeError):
        L.stft(np.zeros(4096, np.float32))
    with pytest.raises(L.NativeError):
        L.feature.melspectrogram(y=np.zeros(4096, np.float32))
    with pytest.raises(L.NativeError):
        L.istft(np.zeros((1025, 4), np.complex64))


def test_product_never_imports_the_oracle():
    """The oracle and the host simulator are test infrastructure: nothing under librosa_amd/ may import,
    include or load them."""
    pkg = os.path.join(ROOT, "librosa_amd")
    bad = re.compile(r"^\s*(import|from)\s+(stft_oracle|ref_shim|golden_cases|hostsim_util)\b|#\s*include\s+\"[^\"]*(oracle|tests)/|_hostsim\.so|oracle/", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                code = "\n".join(line for line in txt.splitlines() if not line.lstrip().startswith(("//", "#  ", "*", '"""')))
                m = bad.search(code)
                assert m is None, (os.path.join(dirpath, f), m.group(0))
