"""Import shim that runs the UNMODIFIED reference (`/root/reference/librosa`) in this container.

TEST INFRASTRUCTURE ONLY.  Nothing under ``librosa_amd/`` may import this module.

The reference needs Python >= 3.12 (PEP 695 syntax) plus numba / lazy_loader / decorator /
soundfile / soxr / pooch, none of which exist here.  Instead of porting anything, this module
patches the *environment*:

* stubs the missing third-party modules in ``sys.modules`` (numba ``jit`` -> identity, so the two
  numba kernels on the hot path, ``core/spectrum.py:629-643`` and ``filters.py:1258-1265``, run
  as the plain NumPy code they are written in);
* installs a meta-path finder that loads ``librosa*`` from ``/root/reference`` and rewrites, in
  memory only, the PEP 695 ``type X = ...`` aliases and ``def f[T](...)`` generics that Python
  3.10 cannot parse (all affected runtime modules use ``from __future__ import annotations``).

``/root/reference`` exists only in the build container.  On the GPU box the same files are available as the
archive ``oracle/_ref/librosa_ref.zip`` (a git-ignored build product of ``oracle/make_ref.py``), which this module
unpacks into a temporary directory; only ``bench.py``'s ``cpu_baseline`` leg uses that (the reference itself timed on
the GPU box's host cores).  The ``-m gpu`` tests and ``smoke()`` use the committed fixtures under ``tests/golden/``
(made by ``oracle/make_golden.py``) and the NumPy restatement in ``oracle/stft_oracle.py``.
"""
from __future__ import annotations

import ast
import functools
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import re
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
PACKED = os.path.join(HERE, "_ref", "librosa_ref.zip")  # built by oracle/make_ref.py where /root/reference exists; travels to the GPU box


def _reference_root():
    """LIBROSA_REFERENCE_ROOT, else /root/reference, else the packed copy of the same files unpacked into a temporary directory."""
    env = os.environ.get("LIBROSA_REFERENCE_ROOT")
    if env:
        return env
    if os.path.isfile("/root/reference/librosa/__init__.py") or not os.path.isfile(PACKED):
        return "/root/reference"
    import atexit
    import shutil
    import zipfile

    d = tempfile.mkdtemp(prefix="lra_ref_")
    atexit.register(shutil.rmtree, d, ignore_errors=True)
    with zipfile.ZipFile(PACKED) as z:
        z.extractall(d)
    return d


REFERENCE_ROOT = _reference_root()


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "librosa", "__init__.py"))


# --------------------------------------------------------------------------- third-party stubs
def _identity_decorator(*dargs, **dkwargs):
    # usable bare (@jit) or called (@jit(nopython=True, cache=True))
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]

    def deco(fn):
        return fn

    return deco


def _raising_decorator(name):
    def factory(*dargs, **dkwargs):
        def deco(fn):
            @functools.wraps(fn)
            def _unavailable(*a, **k):
                raise NotImplementedError(f"numba.{name} kernel '{fn.__name__}' is stubbed in the oracle shim")

            return _unavailable

        if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
            return deco(dargs[0])
        return deco

    return factory


def _array_vectorize(*dargs, **dkwargs):
    """numba.vectorize for scalar kernels whose bodies are plain NumPy expressions (``util/utils.py:2580-2637``:
    ``_cabs2``, ``_phasor_angles``): the undecorated function already works elementwise on arrays; a trailing extra
    positional argument is the ufunc's ``out``."""

    def deco(fn):
        nin = fn.__code__.co_argcount

        @functools.wraps(fn)
        def ufunc(*args):
            res = fn(*args[:nin])
            if len(args) > nin and args[nin] is not None:
                args[nin][...] = res
                return args[nin]
            return res

        return ufunc

    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return deco(dargs[0])
    return deco


def _make_numba():
    m = types.ModuleType("numba")
    m.jit = _identity_decorator
    m.njit = _identity_decorator
    m.stencil = _raising_decorator("stencil")
    m.guvectorize = _raising_decorator("guvectorize")
    m.vectorize = _array_vectorize
    m.__version__ = "0.0-stub"
    return m


def _make_lazy_loader():
    m = types.ModuleType("lazy_loader")

    def attach_stub(package_name, filename):
        stub = os.path.splitext(filename)[0] + ".pyi"
        with open(stub, "r", encoding="utf-8") as fh:
            tree = ast.parse(fh.read())
        submodules, attrs = set(), {}
        for node in tree.body:
            if isinstance(node, ast.ImportFrom) and node.level == 1:
                if node.module is None:
                    for alias in node.names:
                        submodules.add(alias.asname or alias.name)
                else:
                    for alias in node.names:
                        attrs[alias.asname or alias.name] = (node.module, alias.name)
        all_names = sorted(submodules | set(attrs))

        def __getattr__(name):
            if name in submodules:
                return importlib.import_module(f"{package_name}.{name}")
            if name in attrs:
                mod, attr = attrs[name]
                sub = importlib.import_module(f"{package_name}.{mod}")
                val = getattr(sub, attr)
                setattr(sys.modules[package_name], name, val)
                return val
            raise AttributeError(f"No {package_name} attribute {name}")

        def __dir__():
            return all_names

        return __getattr__, __dir__, list(all_names)

    def load(name, *a, **k):
        return types.ModuleType(name)

    m.attach_stub = attach_stub
    m.load = load
    return m


def _make_decorator():
    m = types.ModuleType("decorator")

    class FunctionMaker:
        @staticmethod
        def create(func, body, evaldict, **kw):
            return evaldict["decfunc"]

    def decorator(caller):
        def deco(func):
            @functools.wraps(func)
            def wrapper(*args, **kwargs):
                return caller(func, *args, **kwargs)

            return wrapper

        return deco

    m.FunctionMaker = FunctionMaker
    m.decorator = decorator
    return m


def _make_soundfile():
    m = types.ModuleType("soundfile")
    m.SoundFile = object
    m.SEEK_END = 2
    m.LibsndfileError = RuntimeError
    return m


def _make_pooch():
    m = types.ModuleType("pooch")

    class _Registry:
        registry: dict = {}

        def load_registry(self, *a, **k):
            return None

        def fetch(self, *a, **k):
            raise RuntimeError("pooch is stubbed: no network in the oracle shim")

    m.os_cache = lambda name: os.path.join(tempfile.gettempdir(), name)
    m.create = lambda *a, **k: _Registry()
    return m


# --------------------------------------------------------------------------- PEP 695 rewriting
_RE_TYPE_ALIAS = re.compile(r"^(\s*)type\s+(\w+)(\[[^\]]*\])?\s*=", re.M)
_RE_GENERIC_DEF = re.compile(r"\bdef\s+(\w+)\[[^\]\(\)]*(\[[^\]]*\][^\]\(\)]*)*\]\(")


class _RewritingLoader(importlib.machinery.SourceFileLoader):
    def source_to_code(self, data, path, *, _optimize=-1):  # type: ignore[override]
        src = data.decode("utf-8") if isinstance(data, (bytes, bytearray)) else data
        src = _RE_TYPE_ALIAS.sub(r"\1\2 =", src)
        src = _RE_GENERIC_DEF.sub(r"def \1(", src)
        return compile(src, path, "exec", dont_inherit=True, optimize=_optimize)


class _ReferenceFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != "librosa" and not fullname.startswith("librosa."):
            return None
        rel = fullname.replace(".", os.sep)
        pkg_init = os.path.join(REFERENCE_ROOT, rel, "__init__.py")
        mod_file = os.path.join(REFERENCE_ROOT, rel + ".py")
        if os.path.isfile(pkg_init):
            loader = _RewritingLoader(fullname, pkg_init)
            return importlib.util.spec_from_file_location(
                fullname, pkg_init, loader=loader, submodule_search_locations=[os.path.dirname(pkg_init)]
            )
        if os.path.isfile(mod_file):
            loader = _RewritingLoader(fullname, mod_file)
            return importlib.util.spec_from_file_location(fullname, mod_file, loader=loader)
        return None


_installed = False


def install():
    """Install stubs + finder; afterwards ``import librosa`` yields the reference."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    os.environ.pop("LIBROSA_CACHE_DIR", None)
    for name, factory in (
        ("numba", _make_numba),
        ("lazy_loader", _make_lazy_loader),
        ("decorator", _make_decorator),
        ("soundfile", _make_soundfile),
        ("soxr", lambda: types.ModuleType("soxr")),
        ("pooch", _make_pooch),
    ):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = factory()
    sys.meta_path.insert(0, _ReferenceFinder())
    _installed = True


def load_reference():
    """Return the reference ``librosa`` module (imported through the shim)."""
    install()
    import librosa  # noqa: WPS433 - resolved by _ReferenceFinder

    assert os.path.realpath(librosa.__file__).startswith(os.path.realpath(REFERENCE_ROOT))
    return librosa
