"""ctypes binding of ``include/librosa_amd.h`` (the gfx950 library ``_liblibrosa_amd.so``).

There is NO CPU fallback: if the shared library is missing or no HIP device is present, every
compute entry point raises :class:`NativeError` (a ``LibrosaError``).  The library is built
in-tree by ``__graft_entry__.build()`` / ``python -m librosa_amd.build``.
"""
from __future__ import annotations

import collections
import ctypes
import os
import threading
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

import numpy as np

from .util.exceptions import LibrosaError, ParameterError

_HERE = os.path.dirname(os.path.abspath(__file__))
# LIBROSA_AMD_LIBRARY: load another build of the same ABI (kernel experiments, scripts/gpu_probe.py)
LIB_PATH = os.environ.get("LIBROSA_AMD_LIBRARY") or os.path.join(_HERE, "_liblibrosa_amd.so")

LRA_OK, LRA_EINVAL, LRA_EHIP, LRA_EROCFFT, LRA_ENODEV, LRA_ENOMEM = 0, -1, -2, -3, -4, -5
LRA_F32, LRA_F64 = 0, 1
PAD_MODES = {"constant": 0, "reflect": 1, "edge": 2, "symmetric": 3}


class NativeError(LibrosaError):
    """The HIP library is missing, or a HIP / rocFFT call failed."""


# name -> (restype, argtypes); every symbol declared in include/librosa_amd.h
SIGNATURES = {
    "lra_last_error": (c_char_p, []),
    "lra_version": (c_char_p, []),
    "lra_device_count": (c_int, [POINTER(c_int)]),
    "lra_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "lra_ctx_destroy": (None, [c_void_p]),
    "lra_ctx_set_stream": (c_int, [c_void_p, c_void_p]),
    "lra_ctx_side": (c_int, [c_void_p, c_int]),
    "lra_ctx_use_own_stream": (c_int, [c_void_p]),
    "lra_ctx_sync": (c_int, [c_void_p]),
    "lra_ctx_set_option": (c_int, [c_void_p, c_char_p, c_int]),
    "lra_ctx_device_name": (c_int, [c_void_p, c_char_p, c_size_t]),
    "lra_ctx_nonfinite_reset": (c_int, [c_void_p]),
    "lra_ctx_nonfinite_read": (c_int, [c_void_p, POINTER(c_int)]),
    "lra_stft_plan_is_fused": (c_int, [c_void_p]),
    "lra_stft_plan_tuned_variant": (c_int, [c_void_p, c_int]),
    "lra_istft_plan_tuned_variant": (c_int, [c_void_p]),
    "lra_malloc": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "lra_free": (c_int, [c_void_p, c_void_p]),
    "lra_malloc_placed": (c_int, [c_void_p, c_size_t, c_int, c_int64, c_int, POINTER(c_void_p), POINTER(c_float), POINTER(c_int)]),
    "lra_free_placed": (c_int, [c_void_p, c_void_p]),
    "lra_memset": (c_int, [c_void_p, c_void_p, c_int, c_size_t]),
    "lra_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "lra_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "lra_event_create": (c_int, [c_void_p, POINTER(c_void_p)]),
    "lra_event_destroy": (None, [c_void_p]),
    "lra_event_record": (c_int, [c_void_p]),
    "lra_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "lra_stft_plan_create": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, POINTER(c_void_p)]),
    "lra_stft_plan_destroy": (None, [c_void_p]),
    "lra_stft_num_frames": (c_int, [c_void_p, c_int64, POINTER(c_int64)]),
    "lra_stft_exec": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "lra_spectrogram_exec": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p]),
    "lra_mel_plan_create": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, POINTER(c_void_p)]),
    "lra_mel_plan_destroy": (None, [c_void_p]),
    "lra_melspectrogram_exec": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p]),
    "lra_mel_apply_exec": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "lra_istft_plan_create": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "lra_istft_plan_destroy": (None, [c_void_p]),
    "lra_istft_exec": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64]),
    "lra_istft_exec_norm": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64]),
    "lra_transpose": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int]),
    "lra_probe_stream": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int64, c_int, c_int]),
    "lra_cqt_recursion_exec": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_int, c_void_p, c_int64, c_int, c_int]),
    "lra_probe_stream_window": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int64, c_int, c_int]),
    "lra_pcg64_random_exec": (c_int, [c_void_p, POINTER(ctypes.c_uint64), ctypes.c_uint64, c_void_p, c_int64]),
    "lra_griffinlim_init_pcg64": (c_int, [c_void_p, POINTER(ctypes.c_uint64), c_void_p, c_void_p, c_int64, c_int, c_int64, c_int]),
    "lra_probe_stream_pitched": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int64, c_int, c_int, c_int64, c_int]),
    "lra_stft_exec_strided": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p, c_int64]),
    "lra_item_absmax_exec": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "lra_item_max_exec": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "lra_to_db_exec": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_double, c_double, c_void_p, c_void_p, c_int, c_double]),
    "lra_from_db_exec": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_double]),
    "lra_istft_exec_host": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64]),
    "lra_stft_exec_host": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p, c_int64, POINTER(c_int)]),
    "lra_comm_unique_id": (c_int, [c_void_p]),
    "lra_comm_init": (c_int, [c_void_p, c_int, c_int, c_void_p, POINTER(c_void_p)]),
    "lra_comm_allgather": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "lra_comm_allgatherv": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_size_t), POINTER(c_size_t)]),
    "lra_comm_destroy": (None, [c_void_p]),
    "lra_phase_vocoder_exec": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_int]),
    "lra_griffinlim_init": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int]),
    "lra_griffinlim_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_double, c_double, c_int]),
    "lra_pcen_exec": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_double, c_double, c_double, c_double, c_double, c_void_p, c_double, c_void_p]),
    "lra_maxfilter_exec": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_int]),
    "lra_fir_decimate_exec": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_double, c_double, c_int]),
    "lra_resample_poly_exec": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_int]),
    "lra_resample_fft_exec": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_double, c_int]),
    "lra_resample_band_exec": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_double, c_double, c_double, c_int]),
    "lra_cqt_project_exec": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int64, c_int, c_int, c_int, c_int, c_int]),
    "lra_cqt_octave_supported": (c_int, [c_int]),
    "lra_cqt_octave_exec": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int]),
    "lra_magnitude_exec": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int]),
    "lra_magphase_exec": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_double, c_int]),
    "lra_hpss_exec": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_double, c_double, c_double, c_int, c_int]),
    "lra_dct_exec": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_int, c_double, c_double, c_void_p, c_void_p, c_int, c_double]),
}

_lib = None
_lib_lock = threading.Lock()


def load_library():
    """dlopen the in-tree HIP library and attach signatures; raises NativeError when absent."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise NativeError(
                    f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(librosa_amd has no CPU fallback)"
                )
            # PyTorch wheels bundle their own ROCm runtime (libamdhip64.so.7, libhsa-runtime64, librocfft
            # ... same SONAMEs as /opt/rocm).  Two different HIP runtimes in one process cannot both
            # own the device, so if torch is installed its runtime must be the one that is loaded first;
            # our library then binds to it through the shared SONAMEs.  Without torch the system ROCm
            # runtime is used.
            try:
                import torch  # noqa: F401
            except Exception:
                pass
            try:
                lib = ctypes.CDLL(LIB_PATH)
            except OSError as exc:  # missing libamdhip64 / librocfft
                raise NativeError(f"cannot load {LIB_PATH}: {exc}") from exc
            for name, (restype, argtypes) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = restype
                fn.argtypes = argtypes
            _lib = lib
    return _lib


def _check(rc):
    if rc == LRA_OK:
        return
    msg = load_library().lra_last_error().decode("utf-8", "replace")
    if rc == LRA_EINVAL:
        raise ParameterError(msg)
    raise NativeError(f"librosa_amd native error {rc}: {msg}")


def device_count():
    n = c_int(0)
    lib = load_library()
    rc = lib.lra_device_count(byref(n))
    return n.value if rc == LRA_OK else 0


def dtype_code(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return LRA_F32
    if dtype == np.float64:
        return LRA_F64
    raise ParameterError(f"unsupported real dtype {dtype}")


class DeviceBuffer:
    """Owning device allocation made through lra_malloc (NumPy path)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = c_void_p()
        _check(ctx.lib.lra_malloc(ctx.handle, self.nbytes, byref(p)))
        self.ptr = p.value or 0

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        if arr.nbytes:
            _check(self.ctx.lib.lra_memcpy_h2d(self.ctx.handle, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        if out.nbytes:
            _check(self.ctx.lib.lra_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.lra_free(self.ctx.handle, self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Event:
    def __init__(self, ctx):
        self.ctx = ctx
        h = c_void_p()
        _check(ctx.lib.lra_event_create(ctx.handle, byref(h)))
        self.handle = h

    def record(self):
        _check(self.ctx.lib.lra_event_record(self.handle))
        return self

    def elapsed_ms(self, stop):
        ms = c_float(0)
        _check(self.ctx.lib.lra_event_elapsed_ms(self.handle, stop.handle, byref(ms)))
        return ms.value

    def __del__(self):
        try:
            self.ctx.lib.lra_event_destroy(self.handle)
        except Exception:
            pass


class CqtOctave(ctypes.Structure):
    """``lra_cqt_octave`` (``include/librosa_amd.h``)."""
    _fields_ = [("n_fft", c_int), ("hop", c_int), ("bin0", c_int), ("row0", c_int), ("n_rows", c_int), ("halve", c_int), ("n", c_int64), ("row_ptr", c_void_p), ("col", c_void_p),
                ("val", c_void_p)]


class Context:
    """One device + stream + plan caches.  ``get_context(device)`` returns the shared instance."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = c_void_p()
        _check(self.lib.lra_ctx_create(int(device), byref(h)))
        self.handle = h
        self.device = int(device)
        # least-recently-used caches, bounded: a plan owns device tables (and, on the rocFFT path, scratch buffers)
        self._stft_plans = collections.OrderedDict()
        self._istft_plans = collections.OrderedDict()
        self._mel_plans = collections.OrderedDict()
        self._mel_by_id = {}
        self._tables = {}  # pool name -> OrderedDict (see device_table)
        self._lock = threading.RLock()
        self.call_lock = threading.RLock()  # held by _arrays.Session for the duration of one public call
        # placement-aware result buffers (lra_malloc_placed): `placement_retry` candidates per new buffer (0 = off; env LRA_PLACEMENT_RETRY), and the
        # buffers whose tensors have been dropped, kept for the next call of the same shape (their placement is already known to be a good one)
        # (default 4: on a box with the placement lottery the worst of five such buffers ran the complex STFT within 0.3 % of the best, 15 % between torch.empty ones;
        #  a box without it stops after two candidates, ~12 ms per new shape: profiles/r06b_bench_line.json)
        self.placement_retry = max(0, min(8, int(os.environ.get("LRA_PLACEMENT_RETRY", "4") or 0)))
        self._placed_free = {}   # (nbytes, row_bytes) -> [ptr]
        self._placed_log = []    # (nbytes, row_bytes, probe_ms, candidates tried) of every fresh allocation

    # -- context ------------------------------------------------------------------------------
    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        _check(self.lib.lra_ctx_device_name(self.handle, buf, 256))
        return buf.value.decode()

    SIDE_FORK, SIDE_BACK, SIDE_JOIN, SIDE_END = 1, 2, 3, 4

    def side(self, mode):
        """A second stream of the context (``lra_ctx_side``): fork / back / join / end (= back if forked, then join)."""
        _check(self.lib.lra_ctx_side(self.handle, int(mode)))

    def set_stream(self, stream_ptr):
        """Enqueue on this hipStream_t (0 / None = HIP's default stream, torch's usual current stream)."""
        _check(self.lib.lra_ctx_set_stream(self.handle, c_void_p(stream_ptr or None)))

    def use_own_stream(self):
        _check(self.lib.lra_ctx_use_own_stream(self.handle))

    def sync(self):
        _check(self.lib.lra_ctx_sync(self.handle))

    def set_option(self, key, value):
        _check(self.lib.lra_ctx_set_option(self.handle, key.encode(), int(value)))
        if key == "placement_retry":
            self.placement_retry = max(0, min(8, int(value)))

    # -- placement-aware result buffers (include/librosa_amd.h, lra_malloc_placed) -----------------
    PLACED_MIN_BYTES = 256 << 20
    PLACED_KEEP_PER_SHAPE = 2
    PLACED_KEEP_BYTES = 12 << 30   # all recycled buffers together (variable-length inputs make a new shape per call: the oldest shapes go first)

    def placed_take(self, nbytes, row_bytes, rows_per_item=0):
        """Device pointer of a buffer of ``nbytes`` (rows of ``row_bytes``, ``rows_per_item`` of them per clip): a recycled one of that shape if there is
        one, else the best of ``placement_retry`` fresh candidates."""
        key = (int(nbytes), int(row_bytes), int(rows_per_item))
        with self._lock:
            free = self._placed_free.get(key)
            if free:
                return free.pop()
        p, ms, tried = c_void_p(), c_float(0), c_int(0)
        _check(self.lib.lra_malloc_placed(self.handle, key[0], key[1], key[2], int(self.placement_retry), byref(p), byref(ms), byref(tried)))
        with self._lock:
            self._placed_log.append((key[0], key[1], float(ms.value), int(tried.value)))
            del self._placed_log[:-64]
        return p.value

    def placed_give(self, nbytes, row_bytes, ptr, rows_per_item=0):
        """A placed buffer whose tensor is gone: kept for the next result of that shape, or released when enough are waiting."""
        key = (int(nbytes), int(row_bytes), int(rows_per_item))
        drop = []
        with self._lock:
            free = self._placed_free.pop(key, [])   # (re-inserted below: dict order = least recently returned shape first)
            if len(free) < self.PLACED_KEEP_PER_SHAPE and key[0] <= self.PLACED_KEEP_BYTES:
                free.append(ptr)
            else:
                drop.append(ptr)
            if free:
                self._placed_free[key] = free
            held = sum(k[0] * len(v) for k, v in self._placed_free.items())
            for k in list(self._placed_free):
                if held <= self.PLACED_KEEP_BYTES:
                    break
                if k == key and len(self._placed_free) > 1:
                    continue
                v = self._placed_free[k]
                while v and held > self.PLACED_KEEP_BYTES:
                    drop.append(v.pop(0))
                    held -= k[0]
                if not v:
                    del self._placed_free[k]
        for p in drop:
            try:
                self.lib.lra_free_placed(self.handle, c_void_p(p))
            except Exception:  # pragma: no cover - interpreter shutdown
                pass

    def placed_release_all(self):
        with self._lock:
            ptrs = [p for v in self._placed_free.values() for p in v]
            self._placed_free.clear()
        for p in ptrs:
            _check(self.lib.lra_free_placed(self.handle, c_void_p(p)))

    def nonfinite_reset(self):
        _check(self.lib.lra_ctx_nonfinite_reset(self.handle))

    def nonfinite_read(self):
        f = c_int(0)
        _check(self.lib.lra_ctx_nonfinite_read(self.handle, byref(f)))
        return bool(f.value)

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def event(self):
        return Event(self)

    PLAN_CACHE_SIZE = 48
    # device tables live in two LRU pools, each bounded by entries AND bytes: "small" (DCT bases, lifters, CQT bases, decimator taps: a few
    # KB each) and "large" (istft / griffinlim window sum-square envelopes: one float per output sample, a new key per (n_frames, length)).
    # Separate pools so that a stream of variable-length inverse transforms can neither pin gigabytes of HBM nor push the small tables out.
    TABLE_POOLS = {"small": (512, 64 << 20), "large": (8, 128 << 20)}

    def device_table(self, key, build, pool="small"):
        """Device pointer of a read-only table kept under ``key`` (LRU, bounded per pool by entry count and by bytes); ``build()`` returns the
        host array on a miss.  The upload is synchronous, so the table is usable from any stream of the device afterwards.  A table larger than
        a quarter of its pool's byte budget is not worth caching: callers check ``table_cacheable`` and upload per call instead."""
        max_entries, max_bytes = self.TABLE_POOLS[pool]
        with self._lock:
            tables = self._tables.setdefault(pool, collections.OrderedDict())
            buf = tables.get(key)
            if buf is not None:
                tables.move_to_end(key)
                return buf.ptr
            host = np.ascontiguousarray(build())
            buf = DeviceBuffer(self, max(host.nbytes, 16)).upload(host)
            tables[key] = buf
            while len(tables) > 1 and (len(tables) > max_entries or sum(b.nbytes for b in tables.values()) > max_bytes):
                _, old = tables.popitem(last=False)
                old.free()
            return buf.ptr

    def table_cacheable(self, nbytes, pool="small"):
        return int(nbytes) <= self.TABLE_POOLS[pool][1] // 4

    def table_bytes(self, pool=None):
        with self._lock:
            return sum(b.nbytes for name, t in self._tables.items() if pool in (None, name) for b in t.values())

    def _cached_plan(self, cache, key, create, destroy):
        """LRU lookup; on overflow the oldest plan is destroyed (hipFree synchronises with the device first)."""
        with self._lock:
            plan = cache.get(key)
            if plan is not None:
                cache.move_to_end(key)
                return plan
            plan = create()
            cache[key] = plan
            while len(cache) > self.PLAN_CACHE_SIZE:
                _, old = cache.popitem(last=False)
                destroy(old)
            return plan

    # -- plans (cached; keyed by the bytes of the host tables so any window spec works) -----------
    def stft_plan(self, n_fft, hop, window, center, pad_mode, dtype):
        window = np.ascontiguousarray(window, dtype=dtype)
        key = (int(n_fft), int(hop), window.tobytes(), bool(center), pad_mode, np.dtype(dtype).str)

        def create():
            h = c_void_p()
            _check(self.lib.lra_stft_plan_create(self.handle, int(n_fft), int(hop), window.ctypes.data, int(bool(center)), PAD_MODES[pad_mode], dtype_code(dtype), byref(h)))
            return h

        return self._cached_plan(self._stft_plans, key, create, self.lib.lra_stft_plan_destroy)

    def istft_plan(self, n_fft, hop, window, center, dtype):
        window = np.ascontiguousarray(window, dtype=dtype)
        key = (int(n_fft), int(hop), window.tobytes(), bool(center), np.dtype(dtype).str)

        def create():
            h = c_void_p()
            _check(self.lib.lra_istft_plan_create(self.handle, int(n_fft), int(hop), window.ctypes.data, int(bool(center)), dtype_code(dtype), byref(h)))
            return h

        return self._cached_plan(self._istft_plans, key, create, self.lib.lra_istft_plan_destroy)

    def mel_plan(self, basis):
        # Read-only arrays (what filters.mel_cached hands out) are recognised by identity: hashing the 525 KB of a 128 x 1025
        # basis on every call cost more host time than the kernel it configures takes to run.
        fast = isinstance(basis, np.ndarray) and not basis.flags.writeable and basis.flags.c_contiguous
        if fast:
            hit = self._mel_by_id.get(id(basis))
            if hit is not None and hit[0] is basis:
                plan = self._mel_plans.get(hit[1])
                if plan is not None:
                    return plan
        basis = np.ascontiguousarray(basis)
        key = (basis.shape, basis.dtype.str, basis.tobytes())
        if fast:
            if len(self._mel_by_id) > 4 * self.PLAN_CACHE_SIZE:
                self._mel_by_id.clear()
            self._mel_by_id[id(basis)] = (basis, key)

        def create():
            h = c_void_p()
            _check(self.lib.lra_mel_plan_create(self.handle, int(basis.shape[0]), int(basis.shape[1]), basis.ctypes.data, dtype_code(basis.dtype), byref(h)))
            return h

        return self._cached_plan(self._mel_plans, key, create, self.lib.lra_mel_plan_destroy)

    # -- execution ----------------------------------------------------------------------------------
    def stft_num_frames(self, plan, n):
        out = c_int64(0)
        _check(self.lib.lra_stft_num_frames(plan, int(n), byref(out)))
        return out.value

    def stft_is_fused(self, plan):
        return bool(self.lib.lra_stft_plan_is_fused(plan))

    def tuned_variant(self, plan, mode=None):
        """Kernel variant an stft plan (mode 0/1/2) or istft plan (mode None) settled on; -1 = not measured."""
        if mode is None:
            return int(self.lib.lra_istft_plan_tuned_variant(plan))
        return int(self.lib.lra_stft_plan_tuned_variant(plan, int(mode)))

    def stft_exec(self, plan, y_ptr, batch, n, y_stride, out_ptr):
        _check(self.lib.lra_stft_exec(plan, c_void_p(y_ptr), batch, n, y_stride, c_void_p(out_ptr)))

    def spectrogram_exec(self, plan, y_ptr, batch, n, y_stride, power, out_ptr):
        _check(self.lib.lra_spectrogram_exec(plan, c_void_p(y_ptr), batch, n, y_stride, float(power), c_void_p(out_ptr)))

    def stft_exec_strided(self, plan, kind, y_ptr, batch, n, y_stride, power, out_ptr, out_frame_stride):
        """kind 0 = complex spectrum, 1 = ``|X|**power``, rows ``out_frame_stride`` elements apart (fused power-of-two plans; ``include/librosa_amd.h``)."""
        _check(self.lib.lra_stft_exec_strided(plan, int(kind), c_void_p(y_ptr), batch, n, y_stride, float(power), c_void_p(out_ptr), int(out_frame_stride)))

    def melspectrogram_exec(self, plan, mel_plan, y_ptr, batch, n, y_stride, power, out_ptr):
        _check(self.lib.lra_melspectrogram_exec(plan, mel_plan, c_void_p(y_ptr), batch, n, y_stride, float(power), c_void_p(out_ptr)))

    def stft_exec_host(self, plan, mel_plan, kind, y_host_ptr, batch, n, y_stride, power, out_host_ptr, out_item_stride=0):
        """Host buffers in, host buffers out (chunked, overlapped staging); returns the non-finite flag."""
        flag = c_int(0)
        _check(self.lib.lra_stft_exec_host(plan, mel_plan, int(kind), c_void_p(y_host_ptr), batch, n, y_stride, float(power), c_void_p(out_host_ptr), int(out_item_stride), ctypes.byref(flag)))
        return bool(flag.value)

    def istft_exec_host(self, plan, d_host_ptr, batch, n_frames, n_used, wss_host_ptr, y_host_ptr, out_len, y_stride):
        _check(self.lib.lra_istft_exec_host(plan, c_void_p(d_host_ptr), batch, n_frames, n_used, c_void_p(wss_host_ptr), c_void_p(y_host_ptr), out_len, y_stride))

    def mel_apply_exec(self, mel_plan, s_ptr, batch, n_frames, batch_stride, bin_stride, frame_stride, out_ptr):
        _check(self.lib.lra_mel_apply_exec(mel_plan, c_void_p(s_ptr), batch, n_frames, batch_stride, bin_stride, frame_stride, c_void_p(out_ptr)))

    def istft_exec_norm(self, plan, d_ptr, batch, d_batch_stride, d_frame_stride, n_used, norm_ptr, y_ptr, out_len, y_stride):
        """``istft_exec`` with the normalisation given as factors (``1 / wss`` where ``wss > tiny``, else 1; ``spectrum.wss_to_norm``)."""
        _check(self.lib.lra_istft_exec_norm(plan, c_void_p(d_ptr), batch, d_batch_stride, d_frame_stride, n_used, c_void_p(norm_ptr), c_void_p(y_ptr), out_len, y_stride))

    def istft_exec(self, plan, d_ptr, batch, d_batch_stride, d_frame_stride, n_used, wss_ptr, y_ptr, out_len, y_stride):
        _check(self.lib.lra_istft_exec(plan, c_void_p(d_ptr), batch, d_batch_stride, d_frame_stride, n_used, c_void_p(wss_ptr), c_void_p(y_ptr), out_len, y_stride))

    def item_absmax_exec(self, x_ptr, batch, per_item, dtype, out_ptr):
        _check(self.lib.lra_item_absmax_exec(self.handle, c_void_p(x_ptr), batch, per_item, dtype_code(dtype), c_void_p(out_ptr)))

    def item_max_exec(self, x_ptr, batch, per_item, dtype, out_ptr, absolute):
        """Per-item maximum of |x| (``absolute``: amplitude_to_db) or of max(x, 0) (power_to_db on real input, which stays signed)."""
        _check(self.lib.lra_item_max_exec(self.handle, c_void_p(x_ptr), batch, per_item, dtype_code(dtype), int(bool(absolute)), c_void_p(out_ptr)))

    def to_db_exec(self, x_ptr, out_ptr, batch, per_item, dtype, amplitude, amin, ref_scalar, ref_items_ptr, item_max_ptr, top_db):
        _check(self.lib.lra_to_db_exec(self.handle, c_void_p(x_ptr), c_void_p(out_ptr), batch, per_item, dtype_code(dtype), int(bool(amplitude)), float(amin), float(ref_scalar),
                                       c_void_p(ref_items_ptr or None), c_void_p(item_max_ptr or None), int(top_db is not None), float(top_db if top_db is not None else 0.0)))

    def from_db_exec(self, x_ptr, out_ptr, count, dtype, amplitude, ref):
        _check(self.lib.lra_from_db_exec(self.handle, c_void_p(x_ptr), c_void_p(out_ptr), count, dtype_code(dtype), int(bool(amplitude)), float(ref)))

    def dct_exec(self, s_ptr, out_ptr, batch, n_in, n_out, n_frames, dtype, basis_ptr, lift_ptr, fuse_db=False, amin=1e-10, ref_scalar=1.0, ref_items_ptr=None, item_max_ptr=None, top_db=None):
        _check(self.lib.lra_dct_exec(self.handle, c_void_p(s_ptr), c_void_p(out_ptr), batch, n_in, n_out, n_frames, dtype_code(dtype), c_void_p(basis_ptr), c_void_p(lift_ptr),
                                     int(bool(fuse_db)), float(amin), float(ref_scalar), c_void_p(ref_items_ptr or None), c_void_p(item_max_ptr or None), int(top_db is not None),
                                     float(top_db if top_db is not None else 0.0)))

    def phase_vocoder_exec(self, d_ptr, out_ptr, batch, n_frames, n_bins, t_out, dtype):
        t = np.ascontiguousarray(t_out, dtype=np.float64)
        _check(self.lib.lra_phase_vocoder_exec(self.handle, c_void_p(d_ptr), c_void_p(out_ptr), batch, n_frames, n_bins, c_void_p(t.ctypes.data), len(t), dtype_code(dtype)))

    def pcen_exec(self, s_ptr, ref_ptr, out_ptr, rows, n_frames, dtype, b, gain, bias, power, eps, zi_ptr, zi_scalar, zf_ptr):
        _check(self.lib.lra_pcen_exec(self.handle, c_void_p(s_ptr), c_void_p(ref_ptr or None), c_void_p(out_ptr), rows, n_frames, dtype_code(dtype), float(b), float(gain), float(bias),
                                      float(power), float(eps), c_void_p(zi_ptr or None), float(zi_scalar), c_void_p(zf_ptr or None)))

    def maxfilter_exec(self, s_ptr, out_ptr, outer, n_bands, inner, size, dtype):
        _check(self.lib.lra_maxfilter_exec(self.handle, c_void_p(s_ptr), c_void_p(out_ptr), outer, n_bands, inner, int(size), dtype_code(dtype)))

    def fir_decimate_exec(self, x_ptr, out_ptr, batch, n_in, n_out, taps_ptr, n_taps, down, first, div, mul, dtype):
        _check(self.lib.lra_fir_decimate_exec(self.handle, c_void_p(x_ptr), c_void_p(out_ptr), batch, n_in, n_out, c_void_p(taps_ptr), int(n_taps), int(down), int(first), float(div),
                                              float(mul), dtype_code(dtype)))

    def resample_poly_exec(self, x_ptr, out_ptr, batch, n_in, n_out, taps_ptr, n_taps, up, down, first, div, mul, dtype):
        _check(self.lib.lra_resample_poly_exec(self.handle, c_void_p(x_ptr), c_void_p(out_ptr), batch, n_in, n_out, c_void_p(taps_ptr), int(n_taps), int(up), int(down), int(first),
                                               float(div), float(mul), dtype_code(dtype)))

    def resample_fft_exec(self, x_ptr, out_ptr, batch, n_in, n_out, gain, dtype):
        _check(self.lib.lra_resample_fft_exec(self.handle, c_void_p(x_ptr), c_void_p(out_ptr), batch, n_in, n_out, float(gain), dtype_code(dtype)))

    def resample_band_exec(self, x_ptr, out_ptr, batch, n_in, n_out, fft_in, fft_out, k_mid, k_sigma, gain, dtype):
        _check(self.lib.lra_resample_band_exec(self.handle, c_void_p(x_ptr), c_void_p(out_ptr), batch, n_in, n_out, int(fft_in), int(fft_out), float(k_mid), float(k_sigma), float(gain),
                                               dtype_code(dtype)))

    def cqt_project_exec(self, d_ptr, out_ptr, row_ptr, col_ptr, val_ptr, sqrt_len_ptr, batch, frames_in, n_bins, n_frames, n_total, bin0, row0, n_rows, dtype):
        _check(self.lib.lra_cqt_project_exec(self.handle, c_void_p(d_ptr), c_void_p(out_ptr), c_void_p(row_ptr), c_void_p(col_ptr), c_void_p(val_ptr), c_void_p(sqrt_len_ptr or None), batch,
                                             frames_in, int(n_bins), n_frames, int(n_total), int(bin0), int(row0), int(n_rows), dtype_code(dtype)))

    def cqt_octave_supported(self, n_fft):
        return bool(self.lib.lra_cqt_octave_supported(int(n_fft)))

    def cqt_octave_exec(self, y_ptr, batch, n, y_stride, n_fft, hop, pad_mode, row_ptr, col_ptr, val_ptr, sqrt_len_ptr, out_ptr, n_frames, n_total, bin0, row0, n_rows, dtype):
        """One constant-Q octave in one launch (``include/librosa_amd.h``): STFT with a rectangular window + sparse projection + scaling + stacking."""
        _check(self.lib.lra_cqt_octave_exec(self.handle, c_void_p(y_ptr), batch, n, y_stride, n_fft, hop, PAD_MODES[pad_mode], c_void_p(row_ptr), c_void_p(col_ptr), c_void_p(val_ptr),
                                            c_void_p(sqrt_len_ptr) if sqrt_len_ptr else None, c_void_p(out_ptr), n_frames, n_total, bin0, row0, n_rows, dtype_code(dtype)))

    def cqt_recursion_exec(self, y_ptr, batch, octaves, pad_mode, sqrt_len_ptr, out_ptr, n_frames, n_total, taps_ptr, n_taps, first, scratch_ptr, scratch_bytes, overlap, dtype):
        """The octave recursion of one cqt / vqt call in one native call; ``octaves``: a ``CqtOctave`` ctypes array (``include/librosa_amd.h``: lra_cqt_octave)."""
        _check(self.lib.lra_cqt_recursion_exec(self.handle, c_void_p(y_ptr), batch, ctypes.cast(octaves, c_void_p), len(octaves), PAD_MODES[pad_mode], c_void_p(sqrt_len_ptr) if sqrt_len_ptr else None,
                                               c_void_p(out_ptr), n_frames, int(n_total), c_void_p(taps_ptr) if taps_ptr else None, int(n_taps), int(first), c_void_p(scratch_ptr) if scratch_ptr else None,
                                               int(scratch_bytes), int(bool(overlap)), dtype_code(dtype)))

    def magnitude_exec(self, d_ptr, mag_ptr, count, dtype):
        _check(self.lib.lra_magnitude_exec(self.handle, c_void_p(d_ptr), c_void_p(mag_ptr), count, dtype_code(dtype)))

    def magphase_exec(self, d_ptr, is_complex, mag_ptr, phase_ptr, count, power, dtype):
        _check(self.lib.lra_magphase_exec(self.handle, c_void_p(d_ptr), int(bool(is_complex)), c_void_p(mag_ptr), c_void_p(phase_ptr), count, float(power), dtype_code(dtype)))

    def hpss_exec(self, mag_ptr, d_ptr, out_h_ptr, out_p_ptr, batch, n_frames, n_bins, win_harm, win_perc, power, margin_harm, margin_perc, want_mask, dtype):
        _check(self.lib.lra_hpss_exec(self.handle, c_void_p(mag_ptr), c_void_p(d_ptr or None), c_void_p(out_h_ptr), c_void_p(out_p_ptr), batch, n_frames, int(n_bins), int(win_harm), int(win_perc),
                                      float(power), float(margin_harm), float(margin_perc), int(bool(want_mask)), dtype_code(dtype)))

    def memset(self, ptr, value, nbytes):
        _check(self.lib.lra_memset(self.handle, c_void_p(ptr), int(value), int(nbytes)))

    def griffinlim_init(self, u_ptr, s_ptr, angles_ptr, count, dtype):
        _check(self.lib.lra_griffinlim_init(self.handle, c_void_p(u_ptr), c_void_p(s_ptr), c_void_p(angles_ptr), count, dtype_code(dtype)))

    @staticmethod
    def _pcg64_state4(state, inc):
        m = (1 << 64) - 1
        return (ctypes.c_uint64 * 4)((int(state) >> 64) & m, int(state) & m, (int(inc) >> 64) & m, int(inc) & m)

    def pcg64_random_exec(self, state, inc, offset, out_ptr, count):
        """``np.random.Generator(PCG64)`` with the 128-bit ``state`` / ``inc`` of its ``bit_generator.state["state"]``: draws ``offset .. offset + count`` of ``random()`` (float64, device)."""
        _check(self.lib.lra_pcg64_random_exec(self.handle, self._pcg64_state4(state, inc), int(offset), c_void_p(out_ptr), int(count)))

    def griffinlim_init_pcg64(self, state, inc, s_ptr, angles_ptr, batch, n_bins, n_frames, dtype):
        _check(self.lib.lra_griffinlim_init_pcg64(self.handle, self._pcg64_state4(state, inc), c_void_p(s_ptr), c_void_p(angles_ptr), int(batch), int(n_bins), int(n_frames), dtype_code(dtype)))

    def griffinlim_update(self, rebuilt_ptr, tprev_ptr, s_ptr, angles_ptr, count, dtype, coef, eps, normalize=True):
        _check(self.lib.lra_griffinlim_update(self.handle, c_void_p(rebuilt_ptr), c_void_p(tprev_ptr) if tprev_ptr else None, c_void_p(s_ptr), c_void_p(angles_ptr), count, dtype_code(dtype),
                                              float(coef), float(eps), int(bool(normalize))))

    def probe_stream(self, direction, in_ptr, out_ptr, batch, rows_per_clip, n_fft, hop, clip_samples, strip_rows=0, waves_per_cu=0, row_pitch_bytes=0, piece_bytes=8):
        """The transform's access stream without its arithmetic (measurement aid; ``include/librosa_amd.h``)."""
        _check(self.lib.lra_probe_stream_pitched(self.handle, int(direction), c_void_p(in_ptr), c_void_p(out_ptr), batch, rows_per_clip, n_fft, hop, clip_samples, strip_rows, waves_per_cu,
                                                 int(row_pitch_bytes), int(piece_bytes)))

    def probe_stream_window(self, in_ptr, out_ptr, batch, rows_per_clip, n_fft, hop, clip_samples, strip_rows, waves_per_cu):
        """The forward access stream with the strips dealt out in address order to persistent waves (``include/librosa_amd.h``)."""
        _check(self.lib.lra_probe_stream_window(self.handle, c_void_p(in_ptr), c_void_p(out_ptr), batch, rows_per_clip, n_fft, hop, clip_samples, int(strip_rows), int(waves_per_cu)))

    def transpose(self, src_ptr, dst_ptr, batch, rows, cols, elem_bytes):
        _check(self.lib.lra_transpose(self.handle, c_void_p(src_ptr), c_void_p(dst_ptr), batch, rows, cols, elem_bytes))


_contexts = {}
_ctx_lock = threading.Lock()


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """RCCL unique id for a new communicator (rank 0 calls this; the host program distributes the bytes)."""
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    _check(load_library().lra_comm_unique_id(buf))
    return buf.raw


class Comm:
    """Native RCCL communicator of one rank (``include/librosa_amd.h``: ``lra_comm_*``), bound to a context's device and stream."""

    def __init__(self, ctx, rank, n_ranks, unique_id: bytes):
        self.ctx, self.rank, self.n_ranks = ctx, int(rank), int(n_ranks)
        self.lib = ctx.lib
        h = c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        _check(self.lib.lra_comm_init(ctx.handle, self.rank, self.n_ranks, buf, byref(h)))
        self.handle = h

    def allgather(self, send_ptr, recv_ptr, bytes_per_rank):
        _check(self.lib.lra_comm_allgather(self.handle, c_void_p(send_ptr), c_void_p(recv_ptr), int(bytes_per_rank)))

    def allgatherv(self, send_ptr, recv_ptr, bytes_per_rank, recv_offsets):
        """Unequal shards: rank r's ``bytes_per_rank[r]`` bytes land at ``recv_ptr + recv_offsets[r]`` on every rank (one grouped broadcast per rank)."""
        n = self.n_ranks
        if len(bytes_per_rank) != n or len(recv_offsets) != n:
            raise ParameterError(f"allgatherv: size / offset tables must have one entry per rank ({n})")
        sizes = (c_size_t * n)(*[int(v) for v in bytes_per_rank])
        offs = (c_size_t * n)(*[int(v) for v in recv_offsets])
        _check(self.lib.lra_comm_allgatherv(self.handle, c_void_p(send_ptr) if send_ptr else None, c_void_p(recv_ptr), sizes, offs))

    def close(self):
        if self.handle:
            self.lib.lra_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_devices():
    """Devices the NumPy drop-in spreads the clips of one call over, inside ONE process (a thread, a context and a host pipeline per
    device; ctypes releases the GIL during native calls).  Opt-in (a plain NumPy call must not wake contexts on every GPU of
    the node, least of all under a one-process-per-GPU launcher that this library does not know -- SLURM, MPI, a multiprocessing pool):
    by default a call stays on the process's own device (``get_context()``: ``LIBROSA_AMD_DEVICE`` / ``LOCAL_RANK``, else 0);
    ``LRA_DEVICES=all`` = every visible device, ``LRA_DEVICES=0,2`` = those."""
    n = device_count()
    if n <= 0:
        return [0]
    spec = os.environ.get("LRA_DEVICES", "").strip().lower()
    if spec == "":
        return [get_context().device]
    if spec == "all":
        return list(range(n))
    try:
        devs = [int(t) for t in spec.split(",") if t.strip()]
    except ValueError:
        raise ParameterError(f"LRA_DEVICES={spec!r}: expected comma-separated device indices or 'all'")
    bad = [d for d in devs if not 0 <= d < n]
    if bad:
        raise ParameterError(f"LRA_DEVICES names device(s) {bad}, but {n} device(s) are visible")
    return devs or [0]


def get_context(device=None):
    """Shared per-device context.  ``device=None`` -> LOCAL_RANK's device under torchrun, else 0."""
    if device is None:
        device = int(os.environ.get("LIBROSA_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = device_count()
        if n > 0:
            device %= n
    with _ctx_lock:
        ctx = _contexts.get(device)
        if ctx is None:
            ctx = Context(device)
            _contexts[device] = ctx
    return ctx
