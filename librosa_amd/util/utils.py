"""Host-side helpers whose semantics the STFT/mel/ISTFT path must honour.

Each helper mirrors the behaviour (names, argument meaning, error type) of the reference helper it
cites; they operate on small host arrays (windows, filter tables) or validate arguments before any
device work is enqueued.
"""
from __future__ import annotations

import numpy as np

from .exceptions import ParameterError

# librosa/util/utils.py:40-41 -- the reference's only performance knob (column blocking of its CPU
# loops).  Kept for API compatibility; the device kernels do not block by columns.
MAX_MEM_BLOCK = 2**8 * 2**10


def is_torch_tensor(x) -> bool:
    mod = type(x).__module__
    return mod == "torch" or mod.startswith("torch.")


def valid_audio(y, *, scan=True) -> bool:
    """``librosa/util/utils.py:294-306``: ndarray, floating, >= 1-d, finite everywhere.

    ``scan=False`` skips the pass over the samples (the fused kernels then report non-finite frames themselves and the
    caller confirms on the samples only when they do)."""
    if not isinstance(y, np.ndarray):
        raise ParameterError("Audio data must be of type numpy.ndarray")
    if not np.issubdtype(y.dtype, np.floating):
        raise ParameterError("Audio data must be floating-point")
    if y.ndim == 0:
        raise ParameterError(f"Audio data must be at least one-dimensional, given y.shape={y.shape}")
    if scan and not np.isfinite(y).all():
        raise ParameterError("Audio buffer is not finite everywhere")
    return True


def is_positive_int(x) -> bool:
    """``librosa/util/utils.py:344-358``."""
    return isinstance(x, (int, np.integer)) and (x > 0)


def pad_center(data, *, size, axis=-1, **kwargs):
    """``librosa/util/utils.py:387-458``: centre ``data`` in a length-``size`` axis, lpad=(size-n)//2."""
    kwargs.setdefault("mode", "constant")
    n = data.shape[axis]
    lpad = int((size - n) // 2)
    if lpad < 0:
        raise ParameterError(f"Target size ({size:d}) must be at least input size ({n:d})")
    widths = [(0, 0)] * data.ndim
    widths[axis] = (lpad, int(size - n - lpad))
    return np.pad(data, widths, **kwargs)


def fix_length(data, *, size, axis=-1, **kwargs):
    """``librosa/util/utils.py:532-588``: truncate or right-pad ``axis`` to exactly ``size``."""
    kwargs.setdefault("mode", "constant")
    n = data.shape[axis]
    if n > size:
        index = [slice(None)] * data.ndim
        index[axis] = slice(0, size)
        return data[tuple(index)]
    if n < size:
        widths = [(0, 0)] * data.ndim
        widths[axis] = (0, size - n)
        return np.pad(data, widths, **kwargs)
    return data


def tiny(x):
    """``librosa/util/utils.py:1935-2000``: smallest positive normal of x's float type (f32 otherwise)."""
    x = np.asarray(x)
    if np.issubdtype(x.dtype, np.floating) or np.issubdtype(x.dtype, np.complexfloating):
        dtype = x.dtype
    else:
        dtype = np.dtype(np.float32)
    return np.finfo(dtype).tiny


def dtype_r2c(d, *, default=np.complex64):
    """``librosa/util/utils.py:2362-2416``: float32->complex64, float64->complex128, complex passes."""
    table = {np.dtype(np.float32): np.complex64, np.dtype(np.float64): np.complex128}
    dt = np.dtype(d)
    if dt.kind == "c":
        return dt
    return np.dtype(table.get(dt, default))


def dtype_c2r(d, *, default=np.float32):
    """``librosa/util/utils.py:2419-2476``."""
    table = {np.dtype(np.complex64): np.float32, np.dtype(np.complex128): np.float64}
    dt = np.dtype(d)
    if dt.kind == "f":
        return dt
    return np.dtype(table.get(dt, default))


def normalize(S, *, norm=np.inf, axis=0, threshold=None, fill=None):
    """Scale ``S`` to unit ``norm`` along ``axis``: ``librosa.util.normalize`` (``librosa/util/utils.py:796-1025``).  On this path it is
    reached from ``filters.mel(norm=<number>)`` (``filters.py:238-239``) and ``window_sumsquare(norm=...)`` (``filters.py:1325-1327``),
    both with the default ``fill=None``.

    ``norm``: ``None`` (no scaling), ``+-inf`` (max / min magnitude), ``0`` (number of non-zeros) or ``p > 0`` (the l_p norm).
    Slices whose norm is below ``threshold`` (default: ``tiny`` of the dtype) are left unscaled (``fill=None``), set to the constant
    that has unit norm (``fill=True``: 1 for the max / min norms, ``n ** (-1 / p)`` for l_p; undefined for ``norm=0``) or set to zero
    (``fill=False``).
    """
    if threshold is None:
        threshold = tiny(S)
    elif threshold <= 0:
        raise ParameterError(f"threshold={threshold} must be strictly positive")
    if not (fill is None or fill is True or fill is False):
        raise ParameterError(f"fill={fill} must be None or boolean")
    S = np.asarray(S)
    if not np.isfinite(S).all():
        raise ParameterError("Input must be finite")
    if norm is None:
        return S
    magnitude = np.abs(S).astype(float)
    reduce_kw = dict(axis=axis, keepdims=True)
    is_number = isinstance(norm, (int, float, np.number))
    unit_fill = 1.0  # the constant slice of unit norm (max / min norms)
    if is_number and np.isposinf(norm):
        scale = magnitude.max(**reduce_kw)
    elif is_number and np.isneginf(norm):
        scale = magnitude.min(**reduce_kw)
    elif is_number and norm == 0:
        if fill is True:
            raise ParameterError("Cannot normalize with norm=0 and fill=True")
        scale = np.count_nonzero(magnitude, **reduce_kw).astype(magnitude.dtype)
    elif is_number and norm > 0:
        scale = np.sum(magnitude**norm, **reduce_kw) ** (1.0 / norm)
        unit_fill = float(magnitude.size if axis is None else magnitude.shape[axis]) ** (-1.0 / norm)
    else:
        raise ParameterError(f"Unsupported norm: {norm!r}")
    small = scale < threshold
    out = np.empty_like(S)
    if fill is None:
        out[...] = S / np.where(small, 1.0, scale)
    elif fill:
        out[...] = np.where(np.broadcast_to(small, S.shape), unit_fill, S / np.where(small, 1.0, scale))
    else:
        out[...] = np.where(np.broadcast_to(small, S.shape), 0, S / np.where(small, 1.0, scale))
    return out


def sparsify_rows(x, *, quantile=0.01, dtype=None):
    """Row-wise sparsification: in every row, drop the smallest-magnitude entries that together hold less than ``quantile`` of
    the row's l1 mass; returns a ``scipy.sparse.csr_array`` (``librosa.util.sparsify_rows``, ``util/utils.py:1500-1597``)."""
    import scipy.sparse

    x = np.asarray(x)
    if x.ndim == 1:
        x = x.reshape((1, -1))
    elif x.ndim > 2:
        raise ParameterError(f"Input must have 2 or fewer dimensions. Provided x.shape={x.shape}.")
    if not 0.0 <= quantile < 1:
        raise ParameterError(f"Invalid quantile {quantile:.2f}")
    out_dtype = np.dtype(x.dtype if dtype is None else dtype)
    magnitude = np.abs(x)
    ascending = np.sort(magnitude, axis=1)
    share = np.cumsum(ascending / np.sum(magnitude, axis=1, keepdims=True), axis=1)
    cut = ascending[np.arange(x.shape[0]), np.argmin(share < quantile, axis=1)]          # per row: the smallest magnitude kept
    return scipy.sparse.csr_array((x * (magnitude >= cut[:, np.newaxis])).astype(out_dtype, copy=False))
