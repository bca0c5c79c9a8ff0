"""Window and mel-filterbank construction (host side, float64), mirroring ``librosa/filters.py``.

These tables are tiny (a window of n_fft samples, a 128 x 1025 filterbank with 2018 non-zeros) and
are built once per configuration on the host with the same float64 arithmetic as the reference, so
they are bit-identical to ``librosa.filters.get_window`` (:914-977), ``mel`` (:116-251) and
``window_sumsquare`` (:1268-1339); the device receives them as plan constants.
"""
from __future__ import annotations

import functools
import warnings

import numpy as np
import scipy.signal

from .core.convert import fft_frequencies, mel_frequencies
from .util import utils as _u
from .util.exceptions import ParameterError

__all__ = ["get_window", "mel", "window_sumsquare"]


def get_window(window, Nx, *, fftbins=True):
    """Window of length ``Nx`` from a name / (name, param) tuple / number / callable / vector."""
    if callable(window):
        return window(Nx)
    if isinstance(window, (str, tuple)) or np.isscalar(window):
        return scipy.signal.get_window(window, Nx, fftbins=fftbins)
    if isinstance(window, (np.ndarray, list)):
        if len(window) == Nx:
            return np.asarray(window)
        raise ParameterError(f"Window size mismatch: {len(window):d} != {Nx:d}")
    raise ParameterError(f"Invalid window specification: {window!r}")


def mel(*, sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm="slaney", dtype=np.float32):
    """Triangular mel filterbank, shape ``(n_mels, 1 + n_fft//2)``.

    Row i rises from mel point i to i+1 and falls to i+2; every FFT bin is covered by at most two
    filters, which is what lets the device apply it as a banded (sparse) dot product.
    """
    if fmax is None:
        fmax = float(sr) / 2
    n_mels = int(n_mels)
    n_bins = int(1 + n_fft // 2)
    bin_hz = fft_frequencies(sr=sr, n_fft=n_fft)
    edges = mel_frequencies(n_mels + 2, fmin=fmin, fmax=fmax, htk=htk)
    widths = np.diff(edges)
    dist = np.subtract.outer(edges, bin_hz)  # (n_mels + 2, n_bins)
    weights = np.zeros((n_mels, n_bins), dtype=dtype)
    if n_mels > 0:
        rising = -dist[:-2] / widths[:-1, np.newaxis]
        falling = dist[2:] / widths[1:, np.newaxis]
        weights[...] = np.maximum(0, np.minimum(rising, falling))
    if isinstance(norm, str):
        if norm != "slaney":
            raise ParameterError(f"Unsupported norm={norm}")
        # constant energy per channel: divide by the width of the triangle's base
        weights *= (2.0 / (edges[2 : n_mels + 2] - edges[:n_mels]))[:, np.newaxis]
    else:
        weights = _u.normalize(weights, norm=norm, axis=-1)
    if not np.all((edges[:-2] == 0) | (weights.max(axis=1) > 0)):
        warnings.warn(
            "Empty filters detected in mel frequency basis. Some channels will produce empty responses. "
            "Try increasing your sampling rate (and fmax) or reducing n_mels.",
            stacklevel=2,
        )
    return weights


def window_sumsquare(*, window, n_frames, hop_length=512, win_length=None, n_fft=2048, dtype=np.float32, norm=None):
    """Sum of squared, hop-shifted windows: the ISTFT's normalisation envelope.

    The accumulator has ``dtype`` while the squared window is float64, so every ``+=`` rounds an f64
    sum back to ``dtype`` in frame order -- the same arithmetic as the reference's fill loop
    (``filters.py:1258-1265``), which is why this stays on the host instead of the device.
    """
    if win_length is None:
        win_length = n_fft
    n = n_fft + hop_length * (n_frames - 1)
    env = np.zeros(n, dtype=dtype)
    sq = _u.normalize(get_window(window, win_length), norm=norm) ** 2
    sq = _u.pad_center(sq, size=n_fft)
    for t in range(n_frames):
        s = t * hop_length
        env[s : min(n, s + n_fft)] += sq[: max(0, min(n_fft, n - s))]
    return env


@functools.lru_cache(maxsize=64)
def _mel_cached(sr, n_fft, n_mels, fmin, fmax, htk, norm, dtype_str):
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        B = mel(sr=sr, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax, htk=htk, norm=norm, dtype=np.dtype(dtype_str))
    B.setflags(write=False)
    return B, tuple(str(w.message) for w in caught)


def mel_cached(*, sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm="slaney", dtype=np.float32):
    """``mel`` memoised on its (hashable) arguments; re-emits the empty-filter warning on every call."""
    try:
        B, msgs = _mel_cached(sr, n_fft, int(n_mels), fmin, fmax, bool(htk), norm, np.dtype(dtype).str)
    except TypeError:  # unhashable argument
        return mel(sr=sr, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax, htk=htk, norm=norm, dtype=dtype)
    for m in msgs:
        warnings.warn(m, stacklevel=3)
    return B
