"""Spectrogram decomposition on the device: ``librosa.decompose.hpss`` (``librosa/decompose.py:371-528``; SURVEY.md 8f rank 3, the step
between the forward and the two inverse transforms of ``librosa.effects.hpss``)."""
from __future__ import annotations

import numpy as np

from . import _arrays
from .core.spectrum import _to_frame_major
from .util.exceptions import ParameterError
from .util.utils import is_torch_tensor

__all__ = ["hpss"]


def _pair(value):
    return (value[0], value[1]) if isinstance(value, (tuple, list)) else (value, value)


def hpss(S, *, kernel_size=31, power=2.0, mask=False, margin=1.0):
    """Median-filtering harmonic / percussive source separation; drop-in for ``librosa.decompose.hpss``.

    One device pass over the spectrogram (``csrc/lra_hpss.h``): both running medians (a sorting network in registers; the values
    are exactly the ones ``scipy.ndimage.median_filter`` selects), the two soft masks and the masked spectrogram with the input's
    phase.  ``S`` is ``(..., n_bins, n_frames)``, complex or non-negative real, a NumPy array or a device tensor (device tensors are
    returned for device input; the output of ``librosa_amd.stft`` is consumed and produced without a transpose).  Integer input is
    taken as float32.  For device tensors the non-negativity of a real ``S`` is the caller's responsibility, and so is its finiteness:
    the selection network works on ``min`` / ``max``, which skip a NaN operand where scipy's filter would carry it through, so a NumPy
    ``S`` with non-finite entries is rejected here (``ParameterError``) instead of being separated into finite, wrong medians.
    """
    win_harm, win_perc = _pair(kernel_size)
    margin_harm, margin_perc = _pair(margin)
    if margin_harm < 1 or margin_perc < 1:
        raise ParameterError("Margins must be >= 1.0. A typical range is between 1 and 10.")
    if power <= 0:
        raise ParameterError("power must be strictly positive")
    win_harm, win_perc = int(win_harm), int(win_perc)
    if win_harm < 1 or win_perc < 1:
        raise ParameterError(f"kernel_size={kernel_size} must be positive")
    on_device = is_torch_tensor(S)
    if not on_device:
        S = np.asarray(S)
    if S.ndim < 2:
        raise ParameterError(f"S must have at least 2 dimensions, given shape={tuple(S.shape)}")
    in_dtype = _arrays.numpy_dtype_of(S)
    is_complex = in_dtype.kind == "c"
    if not on_device and in_dtype.kind in "fc" and not np.isfinite(S).all():
        raise ParameterError("S is not finite everywhere")   # (see the docstring: the device medians would skip NaN entries)
    if is_complex:
        cplx = np.dtype(in_dtype)
        real = np.dtype(np.float64) if cplx == np.complex128 else np.dtype(np.float32)
    else:
        real = np.dtype(np.float64) if in_dtype == np.float64 else np.dtype(np.float32)
        if not on_device and np.any(S < 0):
            raise ParameterError("X and X_ref must be non-negative")   # what util.softmask says about the medians of such an S
    n_bins, n_frames = int(S.shape[-2]), int(S.shape[-1])
    lead = tuple(int(v) for v in S.shape[:-2])
    batch = int(np.prod(lead, dtype=np.int64)) if lead else 1
    count = batch * n_bins * n_frames
    out_dtype = real if (mask or not is_complex) else cplx
    if count == 0:
        empty = np.zeros(tuple(S.shape), dtype=out_dtype)
        if on_device:
            empty = _arrays._torch().from_numpy(empty).to(S.device)
        return empty, empty.copy() if not on_device else empty.clone()
    sess = _arrays.Session(S if on_device else np.empty(0))
    try:
        ctx = sess.ctx
        if is_complex:
            d_ptr = _to_frame_major(sess, S, batch, n_bins, n_frames, cplx)
            mag_ptr = sess.scratch(count * real.itemsize)
            ctx.magnitude_exec(d_ptr, mag_ptr, count, real)
        else:
            d_ptr = None
            mag_ptr = _to_frame_major(sess, S, batch, n_bins, n_frames, real)
        h_ptr, h_handle = sess.output((batch, n_frames, n_bins), out_dtype)
        p_ptr, p_handle = sess.output((batch, n_frames, n_bins), out_dtype)
        ctx.hpss_exec(mag_ptr, d_ptr, h_ptr, p_ptr, batch, n_frames, n_bins, win_harm, win_perc, power, margin_harm, margin_perc, mask, real)
        harm, perc = sess.result(h_handle), sess.result(p_handle)
    finally:
        sess.close()
    harm = _arrays.swap_last_two(harm.reshape(lead + (n_frames, n_bins)))
    perc = _arrays.swap_last_two(perc.reshape(lead + (n_frames, n_bins)))
    if mask and np.isinf(power):   # the reference's hard masks are boolean (util/utils.py:1930)
        harm, perc = harm != 0, perc != 0
    return harm, perc
