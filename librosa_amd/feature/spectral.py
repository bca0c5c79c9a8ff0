"""``melspectrogram`` with librosa's signature (``librosa/feature/spectral.py:2022-2161``)."""
from __future__ import annotations

import numpy as np

from .. import _arrays
from .. import filters
from ..core import spectrum as _spectrum
from ..util.exceptions import ParameterError
from ..util.utils import is_torch_tensor

__all__ = ["melspectrogram"]


def melspectrogram(*, y=None, sr=22050, S=None, n_fft=2048, hop_length=512, win_length=None, window="hann", center=True, pad_mode="constant",
                   power=2.0, check_finite=True, **kwargs):
    """Mel-scaled spectrogram ``M[..., m, t] = sum_f mel[m, f] * |stft(y)[..., f, t]|**power``.

    ``kwargs`` go to ``filters.mel`` (``n_mels, fmin, fmax, htk, norm, dtype``).  With ``y`` given the
    whole chain -- framing, window, FFT, ``|.|**power`` and the banded mel reduce -- is ONE kernel for
    power-of-two ``n_fft``; the 1025-bin spectrum is never written to HBM.  With ``S`` given only the
    (banded) filterbank product runs (``feature/spectral.py:2145-2160``).
    """
    if S is not None:
        if n_fft is None or n_fft // 2 + 1 != S.shape[-2]:
            n_fft = 2 * (S.shape[-2] - 1)
        mel_basis = filters.mel_cached(sr=sr, n_fft=n_fft, **kwargs)
        return _apply_mel(S, mel_basis)
    if n_fft is None:
        raise ParameterError(f"Unable to compute spectrogram with n_fft={n_fft}")
    if y is None:
        raise ParameterError("Input signal must be provided to compute a spectrogram")
    mel_basis = filters.mel_cached(sr=sr, n_fft=n_fft, **kwargs)
    return _spectrum._run_stft_family("mel", y, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window, center=center,
                                      pad_mode=pad_mode, power=float(power), mel_basis=mel_basis, check_finite=check_finite)


def _apply_mel(S, mel_basis):
    """einsum('...ft,mf->...mt') with the filterbank in band form on the device."""
    s_dtype = _arrays.numpy_dtype_of(S)
    if s_dtype.kind == "c":
        raise ParameterError("S must be a real-valued (magnitude or power) spectrogram")
    real = np.dtype(np.float64) if (s_dtype == np.float64 or mel_basis.dtype == np.float64) else np.dtype(np.float32)
    if S.ndim < 2:
        raise ParameterError(f"S must have at least 2 dimensions, given shape={tuple(S.shape)}")
    lead = tuple(S.shape[:-2])
    n_bins, n_frames = int(S.shape[-2]), int(S.shape[-1])
    if n_bins != mel_basis.shape[1]:
        raise ParameterError(f"S has {n_bins} frequency bins but the mel basis expects {mel_basis.shape[1]}")
    n_mels = int(mel_basis.shape[0])
    batch = int(np.prod(lead, dtype=np.int64)) if lead else 1
    sess = _arrays.Session(S)
    try:
        ctx = sess.ctx
        mel_plan = ctx.mel_plan(np.ascontiguousarray(mel_basis, dtype=real))
        St = _arrays.swap_last_two(S)
        native = St.is_contiguous() if is_torch_tensor(S) else St.flags["C_CONTIGUOUS"]
        if native and n_frames > 1:
            # a view of the device layout [b][t][f] (what our own _spectrogram returns)
            s_ptr = sess.input_raw(St, real)
            strides = (n_frames * n_bins, 1, n_bins)  # batch, bin, frame
        else:
            s_ptr = sess.input_raw(S, real)
            strides = (n_bins * n_frames, n_frames, 1)
        ptr, handle = sess.output((batch, n_mels, n_frames), real)
        ctx.mel_apply_exec(mel_plan, s_ptr, batch, n_frames, strides[0], strides[1], strides[2], ptr)
        M = sess.result(handle)
    finally:
        sess.close()
    return M.reshape(lead + (n_mels, n_frames))
