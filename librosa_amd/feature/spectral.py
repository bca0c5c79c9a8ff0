"""``melspectrogram`` with librosa's signature (``librosa/feature/spectral.py:2022-2161``)."""
from __future__ import annotations

import numpy as np

from .. import _arrays
from .. import filters
from ..core import spectrum as _spectrum
from ..util.exceptions import ParameterError
from ..util.utils import is_torch_tensor

__all__ = ["melspectrogram", "mfcc"]


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


def _dct_tables(n_in, n_mfcc, dct_type, norm, lifter, real):
    """Host tables of the MFCC contraction: the first ``n_mfcc`` rows of scipy's DCT applied to the identity (float64 recipe,
    rounded once to the compute dtype; stored band-major with the coefficient axis zero-padded to a multiple of 128, the C ABI's contract) and the lifter weights
    ``1 + (lifter / 2) sin(pi (k + 1) / lifter)`` in the result dtype (``feature/spectral.py:2005-2015``)."""
    import scipy.fft

    if lifter < 0:
        raise ParameterError(f"MFCC lifter={lifter} must be a non-negative number")
    full = scipy.fft.dct(np.eye(n_in, dtype=np.float64), axis=0, type=dct_type, norm=norm)
    n_out = min(int(n_mfcc), n_in)
    rows = -(-n_out // 128) * 128
    basis = np.zeros((n_in, rows), dtype=real)  # band-major: the coefficients of one band are contiguous (the kernel's scalar loads)
    basis[:, :n_out] = full[:n_out].T
    if lifter > 0:
        if n_out != int(n_mfcc):
            raise ParameterError(f"n_mfcc={n_mfcc} exceeds the {n_in} bands of the input: the lifter cannot be applied")
        li = np.sin(np.pi * np.arange(1, 1 + n_out, dtype=real) / lifter)
        lift = np.asarray(1 + (lifter / 2) * li, dtype=real)
    else:
        lift = np.ones(n_out, dtype=real)
    return basis, lift, n_out


def mfcc(*, y=None, sr=22050, S=None, n_mfcc=20, dct_type=2, norm="ortho", lifter=0, mel_norm="slaney", check_finite=True, **kwargs):
    """Mel-frequency cepstral coefficients; drop-in for ``librosa.feature.mfcc`` (``librosa/feature/spectral.py:1843-2019``).

    ``S`` (a log-power mel spectrogram) given: ``scipy.fft.dct(S, axis=-2, type=dct_type, norm=norm)[..., :n_mfcc, :]`` times
    the lifter, as one device contraction.  ``y`` given: ``S = power_to_db(melspectrogram(y=y, sr=sr, norm=mel_norm,
    **kwargs))`` (``:2001``) -- here the fused mel kernel, one per-clip maximum reduction (``top_db``) and the DCT kernel, which
    applies the decibel scaling while it reads the mel power spectrogram: nothing but the ``n_mfcc`` rows returns to the host."""
    if lifter < 0:
        raise ParameterError(f"MFCC lifter={lifter} must be a non-negative number")
    if S is not None:
        return _mfcc_of(S, n_mfcc, dct_type, norm, lifter)
    n_fft = kwargs.pop("n_fft", 2048)
    hop_length = kwargs.pop("hop_length", 512)
    win_length = kwargs.pop("win_length", None)
    window = kwargs.pop("window", "hann")
    center = kwargs.pop("center", True)
    pad_mode = kwargs.pop("pad_mode", "constant")
    power = kwargs.pop("power", 2.0)
    if n_fft is None:
        raise ParameterError(f"Unable to compute spectrogram with n_fft={n_fft}")
    if y is None:
        raise ParameterError("Input signal must be provided to compute a spectrogram")
    mel_basis = filters.mel_cached(sr=sr, n_fft=n_fft, norm=mel_norm, **kwargs)
    tables = {}

    def post(sess, mel_ptr, batch, n_mels, n_frames, real):
        basis, lift, n_out = _dct_tables(n_mels, n_mfcc, dct_type, norm, lifter, real)
        tables["n_out"] = n_out
        ctx = sess.ctx
        max_ptr = sess.scratch(batch * real.itemsize)
        ctx.item_max_exec(mel_ptr, batch, n_mels * n_frames, real, max_ptr, absolute=False)  # power_to_db's top_db = 80 (defaults, :2001)
        out_ptr, handle = sess.output((batch, n_out, n_frames), real)
        ctx.dct_exec(mel_ptr, out_ptr, batch, n_mels, n_out, n_frames, real, sess.input_raw(_spectrum._as_like(sess, basis), real),
                     sess.input_raw(_spectrum._as_like(sess, lift), real), fuse_db=True, amin=1e-10, ref_scalar=1.0, item_max_ptr=max_ptr, top_db=80.0)
        return handle, n_out

    return _spectrum._run_stft_family("mel", y, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window, center=center, pad_mode=pad_mode,
                                      power=float(power), mel_basis=mel_basis, check_finite=check_finite, post=post)


def _mfcc_of(S, n_mfcc, dct_type, norm, lifter):
    """DCT over axis -2 of a (log-)mel spectrogram (``feature/spectral.py:2005-2015``)."""
    s_dtype = _arrays.numpy_dtype_of(S)
    if s_dtype.kind == "c":
        raise ParameterError("S must be a real-valued (log-power) mel spectrogram")
    if S.ndim < 2:
        raise ParameterError(f"S must have at least 2 dimensions, given shape={tuple(S.shape)}")
    real = np.dtype(np.float32) if s_dtype == np.float32 else np.dtype(np.float64)
    lead = tuple(S.shape[:-2])
    n_in, n_frames = int(S.shape[-2]), int(S.shape[-1])
    batch = int(np.prod(lead, dtype=np.int64)) if lead else 1
    basis, lift, n_out = _dct_tables(n_in, n_mfcc, dct_type, norm, lifter, real)
    sess = _arrays.Session(S)
    try:
        s_ptr = sess.input_raw(S, real)
        out_ptr, handle = sess.output((batch, n_out, n_frames), real)
        sess.ctx.dct_exec(s_ptr, out_ptr, batch, n_in, n_out, n_frames, real, sess.input_raw(_spectrum._as_like(sess, basis), real),
                          sess.input_raw(_spectrum._as_like(sess, lift), real))
        M = sess.result(handle)
    finally:
        sess.close()
    return M.reshape(lead + (n_out, n_frames))


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
        padded = _spectrum._frame_major_strides(St, n_bins) if (is_torch_tensor(S) and not native and St.dtype == _arrays.torch_dtype(real)) else None
        if padded is not None and n_frames > 1:
            # the device layout with each frame's row padded to whole cache lines (what _spectrogram returns for device tensors): read in place
            sess._keep.append(St)
            s_ptr = St.data_ptr()
            strides = (padded[0], 1, padded[1])
        elif native and n_frames > 1:
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
