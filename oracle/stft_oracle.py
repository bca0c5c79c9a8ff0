"""CPU oracle: a NumPy restatement of librosa's STFT -> mel (+ ISTFT) hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module.  Nothing under ``librosa_amd/``
does; the product path fails loudly when the HIP library is missing.

Parity pinning
--------------
Every function cites the reference lines it restates (paths relative to ``/root/reference``).
The restatement is pinned three ways (see ``tests/test_oracle.py``):

1. against the reference's own known-answer vectors for this path
   (``tests/test_filters.py:35-98`` mel-scale KATs, ``tests/test_core.py:256-292`` rfft-of-frames
   definition of the STFT, ``tests/test_core.py:813-828`` round trip);
2. against outputs of the reference itself, generated in the build container by importing the
   unmodified reference through ``oracle/ref_shim.py`` and committed as fixtures under
   ``tests/golden/`` (script: ``oracle/make_golden.py``);
3. when ``/root/reference`` is present, live against the reference on seeded random cases.

Third-party arithmetic on the path that is NOT under ``/root/reference`` and is used here
through the very same library the reference calls: ``scipy.fft.rfft/irfft`` (pocketfft, scipy
1.15.3 in this image; call sites ``librosa/core/spectrum.py:372,376,388,566,598``),
``scipy.signal.get_window`` (``librosa/filters.py:968``), ``np.pad``
(``librosa/core/spectrum.py:287,298,313``), ``np.fft.rfftfreq`` (``librosa/core/convert.py:1391``).

The restatement deliberately does NOT copy the reference's head/middle/tail copy-avoidance
split (``core/spectrum.py:273-328``) or its ``MAX_MEM_BLOCK`` column blocking (``:380-390``,
``:588-603``): the result is *defined* (and tested by the reference,
``tests/test_core.py:279-292``) as the rfft of the fully padded, framed, windowed signal.
"""
from __future__ import annotations

import warnings

import numpy as np
import scipy.fft
import scipy.signal


class ParameterError(ValueError):
    """Mirror of ``librosa/util/exceptions.py:12`` for the oracle's own argument checks."""


# ----------------------------------------------------------------------------- helpers (L1/L2)
def dtype_r2c(d, default=np.complex64):
    """``librosa/util/utils.py:2400-2416``: f32->c64, f64->c128, complex passes, else default."""
    dt = np.dtype(d)
    if dt.kind == "c":
        return dt
    return np.dtype({np.dtype(np.float32): np.complex64, np.dtype(np.float64): np.complex128}.get(dt, default))


def dtype_c2r(d, default=np.float32):
    """``librosa/util/utils.py:2462-2476``."""
    dt = np.dtype(d)
    if dt.kind == "f":
        return dt
    return np.dtype({np.dtype(np.complex64): np.float32, np.dtype(np.complex128): np.float64}.get(dt, default))


def tiny(x):
    """``librosa/util/utils.py:1985-2000``: smallest normal of x's float dtype (f32 default)."""
    x = np.asarray(x)
    if np.issubdtype(x.dtype, np.floating) or np.issubdtype(x.dtype, np.complexfloating):
        dtype = x.dtype
    else:
        dtype = np.dtype(np.float32)
    return np.finfo(dtype).tiny


def pad_center(data, size):
    """``librosa/util/utils.py:440-458`` (axis=-1, constant mode): lpad=(size-n)//2."""
    n = data.shape[-1]
    lpad = int((size - n) // 2)
    if lpad < 0:
        raise ParameterError(f"Target size ({size:d}) must be at least input size ({n:d})")
    lengths = [(0, 0)] * data.ndim
    lengths[-1] = (lpad, int(size - n - lpad))
    return np.pad(data, lengths, mode="constant")


def fix_length(data, size):
    """``librosa/util/utils.py:570-588`` (axis=-1): truncate or zero-pad on the right."""
    n = data.shape[-1]
    if n > size:
        return data[..., :size]
    if n < size:
        lengths = [(0, 0)] * data.ndim
        lengths[-1] = (0, size - n)
        return np.pad(data, lengths, mode="constant")
    return data


def get_window(window, Nx, fftbins=True):
    """``librosa/filters.py:960-977``: callable / name|tuple|scalar -> scipy / explicit vector."""
    if callable(window):
        return window(Nx)
    if isinstance(window, (str, tuple)) or np.isscalar(window):
        return scipy.signal.get_window(window, Nx, fftbins=fftbins)
    if isinstance(window, (np.ndarray, list)):
        if len(window) == Nx:
            return np.asarray(window)
        raise ParameterError(f"Window size mismatch: {len(window):d} != {Nx:d}")
    raise ParameterError(f"Invalid window specification: {window!r}")


def normalize(S, norm=np.inf, axis=-1):
    """``librosa/util/utils.py:960-1025`` restricted to threshold=None, fill=None."""
    threshold = tiny(S)
    if not np.all(np.isfinite(S)):
        raise ParameterError("Input must be finite")
    mag = np.abs(S).astype(float)
    if norm is None:
        return S
    if norm == np.inf:
        length = np.max(mag, axis=axis, keepdims=True)
    elif norm == -np.inf:
        length = np.min(mag, axis=axis, keepdims=True)
    elif norm == 0:
        length = np.sum(mag > 0, axis=axis, keepdims=True, dtype=mag.dtype)
    elif np.issubdtype(type(norm), np.number) and norm > 0:
        length = np.sum(mag**norm, axis=axis, keepdims=True) ** (1.0 / norm)
    else:
        raise ParameterError(f"Unsupported norm: {norm!r}")
    small_idx = length < threshold
    Snorm = np.empty_like(S)
    length[small_idx] = 1.0
    Snorm[:] = S / length
    return Snorm


def hz_to_mel(frequencies, htk=False):
    """``librosa/core/convert.py:1032-1058``: Slaney (linear<1 kHz, log above) or HTK scale."""
    frequencies = np.asanyarray(frequencies)[()]
    if htk:
        return 2595.0 * np.log10(1.0 + frequencies / 700.0)
    f_min = 0.0
    f_sp = 200.0 / 3
    mels = (frequencies - f_min) / f_sp
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if frequencies.ndim:
        log_t = frequencies >= min_log_hz
        mels[log_t] = min_log_mel + np.log(frequencies[log_t] / min_log_hz) / logstep
    elif frequencies >= min_log_hz:
        mels = min_log_mel + np.log(frequencies / min_log_hz) / logstep
    return mels


def mel_to_hz(mels, htk=False):
    """``librosa/core/convert.py:1098-1121``."""
    mels = np.asanyarray(mels)[()]
    if htk:
        return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    f_min = 0.0
    f_sp = 200.0 / 3
    freqs = f_min + f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if mels.ndim:
        log_t = mels >= min_log_mel
        freqs[log_t] = min_log_hz * np.exp(logstep * (mels[log_t] - min_log_mel))
    elif mels >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (mels - min_log_mel))
    return freqs


def fft_frequencies(sr=22050, n_fft=2048):
    """``librosa/core/convert.py:1391``."""
    return np.fft.rfftfreq(n=n_fft, d=1.0 / sr)


def mel_frequencies(n_mels=128, fmin=0.0, fmax=11025.0, htk=False):
    """``librosa/core/convert.py:1505-1511``: linspace in mel, mapped back to Hz."""
    min_mel = hz_to_mel(fmin, htk=htk)
    max_mel = hz_to_mel(fmax, htk=htk)
    mels = np.linspace(min_mel, max_mel, n_mels)
    return mel_to_hz(mels, htk=htk)


def mel(*, sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm="slaney", dtype=np.float32):
    """``librosa/filters.py:206-251``: triangles in f64, stored in ``dtype``, slaney area norm."""
    if fmax is None:
        fmax = float(sr) / 2
    n_mels = int(n_mels)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)), dtype=dtype)
    fftfreqs = fft_frequencies(sr=sr, n_fft=n_fft)
    mel_f = mel_frequencies(n_mels + 2, fmin=fmin, fmax=fmax, htk=htk)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    if isinstance(norm, str):
        if norm == "slaney":
            enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
            weights *= enorm[:, np.newaxis]
        else:
            raise ParameterError(f"Unsupported norm={norm}")
    else:
        weights = normalize(weights, norm=norm, axis=-1)
    if not np.all((mel_f[:-2] == 0) | (weights.max(axis=1) > 0)):
        warnings.warn("Empty filters detected in mel frequency basis.", stacklevel=2)
    return weights


def window_sumsquare(*, window, n_frames, hop_length=512, win_length=None, n_fft=2048, dtype=np.float32, norm=None):
    """``librosa/filters.py:1325-1339`` + fill loop ``:1258-1265``.

    The accumulator has ``dtype`` (f32 by default) while ``win_sq`` is f64, exactly as in the
    reference: each ``+=`` is an f64 add rounded back to ``dtype``.
    """
    if win_length is None:
        win_length = n_fft
    n = n_fft + hop_length * (n_frames - 1)
    x = np.zeros(n, dtype=dtype)
    win_sq = get_window(window, win_length)
    win_sq = normalize(win_sq, norm=norm) ** 2
    win_sq = pad_center(win_sq, size=n_fft)
    for i in range(n_frames):
        sample = i * hop_length
        x[sample : min(n, sample + n_fft)] += win_sq[: max(0, min(n_fft, n - sample))]
    return x


# ----------------------------------------------------------------------------- L3: stft / istft
def _frame(x, frame_length, hop_length):
    """``librosa/util/utils.py:210-242`` (axis=-1): (..., n) -> (..., frame_length, n_frames) view."""
    if x.shape[-1] < frame_length:
        raise ParameterError(f"Input is too short (n={x.shape[-1]:d}) for frame_length={frame_length:d}")
    xw = np.lib.stride_tricks.sliding_window_view(x, frame_length, axis=-1)  # (..., n-L+1, L)
    xw = np.moveaxis(xw, -1, -2)
    return xw[..., ::hop_length]


_BAD_PAD_MODES = ("wrap", "maximum", "mean", "median", "minimum")


def stft(y, *, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True, dtype=None, pad_mode="constant"):
    """``librosa/core/spectrum.py:230-391``.

    D[..., f, t] = rfft(w * yhat[..., t*hop : t*hop + n_fft])[f]; the f64 window times the
    framed signal is an f64 product, pocketfft runs in double, and the result is rounded once to
    the output complex dtype (``:340-341, :388-390``).
    """
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = int(win_length // 4)
    elif not (isinstance(hop_length, (int, np.integer)) and hop_length > 0):
        raise ParameterError(f"hop_length={hop_length} must be a positive integer")
    if not np.isfinite(y).all():
        raise ParameterError("Audio buffer is not finite everywhere")
    fft_window = pad_center(get_window(window, win_length, fftbins=True), size=n_fft)
    fft_window = fft_window.reshape((1,) * (y.ndim - 1) + (n_fft, 1))
    if center:
        if pad_mode in _BAD_PAD_MODES:
            raise ParameterError(f"pad_mode='{pad_mode}' is not supported by librosa.stft")
        padding = [(0, 0)] * y.ndim
        padding[-1] = (n_fft // 2, n_fft // 2)
        y = np.pad(y, padding, mode=pad_mode)
    elif n_fft > y.shape[-1]:
        raise ParameterError(f"n_fft={n_fft} is too large for uncentered analysis of input signal of length={y.shape[-1]}")
    if dtype is None:
        dtype = dtype_r2c(y.dtype)
    y_frames = _frame(y, n_fft, hop_length)
    shape = list(y_frames.shape)
    shape[-2] = 1 + n_fft // 2
    D = np.zeros(shape, dtype=dtype, order="F")
    # bounded-memory blocking (not the reference's block size; blocking does not change values)
    n_frames = y_frames.shape[-1]
    lead = int(np.prod(y_frames.shape[:-1]))
    block = max(1, (1 << 24) // max(1, lead))
    for s in range(0, n_frames, block):
        t = min(n_frames, s + block)
        D[..., s:t] = scipy.fft.rfft(fft_window * y_frames[..., s:t], axis=-2)
    return D


def istft(D, *, hop_length=None, win_length=None, n_fft=None, window="hann", center=True, dtype=None, length=None):
    """``librosa/core/spectrum.py:506-626`` + ``__overlap_add`` ``:629-643``.

    Frames are inverse-transformed (c64 -> f32 irfft, c128 -> f64), multiplied by the f64 window
    and added in increasing frame order into a buffer of the output dtype, then divided by the
    window sum-square where it exceeds ``tiny``.  The head block / offset bookkeeping of the
    reference (``:557-603``) is equivalent to adding frame t at padded position t*hop and
    dropping the first n_fft//2 samples; frames used are [0, min(T, start_frame)) from the head
    block plus [start_frame, n_frames) from the main loop.
    """
    if n_fft is None:
        n_fft = 2 * (D.shape[-2] - 1)
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = int(win_length // 4)
    ifft_window = pad_center(get_window(window, win_length, fftbins=True), size=n_fft)
    if length:
        padded_length = length + 2 * (n_fft // 2) if center else length
        n_frames = min(D.shape[-1], int(np.ceil(padded_length / hop_length)))
    else:
        n_frames = D.shape[-1]
    if dtype is None:
        dtype = dtype_c2r(D.dtype)
    expected_signal_len = n_fft + hop_length * (n_frames - 1)
    if length:
        expected_signal_len = length
    elif center:
        expected_signal_len -= 2 * (n_fft // 2)
    lead = D.shape[:-2]
    if center:
        start_frame = int(np.ceil((n_fft // 2) / hop_length))
        used = max(n_frames, min(D.shape[-1], start_frame))
        drop = n_fft // 2
    else:
        used = n_frames
        drop = 0
    # accumulate in the padded coordinate system, in the output dtype, frame by frame
    total = max(drop + expected_signal_len, 0)
    buf = np.zeros(lead + (total,), dtype=dtype)
    for t in range(used):
        s = t * hop_length
        if s >= total:
            break
        ytmp = ifft_window * scipy.fft.irfft(D[..., t], n=n_fft, axis=-1)
        N = min(n_fft, total - s)
        buf[..., s : s + N] += ytmp[..., :N]
    y = np.ascontiguousarray(buf[..., drop : drop + expected_signal_len])
    wss = window_sumsquare(window=window, n_frames=n_frames, win_length=win_length, n_fft=n_fft, hop_length=hop_length, dtype=dtype)
    wss = fix_length(wss[drop:], size=y.shape[-1])
    nz = wss > tiny(wss)
    y[..., nz] /= wss[nz]
    return y


# ----------------------------------------------------------------------------- L4: spectrogram / mel
def spectrogram(*, y=None, S=None, n_fft=2048, hop_length=512, power=1, win_length=None, window="hann", center=True, pad_mode="constant"):
    """``librosa/core/spectrum.py:2988-3015`` (``_spectrogram``): abs(stft)**power or pass-through."""
    if S is not None:
        if n_fft is None or n_fft // 2 + 1 != S.shape[-2]:
            n_fft = 2 * (S.shape[-2] - 1)
    else:
        if n_fft is None:
            raise ParameterError(f"Unable to compute spectrogram with n_fft={n_fft}")
        if y is None:
            raise ParameterError("Input signal must be provided to compute a spectrogram")
        S = np.abs(stft(y, n_fft=n_fft, hop_length=hop_length, win_length=win_length, center=center, window=window, pad_mode=pad_mode)) ** power
    return S, n_fft


def melspectrogram(*, y=None, sr=22050, S=None, n_fft=2048, hop_length=512, win_length=None, window="hann", center=True, pad_mode="constant", power=2.0, **kwargs):
    """``librosa/feature/spectral.py:2145-2161``: einsum('...ft,mf->...mt', S, mel_basis)."""
    S, n_fft = spectrogram(y=y, S=S, n_fft=n_fft, hop_length=hop_length, power=power, win_length=win_length, window=window, center=center, pad_mode=pad_mode)
    mel_basis = mel(sr=sr, n_fft=n_fft, **kwargs)
    return np.einsum("...ft,mf->...mt", S, mel_basis, optimize=True)


# ----------------------------------------------------------------------------- L5: decibel scaling and MFCC
def _db_axes(ndim, axes):
    """``librosa/core/spectrum.py:1853-1859``: axes="auto" -> the last two axes (one for 1-d input, None for scalars)."""
    if isinstance(axes, str) and axes == "auto":
        return (-2, -1) if ndim >= 2 else ((-1,) if ndim == 1 else None)
    return axes


def power_to_db(S, *, ref=1.0, amin=1e-10, top_db=80.0, axes="auto"):
    """``librosa/core/spectrum.py:1838-1883``."""
    S = np.asarray(S)
    if amin <= 0:
        raise ParameterError("amin must be strictly positive")
    magnitude = np.abs(S) if np.issubdtype(S.dtype, np.complexfloating) else S
    axes = _db_axes(magnitude.ndim, axes)
    ref_value = ref(magnitude, axis=axes, keepdims=True) if callable(ref) else np.abs(ref)
    log_spec = 10.0 * np.log10(np.maximum(amin, magnitude))
    log_spec -= 10.0 * np.log10(np.maximum(amin, ref_value))
    if top_db is not None:
        if top_db < 0:
            raise ParameterError("top_db must be non-negative")
        log_spec = np.maximum(log_spec, log_spec.max(axis=axes, keepdims=True) - top_db)
    return log_spec[()]


def amplitude_to_db(S, *, ref=1.0, amin=1e-5, top_db=80.0, axes="auto"):
    """``librosa/core/spectrum.py:2000-2038``: power_to_db(|S|**2, ref=ref**2, amin=amin**2)."""
    magnitude = np.abs(np.asarray(S))
    axes = _db_axes(magnitude.ndim, axes)
    ref_value = ref(magnitude, axis=axes, keepdims=True) if callable(ref) else np.abs(ref)
    power = np.square(magnitude)
    return power_to_db(power, ref=ref_value**2, amin=amin**2, top_db=top_db, axes=axes)


def db_to_power(S_db, *, ref=1.0):
    """``librosa/core/spectrum.py:1925``."""
    return ref * np.power(10.0, np.asarray(S_db) * 0.1)


def db_to_amplitude(S_db, *, ref=1.0):
    """``librosa/core/spectrum.py:2082``."""
    return db_to_power(S_db, ref=ref**2) ** 0.5


def mfcc(*, y=None, sr=22050, S=None, n_mfcc=20, dct_type=2, norm="ortho", lifter=0, mel_norm="slaney", **kwargs):
    """``librosa/feature/spectral.py:1999-2019``: DCT over the mel axis of the log-power mel spectrogram (+ lifter)."""
    if S is None:
        S = power_to_db(melspectrogram(y=y, sr=sr, norm=mel_norm, **kwargs))
    M = scipy.fft.dct(S, axis=-2, type=dct_type, norm=norm)[..., :n_mfcc, :]
    if lifter > 0:
        LI = np.sin(np.pi * np.arange(1, 1 + n_mfcc, dtype=M.dtype) / lifter)
        LI = LI.reshape((1,) * (S.ndim - 2) + (-1, 1))
        M *= 1 + (lifter / 2) * LI
        return M
    if lifter == 0:
        return M
    raise ParameterError(f"MFCC lifter={lifter} must be a non-negative number")


# ----------------------------------------------------------------------------- Griffin-Lim
def phasor(angles, mag=None):
    """``librosa/util/utils.py:2629-2637, 2700-2706``: cos + i sin in the precision of ``angles``, times ``mag``."""
    angles = np.asarray(angles)
    z = np.empty_like(angles, dtype=dtype_r2c(angles.dtype))
    z[...] = np.cos(angles) + 1j * np.sin(angles)
    if mag is not None:
        z *= mag
    return z


def griffinlim_update(rebuilt, tprev, S, momentum, eps):
    """One phase update, ``librosa/core/spectrum.py:2896-2902`` (in the precision of ``rebuilt``):
    ``angles = rebuilt - momentum/(1+momentum) tprev; angles /= |angles| + eps; angles *= S``."""
    angles = np.array(rebuilt, copy=True)
    if tprev is not None:
        angles -= (momentum / (1 + momentum)) * tprev
    angles /= np.abs(angles) + eps
    angles *= S
    return angles


def griffinlim(S, *, n_iter=32, hop_length=None, win_length=None, n_fft=None, window="hann", center=True, dtype=None, length=None, pad_mode="constant",
               momentum=0.99, init="random", rng=None):
    """``librosa/core/spectrum.py:2816-2917``: istft <-> stft fixed-point iteration with momentum ("fast" Griffin-Lim)."""
    if not isinstance(rng, np.random.RandomState):
        rng = np.random.default_rng(rng)                                            # :2812-2813
    if momentum > 1:
        warnings.warn(f"Griffin-Lim with momentum={momentum} > 1 can be unstable. Proceed with caution!", stacklevel=2)
    elif momentum < 0:
        raise ParameterError(f"griffinlim() called with momentum={momentum} < 0")
    if n_fft is None:
        n_fft = 2 * (S.shape[-2] - 1)                                               # :2825-2826
    angles = np.empty(S.shape, dtype=dtype_r2c(S.dtype))                            # :2829
    eps = tiny(angles)
    if init == "random":
        angles[:] = phasor(2 * np.pi * rng.random(size=S.shape))                    # :2834
    elif init is None:
        angles[:] = 1.0
    else:
        raise ParameterError(f"init={init} must either None or 'random'")
    tprev = None
    angles *= S                                                                     # :2847
    kw_i = dict(hop_length=hop_length, win_length=win_length, n_fft=n_fft, window=window, center=center, dtype=dtype, length=length)
    for _ in range(n_iter):
        inverse = istft(angles, **kw_i)                                             # :2850-2860
        rebuilt = stft(inverse, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window, center=center, pad_mode=pad_mode)  # :2863-2872
        angles[:] = griffinlim_update(rebuilt, tprev, S, momentum, eps)             # :2875-2880
        tprev = rebuilt                                                             # :2882
    return istft(angles, **kw_i)                                                    # :2885-2895


# ----------------------------------------------------------------------------- phase vocoder / time stretch
def phase_vocoder(D, *, rate=None, t_out=None, kind="linear"):
    """``librosa/core/spectrum.py:1459-1519``: phase advance per output frame = phase difference of the two input frames around
    its (fractional) input time, accumulated; magnitude interpolated (``scipy.interpolate.interp1d``, the reference's own
    call ``:1507-1515``)."""
    import scipy.interpolate

    n_frames = D.shape[-1]
    if (rate is None) == (t_out is None):
        raise ParameterError("Must specify exactly one of `rate` or `t_out`")
    if (rate is not None) and (rate <= 0):
        raise ParameterError(f"rate={rate} must be a positive number")
    if t_out is None:
        t_out = np.arange(0.0, n_frames, rate)
    t_out = np.asarray(t_out, dtype=float)
    if np.any(t_out < 0) or np.any(t_out >= n_frames):
        raise ParameterError("t_out values must be in the range [0, D.shape[-1])")
    i0 = np.floor(t_out).astype(int)                                  # :1491-1492
    i1 = np.minimum(i0 + 1, n_frames - 1)
    ph = np.angle(D)                                                  # :1495
    diff = ph[..., i1] - ph[..., i0]                                  # :1498
    phase = np.empty_like(diff)
    phase[..., 0] = np.angle(D[..., i0[0]])                           # :1503
    phase[..., 1:] = diff[..., :-1]
    np.cumsum(phase, axis=-1, out=phase)                              # :1507
    mag_interp = scipy.interpolate.interp1d(np.arange(n_frames), np.abs(D), kind=kind, axis=-1, fill_value="extrapolate", assume_sorted=True, copy=False)
    return phasor(phase, mag=mag_interp(t_out))                       # :1518


def time_stretch(y, *, rate, **kwargs):
    """``librosa/effects.py:464-484``: stft -> phase_vocoder -> istft(length=round(n / rate))."""
    if rate <= 0:
        raise ParameterError("rate must be a positive number")
    D = stft(y, **kwargs)
    Ds = phase_vocoder(D, rate=rate)
    ikw = {k: v for k, v in kwargs.items() if k in ("hop_length", "win_length", "n_fft", "window", "center")}
    return istft(Ds, dtype=y.dtype, length=round(y.shape[-1] / rate), **ikw)


def pitch_shift(y, *, sr, n_steps, bins_per_octave=12, res_type="fft", scale=False, **kwargs):
    """``librosa/effects.py:573-596``: time_stretch by 2 ** (-n_steps / bins_per_octave), resample from sr / rate back to sr (the
    scipy-backed converters: ``cqt_oracle.resample``), fix_length to the input's length."""
    import cqt_oracle

    if not (isinstance(bins_per_octave, (int, np.integer)) and bins_per_octave > 0):
        raise ParameterError(f"bins_per_octave={bins_per_octave} must be a positive integer.")
    rate = 2.0 ** (-float(n_steps) / bins_per_octave)
    y_shift = cqt_oracle.resample(time_stretch(y, rate=rate, **kwargs), orig_sr=float(sr) / rate, target_sr=sr, res_type=res_type, scale=scale)
    return fix_length(y_shift, y.shape[-1])


# ----------------------------------------------------------------------------- block feeder
def stream_blocks(y, *, block_length, frame_length, hop_length, fill_value=None):
    """Blocks ``librosa.stream`` yields for an already decoded signal ``y`` ((n,) or (channels, n)), stated directly from
    its two constants (``librosa/core/audio.py:409-410``): blocks of ``(block_length-1)*hop + frame_length`` samples,
    ``block_length*hop`` apart, starting at every multiple of the advance below ``n`` (full blocks ``:488-491``, then the
    remainder ``:507-520``, short unless ``fill_value`` pads it)."""
    size = (block_length - 1) * hop_length + frame_length
    advance = block_length * hop_length
    n = y.shape[-1]
    out = []
    for lo in range(0, n, advance):
        blk = y[..., lo : lo + size]
        if blk.shape[-1] < size and fill_value is not None:
            width = [(0, 0)] * (y.ndim - 1) + [(0, size - blk.shape[-1])]
            blk = np.pad(blk, width, mode="constant", constant_values=fill_value)
        out.append(np.array(blk, copy=True))
    return out


# ----------------------------------------------------------------------------- PCEN (the streaming example's consumer)
def maximum_filter1d(x, size, axis):
    """``scipy.ndimage.maximum_filter1d(x, size, axis=axis)`` as the reference calls it (``librosa/core/spectrum.py:2640-2642``:
    default ``mode="reflect"`` = half-sample symmetric extension, ``origin=0``): ``out[i] = max(x[i - size//2 : i - size//2 + size])``."""
    x = np.moveaxis(np.asarray(x), axis, -1)
    n = x.shape[-1]
    left, right = size // 2, size - size // 2 - 1
    idx = np.arange(-left, n + right) % (2 * n)                       # reflect with period 2n: ... b a | a b c ... | c b ...
    idx = np.where(idx < n, idx, 2 * n - 1 - idx)
    ext = x[..., idx]
    out = ext[..., 0:n].copy()
    for j in range(1, size):
        np.maximum(out, ext[..., j : j + n], out=out)
    return np.moveaxis(out, -1, axis)


def lfilter_first_order(b, x, zi, axis):
    """``scipy.signal.lfilter([b], [1, b - 1], x, zi=zi, axis=axis)`` (``:2655``): transposed direct form II in float64,
    ``y[n] = z + b x[n]``, ``z = 0 x[n] - (b - 1) y[n]``; returns ``(y, zf)`` with ``zf`` shaped like ``x`` with ``axis`` of length 1."""
    x = np.moveaxis(np.asarray(x, dtype=np.float64), axis, -1)
    z = np.broadcast_to(np.moveaxis(np.asarray(zi, dtype=np.float64), axis, -1), x.shape[:-1] + (1,))[..., 0].copy()
    y = np.empty_like(x)
    a1 = b - 1.0
    for n in range(x.shape[-1]):
        y[..., n] = z + b * x[..., n]
        z = x[..., n] * 0.0 - y[..., n] * a1
    return np.moveaxis(y, -1, axis), np.moveaxis(z[..., None], -1, axis)


def pcen(S, *, sr=22050, hop_length=512, gain=0.98, bias=2, power=0.5, time_constant=0.400, eps=1e-6, b=None, max_size=1, ref=None, axis=-1, max_axis=None,
         zi=None, return_zf=False):
    """``librosa/core/spectrum.py:2598-2666``."""
    if power < 0:
        raise ParameterError(f"power={power} must be nonnegative")                 # :2598-2599
    if gain < 0:
        raise ParameterError(f"gain={gain} must be non-negative")
    if bias < 0:
        raise ParameterError(f"bias={bias} must be non-negative")
    if eps <= 0:
        raise ParameterError(f"eps={eps} must be strictly positive")
    if time_constant <= 0:
        raise ParameterError(f"time_constant={time_constant} must be strictly positive")
    if not (isinstance(max_size, (int, np.integer)) and max_size > 0):
        raise ParameterError(f"max_size={max_size} must be a positive integer")   # :2613-2614
    if b is None:
        t_frames = time_constant * sr / float(hop_length)                          # :2616-2621
        b = (np.sqrt(1 + 4 * t_frames**2) - 1) / (2 * t_frames**2)
    if not 0 <= b <= 1:
        raise ParameterError(f"b={b} must be between 0 and 1")
    S = np.asarray(S)
    if np.issubdtype(S.dtype, np.complexfloating):                                  # :2626-2633
        warnings.warn("pcen was called on complex input so phase information will be discarded. To suppress this warning, call pcen(np.abs(D)) instead.", stacklevel=2)
        S = np.abs(S)
    if ref is None:                                                                 # :2635-2655
        if max_size == 1:
            ref = S
        elif S.ndim == 1:
            raise ParameterError("Max-filtering cannot be applied to 1-dimensional input")
        else:
            if max_axis is None:
                if S.ndim != 2:
                    raise ParameterError(f"Max-filtering a {S.ndim:d}-dimensional spectrogram requires you to specify max_axis")
                max_axis = np.mod(1 - axis, 2)
            ref = maximum_filter1d(S, max_size, max_axis)
    if zi is None:
        zi = np.empty((1,) * np.ndim(ref))
        zi[:] = scipy.signal.lfilter_zi([b], [1, b - 1])[:]                        # :2649-2652
    S_smooth, zf = lfilter_first_order(b, ref, zi, axis)                            # :2655
    smooth = np.exp(-gain * (np.log(eps) + np.log1p(S_smooth / eps)))               # :2658
    if power == 0:
        S_out = np.log1p(S * smooth)                                                # :2661
    elif bias == 0:
        S_out = np.exp(power * (np.log(S) + np.log(smooth)))                        # :2663
    else:
        S_out = (bias**power) * np.expm1(power * np.log1p(S * smooth / bias))       # :2665
    return (S_out, zf) if return_zf else S_out


# ----------------------------------------------------------------------------- harmonic / percussive separation
def magphase(D, *, power=1):
    """``librosa/core/spectrum.py:1347-1361``: magnitude and unit phasor (1 + 0j where the magnitude is zero)."""
    mag = np.abs(D)
    zeros_to_ones = mag == 0
    mag_nonzero = mag + zeros_to_ones
    phase = np.empty_like(D, dtype=dtype_r2c(D.dtype))
    phase.real = D.real / mag_nonzero + zeros_to_ones
    phase.imag = D.imag / mag_nonzero
    mag **= power
    return mag, phase


def softmask(X, X_ref, *, power=1, split_zeros=False):
    """``librosa/util/utils.py:1895-1932``: ``X**p / (X**p + X_ref**p)`` evaluated relative to the larger of the two."""
    if X.shape != X_ref.shape:
        raise ParameterError(f"Shape mismatch: {X.shape}!={X_ref.shape}")
    if np.any(X < 0) or np.any(X_ref < 0):
        raise ParameterError("X and X_ref must be non-negative")
    if power <= 0:
        raise ParameterError("power must be strictly positive")
    dtype = X.dtype if np.issubdtype(X.dtype, np.floating) else np.float32
    Z = np.maximum(X, X_ref).astype(dtype)
    bad_idx = Z < np.finfo(dtype).tiny
    Z[bad_idx] = 1
    if np.isfinite(power):
        mask = (X / Z) ** power
        ref_mask = (X_ref / Z) ** power
        good_idx = ~bad_idx
        mask[good_idx] /= mask[good_idx] + ref_mask[good_idx]
        mask[bad_idx] = 0.5 if split_zeros else 0.0
    else:
        mask = X > X_ref
    return mask


def hpss(S, *, kernel_size=31, power=2.0, mask=False, margin=1.0):
    """``librosa/decompose.py:371-528`` (body ``:470-528``): median filters along time (harmonic) and frequency (percussive)
    through ``scipy.ndimage.median_filter`` as the reference calls it (``mode="reflect"``), soft masks, masked spectrogram with the
    original phase."""
    from scipy.ndimage import median_filter

    if np.iscomplexobj(S):
        S, phase = magphase(S)
    else:
        phase = 1
    win_harm, win_perc = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size, kernel_size)
    margin_harm, margin_perc = margin if isinstance(margin, (tuple, list)) else (margin, margin)
    if margin_harm < 1 or margin_perc < 1:
        raise ParameterError("Margins must be >= 1.0. A typical range is between 1 and 10.")
    harm_shape = [1] * S.ndim
    harm_shape[-1] = int(win_harm)
    perc_shape = [1] * S.ndim
    perc_shape[-2] = int(win_perc)
    harm = np.empty_like(S)
    harm[:] = median_filter(S, size=harm_shape, mode="reflect")
    perc = np.empty_like(S)
    perc[:] = median_filter(S, size=perc_shape, mode="reflect")
    split_zeros = margin_harm == 1 and margin_perc == 1
    mask_harm = softmask(harm, perc * margin_harm, power=power, split_zeros=split_zeros)
    mask_perc = softmask(perc, harm * margin_perc, power=power, split_zeros=split_zeros)
    if mask:
        return mask_harm, mask_perc
    return ((S * mask_harm) * phase, (S * mask_perc) * phase)


def effects_hpss(y, *, kernel_size=31, power=2.0, mask=False, margin=1.0, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True, pad_mode="constant"):
    """``librosa/effects.py:70-185``: stft -> decompose.hpss -> two istft.  ``window`` is accepted and passed to NONE of the three
    transforms (``:161-183``), exactly as the reference does: all of them run with the default window."""
    del window
    D = stft(y, n_fft=n_fft, hop_length=hop_length, win_length=win_length, center=center, pad_mode=pad_mode)
    Dh, Dp = hpss(D, kernel_size=kernel_size, power=power, mask=mask, margin=margin)
    ikw = dict(dtype=y.dtype, n_fft=n_fft, hop_length=hop_length, win_length=win_length, center=center, length=y.shape[-1])
    return istft(Dh, **ikw), istft(Dp, **ikw)


# ----------------------------------------------------------------------------- synthetic inputs
def config_input(batch, n=661500, sr=22050, seed=440, first_clip=0):
    """SURVEY.md 8(d) config-2/3 generator: 0.1*noise + 0.5*sin(2 pi f_i t), f_i = 110*2^((i%72)/12).

    Clip ``i`` depends only on (seed, i), so shards generated on different ranks agree with the
    unsharded batch.
    """
    t = np.arange(n, dtype=np.float64) / sr
    out = np.empty((batch, n), dtype=np.float32)
    for b in range(batch):
        i = first_clip + b
        rng = np.random.default_rng([seed, i])
        f = 110.0 * 2.0 ** ((i % 72) / 12.0)
        out[b] = np.clip(0.1 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * f * t), -1.0, 1.0).astype(np.float32)
    return out


def config1_input():
    """10 s, 440 Hz sine at 22.05 kHz, f32."""
    return np.sin(2 * np.pi * 440.0 * np.arange(220500) / 22050.0).astype(np.float32)
