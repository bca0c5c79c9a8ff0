"""CPU oracle for the constant-Q / variable-Q transform ("CQT/VQT octave recursion").

TEST INFRASTRUCTURE ONLY (same rules as ``stft_oracle.py``: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` may import it).  A NumPy restatement of ``librosa/core/constantq.py:820-1122`` (``vqt``; ``cqt`` ``:42-225`` is the
``gamma=0`` case) and of the table builders under it: ``filters.wavelet_lengths`` / ``wavelet`` (``librosa/filters.py:424-722``),
``interval_frequencies`` (``core/intervals.py:28-135``, equal temperament or an explicit interval list), ``util.sparsify_rows``
(``util/utils.py:1500-1597``), ``audio.resample`` (``core/audio.py:536-724``).

Third-party arithmetic used through the same library the reference calls: ``scipy.fft.fft`` (filter basis, ``constantq.py:1160``),
``scipy.sparse`` (``util/utils.py:1597``; the projection ``constantq.py:1218``), ``scipy.signal.resample_poly`` / ``resample``
(``core/audio.py:676-693``).  The reference's default resampler, ``soxr_hq`` (``soxr`` package), is NOT installed in the build image:
the reference itself cannot run with its default ``res_type`` here, so parity is pinned for ``res_type`` in {"polyphase", "fft",
"scipy"} (goldens + live reference) and **unpinned for the soxr family**.
"""
from __future__ import annotations

import numpy as np
import scipy.fft
import scipy.signal
import scipy.sparse

import stft_oracle as O
from stft_oracle import ParameterError

# librosa/filters.py:73-113 (the entries a CQT can ask for by name; other names are computed like :905-909)
WINDOW_BANDWIDTHS = {"hann": 1.50018310546875, "hamming": 1.3629455320350348, "blackman": 1.7269681554262326, "blackmanharris": 2.0045975283585014, "ones": 1.0, "boxcar": 1.0,
                     "bartlett": 1.3334961334912805, "triang": 1.3331706523555851, "nuttall": 1.9763500280946082, "bohman": 1.7859588613860062, "flattop": 2.7762255046484143,
                     "cosine": 1.2337005350199792, "parzen": 1.9174603174603191, "barthann": 1.4560255965133932}

C1_HZ = 440.0 * (2.0 ** ((24 - 69) / 12))  # note_to_hz("C1"), librosa/core/convert.py:573-620 -> midi_to_hz(24)


def window_bandwidth(window, n=1000):
    """``librosa/filters.py:838-911``: equivalent noise bandwidth of a window, in FFT bins."""
    key = window.__name__ if hasattr(window, "__name__") else window
    if isinstance(key, str) and key in WINDOW_BANDWIDTHS:
        return WINDOW_BANDWIDTHS[key]
    win = O.get_window(window, n)
    return n * np.sum(win**2) / (np.sum(win) ** 2 + O.tiny(win))


def interval_frequencies(n_bins, *, fmin, intervals="equal", bins_per_octave=12, tuning=0.0, sort=True):
    """``librosa/core/intervals.py:100-135`` for ``intervals="equal"`` or an explicit list of ratios in [1, 2)."""
    if isinstance(intervals, str):
        if intervals != "equal":
            raise ParameterError(f"intervals={intervals!r}: only 'equal' or an explicit interval list (the notation tables are outside the path)")
        ratios = 2.0 ** ((tuning + np.arange(0, bins_per_octave, dtype=float)) / bins_per_octave)
    else:
        ratios = np.array(intervals)
        bins_per_octave = len(ratios)
    n_octaves = np.ceil(n_bins / bins_per_octave)
    all_ratios = np.multiply.outer(2.0 ** np.arange(n_octaves), ratios).flatten()[:n_bins]
    if sort:
        all_ratios = np.sort(all_ratios)
    return all_ratios * fmin


def relative_bandwidth(freqs):
    """``librosa/filters.py:555-585``."""
    if len(freqs) <= 1:
        raise ParameterError(f"2 or more frequencies are required to compute bandwidths. Given freqs={freqs}")
    bpo = np.empty_like(freqs)
    logf = np.log2(freqs)
    bpo[0] = 1 / (logf[1] - logf[0])
    bpo[-1] = 1 / (logf[-1] - logf[-2])
    bpo[1:-1] = 2 / (logf[2:] - logf[:-2])
    return (2.0 ** (2 / bpo) - 1) / (2.0 ** (2 / bpo) + 1)


def wavelet_lengths(*, freqs, sr=22050, window="hann", filter_scale=1, gamma=0, alpha=None):
    """``librosa/filters.py:514-551``: fractional filter lengths and the highest frequency any filter reaches."""
    freqs = np.asarray(freqs)
    if filter_scale <= 0:
        raise ParameterError(f"filter_scale={filter_scale} must be positive")
    if gamma is not None and gamma < 0:
        raise ParameterError(f"gamma={gamma} must be non-negative")
    if np.any(freqs <= 0):
        raise ParameterError("frequencies must be strictly positive")
    if len(freqs) > 1 and np.any(freqs[:-1] > freqs[1:]):
        raise ParameterError(f"Frequency array={freqs} must be in strictly ascending order")
    alpha = relative_bandwidth(freqs) if alpha is None else np.asarray(alpha)
    gamma_ = alpha * 24.7 / 0.108 if gamma is None else gamma
    Q = float(filter_scale) / alpha
    f_cutoff = max(freqs * (1 + 0.5 * window_bandwidth(window) / Q) + 0.5 * gamma_)
    lengths = Q * sr / (freqs + gamma_ / alpha)
    return lengths, f_cutoff


def _float_window(window, n):
    """``librosa/filters.py:397-420``: a window of fractional length ``n`` = the integer window of floor(n), zero beyond."""
    n_min, n_max = int(np.floor(n)), int(np.ceil(n))
    w = O.get_window(window, n_min)
    if len(w) < n_max:
        w = np.pad(w, [(0, n_max - len(w))], mode="constant")
    w[n_min:] = 0.0
    return w


def wavelet(*, freqs, sr=22050, window="hann", filter_scale=1, pad_fft=True, norm=1, dtype=np.complex64, gamma=0, alpha=None):
    """``librosa/filters.py:693-722``: time-domain filters (complex exponentials under the window), centred in a power-of-two frame."""
    lengths, _ = wavelet_lengths(freqs=freqs, sr=sr, window=window, filter_scale=filter_scale, gamma=gamma, alpha=alpha)
    filters = []
    for ilen, freq in zip(lengths, freqs):
        sig = O.phasor(np.arange(-ilen // 2, ilen // 2, dtype=float) * 2 * np.pi * freq / sr)
        sig *= _float_window(window, len(sig))
        filters.append(O.normalize(sig, norm=norm, axis=0))
    max_len = int(2.0 ** (np.ceil(np.log2(max(lengths))))) if pad_fft else int(np.ceil(max(lengths)))
    return np.asarray([O.pad_center(f, max_len) for f in filters], dtype=dtype), lengths


def sparsify_rows(x, *, quantile=0.01, dtype=None):
    """``librosa/util/utils.py:1567-1597``: per row, zero the smallest entries holding ``quantile`` of the row's L1 mass."""
    if x.ndim == 1:
        x = x.reshape((1, -1))
    if not 0.0 <= quantile < 1:
        raise ParameterError(f"Invalid quantile {quantile:.2f}")
    out_dtype = np.dtype(x.dtype if dtype is None else dtype)
    mags = np.abs(x)
    norms = np.sum(mags, axis=1, keepdims=True)
    mag_sort = np.sort(mags, axis=1)
    cumulative_mag = np.cumsum(mag_sort / norms, axis=1)
    threshold_idx = np.argmin(cumulative_mag < quantile, axis=1)
    thresh = mag_sort[np.arange(x.shape[0]), threshold_idx]
    mask = mags >= thresh[:, np.newaxis]
    return scipy.sparse.csr_array((x * mask).astype(out_dtype, copy=False))


def vqt_filter_fft(sr, freqs, filter_scale, norm, sparsity, window="hann", gamma=0.0, dtype=np.complex64, alpha=None):
    """``librosa/core/constantq.py:1137-1165``: frequency-domain basis of one octave, non-negative frequencies, sparsified."""
    basis, lengths = wavelet(freqs=freqs, sr=sr, filter_scale=filter_scale, norm=norm, pad_fft=True, window=window, gamma=gamma, alpha=alpha)
    n_fft = basis.shape[1]
    basis *= lengths[:, np.newaxis] / float(n_fft)
    fft_basis = scipy.fft.fft(basis, n=n_fft, axis=1)[:, : (n_fft // 2) + 1]
    return sparsify_rows(fft_basis, quantile=sparsity, dtype=dtype), n_fft, lengths


def resample(y, *, orig_sr, target_sr, res_type, scale=False):
    """``librosa/core/audio.py:660-724`` along the last axis for the scipy-backed resamplers; ``fix=True``."""
    if orig_sr == target_sr:
        return y
    ratio = float(target_sr) / orig_sr
    n_samples = int(np.ceil(y.shape[-1] * ratio))
    if res_type in ("scipy", "fft"):
        y_hat = scipy.signal.resample(y, n_samples, axis=-1)
    elif res_type == "polyphase":
        if int(orig_sr) != orig_sr or int(target_sr) != target_sr:
            raise ParameterError("polyphase resampling is only supported for integer-valued sampling rates.")
        gcd = np.gcd(int(orig_sr), int(target_sr))
        y_hat = scipy.signal.resample_poly(y, int(target_sr) // gcd, int(orig_sr) // gcd, axis=-1)
    else:
        raise ParameterError(f"res_type={res_type!r}: the oracle restates only the scipy-backed resamplers (soxr / samplerate / resampy are not in this image)")
    y_hat = O.fix_length(y_hat, n_samples)
    if scale:
        y_hat /= np.sqrt(ratio)
    return np.asarray(y_hat, dtype=y.dtype)


def num_two_factors(x):
    """``librosa/core/constantq.py:1271-1284``."""
    n = 0
    while x > 0 and x % 2 == 0:
        n += 1
        x //= 2
    return n


def early_downsample_count(nyquist, filter_cutoff, hop_length, n_octaves):
    """``librosa/core/constantq.py:1226-1232``."""
    count1 = max(0, int(np.ceil(np.log2(nyquist / filter_cutoff)) - 1) - 1)
    count2 = max(0, num_two_factors(hop_length) - n_octaves + 1)
    return min(count1, count2)


def cqt_response(y, n_fft, hop_length, fft_basis, mode, dtype=None):
    """``librosa/core/constantq.py:1197-1223``: rectangular-window STFT, then the sparse basis applied to every frame."""
    D = O.stft(y, n_fft=n_fft, hop_length=hop_length, window="ones", pad_mode=mode, dtype=dtype)
    Dr = D.reshape((-1, D.shape[-2], D.shape[-1]))
    out = np.empty((Dr.shape[0], fft_basis.shape[0], Dr.shape[-1]), dtype=D.dtype)
    for i in range(Dr.shape[0]):
        out[i] = fft_basis.dot(Dr[i])
    return out.reshape(D.shape[:-2] + (fft_basis.shape[0], D.shape[-1]))


def clip_freqs(freqs, window, filter_scale, gamma, sr):
    """``librosa/core/constantq.py:1630-1657``: the longest prefix of ``freqs`` whose filters all stay below the Nyquist frequency."""
    logf = np.log2(freqs)
    window_bw = window_bandwidth(window)
    bpo = 1 / np.diff(logf, prepend=0)
    bpo[0] = 1 / (logf[1] - logf[0])
    alpha = (2.0 ** (2 / bpo) - 1) / (2.0 ** (2 / bpo) + 1)
    gamma_ = alpha * 24.7 / 0.108 if gamma is None else gamma
    Q = float(filter_scale) / alpha
    f_cutoff = np.maximum.accumulate(freqs * (1 + 0.5 * window_bw / Q) + 0.5 * gamma_)
    idx = np.searchsorted(f_cutoff, sr / 2.0, side="left")
    if idx < 1:
        raise ParameterError(f"Unable to construct wavelet basis for fmin={freqs[0]:.2f} Hz and sr={sr:.2f} Hz.")
    return freqs[:idx]


def vqt(y, *, sr=22050, hop_length=512, fmin=None, n_bins=84, intervals="equal", gamma=None, bins_per_octave=12, tuning=0.0, filter_scale=1, norm=1, sparsity=0.01,
        window="hann", scale=True, pad_mode="constant", res_type="polyphase", dtype=None):
    """``librosa/core/constantq.py:977-1122``.  ``tuning=None`` (pitch tracking) is the caller's."""
    if not isinstance(intervals, str):
        bins_per_octave = len(intervals)
    if fmin is None:
        fmin = C1_HZ
    if tuning is None:
        raise ParameterError("tuning=None needs estimate_tuning (pitch tracking), which is outside the path")
    if dtype is None:
        dtype = O.dtype_r2c(y.dtype)
    fmin = fmin * 2.0 ** (tuning / bins_per_octave)
    if fmin >= sr / 2:
        raise ParameterError(f"fmin={fmin} must be less than sr/2={sr/2}")
    auto_n_bins = n_bins is None
    if auto_n_bins:                                                          # :1000-1007: one octave more than fits, clipped below
        n_bins = int(np.ceil(bins_per_octave * (np.log2(sr) - np.log2(fmin))))
    freqs = interval_frequencies(n_bins, fmin=fmin, intervals=intervals, bins_per_octave=bins_per_octave, sort=True)
    if auto_n_bins:
        freqs = clip_freqs(freqs, window, filter_scale, gamma, sr)
        n_bins = len(freqs)
    if n_bins == 1:
        r = 2 ** (1 / bins_per_octave)
        alpha = np.atleast_1d((r**2 - 1) / (r**2 + 1))                       # :1577-1597
    else:
        alpha = relative_bandwidth(freqs)
    lengths, filter_cutoff = wavelet_lengths(freqs=freqs, sr=sr, window=window, filter_scale=filter_scale, gamma=gamma, alpha=alpha)
    nyquist = sr / 2.0
    if filter_cutoff > nyquist:
        raise ParameterError(f"Wavelet basis with max frequency={np.max(freqs[-bins_per_octave:])} would exceed the Nyquist frequency={nyquist}. "
                             "Try reducing the number of frequency bins.")
    n_octaves = int(np.ceil(float(n_bins) / bins_per_octave))
    n_filters = min(bins_per_octave, n_bins)
    # early downsampling (:1235-1268)
    count = early_downsample_count(nyquist, filter_cutoff, hop_length, n_octaves)
    if count > 0:
        factor = 2**count
        hop_length //= factor
        if y.shape[-1] < factor:
            raise ParameterError(f"Input signal length={len(y):d} is too short for {n_octaves:d}-octave CQT")
        y = resample(y, orig_sr=factor, target_sr=1, res_type=res_type, scale=True)
        if not scale:
            y *= np.sqrt(factor)
        sr = sr / float(factor)
    resp = []
    my_y, my_sr, my_hop = y, sr, hop_length
    for i in range(n_octaves):                                               # :1054-1099
        sl = slice(-n_filters, None) if i == 0 else slice(-n_filters * (i + 1), -n_filters * i)
        fft_basis, n_fft, _ = vqt_filter_fft(my_sr, freqs[sl], filter_scale, norm, sparsity, window=window, gamma=gamma, dtype=dtype, alpha=alpha[sl])
        fft_basis[:] *= np.sqrt(sr / my_sr)
        resp.append(cqt_response(my_y, n_fft, my_hop, fft_basis, pad_mode, dtype=dtype))
        if i < n_octaves - 1:
            f_max_next = freqs[sl.start - 1]
            if my_hop % 2 == 0 and f_max_next <= my_sr / 5:
                my_hop //= 2
                my_sr /= 2.0
                my_y = resample(my_y, orig_sr=2, target_sr=1, res_type=res_type, scale=True)
    # trim and stack, lowest octave first (:1168-1194)
    max_col = min(c.shape[-1] for c in resp)
    V = np.empty(resp[0].shape[:-2] + (n_bins, max_col), dtype=dtype)
    end = n_bins
    for c in resp:
        n_oct = c.shape[-2]
        if end < n_oct:
            V[..., :end, :] = c[..., -end:, :max_col]
        else:
            V[..., end - n_oct : end, :] = c[..., :max_col]
        end -= n_oct
    if scale:
        lengths, _ = wavelet_lengths(freqs=freqs, sr=sr, window=window, filter_scale=filter_scale, gamma=gamma, alpha=alpha)
        V /= np.sqrt(lengths)[:, np.newaxis]
    return V


def cqt(y, *, sr=22050, hop_length=512, fmin=None, n_bins=84, bins_per_octave=12, tuning=0.0, filter_scale=1, norm=1, sparsity=0.01, window="hann", scale=True,
        pad_mode="constant", res_type="polyphase", dtype=None):
    """``librosa/core/constantq.py:204-225``."""
    return vqt(y, sr=sr, hop_length=hop_length, fmin=fmin, n_bins=n_bins, intervals="equal", gamma=0, bins_per_octave=bins_per_octave, tuning=tuning, filter_scale=filter_scale,
               norm=norm, sparsity=sparsity, window=window, scale=scale, pad_mode=pad_mode, res_type=res_type, dtype=dtype)
