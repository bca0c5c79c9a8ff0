"""Constant-Q and variable-Q transforms on the device: drop-ins for ``librosa.cqt`` / ``librosa.vqt``
(``librosa/core/constantq.py:42-225, 820-1122``; SURVEY.md 8f rank 4, "CQT/VQT octave recursion").

The recursion the reference runs on the host -- per octave a rectangular-window STFT (``__cqt_response``, ``:1197-1223``), the
sparse frequency-domain filter basis applied to every frame (``:1218``), then ``audio.resample`` by two (``:1095-1098``) -- stays in
HBM from the first sample to the stacked result: ``lra_stft_exec`` -> ``lra_cqt_project_exec`` (projection + length scaling + the
stacking of ``__trim_stack``) -> ``lra_fir_decimate_exec``, one octave after the other on one stream.  The host builds the tables
(frequencies, filter lengths, the wavelet basis and its FFT, the sparsification: float64 / complex64 recipes identical to the
reference's, see ``librosa_amd.filters.wavelet``) once per parameter set and keeps them.

Resampling.  The reference's default ``res_type="soxr_hq"`` calls the ``soxr`` package, which is not in the build image (the
reference itself cannot run its default there).  ``res_type="polyphase"`` is reproduced exactly (scipy's Kaiser-5 design and
summation order: bit-identical signals between octaves).  The band-limited resamplers with an exact rate ratio (the ``soxr_*``
family, resampy's ``kaiser_*``, samplerate's ``sinc_*``) run this library's own linear-phase decimator with soxr-HQ's band edges
(pass band to 0.913 of the new Nyquist, 125 dB stop band from the new Nyquist on): the constant-Q filters of the next octave lie
below 0.8 of the new Nyquist (``:1093``), inside every such filter's pass band, so the transforms agree to the resamplers'
pass-band ripple -- 2.6e-3 of the peak against the reference's ``"polyphase"`` result, whose Kaiser-5 filter droops most; parity
for this family is that tolerance statement, not pinned against soxr.  ``"fft"`` / ``"scipy"`` (whole-signal
``scipy.signal.resample``) run as two rocFFT transforms of the whole signal per halving (``lra_resample_fft_exec``), pinned against
the reference like ``"polyphase"``.  The non-band-limited ``"linear"`` / ``"zero_order_hold"`` are not provided.
"""
from __future__ import annotations

import functools
import warnings

import numpy as np
import scipy.fft
import scipy.signal

from .. import _arrays
from .. import filters
from ..util import utils as util
from ..util.exceptions import ParameterError
from ..util.utils import is_torch_tensor
from .audio import _rational_filter
from .intervals import interval_frequencies
from .spectrum import _DEVICE_PAD_MODES, _all_finite, _as_like

# One launch per octave (csrc/lra_mixed.h, mixed_cqt_kernel: frames + rectangular-window transform + sparse projection + scaling + stacking, the
# octave's spectra never leave LDS) where the octave's frame length is one the kernel is built for; False keeps the round-3 pair of launches
# (forward kernel -> HBM -> cqt_project_kernel) everywhere.  A test / measurement switch.
FUSED_OCTAVES = True
# The octave transforms on the context's side stream, beside the chain of halvings (lra_ctx_side); False: everything on one stream.
OVERLAP_OCTAVES = True
# the whole octave recursion in one native call (lra_cqt_recursion_exec) where it applies; False = the per-octave calls from Python
NATIVE_RECURSION = True

__all__ = ["cqt", "vqt"]

_C1_HZ = 440.0 * (2.0 ** ((24 - 69) / 12))  # note_to_hz("C1") (core/convert.py:573-620): MIDI note 24




def _two_factors(x):
    """How many times 2 divides ``x`` (``constantq.py:1271-1284``)."""
    n = 0
    while x > 0 and x % 2 == 0:
        n += 1
        x //= 2
    return n


def _decimator(down, res_type, real):
    """(taps incl. leading zeros, first output's offset) of the FIR that decimates by ``down``: ``audio._rational_filter(1, down, ...)``
    (``"polyphase"``: scipy's design, reproduced exactly; other names: the library's own band-limited design)."""
    return _rational_filter(1, int(down), res_type, real)


_SPECTRAL = ("fft", "scipy")   # scipy.signal.resample: whole-signal Fourier resampling (core/audio.py:672-675)


def _auto_n_bins(sr, fmin, intervals, gamma, bins_per_octave, filter_scale, window):
    """``n_bins=None``: as many bins as stay below the Nyquist frequency (``constantq.py:1000-1007, 1017-1019`` with ``__clip_freqs``
    ``:1599-1657``: one octave more than fits, then the longest prefix whose filters' upper edges are below ``sr / 2``)."""
    n_over = int(np.ceil(bins_per_octave * (np.log2(sr) - np.log2(fmin))))
    freqs = interval_frequencies(n_over, fmin=fmin, intervals=intervals if isinstance(intervals, str) else list(intervals), bins_per_octave=bins_per_octave, sort=True)
    log_f = np.log2(freqs)
    density = 1 / np.diff(log_f, prepend=0)
    density[0] = 1 / (log_f[1] - log_f[0])
    step = 2.0 ** (2 / density)
    alpha = (step - 1) / (step + 1)
    offset = alpha * 24.7 / 0.108 if gamma is None else gamma
    q = float(filter_scale) / alpha
    upper_edge = np.maximum.accumulate(freqs * (1 + 0.5 * filters.window_bandwidth(window) / q) + 0.5 * offset)
    keep = int(np.searchsorted(upper_edge, sr / 2.0, side="left"))
    if keep < 1:
        raise ParameterError(f"Unable to construct wavelet basis for fmin={freqs[0]:.2f} Hz and sr={sr:.2f} Hz.")
    return keep


@functools.lru_cache(maxsize=16)
def _plan(sr, hop_length, fmin, n_bins, intervals, gamma, bins_per_octave, filter_scale, norm, sparsity, window, scale, cplx_str):
    """Everything that does not depend on the signal: per-octave FFT size, hop, sparse basis (CSR arrays), row selection and
    scaling of the stacked result, and the downsampling schedule (``constantq.py:1000-1099`` without the data)."""
    cplx = np.dtype(cplx_str)
    freqs = interval_frequencies(n_bins, fmin=fmin, intervals=intervals if isinstance(intervals, str) else list(intervals), bins_per_octave=bins_per_octave, sort=True)
    if n_bins == 1:
        r = 2 ** (1 / bins_per_octave)
        alpha = np.atleast_1d((r**2 - 1) / (r**2 + 1))                       # :1577-1597
    else:
        alpha = filters._relative_bandwidth(freqs=freqs)
    lengths, reach = filters.wavelet_lengths(freqs=freqs, sr=sr, window=window, filter_scale=filter_scale, gamma=gamma, alpha=alpha)
    nyquist = sr / 2.0
    if reach > nyquist:
        raise ParameterError(f"Wavelet basis with max frequency={np.max(freqs[-bins_per_octave:])} would exceed the Nyquist frequency={nyquist}. "
                             "Try reducing the number of frequency bins.")
    n_octaves = int(np.ceil(float(n_bins) / bins_per_octave))
    per_octave = min(bins_per_octave, n_bins)
    early = min(max(0, int(np.ceil(np.log2(nyquist / reach)) - 1) - 1), max(0, _two_factors(hop_length) - n_octaves + 1))   # :1226-1232
    sr0 = sr / float(2**early)
    hop0 = hop_length // (2**early)
    octaves, top = [], n_bins
    cur_sr, cur_hop = sr0, hop0
    for i in range(n_octaves):
        sl = slice(-per_octave, None) if i == 0 else slice(-per_octave * (i + 1), -per_octave * i)
        f_oct, a_oct = freqs[sl], alpha[sl]
        basis, f_lengths = filters.wavelet(freqs=f_oct, sr=cur_sr, filter_scale=filter_scale, norm=norm, pad_fft=True, window=window, gamma=gamma, alpha=a_oct)
        n_fft = basis.shape[1]
        basis *= f_lengths[:, np.newaxis] / float(n_fft)                      # :1157
        spectrum = scipy.fft.fft(basis, n=n_fft, axis=1)[:, : (n_fft // 2) + 1]
        sparse = util.sparsify_rows(spectrum, quantile=sparsity, dtype=cplx)
        sparse[:] *= np.sqrt(sr0 / cur_sr)                                    # :1080
        sparse.sort_indices()
        rows = sparse.shape[0]
        keep = min(rows, top)                                                 # __trim_stack (:1183-1192): the last octave may keep only its highest filters
        octaves.append(dict(n_fft=n_fft, hop=cur_hop, row_ptr=np.ascontiguousarray(sparse.indptr, dtype=np.int32), col=np.ascontiguousarray(sparse.indices, dtype=np.int32),
                            val=np.ascontiguousarray(sparse.data, dtype=cplx), row0=rows - keep, n_rows=keep, bin0=top - keep, halve=False))
        top -= keep
        if i < n_octaves - 1 and cur_hop % 2 == 0 and freqs[sl.start - 1] <= cur_sr / 5:     # :1093
            octaves[-1]["halve"] = True
            cur_hop //= 2
            cur_sr /= 2.0
    sqrt_len = None
    if scale:
        l0, _ = filters.wavelet_lengths(freqs=freqs, sr=sr0, window=window, filter_scale=filter_scale, gamma=gamma, alpha=alpha)   # :1103-1112
        sqrt_len = np.sqrt(l0)
    return dict(early=early, octaves=octaves, sqrt_len=sqrt_len)


def vqt(y, *, sr=22050, hop_length=512, fmin=None, n_bins=84, intervals="equal", gamma=None, bins_per_octave=12, tuning=0.0, filter_scale=1, norm=1, sparsity=0.01,
        window="hann", scale=True, pad_mode="constant", res_type="soxr_hq", dtype=None, check_finite=True):
    """Variable-Q transform; drop-in for ``librosa.vqt`` (``librosa/core/constantq.py:820-1122``).

    Returns ``(..., n_bins, n_frames)`` complex (complex64 for float32 audio).  ``y`` may be a device tensor (a device tensor is
    returned; ``check_finite=False`` -- an extension, as for ``stft`` -- then skips ``valid_audio``'s finite test, whose device flag costs one
    host synchronisation per call: back-to-back calls otherwise cannot overlap their launches with the previous call's kernels).  Not provided: ``tuning=None`` (needs the pitch tracker behind ``estimate_tuning``), named just-intonation interval
    sets (``intervals`` must be ``"equal"`` or an explicit list).  See the module docstring for ``res_type``.
    """
    if not isinstance(intervals, str):
        intervals = tuple(float(v) for v in intervals)
        bins_per_octave = len(intervals)
    elif intervals != "equal":
        raise ParameterError(f"intervals={intervals!r}: librosa_amd.vqt takes 'equal' or an explicit list of intervals")
    if fmin is None:
        fmin = _C1_HZ
    if tuning is None:
        raise ParameterError("tuning=None (automatic tuning estimation) is not provided by librosa_amd; pass a number")
    if not util.is_positive_int(hop_length):
        raise ParameterError(f"hop_length={hop_length} must be a positive integer")
    on_device = is_torch_tensor(y)
    if on_device:
        if not y.is_floating_point():
            raise ParameterError("Audio data must be floating-point")
    else:
        util.valid_audio(y)
    in_dtype = _arrays.numpy_dtype_of(y)
    cplx = np.dtype(util.dtype_r2c(in_dtype)) if dtype is None else np.dtype(dtype)
    if cplx.kind != "c":
        raise ParameterError(f"dtype={cplx} must be a complex type")
    real = np.dtype(np.float64) if cplx == np.complex128 else np.dtype(np.float32)
    fmin = fmin * 2.0 ** (tuning / bins_per_octave)
    if fmin >= sr / 2:
        raise ParameterError(f"fmin={fmin} must be less than sr/2={sr/2}")
    if n_bins is None:
        n_bins = _auto_n_bins(float(sr), float(fmin), intervals, None if gamma is None else float(gamma), int(bins_per_octave), float(filter_scale), window)
    if not (isinstance(pad_mode, str) and pad_mode in _DEVICE_PAD_MODES):
        raise ParameterError(f"pad_mode={pad_mode!r} is not supported by librosa_amd.vqt")
    if res_type not in _SPECTRAL:
        _decimator(2, res_type, real)  # validates res_type (also when no octave needs a decimation)
    plan_key = (float(sr), int(hop_length), float(fmin), int(n_bins), intervals, None if gamma is None else float(gamma), int(bins_per_octave), float(filter_scale),
                None if norm is None else float(norm), float(sparsity), window, bool(scale), cplx.str)
    try:
        plan = _plan(*plan_key)
    except TypeError:  # unhashable window specification: build without the caches (host tables here, device copies below)
        plan = _plan.__wrapped__(*plan_key)
        plan_key = None
    octaves = plan["octaves"]
    n = int(y.shape[-1])
    lead = tuple(int(s) for s in y.shape[:-1])
    factor = 2 ** plan["early"]
    if plan["early"] and n < factor:
        raise ParameterError(f"Input signal length={n:d} is too short for {len(octaves):d}-octave CQT")
    # signal length and frame count of every octave (centred frames: 1 + n // hop)
    lens = [-(-n // factor) if plan["early"] else n]
    for o in octaves[:-1]:
        lens.append(-(-lens[-1] // 2) if o["halve"] else lens[-1])
    frames = [1 + ln // o["hop"] for ln, o in zip(lens, octaves)]
    n_frames = min(frames)
    sess = _arrays.Session(y if on_device else np.empty(0))
    overlapped = False
    try:
        ctx = sess.ctx
        overlapped = OVERLAP_OCTAVES
        y_ptr, batch, _, _ = sess.input_2d(y, real)
        check = on_device and bool(check_finite)
        if check:
            ctx.nonfinite_reset()

        def table(name, a, dtype):
            """Device copy of a host table: kept in the context under the plan's key, or uploaded for this call only."""
            if plan_key is None:
                return sess.input_raw(_as_like(sess, a), dtype)
            return ctx.device_table(("cqt", plan_key, name, np.dtype(dtype).str), lambda: np.ascontiguousarray(a, dtype=dtype))

        def decimator(down):
            taps, first = _decimator(down, res_type, real)
            return ctx.device_table(("fir", down, res_type, real.str), lambda: taps), len(taps), first

        def shorten(src, n_from, n_to, down, extra):
            """audio.resample(orig_sr=down, target_sr=1, scale=True) of the (batch, n_from) signal at ``src`` into scratch; ``extra``: a factor on top."""
            dst = sess.scratch(batch * n_to * real.itemsize)
            if res_type in _SPECTRAL:
                ctx.resample_fft_exec(src, dst, batch, n_from, n_to, np.sqrt(float(down)) * extra, real)
            else:
                taps_ptr, n_taps, first = decimator(down)
                ctx.fir_decimate_exec(src, dst, batch, n_from, n_to, taps_ptr, n_taps, down, first, np.sqrt(1.0 / down), extra, real)
            return dst

        if plan["early"]:
            # resample(scale=True) divides by sqrt(1 / factor); an unscaled transform multiplies by sqrt(factor) on top (:1254-1264)
            y_ptr = shorten(y_ptr, n, lens[0], factor, 1.0 if scale else np.sqrt(factor))
        out_ptr, handle = sess.output((batch, n_frames, n_bins), cplx)
        d_ptr = None  # spectrum scratch of the unfused octaves (frame lengths beyond the fused kernel's), allocated on first need
        sqrt_len_ptr = table("sqrt_len", plan["sqrt_len"], np.float64) if plan["sqrt_len"] is not None else None
        # One native call for the whole recursion where every octave has a fused kernel and the halvings are FIR decimations (round 5: the Python
        # loop below spent 0.26 ms in ~40 ctypes calls per transform, more than half of what the 13 launches take on the device)
        native = NATIVE_RECURSION and FUSED_OCTAVES and res_type not in _SPECTRAL and hasattr(ctx, "cqt_recursion_exec") and all(ctx.cqt_octave_supported(o["n_fft"]) for o in octaves)
        if native:
            for i, o in enumerate(octaves):
                if o["n_fft"] > lens[i]:
                    warnings.warn(f"n_fft={o['n_fft']} is too large for input signal of length={lens[i]}", stacklevel=3)
            arr = (_arrays._native.CqtOctave * len(octaves))()
            for i, o in enumerate(octaves):
                arr[i].n_fft, arr[i].hop, arr[i].bin0, arr[i].row0, arr[i].n_rows = o["n_fft"], o["hop"], o["bin0"], o["row0"], o["n_rows"]
                arr[i].halve, arr[i].n = int(bool(o["halve"]) and i + 1 < len(octaves)), lens[i]
                arr[i].row_ptr, arr[i].col, arr[i].val = table(f"row_ptr{i}", o["row_ptr"], np.int32), table(f"col{i}", o["col"], np.int32), table(f"val{i}", o["val"], cplx)
            need = sum(-(-batch * lens[i + 1] * real.itemsize // 256) * 256 for i in range(len(octaves) - 1) if arr[i].halve)
            taps_ptr, n_taps, first = decimator(2) if need else (None, 0, 0)
            ctx.cqt_recursion_exec(y_ptr, batch, arr, pad_mode, sqrt_len_ptr, out_ptr, n_frames, n_bins, taps_ptr, n_taps, first, sess.scratch(need) if need else None, need, OVERLAP_OCTAVES, real)
        for i, o in enumerate(octaves if not native else ()):
            n_fft, hop = o["n_fft"], o["hop"]
            if n_fft > lens[i]:   # the warning of the reference's stft (core/spectrum.py:267-271), once per octave it applies to
                warnings.warn(f"n_fft={n_fft} is too large for input signal of length={lens[i]}", stacklevel=3)
            csr = (table(f"row_ptr{i}", o["row_ptr"], np.int32), table(f"col{i}", o["col"], np.int32), table(f"val{i}", o["val"], cplx))
            scl = (sqrt_len_ptr + 8 * o["bin0"]) if sqrt_len_ptr else None
            # The octave's transform goes to the context's side stream (behind the halving that made its signal), the halving for the next
            # octave stays on the main stream: the two chains of small launches overlap instead of alternating (joined after the loop).
            if OVERLAP_OCTAVES:
                ctx.side(ctx.SIDE_FORK)
            if FUSED_OCTAVES and ctx.cqt_octave_supported(n_fft):
                # one launch per octave: frames, rectangular-window transform, projection, scaling and stacking; the octave's spectra stay in LDS
                ctx.cqt_octave_exec(y_ptr, batch, lens[i], lens[i], n_fft, hop, pad_mode, csr[0], csr[1], csr[2], scl, out_ptr, n_frames, n_bins, o["bin0"], o["row0"], o["n_rows"], real)
            else:
                if d_ptr is None:
                    d_ptr = sess.scratch(max(f * (oo["n_fft"] // 2 + 1) for f, oo in zip(frames, octaves)) * batch * cplx.itemsize)
                splan = ctx.stft_plan(n_fft, hop, np.ones(n_fft, dtype=real), True, pad_mode, real)          # window="ones" (:1197)
                ctx.stft_exec(splan, y_ptr, batch, lens[i], lens[i], d_ptr)
                ctx.cqt_project_exec(d_ptr, out_ptr, csr[0], csr[1], csr[2], scl, batch, frames[i], n_fft // 2 + 1, n_frames, n_bins, o["bin0"], o["row0"], o["n_rows"], real)
            if OVERLAP_OCTAVES:
                ctx.side(ctx.SIDE_BACK)
            if o["halve"]:
                y_ptr = shorten(y_ptr, lens[i], lens[i + 1], 2, 1.0)
        if OVERLAP_OCTAVES and not native:
            ctx.side(ctx.SIDE_JOIN)
        overlapped = False
        if check and ctx.nonfinite_read() and not _all_finite(y):
            raise ParameterError("Audio buffer is not finite everywhere")
        res = sess.result(handle)
    finally:
        if overlapped:   # an error between fork and join: nothing of this call may still run when its buffers are released
            ctx.side(ctx.SIDE_END)
        sess.close()
    return _arrays.swap_last_two(res.reshape(lead + (n_frames, n_bins)))


def cqt(y, *, sr=22050, hop_length=512, fmin=None, n_bins=84, bins_per_octave=12, tuning=0.0, filter_scale=1, norm=1, sparsity=0.01, window="hann", scale=True,
        pad_mode="constant", res_type="soxr_hq", dtype=None, check_finite=True):
    """Constant-Q transform; drop-in for ``librosa.cqt`` (``librosa/core/constantq.py:42-225``): the ``gamma=0`` case of :func:`vqt`.
    Not provided: ``tuning=None`` -- the reference estimates the tuning with its pitch tracker
    (``core/constantq.py:318-319, 985-986`` -> ``estimate_tuning``), which is outside this library's scope: pass a number (default 0.0).
    Unpinned: the reference's DEFAULT ``res_type="soxr_hq"`` runs this library's own band-limited design (see ``resample``); ``"polyphase"`` /
    ``"fft"`` / ``"scipy"`` are pinned against the reference.
    """
    return vqt(y, sr=sr, hop_length=hop_length, fmin=fmin, n_bins=n_bins, intervals="equal", gamma=0, bins_per_octave=bins_per_octave, tuning=tuning, filter_scale=filter_scale,
               norm=norm, sparsity=sparsity, window=window, scale=scale, pad_mode=pad_mode, res_type=res_type, dtype=dtype, check_finite=check_finite)
