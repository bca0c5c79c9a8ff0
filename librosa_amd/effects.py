"""Time-domain effects built on the path: ``librosa/effects.py``.

``time_stretch`` is the reference's stft -> phase vocoder -> istft chain (``librosa/effects.py:464-484``) and ``hpss`` /
``harmonic`` / ``percussive`` its stft -> ``decompose.hpss`` -> istft chain (``:70-301``), ``pitch_shift`` = ``time_stretch`` + ``resample``
(``:487-596``), with every stage on the device: for a
device tensor nothing crosses PCIe, for an ``np.ndarray`` only the signal goes up and the result comes down.
"""
from __future__ import annotations

import numpy as np

from . import _arrays
from . import decompose
from .core import audio as _audio
from .core import spectrum
from .util.exceptions import ParameterError
from .util.utils import fix_length, is_positive_int, is_torch_tensor

__all__ = ["time_stretch", "pitch_shift", "hpss", "harmonic", "percussive"]


def time_stretch(y, *, rate, **kwargs):
    """Time-stretch an audio series by a fixed rate; drop-in for ``librosa.effects.time_stretch`` (``librosa/effects.py:404-484``).

    ``kwargs`` go to ``stft`` / ``istft`` exactly as in the reference, which also passes ``hop_length`` / ``n_fft`` on to
    ``phase_vocoder`` and thereby triggers its deprecation warnings (``:471-476``); so does this function.
    """
    if rate <= 0:
        raise ParameterError("rate must be a positive number")
    on_device = is_torch_tensor(y)
    yd = y
    if not on_device:
        spectrum._validate_audio(y, True)
        try:  # one upload, one download: the spectra stay on the device in between (torch is the device allocator here)
            torch = _arrays._torch()
        except ImportError:  # without torch every stage moves its own arrays
            torch = None
        if torch is not None:
            if not np.isfinite(y).all():
                raise ParameterError("Audio buffer is not finite everywhere")
            yd = torch.from_numpy(np.ascontiguousarray(y)).to(f"cuda:{_arrays._native.get_context().device}")
    staged = is_torch_tensor(yd) and not on_device
    D = spectrum.stft(yd, check_finite=not staged, row_align=0, **kwargs)  # (packed rows: the vocoder walks the buffer as it is)
    Ds = spectrum.phase_vocoder(D, rate=rate, hop_length=kwargs.get("hop_length"), n_fft=kwargs.get("n_fft"))
    len_stretch = round(y.shape[-1] / rate)
    ikw = dict(kwargs)
    ikw.pop("pad_mode", None)
    out = spectrum.istft(Ds, dtype=_arrays.numpy_dtype_of(y), length=len_stretch, **ikw)
    return out.cpu().numpy() if staged else out


def pitch_shift(y, *, sr, n_steps, bins_per_octave=12, res_type="soxr_hq", scale=False, **kwargs):
    """Shift the pitch of an audio series by ``n_steps`` steps; drop-in for ``librosa.effects.pitch_shift`` (``librosa/effects.py:487-596``):
    ``time_stretch`` by ``2 ** (-n_steps / bins_per_octave)``, ``resample`` back from ``sr / rate`` to ``sr``, crop / pad to the input's
    length -- one upload and one download for a NumPy ``y``.  The intermediate rate is not an integer, so of the scipy-backed converters
    only ``res_type="fft"`` / ``"scipy"`` apply (``"polyphase"`` raises, as in the reference); the band-limited names (``soxr_*``, the
    default, ``kaiser_*``, ``sinc_*``) take the Fourier converter too, which is band-limited by construction (parity unpinned: those
    packages are not in the build image)."""
    if not is_positive_int(bins_per_octave):
        raise ParameterError(f"bins_per_octave={bins_per_octave} must be a positive integer.")
    rate = 2.0 ** (-float(n_steps) / bins_per_octave)
    yd, staged = _stage(y)
    stretched = time_stretch(yd, rate=rate, **kwargs)
    orig_sr = float(sr) / rate
    if res_type != "polyphase" and not (int(orig_sr) == orig_sr and int(sr) == sr) and (res_type.startswith("soxr") or res_type in _audio._FIR_LIKE):
        res_type = "fft"
    shifted = _audio.resample(stretched, orig_sr=orig_sr, target_sr=sr, res_type=res_type, scale=scale)
    n = int(y.shape[-1])
    if is_torch_tensor(shifted):
        m = int(shifted.shape[-1])
        shifted = shifted[..., :n] if m >= n else _arrays._torch().nn.functional.pad(shifted, (0, n - m))
        return shifted.cpu().numpy() if staged else shifted
    return fix_length(shifted, size=n)


def _stage(y):
    """(array the transforms run on, whether it was staged from host memory): an ``np.ndarray`` is validated and uploaded once, so
    that the spectra between the transforms never leave the device (torch is the device allocator here)."""
    if is_torch_tensor(y):
        return y, False
    spectrum._validate_audio(y, True)
    try:
        torch = _arrays._torch()
    except ImportError:  # without torch every stage moves its own arrays
        return y, False
    if not np.isfinite(y).all():
        raise ParameterError("Audio buffer is not finite everywhere")
    return torch.from_numpy(np.ascontiguousarray(y)).to(f"cuda:{_arrays._native.get_context().device}"), True


def _hpss_parts(y, which, kernel_size, power, mask, margin, n_fft, hop_length, win_length, window, center, pad_mode):
    yd, staged = _stage(y)
    # The reference accepts `window` and then passes it to NEITHER transform (effects.py:161-183): analysis, synthesis and the
    # window-sum-square normalisation all run with the default window.  Same here (forwarding it to the forward transform only
    # would analyse with one window and synthesise with another, which matches neither the reference nor a consistent pair).
    del window
    D = spectrum.stft(yd, n_fft=n_fft, hop_length=hop_length, win_length=win_length, center=center, pad_mode=pad_mode, check_finite=not staged,
                      row_align=0)  # (packed rows: the separation kernel walks the buffer as it is)
    parts = decompose.hpss(D, kernel_size=kernel_size, power=power, mask=mask, margin=margin)
    ikw = dict(dtype=_arrays.numpy_dtype_of(y), n_fft=n_fft, hop_length=hop_length, win_length=win_length, center=center, length=y.shape[-1])
    if mask:  # the reference then inverts the (real-valued) masks themselves; irfft takes them as spectra with zero phase
        cplx = np.dtype(np.complex128) if _arrays.numpy_dtype_of(D) == np.complex128 else np.dtype(np.complex64)
        parts = [_arrays.cast(p, cplx) for p in parts]
    out = [spectrum.istft(parts[i], **ikw) for i in which]
    return [o.cpu().numpy() if staged else o for o in out]


def hpss(y, *, kernel_size=31, power=2.0, mask=False, margin=1.0, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True, pad_mode="constant"):
    """Decompose an audio series into harmonic and percussive components; drop-in for ``librosa.effects.hpss``
    (``librosa/effects.py:70-185``): ``stft`` -> ``decompose.hpss`` -> two ``istft``, device-resident in between."""
    h, p = _hpss_parts(y, (0, 1), kernel_size, power, mask, margin, n_fft, hop_length, win_length, window, center, pad_mode)
    return h, p


def harmonic(y, *, kernel_size=31, power=2.0, mask=False, margin=1.0, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True, pad_mode="constant"):
    """Harmonic component of an audio series; drop-in for ``librosa.effects.harmonic`` (``librosa/effects.py:188-243``)."""
    return _hpss_parts(y, (0,), kernel_size, power, mask, margin, n_fft, hop_length, win_length, window, center, pad_mode)[0]


def percussive(y, *, kernel_size=31, power=2.0, mask=False, margin=1.0, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True, pad_mode="constant"):
    """Percussive component of an audio series; drop-in for ``librosa.effects.percussive`` (``librosa/effects.py:246-301``)."""
    return _hpss_parts(y, (1,), kernel_size, power, mask, margin, n_fft, hop_length, win_length, window, center, pad_mode)[0]
