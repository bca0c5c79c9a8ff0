"""Time-domain effects built on the path: ``librosa/effects.py``.

``time_stretch`` is the reference's stft -> phase vocoder -> istft chain (``librosa/effects.py:464-484``) with all three
stages on the device; for a device tensor nothing crosses PCIe, for an ``np.ndarray`` only the signal goes up and the
stretched signal comes down.  ``hpss`` (``:161-185``) is the same round trip around ``decompose.hpss`` (median filtering,
outside this repository's scope): compute the masks with any array library on ``librosa_amd.stft``'s device result and
hand the masked spectra to ``librosa_amd.istft`` -- both ends stay on the device.
"""
from __future__ import annotations

import numpy as np

from . import _arrays
from .core import spectrum
from .util.exceptions import ParameterError
from .util.utils import is_torch_tensor

__all__ = ["time_stretch"]


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
    D = spectrum.stft(yd, check_finite=not staged, **kwargs)
    Ds = spectrum.phase_vocoder(D, rate=rate, hop_length=kwargs.get("hop_length"), n_fft=kwargs.get("n_fft"))
    len_stretch = round(y.shape[-1] / rate)
    ikw = dict(kwargs)
    ikw.pop("pad_mode", None)
    out = spectrum.istft(Ds, dtype=_arrays.numpy_dtype_of(y), length=len_stretch, **ikw)
    return out.cpu().numpy() if staged else out
