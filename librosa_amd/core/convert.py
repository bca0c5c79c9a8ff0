"""Frequency-scale conversions feeding ``filters.mel`` (host-side float64 scalar math).

Same formulas, constants and argument meaning as ``librosa/core/convert.py`` (``hz_to_mel``
:1004-1058, ``mel_to_hz`` :1069-1121, ``fft_frequencies`` :1369-1391, ``mel_frequencies``
:1432-1511) so the resulting filterbank is bit-identical to the reference's.
"""
from __future__ import annotations

import numpy as np

# Slaney (Auditory Toolbox) mel scale: linear below 1 kHz, logarithmic above
_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(frequencies, *, htk=False):
    f = np.asanyarray(frequencies)[()]
    if htk:
        return 2595.0 * np.log10(1.0 + f / 700.0)
    mels = f / _F_SP
    if f.ndim:
        hi = f >= _MIN_LOG_HZ
        mels[hi] = _MIN_LOG_MEL + np.log(f[hi] / _MIN_LOG_HZ) / _LOGSTEP
    elif f >= _MIN_LOG_HZ:
        mels = _MIN_LOG_MEL + np.log(f / _MIN_LOG_HZ) / _LOGSTEP
    return mels


def mel_to_hz(mels, *, htk=False):
    m = np.asanyarray(mels)[()]
    if htk:
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    freqs = _F_SP * m
    if m.ndim:
        hi = m >= _MIN_LOG_MEL
        freqs[hi] = _MIN_LOG_HZ * np.exp(_LOGSTEP * (m[hi] - _MIN_LOG_MEL))
    elif m >= _MIN_LOG_MEL:
        freqs = _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL))
    return freqs


def fft_frequencies(*, sr=22050, n_fft=2048):
    return np.fft.rfftfreq(n=n_fft, d=1.0 / sr)


def mel_frequencies(n_mels=128, *, fmin=0.0, fmax=11025.0, htk=False):
    lo = hz_to_mel(fmin, htk=htk)
    hi = hz_to_mel(fmax, htk=htk)
    return mel_to_hz(np.linspace(lo, hi, n_mels), htk=htk)
