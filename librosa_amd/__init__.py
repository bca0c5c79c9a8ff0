"""librosa_amd -- MI355X-native (gfx950) implementation of librosa's STFT -> mel-spectrogram hot
path and its inverse, behind librosa's own Python signatures.

    import librosa_amd as librosa
    D = librosa.stft(y, n_fft=2048, hop_length=512)
    M = librosa.feature.melspectrogram(y=y, sr=22050, n_fft=2048, hop_length=512, n_mels=128)
    y2 = librosa.istft(D, hop_length=512, length=len(y))

Only this path and the callers right next to it are provided -- ``magphase``, decibel scaling, ``feature.mfcc``, ``griffinlim``,
``phase_vocoder`` / ``effects.time_stretch``, ``decompose.hpss`` / ``effects.hpss``, ``pcen``, ``cqt`` / ``vqt``, ``stream``, ``resample`` (see
DESIGN.md for the scope table).  Host-side Python validates
arguments exactly like the reference and builds the small float64 tables (window, mel basis, window
sum-square); all signal arithmetic runs in hand-written HIP kernels through the C ABI declared in
``include/librosa_amd.h``.  There is no CPU fallback: without the built library and a GPU, compute
calls raise ``librosa_amd.NativeError``.
"""
from . import core, decompose, effects, feature, filters, util
from ._native import NativeError, device_count, get_context
from .core import (_spectrogram, amplitude_to_db, cqt, interval_frequencies, vqt, db_to_amplitude, db_to_power, fft_frequencies, griffinlim, hz_to_mel, istft, magphase, pcen, phase_vocoder, mel_frequencies, mel_to_hz,
                   power_to_db, resample, stft, stream)
from .util.exceptions import LibrosaError, ParameterError

__version__ = "0.1.0"

__all__ = ["core", "decompose", "effects", "feature", "filters", "util", "stft", "istft", "_spectrogram", "magphase", "griffinlim", "phase_vocoder", "pcen", "cqt", "vqt", "interval_frequencies", "stream", "resample", "power_to_db", "amplitude_to_db", "db_to_power", "db_to_amplitude", "hz_to_mel", "mel_to_hz", "fft_frequencies", "mel_frequencies",
           "LibrosaError", "ParameterError", "NativeError", "device_count", "get_context"]
