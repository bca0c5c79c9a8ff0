"""Core spectral transforms on the hot path (``librosa/core/spectrum.py``, ``core/convert.py``)."""
from . import audio, convert, intervals, spectrum
from .audio import resample, stream
from .convert import fft_frequencies, hz_to_mel, mel_frequencies, mel_to_hz
from .intervals import interval_frequencies
from .spectrum import _spectrogram, amplitude_to_db, db_to_amplitude, db_to_power, griffinlim, istft, magphase, pcen, phase_vocoder, power_to_db, stft
from . import constantq
from .constantq import cqt, vqt

__all__ = ["audio", "constantq", "convert", "intervals", "spectrum", "cqt", "vqt", "interval_frequencies", "stream", "resample", "stft", "istft", "_spectrogram", "magphase", "griffinlim", "phase_vocoder", "pcen", "power_to_db", "amplitude_to_db", "db_to_power", "db_to_amplitude", "hz_to_mel", "mel_to_hz", "fft_frequencies", "mel_frequencies"]
