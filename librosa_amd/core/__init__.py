"""Core spectral transforms on the hot path (``librosa/core/spectrum.py``, ``core/convert.py``)."""
from . import audio, convert, spectrum
from .audio import stream
from .convert import fft_frequencies, hz_to_mel, mel_frequencies, mel_to_hz
from .spectrum import _spectrogram, amplitude_to_db, db_to_amplitude, db_to_power, griffinlim, istft, pcen, phase_vocoder, power_to_db, stft

__all__ = ["audio", "convert", "spectrum", "stream", "stft", "istft", "_spectrogram", "griffinlim", "phase_vocoder", "pcen", "power_to_db", "amplitude_to_db", "db_to_power", "db_to_amplitude", "hz_to_mel", "mel_to_hz", "fft_frequencies", "mel_frequencies"]
