"""Feature extraction on the hot path: ``melspectrogram`` and ``mfcc`` (``librosa/feature/__init__.pyi:12-13``)."""
from .spectral import melspectrogram, mfcc

__all__ = ["melspectrogram", "mfcc"]
