"""Feature extraction on the hot path: ``melspectrogram`` (``librosa/feature/__init__.pyi:12``)."""
from .spectral import melspectrogram

__all__ = ["melspectrogram"]
