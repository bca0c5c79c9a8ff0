"""Utilities on the STFT/mel/ISTFT path (subset of ``librosa.util`` the path depends on)."""
from . import exceptions
from .exceptions import LibrosaError, ParameterError
from .utils import (MAX_MEM_BLOCK, dtype_c2r, dtype_r2c, fix_length, is_positive_int, normalize, pad_center, sparsify_rows, tiny, valid_audio)

__all__ = ["exceptions", "LibrosaError", "ParameterError", "MAX_MEM_BLOCK", "dtype_c2r", "dtype_r2c", "fix_length", "is_positive_int", "normalize", "sparsify_rows",
           "pad_center", "tiny", "valid_audio"]
