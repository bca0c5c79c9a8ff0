"""Frequency grids for the constant-Q family: ``librosa.interval_frequencies`` (``librosa/core/intervals.py:28-135``) for
equal temperament and explicit interval lists.  The named just-intonation sets (``"pythagorean"``, ``"ji3"`` ...) come from the
reference's notation tables, which are outside this path."""
from __future__ import annotations

import numpy as np

from ..util.exceptions import ParameterError

__all__ = ["interval_frequencies"]


def interval_frequencies(n_bins, *, fmin, intervals="equal", bins_per_octave=12, tuning=0.0, sort=True):
    """``n_bins`` frequencies from ``fmin`` upwards: one octave of ratios in [1, 2), repeated at every power of two."""
    if isinstance(intervals, str):
        if intervals != "equal":
            raise ParameterError(f"intervals={intervals!r}: librosa_amd provides 'equal' or an explicit list of intervals")
        octave = 2.0 ** ((tuning + np.arange(0, bins_per_octave, dtype=float)) / bins_per_octave)
    else:
        octave = np.array(intervals)
        bins_per_octave = len(octave)
    repeats = np.ceil(n_bins / bins_per_octave)
    grid = np.multiply.outer(2.0 ** np.arange(repeats), octave).flatten()[:n_bins]
    if sort:
        grid = np.sort(grid)
    return grid * fmin
