"""Block feeder for out-of-core audio: ``librosa.stream`` (``librosa/core/audio.py:223-533``).

SURVEY.md 8(f) rank 4: the step in front of the path.  ``stream`` yields fixed-size, overlapping blocks of PCM such that the
frames of ``stft(block, center=False)`` tile the frames of the whole signal (block ``i`` holds frames
``i * block_length ... (i + 1) * block_length - 1``); fed to ``librosa_amd.stft(block, n_fft=frame_length,
hop_length=hop_length, center=False, out=D)`` every block goes through the same pinned staging and device buffers
(``include/librosa_amd.h``, ``lra_stft_exec_host``), see ``docs/examples/plot_pcen_stream.py:72-74`` in the reference.

Decoding is not rebuilt here (audio I/O is out of scope): sources are an in-memory ``np.ndarray``, a
``soundfile.SoundFile`` / path when the ``soundfile`` package is importable (what the reference uses), or a PCM ``.wav``
path through the standard library.  Resampling (``sr`` different from the file's rate) is not provided.
"""
from __future__ import annotations

import os
import wave

import numpy as np

from ..util.exceptions import ParameterError
from ..util.utils import is_positive_int

__all__ = ["stream"]


def _chunks_from_array(y, chunk, start, frames):
    """(n,) or (channels, n) array -> chunks shaped like soundfile's: (k,) or (k, channels)."""
    y = np.asarray(y)
    if y.ndim not in (1, 2):
        raise ParameterError(f"in-memory audio must be 1-d or (channels, samples), given shape={y.shape}")
    n = y.shape[-1]
    start = min(max(start, 0), n)
    stop = n if frames < 0 else min(n, start + frames)
    for lo in range(start, stop, chunk):
        piece = y[..., lo : min(lo + chunk, stop)]
        yield piece if y.ndim == 1 else piece.T


def _wav_reader(path):
    w = wave.open(path, "rb")
    width, channels = w.getsampwidth(), w.getnchannels()
    if width not in (1, 2, 3, 4):
        w.close()
        raise ParameterError(f"unsupported PCM sample width {width} in {path!r}")
    return w, width, channels


def _decode_pcm(raw, width, channels, dtype):
    """Integer PCM -> floats in [-1, 1) the way libsndfile scales them (divide by 2**(bits-1))."""
    if width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
    elif width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = np.where(v >= 1 << 23, v - (1 << 24), v).astype(np.float64) / 8388608.0
    else:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0
    x = x.astype(dtype)
    return x if channels == 1 else x.reshape(-1, channels)


def _chunks_from_wav(w, width, channels, chunk, start, frames, dtype):
    total = w.getnframes()
    start = min(max(start, 0), total)
    w.setpos(start)
    left = total - start if frames < 0 else min(frames, total - start)
    while left > 0:
        k = min(chunk, left)
        raw = w.readframes(k)
        if not raw:
            break
        yield _decode_pcm(raw, width, channels, dtype)
        left -= k


def stream(path, *, block_length, frame_length, hop_length, sr=None, mono=True, offset=0.0, duration=None, fill_value=None, res_type="soxr_hq", dtype=np.float32):
    """Stream audio in fixed-length buffers; drop-in for ``librosa.stream`` (``librosa/core/audio.py:223-533``).

    Each block has ``(block_length - 1) * hop_length + frame_length`` samples and successive blocks start
    ``block_length * hop_length`` samples apart (``:409-410``), so block-wise ``stft(..., center=False)`` reproduces the
    frames of the whole signal.  The last block is short unless ``fill_value`` is given (``:503-516``).  Multi-channel
    sources yield ``(channels, samples)`` blocks, or their mean when ``mono`` (``:455-461``).

    ``path``: ``np.ndarray`` (``(n,)`` or ``(channels, n)``; ``sr`` is then only used to convert ``offset`` / ``duration``
    from seconds and defaults to 22050), a ``.wav`` path, or anything ``soundfile.SoundFile`` accepts when that package is
    installed.  ``res_type`` is accepted for signature compatibility; resampling is not provided.
    """
    if not is_positive_int(block_length):
        raise ParameterError(f"block_length={block_length} must be a positive integer")
    if not is_positive_int(frame_length):
        raise ParameterError(f"frame_length={frame_length} must be a positive integer")
    if not is_positive_int(hop_length):
        raise ParameterError(f"hop_length={hop_length} must be a positive integer")
    dtype = np.dtype(dtype)
    target_yield_size = (block_length - 1) * hop_length + frame_length      # :409
    target_advance = block_length * hop_length                              # :410

    closer = None
    if isinstance(path, np.ndarray):
        orig_sr = float(sr) if sr is not None else 22050.0
        channels = 1 if path.ndim == 1 else path.shape[0]
        make = lambda start, frames: _chunks_from_array(path.astype(dtype, copy=False), target_advance, start, frames)  # noqa: E731
        total = path.shape[-1]
    else:
        sf = None
        try:
            import soundfile as sf  # noqa: F811
        except ImportError:
            pass
        if sf is not None:
            sfo = path if isinstance(path, sf.SoundFile) else sf.SoundFile(path)
            if not isinstance(path, sf.SoundFile):
                closer = sfo.close
            orig_sr, channels, total = float(sfo.samplerate), sfo.channels, sfo.frames

            def make(start, frames, _sfo=sfo):
                _sfo.seek(start)
                return _sfo.blocks(blocksize=target_advance, overlap=0, dtype=dtype.name, always_2d=False, frames=frames)
        else:
            if not isinstance(path, (str, os.PathLike)):
                raise ParameterError("without the soundfile package, stream() reads np.ndarray sources and PCM .wav paths only")
            w, width, channels = _wav_reader(os.fspath(path))
            closer = w.close
            orig_sr, total = float(w.getframerate()), w.getnframes()
            make = lambda start, frames: _chunks_from_wav(w, width, channels, target_advance, start, frames, dtype)  # noqa: E731
    try:
        if sr is not None and not isinstance(path, np.ndarray) and float(sr) != orig_sr:
            raise ParameterError(f"sr={sr} differs from the file's {orig_sr:g} Hz: resampling is outside librosa_amd's scope")
        start = int(offset * orig_sr) if offset >= 0 else max(0, total - int(abs(offset) * orig_sr))   # :440-443
        read_frames = int(duration * orig_sr) if duration is not None else -1                           # :420
        process_channels = 1 if (mono or channels == 1) else channels
        capacity = target_yield_size + 2 * target_advance                                               # :431
        buffer = np.zeros((capacity,) if process_channels == 1 else (capacity, process_channels), dtype=dtype)
        write_idx = read_idx = 0
        for chunk in make(start, read_frames):
            chunk = np.asarray(chunk, dtype=dtype)
            if mono and chunk.ndim == 2:
                chunk = chunk.mean(axis=1, dtype=dtype)          # to_mono over the channel axis (:455-461)
            k = chunk.shape[0]
            if write_idx + k > capacity:                         # compact: move the unread tail to the front (:469-473)
                available = write_idx - read_idx
                buffer[:available] = buffer[read_idx:write_idx]
                read_idx, write_idx = 0, available
            buffer[write_idx : write_idx + k] = chunk
            write_idx += k
            while write_idx - read_idx >= target_yield_size:     # :488-491
                yield buffer[read_idx : read_idx + target_yield_size].T.copy()
                read_idx += target_advance
        remainder = buffer[read_idx:write_idx]                   # :507-520: what is left, short blocks padded on request
        rem_idx = 0
        while rem_idx < remainder.shape[0]:
            cur = remainder[rem_idx : rem_idx + target_yield_size]
            if cur.shape[0] < target_yield_size and fill_value is not None:
                pad = target_yield_size - cur.shape[0]
                cur = np.pad(cur, (0, pad) if process_channels == 1 else ((0, pad), (0, 0)), mode="constant", constant_values=fill_value)
            yield cur.T.copy()
            rem_idx += target_advance
    finally:
        if closer is not None:
            closer()
