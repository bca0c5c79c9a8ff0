"""Marshalling between host/device array types and raw device pointers.

Two kinds of caller are served:
  * NumPy arrays (the drop-in path): data is uploaded to HBM, the kernels run, results come back
    as ``np.ndarray`` -- signatures and return types identical to the reference's;
  * ``torch`` tensors already resident on a ROCm device (the fast path used by ``bench.py`` and by
    pipelines that keep audio on the GPU): pointers are passed through, work is enqueued on
    torch's current stream and device tensors are returned.  PyTorch is only plumbing here
    (allocator + streams); all arithmetic is in the HIP library.
"""
from __future__ import annotations

import os

import numpy as np

from . import _native
from .util.exceptions import ParameterError
from .util.utils import is_torch_tensor

_NP2TORCH = {}

# Test hook (tests/test_gpu_parity.py::test_istft_stores_every_sample): results are allocated full of NaN bit patterns, so a kernel
# that leaves any element of its output unwritten shows up as NaN instead of as whatever the allocator handed out.
POISON_OUTPUTS = bool(os.environ.get("LRA_POISON_OUTPUTS"))


def _torch():
    import torch

    if not _NP2TORCH:
        _NP2TORCH.update({np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
                          np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128,
                          np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64, np.dtype(np.uint8): torch.uint8})   # integers: index tables (CSR bases) and integer-valued inputs
    return torch


def torch_dtype(np_dtype):
    _torch()
    return _NP2TORCH[np.dtype(np_dtype)]


def numpy_dtype_of(x):
    """NumPy dtype of a numpy array or torch tensor."""
    if is_torch_tensor(x):
        torch = _torch()
        for k, v in _NP2TORCH.items():
            if v == x.dtype:
                return k
        if x.dtype in (torch.float16, torch.bfloat16):
            return np.dtype(np.float16)
        raise ParameterError(f"unsupported tensor dtype {x.dtype}")
    return x.dtype


class Session:
    """Per-call device session: resolves the context/stream and tracks temporary buffers."""

    def __init__(self, like):
        self.is_torch = is_torch_tensor(like)
        self._keep = []
        self._locked = False
        if self.is_torch:
            torch = _torch()
            if like.device.type != "cuda":
                raise ParameterError("torch inputs must live on a ROCm device (tensor.device.type == 'cuda'); pass a numpy array for host data")
            self.device = like.device
            self.ctx = _native.get_context(like.device.index if like.device.index is not None else torch.cuda.current_device())
        else:
            self.device = None
            self.ctx = _native.get_context()
        # One call at a time per context: the stream selection below, the sticky non-finite flag, the plans' scratch
        # buffers and rocFFT work areas are per-context state, and ctypes releases the GIL during native calls.  The
        # lock is held until close(); the device work of different threads is serialised on one stream anyway.
        self.ctx.call_lock.acquire()
        self._locked = True
        try:
            if self.is_torch:
                self.ctx.set_stream(_torch().cuda.current_stream(like.device).cuda_stream)
            else:
                self.ctx.use_own_stream()
        except Exception:
            self.close()
            raise

    # ---- inputs ---------------------------------------------------------------------------------
    def input_2d(self, x, dtype):
        """(…, n) array/tensor -> (ptr, batch, n, row_stride) of a C-contiguous (batch, n) device array."""
        dtype = np.dtype(dtype)
        if self.is_torch:
            t = x.to(torch_dtype(dtype)).reshape(-1, x.shape[-1]).contiguous()
            self._keep.append(t)
            return t.data_ptr(), t.shape[0], t.shape[1], t.shape[1]
        a = np.ascontiguousarray(x, dtype=dtype).reshape(-1, x.shape[-1])
        buf = self.ctx.alloc(max(a.nbytes, 16)).upload(a)
        self._keep.append(buf)
        return buf.ptr, a.shape[0], a.shape[1], a.shape[1]

    def input_raw(self, a, dtype):
        """Upload/borrow a contiguous array as-is; returns ptr."""
        if self.is_torch:
            t = a.to(torch_dtype(dtype)).contiguous()
            self._keep.append(t)
            return t.data_ptr()
        a = np.ascontiguousarray(a, dtype=dtype)
        buf = self.ctx.alloc(max(a.nbytes, 16)).upload(a)
        self._keep.append(buf)
        return buf.ptr

    # ---- outputs --------------------------------------------------------------------------------
    def output(self, shape, dtype, rows=False):
        """Allocate a device result; returns (ptr, handle) where handle is finalised by ``result``.

        ``rows=True``: a large frame-major spectrum ``(batch, n_frames, row)``; ``rows="flat"``: any other large result the store-bound kernels write (the inverse
        transform's signals: its rate follows where its OUTPUT lands, 0.67-0.71 -> 0.63-0.65 ms for 256 x 30 s, ``profiles/r06_raw/o_istft_placement.txt``), judged as
        8 KiB rows.  With the context's ``placement_retry`` option on, such a result of at
        least 256 MB comes from ``lra_malloc_placed`` -- the best of a few candidate allocations under the kernels' own write stream, because where this
        buffer lands moves the store-bound transforms by up to 15 % on some boxes (``profiles/r05_pitch.md``) -- wrapped as a tensor that hands the
        buffer back to the context when it dies, so a loop of calls allocates once."""
        dtype = np.dtype(dtype)
        if self.is_torch:
            nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
            t = None
            if rows and self.ctx.placement_retry > 0 and nbytes >= self.ctx.PLACED_MIN_BYTES and (rows == "flat" or (len(shape) == 3 and int(shape[0]) * int(shape[1]) >= 4096)):
                try:
                    t = _placed_tensor(self.ctx, tuple(int(s) for s in shape), dtype, self.device, flat=rows == "flat")
                except (_native.NativeError, ParameterError):  # (no room for a candidate next to torch's cached blocks, or no virtual-memory API: an ordinary allocation serves)
                    t = None
            if t is None:
                t = _torch().empty(tuple(int(s) for s in shape), dtype=torch_dtype(dtype), device=self.device)
            if POISON_OUTPUTS and t.numel():
                _torch().as_strided(t, (t.numel(),), (1,)).view(_torch().uint8).fill_(0xFF)
            return t.data_ptr(), t
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        buf = self.ctx.alloc(max(nbytes, 16))
        if POISON_OUTPUTS and nbytes:
            self.ctx.memset(buf.ptr, 0xFF, nbytes)
        return buf.ptr, (buf, tuple(int(s) for s in shape), dtype)

    def result(self, handle):
        """Device result -> tensor (torch) or downloaded ndarray (numpy)."""
        if self.is_torch:
            return handle
        buf, shape, dtype = handle
        out = buf.download(shape, dtype)
        buf.free()
        return out

    def scratch(self, nbytes):
        if self.is_torch:
            # (scratch and the results of the other rows -- hpss, griffinlim, the vocoder -- stay with torch's allocator: measured, placement moves them by 1 % (decompose.hpss
            #  1.179 -> 1.170 ms for 48 clips, scripts/placed_probe3.py), not worth a second allocator's churn under variable shapes)
            t = _torch().empty(int(max(nbytes, 16)), dtype=_torch().uint8, device=self.device)
            self._keep.append(t)
            return t.data_ptr()
        buf = self.ctx.alloc(max(int(nbytes), 16))
        self._keep.append(buf)
        return buf.ptr

    def close(self):
        try:
            for k in self._keep:
                if isinstance(k, _native.DeviceBuffer):
                    k.free()
            self._keep = []
        finally:
            if self._locked:
                self._locked = False
                self.ctx.call_lock.release()


class _PlacedHolder:
    """Owner of one lra_malloc_placed buffer behind a torch tensor (``__cuda_array_interface__``): torch keeps a reference to this object for as long
    as any view of the tensor lives; after that the buffer goes back to its context (recycled for the next result of the same shape)."""

    def __init__(self, ctx, nbytes, row_bytes, rows_per_item=0):
        self.ctx, self.nbytes, self.row_bytes, self.rows_per_item = ctx, int(nbytes), int(row_bytes), int(rows_per_item)
        self.ptr = ctx.placed_take(self.nbytes, self.row_bytes, self.rows_per_item)
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (int(self.ptr), False), "version": 3, "strides": None}

    def __del__(self):
        try:
            self.ctx.placed_give(self.nbytes, self.row_bytes, self.ptr, self.rows_per_item)
        except Exception:  # pragma: no cover - interpreter shutdown
            pass


def _placed_tensor(ctx, shape, dtype, device, flat=False):
    """A tensor of ``shape`` over a buffer from ``lra_malloc_placed``; ``flat``: no row structure of its own -- the buffer is sized and judged as rows of 8 KiB."""
    torch = _torch()
    nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    if flat or (shape[-1] * np.dtype(dtype).itemsize) % 8 or shape[-1] * np.dtype(dtype).itemsize < 512:   # (|X|^p rows of 1025 floats: no 8-byte row structure)
        row = 8192
        holder = _PlacedHolder(ctx, (nbytes + row - 1) // row * row, row)
        return torch.as_tensor(holder, device=device)[:nbytes].view(torch_dtype(dtype)).reshape(shape)
    holder = _PlacedHolder(ctx, nbytes, shape[-1] * np.dtype(dtype).itemsize, shape[-2] if len(shape) >= 2 else 0)  # (rows per clip: the probe cuts strips as the kernels do)
    return torch.as_tensor(holder, device=device).view(torch_dtype(dtype)).reshape(shape)


def swap_last_two(x):
    return x.transpose(-1, -2) if is_torch_tensor(x) else np.swapaxes(x, -1, -2)


def cast(x, dtype):
    dtype = np.dtype(dtype)
    if is_torch_tensor(x):
        td = torch_dtype(dtype)
        return x if x.dtype == td else x.to(td)
    return x if x.dtype == dtype else x.astype(dtype)
