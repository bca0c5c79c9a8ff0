"""Multi-GPU sharding of a batch of clips: one process per GPU, no collective on the data path.

Clips (leading axes) are independent (the reference asserts batch == per-item,
``tests/test_multichannel.py:96-111, 685-714``), so a batch shards by contiguous clip ranges with zero
exchange during compute.  The only collective is the optional final gather of the (small) mel
output -- RCCL over xGMI when the process group uses the ``nccl`` backend (which is RCCL on ROCm),
``gloo`` on CPU for tests.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): the gather of one 338.7 MB mel shard per rank
(BASELINE configs[2]: 512 clips x 128 x 1292 f32) is bound by the per-link rate, ~2.2 ms direct against
~1 ms of compute per shard, so ``ShardedGather`` overlaps it with the compute: the shard is produced and
shipped in chunks of clips, chunk c travelling (on the communication stream RCCL owns) while chunk c+1
is being computed.
"""
from __future__ import annotations


def shard_range(n_items: int, rank: int, world_size: int):
    """Contiguous, balanced [begin, end) of ``n_items`` for ``rank``; the first ``n % world`` ranks get one extra."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, extra = divmod(int(n_items), world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def stft_num_frames(n: int, n_fft: int, hop: int, center: bool) -> int:
    """Frames of a clip of ``n`` samples (``1 + n // hop`` centred, ``1 + (n - n_fft) // hop`` otherwise: reference ``tests/test_core.py:270-273``)."""
    return 1 + (n + (2 * (n_fft // 2) if center else 0) - n_fft) // hop


def shard_frames(n: int, rank: int, world_size: int, n_fft: int, hop: int, center: bool = True):
    """Frame-granularity shard of ONE long clip: rank ``rank`` of ``world_size`` computes frames
    ``[frame_lo, frame_hi)`` of the clip's STFT / mel spectrogram from the samples ``[sample_lo, sample_hi)`` -- its hop samples per frame plus the
    ``n_fft - hop`` halo that the last frame reaches into the neighbour's range.  ``sample_lo`` / ``sample_hi`` are clip coordinates and may lie
    outside ``[0, n)``: ``pad_left`` / ``pad_right`` samples of the reference's centre padding (``np.pad``, ``core/spectrum.py:273-328``) belong to the
    first / last shard and to nobody else.  The shard's frames are exactly the UNCENTRED frames of that (padded) slice:
    ``stft(slice, center=False)[..., k] == stft(y, center=True)[..., frame_lo + k]`` (the block property of ``core/spectrum.py:380-390`` and of
    ``core/audio.py:223-533``, ``stream``).  Returns a dict; ``frame_hi == frame_lo`` for a rank that gets nothing (more ranks than frames)."""
    n_frames = stft_num_frames(n, n_fft, hop, center)
    if n_frames < 1:
        raise ValueError(f"a clip of {n} samples has no frame of length {n_fft}")
    f0, f1 = shard_range(n_frames, rank, world_size)
    return frame_run(n, f0, f1, n_fft, hop, center)


def frame_run(n: int, f0: int, f1: int, n_fft: int, hop: int, center: bool = True):
    """The sample range (and its share of the centre padding) behind frames ``[f0, f1)`` of a clip of ``n`` samples: see ``shard_frames``."""
    n_frames = stft_num_frames(n, n_fft, hop, center)
    pad = n_fft // 2 if center else 0
    lo = f0 * hop - pad
    hi = (f1 - 1) * hop + n_fft - pad if f1 > f0 else lo
    return {"frame_lo": f0, "frame_hi": f1, "n_frames": n_frames, "sample_lo": lo, "sample_hi": hi, "pad_left": max(0, -lo), "pad_right": max(0, hi - n),
            "read_lo": min(max(lo, 0), n), "read_hi": min(max(hi, 0), n)}


def split_padded_edges(n: int, shard, n_fft: int, hop: int, center: bool = True):
    """A shard as runs of frames: the few frames that reach into the centre padding (``pad / hop`` at either end of the CLIP) on their own, so that
    attaching the padding copies a frame or two of samples and the bulk of an edge shard stays a view of the input (``frame_shard_input``)."""
    f0, f1 = shard["frame_lo"], shard["frame_hi"]
    pad = n_fft // 2 if center else 0
    if f1 <= f0 or pad == 0:
        return [shard]
    left_end = min(f1, max(f0, -(-pad // hop)))                       # frames f < ceil(pad / hop) start before sample 0
    right_start = max(left_end, min(f1, (n + pad - n_fft) // hop + 1))  # frames f > (n + pad - n_fft) / hop end beyond sample n
    runs = [(f0, left_end), (left_end, right_start), (right_start, f1)]
    return [frame_run(n, a, b, n_fft, hop, center) for a, b in runs if b > a]


def frame_shard_input(y, shard, pad_mode: str = "constant"):
    """The samples a frame shard transforms with ``center=False``: ``y[..., read_lo:read_hi]`` -- a VIEW for interior shards -- with the shard's own
    share of the centre padding attached (first / last shard only; ``np.pad`` with the reference's mode, whose sources lie inside the slice as long as
    the shard holds a frame or two: checked)."""
    import numpy as np

    piece = y[..., shard["read_lo"] : shard["read_hi"]]
    pl, pr = shard["pad_left"], shard["pad_right"]
    if pl == 0 and pr == 0:
        return piece
    need = max(pl, pr) + (1 if pad_mode == "reflect" else 0)
    if pad_mode != "constant" and piece.shape[-1] < need:
        raise ValueError(f"frame shard of {piece.shape[-1]} samples cannot supply {max(pl, pr)} samples of {pad_mode!r} padding")
    return np.pad(piece, [(0, 0)] * (piece.ndim - 1) + [(pl, pr)], mode=pad_mode)


def shard_sizes(n_items: int, world_size: int):
    return [shard_range(n_items, r, world_size)[1] - shard_range(n_items, r, world_size)[0] for r in range(world_size)]


def chunk_ranges(n: int, n_chunks: int):
    """[begin, end) of ``n_chunks`` near-equal consecutive pieces of range(n) (empty pieces dropped)."""
    n_chunks = max(1, min(int(n_chunks), max(1, n)))
    out = []
    for c in range(n_chunks):
        b, e = shard_range(n, c, n_chunks)
        if e > b:
            out.append((b, e))
    return out


class ShardedGather:
    """Gathers per-rank shards (clips on axis 0) into the full batch on every rank, piece by piece.

    ``full`` is allocated once; ``rank r``'s shard occupies rows ``shard_range(n_items, r, world)``, i.e. the layout of
    the unsharded result: no padding, no concatenation.  ``push(lo, hi, piece)`` ships rows ``[lo, hi)`` of the LOCAL
    shard as soon as they exist and returns immediately (``async_op``); ``wait()`` completes everything.

    * equal shards: one ``all_gather`` per piece straight into views of ``full`` (RCCL: every rank's piece crosses each
      xGMI link once);
    * unequal shards (``n_items % world != 0``): every rank broadcasts its own piece into its rows -- still no padding
      and no extra copy, at the price of ``world`` smaller collectives per piece.
    """

    def __init__(self, like, n_items: int, group=None):
        import torch
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.sizes = shard_sizes(n_items, self.world)
        self.offsets = [shard_range(n_items, r, self.world)[0] for r in range(self.world)]
        self.equal = len(set(self.sizes)) == 1
        self.full = torch.empty((int(n_items),) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
        self.pending = []

    def push(self, lo: int, hi: int, piece):
        """Ship rows [lo, hi) of this rank's shard (``piece`` holds exactly those rows)."""
        dist = self.dist
        if hi <= lo:
            return
        piece = piece.contiguous()
        if self.equal:
            views = [self.full[self.offsets[r] + lo : self.offsets[r] + hi] for r in range(self.world)]
            self.pending.append(dist.all_gather(views, piece, group=self.group, async_op=True))
            return
        # unequal shards: rows [lo, hi) exist on a rank only as far as its shard reaches
        for r in range(self.world):
            r_hi = min(hi, self.sizes[r])
            if r_hi <= lo:
                continue
            view = self.full[self.offsets[r] + lo : self.offsets[r] + r_hi]
            if r == self.rank:
                view.copy_(piece[: r_hi - lo])
            src = dist.get_global_rank(self.group, r) if self.group is not None else r
            self.pending.append(dist.broadcast(view, src=src, group=self.group, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        return self.full


def gather_shards(local, n_items: int, group=None, n_chunks: int = 1):
    """All-gather per-rank results (clips on axis 0, possibly unequal shard sizes) into the full batch.

    ``local`` is a torch tensor holding this rank's whole shard; every rank returns the full ``(n_items, ...)`` tensor.
    For a gather that overlaps the compute, produce the shard in pieces and use :class:`ShardedGather` directly
    (``bench.py``'s ``gathered`` measurement does)."""
    g = ShardedGather(local, n_items, group)
    mine = g.sizes[g.rank]
    biggest = max(g.sizes)
    for lo, hi in chunk_ranges(biggest, n_chunks):
        g.push(lo, hi, local[lo : min(hi, mine)])
    return g.wait()


class NativeGather:
    """The same gather through the C ABI's own RCCL binding (``lra_comm_*``, ``include/librosa_amd.h``) for host programs that
    do not run ``torch.distributed``: the caller distributes ``unique_id`` (from ``librosa_amd._native.comm_unique_id()`` on rank 0)
    by whatever channel it has; ``all_gather(local)`` enqueues ONE ncclAllGather on the context's stream -- stream-ordered after
    the kernels that produced ``local`` -- and returns the full ``(world * clips, ...)`` device tensor.  Unequal shards (``n_items`` given and not a
    multiple of the world size: rank r holds ``shard_range(n_items, r, world)``): one grouped ncclBroadcast per rank straight into its rows of the full
    tensor (``lra_comm_allgatherv``), no padding and no staging copy -- the layout of the unsharded result, as ``ShardedGather`` produces it."""

    def __init__(self, ctx, rank: int, world: int, unique_id: bytes):
        from . import _native

        self.ctx = ctx
        self.world = int(world)
        self.comm = _native.Comm(ctx, rank, world, unique_id)

    def all_gather(self, local, n_items=None):
        import torch

        local = local.contiguous()
        self.ctx.set_stream(torch.cuda.current_stream(local.device).cuda_stream)
        sizes = shard_sizes(n_items, self.world) if n_items is not None else [int(local.shape[0])] * self.world
        if local.shape[0] != sizes[self.comm.rank]:
            raise ValueError(f"rank {self.comm.rank} holds {local.shape[0]} items, shard_range({n_items}, {self.comm.rank}, {self.world}) says {sizes[self.comm.rank]}")
        full = torch.empty((sum(sizes),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        if len(set(sizes)) == 1:
            self.comm.allgather(local.data_ptr(), full.data_ptr(), local.numel() * local.element_size())
            return full
        item = local.element_size()
        for d in local.shape[1:]:
            item *= int(d)
        offs, acc = [], 0
        for sz in sizes:
            offs.append(acc * item)
            acc += sz
        self.comm.allgatherv(local.data_ptr() if local.numel() else 0, full.data_ptr(), [sz * item for sz in sizes], offs)
        return full

    def close(self):
        self.comm.close()
