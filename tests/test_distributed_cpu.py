"""The N>1 path on CPU: two gloo processes shard a batch of clips, each computes its shard (the CPU
oracle stands in for the HIP kernels, which need a GPU), and the gathered result equals the unsharded
batch.  This exercises exactly the host logic bench.py / a multi-GPU caller use: shard_range, the
shard-invariant input generator and gather_shards (RCCL under the nccl backend on the GPU box)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist

    import stft_oracle as O
    from librosa_amd.distributed import gather_shards, shard_range

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    b, e = shard_range(n_clips, rank, world)
    y = O.config_input(e - b, n=8000, first_clip=b)  # clip i depends only on (seed, i)
    M = O.melspectrogram(y=y, sr=22050, n_fft=512, hop_length=128, n_mels=20)
    full = gather_shards(torch.from_numpy(M), n_clips)
    chunked = gather_shards(torch.from_numpy(M), n_clips, n_chunks=2)  # the overlapped form: the shard travels in pieces
    # pieces pushed as they are "computed" (what bench.py's gathered measurement does)
    from librosa_amd.distributed import ShardedGather, chunk_ranges

    g = ShardedGather(torch.from_numpy(M), n_clips)
    for lo, hi in chunk_ranges(max(g.sizes), 3):
        g.push(lo, hi, torch.from_numpy(M[lo : min(hi, e - b)]))
    piecewise = g.wait()
    assert torch.equal(full, chunked) and torch.equal(full, piecewise)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [6, 5])
def test_two_rank_sharding_and_gather(tmp_path, n_clips):
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import stft_oracle as O

    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_clips, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npy"))
    ref = O.melspectrogram(y=O.config_input(n_clips, n=8000), sr=22050, n_fft=512, hop_length=128, n_mels=20)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)
