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


def _run_bench(extra, env_extra=None, timeout=240):
    import json
    import subprocess

    env = dict(os.environ, LRA_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's command when WORLD_SIZE is unset) must
    run TWO ranks and print ONE line with n_gpus = 2 and configs[2]'s split (512 clips per GPU) -- not one rank on 256 clips.  --dry-run keeps
    the kernels out (no GPU here); everything else -- re-exec under torch.distributed.run, rendezvous on 127.0.0.1, barriers, max over ranks, rank 0
    printing -- is the real control flow."""
    r, lines = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-side", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["config"]["clips_per_gpu"] == 512 and line["config"]["self_launched"] is True
    assert line["value"] is None and "dry-run" in line["data"]


def test_bench_refuses_when_gpus_and_world_size_disagree():
    r, lines = _run_bench(["--gpus", "4", "--dry-run"], env_extra={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and not lines
    assert "must agree" in r.stderr


def test_bench_refuses_more_gpus_than_devices():
    """Without the gloo override the self-launch counts devices first and fails loudly (no GPU in this container: 0 < 2)."""
    import subprocess

    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices are visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LRA_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode != 0 and "ROCm device(s) visible" in r.stderr
