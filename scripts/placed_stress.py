"""Stress (GPU box, round 6): stft / _spectrogram / istft on device tensors of random batch sizes with placement_retry on -- fresh candidates, recycling, pool eviction
(lra_free_placed: unmap + release) and torch's own allocator interleaved -- every result checked against the torch.empty path."""
import os, sys, time, gc, faulthandler, random
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
dev = torch.device("cuda", 0)
ctx = L.get_context(0)
ctx.PLACED_KEEP_BYTES = 3 << 30   # small pool: evictions all the time
y = bench.make_batch(torch, 128, 661500, 0, dev)
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + float(os.environ.get("STRESS_S", "60"))
it = 0
keep = []
while time.time() < t_end:
    b = random.choice([26, 31, 40, 48, 64, 77, 100, 128])
    hop = random.choice([512, 256, 1024])
    ctx.set_option("placement_retry", 0)
    D0 = L.stft(y[:b], n_fft=2048, hop_length=hop)
    ctx.set_option("placement_retry", random.choice([1, 2, 4]))
    D = L.stft(y[:b], n_fft=2048, hop_length=hop)
    assert torch.equal(D, D0), ("stft", b, hop)
    if random.random() < 0.5:
        S = L._spectrogram(y=y[:b], n_fft=2048, hop_length=hop, power=2)[0]
        assert torch.isfinite(S).all()
        if random.random() < 0.3: keep.append(S[0:1].clone())
    yh = L.istft(D, hop_length=hop, length=y.shape[-1])
    ctx.set_option("placement_retry", 0)
    yh0 = L.istft(D0, hop_length=hop, length=y.shape[-1])
    assert torch.equal(yh, yh0), ("istft", b, hop)
    if random.random() < 0.3: keep.append(D[1])   # a view keeps its buffer alive for a while
    if len(keep) > 4: del keep[: random.randint(1, 4)]
    del D, D0, yh, yh0
    if random.random() < 0.2: torch.cuda.empty_cache()
    if random.random() < 0.1: gc.collect()
    it += 1
    if it % 20 == 0: print(it, "iterations,", len(ctx._placed_log), "fresh allocations logged, pool", {k[0] >> 20: len(v) for k, v in ctx._placed_free.items()}, flush=True)
print("stress ok:", it, "iterations", flush=True)
