"""Development probe (GPU box, round 6): the other store-heavy rows on device tensors with torch.empty results (placement_retry 0) and lra_malloc_placed results (4):
decompose.hpss, griffinlim, effects.time_stretch, _spectrogram, power_to_db."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
dev = torch.device("cuda", 0)
ctx = L.get_context(0)
y = bench.make_batch(torch, 64, 661500, 0, dev)
def timeit(fn, steps=5, reps=3):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(reps):
        t0 = time.perf_counter()
        for _ in range(steps): r = fn(); del r
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    return best
D = L.stft(y[:48], n_fft=2048, hop_length=512)
S = D.abs()
cases = {
    "decompose.hpss 48 clips": lambda: L.decompose.hpss(D),
    "griffinlim 48 clips x 8 it": lambda: L.griffinlim(S, n_iter=8, hop_length=512, rng=0),
    "time_stretch 48 clips": lambda: L.effects.time_stretch(y[:48], rate=1.25),
    "_spectrogram 64 clips": lambda: L._spectrogram(y=y, n_fft=2048, hop_length=512, power=2)[0],
    "stft+istft 64 clips": lambda: L.istft(L.stft(y, n_fft=2048, hop_length=512), hop_length=512, length=y.shape[-1]),
}
for name, fn in cases.items():
    res = []
    for k in (0, 4, 0, 4):
        ctx.set_option("placement_retry", k)
        gc.collect(); ctx.placed_release_all(); torch.cuda.empty_cache()
        res.append(f"{'placed' if k else 'torch '} {timeit(fn):.3f}")
    print(f"{name}: " + "   ".join(res) + " ms", flush=True)
