"""Development probe (GPU box, round 6): does the inverse kernel care where its OUTPUT (677 MB of samples) or its INPUT spectrum lands?  torch.empty vs lra_malloc_placed, three each."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
from librosa_amd import filters, _arrays
from librosa_amd.core.spectrum import wss_to_norm
dev = torch.device("cuda", 0)
ctx = L.get_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
batch, n, hop, n_fft = 256, 661500, 512, 2048
bins = n_fft // 2 + 1
y = bench.make_batch(torch, batch, n, 0, dev)
w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)
pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
ip = ctx.istft_plan(n_fft, hop, w, True, np.float32)
T = ctx.stft_num_frames(pl, n)
ww = filters.window_sumsquare(window="hann", n_frames=T, n_fft=n_fft, hop_length=hop, dtype=np.float32)[n_fft // 2:]
ww = torch.from_numpy(wss_to_norm(np.ascontiguousarray(np.pad(ww, (0, max(0, n - len(ww))))[:n], dtype=np.float32))).to(dev)
def timeit(fn, steps=10):
    t_end = time.time() + 0.4
    while time.time() < t_end:
        for _ in range(5): fn()
        torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = ctx.event(), ctx.event(); e0.record()
        for _ in range(steps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_ms(e1) / steps)
    return best
ctx.set_option("placement_retry", 4)
def spec(kind):
    return _arrays._placed_tensor(ctx, (batch, T, bins), np.dtype(np.complex64), dev) if kind == "placed" else torch.empty((batch, T, bins), dtype=torch.complex64, device=dev)
def sig(kind):
    if kind == "placed":
        rows = (batch * n * 4 + 8191) // 8192
        return _arrays._placed_tensor(ctx, (rows, 1, 2048), np.dtype(np.float32), dev).reshape(-1)[: batch * n].view(batch, n)
    return torch.empty((batch, n), dtype=torch.float32, device=dev)
keep = []
for dk in ("torch.empty", "placed"):
    for yk in ("torch.empty", "placed"):
        res = []
        for a in range(3):
            D = spec(dk); yr = sig(yk); keep += [D, yr]
            ctx.stft_exec(pl, y.data_ptr(), batch, n, n, D.data_ptr())
            ms = timeit(lambda: ctx.istft_exec_norm(ip, D.data_ptr(), batch, T * bins, bins, T, ww.data_ptr(), yr.data_ptr(), n, n))
            res.append(f"{ms:.4f}")
        print(f"istft: spectrum {dk:11s} output {yk:11s}: " + "  ".join(res), flush=True)
        del D, yr; keep.clear(); gc.collect(); ctx.placed_release_all(); torch.cuda.empty_cache()
