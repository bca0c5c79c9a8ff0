"""Development probe (GPU box, round 6): configs[4]'s n_fft = 512 and 8192 legs on torch.empty results against lra_malloc_placed results (three allocations each)."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
from librosa_amd import filters, _arrays
dev = torch.device("cuda", 0)
ctx = L.get_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
batch, n, hop = 256, 661500, 512
y = bench.make_batch(torch, batch, n, 0, dev)
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
for nf in [int(v) for v in os.environ.get("PROBE_NFFT", "512,2048,8192").split(",")]:
    w = np.asarray(filters.get_window("hann", nf, fftbins=True), dtype=np.float32)
    pl = ctx.stft_plan(nf, hop, w, True, "constant", np.float32)
    T = ctx.stft_num_frames(pl, n)
    shape = (batch, T, nf // 2 + 1)
    for kind in ("torch.empty", "placed"):
        res = []
        keep = []
        for a in range(int(os.environ.get("PROBE_ALLOCS", "3"))):
            if kind == "placed":
                ctx.set_option("placement_retry", 4)
                t0 = time.perf_counter(); D = _arrays._placed_tensor(ctx, shape, np.dtype(np.complex64), dev); dt = (time.perf_counter() - t0) * 1e3
                info = f"(alloc {dt:.0f} ms, tried {ctx._placed_log[-1][3]})"
            else:
                D = torch.empty(shape, dtype=torch.complex64, device=dev); info = ""
                keep.append(D)  # hold them all: a freed block would come straight back
            ms = timeit(lambda: ctx.stft_exec(pl, y.data_ptr(), batch, n, n, D.data_ptr()))
            res.append(f"{ms:.3f} {info}")
            if kind == "placed":
                del D; gc.collect(); ctx.placed_release_all()
        del keep; gc.collect(); torch.cuda.empty_cache()
        print(f"n_fft {nf} {kind}: " + "  ".join(res), flush=True)
