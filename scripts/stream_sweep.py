"""Development probe (GPU box): the arithmetic-free access streams of the forward / inverse transform (lra_probe_stream) over strip length and
resident waves per CU, BASELINE configs[1] shapes.  python scripts/stream_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
dev = torch.device("cuda", 0)
ctx = L.get_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
n, batch, n_fft, hop = 22050 * 30, 256, 2048, 512
T = 1 + n // hop
y = bench.make_batch(torch, batch, n, 0, dev)
D = torch.zeros((batch, T, n_fft // 2 + 1), dtype=torch.complex64, device=dev)
yr = torch.empty((batch, n), dtype=torch.float32, device=dev)
by = batch * T * ((n_fft // 2 + 1) * 8 + hop * 4)
def timeit(fn, steps=20):
    t_end = time.time() + 0.3
    while time.time() < t_end:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = ctx.event(), ctx.event(); e0.record()
        for _ in range(steps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_ms(e1) / steps)
    return best
for direction, src, dst in ((0, y.data_ptr(), D.data_ptr()), (1, D.data_ptr(), yr.data_ptr())):
    for wpc in (8, 12, 16, 24, 32):
        row = []
        for strip in (54, 81, 162, 323, 646):
            ms = timeit(lambda: ctx.probe_stream(direction, src, dst, batch, T, n_fft, hop, n, strip, wpc))
            row.append(f"{strip}: {ms:.3f} ms {by / ms / 1e6:5.0f} GB/s")
        print(f"dir {direction} waves/CU {wpc:2d} | " + " | ".join(row), flush=True)
