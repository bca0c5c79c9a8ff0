"""Development probe (GPU box): is the complex STFT's time sensitive to where its 2.7 GB output lands?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
from librosa_amd import filters
dev = torch.device("cuda", 0)
ctx = L.get_context(0); ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
n, batch = 22050 * 30, 256
y = bench.make_batch(torch, batch, n, 0, dev)
w = np.asarray(filters.get_window("hann", 2048, fftbins=True), dtype=np.float32)
pl = ctx.stft_plan(2048, 512, w, True, "constant", np.float32)
T = ctx.stft_num_frames(pl, n)
def timeit(D):
    for _ in range(10): ctx.stft_exec(pl, y.data_ptr(), batch, n, n, D.data_ptr())
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        e0, e1 = ctx.event(), ctx.event(); e0.record()
        for _ in range(20): ctx.stft_exec(pl, y.data_ptr(), batch, n, n, D.data_ptr())
        e1.record(); torch.cuda.synchronize(); best.append(e0.elapsed_ms(e1) / 20)
    return min(best), max(best)
keep = []
for i in range(7):
    pad = torch.empty(int((i * 37 + 1) * 2**20 * 3.1), dtype=torch.uint8, device=dev) if i else None
    D = torch.empty((batch, T, 1025), dtype=torch.complex64, device=dev)
    lo, hi = timeit(D)
    print(f"alloc {i}: D at 0x{D.data_ptr():x} (mod 2MB {D.data_ptr() % (2<<20):#x}, mod 1GB {D.data_ptr() % (1<<30) / 2**20:.0f} MB): {lo:.3f}-{hi:.3f} ms", flush=True)
    keep.append((pad, D))
    if len(keep) > 2: keep.pop(0)
raw = torch.empty(batch * T * 1025 * 8 + (1 << 20), dtype=torch.uint8, device=dev)
for off in (0, 8, 64, 256, 4096, 65536):
    D = raw[off : off + batch * T * 1025 * 8].view(torch.complex64).view(batch, T, 1025)
    lo, hi = timeit(D)
    print(f"offset {off:6d}: {lo:.3f}-{hi:.3f} ms", flush=True)
