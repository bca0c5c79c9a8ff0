"""Development probe (GPU box): BASELINE configs[4] -- the STFTs at n_fft 512 / 2048 / 8192 over the same batch back to back on one
stream vs each on its own HIP stream (concurrent kernels).  python scripts/multi_stream_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
from librosa_amd import filters
dev = torch.device("cuda", 0)
ctx = L.get_context(0)
main = torch.cuda.current_stream(dev)
ctx.set_stream(main.cuda_stream)
n, batch, hop = 22050 * 30, 256, 512
y = bench.make_batch(torch, batch, n, 0, dev)
plans, outs = {}, {}
for nf in (512, 2048, 8192):
    w = np.asarray(filters.get_window("hann", nf, fftbins=True), dtype=np.float32)
    plans[nf] = ctx.stft_plan(nf, hop, w, True, "constant", np.float32)
    T = ctx.stft_num_frames(plans[nf], n)
    outs[nf] = torch.empty((batch, T, nf // 2 + 1), dtype=torch.complex64, device=dev)
streams = {nf: torch.cuda.Stream(device=dev) for nf in plans}
def sequential(order=(512, 2048, 8192)):
    for nf in order: ctx.stft_exec(plans[nf], y.data_ptr(), batch, n, n, outs[nf].data_ptr())
def concurrent(order=(8192, 2048, 512)):
    for nf in order:
        streams[nf].wait_stream(main)
        ctx.set_stream(streams[nf].cuda_stream)
        ctx.stft_exec(plans[nf], y.data_ptr(), batch, n, n, outs[nf].data_ptr())
    ctx.set_stream(main.cuda_stream)
    for nf in order: main.wait_stream(streams[nf])
def timeit(fn, reps=10):
    t_end = time.time() + 0.4
    while time.time() < t_end:
        fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(reps): fn()
        e1.record(main); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
for name, fn in (("sequential 512,2048,8192", sequential), ("concurrent 8192,2048,512", concurrent), ("concurrent 512,2048,8192", lambda: concurrent((512, 2048, 8192))),
                 ("sequential again", sequential)):
    print(f"{name}: {timeit(fn):.3f} ms", flush=True)
