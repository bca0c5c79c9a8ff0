"""Development probe (GPU box, round 6): the inverse kernel as radices 8-8-16 (8-byte spectrum loads) and as radices 4-16-16 (ctx option istft16: 16-byte loads), alternating on the same
buffers (spectrum and output both from lra_malloc_placed); parity of the two against each other and the round trip."""
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
batch, n, n_fft = 256, 661500, 2048
bins = n_fft // 2 + 1
y = bench.make_batch(torch, batch, n, 0, dev)
w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)
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
for hop in (512, 256, 1024):
    pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
    ip = ctx.istft_plan(n_fft, hop, w, True, np.float32)
    T = ctx.stft_num_frames(pl, n)
    b = batch if hop >= 512 else batch // 2
    ww = filters.window_sumsquare(window="hann", n_frames=T, n_fft=n_fft, hop_length=hop, dtype=np.float32)[n_fft // 2:]
    ww = torch.from_numpy(wss_to_norm(np.ascontiguousarray(np.pad(ww, (0, max(0, n - len(ww))))[:n], dtype=np.float32))).to(dev)
    D = _arrays._placed_tensor(ctx, (b, T, bins), np.dtype(np.complex64), dev)
    outs = {}
    yr = _arrays._placed_tensor(ctx, (b, n), np.dtype(np.float32), dev, flat=True)
    ctx.stft_exec(pl, y.data_ptr(), b, n, n, D.data_ptr())
    fn = lambda: ctx.istft_exec_norm(ip, D.data_ptr(), b, T * bins, bins, T, ww.data_ptr(), yr.data_ptr(), n, n)
    for opt in (0, 1):
        ctx.set_option("istft16", opt)
        yr.fill_(float("nan")); fn(); torch.cuda.synchronize()
        outs[opt] = yr.clone()
    d = float((outs[0] - outs[1]).abs().max()); nan = int(torch.isnan(outs[1]).sum())
    err = ((y[:b] - outs[1]).double() ** 2).sum(-1)
    snr = float((10 * torch.log10((y[:b].double() ** 2).sum(-1) / err)).min())
    print(f"hop {hop}: max |4-16-16 - 8-8-16| {d:.3g}  nan {nan}  round-trip SNR min {snr:.1f} dB", flush=True)
    for r in range(3):
        res = []
        for opt in (0, 1):
            ctx.set_option("istft16", opt)
            res.append(timeit(fn))
        by = b * T * (bins * 8 + hop * 4)
        print(f"hop {hop} round {r}: 8-8-16 {res[0]:.4f} ms ({by / res[0] / 8e9 * 100:.1f} %)   4-16-16 {res[1]:.4f} ms ({by / res[1] / 8e9 * 100:.1f} %)   {100 * (res[1] / res[0] - 1):+.1f} %", flush=True)
    del D, yr, outs; gc.collect(); ctx.placed_release_all()
ctx.set_option("istft16", 0)
