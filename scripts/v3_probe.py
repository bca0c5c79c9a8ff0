"""Development probe (GPU box, round 6): the radix 16-16-4 form of the second-generation forward kernels (ctx option v3 = variant 6: adjacent bins per thread,
16-byte row pieces) against the radix 16-8-8 form -- parity on several shapes, then alternating timings on ONE output buffer (the placement of the 2.7 GB
result moves the kernel 0.63-0.75 ms by itself, profiles/r05_pitch.md), packed rows (8 200 B) and rows padded to 128-byte lines (8 320 B).
python scripts/v3_probe.py [rounds]      (LIBROSA_AMD_LIBRARY=probe/lib_x.so for a probe build)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
from librosa_amd import filters

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
ctx = L.get_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
ctx.set_option("autotune", 0)
n_fft = 2048
bins = n_fft // 2 + 1
w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)


def run(batch, n, hop, kind, v3, center=True, pad="constant", power=2.0):
    y = bench.make_batch(torch, batch, n, 0, dev)
    pl = ctx.stft_plan(n_fft, hop, w, center, pad, np.float32)
    T = ctx.stft_num_frames(pl, n)
    ctx.set_option("v3", 2 * v3)
    if kind == 0:
        out = torch.full((batch, T, bins), float("nan"), dtype=torch.complex64, device=dev)
        ctx.stft_exec(pl, y.data_ptr(), batch, n, n, out.data_ptr())
    else:
        out = torch.full((batch, T, bins), float("nan"), dtype=torch.float32, device=dev)
        ctx.spectrogram_exec(pl, y.data_ptr(), batch, n, n, power, out.data_ptr())
    torch.cuda.synchronize()
    ctx.set_option("v3", 1)
    return out


ok = True
for (batch, n, hop, kind, center, pad, power) in [(2, 22050, 512, 0, True, "constant", 2.0), (3, 9000, 512, 0, True, "reflect", 2.0), (5, 661500, 512, 0, True, "constant", 2.0), (4, 100000, 256, 0, True, "edge", 2.0),
                                                  (4, 100000, 1024, 0, False, "constant", 2.0), (3, 50000, 2048, 0, True, "constant", 2.0), (1, 2048, 512, 0, True, "constant", 2.0), (7, 30011, 512, 1, True, "constant", 2.0),
                                                  (7, 30011, 256, 1, True, "symmetric", 1.0), (3, 30011, 512, 1, True, "constant", 1.5), (64, 661500, 512, 0, True, "constant", 2.0)]:
    A = run(batch, n, hop, kind, 0, center, pad, power)
    B = run(batch, n, hop, kind, 1, center, pad, power)
    nan = int(torch.isnan(torch.view_as_real(B) if kind == 0 else B).sum())
    err = float((A - B).abs().max() / A.abs().max())
    good = nan == 0 and err < 3e-6
    ok &= good
    print(f"batch {batch} n {n} hop {hop} kind {kind} center {center} {pad} power {power}: max|v3 - v2| / max|v2| {err:.3g}  nan {nan}  {'ok' if good else 'MISMATCH'}", flush=True)
print("PARITY", "ok" if ok else "FAILED", flush=True)

batch, n, hop = 256, 661500, 512
y = bench.make_batch(torch, batch, n, 0, dev)
pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
T = ctx.stft_num_frames(pl, n)
pitch_pad = 1040
buf = torch.empty((batch, T, pitch_pad), dtype=torch.complex64, device=dev)  # ONE allocation for every variant and both pitches
S = torch.empty((batch, T, bins), dtype=torch.float32, device=dev)


def timeit(fn, steps=20):
    t_end = time.time() + 0.4
    while time.time() < t_end:
        for _ in range(20): fn()
        torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = ctx.event(), ctx.event(); e0.record()
        for _ in range(steps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_ms(e1) / steps)
    return best


algo = batch * T * (bins * 8 + hop * 4)
for r in range(rounds):
    for v3 in (0, 1):
        ctx.set_option("v3", 2 * v3)
        t_packed = timeit(lambda: ctx.stft_exec(pl, y.data_ptr(), batch, n, n, buf.data_ptr()))
        t_padded = timeit(lambda: ctx.stft_exec_strided(pl, 0, y.data_ptr(), batch, n, n, 2.0, buf.data_ptr(), pitch_pad))
        t_power = timeit(lambda: ctx.spectrogram_exec(pl, y.data_ptr(), batch, n, n, 2.0, S.data_ptr()))
        print(f"round {r} v3 {v3}: stft packed {t_packed:.4f} ms ({algo / t_packed / 8e9 * 100:.1f} % of 8 TB/s)  padded rows {t_padded:.4f} ms ({algo / t_padded / 8e9 * 100:.1f} %)  |X|^2 {t_power:.4f} ms", flush=True)
ctx.set_option("v3", 1)
