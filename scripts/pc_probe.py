"""Development probe (GPU box, round 6): the producer / consumer fused mel kernel (ctx option mel_pc, csrc/lra_kernels_pc.h) against
stft2_kernel<OUT_MELR> -- bit equality on BASELINE configs[1] and smaller / ragged shapes, then alternating timings on the same buffers.
python scripts/pc_probe.py [rounds]      (LIBROSA_AMD_LIBRARY=probe/lib_x.so for a probe build of the PC kernel)"""
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
ctx.set_option("variant", 0)
n_fft = 2048
w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)


def run(batch, n, hop, n_mels, power, pc):
    y = bench.make_batch(torch, batch, n, 0, dev)
    pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
    T = ctx.stft_num_frames(pl, n)
    mp = ctx.mel_plan(filters.mel(sr=22050, n_fft=n_fft, n_mels=n_mels))
    Mo = torch.full((batch, n_mels, T), float("nan"), dtype=torch.float32, device=dev)
    ctx.set_option("mel_pc", pc)
    ctx.melspectrogram_exec(pl, mp, y.data_ptr(), batch, n, n, power, Mo.data_ptr())
    torch.cuda.synchronize()
    ctx.set_option("mel_pc", 0)
    return Mo


ok = True
for (batch, n, hop, n_mels, power) in [(2, 22050, 512, 128, 2.0), (3, 9000, 512, 128, 2.0), (5, 661500, 512, 128, 2.0), (4, 100000, 256, 128, 2.0), (4, 100000, 256, 128, 1.0),
                                       (1, 2048, 512, 128, 2.0), (7, 30011, 512, 96, 2.0), (256, 661500, 512, 128, 2.0)]:
    t0 = time.time()
    A = run(batch, n, hop, n_mels, power, 0)
    for pc in (1, 2):
        B = run(batch, n, hop, n_mels, power, pc)
        nan = int(torch.isnan(B).sum())
        rel = float(((A - B).abs() / A.abs()).max())
        good = nan == 0 and rel < 2e-5
        ok &= good
        print(f"batch {batch} n {n} hop {hop} mels {n_mels} power {power}: mel_pc {pc}: max rel diff to the one-wave kernel {rel:.3g}  nan {nan}  {'ok' if good else 'MISMATCH'}", flush=True)
print("PARITY", "ok" if ok else "FAILED", flush=True)

# timings: BASELINE configs[1], alternating
batch, n, hop, n_mels = 256, 661500, 512, 128
y = bench.make_batch(torch, batch, n, 0, dev)
pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
T = ctx.stft_num_frames(pl, n)
mp = ctx.mel_plan(filters.mel(sr=22050, n_fft=n_fft, n_mels=n_mels))
Mo = torch.empty((batch, n_mels, T), dtype=torch.float32, device=dev)
fn = lambda: ctx.melspectrogram_exec(pl, mp, y.data_ptr(), batch, n, n, 2.0, Mo.data_ptr())


def timeit(steps=20):
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


for r in range(rounds):
    for pc in (0, 1, 2):
        ctx.set_option("mel_pc", pc)
        ms = timeit()
        print(f"round {r} mel_pc {pc}: {ms:.4f} ms  {batch * T / ms / 1e3:.1f} M frames/s", flush=True)
ctx.set_option("stft_iters", 0)
ctx.set_option("mel_pc", 0)
