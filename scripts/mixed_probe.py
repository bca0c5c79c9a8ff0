"""Development probe (GPU box): the mixed-radix fused kernels on the 256 x 30 s batch.  python scripts/mixed_probe.py [n_fft hop n_mels sr] [what=mel|stft] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
n_fft, hop, n_mels, sr = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (400, 160, 80, 16000)))
what = sys.argv[5] if len(sys.argv) > 5 else "mel"
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
dev = torch.device("cuda", 0)
ctx = L.get_context(0)
ctx.set_option("mixed", int(os.environ.get("LRA_MIXED_OPT", "1")))
y = bench.make_batch(torch, 256, sr * 30, 0, dev)
fn = (lambda: L.feature.melspectrogram(y=y, sr=sr, n_fft=n_fft, hop_length=hop, n_mels=n_mels, check_finite=False)) if what == "mel" else (lambda: L.stft(y, n_fft=n_fft, hop_length=hop, check_finite=False))
T = int(fn().shape[-1])
t_end = time.time() + 0.3
while time.time() < t_end:
    fn(); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / steps)
by = 256 * T * (hop * 4 + (n_mels * 4 if what == "mel" else (n_fft // 2 + 1) * 8))
print(f"{what} n_fft {n_fft} hop {hop}: frames {256 * T} {best * 1e3:.3f} ms  {256 * T / best / 1e6:.0f} M frames/s  {by / best / 1e9:.0f} GB/s algorithmic", flush=True)
