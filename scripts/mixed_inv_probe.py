"""Development probe (GPU box): the mixed-radix fused inverse on the 256 x 30 s batch.  python scripts/mixed_inv_probe.py [n_fft hop sr]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
n_fft, hop, sr = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (400, 160, 16000)))
dev = torch.device("cuda", 0)
L.get_context(0).set_option("mixed", int(os.environ.get("LRA_MIXED_OPT", "1")))
L.get_context(0).set_option("mixed_inv_pow2", int(os.environ.get("LRA_MIXED_INV_POW2", "1")))   # powers of two with an unaligned hop: fused gather kernel (1) / istft_kernel's general mode (0)
y = bench.make_batch(torch, 256, sr * 30, 0, dev)
D = L.stft(y, n_fft=n_fft, hop_length=hop, check_finite=False)
fn = lambda: L.istft(D, hop_length=hop, n_fft=n_fft, length=y.shape[-1])
fn(); torch.cuda.synchronize()
t_end = time.time() + 0.3
while time.time() < t_end:
    fn(); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
T = D.shape[-1]
print(f"istft n_fft {n_fft} hop {hop}: frames {256 * T} {best * 1e3:.3f} ms  {256 * T * (hop * 4 + (n_fft // 2 + 1) * 8) / best / 1e9:.0f} GB/s algorithmic  err {float((fn() - y).abs().max()):.2e}", flush=True)
