"""Development check (GPU box): the round's new entry points at BASELINE sizes (256 clips x 30 s)."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import bench, librosa_amd as L, stft_oracle as O
warnings.simplefilter("ignore")
dev = torch.device("cuda", 0)
y = bench.make_batch(torch, 256, 22050 * 30, 0, dev)
yh = y.cpu().numpy()
t0 = time.perf_counter(); D = L.stft(yh); t1 = time.perf_counter() - t0
ref = O.stft(yh[200])
print(f"numpy stft 256 clips: {t1*1e3:.0f} ms, shape {D.shape}, clip 200 err {np.abs(D[200] - ref).max() / np.abs(ref).max():.2e}", flush=True)
t0 = time.perf_counter(); yi = L.istft(D, length=yh.shape[-1]); t1 = time.perf_counter() - t0
print(f"numpy istft 256 clips: {t1*1e3:.0f} ms, max err {np.abs(yi - yh).max():.2e}", flush=True)
del D, yi
S = torch.abs(L.stft(y[:128]))
t0 = time.perf_counter(); yg = L.griffinlim(S, n_iter=4, rng=0); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
Sg = torch.abs(L.stft(yg))
conv = float(torch.linalg.norm(Sg[..., : S.shape[-1]] - S) / torch.linalg.norm(S))
print(f"griffinlim 128 clips x 4 iterations: {t1*1e3:.0f} ms, finite {bool(torch.isfinite(yg).all())}, spectral convergence {conv:.3f}", flush=True)
del S, Sg, yg
Dd = L.stft(y)
t0 = time.perf_counter(); Ds = L.phase_vocoder(Dd, rate=1.25); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
r = O.phase_vocoder(Dd[7].cpu().numpy(), rate=1.25)
e = np.abs(Ds[7].cpu().numpy() - r)
print(f"phase_vocoder 256 clips: {t1*1e3:.1f} ms, shape {tuple(Ds.shape)}, clip 7 err / max {e.max() / np.abs(r).max():.2e}", flush=True)
t0 = time.perf_counter(); ys = L.effects.time_stretch(y[:64], rate=0.8); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
print(f"time_stretch 64 clips: {t1*1e3:.1f} ms, shape {tuple(ys.shape)}, finite {bool(torch.isfinite(ys).all())}")
