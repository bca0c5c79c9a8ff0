"""Development probe (GPU box): NumPy end-to-end rate of the public drop-in through the host pipeline vs chunk size / threads."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import librosa_amd as L
ctx = L.get_context(0)
rng = np.random.default_rng(0)
nb = 64
y = (0.1 * rng.standard_normal((nb, 661500))).astype(np.float32)
T = 1292
L.stft(y[:2], n_fft=2048, hop_length=512)
for mb in (16, 48, 128):
    for th in (1, 4, 8, 16):
        ctx.set_option("pipe_chunk_mb", mb); ctx.set_option("pipe_threads", th)
        L.stft(y, n_fft=2048, hop_length=512)
        t0 = time.perf_counter(); D = L.stft(y, n_fft=2048, hop_length=512); ts = time.perf_counter() - t0
        L.feature.melspectrogram(y=y, sr=22050)
        t0 = time.perf_counter(); M = L.feature.melspectrogram(y=y, sr=22050); tm = time.perf_counter() - t0
        print(f"chunk {mb:4d} MB threads {th:2d}: stft {ts*1e3:7.2f} ms ({(y.nbytes + D.nbytes)/ts/1e9:5.1f} GB/s host bytes)   mel {tm*1e3:7.2f} ms ({(y.nbytes + M.nbytes)/tm/1e9:5.1f} GB/s)", flush=True)
ctx.set_option("pipe_chunk_mb", 128); ctx.set_option("pipe_threads", 8)
D = L.stft(y, n_fft=2048, hop_length=512)
L.istft(D, hop_length=512, length=y.shape[-1])
t0 = time.perf_counter(); yi = L.istft(D, hop_length=512, length=y.shape[-1]); ti = time.perf_counter() - t0
print(f"istft numpy end to end: {ti*1e3:.2f} ms ({(D.nbytes + yi.nbytes)/ti/1e9:.1f} GB/s host bytes), max err {np.abs(yi - y).max():.2e}")
t0 = time.perf_counter(); np.isfinite(y).all(); print("np.isfinite scan", (time.perf_counter() - t0) * 1e3, "ms")
t0 = time.perf_counter(); y.copy(); print("np copy 169 MB", (time.perf_counter() - t0) * 1e3, "ms")
