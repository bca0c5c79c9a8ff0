"""Development probe (GPU box): the NumPy drop-in's host pipeline under its two knobs (ctx options pipe_chunk_mb, pipe_threads): stft / melspectrogram / istft of 64 x 30 s.  python scripts/pipe_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import librosa_amd as L
rng = np.random.default_rng(0)
y = (0.1 * rng.standard_normal((64, 22050 * 30))).astype(np.float32)
ctx = L.get_context(0)
D = L.stft(y)
def best(fn, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0); del r
    return min(ts) * 1e3
for chunk in (128, 64, 32, 16):
    for threads in (8, 16):
        ctx.set_option("pipe_chunk_mb", chunk); ctx.set_option("pipe_threads", threads)
        print(f"chunk {chunk:4d} MB threads {threads:2d}: stft {best(lambda: L.stft(y)):6.1f} ms  mel {best(lambda: L.feature.melspectrogram(y=y, sr=22050)):6.1f} ms  istft {best(lambda: L.istft(D, length=y.shape[-1])):6.1f} ms", flush=True)
