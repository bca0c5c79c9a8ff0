"""Development probe (GPU box), round 6: does MADV_HUGEPAGE on the fresh NumPy result shorten the host pipeline's first-touch-bound stft?  python scripts/thp_probe.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import librosa_amd as L
print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip(), flush=True)
libc = ctypes.CDLL("libc.so.6", use_errno=True)
rng = np.random.default_rng(0)
y = (0.1 * rng.standard_normal((64, 22050 * 30))).astype(np.float32)
D0 = L.stft(y)
shape = D0.shape  # (64, 1025, T)
T = shape[-1]
def fresh(advise):
    buf = np.empty((64, T, 1025), dtype=np.complex64)
    if advise:
        addr = buf.ctypes.data; n = buf.nbytes
        a0 = (addr + (1 << 21) - 1) & ~((1 << 21) - 1); a1 = (addr + n) & ~((1 << 21) - 1)
        rc = libc.madvise(ctypes.c_void_p(a0), ctypes.c_size_t(a1 - a0), 14)
        if rc != 0: print("madvise failed", ctypes.get_errno())
    return np.swapaxes(buf, -1, -2)
for rep in range(3):
    for advise in (0, 1):
        ts = []
        for _ in range(4):
            out = fresh(advise)
            t0 = time.perf_counter(); L.stft(y, out=out); ts.append(time.perf_counter() - t0)
            del out
        print(f"advise {advise}: stft(out=fresh) {min(ts) * 1e3:.1f} ms (all {[round(t * 1e3, 1) for t in ts]})", flush=True)
ts = []
out = fresh(0); L.stft(y, out=out)
for _ in range(4):
    t0 = time.perf_counter(); L.stft(y, out=out); ts.append(time.perf_counter() - t0)
print(f"reused (already touched) out: {min(ts) * 1e3:.1f} ms", flush=True)
ts = []
for _ in range(4):
    t0 = time.perf_counter(); D = L.stft(y); ts.append(time.perf_counter() - t0); del D
print(f"plain stft(y): {min(ts) * 1e3:.1f} ms", flush=True)
