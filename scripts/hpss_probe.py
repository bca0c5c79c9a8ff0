"""Development probe (GPU box): timing of decompose.hpss / effects.hpss on device tensors.  python scripts/hpss_probe.py [clips]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
y = bench.make_batch(torch, clips, 22050 * 30, 0, dev)
D = L.stft(y)
torch.cuda.synchronize()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ctx = L.get_context(0)
if len(sys.argv) > 2:   # counter passes: one call of each kernel
    for tile in (1, 0):
        ctx.set_option("hpss_tile", tile)
        L.decompose.hpss(D, mask=True); L.decompose.hpss(D)
    torch.cuda.synchronize()
    sys.exit(0)
for tile in (1, 0, 1, 0):
  ctx.set_option("hpss_tile", tile)
  for kw in (dict(), dict(kernel_size=(13, 31)), dict(kernel_size=63), dict(mask=True)):
    print(f"tile={tile}", end=" ")
    print(f"decompose.hpss {clips} x 1025 x {D.shape[-1]} {kw}: {timeit(lambda: L.decompose.hpss(D, **kw)):.2f} ms", flush=True)
print(f"effects.hpss {clips} clips x 30 s: {timeit(lambda: L.effects.hpss(y)):.2f} ms (stft {timeit(lambda: L.stft(y)):.2f} ms, istft {timeit(lambda: L.istft(D, length=y.shape[-1])):.2f} ms)", flush=True)
