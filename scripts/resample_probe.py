"""Development probe (GPU box): timing of librosa_amd.resample and of cqt with each resampler family.  python scripts/resample_probe.py [clips]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
y = bench.make_batch(torch, clips, 22050 * 30, 0, dev)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
mb = y.numel() * 4 / 1e6
for orig, target in ((22050, 16000), (22050, 11025), (22050, 8000), (16000, 22050)):
    for rt in ("fft", "polyphase", "soxr_hq"):
        ms = timeit(lambda: L.resample(y, orig_sr=orig, target_sr=target, res_type=rt))
        print(f"resample {clips} x 30 s {orig}->{target} {rt}: {ms:.2f} ms  ({mb * (1 + target / orig) / ms:.0f} GB/s in+out)", flush=True)
for rt in ("polyphase", "soxr_hq", "fft"):
    print(f"cqt {clips} x 30 s res_type={rt}: {timeit(lambda: L.cqt(y, sr=22050, res_type=rt)):.2f} ms", flush=True)
