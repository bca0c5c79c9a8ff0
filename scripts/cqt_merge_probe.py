"""Development probe (GPU box), round 5: the constant-Q transform with its octaves 1.. merged into one launch per frame length (ctx option cqt_merge) against one launch per
octave, with and without the finite check.  python scripts/cqt_merge_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
dev = torch.device("cuda", 0)
ctx = L.get_context(0)
y = bench.make_batch(torch, 64, 22050 * 30, 0, dev)
def timeit(fn, steps=20):
    t_end = time.time() + 0.3
    while time.time() < t_end:
        fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / steps)
    return best * 1e3
ref = {}
for rt in ("polyphase", "soxr_hq"):
    for merge in (2, 1, 2, 1, 0):
        ctx.set_option("cqt_merge", merge)
        out = L.cqt(y, sr=22050, res_type=rt)
        key = rt
        same = "first" if key not in ref else ("== unmerged" if torch.equal(out, ref[key]) else "MISMATCH")
        ref.setdefault(key, out)
        a = timeit(lambda: L.cqt(y, sr=22050, res_type=rt))
        b = timeit(lambda: L.cqt(y, sr=22050, res_type=rt, check_finite=False))
        print(f"cqt 64 x 30 s res_type {rt} cqt_merge {merge}: {a:.3f} ms with the finite check, {b:.3f} without  [{same}]", flush=True)
