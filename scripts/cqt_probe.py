"""Development probe (GPU box): librosa_amd.cqt on 64 x 30 s clips, fused octaves on / off.  python scripts/cqt_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
from librosa_amd.core import constantq
dev = torch.device("cuda", 0)
y = bench.make_batch(torch, 64, 22050 * 30, 0, dev)
for rt in ("polyphase", "soxr_hq"):
    outs = {}
    for fused in (True, False, True, False):
        constantq.FUSED_OCTAVES = fused
        fn = lambda: L.cqt(y, sr=22050, hop_length=512, res_type=rt)
        outs[fused] = fn(); torch.cuda.synchronize()
        t_end = time.time() + 0.3
        while time.time() < t_end:
            fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): fn()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
        print(f"cqt res_type={rt} fused={fused}: {best * 1e3:.3f} ms per 64 clips", flush=True)
    d = (outs[True] - outs[False]).abs().max() / outs[False].abs().max()
    print(f"  fused vs unfused: max |diff| / max = {float(d):.2e}", flush=True)
