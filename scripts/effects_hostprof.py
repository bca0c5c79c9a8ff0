"""Development probe (GPU box): where the host side of effects.hpss / time_stretch / pitch_shift goes (cProfile).  python scripts/effects_hostprof.py"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, librosa_amd as L
dev = torch.device("cuda", 0)
y = bench.make_batch(torch, 32, 22050 * 30, 0, dev)
for name, fn in (("effects.hpss", lambda: L.effects.hpss(y)), ("istft", None)):
    if fn is None:
        D = L.stft(y)
        fn = lambda: L.istft(D, length=y.shape[-1])
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(50): fn()
    e1.record(); t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name}: wall to issue {t_issue * 20:.3f} ms/call; GPU span {e0.elapsed_time(e1) / 50:.3f} ms/call", flush=True)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(100): fn()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
