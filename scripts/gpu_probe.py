"""Development probe (GPU box): ablation timings and variant debugging.  Not part of the product."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch

import bench
import librosa_amd as L
import stft_oracle as O
from librosa_amd import filters

dev = torch.device("cuda", 0)
ctx = L.get_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
n = 22050 * 30
batch = 256
y = bench.make_batch(torch, batch, n, 0, dev)
window = np.asarray(filters.get_window("hann", 2048, fftbins=True), dtype=np.float32)
plan = ctx.stft_plan(2048, 512, window, True, "constant", np.float32)
mel_plan = ctx.mel_plan(filters.mel(sr=22050, n_fft=2048, n_mels=128))
T = ctx.stft_num_frames(plan, n)
D = torch.empty((batch, T, 1025), dtype=torch.complex64, device=dev)
M = torch.empty((batch, 128, T), dtype=torch.float32, device=dev)


iplan = ctx.istft_plan(2048, 512, window, True, np.float32)
_w = filters.window_sumsquare(window="hann", n_frames=T, n_fft=2048, hop_length=512, dtype=np.float32)[1024:]
wss = torch.from_numpy(np.ascontiguousarray(np.pad(_w, (0, max(0, n - len(_w))))[:n], dtype=np.float32)).to(dev)
yrec = torch.empty((batch, n), dtype=torch.float32, device=dev)


def timeit(fn, steps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_ms(e1) / steps


what = sys.argv[1] if len(sys.argv) > 1 else "ablate"
if what == "ablate":
    # run once per probe build:  LIBROSA_AMD_LIBRARY=probe/lib_abN.so python scripts/gpu_probe.py ablate
    # (hipcc -DLRA_PROBE_ONLY -DLRA_ABLATE=N: 1 = no spectrum stores, 2 = no FFT math, stores only)
    for variant in (0, 4):
        ctx.set_option("variant", variant)
        for iters in (16, 32):
            ctx.set_option("stft_iters", iters)
            ms = timeit(lambda: ctx.stft_exec(plan, y.data_ptr(), batch, n, n, D.data_ptr()))
            print(f"{os.environ.get('LIBROSA_AMD_LIBRARY', 'product')}: variant {variant} iters {iters}: stft {ms:.3f} ms ({batch * T / ms / 1e3:.1f} Mframes/s)", flush=True)
elif what == "iters":
    ctx.set_option("mel_runs", int(os.environ.get("PROBE_MEL_RUNS", "1")))
    ctx.set_option("autotune", 0)
    for variant in (0, 1, 4):
        ctx.set_option("variant", variant)
        for iters in [int(v) for v in os.environ.get("PROBE_ITERS", "16,21,27,32,41,54,62,81,108,162,324").split(",")]:
            ctx.set_option("stft_iters", iters)
            ms = timeit(lambda: ctx.stft_exec(plan, y.data_ptr(), batch, n, n, D.data_ptr()))
            msm = timeit(lambda: ctx.melspectrogram_exec(plan, mel_plan, y.data_ptr(), batch, n, n, 2.0, M.data_ptr()))
            msi = timeit(lambda: ctx.istft_exec(iplan, D.data_ptr(), batch, T * 1025, 1025, T, wss.data_ptr(), yrec.data_ptr(), n, n))
            print(f"variant {variant} iters {iters}: stft {ms:.3f} ms ({batch * T / ms / 1e3:.1f} Mframes/s)   mel {msm:.3f} ms ({batch * T / msm / 1e3:.1f} Mframes/s)   istft {msi:.3f} ms ({batch * T / msi / 1e3:.1f} Mframes/s)", flush=True)
elif what == "remap":
    # XCD-aware workgroup map on/off x frames per slot, all three kernels (run once per library build)
    ctx.set_option("autotune", 0)
    ctx.set_option("variant", 0)
    for remap in (0, 1):
        ctx.set_option("xcd_remap", remap)
        for iters in (0, 54, 81, 108, 162, 324):
            ctx.set_option("stft_iters", iters)
            ctx.set_option("istft_strip_groups", iters)
            ms = min(timeit(lambda: ctx.stft_exec(plan, y.data_ptr(), batch, n, n, D.data_ptr())) for _ in range(3))
            msm = min(timeit(lambda: ctx.melspectrogram_exec(plan, mel_plan, y.data_ptr(), batch, n, n, 2.0, M.data_ptr())) for _ in range(3))
            msi = min(timeit(lambda: ctx.istft_exec(iplan, D.data_ptr(), batch, T * 1025, 1025, T, wss.data_ptr(), yrec.data_ptr(), n, n)) for _ in range(3))
            print(f"{os.environ.get('LIBROSA_AMD_LIBRARY', 'product')}: remap {remap} iters {iters:3d}: stft {ms:.3f} ms ({batch * T * 10248 / ms / 1e6:5.0f} GB/s)   mel {msm:.3f} ms ({batch * T / msm / 1e3:.1f} Mframes/s)   "
                  f"istft {msi:.3f} ms ({batch * T * 10248 / msi / 1e6:5.0f} GB/s)", flush=True)
elif what == "v2":
    # second-generation forward kernel on / off x XCD map x frames per slot (run once per library build)
    ctx.set_option("autotune", 0)
    ctx.set_option("variant", 0)
    Sp = torch.empty((batch, T, 1025), dtype=torch.float32, device=dev)
    for v2 in (0, 1):
        ctx.set_option("v2", v2)
        for remap in (0, 1):
            ctx.set_option("xcd_remap", remap)
            for iters in (0, 41, 54, 81, 108, 162, 324):
                ctx.set_option("stft_iters", iters)
                ms = min(timeit(lambda: ctx.stft_exec(plan, y.data_ptr(), batch, n, n, D.data_ptr())) for _ in range(3))
                msp = min(timeit(lambda: ctx.spectrogram_exec(plan, y.data_ptr(), batch, n, n, 2.0, Sp.data_ptr())) for _ in range(3))
                print(f"{os.environ.get('LIBROSA_AMD_LIBRARY', 'product')}: v2 {v2} remap {remap} iters {iters:3d}: stft {ms:.3f} ms ({batch * T * 10248 / ms / 1e6:5.0f} GB/s = {batch * T * 10248 / ms / 8e7:4.1f} %)   "
                      f"power {msp:.3f} ms ({batch * T * 6148 / msp / 1e6:5.0f} GB/s)", flush=True)
elif what == "v2b":
    # second-generation kernel only: XCD map x frames per slot (store-form / ablation variants are separate builds)
    ctx.set_option("autotune", 0)
    ctx.set_option("variant", 0)
    ctx.set_option("v2", 1)
    for remap in (0, 1):
        ctx.set_option("xcd_remap", remap)
        for iters in (0, 54, 81, 108, 162, 324):
            ctx.set_option("stft_iters", iters)
            ms = min(timeit(lambda: ctx.stft_exec(plan, y.data_ptr(), batch, n, n, D.data_ptr())) for _ in range(3))
            print(f"{os.environ.get('LIBROSA_AMD_LIBRARY', 'product')}: remap {remap} iters {iters:3d}: stft {ms:.3f} ms ({batch * T * 10248 / ms / 1e6:5.0f} GB/s = {batch * T * 10248 / ms / 8e7:4.1f} %)", flush=True)
elif what == "mel2":
    # fused melspectrogram: second-generation core on / off x frames per slot
    ctx.set_option("autotune", 0)
    ctx.set_option("variant", 0)
    for v2 in (0, 1):
        ctx.set_option("v2", v2)
        for iters in (0, 27, 41, 54, 81, 108, 162):
            ctx.set_option("stft_iters", iters)
            msm = min(timeit(lambda: ctx.melspectrogram_exec(plan, mel_plan, y.data_ptr(), batch, n, n, 2.0, M.data_ptr())) for _ in range(3))
            print(f"{os.environ.get('LIBROSA_AMD_LIBRARY', 'product')}: v2 {v2} iters {iters:3d}: mel {msm:.3f} ms ({batch * T / msm / 1e3:.1f} Mframes/s)", flush=True)
elif what == "survey":
    # other common configurations: ms and algorithmic GB/s for stft / melspectrogram / istft
    ctx.set_option("variant", -1)
    for n_fft, hop, dt in ((2048, 512, np.float32), (1024, 256, np.float32), (512, 128, np.float32), (4096, 1024, np.float32), (2048, 1024, np.float32), (2048, 256, np.float32),
                           (1024, 512, np.float32), (400, 160, np.float32), (2048, 512, np.float64)):
        tdt = torch.float32 if dt == np.float32 else torch.float64
        es = 4 if dt == np.float32 else 8
        b = batch if dt == np.float32 else batch // 2
        yy = y[:b].to(tdt)
        w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=dt)
        pl = ctx.stft_plan(n_fft, hop, w, True, "constant", dt)
        Tn = ctx.stft_num_frames(pl, n)
        bins = n_fft // 2 + 1
        Dn = torch.empty((b, Tn, bins), dtype=torch.complex64 if dt == np.float32 else torch.complex128, device=dev)
        Mn = torch.empty((b, 128, Tn), dtype=tdt, device=dev)
        mp = ctx.mel_plan(filters.mel(sr=22050, n_fft=n_fft, n_mels=128, dtype=dt))
        ip = ctx.istft_plan(n_fft, hop, w, True, dt)
        ww = filters.window_sumsquare(window="hann", n_frames=Tn, n_fft=n_fft, hop_length=hop, dtype=dt)[n_fft // 2:]
        ww = torch.from_numpy(np.ascontiguousarray(np.pad(ww, (0, max(0, n - len(ww))))[:n], dtype=dt)).to(dev)
        yr = torch.empty((b, n), dtype=tdt, device=dev)
        ms = timeit(lambda: ctx.stft_exec(pl, yy.data_ptr(), b, n, n, Dn.data_ptr()), 5)
        mm = timeit(lambda: ctx.melspectrogram_exec(pl, mp, yy.data_ptr(), b, n, n, 2.0, Mn.data_ptr()), 5)
        mi = timeit(lambda: ctx.istft_exec(ip, Dn.data_ptr(), b, Tn * bins, bins, Tn, ww.data_ptr(), yr.data_ptr(), n, n), 5)
        by = b * Tn * (bins * 2 * es + hop * es)
        print(f"n_fft {n_fft:5d} hop {hop:5d} {np.dtype(dt).name}: frames {b * Tn:8d}  stft {ms:7.3f} ms ({by / ms / 1e6:6.0f} GB/s)  mel {mm:7.3f} ms ({b * Tn / mm / 1e3:6.1f} Mfr/s)  "
              f"istft {mi:7.3f} ms ({by / mi / 1e6:6.0f} GB/s)", flush=True)
        del Dn, Mn, yr
elif what == "occupancy":
    # stft kernel time vs resident waves per CU (one wave64 per workgroup, 17 KB of LDS per slot + pad); run once per (ablation) build
    ctx.set_option("autotune", 0)
    ctx.set_option("variant", 0)
    for waves in (9, 8, 6, 4, 3, 2, 1):
        pad = max(0, int(160 * 1024 / waves) - 17 * 1024 - 512) & ~255 if waves < 9 else 0
        ctx.set_option("lds_pad", pad)
        ms = min(timeit(lambda: ctx.stft_exec(plan, y.data_ptr(), batch, n, n, D.data_ptr()), 5) for _ in range(2))
        print(f"{os.environ.get('LIBROSA_AMD_LIBRARY', 'product')}: {waves} waves/CU (lds_pad {pad // 1024:3d} KB): stft {ms:.3f} ms  -> {ms * 1e-3 * 2.2e9 * 256 * waves / (batch * T):7.0f} cycles per frame per wave @2.2GHz", flush=True)
    ctx.set_option("lds_pad", 0)
elif what == "variant1":
    yh = O.config_input(1, n=44100)
    yt = torch.from_numpy(yh).cuda()
    ref = O.stft(yh, n_fft=2048, hop_length=512)[0]
    for variant in (0, 1, 4):
        ctx.set_option("variant", variant)
        ctx.set_option("stft_iters", 1)
        Dv = L.stft(yt, n_fft=2048, hop_length=512).cpu().numpy()[0]
        err = np.abs(Dv - ref)
        print("variant", variant, "max err", err.max(), "per-frame max (first 12):", np.round(err.max(axis=0)[:12], 3))
        print("   per-bin max (bins 0..15):", np.round(err.max(axis=1)[:16], 3), " bins 1010..1024:", np.round(err.max(axis=1)[1010:], 3))
        bad = np.argwhere(err > 1e-3 * np.abs(ref).max())
        print("   #bad", len(bad), "of", err.size, "first bad (bin, frame):", bad[:10].tolist())
if what == "membw":
    Dr = torch.view_as_real(D)
    src = torch.empty_like(Dr)
    for name, fn in (("fill 2.71 GB", lambda: Dr.fill_(1.0)), ("copy 2.71 GB -> 2.71 GB", lambda: Dr.copy_(src)), ("read-reduce 2.71 GB", lambda: src.sum()),
                     ("mul_ in place 2.71 GB", lambda: Dr.mul_(1.0001)), ("fill y 0.68 GB", lambda: y.fill_(0.5))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            fn()
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 10
        nbytes = Dr.numel() * 4 if "2.71" in name else y.numel() * 4
        mult = 2 if ("copy" in name or "mul_" in name) else 1
        print(f"{name}: {ms:.3f} ms -> {nbytes * mult / ms / 1e6:.0f} GB/s", flush=True)
