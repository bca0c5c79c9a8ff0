"""Development probe (GPU box): per-phase shader-clock ticks of the forward kernel (build with -DLRA_PROBE_ONLY -DLRA_PHASE_TIMER,
run with LIBROSA_AMD_LIBRARY=probe/lib_timer.so).  Not part of the product."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import librosa_amd as L
from librosa_amd import _native, filters

dev = torch.device("cuda", 0)
ctx = L.get_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
lib = _native.load_library()
lib.lra_debug_phase_ticks.restype = ctypes.c_int
lib.lra_debug_phase_ticks.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
n, batch = 22050 * 30, 256
y = bench.make_batch(torch, batch, n, 0, dev)
window = np.asarray(filters.get_window("hann", 2048, fftbins=True), dtype=np.float32)
plan = ctx.stft_plan(2048, 512, window, True, "constant", np.float32)
mel_plan = ctx.mel_plan(filters.mel(sr=22050, n_fft=2048, n_mels=128))
T = ctx.stft_num_frames(plan, n)
D = torch.empty((batch, T, 1025), dtype=torch.complex64, device=dev)
M = torch.empty((batch, 128, T), dtype=torch.float32, device=dev)
ctx.set_option("autotune", 0)
ctx.set_option("variant", 0)
NAMES = {0: "loop", 1: "ring ld+win+pass0+wr", 2: "pass1 read", 3: "pass1 dft+wr", 4: "pass2 read", 5: "pass2 dft+wr", 8: "split read+ring adv", 9: "split+store / accumulate", 10: "mel combine"}


def ticks():
    buf = (ctypes.c_ulonglong * 16)()
    assert lib.lra_debug_phase_ticks(ctx.handle, buf) == 0
    return np.array(list(buf), dtype=np.float64)


def run(name, fn, frames):
    for _ in range(2):
        fn()
    ticks()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    reps = 5
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_ms(e1) / reps
    t = ticks() / (reps * frames)
    tot = t.sum()
    print(f"{name}: {ms:.3f} ms/launch; ticks per frame per wave: total {tot:.0f}")
    for i, v in enumerate(t):
        if v > 0:
            print(f"    [{i:2d}] {NAMES.get(i, '?'):28s} {v:8.0f}  ({100 * v / tot:4.1f} %)")
    sys.stdout.flush()


for pad_kb in (0, 24, 64):
    ctx.set_option("lds_pad", pad_kb * 1024)
    run(f"stft lds_pad={pad_kb}K", lambda: ctx.stft_exec(plan, y.data_ptr(), batch, n, n, D.data_ptr()), batch * T)
ctx.set_option("lds_pad", 0)
run("mel", lambda: ctx.melspectrogram_exec(plan, mel_plan, y.data_ptr(), batch, n, n, 2.0, M.data_ptr()), batch * T)
