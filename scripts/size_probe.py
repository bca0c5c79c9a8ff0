"""Development probe (GPU box): one STFT shape in a loop (for rocprofv3 counter passes).  python scripts/size_probe.py n_fft hop [steps] [what]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
from librosa_amd import filters
n_fft, hop = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
what = sys.argv[4] if len(sys.argv) > 4 else "stft"
dev = torch.device("cuda", 0)
ctx = L.get_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
def apply_opts():
    for kv in os.environ.get("PROBE_OPTS", "").split(","):
        if "=" in kv:
            k, v = kv.split("="); ctx.set_option(k, int(v))
if what != "istft":
    apply_opts()
n, batch = 22050 * 30, 256
y = bench.make_batch(torch, batch, n, 0, dev)
w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)
pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
T = ctx.stft_num_frames(pl, n)
bins = n_fft // 2 + 1
D = torch.empty((batch, T, bins), dtype=torch.complex64, device=dev)
if what == "istft":
    ip = ctx.istft_plan(n_fft, hop, w, True, np.float32)
    ww = filters.window_sumsquare(window="hann", n_frames=T, n_fft=n_fft, hop_length=hop, dtype=np.float32)[n_fft // 2:]
    from librosa_amd.core.spectrum import wss_to_norm
    if not hasattr(ctx, "istft_exec_norm"):  # (older probe builds)
        ctx.istft_exec_norm, wss_to_norm = ctx.istft_exec, (lambda w: w)
    ww = torch.from_numpy(wss_to_norm(np.ascontiguousarray(np.pad(ww, (0, max(0, n - len(ww))))[:n], dtype=np.float32))).to(dev)
    yr = torch.empty((batch, n), dtype=torch.float32, device=dev)
    ctx.stft_exec(pl, y.data_ptr(), batch, n, n, D.data_ptr())
    apply_opts()  # (inverse-only kernel variants: after the forward transform that makes the input)
    fn = lambda: ctx.istft_exec_norm(ip, D.data_ptr(), batch, T * bins, bins, T, ww.data_ptr(), yr.data_ptr(), n, n)
elif what == "mel":
    n_mels = int(os.environ.get("PROBE_MELS", "128"))
    mp = ctx.mel_plan(filters.mel(sr=22050, n_fft=n_fft, n_mels=n_mels))
    Mo = torch.empty((batch, n_mels, T), dtype=torch.float32, device=dev)
    fn = lambda: ctx.melspectrogram_exec(pl, mp, y.data_ptr(), batch, n, n, 2.0, Mo.data_ptr())
elif what == "power":
    S = torch.empty((batch, T, bins), dtype=torch.float32, device=dev)
    fn = lambda: ctx.spectrogram_exec(pl, y.data_ptr(), batch, n, n, 2.0, S.data_ptr())
else:
    fn = lambda: ctx.stft_exec(pl, y.data_ptr(), batch, n, n, D.data_ptr())
if os.environ.get("PROBE_CHECK"):  # the kernel under test against torch.stft in float64 on two clips (probe builds of sizes the parity cases skip)
    got = L.stft(y[:2], n_fft=n_fft, hop_length=hop)
    want = torch.stft(y[:2].double(), n_fft, hop_length=hop, window=torch.from_numpy(w.astype(np.float64)).to(dev), center=True, pad_mode="constant", return_complex=True)
    err = float((got.to(torch.complex128) - want).abs().max() / want.abs().max())
    print(f"check n_fft {n_fft} hop {hop}: max |diff| / max |want| = {err:.3g} {'ok' if err < 2e-6 else 'MISMATCH'}", flush=True)
import time
t_end = time.time() + float(os.environ.get("PROBE_PREWARM_S", "0.4"))  # an idle MI355X runs its first ~0.2 s of work at reduced clocks
while time.time() < t_end:
    for _ in range(20): fn()
    torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = ctx.event(), ctx.event(); e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_ms(e1) / steps)
by = batch * T * (bins * 8 + hop * 4)
print(f"{what} n_fft {n_fft} hop {hop}: frames {batch*T} {best:.3f} ms  {by/best/1e6:.0f} GB/s  ({by/1e6:.0f} MB algorithmic)", flush=True)
