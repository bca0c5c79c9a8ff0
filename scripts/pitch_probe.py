"""Development probe (GPU box), round 5 / VERDICT r04 item 1: does a cache-line-aligned ROW PITCH move the two HBM-bound transforms?

  python scripts/pitch_probe.py [stream] [kernels] [sizes]

stream   lra_probe_stream_pitched (no arithmetic): forward / inverse x row pitch {8200, 8256, 8320, 8448} B x piece {8, 16} B x strip x waves per CU
kernels  the shipped stft2_kernel<OUT_COMPLEX> / istft_kernel at n_fft 2048, hop 512 on rows `pitch` complex64 apart (lra_stft_exec_strided,
         d_frame_stride of lra_istft_exec_norm), checked against the packed result
sizes    the n_fft 512 and 8192 legs of BASELINE configs[4] at packed / padded pitch
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, librosa_amd as L
from librosa_amd import filters
from librosa_amd.core.spectrum import wss_to_norm

what = sys.argv[1:] or ["stream", "kernels", "sizes"]
dev = torch.device("cuda", 0)
ctx = L.get_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
n, batch = 22050 * 30, 256
y = bench.make_batch(torch, batch, n, 0, dev)


def timeit(fn, steps=20, prewarm=0.3):
    t_end = time.time() + prewarm
    while time.time() < t_end:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = ctx.event(), ctx.event(); e0.record()
        for _ in range(steps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_ms(e1) / steps)
    return best


def aligned_buffer(nbytes, align=4096):
    raw = torch.empty(nbytes + align, dtype=torch.uint8, device=dev)
    off = (-raw.data_ptr()) % align
    return raw, raw.data_ptr() + off


if "stream" in what:
    n_fft, hop = 2048, 512
    T = 1 + n // hop
    by = batch * T * ((n_fft // 2 + 1) * 8 + hop * 4)
    yr = torch.empty((batch, n), dtype=torch.float32, device=dev)
    for pitch in (8200, 8256, 8320, 8448):
        keep, dptr = aligned_buffer(batch * T * pitch)
        torch.as_strided(keep, (keep.numel(),), (1,)).zero_()
        for direction, src, dst in ((0, y.data_ptr(), dptr), (1, dptr, yr.data_ptr())):
            for piece in (8, 16):
                if piece == 16 and pitch % 16:
                    continue
                for wpc in (12, 16, 24):
                    row = []
                    for strip in (81, 162, 323):
                        ms = timeit(lambda: ctx.probe_stream(direction, src, dst, batch, T, n_fft, hop, n, strip, wpc, pitch, piece), prewarm=0.15)
                        row.append(f"{strip}: {ms:.3f} ms {by / ms / 1e6:5.0f} GB/s")
                    print(f"stream dir {direction} pitch {pitch} piece {piece:2d} waves/CU {wpc:2d} | " + " | ".join(row), flush=True)
        del keep

if "kernels" in what or "sizes" in what:
    cases = []
    if "kernels" in what:
        cases += [(2048, 512, p) for p in (1025, 1032, 1040, 1056)]
    if "sizes" in what:
        cases += [(512, 512, 257), (512, 512, 272), (512, 128, 257), (512, 128, 272), (8192, 512, 4097), (8192, 512, 4112), (1024, 256, 513), (1024, 256, 528), (4096, 1024, 2049), (4096, 1024, 2064)]
    ref = {}
    for n_fft, hop, pitch in cases:
        bins = n_fft // 2 + 1
        w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)
        pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
        T = ctx.stft_num_frames(pl, n)
        by = batch * T * (bins * 8 + hop * 4)
        keep, dptr = aligned_buffer(batch * T * pitch * 8)
        off = (dptr - keep.data_ptr())
        Dv = keep[off:off + batch * T * pitch * 8].view(torch.complex64).view(batch, T, pitch)
        Dv.zero_()
        fwd = lambda: ctx.stft_exec_strided(pl, 0, y.data_ptr(), batch, n, n, 1.0, dptr, pitch)
        ms_f = timeit(fwd)
        got = Dv[:2, :, :bins].clone()
        if pitch == bins:
            ref[(n_fft, hop)] = got
            ok = "packed"
        else:
            ok = "== packed" if torch.equal(got, ref[(n_fft, hop)]) else "MISMATCH vs packed"
            pad_clean = bool((Dv[:, :, bins:] == 0).all())
            ok += ", padding untouched" if pad_clean else ", PADDING WRITTEN"
        line = f"kernel n_fft {n_fft} hop {hop} pitch {pitch} ({pitch * 8} B): stft {ms_f:.3f} ms {by / ms_f / 1e6:5.0f} GB/s ({100 * by / ms_f / 8e9:.1f} %) [{ok}]"
        # inverse on the same rows
        ip = ctx.istft_plan(n_fft, hop, w, True, np.float32)
        ww = filters.window_sumsquare(window="hann", n_frames=T, n_fft=n_fft, hop_length=hop, dtype=np.float32)[n_fft // 2:]
        ww = torch.from_numpy(wss_to_norm(np.ascontiguousarray(np.pad(ww, (0, max(0, n - len(ww))))[:n], dtype=np.float32))).to(dev)
        yr = torch.empty((batch, n), dtype=torch.float32, device=dev)
        inv = lambda: ctx.istft_exec_norm(ip, dptr, batch, T * pitch, pitch, T, ww.data_ptr(), yr.data_ptr(), n, n)
        ms_i = timeit(inv)
        snr = float(10 * torch.log10((y[:4].double() ** 2).sum() / ((y[:4].double() - yr[:4].double()) ** 2).sum()))
        line += f" | istft {ms_i:.3f} ms {by / ms_i / 1e6:5.0f} GB/s ({100 * by / ms_i / 8e9:.1f} %) round-trip SNR {snr:.1f} dB"
        print(line, flush=True)
        del keep, Dv
