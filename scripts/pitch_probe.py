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


def _opt(key, value):
    """ctx.set_option for keys that only some experiment builds know (wide_store: the round-5 lane-pair-exchange epilogue, not in the product)."""
    try:
        ctx.set_option(key, value)
    except Exception:
        pass


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

if "kernels" in what or "sizes" in what or "wide" in what:
    cases = []
    if "kernels" in what:
        cases += [(2048, 512, p) for p in (1025, 1032, 1040, 1056)]
    if "sizes" in what:
        cases += [(512, 512, 257), (512, 512, 272), (512, 128, 257), (512, 128, 272), (8192, 512, 4097), (8192, 512, 4112), (1024, 256, 513), (1024, 256, 528), (4096, 1024, 2049), (4096, 1024, 2064)]
    ref = {}
    if "wide" in what:  # the 16-byte-piece epilogue (ctx option wide_store) against the 8-byte one, same rows
        cases = [(n_fft, hop, pitch, w) for n_fft, hop, pitch in ((2048, 512, 1040), (2048, 512, 1056), (2048, 256, 1040), (1024, 256, 528), (4096, 1024, 2064), (2048, 512, 1026)) for w in (0, 1, 0, 1)]
        cases = [(2048, 512, 1025, 0)] + cases
    else:
        cases = [c + (None,) for c in cases]
    for n_fft, hop, pitch, wide in cases:
        if wide is not None:
            _opt("wide_store", wide)
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
            if (n_fft, hop) not in ref:
                tmp = torch.empty((2, T, bins), dtype=torch.complex64, device=dev)
                _opt("wide_store", 0)
                ctx.stft_exec(pl, y.data_ptr(), 2, n, n, tmp.data_ptr())
                if wide is not None:
                    _opt("wide_store", wide)
                ref[(n_fft, hop)] = tmp
            ok = "== packed" if torch.equal(got, ref[(n_fft, hop)]) else "MISMATCH vs packed"
            pad_clean = bool((Dv[:, :, bins:] == 0).all())
            ok += ", padding untouched" if pad_clean else ", PADDING WRITTEN"
        line = f"kernel n_fft {n_fft} hop {hop} pitch {pitch} ({pitch * 8} B) wide {wide}: stft {ms_f:.3f} ms {by / ms_f / 1e6:5.0f} GB/s ({100 * by / ms_f / 8e9:.1f} %) [{ok}]"
        if wide is not None:
            print(line, flush=True)
            del keep, Dv
            continue
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

if "grid" in what:  # pitch x frames-per-strip for the shipped kernel (8-byte pieces): which layouts does this box like?
    n_fft, hop, bins = 2048, 512, 1025
    w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)
    pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
    T = ctx.stft_num_frames(pl, n)
    by = batch * T * (bins * 8 + hop * 4)
    _opt("wide_store", 0)
    keep, dptr = aligned_buffer(batch * T * 1064 * 8)
    print("grid: ms per launch, rows = pitch (complex64 elements), columns = frames per strip (0 = the library's choice)", flush=True)
    for pitch in list(range(1025, 1045)) + [1048, 1056, 1064]:
        row = []
        for iters in (0, 54, 81, 108, 162):
            ctx.set_option("stft_iters", iters)
            ms = timeit(lambda: ctx.stft_exec_strided(pl, 0, y.data_ptr(), batch, n, n, 1.0, dptr, pitch), steps=10, prewarm=0.08)
            row.append(f"{ms:.3f}")
        print(f"grid pitch {pitch:4d} ({pitch * 8} B, 2^{(pitch * 8 & -(pitch * 8)).bit_length() - 1} x odd): " + "  ".join(row), flush=True)
    ctx.set_option("stft_iters", 0)
    # the same buffer at other base offsets (pitch 1025 and 1040)
    for pitch in (1025, 1040):
        row = []
        for off in (0, 8, 64, 128, 256, 1024, 4096 + 64):
            ms = timeit(lambda: ctx.stft_exec_strided(pl, 0, y.data_ptr(), batch, n, n, 1.0, dptr + off, pitch), steps=10, prewarm=0.08)
            row.append(f"+{off}: {ms:.3f}")
        print(f"grid base offset, pitch {pitch}: " + "  ".join(row), flush=True)

if "widebug" in what:
    n_fft, hop, bins, pitch = 2048, 512, 1025, 1040
    w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)
    pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
    T = ctx.stft_num_frames(pl, n)
    res = []
    for wide in (0, 1):
        _opt("wide_store", wide)
        D = torch.zeros((2, T, pitch), dtype=torch.complex64, device=dev)
        ctx.stft_exec_strided(pl, 0, y.data_ptr(), 2, n, n, 1.0, D.data_ptr(), pitch)
        torch.cuda.synchronize()
        res.append(D[0, 5, :bins].cpu().numpy())
    bad = np.nonzero(res[0] != res[1])[0]
    print("widebug: mismatching bins of frame 5:", bad.tolist()[:200], flush=True)
    for b in bad[:12]:
        hits = np.nonzero(res[0] == res[1][b])[0]
        print(f"  bin {b}: wide value equals the narrow value of bins {hits.tolist()[:4]}; zero: {res[1][b] == 0}", flush=True)

if "lottery" in what:  # the same kernel, pitch and process on DIFFERENT allocations: is the rate a property of where the buffer landed?
    n_fft, hop, bins = 2048, 512, 1025
    w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)
    pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
    T = ctx.stft_num_frames(pl, n)
    _opt("wide_store", 0)
    held = []
    for trial in range(8):
        keep, dptr = aligned_buffer(batch * T * 1040 * 8 + trial * (3 << 20), align=1 << 21)
        held.append(keep)
        if trial % 2 == 1:
            keep.zero_()
        row = [f"{timeit(lambda: ctx.stft_exec_strided(pl, 0, y.data_ptr(), batch, n, n, 1.0, dptr + off, pitch), steps=15, prewarm=0.2):.3f}" for pitch, off in ((1025, 0), (1040, 0), (1025, 1 << 20), (1040, 1 << 20))]
        print(f"lottery allocation {trial} at {dptr:#x}{' (zeroed first)' if trial % 2 else ''}: pitch 1025 / 1040 / 1025 + 1 MiB / 1040 + 1 MiB: " + "  ".join(row), flush=True)
    # and the input side: a fresh copy of the PCM batch
    y2 = y.clone()
    dptr = held[0].data_ptr() + ((-held[0].data_ptr()) % (1 << 21))
    row = [f"{timeit(lambda: ctx.stft_exec_strided(pl, 0, yy.data_ptr(), batch, n, n, 1.0, dptr, pitch), steps=15, prewarm=0.2):.3f}" for yy in (y, y2) for pitch in (1025, 1040)]
    print("lottery input copies (y / y.clone()) x pitch 1025 / 1040 on allocation 0: " + "  ".join(row), flush=True)

if "lottery2" in what:  # what makes an allocation fast?  size class (the VRAM manager hands out power-of-two blocks) and position inside one arena
    n_fft, hop, bins = 2048, 512, 1025
    w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)
    pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
    T = ctx.stft_num_frames(pl, n)
    _opt("wide_store", 0)
    need = batch * T * 1025 * 8
    held = []
    def run(ptr):
        return timeit(lambda: ctx.stft_exec_strided(pl, 0, y.data_ptr(), batch, n, n, 1.0, ptr, 1025), steps=15, prewarm=0.2)
    for label, size in (("exact", need), ("4 GiB", 4 << 30), ("exact", need), ("4 GiB", 4 << 30), ("8 GiB", 8 << 30), ("3 GiB", 3 << 30), ("exact", need), ("4 GiB", 4 << 30), ("2 GiB + 1 GiB halves", 0)):
        if size == 0:
            continue
        t = torch.empty(size, dtype=torch.uint8, device=dev)
        held.append(t)
        row = [f"+{o >> 20} MiB: {run(t.data_ptr() + o):.3f}" for o in (0, 1 << 20, (size - need) & ~0xfff) if o + need <= size]
        print(f"lottery2 {label:6s} allocation at {t.data_ptr():#x}: " + "  ".join(row), flush=True)
    big = torch.empty(24 << 30, dtype=torch.uint8, device=dev)
    row = [f"+{o >> 30} GiB: {run(big.data_ptr() + o):.3f}" for o in range(0, (24 << 30) - need, 3 << 30)]
    print(f"lottery2 one 24 GiB arena at {big.data_ptr():#x}: " + "  ".join(row), flush=True)
    # the library's own allocator (hipMalloc through the C ABI) instead of torch's
    for trial in range(4):
        buf = ctx.alloc(need)
        held.append(buf)
        print(f"lottery2 hipMalloc (C ABI) allocation {trial} at {buf.ptr:#x}: {run(buf.ptr):.3f}", flush=True)

if "variant" in what:  # one library build per process (LIBROSA_AMD_LIBRARY): the kernel next to the arithmetic-free stream ON THE SAME BUFFER, so that builds compare across processes
    n_fft, hop, bins = 2048, 512, 1025
    w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)
    pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
    T = ctx.stft_num_frames(pl, n)
    keep, dptr = aligned_buffer(batch * T * 1040 * 8, align=1 << 21)
    ref = None
    for pitch in (1025, 1040):
        row = []
        for piece in (8, 16):
            if piece == 16 and pitch % 2:
                continue
            ms = timeit(lambda: ctx.probe_stream(0, y.data_ptr(), dptr, batch, T, n_fft, hop, n, 162, 12, pitch * 8, piece), steps=15, prewarm=0.2)
            row.append(f"stream piece {piece}: {ms:.3f}")
        for wide in (0, 1):
            _opt("wide_store", wide)
            ms = timeit(lambda: ctx.stft_exec_strided(pl, 0, y.data_ptr(), batch, n, n, 1.0, dptr, pitch), steps=15, prewarm=0.2)
            got = torch.as_strided(keep, (1,), (1,))  # (keep alive)
            Dv = torch.empty(0)
            off = dptr - keep.data_ptr()
            view = keep[off:off + 2 * T * pitch * 8].view(torch.complex64).view(2, T, pitch)[:, :, :bins].clone()
            if ref is None:
                ref = view
            row.append(f"kernel wide {wide}: {ms:.3f} ({'==' if torch.equal(view, ref) else 'MISMATCH'})")
        print(f"variant {os.environ.get('LIBROSA_AMD_LIBRARY', 'product')} pitch {pitch}: " + "  ".join(row), flush=True)

if "window" in what:  # concurrent strips inside one moving window (persistent waves, strips dealt in address order) vs the kernels' long private strips, several allocations
    n_fft, hop, bins = 2048, 512, 1025
    w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)
    pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
    T = ctx.stft_num_frames(pl, n)
    by = batch * T * (bins * 8 + hop * 4)
    held = []
    for trial in range(5):
        keep, dptr = aligned_buffer(batch * T * bins * 8 + trial * (3 << 20), align=1 << 21)
        held.append(keep)
        row = [f"kernel {timeit(lambda: ctx.stft_exec(pl, y.data_ptr(), batch, n, n, dptr), steps=15, prewarm=0.2):.3f}",
               f"strips(162, 12/CU) {timeit(lambda: ctx.probe_stream(0, y.data_ptr(), dptr, batch, T, n_fft, hop, n, 162, 12), steps=15, prewarm=0.1):.3f}"]
        for strip in (2, 4, 8, 16, 32):
            for wpc in (8, 12, 16):
                ms = timeit(lambda: ctx.probe_stream_window(y.data_ptr(), dptr, batch, T, n_fft, hop, n, strip, wpc), steps=15, prewarm=0.1)
                row.append(f"win({strip},{wpc}) {ms:.3f}")
        print(f"window allocation {trial}: " + "  ".join(row), flush=True)

if "iters" in what:  # frames per strip = the distance between concurrently written rows: does the rate depend on it (bank / channel conflicts between the strips)?
    n_fft, hop, bins = 2048, 512, 1025
    w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)
    pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
    T = ctx.stft_num_frames(pl, n)
    held = []
    for trial in range(3):
        keep, dptr = aligned_buffer(batch * T * bins * 8 + trial * (3 << 20), align=1 << 21)
        held.append(keep)
        row = []
        for iters in [0] + list(range(120, 181, 4)) + [185, 216, 259, 323]:
            ctx.set_option("stft_iters", iters)
            ms = timeit(lambda: ctx.stft_exec(pl, y.data_ptr(), batch, n, n, dptr), steps=10, prewarm=0.08)
            row.append(f"{iters}:{ms:.3f}")
        ctx.set_option("stft_iters", 0)
        print(f"iters allocation {trial}: " + " ".join(row), flush=True)
    # ... and the input side: clips a different distance apart (y_stride), same output buffer
    dptr = held[0].data_ptr() + ((-held[0].data_ptr()) % (1 << 21))
    row = []
    for pad in (0, 4, 64, 1024, 2500, 4096):
        yp = torch.zeros((batch, n + pad), dtype=torch.float32, device=dev)
        yp[:, :n] = y
        ms = timeit(lambda: ctx.stft_exec(pl, yp.data_ptr(), batch, n, n + pad, dptr), steps=10, prewarm=0.08)
        row.append(f"+{pad}:{ms:.3f}")
    print("iters input clip stride n + pad: " + " ".join(row), flush=True)

if "power" in what:  # socket power / shader clock (rocm-smi, 2.5 s into a loop) beside the rate of every pitch / piece combination, ONE buffer
    import re, shutil, subprocess, threading
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    n_fft, hop, bins = 2048, 512, 1025
    w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)
    pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
    T = ctx.stft_num_frames(pl, n)
    by = batch * T * (bins * 8 + hop * 4)
    keep, dptr = aligned_buffer(batch * T * 1056 * 8, align=1 << 21)
    yr = torch.empty((batch, n), dtype=torch.float32, device=dev)

    def sample(fn):
        stop = threading.Event()
        def feed():
            while not stop.is_set():
                for _ in range(200): fn()
                torch.cuda.synchronize()
        th = threading.Thread(target=feed, daemon=True); th.start()
        try:
            time.sleep(2.5)
            txt = subprocess.run([smi, "--showpower", "--showclocks"], capture_output=True, text=True, timeout=30).stdout
        finally:
            stop.set(); th.join(timeout=30); torch.cuda.synchronize()
        pw = re.search(r"Power \(W\):\s*([0-9.]+)", txt); ck = re.search(r"sclk clock level:\s*\S+\s*\((\d+)Mhz\)", txt)
        return (float(pw.group(1)) if pw else None, int(ck.group(1)) if ck else None)

    print("power: one buffer; ms per launch, GB/s algorithmic, socket W, sclk MHz", flush=True)
    for pitch in (1025, 1032, 1040, 1056):
        for direction, name, src, dst in ((0, "forward stream", y.data_ptr(), dptr), (1, "inverse stream", dptr, yr.data_ptr())):
            for piece in (8, 16):
                if piece == 16 and pitch % 2: continue
                for wpc in (12, 16):
                    fn = lambda: ctx.probe_stream(direction, src, dst, batch, T, n_fft, hop, n, 162, wpc, pitch * 8, piece)
                    ms = timeit(fn, steps=15, prewarm=0.15)
                    p_w, clk = sample(fn)
                    print(f"power pitch {pitch * 8} B {name} piece {piece:2d} waves/CU {wpc}: {ms:.3f} ms {by / ms / 1e6:5.0f} GB/s {p_w} W {clk} MHz", flush=True)
        fn = lambda: ctx.stft_exec_strided(pl, 0, y.data_ptr(), batch, n, n, 1.0, dptr, pitch)
        ms = timeit(fn, steps=15, prewarm=0.15)
        p_w, clk = sample(fn)
        print(f"power pitch {pitch * 8} B stft kernel: {ms:.3f} ms {by / ms / 1e6:5.0f} GB/s {p_w} W {clk} MHz", flush=True)

if "istft_lottery" in what:  # the inverse kernel over several allocations of its input (per-process distribution: builds compare by their sorted lists)
    n_fft, hop, bins = 2048, 512, 1025
    w = np.asarray(filters.get_window("hann", n_fft, fftbins=True), dtype=np.float32)
    pl = ctx.stft_plan(n_fft, hop, w, True, "constant", np.float32)
    ip = ctx.istft_plan(n_fft, hop, w, True, np.float32)
    T = ctx.stft_num_frames(pl, n)
    ww = filters.window_sumsquare(window="hann", n_frames=T, n_fft=n_fft, hop_length=hop, dtype=np.float32)[n_fft // 2:]
    ww = torch.from_numpy(wss_to_norm(np.ascontiguousarray(np.pad(ww, (0, max(0, n - len(ww))))[:n], dtype=np.float32))).to(dev)
    yr = torch.empty((batch, n), dtype=torch.float32, device=dev)
    held, ms_i, ms_f = [], [], []
    for trial in range(7):
        keep, dptr = aligned_buffer(batch * T * bins * 8 + trial * (3 << 20), align=1 << 21)
        held.append(keep)
        ms_f.append(timeit(lambda: ctx.stft_exec(pl, y.data_ptr(), batch, n, n, dptr), steps=10, prewarm=0.15))
        ms_i.append(timeit(lambda: ctx.istft_exec_norm(ip, dptr, batch, T * bins, bins, T, ww.data_ptr(), yr.data_ptr(), n, n), steps=10, prewarm=0.15))
    print(f"istft_lottery {os.environ.get('LIBROSA_AMD_LIBRARY', 'product')}: istft sorted " + " ".join(f"{m:.3f}" for m in sorted(ms_i)) + " | stft sorted " + " ".join(f"{m:.3f}" for m in sorted(ms_f)), flush=True)
