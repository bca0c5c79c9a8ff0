"""ctypes wrapper around tests/hostsim/_hostsim.so (CPU thread simulator of the kernel bodies).

TEST INFRASTRUCTURE ONLY -- see tests/hostsim/hostsim.cpp.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostsim", "hostsim.cpp")
# LRA_HOSTSIM_DEFINES="-DLRA_MEL_ABLATE=1 ...": simulate a kernel experiment (compile-time flag) in its own library
EXTRA_DEFINES = os.environ.get("LRA_HOSTSIM_DEFINES", "").split()
_TAG = "".join(c if c.isalnum() else "_" for c in "".join(EXTRA_DEFINES))
SO = os.path.join(HERE, "hostsim", f"_hostsim{_TAG}.so")
CSRC = os.path.join(os.path.dirname(HERE), "librosa_amd", "csrc")

PAD_MODES = {"constant": 0, "reflect": 1, "edge": 2, "symmetric": 3}


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build():
    if _stale():
        # four objects in parallel (one per transform x dtype), then one link
        from concurrent.futures import ThreadPoolExecutor

        objs = [os.path.join(HERE, "hostsim", f"_part{i}{_TAG}.o") for i in (1, 2, 3, 4)]

        def cc(i):
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-w", "-fPIC", f"-DHOSTSIM_PART={i}"] + EXTRA_DEFINES + ["-c", SRC, "-o", objs[i - 1]])

        with ThreadPoolExecutor(max_workers=4) as pool:
            list(pool.map(cc, (1, 2, 3, 4)))
        subprocess.check_call(["g++", "-shared", "-fPIC", "-o", SO] + objs)
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.hostsim_pad_index.restype = ctypes.c_longlong
        _lib.hostsim_pad_index.argtypes = [ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int]
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def mel_band(B):
    """Dense (n_mels, n_bins) basis -> band form (c0, len, off, val)."""
    n_mels = B.shape[0]
    c0 = np.zeros(n_mels, np.int32)
    ln = np.zeros(n_mels, np.int32)
    off = np.zeros(n_mels, np.int32)
    vals = []
    pos = 0
    for m in range(n_mels):
        nz = np.nonzero(B[m])[0]
        if len(nz):
            c0[m], ln[m] = nz[0], nz[-1] - nz[0] + 1
            vals.append(B[m, nz[0] : nz[-1] + 1])
        off[m] = pos
        pos += ln[m]
    val = np.concatenate(vals) if vals else np.zeros(1, B.dtype)
    return c0, ln, off, np.ascontiguousarray(val)


def stft(y, n_fft, hop, win, center=True, pad_mode="constant", mode=0, iters_per_wg=1, power=2.0, mel_basis=None, variant=0, row_pad=0):
    """y: (batch, n) f32/f64.  mode 0 -> complex (batch, T, M+1); 1 -> power; 2 / 3 / 4 -> mel (batch, n_mels, T)
    through the generic banded path / the two-slope path / its run-ordered form.
    row_pad (modes 0 / 1): rows `M + 1 + row_pad` elements apart (StftArgs::row_pitch); the whole padded buffer comes back, NaN where nothing was stored."""
    y = np.ascontiguousarray(y)
    assert y.ndim == 2
    f64 = y.dtype == np.float64
    batch, n = y.shape
    if center:
        n_frames = 1 + (n + 2 * (n_fft // 2) - n_fft) // hop
    else:
        n_frames = 1 + (n - n_fft) // hop
    M = n_fft // 2
    win = np.ascontiguousarray(0.5 * np.asarray(win, dtype=np.float64), dtype=y.dtype)  # the kernels take 0.5 * window
    pm = 2 if power == 2.0 else (1 if power == 1.0 else 3)
    c0 = ln = off = val = dense = None
    n_mels = 0
    if mode == 0:
        out = np.full((batch, n_frames, M + 1 + row_pad), np.nan, dtype=np.complex128 if f64 else np.complex64)
    elif mode == 1:
        out = np.full((batch, n_frames, M + 1 + row_pad), np.nan, dtype=y.dtype)
    else:
        assert row_pad == 0
        dense = np.ascontiguousarray(mel_basis, dtype=y.dtype)
        c0, ln, off, val = mel_band(dense)
        n_mels = mel_basis.shape[0]
        out = np.full((batch, n_mels, n_frames), np.nan, dtype=y.dtype)
    diag = np.zeros(12, np.int64)
    fn = lib().hostsim_stft_f64 if f64 else lib().hostsim_stft_f32
    if row_pad:
        os.environ["LRA_SIM_ROW_PAD"] = str(int(row_pad))
    else:
        os.environ.pop("LRA_SIM_ROW_PAD", None)
    rc = fn(ctypes.c_int(n_fft), ctypes.c_int(mode), _p(y), ctypes.c_longlong(n), ctypes.c_longlong(batch), ctypes.c_int(n_frames),
            ctypes.c_int(hop), ctypes.c_int(int(center)), ctypes.c_int(PAD_MODES[pad_mode]), _p(win), ctypes.c_int(iters_per_wg), _p(out),
            ctypes.c_int(pm), ctypes.c_double(power), _p(c0), _p(ln), _p(off), _p(val), ctypes.c_int(n_mels), ctypes.c_int(variant), _p(dense), _p(diag))
    assert rc == 0, "unsupported n_fft for the pow2 kernels"
    if mode == 4 and diag[7] != 0:
        return None, dict(unavailable=int(diag[7]))  # run-ordered form not applicable to this configuration (the library falls back too)
    assert diag[7] == 0, "two-slope mel form not applicable"
    return out, dict(races=int(diag[0]), uninit=int(diag[1]), NT=int(diag[2]), FPB=int(diag[3]), P=int(diag[4]), lds=int(diag[5]), wave_sync=int(diag[6]), ring_aligned=int(diag[8]), max_pieces=int(diag[9]), v2=int(diag[10]), mel_many=int(diag[11]))


def istft(D, n_fft, hop, win, wss, out_len, n_used, center=True, strip_groups=4, variant=0):
    """D: (batch, T, M+1) complex, native layout.  win: padded window (n_fft).  wss: (out_len,)."""
    D = np.ascontiguousarray(D)
    f64 = D.dtype == np.complex128
    rt = np.float64 if f64 else np.float32
    batch, T, bins = D.shape
    assert bins == n_fft // 2 + 1
    ws = np.ascontiguousarray(np.asarray(win, dtype=np.float64) / n_fft, dtype=rt)
    # the kernel bodies multiply by the normalisation factors the host wrapper makes of the envelope (lra_api.hip, wss_to_norm_kernel)
    wss = np.asarray(wss, dtype=rt)
    with np.errstate(divide="ignore", over="ignore"):
        wss = np.ascontiguousarray(np.where(wss > np.finfo(rt).tiny, rt(1) / wss, rt(1)), dtype=rt)
    # the buffer arrives full of NaN and only what the host wrapper clears (lra_api.hip, istft_run: samples from istft_written_end() on)
    # is zeroed: every other sample must be stored by the kernel body itself
    y = np.full((batch, out_len), np.nan, dtype=rt)
    lib().hostsim_istft_written_end.restype = ctypes.c_longlong
    end = lib().hostsim_istft_written_end(ctypes.c_int(n_fft), ctypes.c_int(hop), ctypes.c_longlong(n_used), ctypes.c_int(n_fft // 2 if center else 0))
    y[:, min(int(end), out_len):] = 0
    diag = np.zeros(12, np.int64)
    fn = lib().hostsim_istft_f64 if f64 else lib().hostsim_istft_f32
    rc = fn(ctypes.c_int(n_fft), _p(D), ctypes.c_longlong(batch), ctypes.c_int(T), ctypes.c_int(n_used), ctypes.c_int(hop), ctypes.c_int(int(center)),
            _p(ws), _p(wss), ctypes.c_double(float(np.finfo(rt).tiny)), _p(y), ctypes.c_longlong(out_len), ctypes.c_int(strip_groups), ctypes.c_int(variant), _p(diag))
    assert rc == 0
    return y, dict(races=int(diag[0]), uninit=int(diag[1]), NT=int(diag[2]), FPB=int(diag[3]), P=int(diag[4]), lds=int(diag[5]), wave_sync=int(diag[6]), ring_aligned=int(diag[8]))


# ---- PCEN / band max-filter kernels (librosa_amd/csrc/lra_pcen.h) on host threads: tests/hostsim/postsim.cpp -----------------------
POST_SRC = os.path.join(HERE, "hostsim", "postsim.cpp")
POST_SO = os.path.join(HERE, "hostsim", "_postsim.so")
_post = None


def post_lib():
    global _post
    if _post is None:
        deps = [POST_SRC] + [os.path.join(CSRC, h) for h in ("lra_pcen.h", "lra_cqt.h", "lra_hpss.h", "lra_mixed.h", "lra_rng.h")]
        if not os.path.exists(POST_SO) or any(os.path.getmtime(d) > os.path.getmtime(POST_SO) for d in deps):
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-w", "-fPIC", "-shared", "-pthread", POST_SRC, "-o", POST_SO])
        _post = ctypes.CDLL(POST_SO)
        c = ctypes
        _post.postsim_pcen.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_longlong, c.c_longlong, c.c_int] + [c.c_double] * 5 + [c.c_void_p, c.c_double, c.c_void_p]
        _post.postsim_maxfilter.argtypes = [c.c_void_p, c.c_void_p, c.c_longlong, c.c_int, c.c_longlong, c.c_int, c.c_int]
        _post.postsim_fir_decimate.argtypes = [c.c_void_p, c.c_void_p, c.c_longlong, c.c_longlong, c.c_longlong, c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_double, c.c_double, c.c_int]
        _post.postsim_resample_poly.argtypes = [c.c_void_p, c.c_void_p, c.c_longlong, c.c_longlong, c.c_longlong, c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.c_double, c.c_double, c.c_int]
        _post.postsim_magnitude.argtypes = [c.c_void_p, c.c_void_p, c.c_longlong, c.c_int]
        _post.postsim_magphase.argtypes = [c.c_void_p, c.c_int, c.c_void_p, c.c_void_p, c.c_longlong, c.c_double, c.c_int]
        _post.postsim_hpss.argtypes = [c.c_void_p] * 4 + [c.c_longlong, c.c_longlong, c.c_int, c.c_int, c.c_int, c.c_double, c.c_double, c.c_double, c.c_int, c.c_int]
        _post.postsim_cqt_project.argtypes = [c.c_void_p] * 6 + [c.c_longlong, c.c_longlong, c.c_int, c.c_longlong, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int]
    return _post


def _pcg64_state4(rng):
    st = rng.bit_generator.state
    assert st["bit_generator"] == "PCG64"
    state, inc, m = int(st["state"]["state"]), int(st["state"]["inc"]), (1 << 64) - 1
    return np.array([(state >> 64) & m, state & m, (inc >> 64) & m, inc & m], dtype=np.uint64)


def pcg64_random(rng, offset, count):
    """Draws ``offset .. offset + count`` of ``rng.random()`` (the generator itself is left alone) through the kernel body of lra_pcg64_random_exec."""
    out = np.full(count, np.nan)
    fn = post_lib().postsim_pcg64_random
    fn.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_longlong]
    st = _pcg64_state4(rng)
    fn(_p(st), int(offset), _p(out), int(count))
    return out


def griffinlim_init_pcg64(rng, S_frame_major, seg=256):
    """S: (batch, n_frames, n_bins) real, the device layout -> angles (same layout, complex) = S exp(2 pi i u) with u = rng.random((batch, n_bins, n_frames))."""
    S = np.ascontiguousarray(S_frame_major)
    batch, n_frames, n_bins = S.shape
    out = np.full(S.shape, np.nan, dtype=np.complex128 if S.dtype == np.float64 else np.complex64)
    fn = post_lib().postsim_griffinlim_init_pcg64
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int]
    st = _pcg64_state4(rng)
    fn(_p(st), _p(S), _p(out), batch, n_bins, n_frames, int(seg), int(S.dtype == np.float64))
    return out


def pcen(S, *, b, gain, bias, power, eps, zi_scalar, ref=None, zi=None, want_zf=False):
    """S: (rows, n_frames) float32 / float64 -> (out float64, zf or None), through the kernel body of lra_pcen_exec."""
    S = np.ascontiguousarray(S)
    rows, n = S.shape
    out = np.full((rows, n), np.nan)
    zf = np.full(rows, np.nan) if want_zf else None
    ref = None if ref is None else np.ascontiguousarray(ref, dtype=S.dtype)
    zi = None if zi is None else np.ascontiguousarray(zi, dtype=np.float64)
    post_lib().postsim_pcen(_p(S), _p(ref), _p(out), rows, n, int(S.dtype == np.float64), b, gain, bias, power, eps, _p(zi), zi_scalar, _p(zf))
    return out, zf


def maxfilter(S, size):
    """S: (outer, n_bands, inner) -> max over a window of ``size`` bands (scipy.ndimage.maximum_filter1d, mode="reflect")."""
    S = np.ascontiguousarray(S)
    out = np.empty_like(S)
    post_lib().postsim_maxfilter(_p(S), _p(out), S.shape[0], S.shape[1], S.shape[2], int(size), int(S.dtype == np.float64))
    return out


def fir_decimate(x, taps, down, first, n_out, div=1.0, mul=1.0):
    """x: (batch, n_in) -> (batch, n_out) through the kernel body of lra_fir_decimate_exec."""
    x = np.ascontiguousarray(x)
    taps = np.ascontiguousarray(taps, dtype=x.dtype)
    out = np.full((x.shape[0], n_out), np.nan, dtype=x.dtype)
    post_lib().postsim_fir_decimate(_p(x), _p(out), x.shape[0], x.shape[1], n_out, _p(taps), len(taps), int(down), int(first), float(div), float(mul), int(x.dtype == np.float64))
    return out


def resample_poly(x, taps, up, down, first, n_out, div=1.0, mul=1.0):
    """x: (batch, n_in) -> (batch, n_out) through the kernel body of lra_resample_poly_exec."""
    x = np.ascontiguousarray(x)
    taps = np.ascontiguousarray(taps, dtype=x.dtype)
    out = np.full((x.shape[0], n_out), np.nan, dtype=x.dtype)
    post_lib().postsim_resample_poly(_p(x), _p(out), x.shape[0], x.shape[1], n_out, _p(taps), len(taps), int(up), int(down), int(first), float(div), float(mul), int(x.dtype == np.float64))
    return out


def cqt_project(D, csr, n_frames, n_total, bin0, row0, n_rows, sqrt_len=None):
    """D: (batch, frames_in, n_bins) complex -> the octave's rows of a (batch, n_frames, n_total) result (zeros elsewhere)."""
    D = np.ascontiguousarray(D)
    out = np.zeros((D.shape[0], n_frames, n_total), dtype=D.dtype)
    rp, col = np.ascontiguousarray(csr.indptr, dtype=np.int32), np.ascontiguousarray(csr.indices, dtype=np.int32)
    val = np.ascontiguousarray(csr.data, dtype=D.dtype)
    sl = None if sqrt_len is None else np.ascontiguousarray(sqrt_len, dtype=np.float64)
    post_lib().postsim_cqt_project(_p(D), _p(out), _p(rp), _p(col), _p(val), _p(sl), D.shape[0], D.shape[1], D.shape[2], n_frames, n_total, bin0, row0, n_rows, int(D.dtype == np.complex128))
    return out


def hpss(D, *, win_harm=31, win_perc=31, power=2.0, margin_harm=1.0, margin_perc=1.0, want_mask=False):
    """D: (batch, frames, bins) complex or real, the STFT kernel's layout -> (harmonic, percussive) through the kernel bodies of
    lra_magnitude_exec / lra_hpss_exec."""
    D = np.ascontiguousarray(D)
    cplx = np.iscomplexobj(D)
    real = np.dtype(np.float64) if D.dtype in (np.complex128, np.float64) else np.dtype(np.float32)
    if cplx:
        mag = np.full(D.shape, np.nan, dtype=real)
        post_lib().postsim_magnitude(_p(D), _p(mag), D.size, int(real == np.float64))
    else:
        mag = D
    odt = real if (want_mask or not cplx) else D.dtype
    oh, op = np.full(D.shape, np.nan, dtype=odt), np.full(D.shape, np.nan, dtype=odt)
    post_lib().postsim_hpss(_p(mag), _p(D) if cplx else None, _p(oh), _p(op), D.shape[0], D.shape[1], D.shape[2], int(win_harm), int(win_perc), float(power), float(margin_harm),
                            float(margin_perc), int(want_mask), int(real == np.float64))
    return oh, op


def mixed_stft(y, n_fft, hop, window, *, mode="stft", center=True, pad_mode="constant", power=2.0, mel_basis=None):
    """The fused mixed-radix forward kernel (csrc/lra_mixed.h) through the simulator, argument preparation as in lra_api.hip (stft_run / plan create).
    y: (batch, n); returns (batch, n_frames, bins) complex / real for "stft" / "power", (batch, n_mels, n_frames) for "mel"."""
    y = np.ascontiguousarray(y)
    rt = y.dtype.type
    ct = np.complex128 if y.dtype == np.float64 else np.complex64
    batch, n = y.shape
    M = n_fft // 2
    pad = n_fft // 2 if center else 0
    n_frames = 1 + (n + 2 * pad - n_fft) // hop
    win = np.ascontiguousarray(window, dtype=rt)
    t = np.arange(M, dtype=np.float64)
    tw_m = np.ascontiguousarray(np.exp(-2j * np.pi * t / M).astype(ct))
    tw_n = np.ascontiguousarray(np.exp(-2j * np.pi * np.arange(M + 1, dtype=np.float64) / n_fft).astype(ct))
    code = {"stft": 0, "power": 1, "mel": 2}[mode]
    c0 = ln = off = val = None
    n_mels = 0
    if mode == "mel":
        c0, ln, off, val = mel_band(np.asarray(mel_basis, dtype=rt))
        n_mels = mel_basis.shape[0]
        out = np.full((batch, n_mels, n_frames), np.nan, dtype=rt)
    elif mode == "power":
        out = np.full((batch, n_frames, M + 1), np.nan, dtype=rt)
    else:
        out = np.full((batch, n_frames, M + 1), np.nan, dtype=ct)
    pm = {"constant": 0, "reflect": 1, "edge": 2, "symmetric": 3}[pad_mode]
    fn = post_lib().postsim_mixed_stft
    c = ctypes
    fn.argtypes = [c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_longlong, c.c_longlong, c.c_int, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_double,
                   c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int]
    rc = fn(n_fft, code, int(y.dtype == np.float64), _p(y), batch, n, n_frames, hop, pad, pm, _p(win), _p(tw_m), _p(tw_n), _p(out), 2 if power == 2.0 else (1 if power == 1.0 else 0), float(power),
            _p(c0), _p(ln), _p(off), _p(val), n_mels)
    assert rc == 0, f"n_fft={n_fft} is not among the simulator's mixed-radix sizes"
    return out


def mixed_irfft(D, n_fft):
    """mixed_irfft_kernel (csrc/lra_mixed.h) through the simulator: D (batch, T, M + 1) complex, frame-major -> (batch, T, n_fft) real = n_fft * irfft (rocFFT's
    unnormalised C2R); the output arrives full of NaN."""
    D = np.ascontiguousarray(D)
    f64 = D.dtype == np.complex128
    rt = np.float64 if f64 else np.float32
    batch, T, bins = D.shape
    M = n_fft // 2
    assert bins == M + 1
    tw_m = np.ascontiguousarray(np.exp(-2j * np.pi * np.arange(M, dtype=np.float64) / M).astype(D.dtype))
    tw_n = np.ascontiguousarray(np.exp(-2j * np.pi * np.arange(M + 1, dtype=np.float64) / n_fft).astype(D.dtype))
    out = np.full((batch, T, n_fft), np.nan, dtype=rt)
    fn = post_lib().postsim_mixed_irfft
    c = ctypes
    fn.argtypes = [c.c_int, c.c_int, c.c_void_p, c.c_longlong, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p]
    rc = fn(n_fft, int(f64), _p(D), batch, T, _p(tw_m), _p(tw_n), _p(out))
    assert rc == 0, f"simulator: n_fft={n_fft} not served (rc {rc})"
    return out


def mixed_istft(D, n_fft, hop, win, wss, out_len, n_used, center=True):
    """The fused mixed-radix inverse kernel (csrc/lra_mixed.h) through the simulator; D: (batch, T, M + 1) complex, frame-major.  The output arrives full of NaN and
    only what lra_api.hip's wrapper clears (samples past the last frame's end) is zeroed: every other sample must be stored by the kernel."""
    D = np.ascontiguousarray(D)
    f64 = D.dtype == np.complex128
    rt = np.float64 if f64 else np.float32
    ct = D.dtype
    batch, T, bins = D.shape
    M = n_fft // 2
    assert bins == M + 1
    ws = np.ascontiguousarray(np.asarray(win, dtype=np.float64) / n_fft, dtype=rt)
    wss = np.asarray(wss, dtype=rt)
    with np.errstate(divide="ignore", over="ignore"):
        norm = np.ascontiguousarray(np.where(wss > np.finfo(rt).tiny, rt(1) / wss, rt(1)), dtype=rt)
    tw_m = np.ascontiguousarray(np.exp(-2j * np.pi * np.arange(M, dtype=np.float64) / M).astype(ct))
    tw_n = np.ascontiguousarray(np.exp(-2j * np.pi * np.arange(M + 1, dtype=np.float64) / n_fft).astype(ct))
    drop = n_fft // 2 if center else 0
    y = np.full((batch, out_len), np.nan, dtype=rt)
    y[:, max(0, min(out_len, (n_used - 1) * hop + n_fft - drop)):] = 0
    fn = post_lib().postsim_mixed_istft
    c = ctypes
    fn.argtypes = [c.c_int, c.c_int, c.c_void_p, c.c_longlong, c.c_int, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_longlong]
    rc = fn(n_fft, int(f64), _p(D), batch, T, n_used, hop, drop, _p(ws), _p(tw_m), _p(tw_n), _p(norm), _p(y), out_len)
    assert rc == 0, f"simulator: n_fft={n_fft} hop={hop} not served (rc {rc})"
    return y
