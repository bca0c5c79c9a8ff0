"""CPU checks of the gfx950 kernel BODIES (librosa_amd/csrc/lra_kernels.h) through the host thread
simulator: same templated code the GPU runs, executed phase by phase over a workgroup's threads,
with an LDS shadow that flags cross-thread races inside a phase and reads of unwritten LDS.

This validates index math, barrier placement and numerics against the oracle in the build
container (no GPU).  It is test infrastructure; the product library never runs on the CPU.
"""
import warnings

import numpy as np
import pytest

import hostsim_util as H
import stft_oracle as O

warnings.filterwarnings("ignore", message="n_fft=.*is too large")


def _check_diag(d):
    assert d["races"] == 0, f"LDS race inside a phase: {d}"
    assert d["uninit"] == 0, f"read of never-written LDS: {d}"


def _tol(dtype):
    return 2e-6 if dtype == np.float32 else 1e-13


@pytest.mark.parametrize(
    "n_fft,hop,center,pad_mode,iters,dtype,n,variant",
    [
        (32, 8, True, "constant", 1, np.float32, 500, 0),
        (64, 16, True, "reflect", 1, np.float32, 20, 0),  # pad longer than the signal: repeated reflection
        (256, 64, True, "symmetric", 2, np.float32, 879, 0),
        (512, 128, True, "reflect", 1, np.float32, 300, 0),
        (512, 100, True, "edge", 3, np.float32, 1647, 0),
        (1024, 256, False, "constant", 2, np.float32, 3183, 0),
        (2048, 512, True, "constant", 2, np.float32, 22050, 0),
        (2048, 512, True, "constant", 3, np.float32, 9000, 1),
        (2048, 512, True, "reflect", 1, np.float32, 9000, 2),
        (2048, 512, True, "constant", 1, np.float32, 9000, 3),
        (2048, 512, True, "reflect", 5, np.float32, 9000, 4),
        (2048, 511, True, "constant", 5, np.float32, 9001, 0),  # odd hop: unaligned sample pairs in the ring
        (2048, 512, True, "reflect", 7, np.float32, 30000, 0),  # long slot runs: ring + prefetch, edges at both ends
        (2048, 1024, True, "reflect", 5, np.float32, 30000, 0),  # hop = n_fft/2: the widest prefetch
        (1024, 512, True, "constant", 4, np.float32, 12000, 0),
        (1024, 700, True, "constant", 4, np.float32, 12000, 0),  # hop > n_fft/2: no prefetch, direct fetches
        (1024, 256, True, "edge", 6, np.float32, 9000, 0),
        (1024, 512, True, "constant", 4, np.float32, 9000, 0),  # hop = n_fft/2: new block does not fit the prefetch registers
        (512, 512, True, "reflect", 3, np.float32, 9000, 0),  # hop = n_fft
        (256, 300, True, "constant", 3, np.float32, 5000, 0),  # hop > n_fft
        (512, 512, False, "constant", 5, np.float32, 9001, 0),  # direct framing (no ring): odd length -> the second clip's pairs are not 8-byte aligned
        (2048, 2048, True, "edge", 2, np.float32, 30001, 0),   # direct framing, one wave per frame
        (8192, 8192, True, "reflect", 2, np.float32, 70000, 0),  # direct framing, four waves per frame
        (128, 1000, True, "symmetric", 4, np.float32, 9000, 0),
        (8192, 512, True, "reflect", 3, np.float32, 30001, 0),   # register ring, HD = 16 (one new pair per thread and frame), four waves per frame
        (8192, 2048, False, "constant", 2, np.float32, 40000, 0),  # HD = 4
        (16384, 8192, True, "edge", 2, np.float32, 60001, 0),    # HD = 2, eight waves per frame
        (512, 512, True, "reflect", 3, np.float64, 9000, 0),
        (512, 100, False, "constant", 4, np.float32, 3000, 0),
        (4096, 1000, True, "symmetric", 3, np.float32, 30000, 0),  # two waves per frame (workgroup barriers), odd-ish hop
        (2048, 512, True, "constant", 4, np.float32, 700, 0),  # fewer frames than slots x iters
        (4096, 1024, True, "constant", 1, np.float32, 20000, 0),
        (8192, 512, True, "constant", 1, np.float32, 24687, 0),
        (16384, 4096, True, "constant", 1, np.float32, 49263, 0),
        (64, 16, True, "constant", 1, np.float64, 700, 0),
        (512, 128, True, "reflect", 1, np.float64, 1647, 0),
        (2048, 512, True, "constant", 2, np.float64, 9000, 0),
        (8192, 2048, False, "constant", 1, np.float64, 30000, 0),
    ],
)
def test_stft_body(n_fft, hop, center, pad_mode, iters, dtype, n, variant, monkeypatch):
    monkeypatch.setenv("LRA_SIM_NO_V2", "1")  # the first-generation bodies (the second generation has its own cases below)
    rng = np.random.default_rng(n_fft + hop + n)
    y = rng.standard_normal((2, n)).astype(dtype)
    win = O.get_window("hann", n_fft)
    out, d = H.stft(y, n_fft, hop, win, center=center, pad_mode=pad_mode, iters_per_wg=iters, variant=variant)
    _check_diag(d)
    ref = np.moveaxis(O.stft(y, n_fft=n_fft, hop_length=hop, center=center, pad_mode=pad_mode), -1, -2)
    assert out.shape == ref.shape and out.dtype == ref.dtype
    assert np.isfinite(out.view(dtype)).all()
    assert np.abs(out - ref).max() <= _tol(dtype) * np.abs(ref).max()


@pytest.mark.parametrize(
    "n_fft,hop,center,pad_mode,iters,n,power",
    [
        (2048, 512, True, "constant", 2, 22050, None),
        (2048, 512, True, "reflect", 7, 30000, None),   # long runs: register ring shifts, edge blocks at both ends
        (2048, 512, True, "constant", 4, 700, None),    # fewer frames than slots x iters
        (2048, 512, False, "constant", 3, 9001, None),  # uncentred, odd length
        (2048, 512, True, "edge", 5, 9001, None),
        (2048, 1024, True, "reflect", 5, 30000, None),  # HD = 2
        (2048, 2048, True, "symmetric", 3, 30000, None),  # HD = 1: every pair is new
        (2048, 256, True, "constant", 6, 9000, None),   # HD = 8
        (1024, 256, True, "edge", 6, 9000, None),       # two frame slots per wave, radix 8 / 8 / 8
        (1024, 512, True, "constant", 4, 12000, None),
        (1024, 128, False, "constant", 5, 5000, None),
        (4096, 1024, True, "constant", 2, 20000, None),  # two waves per frame: workgroup barriers
        (4096, 512, True, "reflect", 3, 30000, None),
        (2048, 512, True, "constant", 3, 9000, 2.0),    # |X|^power epilogues
        (2048, 512, True, "constant", 3, 9000, 1.0),
        (1024, 256, True, "reflect", 3, 9000, 1.7),
    ],
)
def test_stft_body_second_generation(n_fft, hop, center, pad_mode, iters, n, power):
    """lra_kernels2.h: PCM ring in registers, mirrored last pass, split in registers."""
    rng = np.random.default_rng(n_fft + hop + n)
    y = rng.standard_normal((2, n)).astype(np.float32)
    win = O.get_window("hann", n_fft)
    ref = np.moveaxis(O.stft(y, n_fft=n_fft, hop_length=hop, center=center, pad_mode=pad_mode), -1, -2)
    if power is None:
        out, d = H.stft(y, n_fft, hop, win, center=center, pad_mode=pad_mode, iters_per_wg=iters)
        assert d["v2"] == 1
        _check_diag(d)
        assert out.shape == ref.shape and out.dtype == ref.dtype
        assert np.abs(out - ref).max() <= 2e-6 * np.abs(ref).max()
    else:
        out, d = H.stft(y, n_fft, hop, win, center=center, pad_mode=pad_mode, iters_per_wg=iters, mode=1, power=power)
        assert d["v2"] == 1
        _check_diag(d)
        S = np.abs(ref) ** power
        assert np.abs(out - S).max() <= 4e-6 * S.max()


@pytest.mark.parametrize("n_fft,hop,power,n_mels,dtype,variant", [(2048, 512, 2.0, 128, np.float32, 0), (2048, 512, 2.0, 128, np.float32, 1), (2048, 512, 2.0, 128, np.float32, 4), (1024, 256, 1.0, 40, np.float32, 0),
                                                          (512, 128, 1.5, 20, np.float32, 0), (2048, 512, 2.0, 64, np.float64, 0)])
def test_power_and_mel_body(n_fft, hop, power, n_mels, dtype, variant):
    rng = np.random.default_rng(7)
    y = rng.standard_normal((2, 9000)).astype(dtype)
    win = O.get_window("hann", n_fft)
    S, d = H.stft(y, n_fft, hop, win, mode=1, power=power, variant=variant)
    _check_diag(d)
    Sref = np.moveaxis(O.spectrogram(y=y, n_fft=n_fft, hop_length=hop, power=power)[0], -1, -2)
    assert np.abs(S - Sref).max() <= 4 * _tol(dtype) * Sref.max()
    B = O.mel(sr=22050, n_fft=n_fft, n_mels=n_mels, dtype=dtype)
    Mo, d = H.stft(y, n_fft, hop, win, mode=2, power=power, mel_basis=B, iters_per_wg=2, variant=variant)
    _check_diag(d)
    Mref = O.melspectrogram(y=y, sr=22050, n_fft=n_fft, hop_length=hop, power=power, n_mels=n_mels, dtype=dtype)
    assert Mo.shape == Mref.shape
    # two-slope path (shared filter tables, larger workgroup)
    M2, d2 = H.stft(y, n_fft, hop, win, mode=3, power=power, mel_basis=B, iters_per_wg=5, variant=variant)
    _check_diag(d2)
    assert np.all(np.abs(M2 - Mref) <= 1e-5 * np.abs(Mref) + 1e-5 * Mref.max())
    # mel parity bar |d| <= 1e-4 |ref| + 1e-4 max|ref|; the f32 pipeline is ~100x inside it
    assert np.all(np.abs(Mo - Mref) <= 1e-5 * np.abs(Mref) + 1e-5 * Mref.max())
    # run-ordered two-slope path (16 points per thread only; falls back when a segment needs too many pieces)
    M4, d4 = H.stft(y, n_fft, hop, win, mode=4, power=power, mel_basis=B, iters_per_wg=3, variant=variant)
    if M4 is None:
        assert not (n_fft == 2048 and n_mels == 128 and dtype == np.float32 and variant == 0), d4  # the headline configuration must have it
    else:
        _check_diag(d4)
        assert np.all(np.abs(M4 - Mref) <= 1e-5 * np.abs(Mref) + 1e-5 * Mref.max())


@pytest.mark.parametrize("n_fft,hop,power,n_mels,iters,n,v2", [(2048, 512, 2.0, 128, 3, 9000, 1), (2048, 512, 1.0, 128, 9, 30000, 1), (2048, 512, 2.0, 40, 4, 9000, 1), (2048, 1024, 2.0, 128, 3, 9000, 1),
                                                               (2048, 256, 1.6, 64, 5, 9000, 1), (1024, 256, 2.0, 40, 6, 9000, 1), (2048, 512, 2.0, 128, 3, 9000, 0), (1024, 256, 1.0, 40, 3, 9000, 0)])
def test_mel_body_run_ordered_both_generations(n_fft, hop, power, n_mels, iters, n, v2, monkeypatch):
    """OUT_MELR on the second-generation core (power row in LDS, lra_kernels2.h) and on the first-generation one."""
    if not v2:
        monkeypatch.setenv("LRA_SIM_NO_V2", "1")
    rng = np.random.default_rng(n_fft + n_mels)
    y = rng.standard_normal((3, n)).astype(np.float32)
    win = O.get_window("hann", n_fft)
    B = O.mel(sr=22050, n_fft=n_fft, n_mels=n_mels)
    M4, d4 = H.stft(y, n_fft, hop, win, mode=4, power=power, mel_basis=B, iters_per_wg=iters)
    assert M4 is not None, d4
    assert d4["v2"] == v2
    _check_diag(d4)
    Mref = O.melspectrogram(y=y, sr=22050, n_fft=n_fft, hop_length=hop, power=power, n_mels=n_mels)
    assert np.all(np.abs(M4 - Mref) <= 1e-5 * np.abs(Mref) + 1e-5 * Mref.max())


def _istft_inputs(y, n_fft, hop, center, length, window="hann", win_length=None):
    D = O.stft(y, n_fft=n_fft, hop_length=hop, center=center, window=window, win_length=win_length)
    ref = O.istft(D, hop_length=hop, n_fft=n_fft, center=center, length=length, window=window, win_length=win_length)
    out_len = ref.shape[-1]
    T = D.shape[-1]
    if length:
        padded = length + 2 * (n_fft // 2) if center else length
        n_frames = min(T, int(np.ceil(padded / hop)))
    else:
        n_frames = T
    if center:
        sf = int(np.ceil((n_fft // 2) / hop))
        n_used, drop = max(n_frames, min(T, sf)), n_fft // 2
    else:
        n_used, drop = n_frames, 0
    wss = O.window_sumsquare(window=window, n_frames=n_frames, win_length=win_length, n_fft=n_fft, hop_length=hop, dtype=ref.dtype)
    wss = O.fix_length(wss[drop:], size=out_len)
    win = O.pad_center(O.get_window(window, win_length or n_fft), size=n_fft)
    return np.ascontiguousarray(np.moveaxis(D, -1, -2)), ref, wss, win, out_len, n_used


@pytest.mark.parametrize(
    "n_fft,hop,n,center,length,dtype,strip_groups,window,win_length,variant",
    [
        (2048, 512, 22050, True, "n", np.float32, 2, "hann", None, 0),
        (2048, 512, 22050, True, None, np.float32, 3, "hann", None, 1),
        (2048, 512, 9000, True, "n", np.float32, 1, "hann", None, 3),
        (2048, 512, 9000, True, "n", np.float32, 5, "hann", None, 4),
        (2048, 512, 22050, True, "n", np.float32, 2, "hann", None, 5),  # ascending radices: Hermitian step fused into the first pass
        (2048, 512, 30000, True, None, np.float32, 7, "blackmanharris", None, 5),
        (2048, 512, 9000, False, None, np.float32, 1, "hann", None, 5),
        (2048, 1024, 20000, True, "n", np.float32, 3, "hann", None, 5),
        (2048, 256, 9000, True, 7000, np.float32, 4, "hann", 1200, 5),
        (2048, 128, 9000, True, "n", np.float32, 3, "hann", None, 0),    # hop = n_fft/16: row-aligned overlap-add with one row per hop
        (8192, 512, 30000, True, None, np.float32, 2, "hann", None, 0),  # ... four waves per frame
        (2048, 128, 9000, False, 7000, np.float32, 5, "blackmanharris", None, 5),
        (2048, 1024, 20000, True, "n", np.float32, 3, "hann", None, 0),  # hop = n_fft/2: row-aligned overlap-add with HC = R/2
        (1024, 512, 9000, True, "n", np.float32, 2, "hann", None, 0),
        (1024, 256, 9000, False, None, np.float32, 2, "hann", None, 0),
        (512, 128, 5000, True, 4000, np.float32, 2, "hann", None, 0),
        (512, 128, 5000, True, 6000, np.float32, 2, "hann", None, 0),  # length beyond the frames: zero tail
        (512, 100, 5000, True, "n", np.float32, 2, "hann", None, 0),  # hop does not divide n_fft
        (512, 512, 9000, True, "n", np.float32, 2, "hann", None, 0),  # hop == n_fft
        (256, 300, 5000, True, None, np.float32, 2, "hann", None, 0),  # hop > n_fft: gaps
        (256, 64, 3000, True, "n", np.float64, 2, "hann", None, 0),
        (2048, 512, 9000, True, "n", np.float64, 1, "hann", None, 0),
        (4096, 512, 30000, True, "n", np.float32, 1, "hann", None, 0),
        (512, 128, 300, True, "n", np.float32, 2, "hann", None, 0),
        (64, 16, 1000, True, "n", np.float32, 1, "hann", None, 0),
        (1024, 256, 12000, True, "n", np.float32, 2, "blackmanharris", None, 0),
        (512, 100, 5000, True, "n", np.float32, 2, "hann", 400, 0),
    ],
)
def test_istft_body(n_fft, hop, n, center, length, dtype, strip_groups, window, win_length, variant):
    rng = np.random.default_rng(n_fft + hop)
    y = rng.standard_normal((2, n)).astype(dtype)
    L = n if length == "n" else length
    Dn, ref, wss, win, out_len, n_used = _istft_inputs(y, n_fft, hop, center, L, window, win_length)
    out, d = H.istft(Dn, n_fft, hop, win, wss, out_len, n_used, center=center, strip_groups=strip_groups, variant=variant)
    _check_diag(d)
    assert out.shape == ref.shape
    # Where the window sum-square is tiny (frame edges without overlap) the division amplifies the
    # round-off of ANY implementation by 1/sqrt(wss): compare with that conditioning factored in.
    cond = 1.0 / np.sqrt(np.maximum(wss, np.finfo(np.float32).tiny))
    tol = (4e-6 if dtype == np.float32 else 1e-13) * np.abs(ref).max() * np.maximum(1.0, cond)
    well = wss > 1e-3 * wss.max()
    assert np.all(np.abs(out - ref)[..., well] <= tol[well])
    assert np.all(np.abs(out - ref)[..., ~well] <= 50 * tol[~well] + 1e-3)


def test_pad_index_matches_numpy():
    lib = H.lib()
    for n in (1, 2, 5, 17):
        x = np.arange(n)
        for mode, code in (("reflect", 1), ("edge", 2), ("symmetric", 3)):
            if mode == "reflect" and n == 1:
                ref = np.pad(x, 12, mode="edge")
            else:
                ref = np.pad(x, 12, mode=mode)
            got = np.array([lib.hostsim_pad_index(g, n, code) for g in range(-12, n + 12)])
            assert np.array_equal(got, ref), (n, mode)
        got = [lib.hostsim_pad_index(g, n, 0) for g in (-3, -1, n, n + 4)]
        assert got == [-1, -1, -1, -1]


@pytest.mark.parametrize("n_fft,hop,power,iters,n", [(512, 512, 2.0, 3, 9000), (512, 128, 1.0, 5, 5000), (8192, 512, 1.7, 2, 30000), (256, 300, 2.0, 4, 5000), (8192, 8192, 1.0, 2, 50000)])
def test_power_epilogue_without_lds_ring(n_fft, hop, power, iters, n, monkeypatch):
    """|X|^power through the ring-less framings of the first-generation body: direct framing (hop >= n_fft) and the register ring
    (hop = n_fft / 2 .. 16 at n_fft = 256, 512 and >= 8192)."""
    monkeypatch.setenv("LRA_SIM_NO_V2", "1")
    rng = np.random.default_rng(n_fft + hop + n)
    y = rng.standard_normal((2, n)).astype(np.float32)
    win = O.get_window("hann", n_fft)
    S, d = H.stft(y, n_fft, hop, win, mode=1, power=power, iters_per_wg=iters)
    _check_diag(d)
    ref = np.moveaxis(np.abs(O.stft(y, n_fft=n_fft, hop_length=hop)) ** power, -1, -2)
    assert S.shape == ref.shape and np.abs(S - ref).max() <= 4e-6 * ref.max()
