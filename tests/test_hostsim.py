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
        (8192, 512, True, "reflect", 3, np.float32, 30001, 0),   # register ring, HD = 16 (one new pair per thread and frame), four waves per frame; mirrored last pass
        (8192, 2048, False, "constant", 2, np.float32, 40000, 0),  # HD = 4
        (8192, 4096, True, "reflect", 2, np.float32, 50001, 0),    # HD = 2: last pass mirrored across the halves of a wave (as HD = 16)
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


@pytest.mark.parametrize("n_fft,hop,row_pad,mode,dtype", [(2048, 512, 15, 0, np.float32), (2048, 512, 31, 1, np.float32), (1024, 256, 15, 0, np.float32), (8192, 512, 15, 0, np.float32),
                                                      (512, 512, 15, 0, np.float32), (256, 64, 7, 1, np.float32), (2048, 512, 7, 0, np.float64), (4096, 1024, 15, 0, np.float32)])
def test_stft_body_padded_rows(n_fft, hop, row_pad, mode, dtype):
    """StftArgs::row_pitch (round 5): rows of the complex / power result `M + 1 + row_pad` elements apart -- every kernel family (second generation,
    first generation with the LDS ring, direct framing, the register ring of the large frames) stores the same values as with packed rows
    and writes nothing into the padding."""
    rng = np.random.default_rng(n_fft + hop + row_pad)
    y = rng.standard_normal((2, 3 * n_fft + 777)).astype(dtype)
    win = O.get_window("hann", n_fft)
    packed, d0 = H.stft(y, n_fft, hop, win, iters_per_wg=3, mode=mode)
    padded, d1 = H.stft(y, n_fft, hop, win, iters_per_wg=3, mode=mode, row_pad=row_pad)
    _check_diag(d0)
    _check_diag(d1)
    bins = n_fft // 2 + 1
    assert padded.shape == packed.shape[:-1] + (bins + row_pad,)
    assert np.array_equal(padded[..., :bins], packed) and not np.isnan(packed).any()
    assert np.isnan(padded[..., bins:]).all()


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
                                                               (2048, 256, 1.6, 64, 5, 9000, 1), (1024, 256, 2.0, 40, 6, 9000, 1), (2048, 512, 2.0, 128, 3, 9000, 0), (1024, 256, 1.0, 40, 3, 9000, 0),
                                                               # round 5: 4 / 8 bands per thread where frames share a wave (TF = 32 / 16 / 8), 4-frame output tiles at TF <= 16
                                                               (512, 128, 2.0, 128, 9, 5000, 0), (512, 128, 2.0, 80, 5, 5000, 0), (512, 160, 1.0, 40, 7, 5000, 0), (1024, 256, 2.0, 128, 9, 9000, 0),
                                                               (256, 64, 2.0, 64, 11, 3000, 0), (256, 64, 2.0, 128, 6, 3000, 0), (512, 512, 2.0, 128, 4, 6000, 0)])
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
    assert d4["mel_many"] == int(n_fft == 512 and n_mels > 100)  # eight bands per thread (128-thread workgroups) exactly where lra_api.hip picks that shape
    _check_diag(d4)
    Mref = O.melspectrogram(y=y, sr=22050, n_fft=n_fft, hop_length=hop, power=power, n_mels=n_mels)
    assert np.all(np.abs(M4 - Mref) <= 1e-5 * np.abs(Mref) + 1e-5 * Mref.max())


@pytest.mark.parametrize("hop,power,n_mels,iters,n,variant", [(512, 2.0, 128, 3, 9000, 0), (512, 2.0, 128, 9, 30000, 0), (256, 2.0, 128, 5, 9000, 0), (256, 1.0, 128, 4, 9000, 0), (512, 2.0, 120, 1, 4096, 0),
                                                              (512, 2.0, 128, 3, 9000, 6), (512, 1.0, 128, 7, 20000, 6), (256, 1.0, 128, 4, 9000, 6), (256, 2.0, 125, 2, 6000, 6)])
def test_mel_body_producer_consumer(hop, power, n_mels, iters, n, variant, monkeypatch):
    """The producer / consumer fused mel kernel body (lra_kernels_pc.h: 192-thread workgroups [P, P, C]) against the oracle, and bit for bit against the
    single-wave form it splits (the same operations in the same order; the simulator runs the phase bodies, the flag hand-over itself only runs on the device)."""
    rng = np.random.default_rng(hop + n_mels)
    y = rng.standard_normal((3, n)).astype(np.float32)
    win = O.get_window("hann", 2048)
    B = O.mel(sr=22050, n_fft=2048, n_mels=n_mels)
    M4, d4 = H.stft(y, 2048, hop, win, mode=4, power=power, mel_basis=B, iters_per_wg=iters, variant=variant)  # (variant 6: both on the radix 16-16-4 core)
    monkeypatch.setenv("LRA_SIM_PC", "1")
    Mp, dp = H.stft(y, 2048, hop, win, mode=4, power=power, mel_basis=B, iters_per_wg=iters, variant=variant)
    assert Mp is not None and dp["v2"] == 2 and dp["NT"] == 192 and dp["lds"] <= 40 * 1024, dp  # four workgroups per CU
    _check_diag(dp)
    assert not np.isnan(Mp).any()
    assert np.array_equal(Mp, M4)
    Mref = O.melspectrogram(y=y, sr=22050, n_fft=2048, hop_length=hop, power=power, n_mels=n_mels)
    assert np.all(np.abs(Mp - Mref) <= 1e-5 * np.abs(Mref) + 1e-5 * Mref.max())


def test_mel_body_producer_consumer_declines_wide_segments(monkeypatch):
    """Banks whose pair segments need more than four pieces (40 / 80 bands at n_fft 2048) are not served by the consumer's register lists: the library keeps the one-wave form."""
    monkeypatch.setenv("LRA_SIM_PC", "1")
    y = np.random.default_rng(3).standard_normal((1, 5000)).astype(np.float32)
    Mp, dp = H.stft(y, 2048, 512, O.get_window("hann", 2048), mode=4, power=2.0, mel_basis=O.mel(sr=22050, n_fft=2048, n_mels=40), iters_per_wg=3)
    assert Mp is None and dp == dict(unavailable=3)
    # ... as it does for powers whose producer does not fit three waves per SIMD (pow(): lra_kernels_pc.h, pc_fits_budget)
    Mp, dp = H.stft(y, 2048, 512, O.get_window("hann", 2048), mode=4, power=1.7, mel_basis=O.mel(sr=22050, n_fft=2048, n_mels=128), iters_per_wg=3)
    assert Mp is None and dp == dict(unavailable=3)


@pytest.mark.parametrize("hop,center,pad_mode,iters,n", [(512, True, "constant", 3, 9000), (512, True, "reflect", 5, 20000), (256, True, "symmetric", 4, 9000), (1024, False, "constant", 3, 12000),
                                                         (2048, True, "edge", 2, 9000), (512, False, "constant", 7, 30011)])
def test_stft_body_radix_16_16_4(hop, center, pad_mode, iters, n):
    """Variant 6 (FftCfg PLAN = 1: radices 16, 16, 4 -- four last-pass butterflies per thread, neighbouring bins side by side, 16-byte row pieces):
    complex, |X|^p (also into padded rows) and the run-ordered mel epilogue on that core."""
    rng = np.random.default_rng(hop + n)
    y = rng.standard_normal((2, n)).astype(np.float32)
    win = O.get_window("hann", 2048)
    Dref = np.moveaxis(O.stft(y, n_fft=2048, hop_length=hop, center=center, pad_mode=pad_mode), -1, -2)
    D, d = H.stft(y, 2048, hop, win, center=center, pad_mode=pad_mode, mode=0, iters_per_wg=iters, variant=6)
    assert d["v2"] == 1
    _check_diag(d)
    assert not np.isnan(D).any() and np.abs(D - Dref).max() <= 4 * _tol(np.float32) * np.abs(Dref).max()
    for power in (1.0, 2.0):
        S, d = H.stft(y, 2048, hop, win, center=center, pad_mode=pad_mode, mode=1, power=power, iters_per_wg=iters, variant=6, row_pad=7)
        _check_diag(d)
        Sref = np.abs(Dref) ** power
        assert np.isnan(S[..., 1025:]).all() and np.abs(S[..., :1025] - Sref).max() <= 8 * _tol(np.float32) * Sref.max()
    Dp, d = H.stft(y, 2048, hop, win, center=center, pad_mode=pad_mode, mode=0, iters_per_wg=iters, variant=6, row_pad=15)
    assert np.array_equal(Dp[..., :1025], D) and np.isnan(Dp[..., 1025:]).all()
    if center and hop in (256, 512):
        B = O.mel(sr=22050, n_fft=2048, n_mels=128)
        M4, d4 = H.stft(y, 2048, hop, win, center=center, pad_mode=pad_mode, mode=4, power=2.0, mel_basis=B, iters_per_wg=iters, variant=6)
        assert M4 is not None and d4["v2"] == 1
        _check_diag(d4)
        Mref = O.melspectrogram(y=y, sr=22050, n_fft=2048, hop_length=hop, power=2.0, n_mels=128, center=center, pad_mode=pad_mode)
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
        # round 6: radices 4, 16, 16 (variant 7: four first-pass butterflies per thread, the spectrum row read as 16-byte pieces), every row-aligned hop
        (2048, 512, 9000, True, "n", np.float32, 3, "hann", None, 7),
        (2048, 512, 30011, True, None, np.float32, 7, "hann", None, 7),
        (2048, 1024, 12000, False, "n", np.float32, 2, "hann", None, 7),
        (2048, 256, 9000, True, "n", np.float32, 5, "hamming", None, 7),
        (2048, 128, 5000, True, "n", np.float32, 4, "hann", 1500, 7),
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


# ---- PCEN kernels (librosa_amd/csrc/lra_pcen.h) on host threads ----------------------------------------------------------------
@pytest.mark.parametrize("seed", [0, 1, 440, 2**63 + 12345, None])
def test_pcg64_stream_is_numpys(seed):
    """csrc/lra_rng.h (round 5): the device generator IS np.random.default_rng's PCG64 -- integer arithmetic, so bit for bit --
    for several seeds, sizes that end inside a thread's run, and offsets (jump-ahead) from 0 to beyond 2^32."""
    mk = lambda: np.random.default_rng(seed if seed is not None else np.random.SeedSequence(987654321))
    for count in (1, 127, 128, 129, 5000):
        assert np.array_equal(H.pcg64_random(mk(), 0, count), mk().random(count))
    for offset in (1, 127, 128, 1000003, 2**32 + 17):
        want = mk()
        want.bit_generator.advance(offset)
        assert np.array_equal(H.pcg64_random(mk(), offset, 300), want.random(300))
    # a generator that has already been used: its state says where it is
    g = mk()
    g.random(77)
    g.integers(0, 10, size=5)
    st = g.bit_generator.state
    want = np.random.Generator(np.random.PCG64())
    want.bit_generator.state = st
    assert np.array_equal(H.pcg64_random(g, 0, 1000), want.random(1000))


@pytest.mark.parametrize("dtype,batch,n_bins,n_frames,seg", [(np.float32, 2, 129, 37, 16), (np.float64, 1, 513, 9, 256), (np.float32, 3, 300, 70, 32)])
def test_griffinlim_init_pcg64_body(dtype, batch, n_bins, n_frames, seg):
    """angles = S exp(2 pi i u) with u drawn on the device in the reference's order (rng.random(S.shape), S = (clip, bin, frame)) into the frame-major layout:
    the same values as the host-drawn path (float64 phasor, one rounding)."""
    rng = np.random.default_rng(2024)
    S = np.abs(np.random.default_rng(5).standard_normal((batch, n_bins, n_frames))).astype(dtype)
    got = H.griffinlim_init_pcg64(rng, np.ascontiguousarray(np.swapaxes(S, -1, -2)), seg=seg)
    u = np.random.default_rng(2024).random(S.shape)
    a = 2 * np.pi * u
    want = (np.cos(a).astype(dtype) * S) + 1j * (np.sin(a).astype(dtype) * S)
    want = np.swapaxes(want, -1, -2).astype(got.dtype)
    assert np.abs(got - want).max() <= (1e-15 if dtype == np.float64 else 2e-7) * np.abs(want).max()
    assert np.array_equal(got.real == 0, want.real == 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("rows,n", [(1, 1), (5, 63), (16, 64), (17, 65), (40, 200)])
def test_pcen_body(dtype, rows, n):
    """Rows not a multiple of the 16 a wave owns, frame counts around the 64-frame tile, every output branch, carried state."""
    import scipy.signal

    rng = np.random.default_rng(rows * 1000 + n)
    X = (rng.standard_normal((rows, n)) ** 2).astype(dtype)
    base = dict(b=0.05, gain=0.98, bias=2.0, power=0.5, eps=1e-6)
    for kw in (dict(), dict(power=0.0), dict(bias=0.0, power=0.25), dict(gain=0.8, bias=10.0, power=0.25), dict(b=1.0, gain=1.0, bias=0.0, power=1.0, eps=1e-20)):
        a = {**base, **kw}
        got, zf = H.pcen(X, zi_scalar=float(scipy.signal.lfilter_zi([a["b"]], [1, a["b"] - 1])[0]), want_zf=True, **a)
        exp, ezf = O.pcen(X, return_zf=True, **a)
        # log(S) of float32 input is a float32 log in the reference (NumPy's SIMD loop): last-bit differences scale with |log S|
        tol = 5e-6 if (dtype == np.float32 and a["bias"] == 0) else 1e-13
        assert np.all(np.abs(got - exp) <= tol * np.abs(exp)), (kw, np.max(np.abs(got - exp) / np.abs(exp).clip(1e-300)))
        assert np.array_equal(zf, ezf[:, 0])
    zi = rng.random(rows)
    ref = (rng.standard_normal((rows, n)) ** 2).astype(dtype)
    got, zf = H.pcen(X, ref=ref, zi=zi, zi_scalar=np.nan, want_zf=True, **base)
    exp, ezf = O.pcen(X, ref=ref, zi=zi[:, None], return_zf=True, **base)
    assert np.all(np.abs(got - exp) <= 1e-13 * np.abs(exp)) and np.array_equal(zf, ezf[:, 0])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_maxfilter_body(dtype):
    rng = np.random.default_rng(8)
    for shape in ((1, 1, 1), (2, 7, 5), (3, 40, 70), (1, 5, 300)):
        X = rng.standard_normal(shape).astype(dtype)
        for size in (1, 2, 3, 4, 9, 23):
            assert np.array_equal(H.maxfilter(X, size), O.maximum_filter1d(X, size, 1)), (shape, size)


class _SimCtx:
    """The entry points of librosa_amd._native.Context that the PCEN and constant-Q shims call, on host memory: the kernels of
    lra_pcen.h / lra_cqt.h through the simulator, the STFT through the oracle (the forward kernel bodies have their own cases above)."""

    @staticmethod
    def _view(ptr, shape, dtype):
        import ctypes

        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return np.frombuffer((ctypes.c_char * n).from_address(ptr), dtype=dtype).reshape(shape)

    def device_table(self, key, build):
        tables = self.__dict__.setdefault("_tables", {})
        if key not in tables:
            tables[key] = np.ascontiguousarray(build())
        return tables[key].ctypes.data

    SIDE_FORK, SIDE_BACK, SIDE_JOIN, SIDE_END = 1, 2, 3, 4

    def side(self, mode):
        pass

    def nonfinite_reset(self):
        pass

    def nonfinite_read(self):
        return False

    def stft_plan(self, n_fft, hop, window, center, pad_mode, dtype):
        return dict(n_fft=int(n_fft), hop=int(hop), window=np.asarray(window), center=center, pad_mode=pad_mode, dtype=np.dtype(dtype))

    def stft_exec(self, plan, y_ptr, batch, n, y_stride, out_ptr):
        y = self._view(y_ptr, (batch, n), plan["dtype"])
        # a float64 window, as the reference holds it: the product of window and frame, and with it the FFT, is then float64 (core/spectrum.py:264, 372)
        D = O.stft(y, n_fft=plan["n_fft"], hop_length=plan["hop"], window=plan["window"].astype(np.float64), center=plan["center"], pad_mode=plan["pad_mode"])   # (batch, bins, frames)
        self._view(out_ptr, (batch, D.shape[-1], D.shape[-2]), D.dtype)[...] = np.swapaxes(D, -1, -2)

    def fir_decimate_exec(self, x_ptr, out_ptr, batch, n_in, n_out, taps_ptr, n_taps, down, first, div, mul, dtype):
        H.post_lib().postsim_fir_decimate(x_ptr, out_ptr, batch, n_in, n_out, taps_ptr, int(n_taps), int(down), int(first), float(div), float(mul), int(np.dtype(dtype) == np.float64))

    def resample_poly_exec(self, x_ptr, out_ptr, batch, n_in, n_out, taps_ptr, n_taps, up, down, first, div, mul, dtype):
        H.post_lib().postsim_resample_poly(x_ptr, out_ptr, batch, n_in, n_out, taps_ptr, int(n_taps), int(up), int(down), int(first), float(div), float(mul), int(np.dtype(dtype) == np.float64))

    def resample_fft_exec(self, x_ptr, out_ptr, batch, n_in, n_out, gain, dtype):
        # (the whole-signal transforms are rocFFT's on the device: scipy's here, like the oracle's STFT above)
        import scipy.signal

        x = self._view(x_ptr, (batch, n_in), dtype)
        self._view(out_ptr, (batch, n_out), dtype)[...] = (scipy.signal.resample(x, n_out, axis=-1) * gain).astype(dtype)

    def resample_band_exec(self, x_ptr, out_ptr, batch, n_in, n_out, fft_in, fft_out, k_mid, k_sigma, gain, dtype):
        # (rocFFT's transforms on the device: NumPy's here; the roll-off as resample_shaped_spectrum_kernel applies it)
        import scipy.special

        x = self._view(x_ptr, (batch, n_in), dtype)
        X = np.fft.rfft(x.astype(np.float64), n=fft_in, axis=-1)
        n_copy = min(fft_in, fft_out) // 2 + 1
        Y = np.zeros((batch, fft_out // 2 + 1), dtype=np.complex128)
        Y[:, :n_copy] = X[:, :n_copy] * (0.5 * scipy.special.erfc((np.arange(n_copy) - k_mid) / k_sigma))
        y = np.fft.irfft(Y, n=fft_out, axis=-1) * (fft_out / fft_in) * gain
        self._view(out_ptr, (batch, n_out), dtype)[...] = y[:, :n_out].astype(dtype)

    def cqt_project_exec(self, d_ptr, out_ptr, row_ptr, col_ptr, val_ptr, sqrt_len_ptr, batch, frames_in, n_bins, n_frames, n_total, bin0, row0, n_rows, dtype):
        H.post_lib().postsim_cqt_project(d_ptr, out_ptr, row_ptr, col_ptr, val_ptr, sqrt_len_ptr, batch, frames_in, int(n_bins), n_frames, int(n_total), int(bin0), int(row0), int(n_rows),
                                         int(np.dtype(dtype) == np.float64))

    # the fused octave (lra_cqt_octave_exec): the whole kernel body of csrc/lra_mixed.h, mixed_cqt_kernel, through the simulator
    fused_octaves = False  # (the bit-for-bit shim tests take the oracle's STFT; test_cqt_fused_octaves_through_simulator turns this on)

    def cqt_octave_supported(self, n_fft):
        return self.fused_octaves and int(n_fft) in (32, 64, 128, 256, 512, 1024, 2048)

    def cqt_octave_exec(self, y_ptr, batch, n, y_stride, n_fft, hop, pad_mode, row_ptr, col_ptr, val_ptr, sqrt_len_ptr, out_ptr, n_frames, n_total, bin0, row0, n_rows, dtype):
        import ctypes as c

        f64 = np.dtype(dtype) == np.float64
        ct = np.complex128 if f64 else np.complex64
        M = int(n_fft) // 2
        tw_m = np.ascontiguousarray(np.exp(-2j * np.pi * np.arange(M, dtype=np.float64) / M).astype(ct))
        tw_n = np.ascontiguousarray(np.exp(-2j * np.pi * np.arange(M + 1, dtype=np.float64) / n_fft).astype(ct))
        fn = H.post_lib().postsim_cqt_octave
        fn.argtypes = [c.c_int, c.c_int, c.c_void_p, c.c_longlong, c.c_longlong, c.c_longlong, c.c_int, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p,
                       c.c_longlong, c.c_int, c.c_int, c.c_int, c.c_int]
        rc = fn(int(n_fft), int(f64), y_ptr, batch, n, y_stride, int(hop), {"constant": 0, "reflect": 1, "edge": 2, "symmetric": 3}[pad_mode], tw_m.ctypes.data, tw_n.ctypes.data, row_ptr, col_ptr,
                val_ptr, sqrt_len_ptr, out_ptr, n_frames, int(n_total), int(bin0), int(row0), int(n_rows))
        assert rc == 0

    def transpose(self, src_ptr, dst_ptr, batch, rows, cols, elem_bytes):
        """dst[b][c][r] = src[b][r][c] (lra_transpose)."""
        dt = {4: np.float32, 8: np.float64, 16: np.complex128}[int(elem_bytes)]
        src = self._view(src_ptr, (batch, rows, cols), dt)
        self._view(dst_ptr, (batch, cols, rows), dt)[...] = np.swapaxes(src, 1, 2)

    def magnitude_exec(self, d_ptr, mag_ptr, count, dtype):
        H.post_lib().postsim_magnitude(d_ptr, mag_ptr, count, int(np.dtype(dtype) == np.float64))

    def magphase_exec(self, d_ptr, is_complex, mag_ptr, phase_ptr, count, power, dtype):
        H.post_lib().postsim_magphase(d_ptr, int(bool(is_complex)), mag_ptr, phase_ptr, count, float(power), int(np.dtype(dtype) == np.float64))

    def hpss_exec(self, mag_ptr, d_ptr, out_h_ptr, out_p_ptr, batch, n_frames, n_bins, win_harm, win_perc, power, margin_harm, margin_perc, want_mask, dtype):
        H.post_lib().postsim_hpss(mag_ptr, d_ptr, out_h_ptr, out_p_ptr, batch, n_frames, int(n_bins), int(win_harm), int(win_perc), float(power), float(margin_harm), float(margin_perc),
                                  int(bool(want_mask)), int(np.dtype(dtype) == np.float64))

    def pcen_exec(self, s_ptr, ref_ptr, out_ptr, rows, n_frames, dtype, b, gain, bias, power, eps, zi_ptr, zi_scalar, zf_ptr):
        H.post_lib().postsim_pcen(s_ptr, ref_ptr, out_ptr, rows, n_frames, int(np.dtype(dtype) == np.float64), float(b), float(gain), float(bias), float(power), float(eps), zi_ptr,
                                  float(zi_scalar), zf_ptr)

    def maxfilter_exec(self, s_ptr, out_ptr, outer, n_bands, inner, size, dtype):
        H.post_lib().postsim_maxfilter(s_ptr, out_ptr, outer, n_bands, inner, int(size), int(np.dtype(dtype) == np.float64))


class _SimSession:
    """Stands in for librosa_amd._arrays.Session in the CPU test below: host arrays instead of device buffers, and a context whose
    pcen / max-filter entry points run the kernel bodies on host threads (the argument order of librosa_amd._native.Context)."""

    is_torch = False

    def __init__(self, like):
        self._keep = []
        self.ctx = _SimCtx()

    def input_raw(self, a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        self._keep.append(a)
        return a.ctypes.data

    def scratch(self, nbytes):
        a = np.empty(max(int(nbytes), 16), dtype=np.uint8)
        self._keep.append(a)
        return a.ctypes.data

    def output(self, shape, dtype):
        a = np.full(shape, np.nan, dtype=dtype)
        return a.ctypes.data, a

    def result(self, handle):
        return handle

    def close(self):
        pass

    def input_2d(self, x, dtype):
        a = np.ascontiguousarray(x, dtype=dtype).reshape(-1, x.shape[-1])
        self._keep.append(a)
        return a.ctypes.data, a.shape[0], a.shape[1], a.shape[1]


def _sim_torch_session(real_session):
    """librosa_amd._arrays.Session's own tensor marshalling (input_2d / input_raw / output / scratch / result), on CPU tensors whose
    data_ptr() the simulator can read: the device-tensor code paths of the shims run in the CPU suite."""
    import torch

    class _SimTorchSession(real_session):
        def __init__(self, like):
            assert isinstance(like, torch.Tensor)
            self.is_torch = True
            self._keep = []
            self._locked = False
            self.device = torch.device("cpu")
            self.ctx = _SimCtx()

    return _SimTorchSession


def test_pcen_shim_layouts_through_simulator(monkeypatch):
    """librosa_amd.pcen's host side (axis / max_axis permutations, state shapes, broadcasting of zi and ref, dtype promotion) with the
    kernels run by the simulator: against the oracle on the layouts of the reference's tests (tests/test_core.py:2459-2573)."""
    import librosa_amd
    from librosa_amd import _arrays

    monkeypatch.setattr(_arrays, "Session", _SimSession)
    rng = np.random.default_rng(77)
    X = rng.standard_normal((3, 20, 33)) ** 2

    def close(got, exp):
        if isinstance(got, tuple):
            return all(close(g, e) for g, e in zip(got, exp))
        return got.shape == exp.shape and got.dtype == exp.dtype and np.all(np.abs(got - exp) <= 1e-13 * np.abs(exp))

    for kw in (dict(), dict(axis=0), dict(axis=1), dict(axis=-2, return_zf=True), dict(max_size=3, max_axis=1), dict(max_size=4, max_axis=0, axis=1), dict(max_size=2, max_axis=2, axis=0),
               dict(max_size=3, max_axis=-1), dict(power=0, b=0.3, return_zf=True)):
        assert close(librosa_amd.pcen(X, **kw), O.pcen(X, **kw)), kw
    for kw in (dict(), dict(max_size=3), dict(axis=0, max_size=5), dict(ref=np.ones((20, 1))), dict(ref=X[1].astype(np.float32)), dict(zi=np.full((1, 1), 0.25)), dict(zi=rng.random((20, 1)), return_zf=True)):
        for A in (X[0], X[0].astype(np.float32), np.asfortranarray(X[0])):
            assert close(librosa_amd.pcen(A, **kw), O.pcen(A, **kw)), kw
    assert close(librosa_amd.pcen(np.arange(50)), O.pcen(np.arange(50)))
    assert close(librosa_amd.pcen(np.arange(50), return_zf=True), O.pcen(np.arange(50), return_zf=True))
    s1, z1 = librosa_amd.pcen(X[:, :, :10], axis=-1, return_zf=True)
    assert z1.shape == (3, 20, 1) and close(librosa_amd.pcen(X[:, :, 10:], zi=z1), O.pcen(X[:, :, 10:], zi=z1))
    with pytest.raises(librosa_amd.ParameterError):
        librosa_amd.pcen(X[0], zi=np.zeros((3, 1)))
    with pytest.raises(librosa_amd.ParameterError):
        librosa_amd.pcen(X[0], ref=np.ones((3, 3)))


# ---- constant-Q kernels (librosa_amd/csrc/lra_cqt.h) on the host ---------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("down,n", [(2, 1), (2, 41), (2, 1000), (2, 1001), (4, 999), (8, 4096), (3, 500), (128, 40000)])   # the last: span beyond the LDS -> direct kernel
def test_fir_decimate_body_is_resample_poly(dtype, down, n):
    """The decimator with scipy's own design and alignment == scipy.signal.resample_poly(x, 1, down), bit for bit (same taps,
    same summation order, no contraction), incl. the scale=True division of librosa.resample."""
    import scipy.signal
    from librosa_amd.core.constantq import _decimator

    rng = np.random.default_rng(down * 7919 + n)
    x = rng.standard_normal((3, n)).astype(dtype)
    taps, first = _decimator(down, "polyphase", np.dtype(dtype))
    n_out = -(-n // down)
    got = H.fir_decimate(x, taps, down, first, n_out)
    assert np.array_equal(got, scipy.signal.resample_poly(x, 1, down, axis=-1))
    import cqt_oracle as CQ

    got = H.fir_decimate(x, taps, down, first, n_out, div=np.sqrt(1.0 / down))
    assert np.array_equal(got, CQ.resample(x, orig_sr=down, target_sr=1, res_type="polyphase", scale=True))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("up,down,n", [(2, 1, 300), (3, 2, 1000), (2, 3, 1001), (160, 441, 5000), (441, 160, 700), (320, 441, 4001), (1, 3, 777), (5, 4, 1)])
def test_resample_poly_body_is_resample_poly(dtype, up, down, n):
    """The general rational resampler with scipy's own design and alignment == scipy.signal.resample_poly(x, up, down), bit for bit
    (same taps, same summation order, no contraction)."""
    import scipy.signal
    from librosa_amd.core.audio import _rational_filter

    rng = np.random.default_rng(up * 7919 + down * 31 + n)
    x = rng.standard_normal((2, n)).astype(dtype)
    taps, first = _rational_filter(up, down, "polyphase", np.dtype(dtype))
    n_out = -(-n * up // down)
    got = H.resample_poly(x, taps, up, down, first, n_out)
    exp = scipy.signal.resample_poly(x, up, down, axis=-1)
    assert got.shape == exp.shape and np.array_equal(got, exp)


def test_resample_through_simulator(monkeypatch):
    """librosa_amd.resample(res_type="polyphase") -- axis handling, integer-rate check, fix / scale, dtype, errors -- through the simulator,
    against the oracle's restatement of librosa.resample (bit-identical)."""
    import cqt_oracle as CQ
    import librosa_amd
    from librosa_amd import _arrays
    from librosa_amd.util.exceptions import ParameterError

    monkeypatch.setattr(_arrays, "Session", _SimSession)
    rng = np.random.default_rng(5)
    y = rng.standard_normal((2, 3, 4000)).astype(np.float32)
    for orig, target, kw in ((22050, 8000, {}), (22050, 16000, dict(scale=True)), (8000, 22050, {}), (44100, 22050, dict(scale=True)), (22050, 11025.0, {}), (3, 2, dict(fix=False))):
        got = librosa_amd.resample(y, orig_sr=orig, target_sr=target, res_type="polyphase", **kw)
        exp = CQ.resample(y, orig_sr=orig, target_sr=target, res_type="polyphase", **{k: v for k, v in kw.items() if k != "fix"})
        assert got.dtype == exp.dtype and got.shape == exp.shape and np.array_equal(got, exp), (orig, target, kw)
    yt = np.ascontiguousarray(np.moveaxis(y, -1, 0))                    # time first
    got = librosa_amd.resample(yt, orig_sr=22050, target_sr=8000, res_type="polyphase", axis=0)
    assert np.array_equal(np.moveaxis(got, 0, -1), CQ.resample(y, orig_sr=22050, target_sr=8000, res_type="polyphase"))
    y64 = y[0, 0].astype(np.float64)
    got = librosa_amd.resample(y64, orig_sr=4, target_sr=3, res_type="polyphase")
    assert got.dtype == np.float64 and np.array_equal(got, CQ.resample(y64, orig_sr=4, target_sr=3, res_type="polyphase"))
    assert librosa_amd.resample(y, orig_sr=8000, target_sr=8000) is y
    tone = np.sin(0.07 * np.arange(4000)) + 0.5 * np.cos(0.31 * np.arange(4000))   # well inside both pass bands
    for orig, target in ((2, 1), (3, 2), (2, 3)):                               # the library's own design: band-limited, same output grid
        own = librosa_amd.resample(tone, orig_sr=orig, target_sr=target, res_type="soxr_hq")
        ref = CQ.resample(tone, orig_sr=orig, target_sr=target, res_type="polyphase")
        assert own.shape == ref.shape and np.abs(own - ref)[100:-100].max() < 5e-3, (orig, target)
    for bad in (dict(orig_sr=22050.5, target_sr=8000, res_type="polyphase"), dict(orig_sr=22050, target_sr=8000, res_type="linear"), dict(orig_sr=-1, target_sr=8000, res_type="fft")):
        with pytest.raises(ParameterError):
            librosa_amd.resample(y, **bad)
    with pytest.raises(ParameterError):
        librosa_amd.resample(y.astype(np.int16), orig_sr=2, target_sr=1)
    with pytest.raises(ParameterError):
        librosa_amd.resample(np.array([1.0, np.nan]), orig_sr=2, target_sr=1)


def test_own_decimator_design():
    """The default decimator (soxr-HQ band edges): unit DC gain, pass band flat to 1e-5 up to 0.913 of the new Nyquist, more than
    120 dB down from the new Nyquist on, integer alignment (a centred impulse stays centred)."""
    import scipy.signal
    from librosa_amd.core.constantq import _decimator

    for down in (2, 4, 8):
        taps, first = _decimator(down, "soxr_hq", np.dtype(np.float64))
        assert len(taps) % 2 == 1 and (len(taps) // 2) == first * down and np.allclose(taps, taps[::-1])
        w, h = scipy.signal.freqz(taps, worN=1 << 15)
        f = w / np.pi * down                              # in units of the new Nyquist
        mag = np.abs(h)
        assert abs(mag[0] - 1) < 1e-12 and np.all(np.abs(mag[f <= 0.913] - 1) < 1e-5) and np.all(mag[f >= 1.0] < 1e-6)
        x = np.zeros((1, 64 * down + 1))
        x[0, 32 * down] = 1.0
        y = H.fir_decimate(x, taps, down, first, -(-x.shape[1] // down))
        assert np.argmax(np.abs(y[0])) == 32 and abs(y[0, 32] - taps[len(taps) // 2]) < 1e-15


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_cqt_project_body(dtype):
    """Sparse basis x frames + length scaling + stacking offsets against scipy.sparse's product on the same arrays."""
    import scipy.sparse

    rng = np.random.default_rng(12)
    n_bins, rows = 129, 12
    dense = (rng.standard_normal((rows, n_bins)) + 1j * rng.standard_normal((rows, n_bins))) * (rng.random((rows, n_bins)) < 0.15)
    csr = scipy.sparse.csr_array(dense.astype(dtype))
    D = (rng.standard_normal((2, 37, n_bins)) + 1j * rng.standard_normal((2, 37, n_bins))).astype(dtype)
    sqrt_len = np.sqrt(rng.random(rows) * 100 + 1)
    for (n_frames, n_total, bin0, row0, n_rows, scaled) in ((37, 40, 20, 0, 12, True), (30, 12, 0, 0, 12, False), (37, 7, 0, 5, 7, True), (1, 30, 18, 0, 12, True)):
        got = H.cqt_project(D, csr, n_frames, n_total, bin0, row0, n_rows, sqrt_len[row0 : row0 + n_rows] if scaled else None)
        exp = np.zeros_like(got)
        for b in range(2):
            block = csr.dot(D[b].T)[row0 : row0 + n_rows, :n_frames]           # (rows, frames), scipy's own accumulation order
            if scaled:
                block = (block / sqrt_len[row0 : row0 + n_rows, None]).astype(dtype)
            exp[b, :, bin0 : bin0 + n_rows] = block.T
        assert np.array_equal(got, exp), (n_frames, n_total, bin0, row0)


@pytest.mark.filterwarnings("ignore:n_fft=")
def test_cqt_shim_through_simulator(monkeypatch):
    """librosa_amd.cqt / vqt's host side (tables, octave schedule, early downsampling, stacking, scaling) with the decimator and the
    projection run by the simulator and the STFT by the oracle: bit for bit the oracle's transform for res_type="polyphase"."""
    import cqt_oracle as CQ
    import golden_cases
    import librosa_amd
    from librosa_amd import _arrays

    monkeypatch.setattr(_arrays, "Session", _SimSession)
    y = golden_cases.make_signal("mix", 22050, 3, None, "float32")
    ys = golden_cases.make_signal("mix", 9000, 4, (2,), "float32")
    for kw in (dict(), dict(hop_length=256, n_bins=60), dict(n_bins=24), dict(n_bins=30, bins_per_octave=12), dict(scale=False, n_bins=24), dict(fmin=110.0, n_bins=36, tuning=0.2, norm=2),
               dict(filter_scale=0.5, pad_mode="reflect", window="hamming", sparsity=0.05), dict(n_bins=1)):
        got = librosa_amd.cqt(y, res_type="polyphase", **kw)
        exp = CQ.cqt(y, res_type="polyphase", **kw)
        assert got.shape == exp.shape and got.dtype == exp.dtype and np.array_equal(got, exp), kw
    for kw in (dict(n_bins=None), dict(n_bins=None, fmin=110.0, bins_per_octave=24), dict(n_bins=None, sr=16000, fmin=200.0, filter_scale=0.5)):   # as many bins as fit below Nyquist
        got, exp = librosa_amd.cqt(y, res_type="polyphase", **kw), CQ.cqt(y, res_type="polyphase", **kw)
        assert got.shape == exp.shape and np.array_equal(got, exp), kw
    assert np.array_equal(librosa_amd.vqt(y, n_bins=None, res_type="polyphase"), CQ.vqt(y, n_bins=None, gamma=None, res_type="polyphase"))
    for kw in (dict(gamma=None, bins_per_octave=24, n_bins=96), dict(gamma=5, scale=False), dict(intervals=[1, 1.2, 1.5, 1.8], n_bins=16, fmin=200.0, gamma=0)):
        assert np.array_equal(librosa_amd.vqt(ys, res_type="polyphase", **kw), CQ.vqt(ys, res_type="polyphase", **kw)), kw
    y64 = y.astype(np.float64)
    got, exp = librosa_amd.cqt(y64, res_type="polyphase", n_bins=48), CQ.cqt(y64, res_type="polyphase", n_bins=48)
    assert got.dtype == np.complex128 and np.array_equal(got, exp)
    for bad in (dict(tuning=None), dict(fmin=20000.0), dict(n_bins=200), dict(pad_mode="wrap"), dict(hop_length=0)):
        with pytest.raises(librosa_amd.ParameterError):
            librosa_amd.cqt(y, **bad)
    with pytest.raises(librosa_amd.ParameterError):
        librosa_amd.vqt(y, intervals="pythagorean")


@pytest.mark.filterwarnings("ignore:n_fft=")
def test_tensor_code_paths_through_simulator(monkeypatch):
    """The device-tensor branches of librosa_amd.pcen / cqt (tensor marshalling, table uploads, carried state as a tensor, results
    as tensors), exercised on CPU tensors: same values as the NumPy branches."""
    import torch

    import golden_cases
    import librosa_amd
    from librosa_amd import _arrays

    rng = np.random.default_rng(31)
    X = (rng.standard_normal((2, 20, 33)) ** 2).astype(np.float32)
    y = golden_cases.make_signal("mix", 12000, 8, (2,), "float32")
    real_session = _arrays.Session
    monkeypatch.setattr(_arrays, "Session", _SimSession)
    host = dict(p=librosa_amd.pcen(X), pm=librosa_amd.pcen(X, max_size=3, max_axis=1, axis=2), pz=librosa_amd.pcen(X[..., :10], return_zf=True), pt=librosa_amd.pcen(X, axis=1),
                c=librosa_amd.cqt(y, res_type="polyphase", n_bins=48), v=librosa_amd.vqt(y, n_bins=36, scale=False))
    monkeypatch.setattr(_arrays, "Session", _sim_torch_session(real_session))
    Xt, yt = torch.from_numpy(X), torch.from_numpy(y)
    p = librosa_amd.pcen(Xt)
    assert isinstance(p, torch.Tensor) and p.dtype == torch.float64 and np.array_equal(p.numpy(), host["p"])
    assert np.array_equal(librosa_amd.pcen(Xt, max_size=3, max_axis=1, axis=2).numpy(), host["pm"])
    assert np.array_equal(librosa_amd.pcen(Xt, axis=1).numpy(), host["pt"])
    p1, z1 = librosa_amd.pcen(Xt[..., :10], return_zf=True)
    assert isinstance(z1, torch.Tensor) and np.array_equal(p1.numpy(), host["pz"][0]) and np.array_equal(z1.numpy(), host["pz"][1])
    p2 = librosa_amd.pcen(Xt[..., 10:], zi=z1)                       # tensor state
    p2h = librosa_amd.pcen(Xt[..., 10:], zi=host["pz"][1])           # host state, tensor data
    assert np.array_equal(p2.numpy(), p2h.numpy()) and np.allclose(torch.cat([p1, p2], dim=-1).numpy(), host["p"], rtol=1e-12, atol=0)
    assert np.array_equal(librosa_amd.pcen(Xt.to(torch.int32)).numpy(), librosa_amd.pcen(torch.from_numpy(X.astype(np.int32).astype(np.float64))).numpy())
    c = librosa_amd.cqt(yt, res_type="polyphase", n_bins=48)
    assert isinstance(c, torch.Tensor) and c.dtype == torch.complex64 and tuple(c.shape) == host["c"].shape and np.array_equal(c.numpy(), host["c"])
    assert np.array_equal(librosa_amd.vqt(yt, n_bins=36, scale=False).numpy(), host["v"])
    with pytest.raises(librosa_amd.ParameterError):
        librosa_amd.cqt(yt.to(torch.int32))


@pytest.mark.filterwarnings("ignore")
def test_shims_seeded_sweep_through_simulator(monkeypatch):
    """A seeded sweep over shapes, axes, max-filter axes, carried states and parameter sets (PCEN) and over rates, hops, bin counts,
    octave widths, filter scales, pad modes and signal lengths down to a few hundred samples (CQT): the shims with simulated
    kernels against the oracle; where the oracle raises, the shim raises too."""
    import cqt_oracle as CQ
    import librosa_amd
    from librosa_amd import _arrays

    monkeypatch.setattr(_arrays, "Session", _SimSession)
    rng = np.random.default_rng(2024)
    compared = 0
    for _ in range(60):
        nd = int(rng.integers(1, 5))
        shape = tuple(int(rng.integers(1, 9)) for _ in range(nd))
        dt = rng.choice([np.float32, np.float64])
        X = (rng.standard_normal(shape) ** 2).astype(dt)
        axis = int(rng.integers(-nd, nd))
        kw = dict(axis=axis, gain=float(rng.choice([0.98, 0.5, 0.0])), bias=float(rng.choice([2.0, 0.0, 10.0])), power=float(rng.choice([0.5, 0.0, 0.25, 1.0])),
                  time_constant=float(rng.choice([0.4, 0.06])), eps=float(rng.choice([1e-6, 1e-3])))
        if nd >= 2 and rng.random() < 0.5:
            kw.update(max_size=int(rng.integers(1, 6)), max_axis=int(rng.integers(-nd, nd)))
        if rng.random() < 0.3:
            kw["b"] = float(rng.random())
        if rng.random() < 0.3:
            kw["return_zf"] = True
        if rng.random() < 0.3:
            kw["zi"] = rng.random(tuple(1 if a == axis % nd else shape[a] for a in range(nd)))
        try:
            exp = O.pcen(X, **kw)
        except O.ParameterError:
            with pytest.raises(librosa_amd.ParameterError):
                librosa_amd.pcen(X, **kw)
            continue
        got = librosa_amd.pcen(X, **kw)
        tol = 5e-6 if (kw["bias"] == 0 and dt == np.float32) else 1e-12
        for g, e in zip(got if isinstance(got, tuple) else (got,), exp if isinstance(exp, tuple) else (exp,)):
            assert g.shape == e.shape and g.dtype == e.dtype and np.all((np.abs(g - e) <= tol * np.abs(e)) | (g == e)), (shape, kw)
        compared += 1
    assert compared >= 50
    compared = 0
    for _ in range(20):
        n = int(rng.choice([300, 1000, 5000, 12000]))
        ch = int(rng.choice([0, 1, 2]))
        y = rng.standard_normal((n,) if ch == 0 else (ch, n)).astype(rng.choice([np.float32, np.float64]))
        bpo = int(rng.choice([12, 24, 7]))
        kw = dict(sr=float(rng.choice([22050, 16000, 44100])), hop_length=int(rng.choice([512, 256, 64, 100, 384])), n_bins=int(rng.integers(1, 6 * bpo)), bins_per_octave=bpo,
                  fmin=float(rng.choice([32.7, 55.0, 110.0, 400.0])), filter_scale=float(rng.choice([1.0, 0.5, 2.0])), scale=bool(rng.random() < 0.7),
                  pad_mode=str(rng.choice(["constant", "reflect", "edge"])), tuning=float(rng.choice([0.0, 0.3])), sparsity=float(rng.choice([0.01, 0.0, 0.1])),
                  norm=rng.choice([1, 2, np.inf]), window=str(rng.choice(["hann", "hamming"])))
        try:
            exp = CQ.cqt(y, res_type="polyphase", **kw)
        except (O.ParameterError, ValueError):
            with pytest.raises((librosa_amd.ParameterError, ValueError)):
                librosa_amd.cqt(y, res_type="polyphase", **kw)
            continue
        got = librosa_amd.cqt(y, res_type="polyphase", **kw)
        assert got.shape == exp.shape and got.dtype == exp.dtype and np.array_equal(got, exp), (n, y.shape, kw)
        compared += 1
    assert compared >= 12


# ---- harmonic / percussive separation (librosa_amd/csrc/lra_hpss.h) -------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_hpss_body(dtype):
    """Both medians (sorting networks of 32 and 64 slots, counting selection beyond; odd and even windows; windows longer than the axis),
    soft / hard masks, margins, masked output with the input's phase -- against the oracle (bit-identical to the reference).  Real
    input: bit for bit (the median is a selection); complex input and non-trivial exponents: last-bit differences of |D| and pow."""
    import golden_cases

    y = golden_cases.make_signal("mix", 6000, 7, (2,), "float32").astype(dtype)
    D = O.stft(y, n_fft=128, hop_length=64)                       # (2, 65, 95)
    Dt = np.ascontiguousarray(np.swapaxes(D, -1, -2))
    eps = 1e-6 if dtype == np.float32 else 1e-14
    for kw in (dict(), dict(kernel_size=(13, 31), margin=(1.0, 3.0)), dict(power=1.0, mask=True), dict(kernel_size=8, power=3.5), dict(power=np.inf), dict(margin=2.5, mask=True),
               dict(kernel_size=(40, 5)), dict(kernel_size=(70, 9), power=0.5), dict(kernel_size=(3, 200))):
        wh, wp = kw.get("kernel_size", 31) if isinstance(kw.get("kernel_size", 31), tuple) else (kw.get("kernel_size", 31),) * 2
        mh, mp = kw.get("margin", 1.0) if isinstance(kw.get("margin", 1.0), tuple) else (kw.get("margin", 1.0),) * 2
        for inp, inp_t in ((D, Dt), (np.abs(D) ** 2, np.abs(Dt) ** 2)):
            exp = O.hpss(inp, **kw)
            got = H.hpss(inp_t, win_harm=wh, win_perc=wp, power=kw.get("power", 2.0), margin_harm=mh, margin_perc=mp, want_mask=kw.get("mask", False))
            for g, e in zip(got, exp):
                e = np.swapaxes(e, -1, -2).astype(g.dtype)
                if np.iscomplexobj(inp) or kw.get("power", 2.0) == 3.5:
                    assert np.abs(g - e).max() <= eps * max(np.abs(e).max(), 1e-30), kw
                else:
                    assert np.array_equal(g, e), kw


def test_magphase_through_simulator(monkeypatch):
    """librosa_amd.magphase (core/spectrum.py:1296-1361) with the kernel body on the host: the reference's own test cases (tests/test_core.py:756-804:
    dtypes, zeros -> phase 1 + 0j, denormals, real input with signed zeros) and random spectra against the oracle -- real input bit for bit, |D| of a complex
    value to the last bit (an exactly rounded sum / this libm's hypot here, NumPy's hypot there), every NumPy fast-path exponent."""
    import torch

    import librosa_amd
    from librosa_amd import _arrays

    real_session = _arrays.Session
    monkeypatch.setattr(_arrays, "Session", _SimSession)
    rng = np.random.default_rng(12)
    D = (rng.standard_normal((3, 17, 9)) + 1j * rng.standard_normal((3, 17, 9))).astype(np.complex64)
    D[0, :3] = 0
    S, P = librosa_amd.magphase(D)
    eS, eP = O.magphase(D)
    assert S.dtype == np.float32 and P.dtype == np.complex64 and S.shape == D.shape
    assert np.abs(S - eS).max() <= 1e-7 * np.abs(eS).max() and np.abs(P - eP).max() <= 2e-7 and np.allclose(np.abs(P), 1.0) and np.allclose(S * P, D, atol=1e-6)
    assert np.all(P[0, :3] == 1 + 0j) and np.all(S[0, :3] == 0)
    D64 = D.astype(np.complex128)
    for power in (1, 2, 0.5, -1, 0, 3.5):
        with np.errstate(divide="ignore"):
            got, exp = librosa_amd.magphase(D64, power=power), O.magphase(D64, power=power)
        assert got[0].dtype == np.float64 and got[1].dtype == np.complex128
        assert np.abs(got[1] - exp[1]).max() <= 5e-16 and np.allclose(got[0], exp[0], rtol=1e-14 if power == 3.5 else 2e-15, atol=0), power   # (|D|: this libm's hypot / NumPy's)
    R = np.array([[-1.0, -0.0], [0.0, 1.0]], dtype=np.float64)                       # test_magphase_real
    S, P = librosa_amd.magphase(R)
    assert S.dtype == np.float64 and P.dtype == np.complex128 and np.array_equal(S, [[1.0, 0.0], [0.0, 1.0]]) and np.array_equal(P, [[-1, 1], [1, 1]])
    assert all(np.array_equal(g, e) for g, e in zip(librosa_amd.magphase(R.astype(np.float32), power=2), O.magphase(R.astype(np.float32), power=2)))
    Dn = 1.0e-42j * np.ones((4, 4), dtype=np.complex64)                              # test_magphase_denormalized
    S, P = librosa_amd.magphase(Dn)
    assert np.allclose(S, 1.0e-42) and np.allclose(P, 0 + 1j)
    Z = np.zeros((5, 5), dtype=np.complex64)                                          # test_magphase_zero
    S, P = librosa_amd.magphase(Z)
    assert np.all(S == 0) and np.all(P == 1 + 0j)
    assert librosa_amd.magphase(np.zeros((0, 4), dtype=np.complex64))[1].shape == (0, 4)
    monkeypatch.setattr(_arrays, "Session", _sim_torch_session(real_session))         # tensor code path (CPU tensors)
    St, Pt = librosa_amd.magphase(torch.from_numpy(D64), power=2)
    assert isinstance(St, torch.Tensor) and np.allclose(St.numpy(), O.magphase(D64, power=2)[0], rtol=1e-15, atol=0) and np.abs(Pt.numpy() - O.magphase(D64)[1]).max() <= 5e-16


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_hpss_tile_kernel_windows(dtype):
    """The 4 x 4-tile kernel's corners (lra_hpss.h, hpss_median_quad): shortest and longest windows of both network sizes, even windows,
    axes of exactly win + 4 positions (every tile folds at both ends), axes that are not multiples of four, values with many ties --
    real input, so bit for bit against the oracle."""
    rng = np.random.default_rng(77)
    cases = [(35, 35, 31, 31), (10, 11, 6, 7), (12, 37, 8, 33), (37, 10, 33, 6), (41, 23, 6, 18), (69, 12, 65, 7), (13, 70, 7, 65), (50, 45, 32, 33), (39, 38, 17, 31), (7, 9, 3, 3), (30, 30, 5, 9)]
    for n_frames, n_bins, wh, wp in cases:
        for ties in (False, True):
            S = rng.random((2, n_bins, n_frames)).astype(dtype)
            if ties:
                S = np.round(S * 6).astype(dtype) / 4
            St = np.ascontiguousarray(np.swapaxes(S, -1, -2))
            exp = O.hpss(S, kernel_size=(wh, wp), mask=True, power=1.0)
            got = H.hpss(St, win_harm=wh, win_perc=wp, power=1.0, margin_harm=1.0, margin_perc=1.0, want_mask=True)
            for g, e in zip(got, exp):
                assert np.array_equal(g, np.swapaxes(e, -1, -2).astype(g.dtype)), (n_frames, n_bins, wh, wp, ties)


def test_hpss_shims_through_simulator(monkeypatch):
    """librosa_amd.decompose.hpss (layouts, dtypes, NumPy and tensor code paths, argument errors) and librosa_amd.effects.hpss / harmonic
    / percussive (with the two transforms supplied by the oracle) against the oracle's chain."""
    import torch

    import golden_cases
    import librosa_amd
    from librosa_amd import _arrays, effects
    from librosa_amd.core import spectrum

    real_session = _arrays.Session
    monkeypatch.setattr(_arrays, "Session", _SimSession)
    y = golden_cases.make_signal("mix", 5000, 9, (2,), "float32")
    D = O.stft(y, n_fft=256, hop_length=64)                        # (2, 129, 79)

    def close(got, exp, tol=1e-6):
        return all(g.shape == e.shape and g.dtype == e.dtype and np.abs(g.astype(np.complex128) - e).max() <= tol * max(np.abs(e).max(), 1e-30) for g, e in zip(got, exp))

    for kw in (dict(), dict(kernel_size=(9, 17), margin=(1.0, 2.0)), dict(mask=True, power=1.0), dict(mask=True, power=np.inf), dict(kernel_size=40)):
        assert close(librosa_amd.decompose.hpss(D, **kw), O.hpss(D, **kw)), kw
        assert close(librosa_amd.decompose.hpss(D[0], **kw), O.hpss(D[0], **kw)), kw
        P = np.abs(D) ** 2
        got, exp = librosa_amd.decompose.hpss(P, **kw), O.hpss(P, **kw)
        assert all(np.array_equal(g, e) and g.dtype == e.dtype for g, e in zip(got, exp)), kw
    assert close(librosa_amd.decompose.hpss(D.astype(np.complex128)), O.hpss(D.astype(np.complex128)), 1e-14)
    for bad in (dict(margin=0.5), dict(margin=(1.0, 0.9)), dict(power=0), dict(kernel_size=0)):
        with pytest.raises(librosa_amd.ParameterError):
            librosa_amd.decompose.hpss(D, **bad)
    with pytest.raises(librosa_amd.ParameterError):
        librosa_amd.decompose.hpss(-np.abs(D))
    # tensor code path (CPU tensors; a transposed view of a frame-major buffer, like librosa_amd.stft's result, needs no transpose)
    host = librosa_amd.decompose.hpss(D, margin=(1.0, 2.0))
    monkeypatch.setattr(_arrays, "Session", _sim_torch_session(real_session))
    Dt = torch.from_numpy(np.ascontiguousarray(np.swapaxes(D, -1, -2))).transpose(-1, -2)
    dev = librosa_amd.decompose.hpss(Dt, margin=(1.0, 2.0))
    assert all(isinstance(d, torch.Tensor) and d.dtype == torch.complex64 and np.array_equal(d.numpy(), h) for d, h in zip(dev, host))
    dev = librosa_amd.decompose.hpss(torch.from_numpy(np.abs(D)), mask=True)
    assert all(np.array_equal(d.numpy(), h) for d, h in zip(dev, O.hpss(np.abs(D), mask=True)))
    # effects: the shim's own chaining (which istft arguments, which component), transforms by the oracle
    monkeypatch.setattr(_arrays, "Session", _SimSession)
    monkeypatch.setattr(effects, "_stage", lambda a: (a, False))
    monkeypatch.setattr(spectrum, "stft", lambda a, check_finite=True, row_align=None, **kw: O.stft(a, **kw))
    monkeypatch.setattr(spectrum, "istft", lambda a, **kw: O.istft(np.ascontiguousarray(a), **kw))
    for kw in (dict(n_fft=512), dict(n_fft=256, hop_length=64, margin=(1.0, 3.0), kernel_size=(9, 17)), dict(n_fft=256, window="hamming", win_length=200)):
        eh, ep = O.effects_hpss(y, **kw)
        gh, gp = effects.hpss(y, **kw)
        scale = np.abs(y).max()
        assert gh.shape == y.shape and gh.dtype == y.dtype and np.abs(gh - eh).max() <= 1e-5 * scale and np.abs(gp - ep).max() <= 1e-5 * scale, kw
        assert np.array_equal(effects.harmonic(y, **kw), gh) and np.array_equal(effects.percussive(y, **kw), gp)
    # pitch_shift: time_stretch (oracle's here) + the shim's own resample / crop chaining
    monkeypatch.setattr(effects, "time_stretch", lambda a, rate, **kw: O.time_stretch(a, rate=rate, **kw))
    import warnings

    for kw in (dict(n_steps=4, res_type="fft"), dict(n_steps=-5, bins_per_octave=24, res_type="scipy", scale=True, n_fft=512), dict(n_steps=2, res_type="kaiser_best")):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            exp = O.pitch_shift(y, sr=22050, **{**kw, "res_type": "fft" if kw["res_type"] == "kaiser_best" else kw["res_type"]})
            got = effects.pitch_shift(y, sr=22050, **kw)
        assert got.shape == y.shape and got.dtype == y.dtype and np.abs(got - exp).max() <= 1e-6 * np.abs(y).max(), kw
    with pytest.raises(librosa_amd.ParameterError):
        effects.pitch_shift(y, sr=22050, n_steps=1, res_type="polyphase")
    with pytest.raises(librosa_amd.ParameterError):
        effects.pitch_shift(y, sr=22050, n_steps=1, bins_per_octave=1.5)


# ---- mixed-radix fused forward kernel (csrc/lra_mixed.h): the whole __global__ body on host threads against the oracle ---------------------------
@pytest.mark.parametrize(
    "n_fft,hop,n,center,pad_mode,dtype",
    [
        (400, 160, 4000, True, "constant", np.float32),    # 8 x 5 x 5: the 25 ms / 10 ms frames of 16 kHz front ends
        (400, 160, 1900, True, "reflect", np.float32),
        (400, 100, 2100, False, "constant", np.float32),
        (240, 60, 1500, True, "edge", np.float32),         # 8 x 5 x 3
        (160, 80, 1000, True, "symmetric", np.float32),    # 8 x 2 x 5
        (480, 120, 2500, True, "constant", np.float64),    # 8 x 2 x 5 x 3, float64
        (1000, 250, 5000, True, "reflect", np.float32),    # 4 x 5 x 5 x 5
        (1200, 300, 6100, True, "constant", np.float32),   # 8 x 5 x 5 x 3: several frames per workgroup, a partial last group
        (1280, 320, 5000, True, "constant", np.float32),   # 8 x 8 x 2 x 5
        (882, 441, 4500, True, "reflect", np.float32),     # 3 x 3 x 7 x 7: 20 ms / 10 ms at 44.1 kHz (odd M, odd padding: no 8-byte sample pairs)
        (882, 220, 3000, False, "constant", np.float64),
    ],
)
def test_mixed_radix_stft_body(n_fft, hop, n, center, pad_mode, dtype):
    """librosa/core/spectrum.py:57-391 for frame lengths 2^a 3^b 5^c through ONE fused launch: complex spectrum and |X|^2 against the oracle
    (NaN-prefilled outputs: every element must be stored)."""
    rng = np.random.default_rng(n_fft + hop)
    y = rng.standard_normal((2, n)).astype(dtype)
    win = O.get_window("hann", n_fft).astype(dtype)
    ref = O.stft(y, n_fft=n_fft, hop_length=hop, center=center, pad_mode=pad_mode)
    D = H.mixed_stft(y, n_fft, hop, win, mode="stft", center=center, pad_mode=pad_mode)
    got = np.moveaxis(D, -1, -2)
    assert got.shape == ref.shape and np.isfinite(D.view(dtype)).all()
    scale = np.abs(ref).max()
    tol = 1e-12 if dtype == np.float64 else 2e-6
    assert np.abs(got - ref).max() <= tol * scale, np.abs(got - ref).max() / scale
    assert np.all(got[:, 0].imag == 0) and np.all(got[:, -1].imag == 0)  # DC and Nyquist bins are exactly real, as pocketfft's are
    S = np.moveaxis(H.mixed_stft(y, n_fft, hop, win, mode="power", center=center, pad_mode=pad_mode, power=2.0), -1, -2)
    Sref = np.abs(ref) ** 2
    assert np.all(np.abs(S - Sref) <= (1e-11 if dtype == np.float64 else 1e-5) * Sref.max())
    S1 = np.moveaxis(H.mixed_stft(y[:1], n_fft, hop, win, mode="power", center=center, pad_mode=pad_mode, power=1.0), -1, -2)
    assert np.all(np.abs(S1 - np.abs(ref[:1])) <= (1e-11 if dtype == np.float64 else 1e-5) * np.abs(ref).max())


@pytest.mark.parametrize("n_fft,hop,n_mels,sr,dtype", [(400, 160, 80, 16000, np.float32), (1200, 300, 64, 48000, np.float32), (240, 120, 20, 8000, np.float64), (882, 441, 64, 44100, np.float32)])
def test_mixed_radix_mel_body(n_fft, hop, n_mels, sr, dtype):
    """librosa/feature/spectral.py:2022-2161 through the same launch (banded basis from LDS power rows); the reference's float32 basis values."""
    rng = np.random.default_rng(n_mels)
    y = (0.1 * rng.standard_normal((2, 7 * n_fft + 13))).astype(dtype)
    win = O.get_window("hann", n_fft).astype(dtype)
    B = O.mel(sr=sr, n_fft=n_fft, n_mels=n_mels).astype(dtype)
    ref = O.melspectrogram(y=y, sr=sr, n_fft=n_fft, hop_length=hop, n_mels=n_mels)
    got = H.mixed_stft(y, n_fft, hop, win, mode="mel", mel_basis=B)
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert np.all(np.abs(got - ref) <= (1e-11 if dtype == np.float64 else 1e-4) * np.abs(ref) + (1e-12 if dtype == np.float64 else 1e-6) * ref.max())


@pytest.mark.parametrize("n_fft,frames,dtype", [(400, 13, np.float32), (882, 5, np.float32), (1200, 7, np.float64), (1280, 3, np.float32)])
def test_mixed_radix_irfft_body(n_fft, frames, dtype):
    """The inverse real transform alone (mixed_irfft_kernel, round 6: listed lengths too long for the gather kernel; replaces spec_pack + rocFFT C2R of the general
    inverse path, librosa/core/spectrum.py:598): n_fft * irfft of every frame, the imaginary parts of the DC / Nyquist bins ignored as pocketfft's c2r does,
    NaN-prefilled output, partial last group."""
    rng = np.random.default_rng(n_fft)
    ct = np.complex128 if dtype == np.float64 else np.complex64
    D = (rng.standard_normal((2, frames, n_fft // 2 + 1)) + 1j * rng.standard_normal((2, frames, n_fft // 2 + 1))).astype(ct)
    got = H.mixed_irfft(D, n_fft)
    ref = np.fft.irfft(D.astype(np.complex128), n=n_fft, axis=-1) * n_fft
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert np.abs(got - ref).max() <= (1e-11 if dtype == np.float64 else 3e-6) * np.abs(ref).max()


@pytest.mark.parametrize(
    "n_fft,hop,n,center,length,dtype",
    [
        (400, 160, 4000, True, "n", np.float32),
        (400, 160, 4000, True, None, np.float32),
        (400, 100, 3000, True, 2500, np.float32),
        (400, 160, 3000, True, 4000, np.float32),     # length beyond the frames: zero tail from the wrapper
        (400, 160, 3000, False, None, np.float32),
        (240, 60, 2000, True, "n", np.float32),
        (160, 200, 3000, True, None, np.float32),     # hop > n_fft: gaps between frames
        (480, 240, 3000, True, "n", np.float64),      # (larger frames hold three per workgroup: the fused form needs hop >= n_fft / 2 there)
        (1000, 500, 9000, True, "n", np.float32),     # several groups per clip with a halo frame
        (1200, 600, 7000, True, "n", np.float32),
        (882, 441, 6000, True, "n", np.float32),      # radix 7
        (512, 160, 5000, True, "n", np.float32),      # powers of two with a hop outside n_fft / {2, 4, 8, 16}: the same kernel (LRA_MIXED_INV_POW2)
        (512, 200, 4100, False, 4000, np.float64),
        (256, 100, 3000, True, 3300, np.float32),
        (1024, 441, 9000, True, "n", np.float32),
    ],
)
def test_mixed_radix_istft_body(n_fft, hop, n, center, length, dtype):
    """librosa/core/spectrum.py:394-626 for frame lengths 2^a 3^b 5^c through ONE fused launch (un-split, inverse passes, window, gather overlap-add in frame
    order, normalisation), NaN-prefilled output."""
    rng = np.random.default_rng(n_fft + hop)
    y = rng.standard_normal((2, n)).astype(dtype)
    L_ = n if length == "n" else length
    Dn, ref, wss, win, out_len, n_used = _istft_inputs(y, n_fft, hop, center, L_)
    out = H.mixed_istft(Dn, n_fft, hop, win, wss, out_len, n_used, center=center)
    assert out.shape == ref.shape and np.isfinite(out).all()
    cond = 1.0 / np.sqrt(np.maximum(wss, np.finfo(np.float32).tiny))
    tol = (4e-6 if dtype == np.float32 else 1e-13) * np.abs(ref).max() * np.maximum(1.0, cond)
    well = wss > 1e-3 * wss.max()
    assert np.all(np.abs(out - ref)[..., well] <= tol[well])
    assert np.all(np.abs(out - ref)[..., ~well] <= 50 * tol[~well] + 1e-3)


@pytest.mark.filterwarnings("ignore:n_fft=")
def test_cqt_fused_octaves_through_simulator(monkeypatch):
    """The one-launch octave (csrc/lra_mixed.h, mixed_cqt_kernel: frames, rectangular-window transform, sparse projection, scaling, stacking) run whole
    by the simulator inside librosa_amd.cqt / vqt: the oracle's transform to float32 round-off (2e-5 of the peak, the device tests' bar), 1e-11 in float64,
    for every octave schedule of the bit-for-bit test above (early downsampling, partial lowest octave, reflect padding, unscaled, one bin)."""
    import cqt_oracle as CQ
    import golden_cases
    import librosa_amd
    from librosa_amd import _arrays

    monkeypatch.setattr(_arrays, "Session", _SimSession)
    monkeypatch.setattr(_SimCtx, "fused_octaves", True)
    calls = []
    orig = _SimCtx.cqt_octave_exec
    monkeypatch.setattr(_SimCtx, "cqt_octave_exec", lambda self, *a: (calls.append(a[4]), orig(self, *a))[1])
    y = golden_cases.make_signal("mix", 22050, 3, None, "float32")
    for kw in (dict(), dict(hop_length=256, n_bins=60), dict(n_bins=30, bins_per_octave=12), dict(scale=False, n_bins=24), dict(fmin=110.0, n_bins=36, tuning=0.2, norm=2),
               dict(filter_scale=0.5, pad_mode="reflect", window="hamming", sparsity=0.05), dict(n_bins=1)):
        got = librosa_amd.cqt(y, res_type="polyphase", **kw)
        exp = CQ.cqt(y, res_type="polyphase", **kw)
        assert got.shape == exp.shape and got.dtype == exp.dtype
        assert np.abs(got - exp).max() <= 2e-5 * np.abs(exp).max(), (kw, np.abs(got - exp).max() / np.abs(exp).max())
    assert calls and set(calls) <= {32, 64, 128, 256, 512, 1024, 2048}
    y64 = golden_cases.make_signal("mix", 12000, 5, (2,), "float64")
    got = librosa_amd.vqt(y64, res_type="polyphase", gamma=5.0, n_bins=36)
    exp = CQ.vqt(y64, res_type="polyphase", gamma=5.0, n_bins=36)
    assert got.dtype == np.complex128 and np.abs(got - exp).max() <= 1e-11 * np.abs(exp).max()
