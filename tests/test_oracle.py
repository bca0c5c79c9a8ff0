"""Pins the CPU oracle (oracle/stft_oracle.py): reference KATs, committed reference outputs, and
(when /root/reference exists) the live reference.  CPU only."""
import glob
import json
import os
import warnings

import numpy as np
import pytest

import golden_cases
import ref_shim
import stft_oracle as O

from conftest import GOLDEN_DIR


# ---- reference known-answer vectors: tests/test_filters.py:35-98 (reference repo) -------------
def test_hz_to_mel_kat_slaney():
    freqs = np.array([0, 500, 1000, 2000, 3000])
    mels = np.array([0.0, 7.5, 15.0, 25.08188016, 30.97940199])
    assert np.allclose(O.hz_to_mel(freqs), mels)
    for f, m in zip(freqs, mels):
        assert np.isclose(O.hz_to_mel(f), m)


def test_hz_to_mel_kat_htk():
    # tests/test_filters.py:35-46 (reference repo)
    freqs = np.array([0, 500, 1000, 2000, 3000])
    mels = np.array([0.0, 607.44591966, 999.98553714, 1521.35955416, 1876.45406012])
    assert np.allclose(O.hz_to_mel(freqs, htk=True), mels)
    for f, m in zip(freqs, mels):
        assert np.isclose(O.hz_to_mel(f, htk=True), m)


def test_mel_to_hz_kat():
    # tests/test_filters.py:63-98 (reference repo)
    mels = np.array([0.0, 200, 400, 600, 800, 1000, 1200, 1400, 1600, 1800])
    freqs = np.array([0.0, 135.92888249, 298.25299511, 492.09787234, 723.58434605, 1000.02181646,
                      1330.13905319, 1724.35981432, 2195.13198618, 2757.32063694])
    assert np.allclose(O.mel_to_hz(mels, htk=True), freqs)
    for f, m in zip(freqs, mels):
        assert np.isclose(O.mel_to_hz(m, htk=True), f)
    mels = np.array([0, 5, 10, 15, 25, 30])
    freqs = np.array([0.0, 333.33333333, 666.66666667, 1000.0, 1988.77281813, 2804.64413074])
    assert np.allclose(O.mel_to_hz(mels, htk=False), freqs)
    for f, m in zip(freqs, mels):
        assert np.isclose(O.mel_to_hz(m, htk=False), f)


def test_mel_to_hz_inverse():
    f = np.array([0.0, 220.0, 999.0, 1000.0, 4000.0, 11025.0])
    for htk in (False, True):
        np.testing.assert_allclose(O.mel_to_hz(O.hz_to_mel(f, htk=htk), htk=htk), f, rtol=1e-10, atol=1e-9)


def test_fft_frequencies():
    # tests/test_convert.py:314-325: DC = 0, Nyquist = sr/2, linear
    f = O.fft_frequencies(sr=22050, n_fft=2048)
    assert f[0] == 0 and f[-1] == 11025.0 and len(f) == 1025
    np.testing.assert_allclose(np.diff(f), 22050 / 2048)


def test_mel_basis_properties():
    # tests/test_filters.py:120-189: shape, non-negativity, slaney area normalisation
    B = O.mel(sr=22050, n_fft=2048, n_mels=128)
    assert B.shape == (128, 1025) and B.dtype == np.float32 and (B >= 0).all()
    assert abs(float(B[0, 1]) - 0.016182853) < 1e-9  # filters.py:185 docstring value
    assert np.count_nonzero(B) == 2018
    assert (np.count_nonzero(B, axis=0) <= 2).all()
    mel_f = O.mel_frequencies(130, fmin=0.0, fmax=11025.0)
    df = 22050 / 2048
    assert np.all(np.abs(B.sum(axis=1) * df - 1) < 5e-2)
    with pytest.raises(O.ParameterError):
        O.mel(sr=22050, n_fft=2048, norm="bogus")


def test_window_matches_scipy_and_sumsquare_interior():
    w = O.get_window("hann", 2048)
    n = np.arange(2048)
    np.testing.assert_allclose(w, 0.5 - 0.5 * np.cos(2 * np.pi * n / 2048), atol=3e-16)
    wss = O.window_sumsquare(window="hann", n_frames=50, hop_length=512, n_fft=2048)
    np.testing.assert_allclose(wss[2048:-2048], 1.5, rtol=1e-6)
    assert wss.dtype == np.float32 and len(wss) == 2048 + 512 * 49


# ---- the reference's own definition of the STFT: tests/test_core.py:256-292 ---------------------
@pytest.mark.parametrize("n_fft", [256, 501])
@pytest.mark.parametrize("window", ["hann", "ones"])
@pytest.mark.parametrize("hop", [None, 128])
@pytest.mark.parametrize("center", [False, True])
def test_stft_is_rfft_of_frames(n_fft, window, hop, center):
    import scipy.fft
    import scipy.signal
    y = golden_cases.make_signal("chirp", 22050, 0)
    D = O.stft(y, n_fft=n_fft, hop_length=hop, window=window, center=center)
    h = hop or n_fft // 4
    if center:
        assert D.shape == (1 + n_fft // 2, 1 + len(y) // h)
        yp = np.pad(y, n_fft // 2)
    else:
        assert D.shape == (1 + n_fft // 2, 1 + (len(y) - n_fft) // h)
        yp = y
    w = scipy.signal.get_window(window, n_fft, fftbins=True)
    for t in (0, 1, D.shape[1] // 2, D.shape[1] - 1):
        ref = scipy.fft.rfft(w * yp[t * h : t * h + n_fft])
        assert np.allclose(D[:, t], ref, rtol=1e-5, atol=1e-6)


def test_stft_errors():
    y = np.zeros(100, dtype=np.float32)
    with pytest.raises(O.ParameterError):
        O.stft(y, n_fft=256, center=False)
    with pytest.raises(O.ParameterError):
        O.stft(np.zeros(1000, np.float32), n_fft=256, pad_mode="wrap")
    with pytest.raises(O.ParameterError):
        O.stft(np.zeros(1000, np.float32), n_fft=256, hop_length=0)


# ---- committed reference outputs ---------------------------------------------------------------
def _cases():
    return sorted(golden_cases.CASES)


@pytest.mark.parametrize("name", _cases())
def test_oracle_vs_golden(name):
    case = golden_cases.CASES[name]
    g = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    y = g["y"]
    # the seeded generator must reproduce the stored input bit-for-bit
    kind, n, seed, channels, dtype = case["signal"]
    assert np.array_equal(golden_cases.make_signal(kind, n, seed, channels, dtype), y)
    skw = dict(case["stft"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        D = O.stft(y, **skw)
    assert D.shape == g["D"].shape and D.dtype == g["D"].dtype
    scale = np.abs(g["D"]).max()
    assert np.abs(D - g["D"]).max() <= 1e-6 * scale
    if case["mel"] is not None:
        power, fkw = golden_cases.split_mel_kwargs(case["mel"])
        mkw = dict(skw)
        mkw.setdefault("hop_length", int(skw.get("win_length", skw["n_fft"]) // 4))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            M = O.melspectrogram(y=y, sr=golden_cases.SR, power=power, **mkw, **fkw)
        assert M.shape == g["mel"].shape and M.dtype == g["mel"].dtype
        np.testing.assert_allclose(M, g["mel"], rtol=1e-5, atol=1e-6 * g["mel"].max())
        B = O.mel(sr=golden_cases.SR, n_fft=skw["n_fft"], **fkw)
        assert np.array_equal(B, g["mel_basis"])
    if case["istft"]:
        ikw = {k: v for k, v in skw.items() if k in ("hop_length", "win_length", "n_fft", "window", "center")}
        y1 = O.istft(g["D"], length=y.shape[-1], **ikw)
        y2 = O.istft(g["D"], **ikw)
        assert y1.shape == g["y_istft_len"].shape and y2.shape == g["y_istft_nolen"].shape
        np.testing.assert_allclose(y1, g["y_istft_len"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(y2, g["y_istft_nolen"], rtol=0, atol=2e-6)


def test_oracle_vs_golden_config1():
    g = np.load(os.path.join(GOLDEN_DIR, "config1_sine10s.npz"))
    y = O.config1_input()
    D = O.stft(y, n_fft=2048, hop_length=512)
    assert D.shape == (1025, 431) and D.dtype == np.complex64
    assert np.abs(D[:, g["frames"]] - g["D_frames"]).max() <= 1e-6 * g["D_absmax"]
    M = O.melspectrogram(y=y, sr=22050, n_fft=2048, hop_length=512, n_mels=128)
    np.testing.assert_allclose(M, g["mel"], rtol=1e-5, atol=1e-6 * g["mel"].max())
    yh = O.istft(D, hop_length=512, length=len(y))
    np.testing.assert_allclose(yh[:4096], g["y_istft_head"], atol=2e-6)
    np.testing.assert_allclose(yh[-4096:], g["y_istft_tail"], atol=2e-6)
    snr = 10 * np.log10(np.sum(y.astype(np.float64) ** 2) / np.sum((y - yh).astype(np.float64) ** 2))
    assert snr > 100


def test_oracle_vs_golden_config2():
    g = np.load(os.path.join(GOLDEN_DIR, "config2_clips.npz"))
    for i in (0, 37):
        y = O.config_input(1, first_clip=i)[0]
        assert np.array_equal(y[:2048], g[f"y_head_{i}"])
        D = O.stft(y, n_fft=2048, hop_length=512)
        assert D.shape == (1025, 1292)
        assert np.abs(D[:, g["frames"]] - g[f"D_frames_{i}"]).max() <= 1e-6 * g[f"D_absmax_{i}"]
        M = O.melspectrogram(y=y, sr=22050, n_fft=2048, hop_length=512, n_mels=128)
        assert M.shape == (128, 1292)
        if i == 0:
            np.testing.assert_allclose(M, g["mel_0"], rtol=1e-5, atol=1e-6 * g["mel_0"].max())
        else:
            np.testing.assert_allclose(M[:, g["frames"]], g[f"mel_frames_{i}"], rtol=1e-5, atol=1e-6 * g[f"mel_absmax_{i}"])


def test_config_input_is_shard_invariant():
    a = O.config_input(4, n=4096)
    b = np.concatenate([O.config_input(2, n=4096, first_clip=0), O.config_input(2, n=4096, first_clip=2)])
    assert np.array_equal(a, b)


# ---- live reference (build container only) ------------------------------------------------------
@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_vs_live_reference(seed):
    librosa = ref_shim.load_reference()
    rng = np.random.default_rng(seed)
    n_fft = int(rng.choice([256, 400, 512, 1024, 1025, 2048]))
    hop = int(rng.integers(n_fft // 8, n_fft // 2))
    n = int(rng.integers(2 * n_fft, 6 * n_fft))
    y = rng.standard_normal((2, n)).astype(np.float32)
    pad_mode = str(rng.choice(["constant", "reflect", "edge"]))
    D0 = librosa.stft(y, n_fft=n_fft, hop_length=hop, pad_mode=pad_mode)
    D1 = O.stft(y, n_fft=n_fft, hop_length=hop, pad_mode=pad_mode)
    assert np.abs(D0 - D1).max() <= 1e-6 * np.abs(D0).max()
    # (lengths that cut whole frames off the end make the reference itself raise: its
    #  __overlap_add, core/spectrum.py:640-643, gets a negative N; not a defined behaviour)
    for length in (None, n, n - 37, n + 100):
        y0 = librosa.istft(D0, hop_length=hop, n_fft=n_fft, length=length)
        y1 = O.istft(D0, hop_length=hop, n_fft=n_fft, length=length)
        assert y0.shape == y1.shape
        np.testing.assert_allclose(y0, y1, atol=2e-6)
    M0 = librosa.feature.melspectrogram(y=y, sr=22050, n_fft=n_fft, hop_length=hop, n_mels=40, pad_mode=pad_mode)
    M1 = O.melspectrogram(y=y, sr=22050, n_fft=n_fft, hop_length=hop, n_mels=40, pad_mode=pad_mode)
    np.testing.assert_allclose(M0, M1, rtol=1e-5, atol=1e-6 * M0.max())
    wss0 = librosa.filters.window_sumsquare(window="hann", n_frames=17, hop_length=hop, n_fft=n_fft)
    assert np.array_equal(wss0, O.window_sumsquare(window="hann", n_frames=17, hop_length=hop, n_fft=n_fft))


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("seed", range(100, 116))
def test_oracle_vs_live_reference_wide(seed):
    """The same comparison over the rest of the argument space the path takes: float64 audio, uncentred framing, win_length < n_fft, other windows
    (names, tuples, arrays), np.pad's symmetric mode, power 1 / 2 / 1.5, HTK mel scale, norm=None, fmin / fmax, 1-D and 3-D input."""
    librosa = ref_shim.load_reference()
    rng = np.random.default_rng(seed)
    n_fft = int(rng.choice([64, 200, 256, 512, 1000, 1024, 2048, 4096]))
    hop = int(rng.choice([n_fft // 4, n_fft // 2, n_fft // 3, n_fft, max(1, n_fft // 8), n_fft + 17]))
    dtype = np.float64 if rng.random() < 0.3 else np.float32
    center = bool(rng.random() < 0.7)
    pad_mode = str(rng.choice(["constant", "reflect", "edge", "symmetric"]))
    win_length = None if rng.random() < 0.6 else int(rng.integers(n_fft // 2, n_fft))
    wl = win_length or n_fft
    window = [lambda: "hann", lambda: "hamming", lambda: "blackmanharris", lambda: ("tukey", 0.25), lambda: ("kaiser", 4.0), lambda: rng.random(wl) + 0.1][int(rng.integers(0, 6))]()
    n = int(rng.integers(max(n_fft, 2 * hop) + 5, 5 * n_fft + 3000))
    shape = [(n,), (2, n), (2, 3, n)][int(rng.integers(0, 3))]
    y = rng.standard_normal(shape).astype(dtype)
    kw = dict(n_fft=n_fft, hop_length=hop, win_length=win_length, window=window, center=center, pad_mode=pad_mode)
    tol = 1e-6 if dtype == np.float32 else 1e-13
    D0, D1 = librosa.stft(y, **kw), O.stft(y, **kw)
    assert D0.dtype == D1.dtype and D0.shape == D1.shape
    assert np.abs(D0 - D1).max() <= tol * np.abs(D0).max()
    ikw = dict(hop_length=hop, win_length=win_length, n_fft=n_fft, window=window, center=center)
    try:
        y0 = librosa.istft(D0, **ikw)
    except Exception:
        y0 = None  # the reference rejects the combination
    if y0 is not None:
        y1 = O.istft(D0, **ikw)
        assert y0.shape == y1.shape and y0.dtype == y1.dtype
        # (hops beyond the window leave samples with a tiny window sum-square: compare where the division is well conditioned)
        wss = librosa.filters.window_sumsquare(window=window, n_frames=D0.shape[-1], win_length=win_length, n_fft=n_fft, hop_length=hop, dtype=np.float64)
        wss = wss[(n_fft // 2 if center else 0):]
        wss = np.pad(wss, (0, max(0, y0.shape[-1] - len(wss))))[: y0.shape[-1]]
        ok = wss > 1e-3 * wss.max()
        assert np.abs(y0 - y1)[..., ok].max() <= (2e-5 if dtype == np.float32 else 1e-12) * max(np.abs(y0)[..., ok].max(), 1e-30)
    if n_fft >= 200:
        mk = dict(kw, sr=22050, n_mels=int(rng.choice([13, 40, 80, 128])), power=float(rng.choice([1.0, 2.0, 1.5])), htk=bool(rng.random() < 0.3),
                  norm=[None, "slaney", 1, np.inf][int(rng.integers(0, 4))], fmin=float(rng.choice([0.0, 50.0])), fmax=[None, 8000.0][int(rng.integers(0, 2))])
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # (empty filters at small n_fft / many bands: both sides warn)
            M0, M1 = librosa.feature.melspectrogram(y=y, **mk), O.melspectrogram(y=y, **mk)
        assert M0.dtype == M1.dtype and M0.shape == M1.shape
        np.testing.assert_allclose(M0, M1, rtol=2e-5 if dtype == np.float32 else 1e-11, atol=(1e-6 if dtype == np.float32 else 1e-13) * M0.max())


def test_db_and_mfcc_oracle_matches_reference_goldens():
    """the restated power_to_db / amplitude_to_db / db_to_* / mfcc against outputs of the unmodified
    reference (tests/golden/db_mfcc.npz, oracle/make_golden.py::make_db_mfcc)."""
    g = np.load(os.path.join(GOLDEN_DIR, "db_mfcc.npz"))
    M, A, y, y2 = g["M"], g["A"], g["y"], g["y2"]
    assert np.array_equal(O.power_to_db(M), g["db_default"])
    assert np.array_equal(O.power_to_db(M, ref=np.max), g["db_refmax"])
    assert np.array_equal(O.power_to_db(M, top_db=None, amin=1e-6, ref=2.5), g["db_notop"])
    assert np.array_equal(O.power_to_db(M, ref=np.median, top_db=30.0), g["db_median_top30"])
    assert np.array_equal(O.power_to_db(M, ref=np.max, axes=-1, top_db=40.0), g["db_axes_last"])
    assert np.array_equal(O.power_to_db(M, ref=np.max, axes=None), g["db_axes_none"])
    assert np.array_equal(O.amplitude_to_db(A), g["adb_default"])
    assert np.array_equal(O.amplitude_to_db(A, ref=np.max, top_db=60.0), g["adb_refmax"])
    assert np.array_equal(O.db_to_power(g["db_notop"], ref=2.5), g["pow_back"])
    assert np.array_equal(O.mfcc(S=g["db_default"], n_mfcc=13), g["mfcc_S"])
    assert np.array_equal(O.mfcc(S=g["db_default"], n_mfcc=20, dct_type=3, lifter=22), g["mfcc_S_t3_lift"])
    assert np.array_equal(O.mfcc(S=g["db_default"], n_mfcc=12, dct_type=1, norm=None), g["mfcc_S_t1_none"])
    assert np.array_equal(O.mfcc(y=y, sr=22050, n_mfcc=20, n_fft=1024, hop_length=256, n_mels=40), g["mfcc_y"])
    assert np.array_equal(O.mfcc(y=y2, sr=22050), g["mfcc_y2_default"])
    with pytest.raises(O.ParameterError):
        O.power_to_db(M, amin=0)
    with pytest.raises(O.ParameterError):
        O.mfcc(S=g["db_default"], lifter=-1)


# ---- Griffin-Lim -------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(golden_cases.GRIFFINLIM_CASES))
def test_griffinlim_oracle_matches_reference_golden(name):
    """The restatement of librosa/core/spectrum.py:2816-2917 reproduces the reference's output bit for bit (same rng
    stream, same operation order) on the committed fixtures (oracle/make_golden.py::make_griffinlim)."""
    g = np.load(os.path.join(GOLDEN_DIR, "griffinlim.npz"))
    case = golden_cases.GRIFFINLIM_CASES[name]
    y = O.griffinlim(g[f"{name}__S"], **case["gl"])
    ref = g[f"{name}__y"]
    assert y.dtype == ref.dtype and np.array_equal(y, ref)


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present (GPU box)")
def test_griffinlim_oracle_live_reference():
    librosa = ref_shim.load_reference()
    y = golden_cases.make_signal("mix", 7000, 51, None, "float32")
    for skw, gkw in ((dict(n_fft=512, hop_length=160), dict(hop_length=160, n_iter=5, rng=9, momentum=0.5)),
                     (dict(n_fft=400, hop_length=100, center=False), dict(hop_length=100, n_fft=400, center=False, n_iter=2, init=None))):
        S = np.abs(librosa.stft(y, **skw))
        assert np.array_equal(O.griffinlim(S, **gkw), librosa.griffinlim(S, **gkw))
    assert np.array_equal(O.phasor(np.linspace(-7, 7, 101)), librosa.util.phasor(np.linspace(-7, 7, 101)))


# ---- phase vocoder / time stretch ---------------------------------------------------------------------
def test_vocoder_oracle_matches_reference_golden():
    """Restatements of librosa/core/spectrum.py:1459-1519 and effects.py:464-484 against the committed reference outputs
    (oracle/make_golden.py::make_vocoder): bit for bit (same operation order; scipy's own interp1d)."""
    g = np.load(os.path.join(GOLDEN_DIR, "vocoder.npz"))
    D = g["D"]
    assert np.array_equal(O.phase_vocoder(D, rate=2.0), g["pv_rate2"])
    assert np.array_equal(O.phase_vocoder(D, rate=0.6), g["pv_rate06"])
    assert np.array_equal(O.phase_vocoder(D, t_out=g["t_out"]), g["pv_tout"])
    assert np.array_equal(O.phase_vocoder(g["D64"], rate=0.8), g["pv64_rate08"])
    assert np.array_equal(O.time_stretch(g["y"], rate=1.5, n_fft=1024, hop_length=256), g["ts_15"])
    assert np.array_equal(O.time_stretch(g["ys"], rate=0.7), g["ts_stereo_07_default"])


# ---- PCEN ------------------------------------------------------------------------------------------------
def test_pcen_oracle_matches_reference_golden():
    """librosa.pcen (core/spectrum.py:2396-2666) restated -- incl. its scipy.signal.lfilter recurrence and scipy.ndimage max-filter -- vs
    outputs of the unmodified reference (oracle/make_golden.py::make_pcen): bit for bit, float64 results for every input type."""
    g = np.load(os.path.join(GOLDEN_DIR, "pcen.npz"))
    inputs = golden_cases.pcen_inputs(g)
    for name, (key, kw) in golden_cases.PCEN_CASES.items():
        got = O.pcen(inputs[key], **kw)
        assert got.dtype == np.float64 and np.array_equal(got, g[name]), name
    assert np.array_equal(O.pcen(g["A"], ref=g["ref_in"]), g["with_ref"])
    p1, z1 = O.pcen(g["A"][:, :25], return_zf=True)
    p2, z2 = O.pcen(g["A"][:, 25:], zi=z1, return_zf=True)
    for got, key in ((p1, "block1"), (z1, "zf1"), (p2, "block2"), (z2, "zf2")):
        assert got.shape == g[key].shape and np.array_equal(got, g[key]), key
    # the property the reference tests (tests/test_core.py:2533-2573): block-wise with the carried state == one pass
    assert np.allclose(np.hstack([p1, p2]), O.pcen(g["A"]), rtol=1e-12, atol=0)


def test_pcen_oracle_reference_known_answers():
    """The closed-form cases of the reference's own tests (tests/test_core.py:2386-2456, 2516-2530)."""
    rng = np.random.default_rng(20)
    S = np.abs(rng.standard_normal((9, 30)))
    for p in (0.5, 1, 2):
        assert np.allclose(O.pcen(S, gain=0, bias=0, power=p, b=1, time_constant=0.5, eps=1e-6, max_size=1), S**p)
    assert np.allclose(O.pcen(S, gain=1, bias=0, power=1, b=1, time_constant=0.5, eps=1e-20, max_size=1), np.ones_like(S))
    for max_size in (1, 3):
        assert np.allclose(O.pcen(np.zeros((9, 30)), time_constant=0.395, max_size=max_size), 0)
    X = rng.standard_normal((100, 50)) ** 2
    assert np.allclose(O.pcen(X, gain=1, bias=0, power=1, b=1, ref=np.ones_like(X), eps=1e-20), X)
    with pytest.warns(UserWarning, match="complex"):
        assert np.allclose(O.pcen(np.ones((9, 30), dtype=complex), gain=1, bias=0, power=1, time_constant=0.5, eps=1e-20, b=1, max_size=1), 1)
    import scipy.ndimage

    for size in (1, 2, 3, 6, 130):
        for ax in (0, 1):
            assert np.array_equal(O.maximum_filter1d(X, size, ax), scipy.ndimage.maximum_filter1d(X, size, axis=ax))


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present (GPU box)")
def test_pcen_oracle_live_reference():
    L = ref_shim.load_reference()
    rng = np.random.default_rng(21)
    X = (rng.standard_normal((3, 40, 70)) ** 2).astype(np.float32)
    for kw in (dict(), dict(axis=1), dict(max_size=3, max_axis=1), dict(power=0, b=0.3), dict(bias=0, power=2)):
        assert np.array_equal(O.pcen(X, **kw), L.pcen(X, **kw)), kw
    assert np.array_equal(O.pcen(np.arange(100)), L.pcen(np.arange(100)))


# ---- constant-Q / variable-Q transform ---------------------------------------------------------------------
import cqt_oracle as CQ  # noqa: E402


@pytest.mark.parametrize("name", list(golden_cases.CQT_CASES))
def test_cqt_oracle_matches_reference_golden(name):
    """librosa.cqt / vqt (core/constantq.py:42-225, 820-1122) restated -- wavelet tables, sparsification, octave recursion, stacking --
    vs outputs of the unmodified reference (oracle/make_golden.py::make_cqt): bit for bit."""
    fn, (kind, n, seed, channels, dtype), kw = golden_cases.CQT_CASES[name]
    g = np.load(os.path.join(GOLDEN_DIR, "cqt.npz"))
    y = golden_cases.make_signal(kind, n, seed, channels, dtype)
    got = getattr(CQ, fn)(y, sr=golden_cases.SR, res_type="polyphase", **kw)
    assert got.shape == g[name].shape and got.dtype == g[name].dtype and np.array_equal(got, g[name])


@pytest.mark.parametrize("name", list(golden_cases.RESAMPLE_CASES))
def test_resample_oracle_matches_reference_golden(name):
    """librosa.resample (core/audio.py:1002-1178) restated for the scipy-backed converters vs the unmodified reference
    (oracle/make_golden.py::make_resample): bit for bit."""
    (kind, n, seed, channels, dtype), kw = golden_cases.RESAMPLE_CASES[name]
    g = np.load(os.path.join(GOLDEN_DIR, "resample.npz"))
    got = CQ.resample(golden_cases.make_signal(kind, n, seed, channels, dtype), **kw)
    assert got.shape == g[name].shape and got.dtype == g[name].dtype and np.array_equal(got, g[name])


@pytest.mark.parametrize("name", list(golden_cases.PITCH_SHIFT_CASES))
def test_pitch_shift_oracle_matches_reference_golden(name):
    """librosa.effects.pitch_shift (effects.py:573-596) restated (time_stretch + resample + fix_length) vs the unmodified reference: bit for bit."""
    (kind, n, seed, channels, dtype), kw = golden_cases.PITCH_SHIFT_CASES[name]
    g = np.load(os.path.join(GOLDEN_DIR, "resample.npz"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = O.pitch_shift(golden_cases.make_signal(kind, n, seed, channels, dtype), sr=golden_cases.SR, **kw)
    assert got.shape == g[name].shape and got.dtype == g[name].dtype and np.array_equal(got, g[name])


@pytest.mark.parametrize("name", list(golden_cases.CQT_FFT_CASES))
def test_cqt_fft_oracle_matches_reference_golden(name):
    """The recursion with the whole-signal Fourier resampler between the octaves (res_type="fft" / "scipy"): bit for bit."""
    fn, (kind, n, seed, channels, dtype), kw = golden_cases.CQT_FFT_CASES[name]
    g = np.load(os.path.join(GOLDEN_DIR, "resample.npz"))
    got = getattr(CQ, fn)(golden_cases.make_signal(kind, n, seed, channels, dtype), sr=golden_cases.SR, **kw)
    assert got.shape == g[name].shape and got.dtype == g[name].dtype and np.array_equal(got, g[name])


def test_cqt_oracle_known_answers():
    """What the reference's tests assert about the transform itself (tests/test_constantq.py: shape and dtype, the energy of a pure tone
    sits in its bin, an impulse gives a flat column) and about its tables (filters.wavelet_lengths vs constant_q lengths)."""
    sr = 22050
    for midi in (36, 60, 81):
        f = 440.0 * 2.0 ** ((midi - 69) / 12)
        y = np.sin(2 * np.pi * f * np.arange(sr) / sr).astype(np.float32)
        C = np.abs(CQ.cqt(y, sr=sr))
        assert C.shape == (84, 1 + sr // 512) and np.all(np.argmax(C[:, 5:-5], axis=0) == midi - 24)
    freqs = CQ.interval_frequencies(84, fmin=CQ.C1_HZ, bins_per_octave=12)
    assert np.allclose(freqs[12::12] / freqs[:-12:12], 2.0) and abs(freqs[0] - 32.7032) < 1e-4
    lengths, cutoff = CQ.wavelet_lengths(freqs=freqs, sr=sr)
    assert np.all(np.diff(lengths) < 0) and cutoff < sr / 2
    basis, l2 = CQ.wavelet(freqs=freqs[-12:], sr=sr)
    assert basis.shape == (12, 256) and np.allclose(l2, lengths[-12:], rtol=1e-9) and np.allclose(np.sum(np.abs(basis), axis=1), 1.0, atol=1e-6)   # norm=1


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present (GPU box)")
def test_cqt_oracle_live_reference():
    L = ref_shim.load_reference()
    y = golden_cases.make_signal("mix", 16000, 5, (2,), "float32")
    for rt in ("polyphase", "fft"):
        for kw in (dict(), dict(n_bins=36, hop_length=64), dict(scale=False, n_bins=24)):
            assert np.array_equal(CQ.cqt(y, sr=22050, res_type=rt, **kw), L.cqt(y, sr=22050, res_type=rt, **kw)), (rt, kw)
        assert np.array_equal(CQ.vqt(y, sr=22050, res_type=rt, gamma=None), L.vqt(y, sr=22050, res_type=rt))


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("seed", range(200, 212))
def test_db_mfcc_vocoder_oracle_vs_live_reference_wide(seed):
    """Seeded draws over the arguments of the 8(f) rows that are plain functions of a spectrogram: power_to_db / amplitude_to_db and their inverses
    (ref scalar / callable, amin, top_db), mfcc (DCT types 1-3, norm, lifter, from audio and from S), phase_vocoder (rates either side of one)
    and effects.time_stretch -- oracle against the unmodified reference, bit for bit where the arithmetic is NumPy's own."""
    librosa = ref_shim.load_reference()
    rng = np.random.default_rng(seed)
    S = (rng.random((2, int(rng.integers(8, 96)), int(rng.integers(5, 80)))) ** 4 * 10 ** rng.uniform(-6, 3)).astype(np.float32 if rng.random() < 0.7 else np.float64)
    ref = [1.0, 0.37, np.max, np.median][int(rng.integers(0, 4))]
    kw = dict(ref=ref, amin=float(10 ** rng.uniform(-12, -4)), top_db=[None, 80.0, 35.5][int(rng.integers(0, 3))])
    for name in ("power_to_db", "amplitude_to_db"):
        a, b = getattr(librosa, name)(S, **kw), getattr(O, name)(S, **kw)
        assert a.dtype == b.dtype and np.array_equal(a, b), (name, seed)
    dbv = O.power_to_db(S, **kw)
    rs = 1.0 if callable(ref) else ref
    assert np.array_equal(librosa.db_to_power(dbv, ref=rs), O.db_to_power(dbv, ref=rs))
    assert np.array_equal(librosa.db_to_amplitude(dbv, ref=rs), O.db_to_amplitude(dbv, ref=rs))
    mk = dict(n_mfcc=int(rng.integers(5, 30)), dct_type=int(rng.integers(1, 4)), norm=[None, "ortho"][int(rng.integers(0, 2))], lifter=float(rng.choice([0, 0, 22, 7.5])))
    if mk["dct_type"] == 1 and mk["norm"] == "ortho" and S.shape[-2] < 2:
        mk["norm"] = None
    Sdb = O.power_to_db(S)
    a, b = librosa.feature.mfcc(S=Sdb, **mk), O.mfcc(S=Sdb, **mk)
    assert a.dtype == b.dtype and a.shape == b.shape and np.allclose(a, b, rtol=1e-6, atol=1e-6 * np.abs(a).max()), ("mfcc S", seed, mk)
    y = rng.standard_normal(int(rng.integers(3000, 9000))).astype(np.float32)
    a = librosa.feature.mfcc(y=y, sr=22050, n_fft=512, hop_length=128, n_mels=40, **mk)
    b = O.mfcc(y=y, sr=22050, n_fft=512, hop_length=128, n_mels=40, **mk)
    assert a.shape == b.shape and np.allclose(a, b, rtol=1e-4, atol=2e-4 * np.abs(a).max()), ("mfcc y", seed, mk)
    D = librosa.stft(y, n_fft=256, hop_length=64)
    rate = float(rng.choice([0.5, 0.8, 1.0, 1.37, 2.0, 3.1]))
    a, b = librosa.phase_vocoder(D, rate=rate), O.phase_vocoder(D, rate=rate)
    assert a.shape == b.shape and a.dtype == b.dtype and np.abs(a - b).max() <= 2e-5 * np.abs(a).max(), ("vocoder", seed, rate)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", FutureWarning)  # (the reference's time_stretch passes deprecated arguments to its own phase_vocoder)
        a = librosa.effects.time_stretch(y, rate=rate, n_fft=256, hop_length=64)
    b = O.time_stretch(y, rate=rate, n_fft=256, hop_length=64)
    assert a.shape == b.shape and np.abs(a - b).max() <= 5e-5 * max(np.abs(a).max(), 1e-30), ("time_stretch", seed, rate)


# ---- harmonic / percussive separation ----------------------------------------------------------------------------
def test_hpss_oracle_matches_reference_golden():
    """librosa.decompose.hpss (decompose.py:470-528) with util.softmask and magphase, and the effects.hpss / harmonic / percussive chains
    (effects.py:70-301), restated, vs outputs of the unmodified reference (oracle/make_golden.py::make_hpss): bit for bit."""
    g = np.load(os.path.join(GOLDEN_DIR, "hpss.npz"))
    D, y = g["D"], g["y"]
    for name, kw in golden_cases.HPSS_CASES.items():
        h, p = O.hpss(np.abs(D) ** 2 if name.startswith("power_") else D, **kw)
        for got, key in ((h, f"{name}__h"), (p, f"{name}__p")):
            assert got.dtype == g[key].dtype and np.array_equal(got, g[key]), key
    h, p = O.effects_hpss(y, n_fft=512, margin=(1.0, 2.0))
    assert np.array_equal(h, g["effects_h"]) and np.array_equal(p, g["effects_p"])
    assert np.array_equal(O.effects_hpss(y[0])[0], g["effects_harmonic_default"])
    assert np.array_equal(O.effects_hpss(y[0], kernel_size=9, n_fft=1024, hop_length=256)[1], g["effects_percussive_k9"])


def test_hpss_oracle_known_answers():
    """What the reference's tests assert (tests/test_decompose.py: test_hpss, test_real_hpss, test_hpss_margin_error; tests/test_effects.py:
    test_hpss): H + P == D for margin 1, masks sum to one, components are non-negative parts, margins below one are rejected."""
    rng = np.random.default_rng(33)
    D = (rng.standard_normal((33, 50)) + 1j * rng.standard_normal((33, 50))).astype(np.complex64)
    H_, P_ = O.hpss(D)
    assert np.allclose(H_ + P_, D, atol=1e-6)
    S = np.abs(D)
    Hs, Ps = O.hpss(S)
    assert np.allclose(Hs + Ps, S, atol=1e-6) and np.all(Hs >= 0) and np.all(Ps >= 0)
    mh, mp = O.hpss(S, mask=True)
    assert np.allclose(mh + mp, 1.0, atol=1e-6)
    for m in (0.9, (1.0, 0.5), (0.5, 1.0)):
        with pytest.raises(O.ParameterError):
            O.hpss(S, margin=m)
    Hm, Pm = O.hpss(S, margin=(1.0, 4.0))
    assert np.all(Hm + Pm <= S * (1 + 1e-6))     # wider margins leave a residual


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present (GPU box)")
def test_hpss_oracle_live_reference_ignores_window():
    """effects.hpss accepts `window` and passes it to none of its three transforms (effects.py:161-183): a non-default window changes
    nothing in the reference, and therefore nothing in the restatement (the shim used to forward it to the forward stft)."""
    L = ref_shim.load_reference()
    y = golden_cases.make_signal("mix", 6000, 7, (), "float32")
    rh, rp = L.effects.hpss(y, n_fft=512, window="hamming")
    dh, dp = L.effects.hpss(y, n_fft=512)
    assert np.array_equal(rh, dh) and np.array_equal(rp, dp)
    oh, op = O.effects_hpss(y, n_fft=512, window="hamming")
    assert np.array_equal(oh, rh) and np.array_equal(op, rp)


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present (GPU box)")
def test_db_oracle_live_reference_signed_and_array_ref():
    """power_to_db leaves REAL input signed (negative values floor at amin; ref / top_db from the signed maximum), amplitude_to_db takes
    the modulus; `ref` may be an array that broadcasts against S (core/spectrum.py:1855-1881, 2011-2037).  Restatement == reference."""
    L = ref_shim.load_reference()
    rng = np.random.default_rng(5)
    S = rng.standard_normal((3, 20, 30)).astype(np.float32)            # e.g. a difference of spectrograms: both signs
    neg = -np.abs(S) - 1.0                                             # every value negative
    refc = np.array([0.5, 2.0, 1e-12])[:, None, None]
    for X in (S, neg, S.astype(np.float64)):
        for kw in (dict(), dict(ref=np.max), dict(ref=np.max, top_db=30.0), dict(top_db=None), dict(ref=np.median), dict(ref=refc), dict(ref=refc.astype(np.float32), top_db=None)):
            a, b = L.power_to_db(X, **kw), O.power_to_db(X, **kw)
            assert a.dtype == b.dtype and np.array_equal(a, b), kw
        for kw in (dict(), dict(ref=np.max), dict(ref=refc, top_db=40.0)):
            a, b = L.amplitude_to_db(X, **kw), O.amplitude_to_db(X, **kw)
            assert a.dtype == b.dtype and np.array_equal(a, b), kw
    db = rng.standard_normal((3, 4, 5)).astype(np.float32) * 10
    for r in (refc, np.float32(2.0)):
        assert np.array_equal(L.db_to_power(db, ref=r), O.db_to_power(db, ref=r)) and np.array_equal(L.db_to_amplitude(db, ref=r), O.db_to_amplitude(db, ref=r))


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present (GPU box)")
def test_normalize_matches_the_live_reference():
    """util.normalize incl. ``fill`` (None / True / False) and ``threshold``: bit-equal to librosa.util.normalize (util/utils.py:796-1025) on
    real / complex input, every norm kind and axis, error cases alike (the keyword had been dropped)."""
    import warnings
    from librosa_amd.util import utils as U
    R = ref_shim.load_reference()
    rng = np.random.default_rng(0)
    S = rng.standard_normal((5, 7))
    S[:, 2] = 0
    S[1] *= 1e-320
    Sc = S + 1j * rng.standard_normal((5, 7))
    Sc[:, 2] = 0
    for X in (S, Sc, S.astype(np.float32)):
        for norm in (np.inf, -np.inf, 0, 1, 2, 0.5, None, "bad"):
            for axis in (0, -1, None):
                for fill in (None, True, False, 3):
                    for thr in (None, 0.5, -1.0):
                        res = []
                        for fn in (R.util.normalize, U.normalize):
                            try:
                                with warnings.catch_warnings():
                                    warnings.simplefilter("ignore")
                                    res.append(fn(X, norm=norm, axis=axis, fill=fill, threshold=thr))
                            except Exception as exc:  # noqa: BLE001 -- both sides must raise the same kind
                                res.append(type(exc).__name__)
                        a, b = res
                        assert isinstance(a, str) == isinstance(b, str), (norm, axis, fill, thr, a, b)
                        if isinstance(a, str):
                            assert a == b == "ParameterError", (norm, axis, fill, thr, a, b)
                        else:
                            assert a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True), (norm, axis, fill, thr)


def test_soxr_golden_placeholder():
    """The reference's DEFAULT resampler (``res_type="soxr_hq"``: ``core/audio.py:1100, 1162``, reached from every default ``cqt`` / ``vqt`` / ``pitch_shift``) cannot be
    pinned here: the ``soxr`` package is neither installed nor vendored under the reference tree, so the reference itself cannot run its default, and this library serves
    those names with its own band-limited design (parity UNPINNED).  The moment ``import soxr`` works this test stops xfailing and must be turned into the
    golden comparison: generate ``tests/golden/resample_soxr.npz`` with ``oracle/make_golden.py`` from the unmodified reference and compare ``librosa_amd.resample`` to it."""
    try:
        import soxr
        real = hasattr(soxr, "resample")  # oracle/ref_shim.py leaves an EMPTY stand-in module of that name behind when a test imported the reference in this process
    except Exception:
        real = False
    if not real:
        pytest.xfail("soxr is not installed: the soxr_* resamplers stay unpinned (own band-limited design); nothing to compare against")
    assert os.path.exists(os.path.join(GOLDEN_DIR, "resample_soxr.npz")), "soxr is importable now: generate the golden with oracle/make_golden.py and pin librosa_amd.resample(res_type='soxr_hq') to it"
