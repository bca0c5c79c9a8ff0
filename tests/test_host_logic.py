"""Host-side logic of the drop-in (argument validation, tables, frame bookkeeping): CPU only.

Mirrors the reference's own failure tests: tests/test_core.py:295-314, 2941-2963; tests/test_failures.py:18-60;
tests/test_filters.py:120-210; tests/test_util.py (pad_center, fix_length, tiny, dtype_r2c/c2r).
"""
import os
import warnings

import numpy as np
import pytest

import golden_cases
import librosa_amd as L
import stft_oracle as O
from librosa_amd.core import spectrum
from librosa_amd.util import utils as U

from conftest import GOLDEN_DIR


def test_parameter_errors_are_raised_before_any_device_work():
    y = np.zeros(1000, dtype=np.float32)
    with pytest.raises(L.ParameterError):
        L.stft(y, n_fft=2048, center=False)  # tests/test_core.py:295-300
    with pytest.raises(L.ParameterError):
        L.stft(y, n_fft=512, pad_mode="wrap")
    for hop in (0, -1, 2.5):
        with pytest.raises(L.ParameterError):
            L.stft(y, n_fft=512, hop_length=hop)
    with pytest.raises(L.ParameterError):
        L.stft(np.zeros(1000, dtype=np.int16))  # tests/test_failures.py: non-float audio
    with pytest.raises(L.ParameterError):  # the drop-in runs this scan on its staging threads (GPU test); the helper itself:
        U.valid_audio(np.array([0.0, np.nan] * 500, dtype=np.float32))
    with pytest.raises(L.ParameterError):
        L.stft([0.0] * 1000)
    with pytest.raises(L.ParameterError):
        L.stft(np.float32(1.0))
    with pytest.raises(L.ParameterError):
        L.feature.melspectrogram(y=y, n_fft=512, norm="bogus")
    with pytest.raises(L.ParameterError):
        L._spectrogram(y=None, S=None)
    with pytest.raises(L.ParameterError):
        L.istft(np.zeros((1025, 4), dtype=np.float32))  # not complex
    with pytest.raises(L.ParameterError):
        L.filters.get_window(np.ones(7), 8)


def test_spectrogram_passthrough_infers_n_fft():
    # tests/test_core.py:1726-1772 (reference): odd n_fft inference
    S = np.ones((378, 5), dtype=np.float32)
    S2, n_fft = L._spectrogram(S=S, n_fft=2048)
    assert S2 is S and n_fft == 2 * (378 - 1)
    S3, n_fft = L._spectrogram(S=np.ones((1025, 3), np.float32), n_fft=2048)
    assert n_fft == 2048


@pytest.mark.parametrize("name", sorted(k for k, v in golden_cases.CASES.items() if v["mel"] is not None))
def test_mel_basis_is_bit_identical_to_the_reference(name):
    case = golden_cases.CASES[name]
    g = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    _, fkw = golden_cases.split_mel_kwargs(case["mel"])
    B = L.filters.mel(sr=golden_cases.SR, n_fft=case["stft"]["n_fft"], **fkw)
    assert B.dtype == g["mel_basis"].dtype and np.array_equal(B, g["mel_basis"])
    assert np.array_equal(L.filters.mel_cached(sr=golden_cases.SR, n_fft=case["stft"]["n_fft"], **fkw), g["mel_basis"])


def test_mel_scale_known_answers():
    # tests/test_filters.py:35-98 (reference)
    assert np.allclose(L.hz_to_mel(np.array([0, 500, 1000, 2000, 3000])), [0.0, 7.5, 15.0, 25.08188016, 30.97940199])
    assert np.allclose(L.hz_to_mel(np.array([0, 500, 1000, 2000, 3000]), htk=True), [0.0, 607.44591966, 999.98553714, 1521.35955416, 1876.45406012])
    assert np.allclose(L.mel_to_hz(np.array([0, 5, 10, 15, 25, 30])), [0.0, 333.33333333, 666.66666667, 1000.0, 1988.77281813, 2804.64413074])
    assert np.isclose(L.hz_to_mel(2000.0), 25.08188016) and np.isclose(L.mel_to_hz(25.0), 1988.77281813)
    f = L.fft_frequencies(sr=22050, n_fft=2048)
    assert f[0] == 0 and f[-1] == 11025.0 and len(f) == 1025


def test_empty_filter_warning():
    # tests/test_filters.py:195-210 (reference): this configuration has empty filters
    kw = dict(sr=44100, n_fft=1024, n_mels=128, fmin=0, fmax=2000, htk=True)
    with pytest.warns(UserWarning, match="Empty filters"):
        L.filters.mel(**kw)
    with pytest.warns(UserWarning, match="Empty filters"):
        L.filters.mel_cached(**kw)
    with pytest.warns(UserWarning, match="Empty filters"):
        L.filters.mel_cached(**kw)  # the memoised path re-emits the warning


def test_window_sumsquare_and_window_match_the_oracle():
    for hop, n_fft, wl, win in ((512, 2048, None, "hann"), (100, 512, 400, "hann"), (256, 1024, None, "blackmanharris"), (300, 256, None, "hann")):
        a = L.filters.window_sumsquare(window=win, n_frames=23, hop_length=hop, n_fft=n_fft, win_length=wl)
        b = O.window_sumsquare(window=win, n_frames=23, hop_length=hop, n_fft=n_fft, win_length=wl)
        assert a.dtype == b.dtype and np.array_equal(a, b)
    assert np.array_equal(L.filters.get_window("hann", 2048), O.get_window("hann", 2048))
    assert np.array_equal(L.filters.get_window(("kaiser", 4.0), 100), O.get_window(("kaiser", 4.0), 100))
    assert np.array_equal(L.filters.get_window(lambda n: np.ones(n), 16), np.ones(16))


def test_util_helpers():
    assert np.array_equal(U.pad_center(np.ones(5), size=10), np.pad(np.ones(5), (2, 3)))
    with pytest.raises(L.ParameterError):
        U.pad_center(np.ones(5), size=4)
    assert U.fix_length(np.arange(5), size=3).tolist() == [0, 1, 2] and U.fix_length(np.arange(3), size=5).tolist() == [0, 1, 2, 0, 0]
    assert U.tiny(np.float32(1)) == np.finfo(np.float32).tiny and U.tiny(np.complex128(1)) == np.finfo(np.float64).tiny and U.tiny(5) == np.finfo(np.float32).tiny
    assert U.dtype_r2c(np.float32) == np.complex64 and U.dtype_r2c(np.float64) == np.complex128 and U.dtype_r2c(np.float16) == np.complex64
    assert U.dtype_c2r(np.complex64) == np.float32 and U.dtype_c2r(np.complex128) == np.float64 and U.dtype_c2r(np.float32) == np.float32
    assert U.is_positive_int(3) and not U.is_positive_int(0) and not U.is_positive_int(2.0) and not U.is_positive_int(None)
    assert U.valid_audio(np.zeros(4, np.float64))


@pytest.mark.parametrize("n_fft,hop,center,T,length", [(2048, 512, True, 1292, 661500), (2048, 512, True, 1292, None), (512, 128, True, 40, 4000),
                                                        (1024, 256, False, 32, None), (512, 100, True, 51, 5000), (256, 300, True, 17, None)])
def test_istft_frame_bookkeeping_matches_the_oracle(n_fft, hop, center, T, length):
    n_frames, n_used, expected = spectrum._istft_frame_counts(T, n_fft, hop, center, length)
    D = np.zeros((1 + n_fft // 2, T), dtype=np.complex64)
    ref = O.istft(D, hop_length=hop, n_fft=n_fft, center=center, length=length)
    assert expected == ref.shape[-1]
    assert n_used >= n_frames and n_used <= T


def test_finite_check_coverage_rule():
    """The in-kernel non-finite flag only sees samples that lie in some frame."""
    f = spectrum._finite_check_covers_input

    def brute(n, n_fft, hop, center):
        pad = n_fft // 2 if center else 0
        T = 1 + (n + 2 * pad - n_fft) // hop
        seen = np.zeros(n + 2 * pad, bool)
        for t in range(T):
            seen[t * hop : t * hop + n_fft] = True
        return bool(seen[pad : pad + n].all())

    for args in ((661500, 2048, 512, True), (10000, 2048, 2048, True), (10000, 512, 128, False), (10000, 256, 300, True), (4096, 1024, 256, True),
                 (5000, 512, 100, False), (5120, 512, 512, False)):
        assert f(*args) == brute(*args), args


def test_shard_ranges():
    from librosa_amd.distributed import shard_range, shard_sizes

    for n, w in ((4096, 8), (256, 1), (10, 3), (5, 8)):
        covered = []
        for r in range(w):
            b, e = shard_range(n, r, w)
            covered.extend(range(b, e))
        assert covered == list(range(n))
        assert sum(shard_sizes(n, w)) == n and max(shard_sizes(n, w)) - min(shard_sizes(n, w)) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 3, 3)


# ---- block feeder (librosa/core/audio.py:223-533) -----------------------------------------------------
@pytest.mark.parametrize("n,block_length,frame_length,hop_length,fill", [(50000, 16, 2048, 512, None), (50000, 16, 2048, 512, 0.0), (8192 + 3 * 512, 4, 2048, 512, 0.0),
                                                                        (1000, 3, 64, 100, None), (777, 5, 128, 32, 1.5), (100, 8, 256, 64, 0.0)])
def test_stream_blocks_ndarray(n, block_length, frame_length, hop_length, fill):
    """librosa_amd.stream (ring-buffer restatement of the reference's generator) against the direct statement of its block
    geometry, and the property the streaming STFT relies on: block frames tile the frames of the whole signal."""
    import librosa_amd as L

    rng = np.random.default_rng(n)
    for y in (rng.standard_normal(n).astype(np.float32), rng.standard_normal((2, n)).astype(np.float32)):
        for mono in (True, False):
            got = list(L.stream(y, block_length=block_length, frame_length=frame_length, hop_length=hop_length, fill_value=fill, mono=mono))
            src = y.mean(axis=0, dtype=np.float32) if (mono and y.ndim == 2) else y
            want = O.stream_blocks(src, block_length=block_length, frame_length=frame_length, hop_length=hop_length, fill_value=fill)
            assert len(got) == len(want)
            for a, b in zip(got, want):
                assert a.shape == b.shape and a.dtype == np.float32 and np.array_equal(a, b)
    if frame_length >= hop_length and n >= frame_length:
        y = rng.standard_normal(n).astype(np.float32)
        whole = O.stft(y, n_fft=frame_length, hop_length=hop_length, center=False)
        cols = [O.stft(b, n_fft=frame_length, hop_length=hop_length, center=False) for b in
                L.stream(y, block_length=block_length, frame_length=frame_length, hop_length=hop_length, fill_value=0.0) if b.shape[-1] >= frame_length]
        tiled = np.concatenate(cols, axis=-1)
        assert np.array_equal(tiled[:, : whole.shape[1]], whole)


def test_stream_wav_and_arguments(tmp_path):
    import wave

    import librosa_amd as L

    rng = np.random.default_rng(3)
    pcm = (rng.standard_normal((30000, 2)) * 8000).astype("<i2")
    path = str(tmp_path / "x.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(pcm.tobytes())
    y = (pcm.astype(np.float64) / 32768.0).astype(np.float32).T  # (2, n)
    kw = dict(block_length=8, frame_length=1024, hop_length=256)
    got = list(L.stream(path, mono=False, fill_value=0.0, **kw))
    want = O.stream_blocks(y, fill_value=0.0, **kw)
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
    got = list(L.stream(path, offset=0.5, duration=1.0, **kw))  # mono, 8000 samples in, 16000 long
    want = O.stream_blocks(y.mean(axis=0, dtype=np.float32)[8000:24000], **kw)
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
    with pytest.raises(L.ParameterError):
        next(L.stream(y, block_length=0, frame_length=1024, hop_length=256))
    with pytest.raises(L.ParameterError):
        next(L.stream(path, sr=22050, **kw))


# ---- constant-Q host tables (librosa/filters.py:424-722, 838-911; core/intervals.py:28-135; util/utils.py:1500-1597) -------------------
def test_wavelet_tables_match_the_pinned_oracle():
    """The product's own table builders against the oracle's (which is bit-identical to the unmodified reference, tests/test_oracle.py):
    frequencies, relative bandwidths, lengths, the time-domain basis, the sparsified spectrum -- bit for bit."""
    import cqt_oracle as CQ
    from librosa_amd import filters as F
    from librosa_amd.core import constantq as C

    freqs = L.interval_frequencies(84, fmin=CQ.C1_HZ, bins_per_octave=12)
    assert np.array_equal(freqs, CQ.interval_frequencies(84, fmin=CQ.C1_HZ, bins_per_octave=12))
    assert np.array_equal(L.interval_frequencies(9, fmin=55.0, intervals=[1, 1.3, 1.7]), CQ.interval_frequencies(9, fmin=55.0, intervals=[1, 1.3, 1.7]))
    assert np.array_equal(F._relative_bandwidth(freqs=freqs), CQ.relative_bandwidth(freqs))
    for kw in (dict(), dict(window="hamming", filter_scale=0.5, gamma=None), dict(gamma=3.0), dict(window=("kaiser", 4.0))):
        a, b = F.wavelet_lengths(freqs=freqs, sr=22050, **kw), CQ.wavelet_lengths(freqs=freqs, sr=22050, **kw)
        assert np.array_equal(a[0], b[0]) and a[1] == b[1], kw
        fa, la = F.wavelet(freqs=freqs[-12:], sr=22050, **kw)
        fb, lb = CQ.wavelet(freqs=freqs[-12:], sr=22050, **kw)
        assert fa.dtype == np.complex64 and np.array_equal(fa, fb) and np.array_equal(la, lb), kw
    for w in ("hann", "ones", ("kaiser", 4.0), "tukey", 5.0):
        assert F.window_bandwidth(w) == CQ.window_bandwidth(w), w
    rng = np.random.default_rng(0)
    X = rng.standard_normal((7, 50)) + 1j * rng.standard_normal((7, 50))
    for q in (0.0, 0.01, 0.3):
        a, b = U.sparsify_rows(X, quantile=q, dtype=np.complex64), CQ.sparsify_rows(X, quantile=q, dtype=np.complex64)
        assert a.dtype == b.dtype and (a != b).nnz == 0
    # the per-octave plan of the default transform: seven octaves of n_fft = 256, hops 512 ... 8, each basis == the oracle's
    plan = C._plan(22050.0, 512, CQ.C1_HZ, 84, "equal", 0.0, 12, 1.0, 1.0, 0.01, "hann", True, np.dtype(np.complex64).str)
    assert plan["early"] == 0 and [(o["n_fft"], o["hop"], o["bin0"]) for o in plan["octaves"]] == [(256, 512 >> i, 72 - 12 * i) for i in range(7)]
    alpha = CQ.relative_bandwidth(freqs)
    for i, o in enumerate(plan["octaves"]):
        sl = slice(-12, None) if i == 0 else slice(-12 * (i + 1), -12 * i)
        ref, n_fft, _ = CQ.vqt_filter_fft(22050.0 / 2**i, freqs[sl], 1.0, 1.0, 0.01, window="hann", gamma=0.0, dtype=np.complex64, alpha=alpha[sl])
        ref[:] *= np.sqrt(22050.0 / (22050.0 / 2**i))
        ref.sort_indices()
        assert n_fft == o["n_fft"] and np.array_equal(ref.indptr, o["row_ptr"]) and np.array_equal(ref.indices, o["col"]) and np.array_equal(ref.data, o["val"])


def test_cqt_and_pcen_argument_errors_before_any_device_work():
    y = np.zeros(4000, dtype=np.float32)
    for bad in (dict(tuning=None), dict(fmin=20000.0), dict(n_bins=200), dict(pad_mode="wrap"), dict(hop_length=0), dict(res_type="zero_order_hold"), dict(res_type="linear"),
                dict(dtype=np.float32)):
        with pytest.raises(L.ParameterError):
            L.cqt(y, **bad)
    with pytest.raises(L.ParameterError):
        L.vqt(y, intervals="ji5")
    with pytest.raises(L.ParameterError):
        L.cqt(np.zeros(4000, dtype=np.int16))
    S = np.ones((9, 30))
    for kw in (dict(gain=-1), dict(bias=-1), dict(power=-0.1), dict(b=-2), dict(b=2), dict(time_constant=-2), dict(eps=0), dict(max_size=1.5), dict(max_size=0)):
        with pytest.raises(L.ParameterError):       # tests/test_core.py:2354-2383
            L.pcen(S, **kw)
    with pytest.raises(L.ParameterError):
        L.pcen(np.arange(100), max_size=3)          # :2510-2513
    with pytest.raises(L.ParameterError):
        L.pcen(np.ones((3, 4, 5)), max_size=3)      # 3-d input needs max_axis (:2502-2507 passes it)


# ---- in-process multi-device sharding of a NumPy batch (core/spectrum.py::_sharded_host_exec) and its bounded memos -----------------
class _FakeCtx:
    def __init__(self, device):
        import threading

        self.device = device
        self.call_lock = threading.RLock()
        self.streams = 0

    def use_own_stream(self):
        self.streams += 1


class _FakeSess:
    def __init__(self, ctx):
        self.ctx = ctx


def _fake_devices(monkeypatch, devs):
    from librosa_amd import _native

    ctxs = {d: _FakeCtx(d) for d in set(devs) | {0}}
    monkeypatch.setattr(_native, "host_devices", lambda: list(devs))
    monkeypatch.setattr(_native, "get_context", lambda device=None: ctxs[0 if device is None else device])
    return ctxs


@pytest.mark.parametrize("devs,batch", [([0, 1, 2, 3, 4, 5, 6, 7], 4096), ([0, 1, 2], 10), ([0, 0], 7), ([3, 1], 5), ([0], 9)])
def test_numpy_batches_shard_over_devices_without_gaps_or_overlap(monkeypatch, devs, batch):
    """Every clip is served exactly once, by the context of the device its range belongs to (clip i of B on device i * n // B up to the
    balanced remainder: distributed.shard_range), the session's own device on the calling thread, the others on their own threads under
    their contexts' locks.  Reference property kept: batch == per item (tests/test_multichannel.py:96-111)."""
    import threading

    ctxs = _fake_devices(monkeypatch, devs)
    served = np.zeros(batch, np.int32)
    by_dev = {}
    who = {}

    def run(ctx, b, e):
        served[b:e] += 1
        by_dev.setdefault(ctx.device, []).append((b, e))
        who[ctx.device] = threading.current_thread().name
        return b == 0  # (one shard reports a non-finite sample)

    flagged = spectrum._sharded_host_exec(_FakeSess(ctxs[0]), batch, 1 << 40, run)
    assert np.array_equal(served, np.ones(batch, np.int32))
    assert flagged is True
    if len(devs) > 1:
        assert sorted(by_dev) == sorted(set(devs))
        sizes = sorted(e - b for r in by_dev.values() for b, e in r)
        assert sizes[-1] - sizes[0] <= 1 and len(sizes) == len(devs)
        for d in set(devs) - {0}:
            assert who[d] == f"lra-dev{d}" and ctxs[d].streams == 1   # its own thread, its own stream, under its lock
        if 0 in devs:
            assert who[0] == threading.current_thread().name and ctxs[0].streams == 0
    else:
        assert by_dev == {0: [(0, batch)]}


def test_small_jobs_stay_on_one_device_and_errors_surface(monkeypatch):
    ctxs = _fake_devices(monkeypatch, [0, 1])
    calls = []
    spectrum._sharded_host_exec(_FakeSess(ctxs[0]), 64, 1 << 20, lambda c, b, e: calls.append((c.device, b, e)))   # 1 MB: not worth a second context
    spectrum._sharded_host_exec(_FakeSess(ctxs[0]), 3, 1 << 40, lambda c, b, e: calls.append((c.device, b, e)))    # fewer than two clips per device
    assert calls == [(0, 0, 64), (0, 0, 3)]

    ctxs2 = _fake_devices(monkeypatch, [1])   # one listed device that is not the session's own: everything goes there, under its lock and stream
    spectrum._sharded_host_exec(_FakeSess(ctxs2[0]), 64, 1 << 40, lambda c, b, e: calls.append((c.device, b, e)))
    assert calls[-1] == (1, 0, 64) and ctxs2[1].streams == 1
    ctxs = _fake_devices(monkeypatch, [0, 1])

    def boom(c, b, e):
        if c.device == 1:
            raise L.ParameterError("Audio buffer is not finite everywhere")
        return False

    with pytest.raises(L.ParameterError):
        spectrum._sharded_host_exec(_FakeSess(ctxs[0]), 64, 1 << 40, boom)
    assert ctxs[1].call_lock.acquire(blocking=False)  # released on the error path


@pytest.mark.parametrize("n,n_fft,hop,center,mode,world", [(50000, 2048, 512, True, "constant", 3), (50000, 2048, 512, True, "reflect", 4), (30011, 1024, 256, False, "constant", 5),
                                                           (9000, 512, 128, True, "symmetric", 7), (9000, 512, 100, True, "edge", 2), (3000, 2048, 512, True, "reflect", 2), (2048, 2048, 512, True, "constant", 8)])
def test_shard_frames_are_column_slices_of_the_whole_transform(n, n_fft, hop, center, mode, world):
    """distributed.shard_frames: the uncentred transform of a rank's sample range (+ halo, + its share of the centre padding) IS that
    rank's block of columns of the whole clip's STFT -- bit for bit on the oracle -- and the ranks' blocks tile the frame axis without gaps or overlap.
    Reference property: core/spectrum.py:273-328 (padding belongs to the clip's ends), :380-390 (frames are independent), core/audio.py:223-533 (stream)."""
    from librosa_amd.distributed import frame_shard_input, shard_frames, stft_num_frames

    y = np.random.default_rng(n).standard_normal((2, n)).astype(np.float32)
    D = O.stft(y, n_fft=n_fft, hop_length=hop, center=center, pad_mode=mode)
    assert stft_num_frames(n, n_fft, hop, center) == D.shape[-1]
    nxt = 0
    for r in range(world):
        sh = shard_frames(n, r, world, n_fft, hop, center)
        assert sh["frame_lo"] == nxt and sh["n_frames"] == D.shape[-1]
        nxt = sh["frame_hi"]
        if sh["frame_hi"] == sh["frame_lo"]:
            continue
        assert (sh["pad_left"] > 0) <= (r == 0 or sh["frame_lo"] * hop < n_fft // 2) and sh["sample_hi"] - sh["sample_lo"] == n_fft + hop * (sh["frame_hi"] - sh["frame_lo"] - 1)
        piece = frame_shard_input(y, sh, mode)
        if sh["pad_left"] == 0 and sh["pad_right"] == 0:
            assert np.shares_memory(piece, y)  # interior shards are views: no copy of a ten-hour clip
        d = O.stft(piece, n_fft=n_fft, hop_length=hop, center=False)
        assert np.array_equal(d, D[..., sh["frame_lo"] : sh["frame_hi"]])
    assert nxt == D.shape[-1]


def test_frame_shard_plan_and_threads(monkeypatch):
    """Which calls shard by frames (fewer clips than two per device, enough frames per device, LRA_DEVICES opted in) and who serves what."""
    import threading

    ctxs = _fake_devices(monkeypatch, [0, 1, 2, 3])
    n = 4 * 3600 * 22050  # a four-hour clip
    nf = 1 + n // 512
    plan = spectrum._frame_shard_plan(n, nf, 1, 1 << 40, 2048, 512, True)
    # the frames that touch the centre padding are runs of their own (two frames at either end of the clip): the padded copies stay tiny
    assert [d for d, _ in plan] == [0, 0, 1, 2, 3, 3] and plan[0][1]["pad_left"] == 1024 and plan[0][1]["frame_hi"] == 2 and plan[-1][1]["pad_right"] > 0
    assert all(sh["pad_left"] == 0 and sh["pad_right"] == 0 for _, sh in plan[1:-1]) and plan[-1][1]["frame_hi"] - plan[-1][1]["frame_lo"] <= 3
    assert sum(sh["frame_hi"] - sh["frame_lo"] for _, sh in plan) == nf and all(a[1]["frame_hi"] == b[1]["frame_lo"] for a, b in zip(plan, plan[1:]))
    assert spectrum._frame_shard_plan(n, nf, 8, 1 << 40, 2048, 512, True) is None      # two clips per device: sharded by clips instead
    assert spectrum._frame_shard_plan(22050, 44, 1, 1 << 40, 2048, 512, True) is None  # too few frames per device
    assert spectrum._frame_shard_plan(n, nf, 1, 1 << 20, 2048, 512, True) is None      # too little data
    _fake_devices(monkeypatch, [0])
    assert spectrum._frame_shard_plan(n, nf, 1, 1 << 40, 2048, 512, True) is None      # one device
    ctxs = _fake_devices(monkeypatch, [0, 1, 1])
    plan = spectrum._frame_shard_plan(n, nf, 1, 1 << 40, 2048, 512, True)
    who = {}

    def run(c, sh):
        who.setdefault(c.device, []).append((threading.current_thread().name, sh["frame_lo"]))
        return sh["frame_lo"] > 0 and c.device == 1

    assert spectrum._frame_sharded_host_exec(_FakeSess(ctxs[0]), plan, run) is True
    assert set(t for t, _ in who[0]) == {threading.current_thread().name} and set(t for t, _ in who[1]) == {"lra-dev1"} and len(who[1]) >= 2 and ctxs[1].streams == 1
    assert [f for _, f in who[1]] == sorted(f for _, f in who[1])

    def boom(c, sh):
        if c.device == 1:
            raise L.ParameterError("x")
        return False

    with pytest.raises(L.ParameterError):
        spectrum._frame_sharded_host_exec(_FakeSess(ctxs[0]), plan, boom)
    assert ctxs[1].call_lock.acquire(blocking=False)


def test_host_devices_env(monkeypatch):
    from librosa_amd import _native

    monkeypatch.setattr(_native, "device_count", lambda: 4)
    for k in ("LOCAL_RANK", "WORLD_SIZE", "LIBROSA_AMD_DEVICE", "LRA_DEVICES"):
        monkeypatch.delenv(k, raising=False)

    class _Ctx:
        device = 0

    monkeypatch.setattr(_native, "get_context", lambda device=None: _Ctx())
    assert _native.host_devices() == [0]          # opt-in: a plain call stays on the process's own device
    _Ctx.device = 2
    assert _native.host_devices() == [2]
    monkeypatch.setenv("LRA_DEVICES", "0")
    assert _native.host_devices() == [0]
    monkeypatch.setenv("LRA_DEVICES", "2, 3")
    assert _native.host_devices() == [2, 3]
    monkeypatch.setenv("LRA_DEVICES", "7")
    with pytest.raises(L.ParameterError):
        _native.host_devices()
    monkeypatch.setenv("LRA_DEVICES", "all")
    assert _native.host_devices() == [0, 1, 2, 3]


def test_window_sumsquare_memo_is_bounded_by_bytes():
    """every distinct (n_frames, length) is a new envelope of one value per output sample; the memo holds at most 8 of them and
    at most 128 MB, and never an array above a quarter of that."""
    lru = spectrum._ByteBoundedLRU(max_entries=3, max_bytes=4000)
    made = []

    def build(n):
        made.append(n)
        return np.zeros(n, np.float32)

    for n in (100, 200, 100, 150, 120):      # 400 + 800 + 600 + 480 bytes, three entries at most
        lru.get(n, lambda n=n: build(n))
    assert made == [100, 200, 150, 120] and lru.nbytes() <= 4000 and len(lru._d) == 3
    big = lru.get("big", lambda: np.zeros(2000, np.float32))   # 8 000 bytes > budget / 4: returned, not kept
    assert big.nbytes == 8000 and "big" not in lru._d
    for n in range(300, 320):
        lru.get(n, lambda n=n: build(n))
    assert lru.nbytes() <= 4000 and len(lru._d) <= 3
    assert spectrum._WSS_CACHE.max_entries == 8 and spectrum._WSS_CACHE.max_bytes == 128 << 20


def test_resample_plans_and_filters():
    """Host side of librosa_amd.resample: 7-smooth transform lengths, the Fourier form's padding / output grid, the polyphase filter
    (scipy's design and alignment) and argument errors raised before any device work."""
    import math

    import scipy.signal

    from librosa_amd.core.audio import _band_plan, _rational_filter, _smooth_at_least

    for n in (1, 2, 11, 97, 1501, 65537):
        m = _smooth_at_least(n)
        r = m
        for p in (2, 3, 5, 7):
            while r % p == 0:
                r //= p
        assert m >= n and r == 1 and all(any(k % p for p in (2, 3, 5, 7)) or k < n for k in range(n, m))   # the smallest such number
    for n_in, up, down in ((661500, 320, 441), (9000, 441, 320), (5000, 160, 147), (4096, 2, 3)):
        fft_in, fft_out, k_mid, k_sigma = _band_plan(n_in, up, down)
        f_c = 0.5 * min(1.0, up / down)
        assert fft_in % down == 0 and fft_out == fft_in // down * up and fft_in >= n_in + 110 / f_c      # exact output grid, linear convolution
        assert fft_out >= -(-n_in * up // down)
        assert abs(k_mid / fft_in - 0.9565 * f_c) < 1e-12 and abs((k_mid + 3.5 * k_sigma) / fft_in - f_c) < 1e-12   # -125 dB at the lower Nyquist
    for up, down in ((1, 2), (3, 2), (160, 441), (441, 160)):
        for real in (np.dtype(np.float32), np.dtype(np.float64)):
            taps, first = _rational_filter(up, down, "polyphase", real)
            half = 10 * max(up, down)
            ref = scipy.signal.firwin(2 * half + 1, 1.0 / max(up, down), window=("kaiser", 5.0)).astype(real) * real.type(up)
            lead = down - half % down
            assert taps.dtype == real and len(taps) == lead + 2 * half + 1 and not taps[:lead].any() and np.array_equal(taps[lead:], ref) and first == (half + lead) // down
    y = np.zeros(100, dtype=np.float32)
    for bad in (dict(orig_sr=22050.5, target_sr=8000, res_type="polyphase"), dict(orig_sr=22050.5, target_sr=8000, res_type="soxr_hq"), dict(orig_sr=2, target_sr=1, res_type="linear"),
                dict(orig_sr=0, target_sr=1, res_type="fft")):
        with pytest.raises(L.ParameterError):
            L.resample(y, **bad)
    with pytest.raises(L.ParameterError):
        L.resample(np.zeros(10, dtype=np.int32), orig_sr=2, target_sr=1)
    assert L.resample(y, orig_sr=16000, target_sr=16000) is y
    with pytest.raises(L.ParameterError):
        L.effects.pitch_shift(y, sr=22050, n_steps=1, bins_per_octave=-3)
    assert math.gcd(320, 441) == 1



def test_row_pitch_and_frame_major_strides():
    """Round 5: the padded row pitch of device-resident results and the stride test that lets `istft` / `melspectrogram(S=...)` read such a view in place."""
    import torch

    from librosa_amd.core import spectrum as SP

    assert SP.row_pitch(1025, 8, 128) == 1040 and SP.row_pitch(1025, 4, 128) == 1056 and SP.row_pitch(1025, 16, 128) == 1032
    assert SP.row_pitch(4097, 8, 128) == 4112 and SP.row_pitch(513, 8, 128) == 528
    assert SP.row_pitch(257, 8, 128) == 257          # rows under 4 KiB stay packed (measured: no gain)
    assert SP.row_pitch(1025, 8, 0) == 1025 and SP.row_pitch(1024, 8, 128) == 1024
    buf = torch.zeros((6, 7, 1040), dtype=torch.complex64)
    view = buf[..., :1025]
    assert SP._frame_major_strides(view, 1025) == (7 * 1040, 1040)
    assert SP._frame_major_strides(view.view(2, 3, 7, 1040)[..., :1025] if False else buf.view(2, 3, 7, 1040)[..., :1025], 1025) == (7 * 1040, 1040)
    assert SP._frame_major_strides(buf[::2, :, :1025], 1025) == (2 * 7 * 1040, 1040)   # clips further apart: still one batch stride
    assert SP._frame_major_strides(buf.view(2, 3, 7, 1040)[:, ::2, :, :1025], 1025) is None   # leading axes that do not collapse
    assert SP._frame_major_strides(torch.zeros((3, 7, 1025), dtype=torch.complex64), 1025) == (7 * 1025, 1025)
    assert SP._frame_major_strides(torch.zeros((3, 1025, 7), dtype=torch.complex64).transpose(-1, -2).transpose(-1, -2), 7) == (1025 * 7, 7)
    assert SP._frame_major_strides(torch.zeros((3, 1025, 7), dtype=torch.complex64).transpose(-1, -2), 1025) is None  # bin axis not contiguous
    assert SP._frame_major_strides(buf[:, :, ::2][..., :300], 300) is None
    assert SP._frame_major_strides(buf[0, :, :1025], 1025) == (7 * 1040, 1040)   # no leading axis


def test_bench_compact_line_keeps_the_contract_and_fits():
    """bench.py prints ONE short line (the driver's record truncates long ones): contract keys, roofline with the path's fractions folded in, cpu_baseline; <= 4 KB."""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    prose = "x" * 700
    full = {"metric": "STFT+mel frames/sec (n_fft=2048 hop=512)", "value": 5.8e8, "unit": "frames/s", "n_gpus": 1, "steps": 50, "warmup": 10, "ms_per_step": 0.56, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": prose, "frames_per_step_per_gpu": 330752, "clips_per_gpu": 256, "prewarm_ms": 400.0, "parallelism": prose, "device": "gfx950"},
            "roofline": {"bound": "hbm", "kernel": "k", "achieved": 1500.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1875, "traffic": 8.6e8, "traffic_box": "profile", "bytes_per_frame": 2560,
                         "launch_ms": 0.56, "note": prose, "limited_by": prose},
            "roofline_valu": {"frac": 0.24}, "roofline_stft": {"frac": 0.57, "launch_ms": 0.74, "traffic": 3.4e9, "achievable_note": prose},
            "roofline_istft": {"frac": 0.6, "launch_ms": 0.71, "round_trip_snr_db_min": 136.5, "call_note": prose},
            "stream_ceiling": {"forward": {"frac": 0.59}, "inverse": {"frac": 0.6}, "what": prose}, "cqt_lite": {"frac_of_hbm": 0.51, "ms_total": 3.6, "per_n_fft": {"512": {"ms": 0.3}, "8192": {"ms": 2.6}}},
            "placement": {"allocations": 5, "stft_frac_best": 0.61, "stft_frac_worst": 0.56, "istft_frac_best": 0.62, "stft_ms": [0.7] * 5},
            "cpu_baseline": {"value": 154615.4, "unit": "frames/s", "cores": 1, "kind": "reference", "sample": prose, "host": {"cpu_model": "EPYC"}},
            "cpu_baseline_all_cores": {"value": 1.9e6, "cores": 64}, "parity": {"mel_max_rel_err_vs_reference": 5e-6, "oracle_equals_reference": True, "bar": 1e-4, "sample": prose},
            "scaling_base": {"clips_per_gpu": 512, "value": 5.9e8, "unit": "frames/s", "ms_per_step": 1.12}, "repeats": {"ms_per_step_min": 0.55, "ms_per_step_median": 0.56, "ms_per_step_all": [0.56] * 5},
            "griffinlim": {"ms_per_iteration": 0.43, "ms_setup": 0.5, "what": prose}, "hpss": {"error": "boom"}, "dropin_torch": {"ms_per_call": 0.57, "what": prose},
            "kernel_forms": {"stft_radix_16_8_8": {"ms": 0.64, "frac": 0.658}, "stft_radix_16_16_4": {"ms": 0.63, "frac": 0.672}, "mel_one_wave": {"ms": 0.56}, "mel_producer_consumer": {"ms": 0.557},
                             "what": prose}, "long_clip": {"frac_of_batched": 0.97, "what": prose}}
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) <= 4096
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["frac"] == 0.1875 and line["roofline"]["valu_frac"] == 0.24
    # the path's other fractions are SCALARS of `roofline` (the driver's record drops nested objects: VERDICT r05)
    path = line["roofline"]
    assert all(not isinstance(v, (dict, list)) for v in path.values())
    assert path["stft_v2_frac"] == 0.658 and path["stft_v3_frac"] == 0.672 and path["mel_pc_ms"] == 0.557 and path["long_clip_vs_batched"] == 0.97
    assert path["stft_frac"] == 0.57 and path["istft_frac"] == 0.6 and path["stream_forward_frac"] == 0.59 and path["stream_inverse_frac"] == 0.6 and path["cqt_lite_frac"] == 0.51
    assert path["stft_frac_best_placement"] == 0.61 and path["placements"] == 5
    assert line["cpu_baseline"]["kind"] == "reference" and line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["all_cores"] == 64
    assert line["scaling_base"]["clips_per_gpu"] == 512 and line["side"]["griffinlim_ms_setup"] == 0.5 and line["side_errors"] == ["hpss"]
    assert "workload" in line["config"] and len(line["config"]["workload"]) < 300


def test_native_gather_unequal_shards_tables():
    """NativeGather.all_gather with n_items % world != 0: the byte sizes / offsets handed to lra_comm_allgatherv are those of shard_range (the unsharded layout)."""
    import torch

    from librosa_amd.distributed import NativeGather

    class _Ctx:
        def set_stream(self, s):
            pass

    class _Comm:
        rank = 1

        def __init__(self):
            self.calls = []

        def allgather(self, *a):
            self.calls.append(("equal",) + a)

        def allgatherv(self, send, recv, sizes, offs):
            self.calls.append(("v", list(sizes), list(offs)))

    class _Stream:
        cuda_stream = 0

    g = NativeGather.__new__(NativeGather)
    g.ctx, g.world, g.comm = _Ctx(), 3, _Comm()
    orig = torch.cuda.current_stream
    torch.cuda.current_stream = lambda dev=None: _Stream()
    try:
        local = torch.zeros((2, 4, 5), dtype=torch.float32)     # rank 1 of 3 holds 2 of 7 items
        full = g.all_gather(local, n_items=7)
        assert tuple(full.shape) == (7, 4, 5)
        item = 4 * 5 * 4
        assert g.comm.calls == [("v", [3 * item, 2 * item, 2 * item], [0, 3 * item, 5 * item])]
        g.comm.calls.clear()
        full = g.all_gather(local, n_items=6)                    # equal shards: the single all-gather
        assert tuple(full.shape) == (6, 4, 5) and g.comm.calls[0][0] == "equal"
        with pytest.raises(ValueError):
            g.all_gather(torch.zeros((3, 4, 5)), n_items=7)
    finally:
        torch.cuda.current_stream = orig


def test_placed_buffer_pool_is_bounded():
    """Context.placed_give (the recycling side of lra_malloc_placed results): at most two waiting buffers per shape and PLACED_KEEP_BYTES in all -- a stream of
    variable-length inputs (a new result shape per call) releases the oldest shapes instead of pinning HBM."""
    import threading

    from librosa_amd import _native

    class FakeLib:
        def __init__(self):
            self.freed = []

        def lra_free_placed(self, h, p):
            self.freed.append(p.value)
            return 0

    c = _native.Context.__new__(_native.Context)
    c._lock, c._placed_free, c.lib, c.handle = threading.RLock(), {}, FakeLib(), None
    GB = 1 << 30
    for ptr in (1, 2, 3):
        c.placed_give(5 * GB, 8200, ptr, 1292)
    assert c._placed_free == {(5 * GB, 8200, 1292): [1, 2]} and c.lib.freed == [3]
    c.placed_give(4 * GB, 8192, 4)                      # 14 GB waiting > 12: the oldest shape gives one up
    assert c._placed_free == {(5 * GB, 8200, 1292): [2], (4 * GB, 8192, 0): [4]} and c.lib.freed == [3, 1]
    c.placed_give(20 * GB, 8192, 5)                     # larger than the whole budget: released at once
    assert c.lib.freed == [3, 1, 5] and (20 * GB, 8192, 0) not in c._placed_free
    for i in range(40):                                 # forty different shapes of 1 GB each: never more than the budget waiting
        c.placed_give(GB + i * 8192, 8192, 100 + i)
        assert sum(k[0] * len(v) for k, v in c._placed_free.items()) <= c.PLACED_KEEP_BYTES
