"""Generate ``tests/golden/*.npz`` by running the UNMODIFIED reference through ``ref_shim``.

TEST INFRASTRUCTURE ONLY; runs only where ``/root/reference`` exists (the build container):

    python oracle/make_golden.py

Each fixture stores the input PCM, the call parameters (JSON) and the reference outputs
(``librosa.stft``, ``librosa.feature.melspectrogram``, ``librosa.istft``).  The fixtures travel to
the GPU box with the repo; the reference tree does not.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import golden_cases  # noqa: E402
import ref_shim  # noqa: E402
import stft_oracle  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SUBSET_FRAMES_CFG1 = [0, 1, 2, 3, 215, 427, 428, 429, 430]
SUBSET_FRAMES_CFG2 = [0, 1, 2, 3, 645, 1288, 1289, 1290, 1291]


def main():
    librosa = ref_shim.load_reference()
    import scipy

    os.makedirs(OUT, exist_ok=True)
    meta = dict(numpy=np.__version__, scipy=scipy.__version__, reference_version=str(librosa.__version__))

    for name, case in golden_cases.CASES.items():
        kind, n, seed, channels, dtype = case["signal"]
        y = golden_cases.make_signal(kind, n, seed, channels, dtype)
        skw = dict(case["stft"])
        D = librosa.stft(y, **skw)
        arrays = dict(y=y, D=np.ascontiguousarray(D))
        if case["mel"] is not None:
            power, fkw = golden_cases.split_mel_kwargs(case["mel"])
            mkw = {k: v for k, v in skw.items()}
            mkw.setdefault("hop_length", int(skw.get("win_length", skw["n_fft"]) // 4))
            arrays["mel"] = librosa.feature.melspectrogram(y=y, sr=golden_cases.SR, power=power, **mkw, **fkw)
            arrays["mel_basis"] = librosa.filters.mel(sr=golden_cases.SR, n_fft=skw["n_fft"], **fkw)
        if case["istft"]:
            ikw = {k: v for k, v in skw.items() if k in ("hop_length", "win_length", "n_fft", "window", "center")}
            arrays["y_istft_len"] = librosa.istft(D, length=y.shape[-1], **ikw)
            arrays["y_istft_nolen"] = librosa.istft(D, **ikw)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), params=json.dumps(dict(case=name, **meta)), **arrays)
        print(f"{name}: D{D.shape} {D.dtype}")

    # BASELINE config 1: 10 s 440 Hz sine (stored: all mel frames, a frame subset of D)
    y = stft_oracle.config1_input()
    D = librosa.stft(y, n_fft=2048, hop_length=512)
    M = librosa.feature.melspectrogram(y=y, sr=22050, n_fft=2048, hop_length=512, n_mels=128)
    yh = librosa.istft(D, hop_length=512, length=len(y))
    np.savez_compressed(
        os.path.join(OUT, "config1_sine10s.npz"),
        params=json.dumps(dict(case="config1", **meta)),
        frames=np.array(SUBSET_FRAMES_CFG1),
        D_frames=np.ascontiguousarray(D[:, SUBSET_FRAMES_CFG1]),
        D_absmax=np.abs(D).max(),
        mel=M,
        y_istft_head=yh[:4096],
        y_istft_tail=yh[-4096:],
    )
    print("config1", D.shape, M.shape)

    # BASELINE config 2 clips 0 and 37 (30 s): all mel frames of clip 0, subset of clip 37, D subset
    Y = np.stack([stft_oracle.config_input(1, first_clip=i)[0] for i in (0, 37)])
    store = dict(clips=np.array([0, 37]), frames=np.array(SUBSET_FRAMES_CFG2))
    for j, i in enumerate((0, 37)):
        D = librosa.stft(Y[j], n_fft=2048, hop_length=512)
        M = librosa.feature.melspectrogram(y=Y[j], sr=22050, n_fft=2048, hop_length=512, n_mels=128)
        store[f"D_frames_{i}"] = np.ascontiguousarray(D[:, SUBSET_FRAMES_CFG2])
        store[f"D_absmax_{i}"] = np.abs(D).max()
        store[f"y_head_{i}"] = Y[j][:2048].copy()  # guards the seeded generator itself
        if i == 0:
            store["mel_0"] = M
        else:
            store[f"mel_frames_{i}"] = np.ascontiguousarray(M[:, SUBSET_FRAMES_CFG2])
            store[f"mel_absmax_{i}"] = M.max()
    np.savez_compressed(os.path.join(OUT, "config2_clips.npz"), params=json.dumps(dict(case="config2", **meta)), **store)
    print("config2 done")
    make_db_mfcc(librosa, meta)
    make_griffinlim(librosa, meta)
    make_vocoder(librosa, meta)
    make_pcen(librosa, meta)
    make_cqt(librosa, meta)
    make_resample(librosa, meta)
    make_hpss(librosa, meta)


def make_vocoder(librosa, meta):
    """librosa.phase_vocoder / effects.time_stretch outputs of the reference."""
    import warnings

    warnings.simplefilter("ignore", FutureWarning)  # time_stretch hands phase_vocoder its deprecated keywords (effects.py:471-476)
    y = golden_cases.make_signal("mix", 12000, 61, None, "float32")
    # broadband-floored signals: where a bin is below the float32 leakage floor its phase is rounding noise, and the vocoder
    # ACCUMULATES phase -- on a clean chirp two correct STFT implementations give different outputs once the chirp reaches
    # a bin that was quiet before ("precision")
    ys = np.stack([y, golden_cases.make_signal("mix", 12000, 62, None, "float32")])
    D = librosa.stft(y, n_fft=1024, hop_length=256)
    t_out = np.array([0.0, 0.5, 3.25, 3.25, 10.9, 20.0, 46.999])
    store = dict(y=y, ys=ys, D=np.ascontiguousarray(D), t_out=t_out)
    store["pv_rate2"] = librosa.phase_vocoder(D, rate=2.0)
    store["pv_rate06"] = librosa.phase_vocoder(D, rate=0.6)
    store["pv_tout"] = librosa.phase_vocoder(D, t_out=t_out)
    store["ts_15"] = librosa.effects.time_stretch(y, rate=1.5, n_fft=1024, hop_length=256)
    store["ts_stereo_07_default"] = librosa.effects.time_stretch(ys, rate=0.7)
    D64 = librosa.stft(ys.astype(np.float64), n_fft=512)
    store["D64"] = np.ascontiguousarray(D64)
    store["pv64_rate08"] = librosa.phase_vocoder(D64, rate=0.8)
    np.savez_compressed(os.path.join(OUT, "vocoder.npz"), params=json.dumps(dict(case="vocoder", **meta)), **store)
    print("vocoder done")


def make_pcen(librosa, meta):
    """librosa.pcen outputs of the reference, incl. the block-wise use of docs/examples/plot_pcen_stream.py."""
    y = golden_cases.make_signal("mix", 12000, 81, (2,), "float32")
    M = librosa.feature.melspectrogram(y=y, sr=golden_cases.SR, n_fft=1024, hop_length=256, n_mels=32)   # (2, 32, 47)
    A = np.abs(librosa.stft(y[0], n_fft=128, hop_length=200))                                               # (65, 61)
    store = dict(M=M, A=A)
    inputs = golden_cases.pcen_inputs(store)
    for name, (key, kw) in golden_cases.PCEN_CASES.items():
        store[name] = librosa.pcen(inputs[key], **kw)
    ref = np.maximum(A, np.roll(A, 1, axis=0))
    store["ref_in"] = ref
    store["with_ref"] = librosa.pcen(A, ref=ref)
    p1, z1 = librosa.pcen(A[:, :25], return_zf=True)                 # two blocks, the state carried over
    p2, z2 = librosa.pcen(A[:, 25:], zi=z1, return_zf=True)
    store.update(block1=p1, zf1=z1, block2=p2, zf2=z2)
    np.savez_compressed(os.path.join(OUT, "pcen.npz"), params=json.dumps(dict(case="pcen", **meta)), **store)
    print("pcen done")


def make_cqt(librosa, meta):
    """librosa.cqt / librosa.vqt outputs of the reference (res_type="polyphase": soxr is not in the image)."""
    store = {}
    for name, (fn, (kind, n, seed, channels, dtype), kw) in golden_cases.CQT_CASES.items():
        y = golden_cases.make_signal(kind, n, seed, channels, dtype)
        store[name] = getattr(librosa, fn)(y, sr=golden_cases.SR, res_type="polyphase", **kw)
    np.savez_compressed(os.path.join(OUT, "cqt.npz"), params=json.dumps(dict(case="cqt", **meta)), **store)
    print("cqt done")


def make_resample(librosa, meta):
    """librosa.resample (core/audio.py:1002-1178) with the scipy-backed converters, librosa.cqt / vqt with res_type="fft" / "scipy", and
    librosa.effects.pitch_shift (effects.py:487-596) through the Fourier converter."""
    store = {}
    for name, ((kind, n, seed, channels, dtype), kw) in golden_cases.RESAMPLE_CASES.items():
        y = golden_cases.make_signal(kind, n, seed, channels, dtype)
        store[name] = librosa.resample(y, **kw)
    for name, (fn, (kind, n, seed, channels, dtype), kw) in golden_cases.CQT_FFT_CASES.items():
        y = golden_cases.make_signal(kind, n, seed, channels, dtype)
        store[name] = getattr(librosa, fn)(y, sr=golden_cases.SR, **kw)
    for name, ((kind, n, seed, channels, dtype), kw) in golden_cases.PITCH_SHIFT_CASES.items():
        y = golden_cases.make_signal(kind, n, seed, channels, dtype)
        store[name] = librosa.effects.pitch_shift(y, sr=golden_cases.SR, **kw)
    np.savez_compressed(os.path.join(OUT, "resample.npz"), params=json.dumps(dict(case="resample", **meta)), **store)
    print("resample done")


def make_hpss(librosa, meta):
    """librosa.decompose.hpss / effects.hpss outputs of the reference."""
    y = golden_cases.make_signal("mix", 6000, 85, (2,), "float32")
    D = librosa.stft(y[0], n_fft=128, hop_length=100)                  # (65, 61)
    store = dict(y=y, D=np.ascontiguousarray(D))
    for name, kw in golden_cases.HPSS_CASES.items():
        h, p = librosa.decompose.hpss(D if not name.startswith("power_") else np.abs(D) ** 2, **kw)
        store[f"{name}__h"], store[f"{name}__p"] = h, p
    h, p = librosa.effects.hpss(y, n_fft=512, margin=(1.0, 2.0))
    store["effects_h"], store["effects_p"] = h, p
    store["effects_harmonic_default"] = librosa.effects.harmonic(y[0])
    store["effects_percussive_k9"] = librosa.effects.percussive(y[0], kernel_size=9, n_fft=1024, hop_length=256)
    np.savez_compressed(os.path.join(OUT, "hpss.npz"), params=json.dumps(dict(case="hpss", **meta)), **store)
    print("hpss done")


def make_griffinlim(librosa, meta):
    """librosa.griffinlim outputs of the reference on |stft| of seeded signals."""
    store = {}
    for name, case in golden_cases.GRIFFINLIM_CASES.items():
        kind, n, seed, channels, dtype = case["signal"]
        y = golden_cases.make_signal(kind, n, seed, channels, dtype)
        S = np.abs(librosa.stft(y, **case["stft"]))
        store[f"{name}__S"] = S
        store[f"{name}__y"] = librosa.griffinlim(S, **case["gl"])
    np.savez_compressed(os.path.join(OUT, "griffinlim.npz"), params=json.dumps(dict(case="griffinlim", **meta)), **store)
    print("griffinlim done")


def make_db_mfcc(librosa, meta):
    """power_to_db / amplitude_to_db / db_to_power and mfcc outputs of the reference."""
    rng = np.random.default_rng(2718)
    y = golden_cases.make_signal("mix", 16000, 21, (2,), "float32")
    M = librosa.feature.melspectrogram(y=y, sr=golden_cases.SR, n_fft=1024, hop_length=256, n_mels=40)  # (2, 40, 63)
    A = np.abs(librosa.stft(y[0], n_fft=512, hop_length=128))  # amplitudes (257, 126)
    store = dict(y=y, M=M, A=A)
    store["db_default"] = librosa.power_to_db(M)
    store["db_refmax"] = librosa.power_to_db(M, ref=np.max)
    store["db_notop"] = librosa.power_to_db(M, top_db=None, amin=1e-6, ref=2.5)
    store["db_median_top30"] = librosa.power_to_db(M, ref=np.median, top_db=30.0)
    store["db_axes_last"] = librosa.power_to_db(M, ref=np.max, axes=-1, top_db=40.0)
    store["db_axes_none"] = librosa.power_to_db(M, ref=np.max, axes=None)
    store["adb_default"] = librosa.amplitude_to_db(A)
    store["adb_refmax"] = librosa.amplitude_to_db(A, ref=np.max, top_db=60.0)
    store["pow_back"] = librosa.db_to_power(store["db_notop"], ref=2.5)
    store["amp_back"] = librosa.db_to_amplitude(librosa.amplitude_to_db(A, top_db=None), ref=1.0)
    store["mfcc_S"] = librosa.feature.mfcc(S=store["db_default"], n_mfcc=13)
    store["mfcc_S_t3_lift"] = librosa.feature.mfcc(S=store["db_default"], n_mfcc=20, dct_type=3, lifter=22)
    store["mfcc_S_t1_none"] = librosa.feature.mfcc(S=store["db_default"], n_mfcc=12, dct_type=1, norm=None)
    store["mfcc_y"] = librosa.feature.mfcc(y=y, sr=golden_cases.SR, n_mfcc=20, n_fft=1024, hop_length=256, n_mels=40)
    y2 = stft_oracle.config_input(1, n=22050 * 2)[0]
    store["y2"] = y2
    store["mfcc_y2_default"] = librosa.feature.mfcc(y=y2, sr=22050)  # n_fft 2048, hop 512, 128 mels, 20 coefficients
    store["mfcc_y2_htk_lift"] = librosa.feature.mfcc(y=y2, sr=22050, n_mfcc=13, lifter=2 * 13, htk=True, n_mels=64, fmax=8000.0)
    np.savez_compressed(os.path.join(OUT, "db_mfcc.npz"), params=json.dumps(dict(case="db_mfcc", **meta)), **store)
    print("db_mfcc done")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("griffinlim", "vocoder", "pcen", "cqt", "hpss", "resample"):  # only this fixture (the others are unchanged)
        _librosa = ref_shim.load_reference()
        import scipy as _scipy

        _meta = dict(numpy=np.__version__, scipy=_scipy.__version__, reference_version=str(_librosa.__version__))
        dict(griffinlim=make_griffinlim, vocoder=make_vocoder, pcen=make_pcen, cqt=make_cqt, hpss=make_hpss, resample=make_resample)[sys.argv[1]](_librosa, _meta)
    else:
        main()
