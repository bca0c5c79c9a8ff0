"""Case table for the golden fixtures (``tests/golden/*.npz``).

TEST INFRASTRUCTURE ONLY.  Each case names a seeded synthetic input and the keyword arguments of
the reference calls whose outputs are stored.  ``oracle/make_golden.py`` runs the UNMODIFIED
reference on them (build container only); the tests replay the same inputs through the oracle
(CPU) and through the HIP path (GPU) and compare with the stored reference outputs.

Sizes follow the reference's own test parametrisations (``tests/test_core.py:256-292, 317-371,
813-828``; ``tests/test_multichannel.py:96-111, 266-285, 685-714``) plus BASELINE.json's configs.
"""
from __future__ import annotations

import numpy as np

SR = 22050


def make_signal(kind, n, seed, channels=None, dtype="float32"):
    """Seeded synthetic PCM: 'noise', 'tone' (440 Hz), 'mix' (0.1 noise + 0.5 tone), 'chirp'."""
    rng = np.random.default_rng(seed)
    shape = (n,) if channels is None else tuple(channels) + (n,)
    t = np.arange(n, dtype=np.float64) / SR
    if kind == "noise":
        y = 0.3 * rng.standard_normal(shape)
    elif kind == "tone":
        y = np.broadcast_to(np.sin(2 * np.pi * 440.0 * t), shape).copy()
    elif kind == "mix":
        y = 0.1 * rng.standard_normal(shape) + 0.5 * np.sin(2 * np.pi * 440.0 * t)
    elif kind == "chirp":
        # exponential chirp 55 Hz -> 55*2^7 Hz, like tests/test_core.py:749-753
        dur = n / SR
        k = (55.0 * 2**7 / 55.0) ** (1.0 / dur)
        phase = 2 * np.pi * 55.0 * (k**t - 1.0) / np.log(k)
        y = np.broadcast_to(np.cos(phase), shape).copy()
    else:
        raise ValueError(kind)
    return np.clip(y, -1.0, 1.0).astype(dtype)


# name -> dict(signal=(kind, n, seed, channels, dtype), stft=kwargs, mel=kwargs|None, istft=bool)
CASES = {
    # reference test_stft sizes (tests/test_core.py:256-292)
    "stft_n256_h64": dict(signal=("mix", 4000, 1, None, "float32"), stft=dict(n_fft=256, hop_length=64), mel=dict(n_mels=20), istft=True),
    "stft_n256_default_hop_ones": dict(signal=("noise", 3000, 2, None, "float32"), stft=dict(n_fft=256, window="ones"), mel=None, istft=True),
    "stft_n256_nocenter": dict(signal=("noise", 4000, 3, None, "float32"), stft=dict(n_fft=256, hop_length=128, center=False), mel=dict(n_mels=16), istft=True),
    "stft_n501_odd": dict(signal=("noise", 3000, 4, None, "float32"), stft=dict(n_fft=501, hop_length=128), mel=None, istft=True),
    "stft_n1023_h129": dict(signal=("mix", 8192, 5, None, "float32"), stft=dict(n_fft=1023, hop_length=129), mel=None, istft=False),
    # headline shape at 1 s
    "stft_n2048_h512_1s": dict(signal=("mix", 22050, 6, None, "float32"), stft=dict(n_fft=2048, hop_length=512), mel=dict(n_mels=128), istft=True),
    # pad modes / window variants (tests/test_core.py:2583-2711 use reflect + win_length<n_fft)
    "stft_n512_reflect": dict(signal=("chirp", 6000, 7, None, "float32"), stft=dict(n_fft=512, hop_length=128, pad_mode="reflect"), mel=dict(n_mels=40), istft=True),
    "stft_n512_edge_win400": dict(signal=("mix", 5000, 8, None, "float32"), stft=dict(n_fft=512, hop_length=100, win_length=400, pad_mode="edge"), mel=None, istft=True),
    "stft_n1024_blackmanharris": dict(signal=("chirp", 12000, 9, None, "float32"), stft=dict(n_fft=1024, hop_length=256, window="blackmanharris"), mel=None, istft=True),
    "stft_n4096_h512": dict(signal=("mix", 30000, 10, None, "float32"), stft=dict(n_fft=4096, hop_length=512), mel=dict(n_mels=64), istft=True),
    "stft_n8192_h512": dict(signal=("mix", 40000, 11, None, "float32"), stft=dict(n_fft=8192, hop_length=512), mel=None, istft=False),
    "stft_n512_h512_hop_eq": dict(signal=("noise", 9000, 12, None, "float32"), stft=dict(n_fft=512, hop_length=512), mel=None, istft=False),
    "stft_short_signal": dict(signal=("noise", 300, 13, None, "float32"), stft=dict(n_fft=512, hop_length=128), mel=None, istft=True),
    # multichannel (tests/test_multichannel.py)
    "stft_stereo_n1024": dict(signal=("noise", 6000, 14, (2,), "float32"), stft=dict(n_fft=1024, hop_length=256), mel=dict(n_mels=32), istft=True),
    "stft_batch_2x3_n512": dict(signal=("noise", 4096, 15, (2, 3), "float32"), stft=dict(n_fft=512, hop_length=128), mel=dict(n_mels=24), istft=True),
    # float64 -> complex128
    "stft_f64_n2048": dict(signal=("chirp", 16000, 16, None, "float64"), stft=dict(n_fft=2048, hop_length=512), mel=dict(n_mels=128), istft=True),
    # mel variants
    "mel_htk_fmin_fmax": dict(signal=("mix", 16000, 17, None, "float32"), stft=dict(n_fft=1024, hop_length=256), mel=dict(n_mels=40, htk=True, fmin=50.0, fmax=8000.0), istft=False),
    "mel_norm_none_power1": dict(signal=("mix", 16000, 18, None, "float32"), stft=dict(n_fft=2048, hop_length=512), mel=dict(n_mels=64, norm=None, power=1.0), istft=False),
    "mel_norm_1": dict(signal=("mix", 12000, 19, None, "float32"), stft=dict(n_fft=1024, hop_length=512), mel=dict(n_mels=32, norm=1), istft=False),
}


# SURVEY.md 8f rank 3 -- griffinlim (reference tests: tests/test_core.py:2583-2711: win_length < n_fft, n_fft = 2049,
# pad_mode="reflect", length=, momentum 0 / 0.99, init None / "random").  name -> dict(signal, stft kwargs that produce the
# magnitude, griffinlim kwargs)
GRIFFINLIM_CASES = {
    "gl_n512_h128_rand": dict(signal=("chirp", 6000, 31, None, "float32"), stft=dict(n_fft=512, hop_length=128), gl=dict(n_iter=4, hop_length=128, rng=0)),
    "gl_n512_win400_reflect_len": dict(signal=("chirp", 6000, 32, None, "float32"), stft=dict(n_fft=512, hop_length=128, win_length=400, pad_mode="reflect"),
                                       gl=dict(n_iter=8, hop_length=128, win_length=400, n_fft=512, pad_mode="reflect", length=6000, rng=3)),
    "gl_n1024_init_none_mom0": dict(signal=("mix", 9000, 33, None, "float32"), stft=dict(n_fft=1024), gl=dict(n_iter=3, init=None, momentum=0.0)),
    "gl_n2049_odd": dict(signal=("chirp", 12000, 34, None, "float32"), stft=dict(n_fft=2049, hop_length=512), gl=dict(n_iter=2, n_fft=2049, hop_length=512, rng=5)),
    "gl_stereo_n2048_32iter": dict(signal=("mix", 22050, 35, (2,), "float32"), stft=dict(n_fft=2048, hop_length=512), gl=dict(n_iter=32, hop_length=512, rng=7)),
    "gl_f64_n512": dict(signal=("chirp", 5000, 36, None, "float64"), stft=dict(n_fft=512, hop_length=128), gl=dict(n_iter=3, hop_length=128, rng=11)),
}


def split_mel_kwargs(mel_kwargs):
    """melspectrogram kwargs -> (power, filter kwargs)."""
    kw = dict(mel_kwargs)
    power = kw.pop("power", 2.0)
    return power, kw


PCEN_CASES = {
    # name: (input key, kwargs)      inputs: M = mel power (2, 32, 47) float32, A = |stft| (65, 61) float32, A64 = float64 copy of A
    "default": ("A", dict()),
    "default_f64": ("A64", dict()),
    "log": ("A", dict(power=0)),
    "nobias": ("A", dict(bias=0, power=0.25)),
    "speech_max3": ("A", dict(gain=0.8, bias=10, power=0.25, time_constant=0.06, max_size=3)),
    "b_explicit": ("A", dict(b=0.11, sr=16000, hop_length=160)),
    "time_first": ("At", dict(axis=0, max_size=4)),
    "stereo_max5": ("M", dict(max_size=5, max_axis=-2)),
    "stereo_timeaxis1": ("Mt", dict(axis=1)),
}


def pcen_inputs(g):
    """The input arrays PCEN_CASES names, from the two stored ones."""
    import numpy as np

    M, A = g["M"], g["A"]
    return dict(M=M, A=A, A64=A.astype(np.float64), At=np.ascontiguousarray(A.T), Mt=np.ascontiguousarray(np.swapaxes(M, 1, 2)))


CQT_CASES = {
    # name: (function, signal (kind, n, seed, channels, dtype), kwargs); all with res_type="polyphase" (the resampler both sides can run)
    "cqt_default": ("cqt", ("mix", 33075, 91, None, "float32"), dict()),
    "cqt_stereo_60": ("cqt", ("mix", 20000, 92, (2,), "float32"), dict(hop_length=256, n_bins=60)),
    "cqt_early_downsample": ("cqt", ("chirp", 33075, 93, None, "float32"), dict(n_bins=24)),
    "cqt_partial_octave": ("cqt", ("mix", 22050, 94, None, "float32"), dict(n_bins=30, fmin=65.4, tuning=0.25, norm=2, scale=False)),
    "cqt_f64": ("cqt", ("mix", 22050, 95, None, "float64"), dict(n_bins=48, bins_per_octave=24, fmin=220.0)),
    "cqt_hamming_reflect": ("cqt", ("mix", 22050, 96, None, "float32"), dict(filter_scale=0.5, pad_mode="reflect", window="hamming", sparsity=0.05)),
    "vqt_default": ("vqt", ("mix", 22050, 97, None, "float32"), dict()),
    "vqt_gamma_36": ("vqt", ("mix", 22050, 98, (2,), "float32"), dict(gamma=5.0, bins_per_octave=36, n_bins=108, hop_length=128)),
    "vqt_intervals": ("vqt", ("mix", 22050, 99, None, "float32"), dict(intervals=[1.0, 1.2, 1.5, 1.8], n_bins=16, fmin=200.0, gamma=0)),
}


RESAMPLE_CASES = {
    # name: (signal (kind, n, seed, channels, dtype), librosa.resample kwargs); the scipy-backed converters (the ones the reference runs here)
    "fft_22050_8000": (("mix", 6000, 131, None, "float32"), dict(orig_sr=22050, target_sr=8000, res_type="fft")),
    "fft_halve_even": (("mix", 5000, 132, (2,), "float32"), dict(orig_sr=2, target_sr=1, res_type="fft", scale=True)),
    "fft_halve_odd": (("mix", 5001, 133, None, "float32"), dict(orig_sr=2, target_sr=1, res_type="scipy", scale=True)),
    "fft_up_even_f64": (("mix", 3000, 134, None, "float64"), dict(orig_sr=8000, target_sr=22050, res_type="fft")),
    "fft_up_double": (("chirp", 2048, 135, None, "float32"), dict(orig_sr=1, target_sr=2, res_type="fft", scale=True)),
    "fft_prime_length": (("mix", 4099, 136, None, "float32"), dict(orig_sr=22050, target_sr=16000, res_type="scipy")),
    "poly_22050_8000": (("mix", 6000, 137, None, "float32"), dict(orig_sr=22050, target_sr=8000, res_type="polyphase")),
    "poly_up_f64_scale": (("mix", 3000, 138, (2,), "float64"), dict(orig_sr=16000, target_sr=22050, res_type="polyphase", scale=True)),
    "poly_44100_22050": (("chirp", 9001, 139, None, "float32"), dict(orig_sr=44100, target_sr=22050, res_type="polyphase")),
}

PITCH_SHIFT_CASES = {
    # name: (signal, librosa.effects.pitch_shift kwargs); sr = SR.  res_type="fft": the converter that accepts the non-integer intermediate rate
    "shift_up_4": (("mix", 12000, 151, None, "float32"), dict(n_steps=4, res_type="fft")),
    "shift_down_tritone_stereo": (("mix", 9000, 152, (2,), "float32"), dict(n_steps=-6, res_type="fft", n_fft=1024)),
    # ("mix", not "chirp": the vocoder accumulates the phase differences of EVERY frame, so a bin's rounding-level phase while it is empty
    # becomes its phase offset once the sweep reaches it -- the reference's own output for a pure chirp moves by O(1) with the last bit of the STFT)
    "shift_quarter_tones_scale": (("mix", 8000, 153, None, "float32"), dict(n_steps=3, bins_per_octave=24, res_type="scipy", scale=True, n_fft=512, hop_length=128)),
}

CQT_FFT_CASES = {
    # as CQT_CASES with the whole-signal Fourier resampler between the octaves (res_type in the kwargs)
    "cqt_fft_default": ("cqt", ("mix", 33075, 141, None, "float32"), dict(res_type="fft")),
    "cqt_scipy_early_downsample": ("cqt", ("chirp", 33075, 142, None, "float32"), dict(n_bins=24, res_type="scipy")),
    "vqt_fft_stereo_odd": ("vqt", ("mix", 22051, 143, (2,), "float32"), dict(res_type="fft", hop_length=256, n_bins=60)),
    "cqt_fft_f64": ("cqt", ("mix", 22050, 144, None, "float64"), dict(n_bins=48, bins_per_octave=24, fmin=220.0, res_type="fft")),
}


HPSS_CASES = {
    # name: decompose.hpss kwargs on D = stft(mix, n_fft=128, hop=100) (65, 61) complex64; "power_*": on |D|**2 (real input)
    "default": dict(),
    "kernels_margins": dict(kernel_size=(13, 31), margin=(1.0, 3.0)),
    "masks_p1": dict(power=1.0, mask=True),
    "even_kernel_p35": dict(kernel_size=8, power=3.5),
    "hard": dict(power=np.inf),
    "hard_masks": dict(power=np.inf, mask=True),
    "long_kernel": dict(kernel_size=(70, 9)),
    "power_default": dict(),
    "power_masks_margin": dict(margin=2.5, mask=True),
}
