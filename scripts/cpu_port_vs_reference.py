"""Build container only (needs /root/reference): times the UNMODIFIED reference (through oracle/ref_shim.py) and the oracle port on the
same clips, one core, and writes profiles/cpu_port_vs_reference.json.  bench.py quotes the ratio next to its cpu_baseline ("port")."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy
from threadpoolctl import threadpool_limits
import ref_shim, stft_oracle as O
librosa = ref_shim.load_reference()
y = O.config_input(4, n=22050 * 30)
kw = dict(sr=22050, n_fft=2048, hop_length=512, n_mels=128)


def rate(fn, seconds=10.0):
    fn(y[0])
    frames, k, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        frames += fn(y[k % 4]).shape[-1]
        k += 1
    return frames / (time.perf_counter() - t0)


with threadpool_limits(limits=1):
    r_ref_mel = rate(lambda c: librosa.feature.melspectrogram(y=c, **kw))
    r_port_mel = rate(lambda c: O.melspectrogram(y=c, **kw))
    r_ref_stft = rate(lambda c: librosa.stft(c, n_fft=2048, hop_length=512))
    r_port_stft = rate(lambda c: O.stft(c, n_fft=2048, hop_length=512))
out = {"where": "build container (no GPU)", "cpu_model": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
       "numpy": np.__version__, "scipy": scipy.__version__, "reference_version": str(librosa.__version__), "cores": 1,
       "melspectrogram_frames_per_s": {"reference": r_ref_mel, "port": r_port_mel, "port_over_reference": r_port_mel / r_ref_mel},
       "stft_frames_per_s": {"reference": r_ref_stft, "port": r_port_stft, "port_over_reference": r_port_stft / r_ref_stft},
       "note": "the port omits the reference's MAX_MEM_BLOCK column blocking (core/spectrum.py:380-390) and its head/middle/tail split; numba is stubbed in the shim (affects istft only)"}
json.dump(out, open(os.path.join(ROOT, "profiles", "cpu_port_vs_reference.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
