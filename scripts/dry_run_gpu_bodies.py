"""Dry run of GPU test bodies on the CPU (development aid for when no GPU time is left): the device is replaced by stand-ins --
`Tensor.cuda()` is the identity, sessions are the simulator sessions of tests/test_hostsim.py (kernels of lra_pcen.h / lra_cqt.h /
lra_hpss.h run by tests/hostsim/postsim.cpp), the forward / inverse transforms come from the oracle -- so that the LOGIC of a test
written for hardware (names, shapes, layouts, tolerances) is exercised before its first run there.  Test infrastructure only.

    python scripts/dry_run_gpu_bodies.py          # the two hpss bodies of tests/test_gpu_parity.py
"""
import sys, os, types, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle'), ROOT]
import librosa_amd as L, stft_oracle as O, test_hostsim as TH, test_gpu_parity as TG
from librosa_amd import _arrays, effects
from librosa_amd.core import spectrum
# device stand-ins
torch.Tensor.cuda = lambda self, *a, **k: self
torch.Tensor.is_cuda = property(lambda self: True)
torch.cuda.synchronize = lambda *a, **k: None
class Ev:
    def __init__(self, enable_timing=False): pass
    def record(self): pass
    def elapsed_time(self, other): return 0.0
torch.cuda.Event = Ev
real_session = _arrays.Session
SimT = TH._sim_torch_session(real_session)
class Dispatch:
    def __new__(cls, like):
        return SimT(like) if isinstance(like, torch.Tensor) else TH._SimSession(like)
_arrays.Session = Dispatch
effects._stage = lambda a: (a, False)
def to_np(a): return a.numpy() if isinstance(a, torch.Tensor) else a
def wrap(like, out): return torch.from_numpy(np.ascontiguousarray(out)) if isinstance(like, torch.Tensor) else out
def stft(a, check_finite=True, **kw):
    D = O.stft(to_np(a), **kw)
    if isinstance(a, torch.Tensor):   # frame-major buffer viewed (bins, frames), as the device returns it
        return torch.from_numpy(np.ascontiguousarray(np.swapaxes(D, -1, -2))).transpose(-1, -2)
    return D
def istft(a, **kw): return wrap(a, O.istft(np.ascontiguousarray(to_np(a)), **kw))
spectrum.stft = stft; spectrum.istft = istft; L.stft = stft; L.istft = istft
# smaller "full size"
orig = O.config_input
O.config_input = lambda batch, **kw: orig(min(batch, 4), n=22050 * 3)
TG._hpss_golden_body(L)
print("golden body ok")
# the properties body indexes clip 17 of 32: patch indices through a smaller batch
import inspect
src = inspect.getsource(TG._hpss_properties_body).replace("clip = 17", "clip = 2")
ns = dict(TG.__dict__); exec(src, ns); ns["_hpss_properties_body"](L)
print("properties body ok")
