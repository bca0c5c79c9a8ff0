"""Development probe (GPU box): where the host side of one librosa_amd.cqt call goes (cProfile over 300 calls)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, librosa_amd as L
dev = torch.device("cuda", 0)
y = bench.make_batch(torch, 64, 22050 * 30, 0, dev)
fn = lambda: L.cqt(y, sr=22050, hop_length=512, res_type="polyphase")
for _ in range(20): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(100): fn()
e1.record(); t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"wall to issue 100 calls {t_issue * 10:.3f} ms/call; GPU span {e0.elapsed_time(e1) / 100:.3f} ms/call", flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): fn()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(18)
