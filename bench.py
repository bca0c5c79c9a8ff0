#!/usr/bin/env python
"""bench.py -- STFT+mel frames/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path -- feature.melspectrogram (fused framing + window + FFT +
|X|^2 + banded mel, one kernel launch) -- over one batch of synthetic 22.05 kHz clips already
resident in HBM: BASELINE.json configs[1], batch=256 clips x 30 s, n_fft=2048 hop=512 n_mels=128,
per GPU (weak scaling: clips are independent, every rank gets its own 256, no data-path collective).
Plans/tables are created outside the timed region.  The timed region is bracketed by a
barrier + device synchronize on both sides; the max over ranks is reported.

Extra keys on the JSON line:
  roofline       dominant kernel of the step (the fused mel kernel): algorithmic bytes / HIP-event time
  roofline_stft  the complex64-out STFT kernel on the same input (the north star's >=70 %-of-HBM bar
                 is attached to this kernel: 10 248 B/frame), timed in the same process
  roofline_istft the inverse on the STFT's output (BASELINE configs[3]; round-trip SNR included)
  roofline_valu  the fused mel kernel against the f32 vector peak (it sits on the compute side of the ridge)
  cqt_lite       BASELINE configs[4]: STFTs at n_fft 512 / 2048 / 8192 over the same batch (N=1 only)
  kernel_variants  which n_fft=2048 tuning the plans settled on (first-call autotune)
  cpu_baseline   the NumPy/scipy.fft oracle (a port of the reference path) on this box's host cores,
                 rank 0 at N=1 only, on a bounded sample of the same workload (one core); cpu_baseline_all_cores
                 = the same with one independent process per core (up to 64)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR, N_FFT, HOP, N_MELS = 22050, 2048, 512, 128
CLIP_SECONDS = 30
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured on the chip: plain copy 4.8 TB/s, this kernel's store stream 4.7
BYTES_PER_FRAME_MEL = HOP * 4 + N_MELS * 4            # 2 560 B: PCM read once + mel written once
BYTES_PER_FRAME_STFT = HOP * 4 + (N_FFT // 2 + 1) * 8  # 10 248 B: PCM read once + complex64 spectrum written once


def make_batch(torch, batch, n, first_clip, device):
    """SURVEY.md 8(d) config-2 style input generated on the device: 0.1*noise + 0.5*tone(f_i), clipped."""
    g = torch.Generator(device=device)
    g.manual_seed(440 + first_clip)
    t = torch.arange(n, device=device, dtype=torch.float32) / SR
    idx = torch.arange(first_clip, first_clip + batch, device=device)
    f = 110.0 * torch.pow(torch.tensor(2.0, device=device), (idx % 72).float() / 12.0)
    y = 0.1 * torch.randn(batch, n, device=device, generator=g)
    y += 0.5 * torch.sin(2 * np.pi * f[:, None] * t[None, :])
    return y.clamp_(-1.0, 1.0).contiguous()


def cpu_baseline(seconds=12.0):
    """Oracle (NumPy restatement of the reference path, oracle/stft_oracle.py) timed on this host, 1 core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import stft_oracle as O

    import contextlib

    try:
        from threadpoolctl import threadpool_limits

        limiter = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        limiter = contextlib.nullcontext()
    y = O.config_input(4, n=SR * CLIP_SECONDS)
    O.melspectrogram(y=y[0], sr=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS)  # warm
    frames = 0
    clips = 0
    with limiter:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            M = O.melspectrogram(y=y[clips % 4], sr=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS)
            frames += M.shape[-1]
            clips += 1
    dt = time.perf_counter() - t0
    return {
        "value": frames / dt,
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{clips} clips x {CLIP_SECONDS} s melspectrogram (n_fft={N_FFT} hop={HOP} n_mels={N_MELS}) in {dt:.1f} s, "
                  f"1 process, BLAS limited to 1 thread; host has {os.cpu_count()} logical cores",
    }


def _cpu_worker(seconds):
    """One host process of the all-cores baseline (spawned: no torch / HIP state in the child)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import stft_oracle as O

    try:
        from threadpoolctl import threadpool_limits

        threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        pass
    y = O.config_input(2, n=SR * CLIP_SECONDS)
    O.melspectrogram(y=y[0], sr=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS)
    frames, clips, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        frames += O.melspectrogram(y=y[clips % 2], sr=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS).shape[-1]
        clips += 1
    return frames, time.perf_counter() - t0


def cpu_baseline_all_cores(seconds=8.0, max_procs=64):
    """The reference has no internal parallelism: its fair multi-core mode is one independent process per core
    (SURVEY.md 8d).  P = min(cores, max_procs) spawned processes, each looping over clips for `seconds`."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor

    procs = max(1, min(os.cpu_count() or 1, max_procs))
    with ProcessPoolExecutor(max_workers=procs, mp_context=mp.get_context("spawn")) as pool:
        res = list(pool.map(_cpu_worker, [seconds] * procs))
    return {"value": sum(f / dt for f, dt in res), "unit": "frames/s", "cores": procs, "kind": "port",
            "sample": f"{procs} independent processes x {seconds:.0f} s of 30 s-clip melspectrograms (n_fft={N_FFT} hop={HOP} n_mels={N_MELS}), 1 BLAS thread each; host has {os.cpu_count()} logical cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cqt", action="store_true", help="skip the CQT-lite (config 5) side measurement")
    ap.add_argument("--sweep", action="store_true", help="time kernel tuning variants (development aid), prints extra lines to stderr")
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--iters", type=int, default=None)
    args = ap.parse_args()

    import torch

    import librosa_amd as L
    from librosa_amd import filters

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (no CPU fallback)"
    # LRA_BENCH_BACKEND=gloo: control-flow check of the N > 1 path on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("LRA_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)  # nccl = RCCL; used for the barriers and the max-over-ranks of the timing only

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    n = SR * CLIP_SECONDS
    batch = args.batch
    y = make_batch(torch, batch, n, rank * batch, device)
    ctx = L.get_context(dev_index)
    ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    if args.variant is not None:
        ctx.set_option("variant", args.variant)
    if args.iters is not None:
        ctx.set_option("stft_iters", args.iters)
    window = np.asarray(filters.get_window("hann", N_FFT, fftbins=True), dtype=np.float32)
    plan = ctx.stft_plan(N_FFT, HOP, window, True, "constant", np.float32)
    mel_plan = ctx.mel_plan(filters.mel(sr=SR, n_fft=N_FFT, n_mels=N_MELS))
    n_frames = ctx.stft_num_frames(plan, n)
    frames_per_step = batch * n_frames
    M = torch.empty((batch, N_MELS, n_frames), dtype=torch.float32, device=device)
    D = torch.empty((batch, n_frames, N_FFT // 2 + 1), dtype=torch.complex64, device=device)
    yp, Mp, Dp = y.data_ptr(), M.data_ptr(), D.data_ptr()

    def step_mel():
        ctx.melspectrogram_exec(plan, mel_plan, yp, batch, n, n, 2.0, Mp)

    def step_stft():
        ctx.stft_exec(plan, yp, batch, n, n, Dp)

    # ISTFT (BASELINE config 4: stft -> istft round trip); the window sum-square comes from the host
    iplan = ctx.istft_plan(N_FFT, HOP, window, True, np.float32)
    wss_host = filters.window_sumsquare(window="hann", n_frames=n_frames, n_fft=N_FFT, hop_length=HOP, dtype=np.float32)[N_FFT // 2 :]
    wss_host = np.ascontiguousarray(np.pad(wss_host, (0, max(0, n - len(wss_host))))[:n], dtype=np.float32)
    wss = torch.from_numpy(wss_host).to(device)
    yrec = torch.empty((batch, n), dtype=torch.float32, device=device)
    n_bins = N_FFT // 2 + 1

    def step_istft():
        ctx.istft_exec(iplan, Dp, batch, n_frames * n_bins, n_bins, n_frames, wss.data_ptr(), yrec.data_ptr(), n, n)

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        e0, e1 = ctx.event(), ctx.event()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize(device)
        wall = time.perf_counter() - t0
        barrier()
        return wall, e0.elapsed_ms(e1) / 1e3

    if args.sweep and world == 1:
        for variant in (0, 1, 4):
            for iters in (4, 8, 16, 32, 64):
                ctx.set_option("variant", variant)
                ctx.set_option("stft_iters", iters)
                _, ev_m = timed(step_mel, args.steps, 3)
                _, ev_s = timed(step_stft, args.steps, 3)
                fm, fs = frames_per_step * args.steps / ev_m, frames_per_step * args.steps / ev_s
                print(f"[sweep] variant={variant} iters={iters:2d}  mel {fm / 1e6:8.1f} Mframes/s ({fm * BYTES_PER_FRAME_MEL / 1e9:7.0f} GB/s)   "
                      f"stft {fs / 1e6:8.1f} Mframes/s ({fs * BYTES_PER_FRAME_STFT / 1e9:7.0f} GB/s = {fs * BYTES_PER_FRAME_STFT / 1e9 / HBM_PEAK_GBS:.1%} of HBM peak)",
                      file=sys.stderr, flush=True)
        ctx.set_option("variant", args.variant if args.variant is not None else -1)
        ctx.set_option("stft_iters", args.iters or 0)

    wall, ev = timed(step_mel, args.steps, args.warmup)
    if world > 1:
        tw = torch.tensor([wall], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    _, ev_stft = timed(step_stft, args.steps, args.warmup)
    _, ev_istft = timed(step_istft, args.steps, args.warmup)
    # BASELINE config 5 (CQT-lite): three STFTs at n_fft = 512 / 2048 / 8192 over the same batch, shared hop 512
    cqt = None
    if world == 1 and not args.no_cqt:  # single-GPU runs only: timed() contains collective barriers
        try:
            parts = {}
            total_s = 0.0
            for nf in (512, 2048, 8192):
                w = np.asarray(filters.get_window("hann", nf, fftbins=True), dtype=np.float32)
                pl = plan if nf == N_FFT else ctx.stft_plan(nf, HOP, w, True, "constant", np.float32)
                T_nf = ctx.stft_num_frames(pl, n)
                Dn = D if nf == N_FFT else torch.empty((batch, T_nf, nf // 2 + 1), dtype=torch.complex64, device=device)
                _, ev_n = timed(lambda: ctx.stft_exec(pl, yp, batch, n, n, Dn.data_ptr()), max(3, args.steps // 2), 2)
                per = ev_n / max(3, args.steps // 2)
                bytes_n = batch * T_nf * ((nf // 2 + 1) * 8 + HOP * 4)
                parts[str(nf)] = {"ms": per * 1e3, "GBps": bytes_n / per / 1e9}
                total_s += per
                del Dn
            cqt = {"workload": f"3 STFTs n_fft=512/2048/8192, hop=512, batch={batch} x {CLIP_SECONDS} s (BASELINE config 5)", "per_n_fft": parts, "ms_total": total_s * 1e3,
                   "frame_triples_per_s": frames_per_step / total_s}
        except Exception as exc:  # never let the side measurement break the contract line
            cqt = {"error": repr(exc)}
    snr_db = None
    if rank == 0:
        err = (y[:8] - yrec[:8]).double().pow(2).sum(dim=1)
        snr_db = float((10 * torch.log10(y[:8].double().pow(2).sum(dim=1) / err)).min().item())

    if rank == 0:
        total_frames = frames_per_step * args.steps * world
        value = total_frames / wall
        launch_s = ev / args.steps
        achieved = frames_per_step * BYTES_PER_FRAME_MEL / launch_s / 1e9
        stft_launch_s = ev_stft / args.steps
        achieved_stft = frames_per_step * BYTES_PER_FRAME_STFT / stft_launch_s / 1e9
        line = {
            "metric": "STFT+mel frames/sec (n_fft=2048 hop=512)",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"feature.melspectrogram: batch={batch} clips x {CLIP_SECONDS} s @ 22.05 kHz per GPU, n_fft={N_FFT} hop={HOP} n_mels={N_MELS}, "
                                   f"inputs resident in HBM, outputs left sharded (no gather)", "frames_per_step_per_gpu": frames_per_step,
                       "parallelism": f"clips sharded over {world} GPU(s), no collective on the data path", "device": ctx.device_name()},
            "roofline": {"bound": "hbm", "kernel": "stft_kernel<n_fft=2048, OUT_MEL> (fused melspectrogram)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "bytes_per_frame": BYTES_PER_FRAME_MEL,
                         "launch_ms": launch_s * 1e3, "frames_per_s_single_gpu": frames_per_step / launch_s,
                         "note": "this kernel sits on the VALU side of the ridge (~65 kFLOP per 2 560 B); see roofline_stft for the HBM-bound kernel"},
            "roofline_stft": {"bound": "hbm", "kernel": "stft_kernel<n_fft=2048, OUT_COMPLEX> (librosa.stft, complex64 out)", "achieved": achieved_stft,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_stft / HBM_PEAK_GBS, "traffic": None,
                              "bytes_per_frame": BYTES_PER_FRAME_STFT, "launch_ms": stft_launch_s * 1e3, "frames_per_s_single_gpu": frames_per_step / stft_launch_s},
        }
        istft_launch_s = ev_istft / args.steps
        achieved_istft = frames_per_step * BYTES_PER_FRAME_STFT / istft_launch_s / 1e9
        line["roofline_istft"] = {"bound": "hbm", "kernel": "istft_kernel<n_fft=2048> (librosa.istft: c2r FFT + window + overlap-add + wss normalise)", "achieved": achieved_istft,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_istft / HBM_PEAK_GBS, "traffic": None, "bytes_per_frame": BYTES_PER_FRAME_STFT,
                                  "launch_ms": istft_launch_s * 1e3, "frames_per_s_single_gpu": frames_per_step / istft_launch_s, "round_trip_snr_db_min": snr_db}
        # the fused mel kernel sits on the VALU side of the ridge: the same launch against the f32 vector peak
        # (~65 kFLOP per frame by the 5 N log2 N convention; 157.3 TFLOP/s = 256 CUs x 256 flop/clk x 2.4 GHz,
        # reachable only with packed FMAs -- an FFT is mostly packed adds, 2 flop per lane-instruction instead of 4)
        line["roofline_valu"] = {"bound": "valu", "kernel": line["roofline"]["kernel"], "achieved": frames_per_step * 65e3 / launch_s / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                                 "frac": frames_per_step * 65e3 / launch_s / 1e12 / 157.3, "flop_per_frame": 65e3}
        line["kernel_variants"] = {"stft": ctx.tuned_variant(plan, 0), "melspectrogram": ctx.tuned_variant(plan, 2), "istft": ctx.tuned_variant(iplan),
                                   "note": "n_fft=2048 f32: 0 = one wave64 per frame (16 points/thread), 4 = two waves per frame (8 points/thread); chosen by timing both on the first call (ctx option autotune), -1 = pinned or default"}
        line["roofline_stft"]["achievable_note"] = ("scripts/storepat.hip (same 2 048 B read + 8 200 B write per row, no arithmetic, no LDS) reaches 4.6-4.8 TB/s on this "
                                                    "chip, a plain copy 4.8 TB/s: that, not the 8 TB/s pin rate, is what this store stream can reach")
        if cqt is not None:
            line["cqt_lite"] = cqt
        # HBM bytes per launch measured with rocprofv3 PMC passes (scripts/profile_round.sh), when committed
        try:
            prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))
            if prof:
                tj = json.load(open(os.path.join(ROOT, "profiles", prof[-1])))
                for key, needle in (("roofline", "mel"), ("roofline_stft", "complex64"), ("roofline_istft", "istft")):
                    # several variants of a kernel may appear (autotune candidates): the one launched most is the one timed here
                    cands = [(v.get("launches", 0), kname, v) for kname, v in tj.get("kernels", {}).items() if needle in kname and "n_fft=2048" in kname and v.get("hbm_bytes")]
                    if cands:
                        _, kname, v = max(cands, key=lambda c: c[0])
                        line[key]["traffic"] = v["hbm_bytes"]
                        line[key]["traffic_source"] = f"profiles/{prof[-1]} ({kname})"
        except Exception:
            pass
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            try:
                line["cpu_baseline_all_cores"] = cpu_baseline_all_cores()
            except Exception as exc:  # the single-core object above is the contract; this one is informative
                line["cpu_baseline_all_cores"] = {"error": repr(exc)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
