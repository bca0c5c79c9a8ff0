#!/usr/bin/env python
"""bench.py -- STFT+mel frames/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        # no launcher around it: starts the N ranks itself (the command above), refuses when fewer devices are visible

A "step" is one pass of the hot path -- feature.melspectrogram (fused framing + window + FFT + |X|^2 + two-slope mel,
one kernel launch) -- over one batch of synthetic 22.05 kHz clips already resident in HBM.  Workload:
  N = 1   BASELINE.json configs[1]: 256 clips x 30 s, n_fft=2048 hop=512 n_mels=128;
  N > 1   BASELINE.json configs[2]'s split: 512 clips per GPU (4096 clips at N = 8), clip i on GPU i // 512, no collective
          on the data path; `value` is the sharded-output rate, `gathered` the rate with every rank holding the full
          (N x 512, 128, 1292) result, the gather (RCCL over xGMI) travelling in chunks of clips behind the compute.
Plans/tables are created outside the timed region.  The timed region is EXACTLY --steps steps bracketed by a
barrier + device synchronize on both sides; the max over ranks is reported.

Output (rank 0): ONE short JSON line on stdout -- the contract's keys, `roofline` (timed kernel) with the path's other fractions folded in as SCALARS of `roofline`
(stft_frac / istft_frac / stream_forward_frac / stream_inverse_frac / cqt_lite_frac, stft_v2_frac / stft_v3_frac / mel_pc_ms of `kernel_forms`, + the best / worst of `--placements` allocations of the 2.7 GB spectrum: its
placement moves the store-bound transform 0.63-0.75 ms, profiles/r05_pitch.md), `cpu_baseline`, `parity`, `scaling_base` (N = 1 at 512 clips per GPU), one number per
side measurement (`side`) -- and the FULL record on stderr (and in --detail / gpurun_out/bench_detail.json).  N > 1: `gathered.full_matches_unsharded` = every rank
recomputed its neighbour's shard from that shard's seeded input and found it equal in the gathered tensor.

Keys of the full record (rank 0):
  roofline        dominant kernel of the step (the fused mel kernel): algorithmic bytes / HIP-event time, + HBM traffic from
                  the newest profiles/*_traffic.json (rocprofv3 PMC passes, scripts/profile_round.sh)
  roofline_stft   the complex64-out STFT kernel on the same input (north-star bar: >= 70 % of HBM, 10 248 B/frame)
  roofline_istft  the inverse on the STFT's output (BASELINE configs[3]; round-trip SNR included)
  roofline_valu   the fused mel kernel against the f32 vector peak (it sits on the compute side of the ridge)
  kernel_forms    round 6: both forms of the complex STFT (radices 16-8-8 / 16-16-4) and of the fused mel kernel (one wave per frame / producer + consumer waves), alternating, same buffers
  repeats         min / median ms per step over 5 more repeats of the timed region (box-to-box and run-to-run spread)
  parity          max relative error of a sampled clip's mel spectrogram against the CPU oracle (same input, downloaded)
  dropin_torch    the PUBLIC drop-in (librosa_amd.feature.melspectrogram on a device tensor: validation, plan cache, lock)
  end_to_end_numpy  the public drop-in on NumPy input: H2D + kernel + D2H (PCIe-bound; informative, never `value`)
  power_to_db / mfcc  the SURVEY 8(f) consumers on the same batch (device tensors)
  pcen_cqt        pcen on the mel batch; the true constant-Q transform (84 bins) of 64 clips, device tensors
  cqt_lite        BASELINE configs[4]: STFTs at n_fft 512 / 2048 / 8192 over the same batch (N=1 only)
  stream_ceiling  the forward / inverse access streams WITHOUT arithmetic (lra_probe_stream), same box, batch and clock ramp: what the
                  transforms' rates are to be read against (`transform_over_stream`)
  cpu_baseline    the reference's own CPU path (unmodified librosa through oracle/ref_shim.py, packed for the GPU box by oracle/make_ref.py:
                  kind "reference") with the NumPy port beside it, on this box's host cores, rank 0 at N=1 only, on a bounded sample of the
                  same workload (one core); cpu_baseline_all_cores = one independent process per core
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR, N_FFT, HOP, N_MELS = 22050, 2048, 512, 128
CLIP_SECONDS = 30
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); what this access mix reaches without arithmetic: profiles/r02_store_stream.md
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


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or platform.machine()


def _host_versions():
    import scipy

    v = {"numpy": np.__version__, "scipy": scipy.__version__, "python": platform.python_version(), "cpu_model": _cpu_model(), "logical_cores": os.cpu_count()}
    try:
        from threadpoolctl import threadpool_info

        v["blas"] = sorted({f"{i.get('internal_api')} {i.get('version')}" for i in threadpool_info() if i.get("user_api") == "blas"})
    except Exception:  # pragma: no cover
        pass
    return v


def _load_reference():
    """The UNMODIFIED reference package through oracle/ref_shim.py: from /root/reference where it exists (the build container), else from
    the archive oracle/_ref/librosa_ref.zip that __graft_entry__.build() packs there (oracle/make_ref.py) and that travels with the
    repository snapshot.  None when neither exists: the baseline then falls back to the NumPy port and says so."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import ref_shim

        if not ref_shim.available():
            return None
        return ref_shim.load_reference()
    except Exception:  # pragma: no cover
        return None


def _limit_blas():
    import contextlib

    try:
        from threadpoolctl import threadpool_limits

        return threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        return contextlib.nullcontext()


def _time_mel(fn, y, seconds):
    fn(y[0])  # warm (tables, imports)
    frames = clips = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        frames += fn(y[clips % len(y)]).shape[-1]
        clips += 1
    return frames, clips, time.perf_counter() - t0


def cpu_baseline(seconds=10.0, parity_clip=None):
    """The reference's own CPU path (librosa.feature.melspectrogram: core/spectrum.py:57-391, feature/spectral.py:2022-2161) timed on this
    host, 1 core: `kind` = "reference" when the reference package is available (see _load_reference), with the NumPy port
    (oracle/stft_oracle.py) timed beside it; "port" alone otherwise.

    ``parity_clip = (y_host, M_gpu_host)``: the same leg also checks the timed kernel's output for that clip."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import stft_oracle as O

    ref = _load_reference()
    y = O.config_input(4, n=SR * CLIP_SECONDS)
    kw = dict(sr=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS)
    with _limit_blas():
        pf, pc, pdt = _time_mel(lambda c: O.melspectrogram(y=c, **kw), y, seconds if ref is None else seconds / 2)
        if ref is not None:
            rf, rc, rdt = _time_mel(lambda c: ref.feature.melspectrogram(y=c, **kw), y, seconds)
    port = {"value": pf / pdt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{pc} clips x {CLIP_SECONDS} s melspectrogram in {pdt:.1f} s, oracle/stft_oracle.py, 1 process, BLAS limited to 1 thread"}
    if ref is not None:
        out = {"value": rf / rdt, "unit": "frames/s", "cores": 1, "kind": "reference",
               "sample": f"{rc} clips x {CLIP_SECONDS} s through the unmodified librosa.feature.melspectrogram (n_fft={N_FFT} hop={HOP} n_mels={N_MELS}) in {rdt:.1f} s, "
                         f"1 process, BLAS limited to 1 thread, numba stubbed (plain NumPy bodies; affects istft / window_sumsquare only, not this path)",
               "reference_root": "oracle/_ref/librosa_ref.zip (packed by oracle/make_ref.py)" if not os.path.isdir("/root/reference") else "/root/reference",
               "port": port, "port_vs_reference": port["value"] / (rf / rdt)}
    else:
        out = port
        try:  # the port against the reference, measured where the reference tree exists (the build container), committed
            out["port_vs_reference"] = json.load(open(os.path.join(ROOT, "profiles", "cpu_port_vs_reference.json")))
        except Exception:
            pass
    out["host"] = _host_versions()
    parity = None
    if parity_clip is not None:
        yh, Mg = parity_clip
        Mref = O.melspectrogram(y=yh, **kw)
        rel = np.abs(Mg - Mref) / np.abs(Mref)
        parity = {"mel_max_rel_err": float(rel.max()), "mel_max_abs_err_over_max": float(np.abs(Mg - Mref).max() / Mref.max()), "bar": 1e-4,
                  "sample": "clip 0 of the timed batch (downloaded), all 128 x 1292 values, pure relative error |d| / |ref| against the oracle"}
        if ref is not None:
            Mr = ref.feature.melspectrogram(y=yh, **kw)
            parity["mel_max_rel_err_vs_reference"] = float((np.abs(Mg - Mr) / np.abs(Mr)).max())
            parity["oracle_equals_reference"] = bool(np.array_equal(Mr, Mref))
    return out, parity


def _cpu_worker(seconds):
    """One host process of the all-cores baseline (spawned: no torch / HIP state in the child)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import stft_oracle as O

    ref = _load_reference()
    y = O.config_input(2, n=SR * CLIP_SECONDS)
    kw = dict(sr=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS)
    with _limit_blas():
        frames, _, dt = _time_mel((lambda c: ref.feature.melspectrogram(y=c, **kw)) if ref is not None else (lambda c: O.melspectrogram(y=c, **kw)), y, seconds)
    return frames, dt, ref is not None


def cpu_baseline_all_cores(seconds=8.0, max_procs=None):
    """The reference has no internal parallelism: its fair multi-core mode is one independent process per core
    (P = os.cpu_count()).  P spawned processes, each looping over clips for `seconds`; P is lowered only where the
    host's free memory would not hold that many interpreters (~0.6 GB each with NumPy / SciPy and a 30 s clip's float64 spectra), and the line says so."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor

    procs = max(1, os.cpu_count() or 1)
    if max_procs:
        procs = min(procs, max_procs)
    try:
        import psutil

        procs = max(1, min(procs, int(psutil.virtual_memory().available * 0.5 // (600 << 20))))
    except Exception:  # pragma: no cover
        procs = min(procs, 64)
    with ProcessPoolExecutor(max_workers=procs, mp_context=mp.get_context("spawn")) as pool:
        res = list(pool.map(_cpu_worker, [seconds] * procs))
    kind = "reference" if all(r[2] for r in res) else "port"
    return {"value": sum(f / dt for f, dt, _ in res), "unit": "frames/s", "cores": procs, "kind": kind,
            "logical_cores": os.cpu_count(),
            "sample": f"{procs} independent processes x {seconds:.0f} s of 30 s-clip melspectrograms (n_fft={N_FFT} hop={HOP} n_mels={N_MELS}), 1 BLAS thread each; host has {os.cpu_count()} logical cores"}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, the way the driver's multi-GPU command does
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...), and pass its
    exit code on.  Fails loudly when the node shows fewer than N devices (LRA_BENCH_BACKEND=gloo: ranks share devices, the CPU test)."""
    import socket
    import subprocess

    backend = os.environ.get("LRA_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        import torch

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.exit(f"bench.py --gpus {args.gpus}: only {have} ROCm device(s) visible on this node; refusing to report n_gpus={args.gpus}")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, LRA_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def dry_run(args, torch, world, rank):
    """--dry-run: everything around the step (see the flag's help).  No kernel, no library, no GPU; never a measurement."""
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    batch = args.batch if args.batch else (256 if world == 1 else 512)
    n_frames = 1 + SR * CLIP_SECONDS // HOP
    for _ in range(args.warmup):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    wall = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        tw = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    if rank == 0:
        print(json.dumps({"metric": "STFT+mel frames/sec (n_fft=2048 hop=512)", "value": None, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "dry-run (no kernel executed, not a measurement)",
                          "config": {"workload": "dry run of the launch / barrier / reduction control flow", "clips_per_gpu": batch, "frames_per_step_per_gpu": batch * n_frames,
                                     "self_launched": bool(os.environ.get("LRA_BENCH_SELF_LAUNCHED"))}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def compact_line(line):
    """The stdout line: the contract's keys, `roofline` with the path's other fractions folded in (`roofline.path`), `cpu_baseline`, and one number per side
    measurement -- prose and per-repeat lists stay in the full record (stderr / --detail; the keys are described in this file's docstring and DESIGN.md 6)."""
    def get(d, *keys, default=None):
        for k in keys:
            if not isinstance(d, dict) or k not in d:
                return default
            d = d[k]
        return d

    def r(x, nd=4):
        return round(x, nd) if isinstance(x, float) else x

    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in line}
    cfg = line.get("config", {})
    out["config"] = {"workload": f"feature.melspectrogram, {cfg.get('clips_per_gpu')} clips x {CLIP_SECONDS} s per GPU @ 22.05 kHz, n_fft={N_FFT} hop={HOP} n_mels={N_MELS}, "
                                 f"{'BASELINE configs[1]' if line.get('n_gpus') == 1 else 'BASELINE configs[2] split, clip i on GPU i // ' + str(cfg.get('clips_per_gpu'))}, inputs resident in HBM",
                     "clips_per_gpu": cfg.get("clips_per_gpu"), "frames_per_step_per_gpu": cfg.get("frames_per_step_per_gpu"), "prewarm_ms": cfg.get("prewarm_ms"),
                     "parallelism": f"clips sharded over {line.get('n_gpus')} GPU(s), no data-path collective", "device": cfg.get("device")}
    rf = line.get("roofline", {})
    roof = {k: r(rf.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_box", "bytes_per_frame", "launch_ms") if k in rf}
    roof["valu_frac"] = r(get(line, "roofline_valu", "frac"))
    path = {"stft_frac": get(line, "roofline_stft", "frac"), "stft_ms": get(line, "roofline_stft", "launch_ms"), "istft_frac": get(line, "roofline_istft", "frac"),
            "istft_ms": get(line, "roofline_istft", "launch_ms"), "stream_forward_frac": get(line, "stream_ceiling", "forward", "frac"),
            "stream_inverse_frac": get(line, "stream_ceiling", "inverse", "frac"), "cqt_lite_frac": get(line, "cqt_lite", "frac_of_hbm"), "cqt_lite_ms": get(line, "cqt_lite", "ms_total"),
            "stft_frac_best_placement": get(line, "placement", "stft_frac_best"), "stft_frac_worst_placement": get(line, "placement", "stft_frac_worst"),
            "istft_frac_best_placement": get(line, "placement", "istft_frac_best"), "placements": get(line, "placement", "allocations"),
            "stft_traffic": get(line, "roofline_stft", "traffic"), "istft_traffic": get(line, "roofline_istft", "traffic")}
    # (round 6, VERDICT r05 item 4: the driver's record keeps scalars of `roofline`, not a nested object -- the path's fractions are scalars of `roofline` itself)
    path.update({"stft_v2_frac": get(line, "kernel_forms", "stft_radix_16_8_8", "frac"), "stft_v3_frac": get(line, "kernel_forms", "stft_radix_16_16_4", "frac"),
                 "mel_one_wave_ms": get(line, "kernel_forms", "mel_one_wave", "ms"), "mel_pc_ms": get(line, "kernel_forms", "mel_producer_consumer", "ms"),
                 "mel_pc_16_16_4_ms": get(line, "kernel_forms", "mel_producer_consumer_16_16_4", "ms"), "long_clip_vs_batched": get(line, "long_clip", "frac_of_batched"),
                 "stft_frac_best_placed": get(line, "placement_placed", "stft_frac_best"), "stft_frac_worst_placed": get(line, "placement_placed", "stft_frac_worst"),
                 "stft_frac_median_placed": get(line, "placement_placed", "stft_frac_median")})
    for k, v in path.items():
        if v is not None:
            roof[k] = r(v)
    out["roofline"] = roof
    cb = line.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = {k: r(cb.get(k), 1) for k in ("value", "unit", "cores", "kind") if k in cb}
        out["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
        if get(line, "cpu_baseline_all_cores", "value") is not None:
            out["cpu_baseline"]["all_cores_value"] = r(get(line, "cpu_baseline_all_cores", "value"), 1)
            out["cpu_baseline"]["all_cores"] = get(line, "cpu_baseline_all_cores", "cores")
        out["cpu_baseline"]["cpu_model"] = get(cb, "host", "cpu_model") or get(line, "cpu_baseline", "cpu_model")
    par = line.get("parity")
    if isinstance(par, dict):
        out["parity"] = {k: par[k] for k in ("mel_max_rel_err_vs_reference", "mel_max_rel_err", "oracle_equals_reference", "bar") if k in par}
        out["parity"]["round_trip_snr_db_min"] = r(get(line, "roofline_istft", "round_trip_snr_db_min"), 1)
    if "scaling_base" in line:
        out["scaling_base"] = {k: r(v) for k, v in line["scaling_base"].items()} if isinstance(line["scaling_base"], dict) else line["scaling_base"]
    if "gathered" in line and isinstance(line["gathered"], dict):
        out["gathered"] = {k: r(line["gathered"].get(k)) for k in ("value", "unit", "ms_per_step", "chunks", "bytes_per_rank", "backend", "own_rows_match", "full_matches_unsharded", "error") if k in line["gathered"]}
    out["repeats"] = {"ms_per_step_min": r(get(line, "repeats", "ms_per_step_min")), "ms_per_step_median": r(get(line, "repeats", "ms_per_step_median"))}
    side = {"dropin_torch_ms": get(line, "dropin_torch", "ms_per_call"), "numpy_mel_ms_64clips": get(line, "end_to_end_numpy", "melspectrogram", "ms"),
            "numpy_stft_ms_64clips": get(line, "end_to_end_numpy", "stft", "ms"), "power_to_db_ms": get(line, "power_to_db", "ms_per_call"), "mfcc_ms": get(line, "mfcc", "ms_per_call"),
            "griffinlim_ms_per_iter_32clips": get(line, "griffinlim", "ms_per_iteration"), "griffinlim_ms_setup": get(line, "griffinlim", "ms_setup"),
            "cqt_polyphase_ms_64clips": get(line, "pcen_cqt", "cqt_polyphase", "ms_per_call"), "cqt_default_ms_64clips": get(line, "pcen_cqt", "cqt_default", "ms_per_call"), "cqt_polyphase_nocheck_ms_64clips": get(line, "pcen_cqt", "cqt_polyphase_nocheck", "ms_per_call"),
            "pcen_ms": get(line, "pcen_cqt", "pcen", "ms_per_call"), "hpss_ms_32clips": get(line, "hpss", "ms_per_call"), "mixed_400_mel_ms": get(line, "mixed_radix_400", "fused", "mel_ms"),
            "mixed_400_stft_ms": get(line, "mixed_radix_400", "fused", "stft_ms"), "mixed_1200_stft_GBps": get(line, "mixed_radix_1200", "stft_GBps_algorithmic"), "mixed_1200_mel_ms": get(line, "mixed_radix_1200", "mel_ms"),
            "mixed_3200_stft_GBps": get(line, "mixed_radix_3200", "stft_GBps_algorithmic"), "mixed_3200_mel_ms": get(line, "mixed_radix_3200", "mel_ms"),
            "mixed_1200_istft_ms": get(line, "mixed_radix_1200", "istft_ms"), "mixed_3200_istft_ms": get(line, "mixed_radix_3200", "istft_ms"), "speech_512_mel_ms": get(line, "speech_512", "mel_ms"), "cqt_lite_512_ms": get(line, "cqt_lite", "per_n_fft", "512", "ms"),
            "cqt_lite_8192_ms": get(line, "cqt_lite", "per_n_fft", "8192", "ms"), "cqt_lite_default_hop_ms": get(line, "cqt_lite", "default_hop_variant", "ms_total"),
            "stft_power_w": get(line, "board_power", "stft", "socket_power_w"), "stft_sclk_mhz": get(line, "board_power", "stft", "sclk_mhz"),
            "mel_power_w": get(line, "board_power", "mel", "socket_power_w"), "mel_variant": get(line, "kernel_variants", "melspectrogram")}
    out["side"] = {k: r(v) for k, v in side.items() if v is not None}
    errs = [k for k, v in line.items() if isinstance(v, dict) and "error" in v]
    if errs:
        out["side_errors"] = errs
    out["detail"] = "full record: stderr of this run (and --detail / gpurun_out/bench_detail.json); keys: bench.py docstring, DESIGN.md 6"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--prewarm-ms", type=float, default=400.0, help="untimed run of the timed launch before the warm-up steps (clock ramp)")
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: 256 at N=1 = configs[1], 512 at N>1 = configs[2]'s split)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cqt", action="store_true", help="skip the CQT-lite (config 5) side measurement")
    ap.add_argument("--no-side", action="store_true", help="skip every side measurement (profiling runs)")
    ap.add_argument("--no-scaling-base", action="store_true", help="skip the N = 1 run at 512 clips (profiling runs: it launches the headline kernel on another grid)")
    ap.add_argument("--no-power", action="store_true", help="skip the rocm-smi power / clock samples (profiling runs: they loop the kernels for seconds)")
    ap.add_argument("--placements", type=int, default=5, help="allocations of the 2.7 GB spectrum the complex STFT / ISTFT side figures are sampled over (median reported, best / worst beside it)")
    ap.add_argument("--detail", default=None, help="file the full (long) measurement record goes to; default gpurun_out/bench_detail.json when that directory exists; it always goes to stderr too")
    ap.add_argument("--gather-chunks", type=int, default=4, help="pieces the shard travels in during the `gathered` measurement (N>1)")
    ap.add_argument("--dry-run", action="store_true", help="control-flow check without a GPU (tests/test_distributed_cpu.py): launch, rendezvous, barriers, max over ranks and "
                                                            "the JSON line with the step replaced by a 1 ms host sleep; `value` is null and `data` says so")
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--iters", type=int, default=None)
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # (does not return)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: one rank per GPU, the two must agree")

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return dry_run(args, torch, world, rank)

    import librosa_amd as L
    from librosa_amd import filters
    from librosa_amd.distributed import ShardedGather, chunk_ranges

    assert torch.cuda.is_available(), "bench.py needs a ROCm device (no CPU fallback)"
    # LRA_BENCH_BACKEND=gloo: control-flow check of the N > 1 path on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("LRA_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)  # nccl = RCCL: barriers, the max-over-ranks of the timing, and the `gathered` measurement

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    n = SR * CLIP_SECONDS
    batch = args.batch if args.batch else (256 if world == 1 else 512)
    total_clips = batch * world
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
    # the synthetic input is generated LAST, right before the warm-up: plan / table construction above is host work during
    # which an idle GPU drops to its low power state, and the first few milliseconds after that run ~10 % slow
    y = make_batch(torch, batch, n, rank * batch, device)  # clip i of the job lives on rank i // batch
    yp, Mp = y.data_ptr(), M.data_ptr()

    def step_mel():
        ctx.melspectrogram_exec(plan, mel_plan, yp, batch, n, n, 2.0, Mp)

    def timed(fn, steps, warmup, collective=True, ramp_ms=0.0):
        # side measurements start after host-side set-up during which the GPU clocked down: same untimed ramp as the headline
        t_ramp = time.perf_counter()
        while ramp_ms and time.perf_counter() - t_ramp < ramp_ms / 1e3:
            fn()
            torch.cuda.synchronize(device)
        for _ in range(warmup):
            fn()
        barrier() if collective else torch.cuda.synchronize(device)
        e0, e1 = ctx.event(), ctx.event()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize(device)
        wall = time.perf_counter() - t0
        if collective:
            barrier()
        return wall, e0.elapsed_ms(e1) / 1e3

    # An idle MI355X sits in a low power state and its first ~0.2 s of work run up to 10 % slow (the driver's default
    # --warmup 5 is 4 ms of work): bring the clocks up with the same launch, untimed and reported as `prewarm_ms`, before the
    # contract's own --warmup steps.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_ms / 1e3:
        for _ in range(20):
            step_mel()
        torch.cuda.synchronize(device)
    # ---- the timed region of the contract: exactly --steps steps, barrier + synchronize on both sides, max over ranks -------
    wall, ev = timed(step_mel, args.steps, args.warmup)
    if world > 1:
        tw = torch.tensor([wall], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    # five more repeats of the same region (no collectives): run-to-run spread
    rep = [timed(step_mel, args.steps, 0, collective=False)[1] / args.steps * 1e3 for _ in range(5)]

    side = {}
    only = set(filter(None, os.environ.get("LRA_BENCH_ONLY", "").split(",")))  # debugging aid: run only these side measurements

    def measure(name, fn):
        """A side measurement never breaks the contract line."""
        if args.no_side or (only and name not in only):
            return
        try:
            side[name] = fn()
        except Exception as exc:  # pragma: no cover
            side[name] = {"error": repr(exc)}

    # ---- N > 1: the same step with every rank ending up with the full result (chunked gather behind the compute) --------------
    if world > 1 and not args.no_side:
        pieces = chunk_ranges(batch, args.gather_chunks)

        def step_gathered():
            g = ShardedGather(M, total_clips)
            for lo, hi in pieces:
                ctx.melspectrogram_exec(plan, mel_plan, yp + lo * n * 4, hi - lo, n, n, 2.0, Mp + lo * N_MELS * n_frames * 4)
                g.push(lo, hi, M[lo:hi])  # async: travels while the next piece is computed
            return g.wait()

        g_steps = max(3, min(10, args.steps // 5))
        g_wall, _ = timed(step_gathered, g_steps, 2)
        tw = torch.tensor([g_wall], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        g_wall = float(tw.item())
        full = step_gathered()
        ok = bool(torch.equal(full[rank * batch : (rank + 1) * batch], M))
        # ... and the rows that came from elsewhere: every rank recomputes its neighbour's shard from that shard's own (seeded) input -- the
        # gathered tensor must equal the unsharded call clip for clip (the reference's batch == per item property, tests/test_multichannel.py:96-111)
        other = (rank + 1) % world
        y_o = make_batch(torch, batch, n, other * batch, device)
        M_o = torch.empty_like(M)
        ctx.melspectrogram_exec(plan, mel_plan, y_o.data_ptr(), batch, n, n, 2.0, M_o.data_ptr())
        ok_other = bool(torch.equal(full[other * batch : (other + 1) * batch], M_o))
        del y_o, M_o
        tf = torch.tensor([int(ok), int(ok_other)], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tf, op=dist.ReduceOp.MIN)
        ok, ok_other = bool(tf[0].item()), bool(tf[1].item())
        side["gathered"] = {"value": frames_per_step * world * g_steps / g_wall, "unit": "frames/s", "ms_per_step": g_wall / g_steps * 1e3, "steps": g_steps, "chunks": len(pieces),
                            "bytes_per_rank": batch * N_MELS * n_frames * 4, "backend": backend, "own_rows_match": ok, "full_matches_unsharded": ok_other,
                            "what": "every rank holds the full (N x clips, 128, frames) mel tensor; the shard is computed and all-gathered in chunks of clips (librosa_amd.distributed.ShardedGather)"}
        del full

    # ---- single-GPU side measurements (their timing helpers contain no collectives) -------------------------------------------
    snr_db = None
    if world == 1 and not args.no_side:
        # The complex spectrum is 2.7 GB, and WHERE that allocation lands moves the store-bound transform between 0.63 and 0.75 ms inside one process
        # (profiles/r05_pitch.md): one allocation is a draw.  Several are made, the transform is timed on each, and everything below runs on the
        # MEDIAN one; best / worst are reported beside it (`placement`).
        n_bins = N_FFT // 2 + 1
        D_cands = [torch.empty((batch, n_frames, n_bins), dtype=torch.complex64, device=device) for _ in range(max(1, args.placements))]
        D = D_cands[0]
        Dp = D.data_ptr()
        iplan = ctx.istft_plan(N_FFT, HOP, window, True, np.float32)
        wss_host = filters.window_sumsquare(window="hann", n_frames=n_frames, n_fft=N_FFT, hop_length=HOP, dtype=np.float32)[N_FFT // 2 :]
        wss_host = np.ascontiguousarray(np.pad(wss_host, (0, max(0, n - len(wss_host))))[:n], dtype=np.float32)
        from librosa_amd.core.spectrum import wss_to_norm

        wss = torch.from_numpy(wss_to_norm(wss_host)).to(device)  # the factor form the public istft hands to the kernels (1 / wss where wss > tiny)
        yrec = torch.empty((batch, n), dtype=torch.float32, device=device)
        if ctx.placement_retry > 0:  # (what `istft(<device tensor>)` returns since round 6: the inverse's rate follows where its OUTPUT lands, profiles/r06_experiments.md 3)
            try:
                from librosa_amd import _arrays as _arr

                yrec = _arr._placed_tensor(ctx, (batch, n), np.dtype(np.float32), device, flat=True)
            except Exception:  # pragma: no cover
                pass

        def roof(fn, bytes_per_frame, kernel, read_bytes):
            _, e = timed(fn, args.steps, args.warmup, collective=False, ramp_ms=args.prewarm_ms / 2)
            s = e / args.steps
            ach = frames_per_step * bytes_per_frame / s / 1e9
            return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                    "bytes_per_frame": bytes_per_frame, "launch_ms": s * 1e3, "frames_per_s_single_gpu": frames_per_step / s,
                    # the north star words its bar as "HBM-read roofline": the compulsory READ bytes alone, next to the write side
                    "read_bytes_per_frame": read_bytes, "read_frac": frames_per_step * read_bytes / s / 1e9 / HBM_PEAK_GBS,
                    "write_frac": frames_per_step * (bytes_per_frame - read_bytes) / s / 1e9 / HBM_PEAK_GBS,
                    "call_ms": s * 1e3, "launches_per_call": 1}

        placement = None
        if len(D_cands) > 1:
            try:
                ms_f, ms_i = [], []
                for Dc in D_cands:
                    p = Dc.data_ptr()
                    _, e = timed(lambda: ctx.stft_exec(plan, yp, batch, n, n, p), 10, 3, collective=False, ramp_ms=args.prewarm_ms / 4)
                    ms_f.append(e / 10 * 1e3)
                    _, e = timed(lambda: ctx.istft_exec_norm(iplan, p, batch, n_frames * n_bins, n_bins, n_frames, wss.data_ptr(), yrec.data_ptr(), n, n), 10, 3, collective=False)
                    ms_i.append(e / 10 * 1e3)
                order = sorted(range(len(D_cands)), key=lambda i: ms_f[i])
                pick = order[len(order) // 2]
                D = D_cands[pick]
                Dp = D.data_ptr()
                to_frac = lambda ms: frames_per_step * BYTES_PER_FRAME_STFT / (ms / 1e3) / 1e9 / HBM_PEAK_GBS
                placement = {"allocations": len(D_cands), "picked": "median by stft time", "stft_ms": ms_f, "istft_ms": ms_i, "stft_frac_best": to_frac(min(ms_f)), "stft_frac_worst": to_frac(max(ms_f)),
                             "istft_frac_best": to_frac(min(ms_i)), "istft_frac_worst": to_frac(max(ms_i))}
            except Exception as exc:  # pragma: no cover
                placement = {"error": repr(exc)}
        D_cands = [D]  # (the others go back to the allocator)
        if placement is not None:
            side["placement"] = placement
        # What `stft(<device tensor>)` returns since round 6: a buffer from lra_malloc_placed (ctx.placement_retry, default 4; csrc/lra_api.hip) -- the contract
        # figures below (roofline_stft / _istft, the stream probes) are measured on such a buffer; `placement` above stays the raw torch.empty lottery.
        stft_buffer = "torch.empty, the median of `placement`"
        if ctx.placement_retry > 0:
            try:
                from librosa_amd import _arrays

                D = _arrays._placed_tensor(ctx, (batch, n_frames, n_bins), np.dtype(np.complex64), device)
                D_cands = [D]
                Dp = D.data_ptr()
                stft_buffer = f"lra_malloc_placed (library default for results >= 256 MB; best of <= {ctx.placement_retry} candidates, tried {ctx._placed_log[-1][3]})"
            except Exception as exc:  # pragma: no cover
                stft_buffer += f" (lra_malloc_placed failed: {exc!r})"

        def placed_allocations():
            """The same five-allocation experiment through lra_malloc_placed (ctx option placement_retry = 4: each buffer the best of up to four candidates built
            from 64 MiB physical handles, judged by the kernels' write stream; round 6): what `stft(<device tensor>)` results get with
            the option on.  One buffer at a time is alive besides its candidates."""
            from librosa_amd import _arrays

            old = ctx.placement_retry
            ctx.set_option("placement_retry", 4)
            try:
                ms_f, tries, alloc_ms = [], [], []
                for _ in range(max(1, args.placements)):
                    torch.cuda.synchronize(device)
                    t0 = time.perf_counter()
                    t = _arrays._placed_tensor(ctx, (batch, n_frames, n_bins), np.dtype(np.complex64), device)
                    alloc_ms.append((time.perf_counter() - t0) * 1e3)
                    tries.append(ctx._placed_log[-1][3] if ctx._placed_log else None)
                    p = t.data_ptr()
                    _, e = timed(lambda: ctx.stft_exec(plan, yp, batch, n, n, p), 10, 3, collective=False, ramp_ms=args.prewarm_ms / 4)
                    ms_f.append(e / 10 * 1e3)
                    del t
                    ctx.placed_release_all()  # (a recycled buffer would just repeat its own figure: every round starts from fresh candidates)
                to_frac = lambda ms: frames_per_step * BYTES_PER_FRAME_STFT / (ms / 1e3) / 1e9 / HBM_PEAK_GBS
                return {"allocations": len(ms_f), "stft_ms": ms_f, "candidates_tried": tries, "alloc_ms": alloc_ms, "stft_frac_best": to_frac(min(ms_f)), "stft_frac_worst": to_frac(max(ms_f)),
                        "stft_frac_median": to_frac(statistics.median(ms_f)),
                        "what": "complex STFT on buffers from lra_malloc_placed (best of <= 4 candidates each; early exit where candidates agree within 1.5 %); alloc_ms includes the probes"}
            finally:
                ctx.set_option("placement_retry", old)
                ctx.placed_release_all()

        if len(side.get("placement", {}).get("stft_ms", [])) > 1:
            measure("placement_placed", placed_allocations)
        step_stft = lambda: ctx.stft_exec(plan, yp, batch, n, n, Dp)
        step_istft = lambda: ctx.istft_exec_norm(iplan, Dp, batch, n_frames * n_bins, n_bins, n_frames, wss.data_ptr(), yrec.data_ptr(), n, n)
        measure("roofline_stft", lambda: roof(step_stft, BYTES_PER_FRAME_STFT, "stft2_kernel<n_fft=2048 as radices 16-16-4, OUT_COMPLEX> (librosa.stft, complex64 out)", HOP * 4))
        measure("roofline_istft", lambda: roof(step_istft, BYTES_PER_FRAME_STFT, "istft_kernel<n_fft=2048> (librosa.istft: c2r FFT + window + overlap-add + wss normalise)", n_bins * 8))
        for k_ in ("roofline_stft", "roofline_istft"):
            if k_ in side and "error" not in side[k_]:
                side[k_]["spectrum_buffer"] = stft_buffer
        if "roofline_istft" in side and "error" not in side["roofline_istft"]:
            side["roofline_istft"]["call_note"] = ("lra_istft_exec_norm (what librosa_amd.istft calls) = ONE launch since round 4: the kernel stores every sample it covers and the wrapper zeroes only what no frame reaches "
                                                   "(nothing here); round 3's call carried an 85 us hipMemsetAsync of the whole output (677 MB)")

        def kernel_forms():
            """Round 6: the alternative forms of the two contract kernels, timed alternately with the defaults on the SAME buffers (ctx options v3 / mel_pc):
            the complex STFT as radices 16-8-8 (8-byte row pieces) and 16-16-4 (16-byte pieces by butterfly assignment); the fused mel kernel as one wave
            per frame (FFT + epilogue, 240 VGPRs, two waves per SIMD) and as producer / consumer waves (192-thread workgroups, three waves per SIMD)."""
            out = {}
            to_frac = lambda s_: frames_per_step * BYTES_PER_FRAME_STFT / s_ / 1e9 / HBM_PEAK_GBS
            try:
                for rnd in range(2):
                    for name, opt in (("stft_radix_16_8_8", 0), ("stft_radix_16_16_4", 1)):
                        ctx.set_option("v3", opt)
                        _, e = timed(step_stft, 10, 3, collective=False, ramp_ms=args.prewarm_ms / 4)
                        ms = e / 10 * 1e3
                        if name not in out or ms < out[name]["ms"]:
                            out[name] = {"ms": ms, "frac": to_frac(e / 10)}
                    for name, opt in (("mel_one_wave", 0), ("mel_producer_consumer", 1), ("mel_producer_consumer_16_16_4", 2)):
                        ctx.set_option("mel_pc", opt)
                        _, e = timed(step_mel, 10, 3, collective=False, ramp_ms=args.prewarm_ms / 4)
                        ms = e / 10 * 1e3
                        if name not in out or ms < out[name]["ms"]:
                            out[name] = {"ms": ms, "frames_per_s": frames_per_step / (e / 10)}
            finally:
                ctx.set_option("v3", 1)
                ctx.set_option("mel_pc", 1)
            out["what"] = ("best of two alternating rounds of 10 launches each, same input / output buffers as roofline_stft and the timed step; the library's defaults: 16-16-4 for the "
                           "complex STFT, producer / consumer waves on the 16-8-8 core for the fused mel kernel (= the timed step)")
            return out

        measure("kernel_forms", kernel_forms)

        def stream_ceiling():
            """The same access streams WITHOUT arithmetic (lra_probe_stream, csrc/lra_probe.h): what this mix of PCM reads and 8-byte-aligned 8 200-byte row
            writes reaches on THIS box at the forward kernel's residency -- the figure the transform's own rate is to be read against."""
            out = {}
            for key, direction, src, dst in (("forward", 0, yp, Dp), ("inverse", 1, Dp, yrec.data_ptr())):
                fn = lambda: ctx.probe_stream(direction, src, dst, batch, n_frames, N_FFT, HOP, n, 162, 12)
                _, e = timed(fn, args.steps, args.warmup, collective=False, ramp_ms=args.prewarm_ms / 2)
                per = e / args.steps
                ach = frames_per_step * BYTES_PER_FRAME_STFT / per / 1e9
                out[key] = {"launch_ms": per * 1e3, "achieved": ach, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
            for key, rk in (("forward", "roofline_stft"), ("inverse", "roofline_istft")):
                if rk in side and "error" not in side[rk]:
                    out[key]["transform_over_stream"] = side[rk]["achieved"] / out[key]["achieved"]
            out["what"] = ("stream_probe_kernel: one wave64 per strip of 162 rows, per row 2 048 B of PCM and one 8 200-byte row (16 x dwordx2 + the middle bin, the kernels' own order), "
                           "next row's loads ahead of this row's stores, 12 waves per CU, no FFT, no LDS traffic; same batch, same clock ramp, same HIP-event timing as roofline_stft / _istft")
            return out

        measure("stream_ceiling", stream_ceiling)
        def board_power():
            """Socket power and shader clock reported by rocm-smi while each of the three kernels runs in a loop (a second thread keeps the queue full):
            the complex STFT runs at the board's power limit with the clock lowered, profiles/r03_experiments.md section 9."""
            import re
            import shutil
            import subprocess
            import threading

            smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
            if not os.path.exists(smi):
                return {"error": "rocm-smi not found"}

            def parse(txt):
                pw = re.search(r"Power \(W\):\s*([0-9.]+)", txt)
                ck = re.search(r"sclk clock level:\s*\S+\s*\((\d+)Mhz\)", txt)
                return {"socket_power_w": float(pw.group(1)) if pw else None, "sclk_mhz": int(ck.group(1)) if ck else None}

            def sample(fn):
                stop = threading.Event()

                def feed():
                    while not stop.is_set():
                        for _ in range(200):
                            fn()
                        torch.cuda.synchronize(device)

                th = threading.Thread(target=feed, daemon=True)
                th.start()
                try:
                    time.sleep(2.5)  # (the power controller takes a second or two to settle on a clock)
                    txt = subprocess.run([smi, "--showpower", "--showclocks"], capture_output=True, text=True, timeout=30).stdout
                finally:
                    stop.set()
                    th.join(timeout=30)
                    torch.cuda.synchronize(device)
                return parse(txt)

            return {"mel": sample(step_mel), "stft": sample(step_stft), "istft": sample(step_istft),
                    "stream_forward": sample(lambda: ctx.probe_stream(0, yp, Dp, batch, n_frames, N_FFT, HOP, n, 162, 12)),
                    "what": "rocm-smi --showpower --showclocks sampled once, 2.5 s into a loop of the kernel (MI355X board limit: 1400 W; shader clock at most 2400 MHz)"}

        if not args.no_power:
            measure("board_power", board_power)
        try:
            err = (y[:8] - yrec[:8]).double().pow(2).sum(dim=1)
            snr_db = float((10 * torch.log10(y[:8].double().pow(2).sum(dim=1) / err)).min().item())
            side["roofline_istft"]["round_trip_snr_db_min"] = snr_db
        except Exception:
            pass
        del yrec

        def public_torch():
            fn = lambda: L.feature.melspectrogram(y=y, sr=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS, check_finite=False)
            _, e = timed(fn, 10, 3, collective=False, ramp_ms=args.prewarm_ms / 4)
            fn2 = lambda: L.feature.melspectrogram(y=y, sr=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS)
            _, e2 = timed(fn2, 10, 3, collective=False, ramp_ms=args.prewarm_ms / 4)
            return {"ms_per_call": e / 10 * 1e3, "frames_per_s": frames_per_step / (e / 10), "ms_per_call_with_finite_check": e2 / 10 * 1e3,
                    "what": "librosa_amd.feature.melspectrogram(y=<device tensor>): argument validation, plan-cache lookup, per-context lock, output allocation + the kernel"}

        measure("dropin_torch", public_torch)

        def scaling_base():
            """The N = 1 figure at the per-GPU work of the N > 1 runs (512 clips = BASELINE configs[2]'s split), so that a 1 -> 8 curve compares equal per-GPU work."""
            b2 = 512
            y2 = make_batch(torch, b2, n, 0, device)
            M2 = torch.empty((b2, N_MELS, n_frames), dtype=torch.float32, device=device)
            fn = lambda: ctx.melspectrogram_exec(plan, mel_plan, y2.data_ptr(), b2, n, n, 2.0, M2.data_ptr())
            _, e = timed(fn, args.steps, args.warmup, collective=False, ramp_ms=args.prewarm_ms / 2)
            per = e / args.steps
            return {"clips_per_gpu": b2, "value": b2 * n_frames / per, "unit": "frames/s", "ms_per_step": per * 1e3}

        if batch != 512 and not args.no_scaling_base:
            measure("scaling_base", scaling_base)

        def public_numpy():
            nb = 64  # a quarter of the batch: 169 MB up, 42 MB down
            yh = y[:nb].cpu().numpy()
            # first calls (untimed): the context sizes its pinned staging / device buffers for this batch; they persist
            t0 = time.perf_counter()
            L.feature.melspectrogram(y=yh, sr=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS)
            L.stft(yh, n_fft=N_FFT, hop_length=HOP)
            first = time.perf_counter() - t0
            t0 = time.perf_counter()
            Mh = L.feature.melspectrogram(y=yh, sr=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS)
            dt = time.perf_counter() - t0
            t0 = time.perf_counter()
            Dh = L.stft(yh, n_fft=N_FFT, hop_length=HOP)
            dts = time.perf_counter() - t0
            L.istft(Dh, hop_length=HOP, length=yh.shape[-1])
            t0 = time.perf_counter()
            yi = L.istft(Dh, hop_length=HOP, length=yh.shape[-1])
            dti = time.perf_counter() - t0
            return {"melspectrogram": {"clips": nb, "ms": dt * 1e3, "frames_per_s": nb * n_frames / dt, "host_bytes": int(yh.nbytes + Mh.nbytes)},
                    "stft": {"clips": nb, "ms": dts * 1e3, "frames_per_s": nb * n_frames / dts, "host_bytes": int(yh.nbytes + Dh.nbytes)},
                    "istft": {"clips": nb, "ms": dti * 1e3, "frames_per_s": nb * n_frames / dti, "host_bytes": int(yi.nbytes + Dh.nbytes)},
                    "first_calls_ms": first * 1e3,
                    "what": "public drop-in on np.ndarray through the native host pipeline (lra_stft_exec_host: pinned two-slot staging, finite scan fused into the staging copy, "
                            "upload / kernel / download overlapped); steady state, `first_calls_ms` = the two calls that sized the staging buffers; informative only"}

        measure("end_to_end_numpy", public_numpy)

        def long_clip():
            """ONE long clip (1 x 1 h @ 22.05 kHz = 79.4 M samples) through the NumPy drop-in: a call with fewer clips than devices shards by FRAMES
            (librosa_amd.distributed.shard_frames; opt-in LRA_DEVICES).  On this 1-GPU box the two frame ranges both run on device 0 -- what is checked is that
            the sharded result equals the unsharded one bit for bit, and what is timed is the long clip against the same number of samples as a batch of clips."""
            from librosa_amd.core import spectrum as SP

            secs = 3600
            rng = np.random.default_rng(440)
            yl = (0.1 * rng.standard_normal(SR * secs, dtype=np.float32) + 0.5 * np.sin(2 * np.pi * 440.0 / SR * np.arange(SR * secs, dtype=np.float32))).astype(np.float32)[None, :]
            kw = dict(sr=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS)
            old = os.environ.get("LRA_DEVICES")
            try:
                os.environ["LRA_DEVICES"] = "0"
                L.feature.melspectrogram(y=yl, **kw)
                t0 = time.perf_counter()
                M1 = L.feature.melspectrogram(y=yl, **kw)
                dt1 = time.perf_counter() - t0
                os.environ["LRA_DEVICES"] = "0,0"
                assert SP._frame_shard_plan(yl.shape[-1], M1.shape[-1], 1, yl.nbytes, N_FFT, HOP, True) is not None
                L.feature.melspectrogram(y=yl, **kw)
                t0 = time.perf_counter()
                M2 = L.feature.melspectrogram(y=yl, **kw)
                dt2 = time.perf_counter() - t0
            finally:
                if old is None:
                    os.environ.pop("LRA_DEVICES", None)
                else:
                    os.environ["LRA_DEVICES"] = old
            batched = side.get("end_to_end_numpy", {}).get("melspectrogram", {}).get("frames_per_s")
            fps = M1.shape[-1] / dt1
            return {"seconds_of_audio": secs, "frames": int(M1.shape[-1]), "ms_unsharded": dt1 * 1e3, "ms_two_frame_ranges_on_one_device": dt2 * 1e3, "frames_per_s": fps,
                    "sharded_equals_unsharded": bool(np.array_equal(M1, M2)), "frac_of_batched": (fps / batched) if batched else None,
                    "what": "feature.melspectrogram(y=<(1, 79.38 M) np.ndarray>) end to end through the host pipeline; `frac_of_batched` = its frames/s over end_to_end_numpy's 64 x 30 s batch"}

        measure("long_clip", long_clip)

        def db_and_mfcc():
            out = {}
            fn = lambda: L.power_to_db(M, ref=np.max)
            _, e = timed(fn, 10, 3, collective=False, ramp_ms=args.prewarm_ms / 4)
            out["power_to_db"] = {"ms_per_call": e / 10 * 1e3, "GBps": 2 * M.numel() * 4 / (e / 10) / 1e9, "what": "power_to_db(M, ref=np.max) on the mel batch: per-clip max reduction + one elementwise pass (reads M twice, writes once)"}
            fn = lambda: L.feature.mfcc(y=y, sr=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS, check_finite=False)
            _, e = timed(fn, 10, 3, collective=False, ramp_ms=args.prewarm_ms / 4)
            out["mfcc"] = {"ms_per_call": e / 10 * 1e3, "frames_per_s": frames_per_step / (e / 10), "what": "feature.mfcc(y=<device tensor>): fused mel kernel + per-clip max + DCT kernel with the dB scaling fused into its read (20 coefficients)"}
            return out

        measure("post", db_and_mfcc)

        def griffinlim_key():
            nb, it_a, it_b = 32, 4, 132  # (128 iterations apart: ~50 ms of device work against a few ms of run-to-run noise in the host-drawn phases)
            S = torch.abs(L.stft(y[:nb], n_fft=N_FFT, hop_length=HOP))
            L.griffinlim(S, n_iter=1, hop_length=HOP, rng=0)

            def run(iters):
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                out = L.griffinlim(S, n_iter=iters, hop_length=HOP, rng=0)
                torch.cuda.synchronize(device)
                return time.perf_counter() - t0, out

            # (the host-drawn phases are ~100 ms of either call and vary by a few ms: minima of three runs each, or the difference is noise)
            ta = min(run(it_a)[0] for _ in range(3))
            tb, yr = min((run(it_b) for _ in range(3)), key=lambda r: r[0])
            per_iter = (tb - ta) / (it_b - it_a)
            return {"clips": nb, "ms_per_iteration": per_iter * 1e3, "frames_per_s_per_iteration": nb * n_frames / per_iter, "ms_32_iterations": (ta + (32 - it_a) * per_iter) * 1e3,
                    "ms_setup": (ta - it_a * per_iter) * 1e3, "finite": bool(torch.isfinite(yr).all()),
                    "what": "librosa_amd.griffinlim(<device |stft|>): per iteration istft + stft + phase update, all device-resident (difference of a 132- and a 4-iteration call, minima of three runs each); "
                            "ms_setup = the initial phases (round 5: the reference's own PCG64 stream drawn on the device bit for bit, csrc/lra_rng.h; rounds 1-4 drew 42 M float64 on the host: 92 ms) + the final istft"}

        measure("griffinlim", griffinlim_key)

        def pcen_and_cqt():
            out = {}
            fn = lambda: L.pcen(M, sr=SR, hop_length=HOP)
            _, e = timed(fn, 10, 3, collective=False, ramp_ms=args.prewarm_ms / 4)
            out["pcen"] = {"ms_per_call": e / 10 * 1e3, "GBps": M.numel() * (4 + 8) / (e / 10) / 1e9,
                           "what": "pcen(M) on the mel batch: first-order smoother along time fused with the normalisation, float32 in, float64 out (the reference's result type); "
                                   "bound by the float64 transcendentals, not by HBM"}
            nb = min(64, batch)
            yc = y[:nb]
            for key, rt, cf in (("cqt_polyphase", "polyphase", True), ("cqt_default", "soxr_hq", True), ("cqt_polyphase_nocheck", "polyphase", False)):
                fn = lambda: L.cqt(yc, sr=SR, hop_length=HOP, res_type=rt, check_finite=cf)
                _, e = timed(fn, 5, 2, collective=False, ramp_ms=args.prewarm_ms / 4)
                alg_bytes = nb * (n * 4 + 84 * n_frames * 8)  # PCM read once + the stacked complex64 result written once
                out[key] = {"clips": nb, "ms_per_call": e / 5 * 1e3, "frames_per_s": nb * n_frames / (e / 5), "GBps_algorithmic": alg_bytes / (e / 5) / 1e9,
                            "what": f"librosa_amd.cqt(<device audio>, 84 bins, res_type={rt!r}): 7 octaves of rectangular-window STFT + sparse basis projection + FIR decimation, "
                                    "device-resident (the constant-Q transform BASELINE config 5 approximates); round 4: one fused launch per octave (frames + transform + projection + stacking, "
                                    "csrc/lra_mixed.h) and a register-blocked halving kernel: 13 launches"}
            # the same two rows through the oracle (port of the reference) on one host core, one clip each: a reported baseline
            try:
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import cqt_oracle as CQ
                import stft_oracle as O

                y1 = y[0].cpu().numpy()
                t0 = time.perf_counter()
                CQ.cqt(y1, sr=SR, hop_length=HOP, res_type="polyphase")
                t_cqt = time.perf_counter() - t0
                m1 = M[0].cpu().numpy()
                t0 = time.perf_counter()
                O.pcen(m1, sr=SR, hop_length=HOP)
                t_pcen = time.perf_counter() - t0
                out["cpu_baseline"] = {"kind": "port", "cores": 1, "sample": "one 30 s clip each", "cqt_ms_per_clip": t_cqt * 1e3, "pcen_ms_per_clip": t_pcen * 1e3,
                                       "gpu_cqt_ms_per_clip": out["cqt_polyphase"]["ms_per_call"] / nb, "gpu_pcen_ms_per_clip": out["pcen"]["ms_per_call"] / batch}
            except Exception as exc:  # pragma: no cover
                out["cpu_baseline"] = {"error": repr(exc)}
            return out

        measure("pcen_cqt", pcen_and_cqt)

        def hpss_key():
            nb = min(32, batch)
            Dh = L.stft(y[:nb], n_fft=N_FFT, hop_length=HOP)  # (the layout decompose.hpss consumes without a transpose)
            fn = lambda: L.decompose.hpss(Dh)
            _, e = timed(fn, 5, 2, collective=False, ramp_ms=args.prewarm_ms / 4)
            per = e / 5
            return {"clips": nb, "ms_per_call": per * 1e3, "GBps_algorithmic": 3 * Dh.numel() * 8 / per / 1e9,
                    "what": "decompose.hpss(<device stft>, kernel_size=31): |D|, both running medians (sorting networks in registers), soft masks, two masked spectra: "
                            "one complex64 spectrogram read, two written"}

        measure("hpss", hpss_key)

        def mixed_radix():
            """n_fft = 400, hop 160, 80 mels at 16 kHz (the front end of Whisper-style speech models) over 256 x 30 s: ONE fused launch (csrc/lra_mixed.h) against the
            framing + rocFFT + transpose + banded-product path it replaces (ctx option mixed = 0)."""
            sr2, nf, hp, nm = 16000, 400, 160, 80
            n2 = sr2 * CLIP_SECONDS
            y2 = y[:, :n2].contiguous()
            out = {}
            for key, opt in (("fused", 1), ("rocfft_path", 0)):
                ctx.set_option("mixed", opt)
                try:
                    fn = lambda: L.feature.melspectrogram(y=y2, sr=sr2, n_fft=nf, hop_length=hp, n_mels=nm, check_finite=False)
                    T2 = int(fn().shape[-1])
                    _, e = timed(fn, 5, 2, collective=False, ramp_ms=args.prewarm_ms / 4)
                    per = e / 5
                    fs = lambda: L.stft(y2, n_fft=nf, hop_length=hp, check_finite=False)
                    _, e2 = timed(fs, 5, 2, collective=False, ramp_ms=args.prewarm_ms / 4)
                    per2 = e2 / 5
                    Dm = fs()
                    fi = lambda: L.istft(Dm, hop_length=hp, n_fft=nf, length=n2)
                    _, e3 = timed(fi, 5, 2, collective=False, ramp_ms=args.prewarm_ms / 4)
                    per3 = e3 / 5
                    out[key] = {"mel_ms": per * 1e3, "mel_frames_per_s": batch * T2 / per, "mel_GBps_algorithmic": batch * T2 * (hp * 4 + nm * 4) / per / 1e9,
                                "stft_ms": per2 * 1e3, "stft_GBps_algorithmic": batch * T2 * (hp * 4 + (nf // 2 + 1) * 8) / per2 / 1e9,
                                "istft_ms": per3 * 1e3, "istft_GBps_algorithmic": batch * T2 * (hp * 4 + (nf // 2 + 1) * 8) / per3 / 1e9}
                    del Dm
                finally:
                    ctx.set_option("mixed", 1)
            out["workload"] = f"feature.melspectrogram / stft, n_fft=400 hop=160 n_mels=80 @ 16 kHz, {batch} clips x {CLIP_SECONDS} s (device tensors, public drop-in)"
            return out

        measure("mixed_radix_400", mixed_radix)

        def mixed_large(nf, hp):
            """The larger mixed-radix frames (25 ms at 48 kHz = 1200 samples; 3200) at 22.05 kHz over the same batch, hop n_fft / 4, 128 bands: stft and melspectrogram, fused launch only
            (reference sizes tests/test_core.py:256-292)."""
            def run():
                fs = lambda: L.stft(y, n_fft=nf, hop_length=hp, check_finite=False)
                T2 = int(fs().shape[-1])
                _, e2 = timed(fs, 5, 2, collective=False, ramp_ms=args.prewarm_ms / 4)
                fm = lambda: L.feature.melspectrogram(y=y, sr=SR, n_fft=nf, hop_length=hp, n_mels=N_MELS, check_finite=False)
                _, e = timed(fm, 5, 2, collective=False, ramp_ms=args.prewarm_ms / 4)
                Dm = fs()
                fi = lambda: L.istft(Dm, hop_length=hp, n_fft=nf, length=y.shape[-1])
                _, e3 = timed(fi, 5, 2, collective=False, ramp_ms=args.prewarm_ms / 4)
                del Dm
                return {"stft_ms": e2 / 5 * 1e3, "stft_GBps_algorithmic": batch * T2 * (hp * 4 + (nf // 2 + 1) * 8) / (e2 / 5) / 1e9, "mel_ms": e / 5 * 1e3, "mel_frames_per_s": batch * T2 / (e / 5),
                        "istft_ms": e3 / 5 * 1e3, "istft_GBps_algorithmic": batch * T2 * (hp * 4 + (nf // 2 + 1) * 8) / (e3 / 5) / 1e9,
                        "mel_GBps_algorithmic": batch * T2 * (hp * 4 + N_MELS * 4) / (e / 5) / 1e9, "frames": batch * T2,
                        "workload": f"stft / feature.melspectrogram, n_fft={nf} hop={hp} n_mels={N_MELS} @ {SR} Hz, {batch} clips x {CLIP_SECONDS} s (device tensors, public drop-in; csrc/lra_mixed.h)"}
            return run

        measure("mixed_radix_1200", mixed_large(1200, 300))
        measure("mixed_radix_3200", mixed_large(3200, 800))

        def speech_512():
            """n_fft = 512, hop 160, 80 mels at 16 kHz (32 ms frames every 10 ms): small frames share a wave (16 threads per frame), four bands per thread since round 5."""
            sr2, nf, hp, nm = 16000, 512, 160, 80
            y2 = y[:, : sr2 * CLIP_SECONDS].contiguous()
            fn = lambda: L.feature.melspectrogram(y=y2, sr=sr2, n_fft=nf, hop_length=hp, n_mels=nm, check_finite=False)
            T2 = int(fn().shape[-1])
            _, e = timed(fn, 5, 2, collective=False, ramp_ms=args.prewarm_ms / 4)
            per = e / 5
            return {"mel_ms": per * 1e3, "mel_frames_per_s": batch * T2 / per, "workload": f"feature.melspectrogram n_fft=512 hop=160 n_mels=80 @ 16 kHz, {batch} clips x {CLIP_SECONDS} s"}

        measure("speech_512", speech_512)
        # BASELINE config 5 (CQT-lite): three STFTs at n_fft = 512 / 2048 / 8192 over the same batch, shared hop 512
        if not args.no_cqt:
            def spec_buffer(shape):
                """A complex64 result buffer the way `stft(<device tensor>)` allocates it: lra_malloc_placed for results of 256 MB and more (ctx.placement_retry), else torch's allocator."""
                nb = int(np.prod(shape)) * 8
                if ctx.placement_retry > 0 and nb >= ctx.PLACED_MIN_BYTES:
                    try:
                        from librosa_amd import _arrays

                        return _arrays._placed_tensor(ctx, tuple(int(v) for v in shape), np.dtype(np.complex64), device)
                    except Exception:  # pragma: no cover
                        pass
                return torch.empty(shape, dtype=torch.complex64, device=device)

            def cqt_lite():
                parts = {}
                total_s = 0.0
                for nf in (512, 2048, 8192):
                    w = np.asarray(filters.get_window("hann", nf, fftbins=True), dtype=np.float32)
                    pl = plan if nf == N_FFT else ctx.stft_plan(nf, HOP, w, True, "constant", np.float32)
                    T_nf = ctx.stft_num_frames(pl, n)
                    Dn = D if nf == N_FFT else spec_buffer((batch, T_nf, nf // 2 + 1))
                    _, ev_n = timed(lambda: ctx.stft_exec(pl, yp, batch, n, n, Dn.data_ptr()), max(3, args.steps // 2), 2, collective=False, ramp_ms=args.prewarm_ms / 4)
                    per = ev_n / max(3, args.steps // 2)
                    bytes_n = batch * T_nf * ((nf // 2 + 1) * 8 + HOP * 4)
                    parts[str(nf)] = {"ms": per * 1e3, "GBps": bytes_n / per / 1e9}
                    total_s += per
                    del Dn
                out = {"workload": f"3 STFTs n_fft=512/2048/8192, hop=512, batch={batch} x {CLIP_SECONDS} s (BASELINE config 5)", "per_n_fft": parts, "ms_total": total_s * 1e3,
                       "frame_triples_per_s": frames_per_step / total_s, "frac_of_hbm": frames_per_step * (HOP * 4 + (257 + 1025 + 4097) * 8) / total_s / 1e9 / HBM_PEAK_GBS}
                # secondary variant: each transform at librosa's default hop = n_fft // 4 (core/spectrum.py:235-236): 128 / 512 / 2048
                parts4, total4, bytes4 = {}, 0.0, 0
                for nf in (512, 2048, 8192):
                    hp = nf // 4
                    w = np.asarray(filters.get_window("hann", nf, fftbins=True), dtype=np.float32)
                    pl = plan if nf == N_FFT else ctx.stft_plan(nf, hp, w, True, "constant", np.float32)
                    T_nf = ctx.stft_num_frames(pl, n)
                    Dn = spec_buffer((batch, T_nf, nf // 2 + 1))
                    _, ev_n = timed(lambda: ctx.stft_exec(pl, yp, batch, n, n, Dn.data_ptr()), max(3, args.steps // 2), 2, collective=False, ramp_ms=args.prewarm_ms / 4)
                    per = ev_n / max(3, args.steps // 2)
                    bytes_n = batch * T_nf * ((nf // 2 + 1) * 8 + hp * 4)
                    parts4[str(nf)] = {"hop": hp, "frames_per_clip": int(T_nf), "ms": per * 1e3, "GBps": bytes_n / per / 1e9}
                    total4 += per
                    bytes4 += bytes_n
                    del Dn
                out["default_hop_variant"] = {"workload": "the same three transforms at hop = n_fft // 4 (128 / 512 / 2048)", "per_n_fft": parts4, "ms_total": total4 * 1e3,
                                              "GBps": bytes4 / total4 / 1e9, "frac_of_hbm": bytes4 / total4 / 1e9 / HBM_PEAK_GBS}
                return out

            measure("cqt_lite", cqt_lite)
        del D

    if rank == 0:
        total_frames = frames_per_step * args.steps * world
        value = total_frames / wall
        launch_s = ev / args.steps
        achieved = frames_per_step * BYTES_PER_FRAME_MEL / launch_s / 1e9
        split = (f"BASELINE configs[1]: batch={batch} clips x {CLIP_SECONDS} s on 1 GPU" if world == 1 else
                 f"BASELINE configs[2]'s split: {batch} clips x {CLIP_SECONDS} s per GPU, {total_clips} clips on {world} GPUs (clip i on GPU i // {batch})")
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
            "config": {"workload": f"feature.melspectrogram, {split}, @ 22.05 kHz, n_fft={N_FFT} hop={HOP} n_mels={N_MELS}, inputs resident in HBM, outputs left sharded "
                                   f"(`gathered`: all-gathered)", "frames_per_step_per_gpu": frames_per_step, "clips_per_gpu": batch, "prewarm_ms": args.prewarm_ms,
                       "parallelism": f"clips sharded over {world} GPU(s), one process per GPU, no collective on the data path", "device": ctx.device_name(),
                       "self_launched": bool(os.environ.get("LRA_BENCH_SELF_LAUNCHED"))},
            "roofline": {"bound": "hbm", "kernel": "stft_pc_kernel<n_fft=2048> (fused melspectrogram: producer / consumer waves, csrc/lra_kernels_pc.h)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "bytes_per_frame": BYTES_PER_FRAME_MEL,
                         "launch_ms": launch_s * 1e3, "frames_per_s_single_gpu": frames_per_step / launch_s,
                         "limited_by": "valu/lds issue, not HBM (achieved / peak / frac here ARE the HBM figures BASELINE's metric asks for; `valu_frac` = the same launch against the f32 vector peak, "
                                       "which is the side of the ridge this kernel sits on)",
                         "note": "this kernel sits on the LDS / VALU side of the ridge (~65 kFLOP and ~500 LDS cycles per 2 560 B); see roofline_stft for the HBM-bound kernel"},
            "repeats": {"ms_per_step_min": min(rep), "ms_per_step_median": statistics.median(rep), "ms_per_step_all": rep, "what": "5 more repeats of the timed region (HIP events, no collectives)"},
        }
        # the fused mel kernel against the f32 vector peak (~65 kFLOP per frame by the 5 N log2 N convention)
        line["roofline_valu"] = {"bound": "valu", "kernel": line["roofline"]["kernel"], "achieved": frames_per_step * 65e3 / launch_s / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                                 "frac": frames_per_step * 65e3 / launch_s / 1e12 / 157.3, "flop_per_frame": 65e3}
        for k, v in side.items():
            if k == "post" and isinstance(v, dict) and "error" not in v:
                line.update(v)
            else:
                line[k] = v
        line["kernel_variants"] = {"melspectrogram": ctx.tuned_variant(plan, 2),
                                   "note": "n_fft=2048 f32 fused mel: 0 = second-generation core, one wave64 per frame; 4 = first generation, two waves per frame; chosen by timing both on the first call "
                                           "(ctx option autotune); stft / |X|^p always run the second-generation kernel"}
        if "roofline_stft" in line and "error" not in line["roofline_stft"]:
            line["roofline_stft"]["achievable_note"] = ("the same stream without arithmetic is measured in this line: stream_ceiling.forward (lra_probe_stream); rounds 2-3 saw 5.0-5.55 TB/s "
                                                        "over the pool's boxes (profiles/r02_store_stream.md)")
        # HBM bytes per launch measured with rocprofv3 PMC passes (scripts/profile_round.sh), when committed
        try:
            prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))
            if prof:
                tj = json.load(open(os.path.join(ROOT, "profiles", prof[-1])))
                for key, needle in (("roofline", "mel"), ("roofline_stft", "complex64"), ("roofline_istft", "istft")):
                    if key not in line or "error" in line[key]:
                        continue
                    # several variants of a kernel may appear (autotune candidates): the one launched most is the one timed here
                    cands = [(v.get("launches", 0), kname, v) for kname, v in tj.get("kernels", {}).items() if needle in kname and "n_fft=2048" in kname and v.get("hbm_bytes")]
                    if cands:
                        _, kname, v = max(cands, key=lambda c: c[0])
                        line[key]["traffic"] = v["hbm_bytes"]
                        line[key]["traffic_source"] = f"profiles/{prof[-1]} ({kname})"
                        line[key]["traffic_box"] = "profile"  # measured under rocprofv3 on the builder's box, not on the box this line was timed on
        except Exception:
            pass
        if world == 1 and not args.no_cpu_baseline:
            try:
                clip = (y[0].cpu().numpy(), M[0].cpu().numpy())
            except Exception:
                clip = None
            line["cpu_baseline"], parity = cpu_baseline(parity_clip=clip)
            if parity is not None:
                line["parity"] = parity
            try:
                line["cpu_baseline_all_cores"] = cpu_baseline_all_cores()
            except Exception as exc:  # the single-core object above is the contract; this one is informative
                line["cpu_baseline_all_cores"] = {"error": repr(exc)}
        # ONE short line on stdout (the driver's record keeps it whole); the full record goes to stderr and, where it can, to a file
        full = json.dumps(line)
        print(full, file=sys.stderr, flush=True)
        detail = args.detail or (os.path.join(ROOT, "gpurun_out", "bench_detail.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
        if detail:
            try:
                with open(detail, "w") as fh:
                    fh.write(full + "\n")
            except OSError:
                pass
        print(json.dumps(compact_line(line)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
