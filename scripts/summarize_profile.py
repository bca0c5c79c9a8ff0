"""Turn gpurun_out/prof_<tag>/ (made by scripts/profile_round.sh on the GPU box) into the committed
profiles/<tag>_* files: kernel stats CSV, a per-kernel counter table (markdown) and <tag>_traffic.json
(HBM bytes per launch from FETCH_SIZE / WRITE_SIZE, read back by bench.py for roofline.traffic).

FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B? -> rocprofv3 reports them in kilobytes
(x1024).  On gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM);
the run contains torch's own elementwise kernels over the 677 MB input, which are used here as the
in-situ calibration of both counters against a known byte count.
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    import re
    cfg = re.search(r"FftCfg<(\d+), (\d), (\w+), (\d+), (\d), (\w+)(?:, (\w+))?(?:, \d+)?>", name)
    plan1 = bool(cfg and re.search(r"FftCfg<[^>]*, 1>", name))
    tag = f" [n_fft={2 ** (int(cfg.group(1)) + 1)}, 2^{cfg.group(2)} pts/thread, NT={cfg.group(4)}{', ascending radices' if cfg.group(7) == 'true' else ''}{', radices 16-16-4' if plan1 else ''}]" if cfg else ""
    modes = {"0": "complex64 out", "1": "power out", "2": "mel, generic", "3": "mel, two-slope", "4": "mel, run-ordered two-slope"}
    if "istft_kernel" in name:
        return "istft_kernel" + tag
    if "stft_pc_kernel" in name:  # producer / consumer fused mel kernel: <Cfg, n_fft / hop, PM>
        return "stft_pc_kernel<mel, producer / consumer waves>" + tag
    if "stft2_kernel" in name:  # second generation: <Cfg, n_fft / hop, MODE, PM>
        m = re.search(r">, (\d+), (\d), (\d)>\(", name)
        return f"stft2_kernel<{modes.get(m.group(2), '?') if m else '?'}>" + tag
    if "stft_kernel" in name:
        m = re.search(r">, (\d), (\d), (true|false|\d)>\(", name) or re.search(r">, (\d)(?:, \d)?>\(", name)
        return f"stft_kernel<{modes.get(m.group(1) if m else '?', '?')}>" + tag
    return name.split("(")[0][-60:]


def counters(sub):
    path = os.path.join(src, sub, "r_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return agg
    rows = list(csv.DictReader(open(path)))
    # a kernel is also launched on other batches (autotune probes, the 64-clip NumPy end-to-end key, the 32-clip Griffin-Lim loop -- which
    # outnumbers everything else): keep the launches with the grid that most of the kernel's TIME goes to, i.e. the 256-clip batch of the
    # timed step every per-launch figure refers to
    grids = collections.defaultdict(collections.Counter)
    for r in rows:
        grids[r["Kernel_Name"]][int(r["Grid_Size"])] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    timed = {k: v.most_common(1)[0][0] for k, v in grids.items()}  # the grid most of the kernel's time goes to
    for r in rows:
        if int(r["Grid_Size"]) != timed[r["Kernel_Name"]]:
            continue
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[r["Kernel_Name"]]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        agg[r["Kernel_Name"]]["_vgpr"].append(float(r["VGPR_Count"]) + float(r["Accum_VGPR_Count"]))
        agg[r["Kernel_Name"]]["_lds"].append(float(r["LDS_Block_Size"]))
    return agg


def mean(x):
    return sum(x) / max(len(x), 1)


stats_csv = os.path.join(src, "stats", "r_kernel_stats.csv")
if os.path.exists(stats_csv):
    shutil.copy(stats_csv, os.path.join(dst, f"{tag}_kernel_stats.csv"))
bench_line = {}
bl = os.path.join(src, "bench_line.json")
if os.path.exists(bl) and os.path.getsize(bl):
    bench_line = json.loads(open(bl).read())
    json.dump(bench_line, open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w"), indent=1)

fetch, write = counters("fetch"), counters("write")
# calibration on torch's clamp_ (reads and writes the whole 256 x 661500 f32 batch once)
calib = {}
known = 256 * 661500 * 4
for k, v in fetch.items():
    if "clamp" in k.lower() and "FETCH_SIZE" in v:
        calib["fetch_kb_per_true_byte"] = mean(v["FETCH_SIZE"]) * 1024 / known
for k, v in write.items():
    if "clamp" in k.lower() and "WRITE_SIZE" in v:
        calib["write_kb_per_true_byte"] = mean(v["WRITE_SIZE"]) * 1024 / known
traffic = {"tag": tag, "calibration": calib, "kernels": {}}
for k in set(list(fetch) + list(write)):
    if "stft_kernel" not in k and "stft2_kernel" not in k and "stft_pc_kernel" not in k:
        continue
    f_raw = mean(fetch[k]["FETCH_SIZE"]) * 1024 if "FETCH_SIZE" in fetch.get(k, {}) else None
    w_raw = mean(write[k]["WRITE_SIZE"]) * 1024 if "WRITE_SIZE" in write.get(k, {}) else None
    fc = calib.get("fetch_kb_per_true_byte") or 1.0
    wc = calib.get("write_kb_per_true_byte") or 1.0
    traffic["kernels"][short(k)] = {
        "fetch_bytes_raw": f_raw, "write_bytes_raw": w_raw,
        "fetch_bytes": f_raw / fc if f_raw is not None else None, "write_bytes": w_raw / wc if w_raw is not None else None,
        "hbm_bytes": (f_raw / fc if f_raw is not None else 0) + (w_raw / wc if w_raw is not None else 0),
        "launches": max(len(fetch.get(k, {}).get("FETCH_SIZE", [])), len(write.get(k, {}).get("WRITE_SIZE", []))),
    }
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)

lines = [f"# rocprofv3 summary, round tag `{tag}`", "", "Command: `scripts/profile_round.sh` (bench.py workload: 256 clips x 30 s, n_fft=2048 hop=512 n_mels=128).", ""]
if bench_line:
    lines += [f"bench line under `rocprofv3 --kernel-trace --stats`: value {bench_line.get('value', 0) / 1e6:.1f} Mframes/s, "
              f"mel kernel {bench_line.get('roofline', {}).get('launch_ms')} ms/launch, stft kernel {bench_line.get('roofline_stft', {}).get('launch_ms') or bench_line.get('roofline', {}).get('stft_ms')} ms/launch", ""]
if os.path.exists(stats_csv):
    lines += ["## kernel stats (top 6 by total time)", "", "| kernel | calls | avg (us) | min (us) | max (us) | % |", "|---|---|---|---|---|---|"]
    rows = list(csv.DictReader(open(stats_csv)))[:6]
    for r in rows:
        lines.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {r['Percentage']} |")
    lines.append("")
# the same trace restricted to the timed 256-clip batch (the grid most of the kernel's time goes to): these are the averages the bench line's launch_ms must agree with
trace_csv = os.path.join(src, "stats", "r_kernel_trace.csv")
if os.path.exists(trace_csv):
    rows = list(csv.DictReader(open(trace_csv)))
    grids = collections.defaultdict(collections.Counter)
    for r in rows:
        grids[r["Kernel_Name"]][int(r["Grid_Size_X"])] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    durs = collections.defaultdict(list)
    for r in rows:
        k = r["Kernel_Name"]
        if ("stft" in k) and int(r["Grid_Size_X"]) == grids[k].most_common(1)[0][0]:
            durs[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    lines += ["## launches of the timed batch only (the grid most of the kernel's time goes to), from the same trace", "", "| kernel | launches | avg (us) | median (us) | min (us) |", "|---|---|---|---|---|"]
    for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1]))[:6]:
        v2 = sorted(v)
        lines.append(f"| {short(k)} | {len(v)} | {mean(v) / 1e3:.1f} | {v2[len(v2) // 2] / 1e3:.1f} | {v2[0] / 1e3:.1f} |")
    lines.append("")
    # the kernels of the SURVEY 8(f) rows and the stream probe, same trace: every launch of the grid most of their time goes to
    other = collections.defaultdict(list)
    FRAMES = 256 * 1292
    known_bytes = {"pcen_kernel": ("256 x 128 x 1292 values, 4 B read + 8 B written", 256 * 128 * 1292 * 12),
                   "hpss_kernel": ("32 x 1292 x 1025 bins: |D| (4 B) + D (8 B) read, two spectra (16 B) written", 32 * 1292 * 1025 * 28),
                   "hpss_tile_kernel": ("32 x 1292 x 1025 bins: |D| (4 B) + D (8 B) read, two spectra (16 B) written", 32 * 1292 * 1025 * 28),
                   "magnitude_kernel": ("32 x 1292 x 1025 bins: D (8 B) read, |D| (4 B) written", 32 * 1292 * 1025 * 12),
                   "mixed_cqt_kernel": ("one octave of 64 x 30 s: the octave's signal read once (<= 169 MB), 12 bins x frames written", 0),
                   "stream_probe_kernel<0>": ("forward stream, 10 248 B per row", FRAMES * 10248), "stream_probe_kernel<1>": ("inverse stream, 10 248 B per row", FRAMES * 10248),
                   "to_db_kernel": ("256 x 128 x 1292 values read and written", 256 * 128 * 1292 * 8)}
    for r in rows:
        k = r["Kernel_Name"]
        if any(n in k for n in ("pcen_kernel", "hpss_kernel", "hpss_tile_kernel", "mixed_cqt_kernel", "fir_halve4", "resample_", "mixed_stft_kernel", "mixed_istft_kernel", "magnitude_kernel", "cqt_project_kernel", "fir_decimate", "stream_probe_kernel", "to_db_kernel", "dct_rows_kernel",
                                "item_absmax_kernel", "griffinlim_update_kernel", "phase_vocoder_kernel", "wss_to_norm_kernel", "fillBuffer")):
            if int(r["Grid_Size_X"]) == grids[k].most_common(1)[0][0]:
                other[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if other:
        lines += ["## other kernels of the path (SURVEY 8f rows, the stream probe; launches of the grid most of each kernel's time goes to)", "",
                  "| kernel | launches | avg (us) | algorithmic bytes per launch | GB/s | % of 8 TB/s |", "|---|---|---|---|---|---|"]
        for k, v in sorted(other.items(), key=lambda kv: -sum(kv[1])):
            name = k.split("(")[0][-70:]
            kb = next((val for key, val in known_bytes.items() if key in k.replace("lra::", "") and val[1]), None)
            avg = mean(v)
            lines.append(f"| {name} | {len(v)} | {avg / 1e3:.1f} | {kb[0] if kb else '-'} | {(kb[1] / avg) if kb else 0:.0f} | {(kb[1] / avg / 80) if kb else 0:.1f} |" if kb else f"| {name} | {len(v)} | {avg / 1e3:.1f} | - | - | - |")
        lines.append("")
lines += ["## HBM traffic per launch (FETCH_SIZE / WRITE_SIZE, separate passes)", "", f"calibration on torch clamp_ over the 677.4 MB batch: {calib}", "",
          "| kernel | FETCH raw (MB) | WRITE raw (MB) | FETCH calibrated (MB) | WRITE calibrated (MB) |", "|---|---|---|---|---|"]
for k, v in sorted(traffic["kernels"].items()):
    fmt = lambda x: "-" if x is None else f"{x / 1e6:.1f}"
    lines.append(f"| {k} | {fmt(v['fetch_bytes_raw'])} | {fmt(v['write_bytes_raw'])} | {fmt(v['fetch_bytes'])} | {fmt(v['write_bytes'])} |")
lines.append("")
for sub in ("sq1", "sq2"):
    agg = counters(sub)
    keys = sorted({c for k, v in agg.items() if ("stft_kernel" in k or "stft2_kernel" in k or "stft_pc_kernel" in k) for c in v})
    if not keys:
        continue
    lines += [f"## SQ counters, pass `{sub}` (mean per launch)", "", "| kernel | " + " | ".join(keys) + " |", "|---|" + "---|" * len(keys)]
    for k, v in agg.items():
        if "stft_kernel" in k or "stft2_kernel" in k or "stft_pc_kernel" in k:
            lines.append(f"| {short(k)} | " + " | ".join(f"{mean(v[c]):.4g}" if c in v else "-" for c in keys) + " |")
    lines.append("")
# derived per-frame figures of the three BASELINE kernels (330 752 frames per launch): vector instructions, LDS instructions, bank-conflict share
sq1, sq2 = counters("sq1"), counters("sq2")
der = []
for k in sq1:
    if not ("stft2_kernel" in k or "istft_kernel" in k or "stft_pc_kernel" in k) or "Li10ELi4Ef" not in k and "10, 4, float" not in k:
        continue
    a, b = sq1[k], sq2.get(k, {})
    g = lambda d, c: mean(d[c]) if c in d else float("nan")
    frames = 256 * 1292
    der.append(f"| {short(k)} | {g(a, 'SQ_INSTS_VALU') / frames:.0f} | {g(b, 'SQ_INSTS_LDS') / frames:.1f} | {g(b, 'SQ_LDS_BANK_CONFLICT') / max(g(b, 'SQ_LDS_IDX_ACTIVE'), 1):.3f} | "
               f"{g(a, 'SQ_WAIT_INST_ANY') / max(g(a, 'SQ_WAVE_CYCLES'), 1):.2f} | {g(a, 'SQ_ACTIVE_INST_VALU') / max(g(a, 'SQ_WAVE_CYCLES'), 1):.2f} | {mean(a.get('_vgpr', [0])):.0f} |")
if der:
    lines += ["## derived, per frame (330 752 frames per launch of the timed batch)", "",
              "| kernel | VALU instructions / frame | LDS instructions / frame | LDS bank-conflict cycles / LDS active cycles | SQ_WAIT_INST_ANY / wave cycles | VALU active / wave cycles | VGPRs (granules x 2?) |",
              "|---|---|---|---|---|---|---|"] + der + [""]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
