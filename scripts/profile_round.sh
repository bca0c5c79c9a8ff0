#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes for the bench.py workload.
# Usage: scripts/profile_round.sh <tag>     -> gpurun_out/prof_<tag>/{stats,fetch,write,sq1,sq2}/...
# Counters are collected in their own passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-cqt --no-power --no-scaling-base --placements 1"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o r -- $BENCH --steps 20 --warmup 5 > "$OUT/stats.log" 2>&1
# (round 5: on one box every pass died with a GPU memory fault right after HSA initialisation and then sat out its timeout -- 20 GPU-minutes; stop at the first such pass)
if grep -q "Memory access fault" "$OUT/stats.log" || [ ! -d "$OUT/stats" ]; then echo "profile_round: the stats pass failed on this box, giving up"; tail -5 "$OUT/stats.log"; exit 1; fi
timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch" -o r -- $BENCH --steps 3 --warmup 1 > "$OUT/fetch.log" 2>&1
timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/write" -o r -- $BENCH --steps 3 --warmup 1 > "$OUT/write.log" 2>&1
timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d "$OUT/sq1" -o r -- $BENCH --steps 3 --warmup 1 > "$OUT/sq1.log" 2>&1
timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/sq2" -o r -- $BENCH --steps 3 --warmup 1 > "$OUT/sq2.log" 2>&1
grep -h '"metric"' "$OUT/stats.log" | tail -1 > "$OUT/bench_line.json"
# keep only the small CSVs (gpurun merges <= 64 MiB back): the PMC passes' own kernel traces repeat what their counter files carry, and of the
# counter rows only the path's kernels (+ torch's clamp_, the FETCH_SIZE / WRITE_SIZE calibration) are read by scripts/summarize_profile.py
find "$OUT" -name '*_agent_info.csv' -delete
for d in fetch write sq1 sq2; do rm -f "$OUT/$d"/*kernel_trace.csv; done
python - "$OUT" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    if not rows:
        continue
    keep = [r for r in rows if any(t in r["Kernel_Name"] for t in ("stft", "clamp", "stream_probe", "placed_probe"))]
    with open(f, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
PY
du -sh "$OUT"
ls -la "$OUT"/*/ | head -40
