#!/bin/bash
# Development probe (GPU box), round 6: every kernel of one bench.py run ranked by the time its waves spend waiting (one SQ counter pass, kernel trace only):
# which kernels are waiting for memory rather than computing?   scripts/wait_rank.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/wait_rank; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $O/a -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-power --no-scaling-base --placements 1 --prewarm-ms 50 > $O/a.log 2>&1
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$O/a/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:100]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
rows = []
for k, m in agg.items():
    if m["SQ_WAVE_CYCLES"] <= 0 or n[k] == 0: continue
    busy = m["SQ_BUSY_CYCLES"] / 32 / n[k]   # ~ cycles per launch
    rows.append((busy * n[k], k, n[k], busy, m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (m["SQ_BUSY_CYCLES"] / 32)))
print("%-100s %6s %12s %9s %9s %9s" % ("kernel", "n", "cycles/launch", "wait_any", "wait_inst", "VALU busy"))
for tot, k, cnt, busy, wa, wi, vb in sorted(rows, reverse=True)[:60]:
    print("%-100s %6d %12.0f %9.3f %9.3f %9.3f" % (k, cnt, busy, wa, wi, vb))
PY
rm -rf $O/a
