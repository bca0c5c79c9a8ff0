#!/bin/bash
# GPU box: per-kernel durations of one librosa_amd.cqt configuration (rocprofv3 kernel trace).  scripts/cqt_trace.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/cqt_trace; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python $R/scripts/cqt_hostprof.py > $O/log.txt 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$O/r_kernel_trace.csv")))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"][:70], r["Grid_Size_X"], r["LDS_Block_Size"])].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:22]:
    print(f"{k[0]:70s} grid {k[1]:>9s} lds {k[2]:>6s} n {len(v):5d} avg {sum(v)/len(v):8.1f} us")
PY
