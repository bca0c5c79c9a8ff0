#!/bin/bash
# Development probe (GPU box): kernel trace of decompose.hpss / effects.hpss.  bash scripts/hpss_trace.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/hpss_trace -o hpss -- python $R/scripts/hpss_probe.py 32 > $R/gpurun_out/hpss_trace.log 2>&1
python - <<'PY'
import csv, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for f in glob.glob(R + "/gpurun_out/hpss_trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
