#!/bin/bash
# GPU box: SQ counters of the mixed-radix kernel (separate --pmc passes, kernel-trace only).  scripts/mixed_counters.sh [what]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; W=${1:-mel}
export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/mixed_$W; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_ANY -d $O/a -o r -- python $R/scripts/mixed_probe.py 400 160 80 16000 $W 3 > $O/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS -d $O/b -o r -- python $R/scripts/mixed_probe.py 400 160 80 16000 $W 3 > $O/b.log 2>&1
python - <<PY
import csv, collections
for sub in "ab":
    try: rows = list(csv.DictReader(open("$O/%s/r_counter_collection.csv" % sub)))
    except Exception as e: print(sub, e); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "mixed" in r["Kernel_Name"]: agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, {c: "%.4g" % (sum(x) / len(x)) for c, x in v.items()}, "vgpr", rows[0].get("VGPR_Count"), "lds", rows[0].get("LDS_Block_Size"))
PY
