#!/bin/bash
# GPU box: SQ / cache counters of the hpss kernels (separate --pmc passes, kernel-trace only).  scripts/hpss_counters.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/hpss_pmc; mkdir -p $O
P="python $R/scripts/hpss_probe.py 32 once"
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_WAIT_ANY -d $O/a -o r -- $P > $O/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_LDS -d $O/b -o r -- $P > $O/b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum -d $O/c -o r -- $P > $O/c.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TOTAL_ACCESSES_sum -d $O/d -o r -- $P > $O/d.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE -d $O/e -o r -- $P > $O/e.log 2>&1
python - <<PY
import csv, collections
for sub in "abcde":
    try: rows = list(csv.DictReader(open("$O/%s/r_counter_collection.csv" % sub)))
    except Exception as e: print(sub, e); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "hpss" in r["Kernel_Name"]: agg[r["Kernel_Name"][8:36]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, {c: "%.4g" % (sum(x) / len(x)) for c, x in v.items()})
PY
tail -3 $O/c.log $O/d.log $O/e.log | grep -iE "error|invalid|not" | head
