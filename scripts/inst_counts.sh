#!/bin/bash
# Development probe (round 6): dynamic instruction counts per frame of the forward kernels' forms (one SQ pass each).  -> gpurun_out/r06/instq
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r06/instq; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for cfg in "stft:v3=0" "stft:v3=1" "power:v3=0" "power:v3=1" "mel:mel_pc=0" "mel:mel_pc=1" "mel:mel_pc=2"; do
  what=${cfg%%:*}; opt=${cfg##*:}; i=$((i+1))
  PROBE_OPTS="autotune=0,$opt" timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/p$i -o r -- python $R/scripts/size_probe.py 2048 512 2 $what > $OUT/p$i.log 2>&1 || { echo "pass $cfg failed"; tail -3 $OUT/p$i.log; continue; }
  python - <<PY
import csv, collections, glob
agg = collections.defaultdict(list); names = set()
for f in glob.glob("$OUT/p$i/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "stft2_kernel" in r["Kernel_Name"] or "stft_pc_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"])); names.add(r["Kernel_Name"][:110])
fr = 330752.0
print("$cfg", sorted(names))
print("   per frame: VALU %.1f  SALU %.1f  LDS %.1f  | waves %d  wait_any/wave_cycles %.3f  LDS conflict share %.3f" % (
    sum(agg["SQ_INSTS_VALU"]) / len(agg["SQ_INSTS_VALU"]) / fr, sum(agg["SQ_INSTS_SALU"]) / len(agg["SQ_INSTS_SALU"]) / fr, sum(agg["SQ_INSTS_LDS"]) / len(agg["SQ_INSTS_LDS"]) / fr,
    sum(agg["SQ_WAVES"]) / len(agg["SQ_WAVES"]), sum(agg["SQ_WAIT_INST_ANY"]) / sum(agg["SQ_WAVE_CYCLES"]), sum(agg["SQ_LDS_BANK_CONFLICT"]) / sum(agg["SQ_LDS_IDX_ACTIVE"])))
PY
done
