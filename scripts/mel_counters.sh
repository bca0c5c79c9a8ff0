#!/bin/bash
# Development probe: wait / issue counters of the fused mel kernel (two SQ passes) -> gpurun_out/melq/
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/melq; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d $OUT/a -o r -- python $R/scripts/size_probe.py 2048 512 2 mel > $OUT/a.log 2>&1
timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES -d $OUT/b -o r -- python $R/scripts/size_probe.py 2048 512 2 mel > $OUT/b.log 2>&1
python - <<PY
import csv, collections, glob
for sub in ("a", "b"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            if "stft2_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print({k: "%.4g" % (sum(v) / len(v)) for k, v in agg.items()})
PY
