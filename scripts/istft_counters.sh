#!/bin/bash
# Development probe (GPU box): wait / issue / memory-latency counters of the inverse kernel (two SQ passes per library) -> gpurun_out/istq/
#   scripts/istft_counters.sh [probe/lib_a.so ...]     (no argument: the product library)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/istq; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=("")
for lib in "${libs[@]}"; do
  tag=$(basename "${lib:-product}" .so)
  [ -n "$lib" ] && export LIBROSA_AMD_LIBRARY=$R/$lib
  timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/${tag}_a -o r -- python $R/scripts/size_probe.py 2048 512 2 istft > $OUT/${tag}_a.log 2>&1
  timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SALU -d $OUT/${tag}_b -o r -- python $R/scripts/size_probe.py 2048 512 2 istft > $OUT/${tag}_b.log 2>&1
  python - <<PY
import csv, collections, glob
for sub in ("a", "b"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/${tag}_%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "istft_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("$tag", {k: "%.4g" % (sum(v) / len(v)) for k, v in agg.items()})
PY
done
