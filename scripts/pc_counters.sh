#!/bin/bash
# Development probe (round 6): issue / wait counters of the fused mel kernel in its two forms -- stft2_kernel<OUT_MELR> (ctx option mel_pc = 0) and the
# producer / consumer kernel stft_pc_kernel (mel_pc = 1) -- two SQ passes each, plus board power / clocks while each loops.  -> gpurun_out/r06/pcq_*
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r06/pcq; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for pc in 0 1; do
  export PROBE_OPTS="autotune=0,variant=0,mel_pc=$pc"
  timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d $OUT/a$pc -o r -- python $R/scripts/size_probe.py 2048 512 2 mel > $OUT/a$pc.log 2>&1 || { echo "pass a$pc failed"; tail -3 $OUT/a$pc.log; exit 1; }
  timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/b$pc -o r -- python $R/scripts/size_probe.py 2048 512 2 mel > $OUT/b$pc.log 2>&1 || { echo "pass b$pc failed"; tail -3 $OUT/b$pc.log; exit 1; }
  timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES_EQ_64 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM -d $OUT/c$pc -o r -- python $R/scripts/size_probe.py 2048 512 2 mel > $OUT/c$pc.log 2>&1 || echo "pass c$pc failed (optional)"
done
python - <<PY
import csv, collections, glob
for pc in (0, 1):
    agg = collections.defaultdict(list); dur = []
    for sub in ("a", "b", "c"):
        for f in glob.glob("$OUT/%s%d/*counter_collection.csv" % (sub, pc)):
            for r in csv.DictReader(open(f)):
                if "stft2_kernel" in r["Kernel_Name"] or "stft_pc_kernel" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob("$OUT/%s%d/*kernel_trace.csv" % (sub, pc)):
            for r in csv.DictReader(open(f)):
                if "stft2_kernel" in r["Kernel_Name"] or "stft_pc_kernel" in r["Kernel_Name"]:
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("mel_pc", pc, "launches", len(dur), "avg us under counters %.1f" % (sum(dur) / max(1, len(dur))))
    print({k: "%.5g" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
for pc in 0 1; do
  echo "== power, mel_pc $pc"
  PROBE_OPTS="autotune=0,variant=0,mel_pc=$pc" bash $R/scripts/power_probe.sh 2048 512 mel 8
done
