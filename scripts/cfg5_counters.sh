#!/bin/bash
# Development probe (round 6): counters of configs[4]'s three legs (n_fft 512 / 2048 / 8192 at hop 512), one SQ pass each.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r06/c5q; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for nf in 512 8192; do
  timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/a$nf -o r -- python $R/scripts/size_probe.py $nf 512 2 stft > $OUT/a$nf.log 2>&1
  timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -d $OUT/b$nf -o r -- python $R/scripts/size_probe.py $nf 512 2 stft > $OUT/b$nf.log 2>&1
  python - <<PY
import csv, collections, glob
agg = collections.defaultdict(list); meta = {}
for sub in ("a", "b"):
    for f in glob.glob("$OUT/%s$nf/*counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            if "stft" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"])); meta = {"vgpr": r["VGPR_Count"], "lds": r["LDS_Block_Size"], "wg": r["Workgroup_Size"], "grid": r["Grid_Size"], "name": r["Kernel_Name"][:90]}
m = {k: sum(v) / len(v) for k, v in agg.items()}
fr = 330752.0
print("n_fft $nf", meta)
print("   per frame: VALU %.0f SALU %.0f LDS %.1f VMEM wr %.1f rd %.1f | waves %d  wait_inst_any/wave_cycles %.3f  wait_any/wave_cycles %.3f  wait_inst_lds/wave_cycles %.3f  LDS conflict share %.3f  LDS active cycles/frame %.0f" % (
    m["SQ_INSTS_VALU"] / fr, m["SQ_INSTS_SALU"] / fr, m["SQ_INSTS_LDS"] / fr, m["SQ_INSTS_VMEM_WR"] / fr, m["SQ_INSTS_VMEM_RD"] / fr, m["SQ_WAVES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
    m["SQ_WAIT_INST_LDS"] / m["SQ_WAVE_CYCLES"], m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], m["SQ_LDS_IDX_ACTIVE"] / fr))
PY
done
rm -rf $OUT
