#!/bin/bash
# GPU box: SQ counters of the mixed-radix kernel (separate --pmc passes, kernel-trace only).  scripts/mixed_counters.sh [what] [n_fft hop n_mels sr]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; W=${1:-mel}; NF=${2:-400}; HOP=${3:-160}; NM=${4:-80}; SR=${5:-16000}
export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/mixed_${W}_$NF; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_ANY -d $O/a -o r -- python $R/scripts/mixed_probe.py $NF $HOP $NM $SR $W 3 > $O/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $O/b -o r -- python $R/scripts/mixed_probe.py $NF $HOP $NM $SR $W 3 > $O/b.log 2>&1
tail -1 $O/a.log
python - <<PY
import csv, collections, glob
m = {}; meta = {}
for sub in "ab":
    for f in glob.glob("$O/%s/*counter_collection.csv" % sub):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mixed" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"])); meta = {"vgpr": r["VGPR_Count"], "lds": r["LDS_Block_Size"], "wg": r["Workgroup_Size"], "grid": r["Grid_Size"], "name": r["Kernel_Name"][:70]}
        m.update({k: sum(v) / len(v) for k, v in agg.items()})
print(meta)
frames = 256.0 * (1 + ($SR * 30) // $HOP)
g = lambda k: m.get(k, float("nan"))
print("   per frame: VALU %.0f SALU %.0f LDS %.1f VMEM wr %.1f rd %.1f | waves %d  VALU busy (x4 / 1024 SIMDs / busy cycles) %.3f  wait_inst_any/wave_cycles %.3f  wait_any/wave_cycles %.3f  wait_inst_lds/wave_cycles %.3f  LDS conflict share %.3f  LDS active cycles/frame %.0f  busy cycles %.3g" % (
    g("SQ_INSTS_VALU") / frames, g("SQ_INSTS_SALU") / frames, g("SQ_INSTS_LDS") / frames, g("SQ_INSTS_VMEM_WR") / frames, g("SQ_INSTS_VMEM_RD") / frames, g("SQ_WAVES"), g("SQ_ACTIVE_INST_VALU") * 4 / 1024 / (g("SQ_BUSY_CYCLES") / 32 if g("SQ_BUSY_CYCLES") == g("SQ_BUSY_CYCLES") else 1), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
    g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES"), g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), g("SQ_LDS_IDX_ACTIVE") / frames, g("SQ_BUSY_CYCLES")))
print("   raw", {k: "%.4g" % v for k, v in m.items()})
PY
rm -rf $O/a $O/b
