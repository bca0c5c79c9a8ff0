#!/bin/bash
# scripts/size_prof.sh <tag> <n_fft> <hop> [what]: timing + two rocprofv3 SQ counter passes of one shape -> gpurun_out/sz_<tag>/
# (TCC counters -- FETCH_SIZE / WRITE_SIZE -- must not share a pass with SQ counters: rocprofv3 aborts and then hangs)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/sz_$1; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 120 python $R/scripts/size_probe.py $2 $3 10 ${4:-stft} | tee $OUT/time.txt
timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/a -o r -- python $R/scripts/size_probe.py $2 $3 2 ${4:-stft} > $OUT/a.log 2>&1
timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY -d $OUT/b -o r -- python $R/scripts/size_probe.py $2 $3 2 ${4:-stft} > $OUT/b.log 2>&1
find $OUT -name '*_agent_info.csv' -delete
python - <<PY
import csv, collections, glob
for sub in ("a", "b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            if "stft" in r["Kernel_Name"]:
                k = r["Kernel_Name"][:90] + " grid=" + r["Grid_Size"] + " wg=" + r["Workgroup_Size"] + " vgpr=" + r["VGPR_Count"] + " lds=" + r["LDS_Block_Size"]
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                agg[k]["dur_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    for k, v in agg.items():
        print(k); print("   ", {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "launches", len(v["dur_us"]))
PY
