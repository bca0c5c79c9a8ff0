#!/bin/bash
# GPU box: per-kernel durations of one librosa_amd.cqt configuration (rocprofv3 kernel trace).  scripts/cqt_trace.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/cqt_trace; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python $R/scripts/cqt_hostprof.py > $O/log.txt 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$O/r_kernel_trace.csv")))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"][:70], r["Grid_Size_X"], r["LDS_Block_Size"])].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:22]:
    print(f"{k[0]:70s} grid {k[1]:>9s} lds {k[2]:>6s} n {len(v):5d} avg {sum(v)/len(v):8.1f} us")
PY
python - <<PY
# one call's timeline (the 300th cqt call): start / end in us relative to the call's first kernel
import csv
rows = sorted(csv.DictReader(open("$O/r_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
ours = [r for r in rows if "lra::" in r["Kernel_Name"]]
per = 9  # kernels per call when 7 octaves: 6 halvings + first octave + two merged launches (cqt_merge 1)
firsts = [i for i, r in enumerate(ours) if "mixed_cqt_kernel" in r["Kernel_Name"] or ("fir_halve4" in r["Kernel_Name"] and r["Grid_Size_X"] == "5292032")]
i0 = [i for i, r in enumerate(ours) if "fir_halve4" in r["Kernel_Name"] and r["Grid_Size_X"] == "5292032"][200]
t0 = min(int(ours[i0]["Start_Timestamp"]), int(ours[i0 - 1]["Start_Timestamp"]) if i0 else 1 << 62)
for r in ours[max(0, i0 - 1): i0 + per + 1]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:8.1f} us  queue {r.get('Queue_Id', '?'):>3s}  grid {r['Grid_Size_X']:>9s}  {r['Kernel_Name'][:60]}")
PY
