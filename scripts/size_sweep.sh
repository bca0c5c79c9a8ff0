export TMPDIR=/tmp; R=$PWD
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mel or golden or config or full_size or random" 2>&1 | tail -2
for i in 1 2 3; do echo -n "al0: "; LIBROSA_AMD_LIBRARY=probe/lib_al0.so timeout 120 python scripts/size_probe.py 2048 512 30 mel; echo -n "new: "; timeout 120 python scripts/size_probe.py 2048 512 30 mel; done
cd /tmp
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/melw -o r -- python $R/scripts/size_probe.py 2048 512 2 mel > /dev/null 2>&1
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$R/gpurun_out/melw/r_counter_collection.csv")) if "stft2_kernel" in r["Kernel_Name"]]
v=[float(r["Counter_Value"]) for r in rows]
print("WRITE_SIZE MB per launch", sum(v)/len(v)*1024/1e6, "launches", len(v))
PY
