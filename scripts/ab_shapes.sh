#!/bin/bash
# Development aid (GPU box): A/B of probe builds over the small-frame mel shapes, one box, alternating.  bash scripts/ab_shapes.sh <n_fft> lib1 lib2 ...
cd "$(dirname "$0")/.."
NF=$1; shift
run() { LIBROSA_AMD_LIBRARY=probe/lib_$1.so PROBE_MELS=$4 timeout 100 python scripts/size_probe.py $2 $3 20 mel 2>&1 | grep n_fft | sed "s/^/$1 /; s/$/ n_mels $4/"; }
for round in 1 2; do
  for hm in "$NF 128" "$((NF / 4)) 80" "$((NF / 4)) 128" "160 80" "$((NF / 4)) 40" "$((NF / 4)) 64"; do set -- $hm "$@"; h=$1; m=$2; shift 2; for lib in "$@"; do run $lib $NF $h $m; done; done
done
