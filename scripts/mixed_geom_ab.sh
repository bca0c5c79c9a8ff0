#!/bin/bash
# Development aid (GPU box), round 6: the mixed-radix / constant-Q octave kernels under different workgroup geometries (probe builds: -DLRA_MIXED_NT, _FMAX, _LDS_KB), one box.
cd "$(dirname "$0")/.."
for lib in "$@"; do
  echo "== $lib"
  for cfg in "400 160 80 16000" "1200 300 128 22050" "3200 800 128 22050"; do
    for what in stft mel; do LIBROSA_AMD_LIBRARY=probe/lib_$lib.so timeout 100 python scripts/mixed_probe.py $cfg $what 5 2>&1 | grep n_fft; done
  done
  LIBROSA_AMD_LIBRARY=probe/lib_$lib.so timeout 100 python scripts/cqt_merge_probe.py 2>&1 | grep "polyphase cqt_merge 1" | tail -1
done
