#!/bin/bash
# Development aid (GPU box): the mixed-radix fused kernels under different LDS budgets per workgroup (probe builds -DLRA_MIXED_LDS_KB=..), one box, alternating.
cd "$(dirname "$0")/.."
for round in 1 2; do
  for cfg in "400 160 80 16000" "800 200 128 16000" "1200 300 128 22050" "3200 800 128 22050" "1920 480 128 48000"; do
    for lib in "$@"; do
      for what in mel stft; do echo -n "$lib "; LIBROSA_AMD_LIBRARY=probe/lib_$lib.so timeout 100 python scripts/mixed_probe.py $cfg $what 5 2>&1 | grep n_fft; done
    done
  done
done
