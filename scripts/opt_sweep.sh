#!/bin/bash
# Development aid (GPU box): one STFT shape under a list of run-time option settings (Context.set_option), product library or $LIBROSA_AMD_LIBRARY.
#   scripts/opt_sweep.sh n_fft hop what "stft_iters=48" "stft_iters=72,xcd_remap=0" ...
cd "$(dirname "$0")/.."
nf=$1; hop=$2; what=$3; shift 3
echo -n "(defaults) "; timeout 120 python scripts/size_probe.py $nf $hop 30 $what 2>&1 | grep n_fft
for o in "$@"; do echo -n "$o: "; PROBE_OPTS=$o timeout 120 python scripts/size_probe.py $nf $hop 30 $what 2>&1 | grep n_fft; done
