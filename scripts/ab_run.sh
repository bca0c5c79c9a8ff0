#!/bin/bash
# A/B of probe builds against the product library on the GPU box (BASELINE configs[1] shapes).
#   scripts/probe_build.sh <name> "<flags>" ...           in the build container (probe/lib_<name>.so)
#   gpurun -- 'bash scripts/ab_run.sh "mel stft istft" name1 name2 ...'      (probe/ must not be listed in .gpurunignore)
# Every variant is timed twice, alternating with the product; then the parity cases that the probe builds cover.
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
kinds=$1; shift
NF=${AB_NFFT:-2048}; HOP=${AB_HOP:-512}   # other sizes: probe builds with -DLRA_PROBE_LOGM=.. (checked against torch.stft instead of the parity cases)
for round in 1 2; do
  for what in $kinds; do
    echo -n "product $what: "; timeout 120 python scripts/size_probe.py $NF $HOP 30 $what 2>&1 | grep n_fft
    for v in "$@"; do
      [ -f probe/lib_$v.so ] || continue
      echo -n "$v $what: "; LIBROSA_AMD_LIBRARY=probe/lib_$v.so timeout 120 python scripts/size_probe.py $NF $HOP 30 $what 2>&1 | grep n_fft
    done
  done
done
for v in "$@"; do
  [ -f probe/lib_$v.so ] || continue
  if [ "$NF" != 2048 ]; then echo -n "$v "; PROBE_CHECK=1 LIBROSA_AMD_LIBRARY=probe/lib_$v.so timeout 120 python scripts/size_probe.py $NF $HOP 2 stft 2>&1 | grep check; continue; fi
  echo -n "$v parity: "; LIBROSA_AMD_LIBRARY=probe/lib_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "golden_config or full_size_mel or full_size_stft_istft" 2>&1 | tail -1
done
