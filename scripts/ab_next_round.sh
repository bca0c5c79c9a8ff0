#!/bin/bash
# A/B of the kernel experiments that are compiled in behind default-off flags, prepared at the end of round 2 when
# no GPU time was left.  Each variant passed the host simulator (LRA_HOSTSIM_DEFINES="-DLRA_V2_EARLY_PASS0=<n>" pytest
# tests/test_hostsim.py -k mel) and compiles without spills (scripts/kernel_resources.py).
#
#   scripts/ab_next_round.sh build     in the build container: probe/lib_early{1,2,3}.so (single-TU probe builds, ~1 min each)
#   gpurun -- 'bash scripts/ab_next_round.sh run'    on the GPU box; probe/ must travel: drop the "probe/" line from .gpurunignore
#                                                   for that call and restore it afterwards
# LRA_V2_EARLY_PASS0 = n: the window multiply and pass-0 butterflies of frame t + 1 (register-only work on registers that are dead
# once the power row is written) are issued inside the mel epilogue of frame t -- 1: after the run reads, 2: after the running
# sums' stores, 3: after the band combine, 4: window multiply at 1 and butterflies at 3 -- instead of at the top of the next frame, to fill the epilogue's LDS round trips.
# LRA_ISTFT_EARLY = 1: the same idea for the inverse kernel (see below).
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
case "$1" in
  build)
    bash scripts/probe_build.sh early1 "-DLRA_V2_EARLY_PASS0=1" early2 "-DLRA_V2_EARLY_PASS0=2" early3 "-DLRA_V2_EARLY_PASS0=3" early4 "-DLRA_V2_EARLY_PASS0=4" iearly "-DLRA_ISTFT_EARLY=1"
    ;;
  run)
    for round in 1 2; do
      echo -n "product: "; timeout 120 python scripts/size_probe.py 2048 512 30 mel 2>&1 | grep n_fft
      for v in early1 early2 early3 early4; do
        [ -f probe/lib_$v.so ] || continue
        echo -n "$v:  "; LIBROSA_AMD_LIBRARY=probe/lib_$v.so timeout 120 python scripts/size_probe.py 2048 512 30 mel 2>&1 | grep n_fft
      done
    done
    # the inverse kernel's analogue (LRA_ISTFT_EARLY=1: Hermitian step + first-pass butterflies of frame t + 1 at the end of frame t's
    # overlap-add phase; 27 simulator cases green, 186-201 VGPRs without spills)
    if [ -f probe/lib_iearly.so ]; then
      for round in 1 2; do
        echo -n "product istft: "; timeout 120 python scripts/size_probe.py 2048 512 30 istft 2>&1 | grep n_fft
        echo -n "iearly istft:  "; LIBROSA_AMD_LIBRARY=probe/lib_iearly.so timeout 120 python scripts/size_probe.py 2048 512 30 istft 2>&1 | grep n_fft
      done
      echo -n "iearly parity: "; LIBROSA_AMD_LIBRARY=probe/lib_iearly.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "full_size_stft_istft or golden_config1" 2>&1 | tail -1
    fi
    # parity of every variant on the mel cases (the probe builds hold the n_fft = 2048 float32 kernels only)
    for v in early1 early2 early3 early4; do
      [ -f probe/lib_$v.so ] || continue
      echo -n "$v parity: "; LIBROSA_AMD_LIBRARY=probe/lib_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "golden_config or full_size_mel" 2>&1 | tail -1
    done
    ;;
  *) echo "usage: $0 build | run"; exit 2;;
esac
