#!/bin/bash
# Development aid: gfx950 assembly of the probe build (n_fft = 2048 f32 kernels) -> /tmp/isa/<tag>.s, and one kernel cut out of it.
#   scripts/isa_dump.sh <tag> "<extra flags>" ; scripts/isa_dump.sh cut <tag> '<mangled-name regex>' out.txt
if [ "$1" = cut ]; then
  f=/tmp/isa/$2.s
  L=$(grep -n "^$3.*:" $f | head -1 | cut -d: -f1)
  sed -n "${L},\$p" $f | awk '/\.amdhsa_next_free_sgpr/{print; exit} {print}' | grep -v "^\s*;" | sed 's/\s*;.*$//' > $4
  wc -l $4
  exit 0
fi
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLRA_PROBE_ONLY $2 -S --cuda-device-only -o /tmp/isa/$1.s /root/repo/librosa_amd/csrc/lra_api.hip 2>/dev/null
ls -la /tmp/isa/$1.s
