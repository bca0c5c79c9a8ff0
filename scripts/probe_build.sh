#!/bin/bash
# Development aid: single-TU probe builds of the library (n_fft = 2048 f32 kernels only, ~1 min each, built side by side).
# Usage: scripts/probe_build.sh name1 "flags1" [name2 "flags2" ...]   ->  probe/lib_<name>.so
R=/root/repo
mkdir -p $R/probe
pids=()
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DLRA_PROBE_ONLY $flags $R/librosa_amd/csrc/lra_api.hip $R/librosa_amd/csrc/lra_mixed_inst.hip -o $R/probe/lib_$name.so \
      -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib > /tmp/probe_$name.log 2>&1 || { echo "probe build $name FAILED"; grep -E "error" /tmp/probe_$name.log | head -5; } ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
ls -la $R/probe/*.so
