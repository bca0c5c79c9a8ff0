#!/bin/bash
# Build the gfx950 library (parallel, see librosa_amd/build.py).  With an argument: also print the register /
# scratch summary of the kernels whose demangled name matches it (recompiles one instance group with
# -Rpass-analysis=kernel-resource-usage; group 0 = n_fft 2048 f32 default configuration).
R=/root/repo
cd $R && python -m librosa_amd.build --force > /tmp/build_py.log 2>&1 || { tail -30 /tmp/build_py.log; echo "errors: 1"; exit 1; }
echo "errors: 0"
if [ -n "$1" ] && [ -f $R/scripts/kernel_resources.py ]; then
  cd $R/librosa_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DLRA_INST_GROUP=${2:-0} lra_inst.hip -o /tmp/inst_res.o -Rpass-analysis=kernel-resource-usage > /tmp/res.txt 2>&1
  python $R/scripts/kernel_resources.py /tmp/res.txt "$1" | grep -E "$1" | head -40
fi
exit 0
