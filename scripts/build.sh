#!/bin/bash
# Build the gfx950 library with absolute paths; prints errors and the resource summary of the headline kernels.
R=/root/repo
cd $R/librosa_amd/csrc || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o $R/librosa_amd/_liblibrosa_amd.so lra_api.hip -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib -Rpass-analysis=kernel-resource-usage > /tmp/res.txt 2>&1
n=$(grep -c " error" /tmp/res.txt)
echo "errors: $n"
if [ "$n" != "0" ]; then grep " error" -A3 /tmp/res.txt | head -20; exit 1; fi
[ -f /tmp/summ2.py ] && python /tmp/summ2.py /tmp/res.txt "${1:-Cfg<10,R.,float,W.,true>}" | grep -E "${1:-true}" | head -20
exit 0
