#!/bin/bash
# Development aid: per-kernel register / spill / LDS figures of a built library (from the code object's metadata notes).
#   scripts/co_info.sh probe/lib_x.so [regex on the demangled kernel name]
so=$1; pat=${2:-stft2_kernel}
t=$(mktemp -d)
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy -O binary --only-section=.hip_fatbin "$so" $t/fat.bin
$L/clang-offload-bundler --unbundle --type=o --input=$t/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/k.co
$L/llvm-readelf --notes $t/k.co | python3 -c "
import sys,re,subprocess
txt=sys.stdin.read()
ks=re.split(r'\n\s+- \.agpr_count', txt)
rows=[]
for k in ks[1:]:
    g=lambda key:(re.search(r'\.'+key+r':\s+(\S+)',k) or [None,'?'])[1]
    rows.append((g('name'),g('vgpr_count'),g('vgpr_spill_count'),g('sgpr_count'),g('group_segment_fixed_size'),g('private_segment_fixed_size')))
dem=subprocess.run(['c++filt'],input='\n'.join(r[0] for r in rows),capture_output=True,text=True).stdout.split('\n')
for n,r in zip(dem,rows):
    n=re.sub(r'lra::FftCfg<(\d+), (\d), (\w+), (\d+), (\d), (\w+), (\w+)>',r'Cfg<\1,R\2,\3,NT\4,W\5,\6,\7>',n); n=re.sub(r'\(.*','',n)
    if re.search(r'''$pat''',n):
        print(f'{n:70s} vgpr {r[1]} spill {r[2]} sgpr {r[3]} lds {r[4]} scratch {r[5]}')
"
rm -rf $t
