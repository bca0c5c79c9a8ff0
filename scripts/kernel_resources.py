"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: VGPRs / SGPRs / scratch / occupancy / spills per kernel.
Usage: python scripts/kernel_resources.py <remarks.txt> [regex on the demangled name]  (see scripts/build.sh)"""
import re,subprocess,sys
txt=open(sys.argv[1]).read()
blocks=txt.split('Function Name: ')[1:]
rows=[]
for b in blocks:
    name=b.split()[0]
    g=lambda key: (re.search(key+r': (\d+)',b) or [None,None])[1]
    rows.append((name,g('    VGPRs'),g('TotalSGPRs'),g(r'ScratchSize \[bytes/lane\]'),g(r'Occupancy \[waves/SIMD\]'),g('VGPRs Spill')))
dem=subprocess.run(['c++filt'],input='\n'.join(r[0] for r in rows),capture_output=True,text=True).stdout.split('\n')
pat=sys.argv[2] if len(sys.argv)>2 else 'Cfg<10'
for n,r in zip(dem,rows):
    n=re.sub(r'lra::FftCfg<(\d+), (\d), (\w+), \d+, (\d), (\w+)>',r'Cfg<\1,R\2,\3,W\4,\5>',n); n=re.sub(r'\(.*','',n)
    if (r[3] and int(r[3])>0) or re.search(pat,n): print(f"{n:58s} VGPR {r[1]} SGPR {r[2]} scratch {r[3]} occ {r[4]} spill {r[5]}")
