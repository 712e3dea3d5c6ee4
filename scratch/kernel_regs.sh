#!/bin/bash
# scratch/kernel_regs.sh [file.hip]: registers, spills, LDS and scratch of every kernel of a .hip file (compiled for gfx950 with the
# Makefile's flags into /tmp; reads the metadata notes of the device code object).
f=${1:-raytracer_amd/csrc/render_wavefront.hip}
shift
out=/tmp/regs_$$
mkdir -p $out
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -S "$f" -o $out/dev.s "$@" || exit 1
cp $out/dev.s /tmp/last_kernels.s
sed -n '/^amdhsa.kernels:/,$p' $out/dev.s | python3 -c "
import sys,re,subprocess
txt=sys.stdin.read()
blocks=re.split(r'\n\s+- \.agpr_count', txt)
rows=[]
for b in blocks[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)',b) or [None,'?'])[1]
    n=g('name')
    d=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()
    d=re.sub(r'^void rt::\(anonymous namespace\)::','',d); d=re.sub(r'\(.*','',d)
    rows.append((d,g('vgpr_count'),g('vgpr_spill_count'),g('sgpr_count'),g('group_segment_fixed_size'),g('private_segment_fixed_size')))
for r in sorted(rows): print('%-60s vgpr %3s spill %3s sgpr %3s lds %6s scratch %5s'%r)
"
rm -rf $out
