"""Register / LDS / scratch table of the kernels in a -save-temps gfx950 assembly file (hipcc ... -save-temps=obj).
usage: python tools/kernel_regs.py exponentialutilities.jl_amd/build/pipe-hip-amdgcn-amd-amdhsa-gfx950.s [substring ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
filt = sys.argv[2:]
rows = []
for b in re.split(r'\n\s*- \.agpr_count:', txt)[1:]:
    name = re.search(r'\.name:\s+(\S+)', b)
    if not name:
        continue
    g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, b).group(1))
    rows.append((name.group(1), g('vgpr_count'), int(re.match(r'\s*(\d+)', b).group(1)), g('vgpr_spill_count'), g('sgpr_spill_count'),
                 g('group_segment_fixed_size'), g('private_segment_fixed_size')))
names = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("vgpr agpr vspill sspill   lds scratch  kernel")
for r, d in zip(rows, names):
    d = d.replace('expv_mi::dev::', '').replace('HIP_vector_type<double, 2u>', 'cplx').replace('HIP_vector_type<float, 2u>', 'cplx32')
    if all(f in d for f in filt):
        print("%4d %4d %6d %6d %6d %6d  %s" % (r[1:] + (d[:160],)))
