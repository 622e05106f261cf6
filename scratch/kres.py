#!/usr/bin/env python3
"""Kernel resource table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage): name, VGPRs, AGPRs, scratch, occupancy.
usage: python scratch/kres.py dp_gp_lvm_amd/csrc/psi2_pairs_grad.hip [name filter]"""
import re, subprocess, sys
src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=fast', '-mllvm', '-amdgpu-mfma-vgpr-form',
       '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/tmp/kres.o'] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r'remark: (.*) \[-Rpass', line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}; rows.append(cur)
    elif ':' in t:
        k, v = t.split(':', 1); cur[k.strip()] = v.strip()
for r in rows:
    if flt in r['name']:
        name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
        print(name[:70].ljust(70), 'v', r.get('VGPRs', '?').rjust(3), 'a', r.get('AGPRs', '?').rjust(3), 'scr', r.get('ScratchSize [bytes/lane]', '?').rjust(4),
              'occ', r.get('Occupancy [waves/SIMD]', '?'), 'lds', r.get('LDS Size [bytes/block]', '?'))
