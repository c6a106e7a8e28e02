#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: registers, spills, LDS and occupancy per kernel.
usage: hipcc ... --offload-device-only -Rpass-analysis=kernel-resource-usage -c x.hip -o /dev/null 2> res.txt; kres.py res.txt [filter ...]"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
flt = sys.argv[2:]
for b in t.split('Function Name: ')[1:]:
    name = b.split()[0]
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    if flt and not any(x in dn for x in flt):
        continue
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    print(dn, 'VGPR', g('VGPRs'), 'AGPR', g('AGPRs'), 'spill', g('VGPRs Spill'), 'scratch', g(r'ScratchSize \[bytes/lane\]'),
          'occ', g(r'Occupancy \[waves/SIMD\]'), 'LDS', g(r'LDS Size \[bytes/block\]'))
