#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (one line per kernel)."""
import re, subprocess, sys
txt = open(sys.argv[1] if len(sys.argv) > 1 else 'tsqr_gpu_amd/csrc/resource_usage.txt').read()
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split()[0]
    def g(k):
        m = re.search(re.escape(k) + r': (\d+)', b)
        return m.group(1) if m else '?'
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'\(.*', '', dn)[:44]
    print("%-44s SGPR %4s VGPR %4s AGPR %3s scratch %4s occ %2s sspill %4s vspill %3s LDS %6s" % (
        dn, g('TotalSGPRs'), g('VGPRs'), g('AGPRs'), g('ScratchSize [bytes/lane]'), g('Occupancy [waves/SIMD]'),
        g('SGPRs Spill'), g('VGPRs Spill'), g('LDS Size [bytes/block]')))
