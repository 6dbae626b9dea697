#!/usr/bin/env python3
"""Condense `hipcc -Rpass-analysis=kernel-resource-usage` stderr: one line per kernel (VGPRs, SGPRs, spills, scratch, occupancy).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -c X.hip -o /tmp/x.o -Rpass-analysis=kernel-resource-usage 2> /tmp/x.res
    python experiments/tools/resusage.py /tmp/x.res [name filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r'remark: Function Name: ', txt)[1:]
for b in blocks:
    name = b.split()[0]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '-'
    if flt and flt not in name:
        continue
    short = re.sub(r'\(.*', '', name)
    print(f"{short[:100]:100s} V {g('    VGPRs'):>3s} A {g('AGPRs'):>3s} S {g('TotalSGPRs'):>3s} vspill {g('VGPRs Spill'):>3s} sspill {g('SGPRs Spill'):>3s} scratch {g('ScratchSize .bytes/lane.'):>4s} occ {g('Occupancy .waves/SIMD.')}")
