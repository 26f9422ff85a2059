#!/usr/bin/env python
"""Registers / scratch per kernel of one translation unit: hipcc ... -save-temps=obj leaves <unit>-hip-amdgcn-amd-amdhsa-gfx950.s;
usage: scripts/kernel_regs.py recsys_amd/csrc/cin_split.hip [grep-pattern]"""
import os, re, subprocess, sys, tempfile
src = os.path.abspath(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                    "-I" + os.path.join(root, "include"), "-I" + os.path.dirname(src), "-c", src, "-o", os.path.join(d, "o.o"),
                    "-save-temps=obj"] + sys.argv[3:], check=True, cwd=d)
    name = os.path.splitext(os.path.basename(src))[0]
    s = open(os.path.join(d, name + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', s):
    if pat in m.group(1):
        print("%-90s scratch %4s  vgpr %4s  spill %3s" % (m.group(1)[:90], m.group(2), m.group(3), m.group(4)))
