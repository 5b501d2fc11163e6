"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: VGPRs, scratch, occupancy per kernel.
usage: python tools/res_usage.py remarks.txt [substring filter ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2:]
names, rows = [], []
for b in re.split(r"remark:[^\n]*?Function Name: ", txt)[1:]:
    name = b.split()[0]
    g = lambda k: int(m.group(1)) if (m := re.search(re.escape(k) + r": (\d+)", b)) else -1
    names.append(name)
    rows.append((g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("ScratchSize [bytes/lane]"), g("VGPRs Spill"), g("Occupancy [waves/SIMD]")))
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
print("vgpr agpr sgpr scratch spill occ  kernel")
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d)
    if all(f in d for f in flt):
        print("%4d %4d %4d %7d %5d %3d  %s" % (*r, d))
