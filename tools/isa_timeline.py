"""Where a kernel spills: order of barriers, global loads / stores and scratch loads / stores in a `hipcc -save-temps` gfx950 .s file.
usage: python tools/isa_timeline.py file.s mangled-name-prefix"""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r"^(" + re.escape(sys.argv[2]) + r"\w*):[^\n]*\n(.*?)s_endpgm", s, flags=re.M | re.S)
idx, out = 0, []
for l in m.group(2).split("\n"):
    t = l.strip()
    if not l.startswith("\t") or not t or t.startswith((".", ";")):
        continue
    idx += 1
    op = t.split()[0]
    k = ("BARRIER" if op == "s_barrier" else "scratch_st" if op.startswith("scratch_store") else "scratch_ld" if op.startswith("scratch_load")
         else "gload" if op.startswith("global_load") else "gstore" if op.startswith("global_store") else None)
    if k:
        out.append((idx, k))
print(m.group(1), "instructions:", idx)
cur, cnt, start = None, 0, 0
for i, k in out + [(0, None)]:
    if k != cur:
        if cur:
            print(f"{start:6d} {cur} x{cnt}")
        cur, cnt, start = k, 1, i
    else:
        cnt += 1
