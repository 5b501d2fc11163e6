"""Instruction-mix histogram per kernel from a `hipcc -save-temps` gfx950 .s file.
usage: python tools/isa_hist.py file.s [name substring ...]"""
import collections
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2:]
parts = re.split(r"^(_ZN4cmbl\w+):[^\n]*\n", s, flags=re.M)
names, bodies = parts[1::2], parts[2::2]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")


def group(k):
    for p, g in (("v_pk", "valu_pk"), ("v_", "valu"), ("ds_read", "ds_read"), ("ds_write", "ds_write"), ("global_load", "gload"),
                 ("global_store", "gstore"), ("scratch", "scratch"), ("s_waitcnt", "waitcnt"), ("s_barrier", "barrier"), ("s_", "salu")):
        if k.startswith(p):
            return g
    return k


for body, d in zip(bodies, dem):
    d = re.sub(r"\(.*", "", d)
    if not all(f in d for f in flt):
        continue
    body = body.split("s_endpgm")[0]
    ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    g = collections.Counter()
    for k, v in c.items():
        g[group(k)] += v
    print(d, "total", len(ins))
    print("   ", dict(g))
    print("    top:", c.most_common(12))
