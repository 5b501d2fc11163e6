#!/usr/bin/env python3
"""Summarise a CMBL_PARITY_LOG (tests/_tol.py) of a GPU run: per test-function and comparison, the largest measured error, the
tolerance it was checked against and their ratio.    python tools/parity_report.py gpurun_out/parity.jsonl > profiles/rNN_parity_measured.txt"""
import collections
import json
import re
import sys

rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    r = json.loads(line)
    m = re.match(r"(.*?)::(\w+)(\[(.*)\])?", r["test"])
    fn, par = (m.group(1).split("/")[-1] + "::" + m.group(2), m.group(4) or "") if m else (r["test"], "")
    prec = "f32" if "f32" in par else ("f64" if "f64" in par else "")
    key = (fn, prec, re.sub(r"\d+", "#", r["what"]) if "--collapse" in sys.argv else r["what"])
    e = rows.setdefault(key, {"err": 0.0, "tol": r["tol"], "n": 0})
    e["err"] = max(e["err"], r["err"]); e["tol"] = min(e["tol"], r["tol"]); e["n"] += 1
print(f"{'test':58s} {'prec':4s} {'comparison':44s} {'n':>4s} {'max err':>10s} {'tol':>9s} {'tol/err':>8s}")
for (fn, prec, what), e in rows.items():
    ratio = e["tol"] / e["err"] if e["err"] > 0 else float("inf")
    print(f"{fn[:58]:58s} {prec:4s} {what[:44]:44s} {e['n']:4d} {e['err']:10.2e} {e['tol']:9.1e} {ratio:8.1f}")
