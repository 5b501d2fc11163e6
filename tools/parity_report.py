#!/usr/bin/env python3
"""Turn a CMBL_PARITY_LOG (tests/_tol.py) of a GPU run into
    --json      tests/golden/parity_measured.json : {key: error} -- the table the GPU suite tightens its tolerances with (<= 3 x measured)
    (default)   a text summary per test function and comparison: largest error, the tolerance it was checked against, their ratio
                (--collapse folds numbers in the comparison labels)
  python tools/parity_report.py gpurun_out/parity.jsonl --json > tests/golden/parity_measured.json
  python tools/parity_report.py gpurun_out/parity.jsonl --collapse > profiles/rNN_parity_measured.txt"""
import collections
import json
import re
import sys

recs = [json.loads(l) for l in open(sys.argv[1])]
if "--json" in sys.argv:
    out = {}
    for r in recs:
        out[r["key"]] = max(out.get(r["key"], 0.0), r["err"])          # a key seen in several runs (three-in-a-row): keep the largest
    json.dump(out, sys.stdout, indent=0, sort_keys=True, ensure_ascii=False)
    sys.exit(0)
rows = collections.OrderedDict()
for r in recs:
    test, what = r["key"].rsplit("#", 1)[0].split("|", 1)
    m = re.match(r"(.*?)::(\w+)(\[(.*)\])?", test)
    fn, par = (m.group(1).split("/")[-1] + "::" + m.group(2), m.group(4) or "") if m else (test, "")
    prec = "f32" if "f32" in par else ("f64" if "f64" in par else "")
    key = (fn, prec, re.sub(r"\d+", "#", what) if "--collapse" in sys.argv else what)
    e = rows.setdefault(key, {"err": 0.0, "tol": r["tol"], "n": 0})
    e["err"] = max(e["err"], r["err"]); e["tol"] = min(e["tol"], r["tol"]); e["n"] += 1
print(f"{'test':58s} {'prec':4s} {'comparison':44s} {'n':>4s} {'max err':>10s} {'min tol':>9s} {'tol/err':>8s}")
for (fn, prec, what), e in rows.items():
    ratio = e["tol"] / e["err"] if e["err"] > 0 else float("inf")
    print(f"{fn[:58]:58s} {prec:4s} {what[:44]:44s} {e['n']:4d} {e['err']:10.2e} {e['tol']:9.1e} {ratio:8.1f}")
