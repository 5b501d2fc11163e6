"""Aggregate a rocprofv3 kernel trace CSV by (kernel, grid, workgroup, LDS): mean duration and calls.
usage: python tools/trace_by_shape.py kernel_trace.csv [name filter]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(list)
for r in rows:
    if flt not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y"), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("LDS_Block_Size"))
    agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%8d calls  mean %8.1f us  total %8.1f ms  %s" % (len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6, k))
