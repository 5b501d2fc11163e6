"""Gaps between consecutive kernels of the delta flow from a rocprofv3 kernel trace: python tools/trace_gaps.py x_kernel_trace.csv"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: (re.search(r"cmbl::(k_\w+)", n) or [None, n[:20]])[1]
main = [r for r in rows if short(r["Kernel_Name"]) in ("k_delta_y", "k_delta_rows")]
side = [r for r in rows if short(r["Kernel_Name"]) in ("k_dphi_y", "k_dphi_x")]
for name, seq in (("main (delta_y, delta_rows)", main), ("side (dphi_y, dphi_x)", side)):
    gaps = collections.defaultdict(list)
    for a, b in zip(seq, seq[1:]):
        g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        if g < 200000:
            gaps[short(a["Kernel_Name"]) + " -> " + short(b["Kernel_Name"])].append(g)
    print(name)
    for k, v in gaps.items():
        v.sort()
        print(f"   {k:30s} n={len(v):5d} median gap {v[len(v)//2]/1e3:7.2f} us   mean {sum(v)/len(v)/1e3:7.2f} us")
dur = collections.defaultdict(list)
for r in rows:
    dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
# stage period on the main stream
dy = [int(r["Start_Timestamp"]) for r in main if short(r["Kernel_Name"]) == "k_delta_y"]
per = sorted(b - a for a, b in zip(dy, dy[1:]) if b - a < 300000)
print("delta_y start-to-start period: median %.1f us" % (per[len(per)//2] / 1e3))
fy = [int(r["Start_Timestamp"]) for r in rows if short(r["Kernel_Name"]) == "k_flow_y_fwd"]
per = sorted(b - a for a, b in zip(fy, fy[1:]) if b - a < 300000)
print("flow_y_fwd start-to-start period: median %.1f us" % (per[len(per)//2] / 1e3))
