"""Gaps between consecutive kernels of the delta flow from a rocprofv3 kernel trace: python tools/trace_gaps.py x_kernel_trace.csv"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: (re.search(r"cmbl::(k_\w+)", n) or [None, n[:20]])[1]
chains = {"delta flow (k_delta_cols, k_delta_rows)": ("k_delta_cols", "k_delta_rows"), "forward flow (k_x_fft, k_flow_y_fwd)": ("k_x_fft", "k_flow_y_fwd"),
          "adjoint flow (k_adj_y, k_adj_x)": ("k_adj_y", "k_adj_x")}
for name, ks in chains.items():
    seq = [r for r in rows if short(r["Kernel_Name"]) in ks]
    gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
    for a, b in zip(seq, seq[1:]):
        g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        if g < 100000:
            gaps[short(a["Kernel_Name"]) + " -> " + short(b["Kernel_Name"])].append(g)
    for r in seq:
        durs[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if not seq:
        continue
    print(name)
    for k, v in durs.items():
        v.sort()
        print(f"   {k:34s} n={len(v):5d} median duration {v[len(v)//2]/1e3:7.2f} us")
    for k, v in gaps.items():
        v.sort()
        print(f"   {k:34s} n={len(v):5d} median gap      {v[len(v)//2]/1e3:7.2f} us")
