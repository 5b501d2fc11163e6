#!/usr/bin/env python3
"""Build profiles/rNN_traffic_<workload>.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected separately):

    python tools/make_traffic_json.py fetch_counter_collection.csv write_counter_collection.csv out.json nside npol nbatch dtype [nevals]

Mean KB per launch per kernel, then grouped into bench.py's kernel classes.  gfx950 correction (MI355X_MICROARCH.md, HBM
section): FETCH_SIZE reports half the bytes of wide coalesced reads -> traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.
"""
import collections
import csv
import json
import re
import sys

CLASSES = {"k_flow_y_fwd": "flow_y_fwd", "k_x_fft<float, 2": "x_grad", "k_x_fft<double, 2": "x_grad", "k_adj_y": "adj_y",
           "k_adj_x": "adj_x", "k_delta_rows": "delta_rows", "k_delta_cols": "delta_cols", "k_dphi_reduce": "dphi_reduce"}


def mean_per_kernel(path, counter):
    acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "")
        acc[k] += float(row["Counter_Value"])
        cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}, cnt


def main():
    fcsv, wcsv, out, nside, npol, nbatch, dtype = sys.argv[1:8]
    nsteps = int(sys.argv[8]) if len(sys.argv) > 8 else 0            # ∇lnP evaluations in the profiled run (steps + warmup): per-step total
    f, nf = mean_per_kernel(fcsv, "FETCH_SIZE")
    w, _ = mean_per_kernel(wcsv, "WRITE_SIZE")
    kernels = {k: {"FETCH_SIZE_KB": f[k], "WRITE_SIZE_KB": w.get(k, 0.0), "launches": nf[k]} for k in f if k.startswith("cmbl::")}
    by_class = {}
    for k, v in kernels.items():
        for pat, cls in CLASSES.items():
            if pat in k:
                by_class[cls] = {"traffic_bytes_per_launch": (2 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024, **v, "kernel": k}
    total = sum((2 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024 * v["launches"] for v in kernels.values())
    json.dump({"workload": {"nside": int(nside), "npol": int(npol), "nbatch": int(nbatch), "dtype": dtype},
               "total_bytes_per_step": (total / nsteps) if nsteps else None,
               "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 4 --warmup 1`; "
                       "values are mean KB per launch. Per MI355X_MICROARCH.md §HBM, FETCH_SIZE on gfx950 reports half of the bytes "
                       "of wide coalesced reads: traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.",
               "kernels": kernels, "by_class": by_class}, open(out, "w"), indent=1, ensure_ascii=False)
    for c, v in by_class.items():
        print(f"{c:12s} {v['traffic_bytes_per_launch'] / 1e6:8.1f} MB/launch")


if __name__ == "__main__":
    main()
