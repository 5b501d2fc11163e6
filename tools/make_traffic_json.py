#!/usr/bin/env python3
"""Build profiles/rNN_traffic_<workload>.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected separately):

    python tools/make_traffic_json.py fetch_counter_collection.csv write_counter_collection.csv out.json nside npol nbatch dtype nrk \
           [nunits] [calibration.json] [unit name]

Mean bytes per launch per kernel, grouped into the library's kernel classes (the names `cmbl_prof_*` / bench.py use).  The raw
counters are multiplied by the factors of the calibration file (tools/make_calib_json.py; without one: the guide's gfx950 rule
FETCH x 2, WRITE x 1 -- MI355X_MICROARCH.md, HBM section).  `nunits` = how many units of work (∇lnP evaluations, CG iterations) the
profiled command ran, for the per-unit total.
"""
import collections
import csv
import json
import re
import sys

# first match wins
CLASSES = [(r"k_flow_y_fwd", "flow_y_fwd"), (r"k_x_fft<\w+, 2", "x_grad"), (r"k_x_fft<\w+, [01]", "x_fft"), (r"k_adj_y", "adj_y"),
           (r"k_adj_x", "adj_x"), (r"k_delta_rows", "delta_rows"), (r"k_delta_cols", "delta_cols"), (r"k_delta_pol", "delta_cols"),
           (r"k_dphi_reduce", "dphi_reduce"),
           (r"k_dphi_combine", "dphi_combine"), (r"k_y_r2c", "y_r2c"), (r"k_y_c2r", "y_c2r"), (r"k_y_mask", "mask_mul"),
           (r"k_harm", "harm_apply"), (r"k_reduce|k_sign|k_max_step|k_min_final", "reduce"), (r"k_cg_", "cg_update"),
           (r"k_lincomb|k_map_fma|k_randn|k_mask_mul", "lincomb"), (r"k_gradhess|k_pcache", "gradhess_mult"),
           (r"k_ref2F|k_F2ref|k_transpose", "layout"), (r"k_gen_dft", "generic_dft"), (r"k_gen_", "generic_pointwise"), (r"k_qe_leg", "harm_apply")]


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
    fcsv, wcsv, out, nside, npol, nbatch, dtype, nrk = sys.argv[1:9]
    nunits = float(sys.argv[9]) if len(sys.argv) > 9 else 0
    calib = sys.argv[10] if len(sys.argv) > 10 and sys.argv[10] not in ("", "-") else None
    unit = sys.argv[11] if len(sys.argv) > 11 else "∇lnP evaluation"
    kf, kw, src = 2.0, 1.0, "MI355X_MICROARCH.md rule (FETCH x 2, WRITE x 1), uncalibrated"
    if calib:
        z = json.load(open(calib))["factors"]
        kf, kw, src = z["fetch"], z["write"], calib
    f, nf = mean_per_kernel(fcsv, "FETCH_SIZE")
    w, _ = mean_per_kernel(wcsv, "WRITE_SIZE")
    kernels = {k: {"FETCH_SIZE_KB": f[k], "WRITE_SIZE_KB": w.get(k, 0.0), "launches": nf[k],
                   "traffic_bytes_per_launch": (kf * f[k] + kw * w.get(k, 0.0)) * 1024} for k in f if k.startswith("cmbl::")}
    by_class = {}
    for k, v in kernels.items():
        for pat, cls in CLASSES:
            if re.search(pat, k):
                e = by_class.setdefault(cls, {"launches": 0, "bytes": 0.0, "read_bytes": 0.0, "write_bytes": 0.0, "kernels": []})
                e["launches"] += v["launches"]
                e["bytes"] += v["traffic_bytes_per_launch"] * v["launches"]
                e["read_bytes"] += kf * v["FETCH_SIZE_KB"] * 1024 * v["launches"]
                e["write_bytes"] += kw * v["WRITE_SIZE_KB"] * 1024 * v["launches"]
                e["kernels"].append(k)
                break
    for cls, e in by_class.items():
        e["traffic_bytes_per_launch"] = e.pop("bytes") / e["launches"]
        e["read_bytes_per_launch"] = e.pop("read_bytes") / e["launches"]
        e["write_bytes_per_launch"] = e.pop("write_bytes") / e["launches"]
        if nunits:
            e["launches_per_unit"] = e["launches"] / nunits
    total = sum(v["traffic_bytes_per_launch"] * v["launches"] for v in kernels.values())
    json.dump({"workload": {"nside": int(nside), "npol": int(npol), "nbatch": int(nbatch), "dtype": dtype, "nrk": int(nrk), "unit": unit},
               "counter_factors": {"fetch": kf, "write": kw, "source": src},
               "total_bytes_per_step": (total / nunits) if nunits else None,
               "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (kernel trace only alongside); mean per launch. "
                       "traffic_bytes = (fetch_factor*FETCH_SIZE + write_factor*WRITE_SIZE)*1024.  The counters sit between the L2 and the "
                       "fabric: Infinity-Cache hits are included, i.e. these are L2<->fabric bytes, an upper bound of the HBM bytes.",
               "by_class": by_class, "kernels": kernels}, open(out, "w"), indent=1, ensure_ascii=False)
    for c, v in sorted(by_class.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"]):
        print(f"{c:14s} {v['traffic_bytes_per_launch'] / 1e6:9.2f} MB/launch x {v['launches']}")
    if nunits:
        print(f"total {total / nunits / 1e9:.3f} GB per {unit}")


if __name__ == "__main__":
    main()
