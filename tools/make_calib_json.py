#!/usr/bin/env python3
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on known-size copies (tools/micro/calib_copy.hip):

    python tools/make_calib_json.py fetch_counter_collection.csv write_counter_collection.csv out.json

For every (variant, bytes per lane, array size) the JSON holds the counter value in bytes (KB x 1024), the true byte count and their
ratio; `factors` is what tools/make_traffic_json.py multiplies the raw counters with (the mean over the 1 GiB launches of the
8- and 16-byte variants, the widths the flow kernels use).
"""
import collections
import csv
import json
import re
import sys

SIZES = {0: 64 << 20, 1: 1 << 30}
READS = {"copy_plain": 1, "copy_ntstore": 1, "copy_ntload": 1, "read_only": 1, "write_only": 0}
WRITES = {"copy_plain": 1, "copy_ntstore": 1, "copy_ntload": 1, "read_only": 0, "write_only": 1}


def collect(path, counter):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        m = re.search(r"(copy_plain|copy_ntstore|copy_ntload|read_only|write_only)<(\d+), (\d+)", row["Kernel_Name"])
        if m:
            acc[(m.group(1), int(m.group(2)), int(m.group(3)))].append(float(row["Counter_Value"]) * 1024.0)
    return acc


def main():
    fcsv, wcsv, out = sys.argv[1:4]
    f, w = collect(fcsv, "FETCH_SIZE"), collect(wcsv, "WRITE_SIZE")
    rows = []
    for key in sorted(set(f) | set(w)):
        var, width, big = key
        nbytes = SIZES[big]
        fv = sum(f[key]) / len(f[key]) if key in f else None
        wv = sum(w[key]) / len(w[key]) if key in w else None
        rows.append({"kernel": var, "bytes_per_lane": width, "array_bytes": nbytes,
                     "true_read_bytes": nbytes * READS[var], "true_write_bytes": nbytes * WRITES[var],
                     "FETCH_SIZE_bytes": fv, "WRITE_SIZE_bytes": wv,
                     "read_over_FETCH": (nbytes * READS[var] / fv) if fv and READS[var] else None,
                     "write_over_WRITE": (nbytes * WRITES[var] / wv) if wv and WRITES[var] else None})

    def mean(vals):
        vals = [v for v in vals if v]
        return sum(vals) / len(vals) if vals else None
    sel = [r for r in rows if r["array_bytes"] == SIZES[1] and r["bytes_per_lane"] in (8, 16)]
    factors = {"fetch": mean(r["read_over_FETCH"] for r in sel if r["kernel"] in ("copy_plain", "read_only")),
               "write": mean(r["write_over_WRITE"] for r in sel if r["kernel"] in ("copy_plain", "write_only")),
               "write_nontemporal": mean(r["write_over_WRITE"] for r in sel if r["kernel"] == "copy_ntstore"),
               "fetch_nontemporal": mean(r["read_over_FETCH"] for r in sel if r["kernel"] == "copy_ntload")}
    json.dump({"note": "true bytes / counter bytes of streaming kernels of known size (tools/micro/calib_copy.hip), separate --pmc passes. "
                       "The counters sit on the fabric side of the L2: Infinity-Cache (256 MiB) hits are INCLUDED, so `traffic` in the bench "
                       "lines is L2<->fabric bytes, an upper bound of the HBM bytes; the 64 MiB rows show a cache-resident working set "
                       "counted like the 1 GiB one.",
               "factors": factors, "rows": rows}, open(out, "w"), indent=1)
    print(json.dumps(factors))
    for r in rows:
        print(f"{r['kernel']:13s} {r['bytes_per_lane']:2d} B/lane {r['array_bytes'] >> 20:5d} MiB  read/FETCH {r['read_over_FETCH']}  write/WRITE {r['write_over_WRITE']}")


if __name__ == "__main__":
    main()
