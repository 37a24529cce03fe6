"""Turns the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as gpurun requires) into
profiles/traffic.json: average HBM bytes per dispatch of every kernel.

    python profiles/make_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> "<command profiled>"

Units (MI355X_MICROARCH.md, HBM / rocprofv3 section): the counters are in KiB per dispatch; on gfx950 FETCH_SIZE
reports half of wide coalesced reads, so fetch is doubled.  Scattered 8-byte accesses are uncalibrated (the
doubling overstates them), which makes the figure an upper bound for the gather-heavy phases."""
import csv
import json
import os
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        kernels[k] = {"dispatches": nf.get(k, 0), "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                      "hbm_bytes_corrected": int((2.0 * f + w) * 1024)}
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), " + sys.argv[3],
           "units": "counter values are KiB per dispatch (x1024 = bytes), averaged over the dispatches of a kernel; "
                    "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> fetch "
                    "doubled; scattered 8-byte accesses are uncalibrated",
           "kernels": kernels,
           "bzx_bwt_kernel_hbm_bytes_per_launch": kernels.get("bzx_bwt_kernel", {}).get("hbm_bytes_corrected")}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, out["bzx_bwt_kernel_hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
