"""Turns the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as gpurun requires) into
profiles/traffic.json: HBM bytes per bench step of every kernel.

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
    """Counter total per bench step for every kernel (steps = dispatches of the once-per-step layout kernel).
    Per step rather than per dispatch: the sort kernel is launched twice per step, and under counter collection
    the profiler serialises kernels, so the library's overlap check falls back to one launch after the first step --
    the per-step total is the same either way."""
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    steps = max(1, len(acc.get("bzx_layout_kernel", [])))
    return {k: sum(v) / steps for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}, steps


def main():
    fetch, nf, steps = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        kernels[k] = {"dispatches": nf.get(k, 0), "FETCH_SIZE_KiB_per_step": round(f, 1),
                      "WRITE_SIZE_KiB_per_step": round(w, 1), "hbm_bytes_per_step": int((2.0 * f + w) * 1024)}
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), " + sys.argv[3],
           "steps": steps,
           "units": "counter values are KiB (x1024 = bytes), summed over a kernel's dispatches and divided by the bench steps; "
                    "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> fetch "
                    "doubled; scattered 8-byte accesses are uncalibrated",
           "kernels": kernels}
    # per launch of the kernels bench.py's roofline object may name (one launch per step each)
    for k in ("bzx_bsort_kernel", "bzx_bsplit_kernel", "bzx_bwt_kernel"):
        if k in kernels:
            out[k + "_hbm_bytes_per_launch"] = kernels[k]["hbm_bytes_per_step"] * steps // max(1, kernels[k]["dispatches"])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, {k: v for k, v in out.items() if k.endswith("per_launch")})


if __name__ == "__main__":
    main()
