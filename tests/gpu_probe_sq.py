"""Ad-hoc GPU probe: sums the SQ counters of a rocprofv3 --pmc run per kernel (counter_collection.csv -> table).

    python tests/gpu_probe_sq.py <dir with *_counter_collection.csv> [out.csv]"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key); calls[k] += 1
names = sorted({c for k in acc for c in acc[k]})
rows = [["kernel", "dispatches"] + names]
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    rows.append([k, calls[k]] + [f"{acc[k].get(c, 0):.0f}" for c in names])
out = "\n".join(",".join(map(str, r)) for r in rows)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
