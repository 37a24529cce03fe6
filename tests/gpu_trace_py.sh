#!/bin/bash
# GPU probe: per-launch kernel durations of one compression of real files (python sources), in launch order.
#   tests/gpu_trace_py.sh [corpus]      -> gpurun_out/trace_<corpus>.txt
set -e
which=${1:-py}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $out/tmp_trace
rocprofv3 --kernel-trace --output-format csv -d $out/tmp_trace -o t -- python3 $root/tests/gpu_probe_bsort.py ${MIB:-128} $which > $out/trace_${which}_probe.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$out/tmp_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last compression: everything after the last split/rle launch sequence start
names = [r["Kernel_Name"].split("(")[0] for r in rows]
last = max(i for i, n in enumerate(names) if "bsplit_kernel" in n)
t0 = int(rows[last]["Start_Timestamp"])
with open("$out/trace_${which}.txt", "w") as o:
    for r in rows[last:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        o.write(f"{(s - t0)/1e6:9.3f} ms  +{(e - s)/1e6:8.3f} ms  {r['Kernel_Name'].split('(')[0]}  grid {r.get('Grid_Size_X', '?')}\n")
PY
rm -rf $out/tmp_trace
