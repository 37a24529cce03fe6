#!/bin/bash
# GPU probe: the rocprofv3 summaries that profiles/ keeps for a round.  Kernel statistics and the HBM counters of the
# bench command (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, no tracing options beside them), SQ counters of the
# three heaviest kernels, kernel statistics of the random and deep-repeat workloads.
#   tests/gpu_profiles.sh <tag>      -> gpurun_out/profiles_<tag>/
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
BENCH="$root/bench.py --steps 4 --warmup 1 --profile-run"
stats() {      # name, bench args...
    name=$1; shift
    rm -rf $out/tmp_$name
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/tmp_$name -o k -- python3 $BENCH "$@" > $out/${name}_bench.log 2>&1
    cp $(find $out/tmp_$name -name '*kernel_stats.csv' | head -1) $out/${tag}_${name}kernel_stats.csv
    grep "^{\"metric\"" $out/${name}_bench.log | tail -1 > $out/${tag}_${name}bench_line.json
    rm -rf $out/tmp_$name
}
pmc() {        # name, counters...
    name=$1; shift
    rm -rf $out/tmp_$name
    rocprofv3 --pmc "$@" --output-format csv -d $out/tmp_$name -o p -- python3 $BENCH > $out/${name}_bench.log 2>&1
    cp $(find $out/tmp_$name -name '*counter_collection.csv' | head -1) $out/${tag}_pmc_${name}.csv
    rm -rf $out/tmp_$name
}
stats ""
stats random_ --workload random --mib 256
stats dups_ --workload dups --mib 256
pmc fetch_size FETCH_SIZE
pmc write_size WRITE_SIZE
pmc sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY
python3 $root/tests/gpu_probe_sq.py $out $out/${tag}_sq_summary.csv > /dev/null || true
python3 - <<PY
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open("$out/${tag}_pmc_sq.csv")):
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"]
with open("$out/${tag}_sq_summary.csv", "w") as f:
    f.write("kernel," + ",".join(names) + ",wait_any/wave_cycles,lds_bank_conflict/lds_idx_active\n")
    for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"])[:8]:
        a = acc[k]
        f.write(k + "," + ",".join(f"{a[c]:.0f}" for c in names) + f",{a['SQ_WAIT_ANY']/max(1,a['SQ_WAVE_CYCLES']):.3f},{a['SQ_LDS_BANK_CONFLICT']/max(1,a['SQ_LDS_IDX_ACTIVE']):.3f}\n")
PY
ls $out
