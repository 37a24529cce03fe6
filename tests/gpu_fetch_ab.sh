#!/bin/bash
# Ad-hoc GPU probe: FETCH_SIZE (rocprofv3 --pmc, KiB -> GB, not doubled) of the three heaviest kernels per launch, for
# libbzx.so and any variant builds libbzx_<v>.so named in VARIANTS.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp; export TMPDIR=/tmp
for v in ${VARIANTS:-""}; do
  lib=$root/bzip2-rust_amd/libbzx${v:+_$v}.so
  rm -rf $out/tmp_ab
  BZX_LIB=$lib rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/tmp_ab -o p -- python3 $root/bench.py --steps 2 --warmup 1 --profile-run > $out/ab_$v.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$out/tmp_ab/**/*counter_collection.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if r["Counter_Name"] == "FETCH_SIZE": acc.setdefault(k, []).append(float(r["Counter_Value"]))
for k in ("bzx_bsort_kernel", "bzx_bsplit_kernel", "bzx_mtf_kernel"):
    print("$v", k, [round(x * 1024 / 1e9, 2) for x in acc.get(k, [])])
PY
done
rm -rf $out/tmp_ab
