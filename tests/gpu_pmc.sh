#!/bin/bash
# GPU probe: rocprofv3 counter passes (one --pmc set per run, no tracing options) over a command; per-kernel sums go to
# gpurun_out/<tag>_<pass>.csv through tests/gpu_probe_sq.py.
#   tests/gpu_pmc.sh <tag> <python script and args...>
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp; export TMPDIR=/tmp
pass() {
    name=$1; shift
    rm -rf $out/pmc_${tag}_$name
    rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_${tag}_$name -o p -- python3 "${CMD[@]}" > $out/pmc_${tag}_$name.log 2>&1
    python3 $root/tests/gpu_probe_sq.py $out/pmc_${tag}_$name $out/${tag}_$name.csv > /dev/null
}
CMD=("$@")
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE
pass wr WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass rd FETCH_SIZE TCC_HIT_sum
pass l2 TCC_REQ_sum TCC_MISS_sum TCC_WRITE_sum TCC_READ_sum
echo done
