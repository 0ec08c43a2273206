#!/bin/bash
# Kernel-level profile of the classic bilateral path on the GPU box (1080p D=128 win 15):
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/prof_bilateral.sh <tag> [variants...]'
# per ASW_XQ_ABLATE value (0 = the product; 1 no staging, 2 no barriers, 3 both: timing experiments -- these need a library
# built with  HIPCC_EXTRA=-DASW_XQ_ABLATION python -m aswstereomatch_amd.build --force;  the shipped build ignores the variable) or "old": rocprofv3 kernel stats of tools/run_one.py and two SQ counter passes (separate runs, --pmc never
# combined with other trace domains).  Output: gpurun_out/<tag>/.
set -e -o pipefail
TAG=${1:-bil}; shift || true
VARS=${@:-0}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SQ_A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES"
SQ_B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM"
for v in $VARS; do
    export ASW_XQ_ABLATE=$v
    if [ "$v" = "old" ]; then export ASW_BILATERAL_XQ=0; else unset ASW_BILATERAL_XQ; fi
    rocprofv3 --kernel-trace --stats -d /tmp/st_$v -o p --output-format csv -- python3 "$ROOT/tools/run_one.py" --alg 2 --reps 5 > "$OUT/stats_$v.log" 2>&1
    cp /tmp/st_$v/p_kernel_stats.csv "$OUT/kernel_stats_var$v.csv"
    grep -E "bilateral|Name" "$OUT/kernel_stats_var$v.csv" | cut -c1-200
    rocprofv3 --kernel-trace --pmc $SQ_A -d /tmp/pa_$v -o p --output-format csv -- python3 "$ROOT/tools/run_one.py" --alg 2 --reps 2 > "$OUT/pmcA_$v.log" 2>&1
    grep bilateral /tmp/pa_$v/p_counter_collection.csv > "$OUT/pmc_SQ_A_var$v.csv" || true
    rocprofv3 --kernel-trace --pmc $SQ_B -d /tmp/pb_$v -o p --output-format csv -- python3 "$ROOT/tools/run_one.py" --alg 2 --reps 2 > "$OUT/pmcB_$v.log" 2>&1 || echo "pmc B failed for $v"
    grep bilateral /tmp/pb_$v/p_counter_collection.csv > "$OUT/pmc_SQ_B_var$v.csv" || true
    echo "variant $v done"
done
