#!/bin/bash
# Counter passes of one method on one resident frame (tools/run_one.py), the tuning switches taken from the caller's environment:
# FETCH_SIZE, WRITE_SIZE, L2 hits/misses and the SQ set, each in a pass of its own (--pmc is never combined with other traces).
#   ASW_RING_Q=2 ... bash tools/pmc_run.sh <tag> <name> --alg 8 [--width .. --height .. --disp ..]
set -e -o pipefail
TAG=$1; NAME=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pmc() { local c=$1; local ctr=$2; shift 2; rocprofv3 --kernel-trace --pmc $ctr -d "/tmp/pm_${NAME}_$c" -o p --output-format csv -- python3 "$ROOT/tools/run_one.py" "$@" --reps 1 > "$OUT/pmc_${NAME}_$c.log" 2>&1; cp "/tmp/pm_${NAME}_$c/p_counter_collection.csv" "$OUT/pmc_${NAME}_$c.csv"; }
pmc FETCH_SIZE "FETCH_SIZE" "$@"
pmc WRITE_SIZE "WRITE_SIZE" "$@"
pmc L2 "TCC_HIT_sum TCC_MISS_sum" "$@"
pmc SQ "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES" "$@"
python3 "$ROOT/tools/pmc_table.py" "$OUT" "$NAME"
