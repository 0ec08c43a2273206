#!/bin/bash
# Stall analysis: several counter groups (one rocprofv3 --pmc pass each) of one method on one resident frame; per-kernel sums.
#   ASW_RING_AB=1 bash tools/pmc_groups.sh <tag> <name> <kernel substring> --alg 8
set -e -o pipefail
TAG=$1; NAME=$2; KSUB=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d "/tmp/pg_${NAME}_$i" -o p --output-format csv -- python3 "$ROOT/tools/run_one.py" "$@" --reps 1 > "$OUT/pg_${NAME}_$i.log" 2>&1 || { echo "group $i failed: $grp"; tail -3 "$OUT/pg_${NAME}_$i.log"; continue; }
  cp "/tmp/pg_${NAME}_$i/p_counter_collection.csv" "$OUT/pg_${NAME}_$i.csv"
  python3 - <<PY
import csv, collections
acc = collections.defaultdict(float)
for r in csv.DictReader(open("/tmp/pg_${NAME}_$i/p_counter_collection.csv")):
    if "$KSUB" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
print("$NAME", {k: "%.4g" % v for k, v in acc.items()}, flush=True)
PY
done <<GROUPS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_IFETCH
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum
GROUPS
