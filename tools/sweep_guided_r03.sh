#!/bin/bash
# Round 3: variants of the two guided-filter passes at 1080p D=128 (GuidedF_2): register ring / pipeline depth (ASW_RING_AB,
# ASW_RING_Q: launch_ab_q3 in k_guided.hip), rows per band, workgroup shape of the q pass.  The two passes are independent, so
# every run varies both.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/sweep_guided_r03.sh <tag> [alg]'
set -e -o pipefail
TAG=${1:-gsweep}
ALG=${2:-8}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name ring_ab band_ab ring_q band_q wg_strips
  local name=$1
  export ASW_RING_AB=$2 ASW_BAND_AB=$3 ASW_RING_Q=$4 ASW_BAND_Q=$5 ASW_Q_WG_STRIPS=$6
  rocprofv3 --kernel-trace --stats -d /tmp/gs_$name -o p --output-format csv -- python3 "$ROOT/tools/run_one.py" --alg $ALG --reps 3 > "$OUT/log_$name.txt" 2>&1
  cp /tmp/gs_$name/p_kernel_stats.csv "$OUT/kernel_stats_$name.csv"
  python3 - <<PY
import csv,re
t={}
for r in csv.DictReader(open("/tmp/gs_$name/p_kernel_stats.csv")):
    n=r["Name"]
    k="ab" if "ABSrc" in n else "q" if "QSrc" in n else "stats" if "StatsSrc" in n else (re.search(r"(k_\w+)",n).group(1) if re.search(r"(k_\w+)",n) else n[:20])
    t[k]=t.get(k,0)+float(r["TotalDurationNs"])/int(r["Calls"])
print("$name ring_ab=$2 band_ab=$3 ring_q=$4 band_q=$5 wg_strips=$6", {k: round(v/1e6,3) for k,v in t.items() if v>1e5}, flush=True)
PY
}
while read -r line; do [ -n "$line" ] && run $line; done <<CASES
$(cat "$ROOT/tools/sweep_guided_cases.txt")
CASES
echo "sweep done"
