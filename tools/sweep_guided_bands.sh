#!/bin/bash
# Band-height sweep of the guided-filter passes at 1080p D=128 (GuidedF_2): the a/b pass and the q pass separately.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/sweep_guided_bands.sh <tag>'
set -e -o pipefail
TAG=${1:-gbands}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for ab in 32 64; do
  for q in 16 32 64 128 270; do
    export ASW_BAND_AB=$ab ASW_BAND_Q=$q
    rocprofv3 --kernel-trace --stats -d /tmp/gb_${ab}_$q -o p --output-format csv -- python3 "$ROOT/tools/run_one.py" --alg 8 --reps 4 > "$OUT/log_${ab}_$q.txt" 2>&1
    cp /tmp/gb_${ab}_$q/p_kernel_stats.csv "$OUT/kernel_stats_ab${ab}_q$q.csv"
    python3 - <<PY
import csv,re
t={}
for r in csv.DictReader(open("/tmp/gb_${ab}_$q/p_kernel_stats.csv")):
    n=r["Name"]
    k="ab" if "ABSrc" in n else "q" if "QSrc" in n else "stats" if "StatsSrc" in n else (re.search(r"(k_\w+)",n).group(1) if re.search(r"(k_\w+)",n) else n[:20])
    t[k]=t.get(k,0)+float(r["TotalDurationNs"])/int(r["Calls"])*1  # avg per call
print("ab=$ab q=$q", {k: round(v/1e6,3) for k,v in t.items() if v>2e4})
PY
  done
done
