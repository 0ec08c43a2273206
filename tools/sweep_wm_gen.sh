#!/bin/bash
# Weighted median, general tile form (k_wmedian_tile_gen.hip), KITTI shape: kernel times per window and per number of block rows a
# workgroup holds (ASW_WMEDIAN_GEN_ROWS).   gpurun -- 'bash tools/sweep_wm_gen.sh <tag> "<win:rows> ..."'
set -e -o pipefail
TAG=${1:-wmgen}
CASES=${2:-"21:0 21:1 21:4 35:0 35:2"}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in $CASES; do
  win=${c%%:*}; rows=${c##*:}
  export ASW_WMEDIAN_GEN_ROWS=$rows
  rocprofv3 --kernel-trace --stats -d /tmp/wg_$win_$rows -o p --output-format csv -- python3 "$ROOT/tools/run_one.py" --alg 10 --width 1242 --height 375 --disp 192 --win $win --reps 2 > "$OUT/log_${win}_$rows.txt" 2>&1
  cp /tmp/wg_$win_$rows/p_kernel_stats.csv "$OUT/kernel_stats_win${win}_rows$rows.csv"
  python3 - <<PY
import csv
t={}
for r in csv.DictReader(open("$OUT/kernel_stats_win${win}_rows$rows.csv")):
    n=r["Name"]
    k="pick" if "k_wmg_pick" in n else "sort" if "k_wmg_sort" in n else "weights" if "k_wm_weights" in n else None
    if k: t[k]=t.get(k,0)+float(r["TotalDurationNs"])/2e6
print("win $win rows $rows  per frame (ms):", {k: round(v,2) for k,v in t.items()}, flush=True)
PY
done
