#!/bin/bash
# Round-3 measurements quoted in DESIGN.md:  /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/refresh_r03.sh r03 [parts]'
# parts: any of  bench stats pmc  (default: all).  Counter passes (--pmc) are never combined with other trace domains.
set -e -o pipefail
TAG=${1:-r03}
PARTS=${2:-"bench stats pmc"}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
hostname > "$OUT/box.txt" 2>/dev/null || true
KITTI="--width 1242 --height 375 --disp 192"
bench() { local name=$1; shift; python3 "$ROOT/bench.py" "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "bench $name: $(cut -c1-150 "$OUT/bench_$name.json")"; }
stats() { local name=$1; shift; rocprofv3 --kernel-trace --stats -d "/tmp/st_$name" -o p --output-format csv -- python3 "$@" > "$OUT/prof_$name.log" 2>&1; cp "/tmp/st_$name/p_kernel_stats.csv" "$OUT/${name}_kernel_stats.csv"; echo "stats $name done"; }
pmc() { local name=$1; local ctr=$2; shift 2; rocprofv3 --kernel-trace --pmc $ctr -d "/tmp/pm_$name" -o p --output-format csv -- python3 "$@" > "$OUT/pmc_$name.log" 2>&1; cp "/tmp/pm_$name/p_counter_collection.csv" "$OUT/pmc_$name.csv"; echo "pmc $name done"; }
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
for part in $PARTS; do
case $part in
bench)
  bench default
  bench guided2 --workload guided2 --frames 4 --steps 3
  bench guided --workload guided --frames 4 --steps 3
  bench geodesic --workload geodesic $KITTI --frames 4 --steps 3
  bench wmedian --workload wmedian $KITTI --frames 4 --steps 3
  ;;
stats)
  stats bilateral_bench "$ROOT/bench.py" --no-cpu --batch-frames 0
  stats alg8 "$ROOT/tools/run_one.py" --alg 8 --reps 3
  stats alg7 "$ROOT/tools/run_one.py" --alg 7 --reps 3
  stats alg4 "$ROOT/tools/run_one.py" --alg 4 $KITTI --reps 4
  stats alg10 "$ROOT/tools/run_one.py" --alg 10 $KITTI --reps 3
  ;;
pmc)
  for a in "guided2 8" "bilateral 2"; do
    set -- $a
    pmc $1_FETCH_SIZE "FETCH_SIZE" "$ROOT/tools/run_one.py" --alg $2 --reps 2
    pmc $1_WRITE_SIZE "WRITE_SIZE" "$ROOT/tools/run_one.py" --alg $2 --reps 2
  done
  pmc guided2_L2 "TCC_HIT_sum TCC_MISS_sum" "$ROOT/tools/run_one.py" --alg 8 --reps 2
  pmc guided2_SQ "$SQ" "$ROOT/tools/run_one.py" --alg 8 --reps 1
  pmc geodesic_FETCH_SIZE "FETCH_SIZE" "$ROOT/tools/run_one.py" --alg 4 $KITTI --reps 2
  pmc geodesic_WRITE_SIZE "WRITE_SIZE" "$ROOT/tools/run_one.py" --alg 4 $KITTI --reps 2
  ;;
esac
done
echo "all done"
