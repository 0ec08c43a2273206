#!/bin/bash
# Re-measures everything DESIGN.md quotes, on the GPU box:
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r01'
# Writes gpurun_out/<tag>/: one bench line per workload, rocprofv3 kernel stats (bench itself for the headline
# workload, tools/run_one.py for the others) and the FETCH_SIZE / WRITE_SIZE / SQ counter passes (separate passes, never
# combined with other trace domains).  Copy the files you want judged into profiles/<tag>/ afterwards.
set -e -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

bench() {  # name, extra args...
    local name=$1; shift
    python3 "$ROOT/bench.py" "$@" > "$OUT/final_bench_$name.json" 2> "$OUT/final_bench_$name.err"
    echo "bench $name: $(cut -c1-160 "$OUT/final_bench_$name.json")"
}
stats() {  # name, program args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats -d "/tmp/prof_$name" -o p --output-format csv -- python3 "$@" > "$OUT/prof_$name.log" 2>&1
    cp "/tmp/prof_$name/p_kernel_stats.csv" "$OUT/final_${name}_kernel_stats.csv"
    echo "stats $name done"
}
pmc() {  # name, counter list (quoted), program args...
    local name=$1; local ctr=$2; shift 2
    rocprofv3 --kernel-trace --pmc $ctr -d "/tmp/pmc_$name" -o p --output-format csv -- python3 "$@" > "$OUT/pmc_$name.log" 2>&1
    cp "/tmp/pmc_$name/p_counter_collection.csv" "$OUT/pmc_$name.csv"
    echo "pmc $name done"
}

KITTI="--width 1242 --height 375 --disp 192"
bench default
bench guided2 --workload guided2 --frames 4 --steps 3
bench guided --workload guided --frames 4 --steps 3
bench geodesic --workload geodesic $KITTI --frames 4 --steps 3
bench wmedian --workload wmedian $KITTI --frames 4 --steps 3
bench blo1 --workload blo1 --frames 2 --steps 2
bench direct8 --workload direct8 --frames 8 --steps 3
bench guided3 --workload guided3 --frames 2 --steps 2
bench ncc --workload ncc --frames 4 --steps 3
bench bilgrid --workload bilgrid --frames 2 --steps 2

# the rocprofv3 summary of the SAME command as the headline bench line (CPU baseline skipped: it is host-only work)
stats bilateral_bench "$ROOT/bench.py" --no-cpu
stats alg8 "$ROOT/tools/run_one.py" --alg 8 --reps 3
stats alg7 "$ROOT/tools/run_one.py" --alg 7 --reps 3
stats alg4 "$ROOT/tools/run_one.py" --alg 4 $KITTI --reps 3
stats alg10 "$ROOT/tools/run_one.py" --alg 10 $KITTI --reps 3
stats alg6 "$ROOT/tools/run_one.py" --alg 6 --reps 2
stats alg3 "$ROOT/tools/run_one.py" --alg 3 --reps 3
stats alg9 "$ROOT/tools/run_one.py" --alg 9 --reps 2
stats alg11 "$ROOT/tools/run_one.py" --alg 11 --reps 3
stats alg5 "$ROOT/tools/run_one.py" --alg 5 --reps 2

pmc bilateral_FETCH_SIZE "FETCH_SIZE" "$ROOT/tools/run_one.py" --alg 2 --reps 2
pmc bilateral_WRITE_SIZE "WRITE_SIZE" "$ROOT/tools/run_one.py" --alg 2 --reps 2
pmc bilateral_SQ "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" "$ROOT/tools/run_one.py" --alg 2 --reps 2
pmc guided2_FETCH_SIZE "FETCH_SIZE" "$ROOT/tools/run_one.py" --alg 8 --reps 2
pmc guided2_WRITE_SIZE "WRITE_SIZE" "$ROOT/tools/run_one.py" --alg 8 --reps 2
pmc guided2_L2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "$ROOT/tools/run_one.py" --alg 8 --reps 2
pmc guided2_SQ "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" "$ROOT/tools/run_one.py" --alg 8 --reps 2
echo "all done"
