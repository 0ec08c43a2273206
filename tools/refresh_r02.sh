#!/bin/bash
# Round-2 measurements quoted in DESIGN.md:  /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/refresh_r02.sh r02'
# bench lines, rocprofv3 kernel stats (separate runs) and counter passes (--pmc never combined with other trace domains).
set -e -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
KITTI="--width 1242 --height 375 --disp 192"
bench() { local name=$1; shift; python3 "$ROOT/bench.py" "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "bench $name: $(cut -c1-150 "$OUT/bench_$name.json")"; }
stats() { local name=$1; shift; rocprofv3 --kernel-trace --stats -d "/tmp/st_$name" -o p --output-format csv -- python3 "$@" > "$OUT/prof_$name.log" 2>&1; cp "/tmp/st_$name/p_kernel_stats.csv" "$OUT/${name}_kernel_stats.csv"; echo "stats $name done"; }
pmc() { local name=$1; local ctr=$2; shift 2; rocprofv3 --kernel-trace --pmc $ctr -d "/tmp/pm_$name" -o p --output-format csv -- python3 "$@" > "$OUT/pmc_$name.log" 2>&1; cp "/tmp/pm_$name/p_counter_collection.csv" "$OUT/pmc_$name.csv"; echo "pmc $name done"; }
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES"
bench default
ASW_BILATERAL_XQ=0 bench default_one_kernel --no-cpu --batch-frames 0
bench geodesic --workload geodesic $KITTI --frames 4 --steps 3
ASW_GEODESIC_XQ=0 bench geodesic_one_kernel --workload geodesic $KITTI --frames 4 --steps 3 --no-cpu --batch-frames 0
bench guided2 --workload guided2 --frames 4 --steps 3
bench guided --workload guided --frames 4 --steps 3
bench wmedian --workload wmedian $KITTI --frames 4 --steps 3
ASW_WMEDIAN_TILE=0 bench wmedian_per_pixel_sort --workload wmedian $KITTI --frames 4 --steps 3 --no-cpu --batch-frames 0
stats bilateral_bench "$ROOT/bench.py" --no-cpu --batch-frames 0
stats alg2 "$ROOT/tools/run_one.py" --alg 2 --reps 5
stats alg4 "$ROOT/tools/run_one.py" --alg 4 $KITTI --reps 4
stats alg8 "$ROOT/tools/run_one.py" --alg 8 --reps 3
stats alg7 "$ROOT/tools/run_one.py" --alg 7 --reps 3
stats alg10 "$ROOT/tools/run_one.py" --alg 10 $KITTI --reps 3
ASW_WMEDIAN_TILE=0 stats alg10_per_pixel_sort "$ROOT/tools/run_one.py" --alg 10 $KITTI --reps 2
pmc bilateral_FETCH_SIZE "FETCH_SIZE" "$ROOT/tools/run_one.py" --alg 2 --reps 2
pmc bilateral_WRITE_SIZE "WRITE_SIZE" "$ROOT/tools/run_one.py" --alg 2 --reps 2
pmc bilateral_SQ "$SQ" "$ROOT/tools/run_one.py" --alg 2 --reps 2
LDSQ="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"
pmc wmedian_LDS "$LDSQ" "$ROOT/tools/run_one.py" --alg 10 $KITTI --reps 1
pmc guided2_SQ "$LDSQ" "$ROOT/tools/run_one.py" --alg 8 --reps 1
pmc bilateral_LDS "$LDSQ" "$ROOT/tools/run_one.py" --alg 2 --reps 2
pmc wmedian_SQ "$SQ" "$ROOT/tools/run_one.py" --alg 10 $KITTI --reps 2
pmc geodesic_SQ "$SQ" "$ROOT/tools/run_one.py" --alg 4 $KITTI --reps 2
pmc geodesic_FETCH_SIZE "FETCH_SIZE" "$ROOT/tools/run_one.py" --alg 4 $KITTI --reps 2
pmc geodesic_WRITE_SIZE "WRITE_SIZE" "$ROOT/tools/run_one.py" --alg 4 $KITTI --reps 2
echo "all done"
