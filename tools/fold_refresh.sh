#!/bin/bash
# Copy one tools/refresh_r03.sh run (gpurun_out/<tag>/) into profiles/r03 and rebuild the traffic records from its counter passes:
#   bash tools/fold_refresh.sh r03e
set -e
TAG=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/gpurun_out/$TAG
DST=$ROOT/profiles/r03
cp "$SRC"/bench_{default,guided2,guided,geodesic,wmedian}.json "$DST"/
cp "$SRC"/{bilateral_bench,alg8,alg7,alg4,alg10}_kernel_stats.csv "$DST"/
cp "$SRC"/pmc_{guided2_FETCH_SIZE,guided2_WRITE_SIZE,guided2_L2,guided2_SQ,bilateral_FETCH_SIZE,bilateral_WRITE_SIZE,geodesic_FETCH_SIZE,geodesic_WRITE_SIZE}.csv "$DST"/
cd "$ROOT"
python3 tools/pmc_traffic.py $DST/pmc_guided2_FETCH_SIZE.csv $DST/pmc_guided2_WRITE_SIZE.csv 2 profiles/pmc_guided2.json \
  "1920x1080 D=128 win=15 guided2 (GuidedF_2): statistics + a/b + q + WTA launches" 8 1920 1080 128 15 "box_walk|k_wta|pair3" > /dev/null
python3 tools/pmc_traffic.py $DST/pmc_bilateral_FETCH_SIZE.csv $DST/pmc_bilateral_WRITE_SIZE.csv 2 profiles/pmc_bilateral.json \
  "1920x1080 D=128 classic bilateral (xq + border tiles + tail + merge)" 2 1920 1080 128 15 > /dev/null
python3 tools/pmc_traffic.py $DST/pmc_geodesic_FETCH_SIZE.csv $DST/pmc_geodesic_WRITE_SIZE.csv 2 profiles/pmc_geodesic.json \
  "1242x375 D=192 geodesic (weights + xq passes + remainder + merge)" 4 1242 375 192 15 > /dev/null
for w in guided2 bilateral geodesic; do python3 -c "
import json; d=json.load(open('profiles/pmc_$w.json')); print('$w', d['hbm_bytes_per_frame'], d['ratio_to_algorithmic'])"; done
