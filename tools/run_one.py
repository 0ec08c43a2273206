"""Run one method a few times on one resident synthetic frame (profiling driver: no torch, fast start)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aswstereomatch_amd as asw
from aswstereomatch_amd.synth import make_pair

ap = argparse.ArgumentParser()
ap.add_argument("--alg", type=int, default=2)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--disp", type=int, default=128)
ap.add_argument("--win", type=int, default=15)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--keep-volume", type=int, default=1)
ap.add_argument("--disparity-type", type=int, default=0, help="0 = DISPARITY_LEFT, 1 = DISPARITY_RIGHT")
a = ap.parse_args()
L, R, _ = make_pair(a.height, a.width, a.disp, seed=1234)
ctx = asw.Context(0)
ctx.upload_pair(0, L, R)
for i in range(a.reps):
    ctx.match_resident(0, a.disparity_type, a.alg, a.win, 0, a.disp, keep_volume=bool(a.keep_volume))
    print(i, ctx.timing(), flush=True)
ctx.close()
