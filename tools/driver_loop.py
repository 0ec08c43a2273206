"""The reference driver's own workload (aswStereoMatch.cpp:30-98): a full-resolution rectified pair is resized to 640x360,
detail-boosted, matched with the enabled method (GuidedF_2, winSize 15, minDisparity 0, numDisparity 64) and written as an
8-bit normalised map.  Times the per-frame latency of that loop through the resident API, stage by stage.

    python tools/driver_loop.py [--frames 50] [--alg 8]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aswstereomatch_amd as asw  # noqa: E402
from aswstereomatch_amd.synth import make_pair  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=50)
ap.add_argument("--alg", type=int, default=8)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
pairs = [make_pair(a.height, a.width, 128, seed=70 + i)[:2] for i in range(2)]
ctx = asw.Context(0)
t_pre = t_match = t_down = 0.0
for i in range(a.frames + 3):
    L, R = pairs[i % 2]
    t0 = time.perf_counter()
    ctx.preprocess_pair(0, L, R, (640, 360), detail_boost=True)              # main.cpp:30-31, 67-89
    t1 = time.perf_counter()
    ctx.match_resident(0, asw.DISPARITY_LEFT, a.alg, 15, 0, 64)               # main.cpp:94
    t2 = time.perf_counter()
    d8 = ctx.download_disparity_u8(0, (360, 640), normalize=True)            # main.cpp:97-98
    t3 = time.perf_counter()
    if i >= 3:
        t_pre += t1 - t0
        t_match += t2 - t1
        t_down += t3 - t2
n = a.frames
print("alg %d, %dx%d -> 640x360 D=64: %.3f ms/frame (upload+preprocess %.3f, match %.3f [kernels %.3f], u8 download %.3f), %.1f frames/s"
      % (a.alg, a.width, a.height, (t_pre + t_match + t_down) / n * 1e3, t_pre / n * 1e3, t_match / n * 1e3, ctx.timing()["total_ms"],
         t_down / n * 1e3, n / (t_pre + t_match + t_down)))
print("disparity bytes: min %d max %d" % (d8.min(), d8.max()))
ctx.close()
