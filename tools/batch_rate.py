"""PCIe-inclusive frame rate of the batch scheduler (asw_stereo_match_batch) vs sequential asw_stereo_match calls."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import aswstereomatch_amd as asw
from aswstereomatch_amd.synth import make_pair

ap = argparse.ArgumentParser()
ap.add_argument("--alg", type=int, default=8)
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--disp", type=int, default=128)
ap.add_argument("--devs", default="0", help="device list of the batch, e.g. 0,0: two scheduler threads (contexts, streams) on GPU 0")
a = ap.parse_args()
devs = [int(v) for v in a.devs.split(",")]
pairs = [make_pair(a.height, a.width, a.disp, seed=50 + i)[:2] for i in range(2)]
Ls = [pairs[i % 2][0] for i in range(a.frames)]
Rs = [pairs[i % 2][1] for i in range(a.frames)]
ctx = asw.Context(0)
ctx.stereoMatching(Ls[0], Rs[0], 0, a.alg, 15, 0, a.disp)
t = time.perf_counter()
seq = [ctx.stereoMatching(Ls[i], Rs[i], 0, a.alg, 15, 0, a.disp) for i in range(a.frames)]
t_seq = time.perf_counter() - t
asw.stereoMatchingBatch(Ls[:2 * len(devs)], Rs[:2 * len(devs)], 0, a.alg, 15, 0, a.disp, device_ids=devs)
t = time.perf_counter()
outs = [np.zeros((a.height, a.width), np.float32) + 1 for _ in range(a.frames)]
bat = asw.stereoMatchingBatch(Ls, Rs, 0, a.alg, 15, 0, a.disp, device_ids=devs, out=outs)
t_bat = time.perf_counter() - t
ok = all(np.array_equal(x, y) for x, y in zip(seq, bat))
mp = a.width * a.height * a.frames / 1e6
print("devs %s " % a.devs, end="")
print("alg %d: sequential %.2f ms/frame (%.1f Mpix/s), pipelined batch %.2f ms/frame (%.1f Mpix/s), identical=%s"
      % (a.alg, t_seq / a.frames * 1e3, mp / t_seq, t_bat / a.frames * 1e3, mp / t_bat, ok))
ctx.close()
if os.environ.get("ASW_BATCH_SWEEP"):  # fixed vs per-frame cost of one batch call
    for n in (1, 2, 4, 8, 16, 32, 32):
        Ln = [pairs[i % 2][0] for i in range(n)]
        Rn = [pairs[i % 2][1] for i in range(n)]
        t = time.perf_counter()
        asw.stereoMatchingBatch(Ln, Rn, 0, a.alg, 15, 0, a.disp, device_ids=[0])
        dt = time.perf_counter() - t
        print("batch of %2d: %.2f ms total, %.2f ms/frame" % (n, dt * 1e3, dt * 1e3 / n))
