"""Kernel times (HIP events of the library) of the classic method at a few shapes: python tools/time_c2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aswstereomatch_amd as asw
from aswstereomatch_amd.synth import make_pair
for (H, W, D, win) in [(1080, 1920, 64, 15), (1080, 1920, 100, 15), (360, 640, 64, 15), (375, 450, 64, 35), (1080, 1920, 128, 15)]:
    L, R, _ = make_pair(H, W, D, seed=1)
    for env in ({}, {"ASW_BILATERAL_XQ": "0"}):
        c = asw.Context(0, env=env)
        c.upload_pair(0, L, R)
        best = 1e9
        for i in range(4):
            c.match_resident(0, 0, 2, win, 0, D, keep_volume=True)
            best = min(best, c.timing()["aggregate_ms"])
        print("%dx%d D=%d win=%d %-22s aggregate %.3f ms  launches %d" % (W, H, D, win, env or "default", best, c.timing()["aggregate_launches"]), flush=True)
        c.close()
# the 35x35 stress point of SURVEY 8d (1080p, D=128)
L, R, _ = make_pair(1080, 1920, 128, seed=1)
c = asw.Context(0)
c.upload_pair(0, L, R)
best = 1e9
for i in range(2):
    c.match_resident(0, 0, 2, 35, 0, 128, keep_volume=True)
    best = min(best, c.timing()["aggregate_ms"])
print("1920x1080 D=128 win=35 aggregate %.3f ms" % best, flush=True)
c.close()
