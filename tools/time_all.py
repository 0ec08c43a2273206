"""Aggregation / total kernel time (library events) of every method on one resident 1080p D=128 frame: python tools/time_all.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aswstereomatch_amd as asw
from aswstereomatch_amd.synth import make_pair
H, W, D = 1080, 1920, 128
L, R, _ = make_pair(H, W, D, seed=1)
c = asw.Context(0)
c.upload_pair(0, L, R)
for alg, name in ((2, "classic"), (3, "direct8"), (4, "geodesic"), (5, "bilgrid"), (6, "blo1"), (7, "guided"), (8, "guided2"), (9, "guided3"), (10, "wmedian"), (11, "ncc")):
    best = (1e9, 1e9)
    for i in range(3):
        c.match_resident(0, 0, alg, 15, 0, D, keep_volume=True)
        t = c.timing()
        best = min(best, (t["total_ms"], t["aggregate_ms"]))
    print("%-9s total %8.3f ms  aggregate %8.3f ms" % (name, best[0], best[1]), flush=True)
c.close()
