"""Weighted median at windows above 15x15, KITTI shape (1242x375 D=192): tile form (k_wmedian_tile_gen.hip) against the per-pixel
sort (ASW_WMEDIAN_TILE=0), library events; the two volumes are compared bit for bit.
    python tools/time_wm_gen.py [win ...]      (default 21 35)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aswstereomatch_amd as asw
from aswstereomatch_amd.synth import make_pair
H, W, D = 375, 1242, 192
wins = [int(a) for a in sys.argv[1:]] or [21, 35]
L, R, _ = make_pair(H, W, D, seed=1)
for win in wins:
    res = {}
    for name, env in (("tile", None), ("sort", {"ASW_WMEDIAN_TILE": "0"})):
        c = asw.Context(0, env=env)
        c.upload_pair(0, L, R)
        best = 1e9
        for i in range(2 if name == "tile" else 1):
            c.match_resident(0, 0, 10, win, 0, D, keep_volume=True)
            best = min(best, c.timing()["aggregate_ms"])
        res[name] = (best, c.download_volume(0, (D, H, W)))
        c.close()
        print("win %2d %s %10.2f ms" % (win, name, best), flush=True)
    print("win %2d volumes equal: %s   speed-up %.1fx" % (win, np.array_equal(res["tile"][1], res["sort"][1]), res["sort"][0] / res["tile"][0]), flush=True)
