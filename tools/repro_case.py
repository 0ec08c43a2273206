"""Re-run one guided3 fuzz case (diagnostic)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aswstereomatch_amd as asw
from aswstereomatch_amd.synth import make_pair
from oracle import asw_oracle as O
H, W, win, minD, numD, dt, seed = 1, 247, 3, 1, 42, 1, 181930660
for block in (4, 8, 16):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=block)
    rc, dw, vw = O.asw_guided3(L, R, dt, 1e-6, win, minD, numD, want_vol=True)
    for trial in range(2):
        ctx = asw.Context(0)
        d, v = ctx.computeAdaptiveWeight_GuidedF_3(L, R, dt, 1e-6, win, minD, numD, return_cost_volume=True)
        d2, v2 = ctx.computeAdaptiveWeight_GuidedF_3(L, R, dt, 1e-6, win, minD, numD, return_cost_volume=True)
        ctx.close()
        fin = np.isfinite(vw)
        nanok = np.array_equal(np.isnan(v), np.isnan(vw))
        err = np.abs(v[fin] - vw[fin]) / np.maximum(np.abs(vw[fin]), 1e-30) if fin.any() else np.zeros(1)
        allfin = np.isfinite(vw).all(axis=0)
        print("block", block, "trial", trial, "nan pattern equal", nanok, "max rel err", err.max() if err.size else 0,
              "finite frac", fin.mean(), "disp equal where finite", np.array_equal(d[allfin], dw[allfin]),
              "second call identical", np.array_equal(v, v2, equal_nan=True))
        bad = np.argwhere(~np.isclose(v, vw, rtol=1e-4, atol=1e-30, equal_nan=True))
        if len(bad):
            print("   first bad", bad[:5].tolist(), [ (float(v[tuple(b)]), float(vw[tuple(b)])) for b in bad[:5]])
