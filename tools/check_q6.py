"""k_ab6_pair + k_q6_pair (the 6-channel guide: each pass as a pair of wavefronts, one per guide word) against the k_box_walk forms (ASW_AB6_PAIR=0 ASW_Q6_PAIR=0): GuidedF volumes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aswstereomatch_amd as asw
from aswstereomatch_amd.synth import make_pair
old = asw.Context(0, env={"ASW_Q6_PAIR": "0", "ASW_AB6_PAIR": "0"})
new = asw.Context(0)
ok = True
for (H, W, D, dt) in ((40, 64, 8, 0), (33, 230, 12, 0), (100, 333, 20, 1), (16, 100, 4, 0), (61, 1000, 6, 1), (270, 480, 16, 0), (5, 30, 3, 0)):
    L, R, _ = make_pair(H, W, D, seed=H + W)
    d0, v0 = old.computeAdaptiveWeight_GuidedF(L, R, dt, 1e-6, 15, 0, D, return_cost_volume=True)
    d1, v1 = new.computeAdaptiveWeight_GuidedF(L, R, dt, 1e-6, 15, 0, D, return_cost_volume=True)
    err = float(np.max(np.abs(v0 - v1)))
    print("%4dx%-4d D=%-3d type %d  max abs diff %.3g  bit-equal %s  disparity diffs %d" % (W, H, D, dt, err, np.array_equal(v0, v1), int((d0 != d1).sum())), flush=True)
    ok = ok and err < 1e-5
print("OK" if ok else "MISMATCH")
# GuidedF_3 (NCC costs: 0/0 = NaN where a window is flat) through the NaN-safe instantiations: the horizontal sums are associated
# differently from the k_box_walk NaN-safe form (pair sums), so the comparison is by tolerance, with identical NaN patterns
ok3 = True
for (H, W, D, flat) in ((40, 64, 8, False), (33, 230, 12, True), (100, 333, 20, True), (61, 500, 6, True)):
    L, R, _ = make_pair(H, W, D, seed=H + W + 1)
    if flat:
        L[5:25, 10:60] = 77
        R[5:25, 5:70] = 77
    d0, v0 = old.computeAdaptiveWeight_GuidedF_3(L, R, 0, 1e-6, 15, 0, D, return_cost_volume=True)
    d1, v1 = new.computeAdaptiveWeight_GuidedF_3(L, R, 0, 1e-6, 15, 0, D, return_cost_volume=True)
    same_nan = np.array_equal(np.isnan(v0), np.isnan(v1))
    fin = np.isfinite(v0) & np.isfinite(v1)
    err = float(np.max(np.abs(v0[fin] - v1[fin]))) if fin.any() else 0.0
    print("GuidedF_3 %4dx%-4d D=%-3d flat %s  NaN %d  same NaN pattern %s  max abs diff %.3g  disparity diffs %d" % (W, H, D, flat, int(np.isnan(v0).sum()), same_nan, err, int((d0 != d1).sum())), flush=True)
    ok3 = ok3 and same_nan and err < 1e-5
print("OK3" if ok3 else "MISMATCH3")
