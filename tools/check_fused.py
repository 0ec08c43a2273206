"""Experiment: the fused a/b -> q walk of the 3-channel guided filter (ASW_GUIDED_FUSED) against the two-pass path and the oracle.
    python tools/check_fused.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aswstereomatch_amd as asw
from aswstereomatch_amd.synth import make_pair
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
modes = sys.argv[1:] or ["1"]
base = asw.Context(0)
ok = True
for mode in modes:
    c = asw.Context(0, env={"ASW_GUIDED_FUSED": mode})
    for (H, W, D) in ((40, 64, 8), (33, 230, 12), (100, 333, 20), (16, 100, 4), (61, 1000, 6), (270, 480, 16)):
        L, R, _ = make_pair(H, W, D, seed=H + W)
        d0, v0 = base.computeAdaptiveWeight_GuidedF_2(L, R, 0, 1e-6, 15, 0, D, return_cost_volume=True)
        d1, v1 = c.computeAdaptiveWeight_GuidedF_2(L, R, 0, 1e-6, 15, 0, D, return_cost_volume=True)
        err = float(np.max(np.abs(v0 - v1) / np.maximum(1e-6, np.abs(v0))))
        same = np.array_equal(v0, v1)
        nd = int((d0 != d1).sum())
        print("mode %s %4dx%-4d D=%-3d rel err %.3g  bit-equal %s  disparity diffs %d" % (mode, W, H, D, err, same, nd), flush=True)
        ok = ok and err < 1e-4
    c.close()
print("OK" if ok else "MISMATCH")
