"""GuidedF_2 at 15x15: the fused walk (ASW_GUIDED_FUSED=1) against the two-pass path over several frame shapes (library events, best of 5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aswstereomatch_amd as asw
from aswstereomatch_amd.synth import make_pair
shapes = [(1080, 1920, 128), (720, 1280, 96), (360, 640, 64), (375, 1242, 192), (288, 384, 16), (2160, 3840, 64), (1080, 1920, 32)]
for (H, W, D) in shapes:
    L, R, _ = make_pair(H, W, D, seed=3)
    res = []
    for env in ({"ASW_GUIDED_FUSED": "0"}, {"ASW_GUIDED_FUSED": "1"}, {}):
        c = asw.Context(0, env=env)
        c.upload_pair(0, L, R)
        best = 1e9
        for i in range(5):
            c.match_resident(0, 0, 8, 15, 0, D, keep_volume=True)
            best = min(best, c.timing()["aggregate_ms"])
        v = c.download_volume(0, (D, H, W))
        res.append((best, v))
        c.close()
    print("%4dx%-4d D=%-3d  two-pass %7.3f ms   fused %7.3f ms   ratio %.3f   default %7.3f ms   bit-equal %s" % (W, H, D, res[0][0], res[1][0], res[1][0] / res[0][0], res[2][0], np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][1], res[2][1])), flush=True)
