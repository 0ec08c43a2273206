"""Generates tests/golden/asw_golden_v1.npz: seeded inputs + outputs of the CPU oracle (oracle/asw_oracle.c).

The reference (ZhangYY12345/aswStereoMatch) ships no fixtures and cannot be built here (it needs OpenCV
4.1.0), so these vectors pin the ORACLE, not the reference: they are regression fixtures that make any
later change of the restatement (or of the HIP kernels checked against it) visible.  "Parity unpinned"
at the OpenCV boundary remains (see oracle/asw_oracle.c header).

    python tests/golden/make_golden.py      # rewrites the .npz (plain arrays, no pickle)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aswstereomatch_amd.synth import make_pair  # noqa: E402
from oracle import asw_oracle as O  # noqa: E402

H, W, D, WIN, SEED = 32, 48, 8, 5, 20261004


def main():
    O.set_threads(4)
    L, R, gt = make_pair(H, W, D, seed=SEED, block=12)
    out = {"L": L, "R": R, "gt": gt.astype(np.int32), "params": np.array([H, W, D, WIN, SEED], np.int64)}
    out["gray_L"] = O.bgr2gray(L)
    out["ad"] = O.compute_ad(L, R, 0, 0, D)[1]
    out["tad"] = O.compute_tad(L, R, 0, 30, 0, D)[1]
    out["sd"] = O.compute_sd(L, R, 0, 0, D)[1]
    out["similarity"] = O.compute_similarity(L, R, 0.4, 10, 50, 0, 0, D)[1]
    out["sad"] = O.cost_sad(L, R, 0, WIN, 0, D)[1]
    out["geodesic_dist_L"] = O.geodesic_dist(L, WIN, 3)[1]
    out["prep_L"] = O.preprocess(L, (40, 26), True)            # row f3: resize + HSV-V bilateral detail boost
    out["prep_R_noboost"] = O.preprocess(R, (24, 16), False)
    out["ncc_raw"] = O.cost_ncc(L, R, 0, WIN, 0, D, raw=True)[1]
    out["ncc"] = O.cost_ncc(L, R, 0, WIN, 0, D)[1]
    out["ncc_disp"] = O.ncc_disparity(L, R, 0, WIN, 0, D)[1]
    for name, fn in [("classic", lambda: O.asw_classic(L, R, 30, 20, 0, WIN, 0, D, want_vol=True)),
                     ("geodesic", lambda: O.asw_geodesic(L, R, 0, WIN, 0, D, want_vol=True)),
                     ("guided", lambda: O.asw_guided(L, R, 0, 1e-6, WIN, 0, D, want_vol=True)),
                     ("guided2", lambda: O.asw_guided2(L, R, 0, 1e-6, WIN, 0, D, want_vol=True)),
                     ("wmedian", lambda: O.asw_wmedian(L, R, 0, WIN, 10, 10, 0, D, want_vol=True)),
                     ("blo1", lambda: O.asw_blo1(L, R, 0, 0.015, WIN, 0, D, want_vol=True)),
                     ("bilgrid", lambda: O.asw_bilgrid((L // 64) * 64, (R // 64) * 64, 0, 6, 64, 0, D, want_vol=True)),
                     ("direct8", lambda: O.asw_direct8(L, R, 0, WIN, 0, D, want_vol=True)),
                     ("guided3", lambda: O.asw_guided3(L, R, 0, 1e-6, WIN, 0, D, want_vol=True)),
                     ("guided3_right", lambda: O.asw_guided3(L, R, 1, 1e-6, WIN, 0, D, want_vol=True)),
                     ("classic_right", lambda: O.asw_classic(L, R, 30, 20, 1, WIN, 0, D, want_vol=True)),
                     ("geodesic_right", lambda: O.asw_geodesic(L, R, 1, WIN, 0, D, want_vol=True)),
                     ("guided_right", lambda: O.asw_guided(L, R, 1, 1e-6, WIN, 0, D, want_vol=True))]:
        rc, disp, vol = fn()
        assert rc == 0
        out[name + "_disp"] = disp
        out[name + "_vol"] = vol
    out["classic_disp_u8"] = O.disparity_to_u8(out["classic_disp"], True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "asw_golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
