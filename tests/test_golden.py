"""Golden fixtures (tests/golden/asw_golden_v1.npz, made by tests/golden/make_golden.py from the oracle).
CPU: the oracle still reproduces them bit for bit.  GPU: the HIP path reproduces them (C-ABI calls)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "asw_golden_v1.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD, allow_pickle=False)


def test_oracle_reproduces_golden(oracle, gold):
    L, R = gold["L"], gold["R"]
    H, W, D, WIN, _ = [int(v) for v in gold["params"]]
    assert np.array_equal(oracle.bgr2gray(L), gold["gray_L"])
    assert np.array_equal(oracle.compute_ad(L, R, 0, 0, D)[1], gold["ad"])
    assert np.array_equal(oracle.compute_tad(L, R, 0, 30, 0, D)[1], gold["tad"])
    assert np.array_equal(oracle.compute_sd(L, R, 0, 0, D)[1], gold["sd"])
    assert np.array_equal(oracle.compute_similarity(L, R, 0.4, 10, 50, 0, 0, D)[1], gold["similarity"])
    assert np.array_equal(oracle.cost_sad(L, R, 0, WIN, 0, D)[1], gold["sad"])
    assert np.array_equal(oracle.geodesic_dist(L, WIN, 3)[1], gold["geodesic_dist_L"])
    assert np.array_equal(oracle.preprocess(L, (40, 26), True), gold["prep_L"])
    assert np.array_equal(oracle.preprocess(R, (24, 16), False), gold["prep_R_noboost"])
    assert np.array_equal(oracle.disparity_to_u8(gold["classic_disp"], True), gold["classic_disp_u8"])
    assert np.array_equal(oracle.cost_ncc(L, R, 0, WIN, 0, D, raw=True)[1], gold["ncc_raw"])
    assert np.array_equal(oracle.cost_ncc(L, R, 0, WIN, 0, D)[1], gold["ncc"])
    assert np.array_equal(oracle.ncc_disparity(L, R, 0, WIN, 0, D)[1], gold["ncc_disp"])
    for name, fn in [("classic", lambda: oracle.asw_classic(L, R, 30, 20, 0, WIN, 0, D, want_vol=True)),
                     ("direct8", lambda: oracle.asw_direct8(L, R, 0, WIN, 0, D, want_vol=True)),
                     ("bilgrid", lambda: oracle.asw_bilgrid((L // 64) * 64, (R // 64) * 64, 0, 6, 64, 0, D, want_vol=True)),
                     ("guided3", lambda: oracle.asw_guided3(L, R, 0, 1e-6, WIN, 0, D, want_vol=True)),
                     ("guided3_right", lambda: oracle.asw_guided3(L, R, 1, 1e-6, WIN, 0, D, want_vol=True)),
                     ("geodesic", lambda: oracle.asw_geodesic(L, R, 0, WIN, 0, D, want_vol=True)),
                     ("guided", lambda: oracle.asw_guided(L, R, 0, 1e-6, WIN, 0, D, want_vol=True)),
                     ("guided2", lambda: oracle.asw_guided2(L, R, 0, 1e-6, WIN, 0, D, want_vol=True)),
                     ("wmedian", lambda: oracle.asw_wmedian(L, R, 0, WIN, 10, 10, 0, D, want_vol=True)),
                     ("blo1", lambda: oracle.asw_blo1(L, R, 0, 0.015, WIN, 0, D, want_vol=True)),
                     ("classic_right", lambda: oracle.asw_classic(L, R, 30, 20, 1, WIN, 0, D, want_vol=True)),
                     ("geodesic_right", lambda: oracle.asw_geodesic(L, R, 1, WIN, 0, D, want_vol=True)),
                     ("guided_right", lambda: oracle.asw_guided(L, R, 1, 1e-6, WIN, 0, D, want_vol=True))]:
        rc, disp, vol = fn()
        assert rc == 0 and np.array_equal(disp, gold[name + "_disp"]), name
        assert np.array_equal(vol, gold[name + "_vol"], equal_nan=True), name


def test_golden_is_meaningful(gold):
    # the synthetic pair has a recoverable ground truth: every method gets most interior pixels right
    gt = gold["gt"]
    for name in ("classic", "geodesic", "guided2", "wmedian"):
        ok = (gold[name + "_disp"][6:-6, 14:-6] == gt[6:-6, 14:-6]).mean()
        assert ok > 0.5, (name, ok)


@pytest.mark.gpu
def test_hip_reproduces_golden(gold):
    import aswstereomatch_amd as asw

    ctx = asw.Context(0)
    L, R = gold["L"], gold["R"]
    H, W, D, WIN, _ = [int(v) for v in gold["params"]]
    A, LEFT = asw.StereoMatchingAlgorithms, asw.DISPARITY_LEFT
    assert np.array_equal(ctx.bgr2gray(L), gold["gray_L"])
    assert np.array_equal(np.stack(ctx.computeAD(L, R, LEFT, 0, D)), gold["ad"])
    assert np.array_equal(np.stack(ctx.computeTAD(L, R, LEFT, 30, 0, D)), gold["tad"])
    assert np.array_equal(np.stack(ctx.computeSD(L, R, LEFT, 0, D)), gold["sd"])
    assert np.array_equal(np.stack(ctx.computeSimilarity(L, R, 0.4, 10, 50, LEFT, 0, D)), gold["similarity"])
    assert np.array_equal(np.stack(ctx.getCostSAD(L, R, LEFT, WIN, 0, D)), gold["sad"])
    assert np.array_equal(ctx.getGeodesicDist(L, WIN, 3), gold["geodesic_dist_L"])
    assert ctx.preprocess_pair(5, L, R, (40, 26), detail_boost=True)
    assert np.array_equal(ctx.download_pair(5, (26, 40, 3))[0], gold["prep_L"])
    assert ctx.preprocess_pair(5, L, R, (24, 16), detail_boost=False)
    assert np.array_equal(ctx.download_pair(5, (16, 24, 3))[1], gold["prep_R_noboost"])
    ctx.upload_pair(6, L, R)
    ctx.match_resident(6, LEFT, A.ADAPTIVE_WEIGHT, WIN, 0, D)
    assert np.array_equal(ctx.download_disparity_u8(6, (H, W), normalize=True), gold["classic_disp_u8"])
    assert np.array_equal(np.stack(ctx.computeNCC_costs(L, R, LEFT, WIN, 0, D, normalized=False)), gold["ncc_raw"])
    assert np.array_equal(np.stack(ctx.computeNCC_costs(L, R, LEFT, WIN, 0, D)), gold["ncc"])
    assert np.array_equal(ctx.stereoMatching(L, R, LEFT, A.NCC, WIN, 0, D), gold["ncc_disp"])
    exact = {"classic": A.ADAPTIVE_WEIGHT, "geodesic": A.ADAPTIVE_WEIGHT_GEODESIC, "wmedian": A.ADAPTIVE_WEIGHT_MEDIAN,
             "direct8": A.ADAPTIVE_WEIGHT_8DIRECT}
    for name, alg in exact.items():
        d, v = ctx.stereoMatching(L, R, LEFT, alg, WIN, 0, D, return_cost_volume=True)
        assert np.array_equal(d, gold[name + "_disp"]), name
        assert np.array_equal(v, gold[name + "_vol"], equal_nan=True), name
    for name, alg in {"guided": A.ADAPTIVE_WEIGHT_GUIDED_FILTER, "guided2": A.ADAPTIVE_WEIGHT_GUIDED_FILTER_2,
                      "guided3": A.ADAPTIVE_WEIGHT_GUIDED_FILTER_3}.items():
        d, v = ctx.stereoMatching(L, R, LEFT, alg, WIN, 0, D, return_cost_volume=True)
        assert np.array_equal(d, gold[name + "_disp"]), name          # WTA index bit-exact
        assert np.abs(v - gold[name + "_vol"]).max() < 1e-4, name      # float cost volume within 1e-4 (north_star)
    RIGHT = asw.DISPARITY_RIGHT
    for name, alg in {"classic_right": A.ADAPTIVE_WEIGHT, "geodesic_right": A.ADAPTIVE_WEIGHT_GEODESIC}.items():
        d, v = ctx.stereoMatching(L, R, RIGHT, alg, WIN, 0, D, return_cost_volume=True)
        assert np.array_equal(d, gold[name + "_disp"]) and np.array_equal(v, gold[name + "_vol"], equal_nan=True), name
    d, v = ctx.stereoMatching(L, R, RIGHT, A.ADAPTIVE_WEIGHT_GUIDED_FILTER, WIN, 0, D, return_cost_volume=True)
    assert np.array_equal(d, gold["guided_right_disp"]) and np.abs(v - gold["guided_right_vol"]).max() < 1e-4
    d, v = ctx.stereoMatching(L, R, RIGHT, A.ADAPTIVE_WEIGHT_GUIDED_FILTER_3, WIN, 0, D, return_cost_volume=True)
    assert np.array_equal(d, gold["guided3_right_disp"]) and np.abs(v - gold["guided3_right_vol"]).max() < 1e-4
    d, v = ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT_BLO1, WIN, 0, D, return_cost_volume=True)
    gv = gold["blo1_vol"]
    fin = np.isfinite(gv)
    assert np.array_equal(d, gold["blo1_disp"]) and np.allclose(v[fin], gv[fin], rtol=1e-4, atol=0)
    d, v = ctx.computeAdaptiveWeight_bilateralGrid((L // 64) * 64, (R // 64) * 64, LEFT, 6, 64, 0, D, return_cost_volume=True)
    assert np.array_equal(d, gold["bilgrid_disp"]) and np.array_equal(v, gold["bilgrid_vol"], equal_nan=True)
    assert np.isfinite(gold["bilgrid_vol"]).mean() > 0.3
    ctx.close()
