"""GPU parity tests: every HIP path, called through the C-ABI (ctypes), against the CPU oracle on the
same seeded inputs.  Integer / index outputs bit-exact; float cost volumes within rtol 1e-4
(BASELINE.json north_star).  Run with `-m gpu` on an MI355X."""
import numpy as np
import pytest

import aswstereomatch_amd as asw
from aswstereomatch_amd.synth import make_pair, shifted_pair

pytestmark = pytest.mark.gpu

A = asw.StereoMatchingAlgorithms
LEFT, RIGHT = asw.DISPARITY_LEFT, asw.DISPARITY_RIGHT


@pytest.fixture(scope="module")
def ctx():
    c = asw.Context(0)
    yield c
    c.close()


def _close(a, b, rtol=1e-4):
    return np.allclose(a, b, rtol=rtol, atol=1e-30)


# ---------------------------------------------------------------- streaming kernels
@pytest.mark.parametrize("shape", [(7, 13), (37, 64), (288, 384)])
def test_bgr2gray(ctx, oracle, shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, shape + (3,)).astype(np.uint8)
    assert np.array_equal(ctx.bgr2gray(img), oracle.bgr2gray(img))


@pytest.mark.parametrize("H,W,C,dt,minD,numD", [
    (9, 21, 3, 0, 0, 5), (9, 21, 3, 1, 0, 5), (16, 64, 1, 0, 0, 9), (16, 64, 1, 1, 2, 7),
    (33, 100, 3, 0, 3, 40), (5, 8, 3, 0, 0, 30),  # disparity range larger than the width: repeated reflection
    (288, 384, 3, 0, 0, 16)])
def test_cost_ad_tad(ctx, oracle, H, W, C, dt, minD, numD):
    rng = np.random.default_rng(H * W + C)
    shp = (H, W, 3) if C == 3 else (H, W)
    L = rng.integers(0, 256, shp).astype(np.uint8)
    R = rng.integers(0, 256, shp).astype(np.uint8)
    rc, want = oracle.compute_ad(L, R, dt, minD, numD)
    got = ctx.computeAD(L, R, dt, minD, numD)
    assert rc == 0 and len(got) == numD and np.array_equal(np.stack(got), want)
    rc, want = oracle.compute_tad(L, R, dt, 30, minD, numD)
    got = ctx.computeTAD(L, R, dt, 30, minD, numD)
    assert np.array_equal(np.stack(got), want) and set(np.unique(np.stack(got))) <= {0, 255}
    rc, want = oracle.compute_sd(L // 8, R // 8, dt, minD, numD)  # small differences, so that not every square saturates
    assert rc == 0 and np.array_equal(np.stack(ctx.computeSD(L // 8, R // 8, dt, minD, numD)), want)
    assert np.array_equal(np.stack(ctx.computeSD(L, R, dt, minD, numD)), oracle.compute_sd(L, R, dt, minD, numD)[1])


def test_cost_ad_reference_error_behaviour(ctx):
    L = np.zeros((8, 8, 3), np.uint8)
    assert ctx.computeAD(L, np.zeros((8, 9, 3), np.uint8)) == [] and asw.last_status() == asw.ERR_SIZE_MISMATCH


@pytest.mark.parametrize("n,H,W,minD", [(5, 7, 9, 0), (17, 36, 64, 3), (64, 100, 128, 0)])
def test_wta(ctx, oracle, n, H, W, minD):
    rng = np.random.default_rng(n)
    vol = rng.random((n, H, W), dtype=np.float32)
    vol[:, 0, 0] = np.nan            # all-NaN column -> 0
    vol[:, 1, 1] = 0.5               # ties -> lowest d
    vol[0, 2, 2] = np.inf
    vol[:, 3, 3] = np.inf            # (double)inf < DBL_MAX is false -> never selected -> 0
    vol[1:, 4, 4] = -np.inf
    assert np.array_equal(ctx.winnerTakeAll(vol, minD), oracle.wta(vol, minD))


# ---------------------------------------------------------------- classic bilateral ASW
@pytest.mark.parametrize("H,W,win,minD,numD,seed", [
    (24, 40, 5, 0, 8, 3),       # smaller than one tile
    (37, 130, 7, 0, 20, 4),     # ragged tile edges, two d-chunks
    (20, 70, 15, 2, 33, 5),     # reference window, minDisparity != 0, three d-chunks
    (12, 64, 3, 0, 80, 6),      # disparity range larger than the width (max(0,x-d) clamps)
    (64, 96, 35, 0, 16, 7),     # Cones-config window (DC=8 variant)
])
def test_classic_bilateral_parity(ctx, oracle, H, W, win, minD, numD, seed):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=16)
    rc, d_want, v_want = oracle.asw_classic(L, R, 30, 20, 0, win, minD, numD, want_vol=True)
    d_got, v_got = ctx.computeAdaptiveWeight(L, R, 30, 20, LEFT, win, minD, numD, return_cost_volume=True)
    assert rc == 0 and v_got.shape == (numD + 1, H, W)
    assert np.array_equal(v_got, v_want)   # same summation order -> E bit-identical (not just 1e-4)
    assert np.array_equal(d_got, d_want)   # WTA index bit-exact
    assert _close(v_got, v_want)


def test_classic_selector_and_shift(ctx, oracle):
    d0 = 5
    L, R = shifted_pair(40, 64, d0)
    d = ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT, 7, 0, 8)
    assert (d[8:-8, 16:-8] == d0).all()
    rc, want = oracle.stereo_matching(L, R, 0, 2, 7, 0, 8)
    assert np.array_equal(d, want)
    # inclusive range: numDisparity = d0 still finds d0 (K7)
    d = ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT, 7, 0, d0)
    assert (d[8:-8, 16:-8] == d0).all()


def test_classic_tsukuba_shape_config1(ctx, oracle):
    # BASELINE.json configs[0]: 384x288, D=16, win 15
    L, R, _ = make_pair(288, 384, 16, seed=1234)
    rc, want = oracle.stereo_matching(L, R, 0, 2, 15, 0, 16)
    got = ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT, 15, 0, 16)
    assert rc == 0 and np.array_equal(got, want)


def test_error_behaviour(ctx):
    L, R, _ = make_pair(16, 32, 4, seed=1)
    assert ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT, 6, 0, 8) is None and asw.last_status() == asw.ERR_EVEN_WINDOW
    assert ctx.stereoMatching(L, R[:, :30], LEFT, A.ADAPTIVE_WEIGHT, 7, 0, 8) is None
    assert asw.last_status() == asw.ERR_SIZE_MISMATCH
    for alg in (A.BM, A.SGBM):  # OpenCV's own block matchers: not ASW, not rebuilt
        with pytest.raises(asw.AswError) as e:
            ctx.stereoMatching(L, R, LEFT, alg, 7, 0, 8)
        assert e.value.status == asw.ERR_UNSUPPORTED_METHOD


# ---------------------------------------------------------------- direct8
@pytest.mark.parametrize("H,W,win,minD,numD,seed", [
    (24, 40, 5, 0, 8, 3), (37, 130, 7, 0, 20, 4), (20, 70, 15, 2, 33, 5), (12, 64, 3, 0, 80, 6), (64, 96, 35, 0, 16, 7),
    (10, 12, 1, 0, 3, 8)])
def test_direct8_parity(ctx, oracle, H, W, win, minD, numD, seed):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=16)
    rc, d_want, v_want = oracle.asw_direct8(L, R, 0, win, minD, numD, want_vol=True)
    d_got, v_got = ctx.computeAdaptiveWeight_direct8(L, R, LEFT, win, minD, numD, return_cost_volume=True)
    assert rc == 0 and v_got.shape == (numD + 1, H, W)
    assert np.array_equal(v_got, v_want, equal_nan=True)   # same tap order -> E bit-identical
    assert np.array_equal(d_got, d_want)


def test_direct8_selector_shift_and_errors(ctx, oracle):
    d0 = 5
    L, R = shifted_pair(40, 64, d0)
    d = ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT_8DIRECT, 7, 0, 8)
    assert (d[8:-8, 16:-8] == d0).all()
    assert np.array_equal(d, oracle.stereo_matching(L, R, 0, 3, 7, 0, 8)[1])
    with pytest.raises(asw.AswError) as e:      # M.cpp:1291-1295: negative vector index in the reference
        ctx.stereoMatching(L, R, RIGHT, A.ADAPTIVE_WEIGHT_8DIRECT, 7, 0, 8)
    assert e.value.status == asw.ERR_UNSUPPORTED_LAYOUT
    assert ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT_8DIRECT, 6, 0, 8) is None and asw.last_status() == asw.ERR_EVEN_WINDOW
    # the classic tables must come back after a direct8 call (shared table cache)
    a = ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT, 7, 0, 8)
    assert np.array_equal(a, oracle.stereo_matching(L, R, 0, 2, 7, 0, 8)[1])


# ---------------------------------------------------------------- NCC cost, NCC disparity, GuidedF_3
@pytest.mark.parametrize("H,W,dt,win,minD,numD,seed", [
    (9, 16, 0, 3, 0, 4, 1), (37, 70, 0, 7, 0, 24, 2), (20, 33, 0, 15, 3, 9, 3), (6, 10, 0, 5, 0, 25, 4),
    (37, 70, 1, 7, 0, 24, 5), (20, 133, 1, 5, 2, 18, 6), (12, 64, 0, 1, 0, 3, 7)])
def test_cost_ncc(ctx, oracle, H, W, dt, win, minD, numD, seed):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=8)
    rc, want = oracle.cost_ncc(L, R, dt, win, minD, numD, raw=True)
    got = np.stack(ctx.computeNCC_costs(L, R, dt, win, minD, numD, normalized=False))
    assert rc == 0 and got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True)     # same summation order -> bit-identical
    rc, want = oracle.cost_ncc(L, R, dt, win, minD, numD)
    got = np.stack(ctx.computeNCC_costs(L, R, dt, win, minD, numD))
    assert np.array_equal(got, want, equal_nan=True)
    # single-channel input: used as it is (no colour conversion)
    g = oracle.rgb2gray(L), oracle.rgb2gray(R)
    got1 = np.stack(ctx.computeNCC_costs(g[0], g[1], dt, win, minD, numD))
    assert np.array_equal(got1, got, equal_nan=True)


@pytest.mark.parametrize("H,W,win,minD,numD,seed", [(24, 40, 5, 0, 8, 3), (37, 130, 7, 0, 20, 4), (20, 70, 15, 2, 33, 5)])
def test_ncc_disparity(ctx, oracle, H, W, win, minD, numD, seed):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=16)
    rc, want = oracle.ncc_disparity(L, R, 0, win, minD, numD)
    got = ctx.computeNCC(L, R, LEFT, win, minD, numD)
    assert rc == 0 and np.array_equal(got, want)
    assert got.max() <= minD + numD - 2                  # the last offset is never tested (M.cpp:864)
    assert np.array_equal(ctx.stereoMatching(L, R, LEFT, A.NCC, win, minD, numD), want)
    right = ctx.computeNCC(L, R, RIGHT, win, minD, numD)  # `cost > DBL_MAX` never holds (M.cpp:896): nothing written
    assert (right == 0).all() and np.array_equal(right, oracle.ncc_disparity(L, R, 1, win, minD, numD)[1])
    assert ctx.computeNCC(L, R, LEFT, 6, minD, numD) is None and asw.last_status() == asw.ERR_EVEN_WINDOW


@pytest.mark.parametrize("H,W,dt,win,minD,numD,seed", [(24, 40, 0, 5, 0, 8, 3), (40, 270, 0, 15, 0, 12, 4), (30, 100, 0, 7, 2, 9, 5),
                                                     (24, 40, 1, 5, 0, 8, 6), (30, 100, 1, 7, 2, 9, 7)])
def test_guided3_parity(ctx, oracle, H, W, dt, win, minD, numD, seed):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=16)
    rc, d_want, v_want = oracle.asw_guided3(L, R, dt, 1e-6, win, minD, numD, want_vol=True)
    d_got, v_got = ctx.computeAdaptiveWeight_GuidedF_3(L, R, dt, 1e-6, win, minD, numD, return_cost_volume=True)
    assert rc == 0 and v_got.shape == (numD, H, W)
    assert _close(v_got, v_want)             # 1e-4 on the float cost volume
    assert np.array_equal(d_got, d_want)     # WTA index bit-exact
    sel = ctx.stereoMatching(L, R, dt, A.ADAPTIVE_WEIGHT_GUIDED_FILTER_3, win, minD, numD)
    assert np.array_equal(sel, oracle.stereo_matching(L, R, dt, 9, win, minD, numD)[1])


# ---------------------------------------------------------------- TAD C+G similarity, SAD cost
@pytest.mark.parametrize("H,W,minD,numD,seed", [(9, 16, 0, 4, 1), (37, 70, 0, 24, 2), (20, 33, 3, 9, 3), (6, 10, 0, 25, 4)])
def test_cost_similarity(ctx, oracle, H, W, minD, numD, seed):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=8)
    rc, want = oracle.compute_similarity(L, R, 0.4, 10, 50, 0, minD, numD)
    got = np.stack(ctx.computeSimilarity(L, R, 0.4, 10, 50, LEFT, minD, numD))
    assert rc == 0 and np.array_equal(got, want)          # same f32 operation order -> bit-exact
    rc, wantp = oracle.compute_similarity(L, R, 0.4, 10, 50, 0, minD, numD, win=7)
    gotp = np.stack(ctx.computeSimilarity(L, R, 0.4, 10, 50, LEFT, minD, numD, winSize=7))
    assert gotp.shape == (numD, H + 6, W + 6) and np.array_equal(gotp, wantp)


def test_cost_similarity_reference_errors(ctx):
    L, R, _ = make_pair(12, 16, 4, seed=1)
    assert ctx.computeSimilarity(L, R, 0.4, 10, 50, LEFT, 0, 4, winSize=6) == [] and asw.last_status() == asw.ERR_EVEN_WINDOW
    with pytest.raises(asw.AswError) as e:   # RIGHT branch throws cv::Exception in the reference (App. B-7)
        ctx.computeSimilarity(L, R, 0.4, 10, 50, RIGHT, 0, 4)
    assert e.value.status == asw.ERR_UNSUPPORTED_LAYOUT


def _ulp_stats(got, want):
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    scale = np.maximum(np.abs(want.astype(np.float64)), 1e-30)
    return float((d / scale).max()), int((got != want).sum())


@pytest.mark.parametrize("H,W,dt,win,minD,numD", [(20, 40, 0, 5, 0, 6), (33, 300, 0, 15, 0, 10), (33, 300, 1, 15, 2, 5), (70, 64, 0, 7, 0, 3)])
def test_cost_sad(ctx, oracle, H, W, dt, win, minD, numD):
    L, R, _ = make_pair(H, W, 8, seed=H + W, block=16)
    rc, want = oracle.cost_sad(L, R, dt, win, minD, numD)
    got = np.stack(ctx.getCostSAD(L, R, dt, win, minD, numD))
    rel, nbad = _ulp_stats(got, want)
    # integer inputs: every f64 window sum is exact in any order -> bit-exact
    assert rc == 0 and nbad == 0, (rel, nbad)


@pytest.mark.parametrize("H,W,C,r,seed", [(18, 22, 3, 5, 4), (40, 300, 3, 15, 5), (40, 300, 6, 15, 6), (70, 50, 3, 6, 7)])
def test_guided_filter(ctx, oracle, H, W, C, r, seed):
    rng = np.random.default_rng(seed)
    guide = rng.integers(0, 256, (H, W, C)).astype(np.uint8)
    P = (rng.random((H, W), dtype=np.float32) * 150 + 5100).astype(np.float32)
    rc, want = oracle.guided_filter(guide, P, r, 1e-6)
    got = ctx.getGuidedFilter(guide, P, r, 1e-6)
    assert rc == 0 and got.shape == want.shape
    # q lives on the normalised [0,1] scale of P (M.cpp:2775): tolerance 1e-4 absolute there
    assert np.abs(got - want).max() < 1e-4, np.abs(got - want).max()
    # sliding f64 sums vs the oracle's direct sums: expect (almost) all values identical
    assert (got != want).mean() < 0.02, (got != want).mean()


def test_guided_filter_reference_errors(ctx):
    rng = np.random.default_rng(0)
    g1 = rng.integers(0, 256, (8, 8, 1)).astype(np.uint8)
    with pytest.raises(asw.AswError) as e:   # 1-channel guide: empty Mat then cv::Exception (M.cpp:2732, 2847)
        ctx.getGuidedFilter(g1, np.zeros((8, 8), np.float32), 3, 1e-6)
    assert e.value.status == asw.ERR_UNSUPPORTED_LAYOUT
    g3 = rng.integers(0, 256, (8, 8, 3)).astype(np.uint8)
    assert ctx.getGuidedFilter(g3, np.zeros((8, 9), np.float32), 3, 1e-6) is None  # size mismatch -> Mat()


@pytest.mark.parametrize("H,W,win,minD,numD,seed", [(24, 40, 5, 0, 8, 3), (40, 270, 15, 0, 20, 4), (30, 100, 7, 2, 12, 5)])
def test_guided2_parity(ctx, oracle, H, W, win, minD, numD, seed):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=16)
    rc, d_want, v_want = oracle.asw_guided2(L, R, 0, 1e-6, win, minD, numD, want_vol=True)
    d_got, v_got = ctx.computeAdaptiveWeight_GuidedF_2(L, R, LEFT, 1e-6, win, minD, numD, return_cost_volume=True)
    assert rc == 0 and v_got.shape == (numD, H, W)
    assert np.abs(v_got - v_want).max() < 1e-4
    assert np.array_equal(d_got, d_want)
    sel = ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT_GUIDED_FILTER_2, win, minD, numD)
    assert np.array_equal(sel, d_want)


@pytest.mark.parametrize("H,W,win,minD,numD,seed", [(24, 40, 5, 0, 8, 3), (40, 270, 15, 0, 12, 4), (30, 100, 7, 2, 9, 5)])
def test_guided_parity(ctx, oracle, H, W, win, minD, numD, seed):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=16)
    rc, d_want, v_want = oracle.asw_guided(L, R, 0, 1e-6, win, minD, numD, want_vol=True)
    d_got, v_got = ctx.computeAdaptiveWeight_GuidedF(L, R, LEFT, 1e-6, win, minD, numD, return_cost_volume=True)
    assert rc == 0 and np.abs(v_got - v_want).max() < 1e-4
    assert np.array_equal(d_got, d_want)
    sel = ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT_GUIDED_FILTER, win, minD, numD)
    assert np.array_equal(sel, d_want)
    assert ctx.computeAdaptiveWeight_GuidedF(L, R, LEFT, 1e-6, 6, minD, numD) is None  # even window -> Mat()


# ---------------------------------------------------------------- geodesic ASW
@pytest.mark.parametrize("H,W,win,iters", [(9, 9, 5, 3), (20, 70, 7, 3), (16, 40, 15, 3), (12, 20, 5, 1), (12, 20, 5, 2), (12, 20, 3, 0)])
def test_geodesic_dist(ctx, oracle, H, W, win, iters):
    L, _, _ = make_pair(H, W, 4, seed=win + H, block=8)
    rc, want = oracle.geodesic_dist(L, win, iters)
    got = ctx.getGeodesicDist(L, win, iters)
    assert rc == 0 and np.array_equal(got, want)   # exact integers / FLT_MAX


def test_geodesic_dist_even_window(ctx):
    L, _, _ = make_pair(8, 8, 2, seed=1)
    assert ctx.getGeodesicDist(L, 4) is None and asw.last_status() == asw.ERR_EVEN_WINDOW


@pytest.mark.parametrize("H,W,win,minD,numD,seed", [
    (24, 40, 5, 0, 8, 3), (37, 130, 7, 0, 20, 4), (20, 70, 15, 2, 33, 5), (12, 64, 3, 0, 80, 6)])
def test_geodesic_parity(ctx, oracle, H, W, win, minD, numD, seed):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=16)
    rc, d_want, v_want = oracle.asw_geodesic(L, R, 0, win, minD, numD, want_vol=True)
    d_got, v_got = ctx.computeAdaptiveWeight_geodesic(L, R, LEFT, win, minD, numD, return_cost_volume=True)
    assert rc == 0 and v_got.shape == (numD + 1, H, W)
    assert np.array_equal(v_got, v_want, equal_nan=True)   # all f64 sums are exact -> E bit-identical
    assert np.array_equal(d_got, d_want)
    sel = ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT_GEODESIC, win, minD, numD)
    assert np.array_equal(sel, d_want)


def test_geodesic_flat_windows_are_nan(ctx, oracle):
    # windows flat in both images: weights all 0 -> 0/0 = NaN -> never selected -> build value 0 (App. B-9)
    L = np.full((12, 24, 3), 77, np.uint8)
    R = L.copy()
    rc, d_want, v_want = oracle.asw_geodesic(L, R, 0, 5, 0, 4, want_vol=True)
    d_got, v_got = ctx.computeAdaptiveWeight_geodesic(L, R, LEFT, 5, 0, 4, return_cost_volume=True)
    assert np.isnan(v_want).all() and np.isnan(v_got).all() and (d_got == 0).all() and np.array_equal(d_got, d_want)


# ---------------------------------------------------------------- weighted-median ASW
@pytest.mark.parametrize("H,W,win,minD,numD,seed", [(16, 24, 5, 0, 6, 3), (20, 70, 7, 0, 12, 4), (14, 40, 15, 1, 9, 5), (10, 12, 3, 0, 20, 6)])
def test_wmedian_parity(ctx, oracle, H, W, win, minD, numD, seed):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=8)
    rc, d_want, v_want = oracle.asw_wmedian(L, R, 0, win, 10, 10, minD, numD, want_vol=True)
    d_got, v_got = ctx.computeAdaptiveWeight_WeightedMedian(L, R, LEFT, win, 10, 10, minD, numD, return_cost_volume=True)
    assert rc == 0 and v_got.shape == (numD, H, W)
    assert np.array_equal(v_got, v_want)    # the median is one of the (bit-exact) input costs
    assert np.array_equal(d_got, d_want)
    sel = ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT_MEDIAN, win, minD, numD)
    assert np.array_equal(sel, d_want)


def test_wmedian_flat_ties(ctx, oracle):
    # heavy cost ties (identical images, d=0 exact match): stable multimap order decides the crossing
    L, _, _ = make_pair(16, 32, 4, seed=8, block=8)
    R = L.copy()
    rc, d_want, v_want = oracle.asw_wmedian(L, R, 0, 5, 10, 10, 0, 5, want_vol=True)
    d_got, v_got = ctx.computeAdaptiveWeight_WeightedMedian(L, R, LEFT, 5, 10, 10, 0, 5, return_cost_volume=True)
    assert np.array_equal(v_got, v_want) and np.array_equal(d_got, d_want)


def test_all_methods_recover_shift_k6(ctx):
    d0 = 5
    L, R = shifted_pair(40, 64, d0)
    for alg in (A.ADAPTIVE_WEIGHT, A.ADAPTIVE_WEIGHT_GEODESIC, A.ADAPTIVE_WEIGHT_GUIDED_FILTER_2, A.ADAPTIVE_WEIGHT_MEDIAN):
        d = ctx.stereoMatching(L, R, LEFT, alg, 7, 0, 8)
        assert (d[8:-8, 16:-8] == d0).all(), alg


# ---------------------------------------------------------------- DISPARITY_RIGHT
@pytest.mark.parametrize("H,W,win,minD,numD,seed", [(24, 40, 5, 0, 8, 3), (37, 130, 7, 0, 20, 4), (20, 70, 15, 2, 33, 5), (12, 64, 3, 0, 80, 6)])
def test_classic_right_parity(ctx, oracle, H, W, win, minD, numD, seed):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=16)
    rc, d_want, v_want = oracle.asw_classic(L, R, 30, 20, 1, win, minD, numD, want_vol=True)
    d_got, v_got = ctx.computeAdaptiveWeight(L, R, 30, 20, RIGHT, win, minD, numD, return_cost_volume=True)
    assert rc == 0 and np.array_equal(v_got, v_want) and np.array_equal(d_got, d_want)
    assert np.array_equal(ctx.stereoMatching(L, R, RIGHT, A.ADAPTIVE_WEIGHT, win, minD, numD), d_want)


@pytest.mark.parametrize("H,W,win,minD,numD,seed", [(24, 40, 5, 0, 8, 3), (37, 130, 7, 0, 20, 4), (20, 70, 15, 2, 33, 5)])
def test_geodesic_right_parity(ctx, oracle, H, W, win, minD, numD, seed):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=16)
    rc, d_want, v_want = oracle.asw_geodesic(L, R, 1, win, minD, numD, want_vol=True)
    d_got, v_got = ctx.computeAdaptiveWeight_geodesic(L, R, RIGHT, win, minD, numD, return_cost_volume=True)
    assert rc == 0 and np.array_equal(v_got, v_want, equal_nan=True) and np.array_equal(d_got, d_want)


@pytest.mark.parametrize("H,W,win,minD,numD,seed", [(24, 40, 5, 0, 8, 3), (40, 270, 15, 0, 12, 4), (30, 100, 7, 2, 9, 5)])
def test_guided_right_parity(ctx, oracle, H, W, win, minD, numD, seed):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=16)
    rc, d_want, v_want = oracle.asw_guided(L, R, 1, 1e-6, win, minD, numD, want_vol=True)
    d_got, v_got = ctx.computeAdaptiveWeight_GuidedF(L, R, RIGHT, 1e-6, win, minD, numD, return_cost_volume=True)
    assert rc == 0 and np.abs(v_got - v_want).max() < 1e-4 and np.array_equal(d_got, d_want)


def test_right_unsupported_where_reference_throws(ctx):
    L, R, _ = make_pair(16, 32, 4, seed=1)
    for alg in (A.ADAPTIVE_WEIGHT_GUIDED_FILTER_2, A.ADAPTIVE_WEIGHT_MEDIAN):   # App. B-7 / B-13
        with pytest.raises(asw.AswError) as e:
            ctx.stereoMatching(L, R, RIGHT, alg, 5, 0, 4)
        assert e.value.status == asw.ERR_UNSUPPORTED_LAYOUT


def test_right_recovers_shift(ctx):
    # L(x) = R(x - d0)  <=>  R(x) = L(x + d0): the right-referenced disparity is d0 as well
    d0 = 5
    L, R = shifted_pair(40, 64, d0)
    for alg in (A.ADAPTIVE_WEIGHT, A.ADAPTIVE_WEIGHT_GEODESIC):
        d = ctx.stereoMatching(L, R, RIGHT, alg, 7, 0, 8)
        assert (d[8:-8, 8:-16] == d0).all(), alg


# ---------------------------------------------------------------- O(1)-bilateral ASW (BLO1), SURVEY 8f row f1
@pytest.mark.parametrize("H,W,win,numD,dt,rate,seed", [(24, 40, 5, 8, 0, 0.015, 3), (40, 150, 15, 12, 0, 0.015, 4), (30, 100, 7, 9, 1, 0.015, 5),
                                                       (20, 64, 5, 6, 0, 0.04, 6), (1, 1, 3, 2, 0, 0.015, 7), (5, 129, 7, 20, 1, 0.1, 8)])
def test_blo1_parity(ctx, oracle, H, W, win, numD, dt, rate, seed):
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=16)
    rc, d_want, v_want = oracle.asw_blo1(L, R, dt, rate, win, 0, numD, want_vol=True)
    d_got, v_got = ctx.computeAdaptiveWeight_BLO1(L, R, dt, rate, win, 0, numD, return_cost_volume=True)
    assert rc == 0 and v_got.shape == (numD, H, W)
    fin = np.isfinite(v_want)
    assert np.array_equal(fin, np.isfinite(v_got)) and np.array_equal(np.isnan(v_want), np.isnan(v_got))
    assert np.array_equal(v_got[fin], v_want[fin])   # the gather kernel adds in the oracle's association: bit-identical
    assert np.array_equal(d_got, d_want)
    if rate == 0.015:
        assert np.array_equal(ctx.stereoMatching(L, R, dt, A.ADAPTIVE_WEIGHT_BLO1, win, 0, numD), d_want)  # selector literal M.cpp:70


def test_blo1_reference_limits(ctx):
    L, R, _ = make_pair(16, 32, 4, seed=1)
    assert ctx.computeAdaptiveWeight_BLO1(L, R, LEFT, 0.015, 6, 0, 4) is None and asw.last_status() == asw.ERR_EVEN_WINDOW
    with pytest.raises(asw.AswError) as e:   # absolute-offset indexing of the reference: only minDisparity == 0 is defined
        ctx.computeAdaptiveWeight_BLO1(L, R, LEFT, 0.015, 5, 2, 4)
    assert e.value.status == asw.ERR_BAD_ARGUMENT
    with pytest.raises(asw.AswError):        # 256*rate < 1: the reference's key loop never terminates
        ctx.computeAdaptiveWeight_BLO1(L, R, LEFT, 0.001, 5, 0, 4)
    d0 = 5
    Ls, Rs = shifted_pair(40, 64, d0)
    assert (ctx.stereoMatching(Ls, Rs, LEFT, A.ADAPTIVE_WEIGHT_BLO1, 7, 0, 8)[8:-8, 16:-8] == d0).all()


# ---------------------------------------------------------------- row f3: driver-side pre/post-processing on the device
@pytest.mark.parametrize("sh,sw,dw,dh,seed", [(90, 160, 80, 45, 1), (72, 128, 64, 36, 2), (97, 131, 64, 36, 3), (360, 640, 640, 360, 4),
                                              (50, 70, 113, 81, 5), (1080, 1920, 640, 360, 6)])
def test_preprocess_pair(ctx, oracle, sh, sw, dw, dh, seed):
    rng = np.random.default_rng(seed)
    L = rng.integers(0, 256, (sh, sw, 3)).astype(np.uint8)
    R = rng.integers(0, 256, (sh, sw, 3)).astype(np.uint8)
    for boost in (False, True):
        assert ctx.preprocess_pair(0, L, R, (dw, dh), detail_boost=boost)
        gl, gr = ctx.download_pair(0, (dh, dw, 3))
        assert np.array_equal(gl, oracle.preprocess(L, (dw, dh), boost)), boost   # u8 results: bit-exact
        assert np.array_equal(gr, oracle.preprocess(R, (dw, dh), boost)), boost


def test_preprocess_then_match_then_u8(ctx, oracle):
    # the reference's main(): resize + boost, stereoMatching(..., 15, 0, 64), convertTo(8U) + normalize(0,255)
    L, R, _ = make_pair(144, 256, 24, seed=11, block=24)
    assert ctx.preprocess_pair(3, L, R, (128, 72), detail_boost=True)
    ctx.match_resident(3, LEFT, A.ADAPTIVE_WEIGHT_GUIDED_FILTER_2, 15, 0, 24)
    d = ctx.download_disparity(3, (72, 128))
    pl, pr = oracle.preprocess(L, (128, 72), True), oracle.preprocess(R, (128, 72), True)
    rc, want = oracle.stereo_matching(pl, pr, 0, 8, 15, 0, 24)
    assert rc == 0 and np.array_equal(d, want)
    for norm in (False, True):
        assert np.array_equal(ctx.download_disparity_u8(3, (72, 128), normalize=norm), oracle.disparity_to_u8(want, norm))
    with pytest.raises(asw.AswError):   # no frame in that slot
        ctx.download_disparity_u8(9, (72, 128))
    with pytest.raises(asw.AswError) as e:   # the resident API has no reference behaviour to mimic: it raises
        ctx.preprocess_pair(3, L, R[:, :200], (128, 72))
    assert e.value.status == asw.ERR_SIZE_MISMATCH and asw.last_status() == asw.ERR_SIZE_MISMATCH


# ---------------------------------------------------------------- bilateral grid (enum 5, inventory #12)
def _smooth_pair(H, W, d, seed):
    """Low-texture pair (a few flat regions): grid bins then hold enough pixels for the int counts to survive the smoothing."""
    rng = np.random.default_rng(seed)
    base = np.kron(rng.integers(0, 6, (H // 8 + 1, W // 8 + 1)) * 45, np.ones((8, 8), int))[:H, :W]
    L = np.repeat(base[:, :, None], 3, axis=2).astype(np.uint8)
    R = np.roll(L, -d, axis=1)
    return L, R


@pytest.mark.parametrize("H,W,sS,sR,minD,numD,smooth", [
    (40, 64, 10, 10, 0, 6, True), (33, 50, 6, 128, 0, 5, False), (20, 30, 6, 64, 1, 3, False), (6, 5, 10, 10, 0, 2, False),
    (48, 70, 5.5, 100, 0, 4, False), (17, 200, 7, 33.3, 2, 9, True), (64, 96, 10, 40, 0, 8, False), (9, 9, 3, 300, 0, 12, False),
    (48, 70, 2.5, 100, 0, 4, False),
    (33, 50, 12, 6.5, 1, 3, True), (48, 80, 16, 4.0, 0, 3, True), (21, 33, 8, 3.0, 1, 3, True)])   # > 36 bins per range axis: the slicing gathers from global memory
def test_bilateral_grid_vs_oracle(ctx, oracle, H, W, sS, sR, minD, numD, smooth):
    L, R = _smooth_pair(H, W, 3, H + W) if smooth else make_pair(H, W, max(2, numD), seed=H * W, block=8)[:2]
    rc, dw, vw = oracle.asw_bilgrid(L, R, 0, sS, sR, minD, numD, want_vol=True)
    d, v = ctx.computeAdaptiveWeight_bilateralGrid(L, R, LEFT, sS, sR, minD, numD, return_cost_volume=True)
    assert rc == 0 and v.shape == (numD + 1, H, W)
    assert np.array_equal(v, vw, equal_nan=True)   # bit-exact: same f64 expression order, integer bin sums
    assert np.array_equal(d, dw)
    if sS >= 5 and H >= 17 and sR > 3:
        assert np.isfinite(vw).mean() > 0.3        # not vacuous: a share of the interpolated counts is non-zero


def test_bilateral_grid_selector_and_errors(ctx, oracle):
    L, R = _smooth_pair(40, 64, 2, 5)
    d = ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT_BILATERAL_GRID, 15, 0, 5)   # rates 10, 10 (M.cpp:67)
    assert np.array_equal(d, oracle.asw_bilgrid(L, R, 0, 10, 10, 0, 5)[1])
    assert np.array_equal(d, oracle.stereo_matching(L, R, 0, 5, 15, 0, 5)[1])
    d2, v2 = ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT_BILATERAL_GRID, 15, 0, 5, return_cost_volume=True)
    assert v2.shape == (6, 40, 64) and np.array_equal(d2, d)   # inclusive range: numD + 1 planes (asw_volume_planes)
    assert np.array_equal(v2, oracle.asw_bilgrid(L, R, 0, 10, 10, 0, 5, want_vol=True)[2], equal_nan=True)
    gray = oracle.bgr2gray(L), oracle.bgr2gray(R)
    assert np.array_equal(ctx.computeAdaptiveWeight_bilateralGrid(gray[0], gray[1], LEFT, 10, 10, 0, 5), d)  # 1-channel input
    with pytest.raises(asw.AswError) as e:   # the reference's RIGHT branch reads one past the image row
        ctx.computeAdaptiveWeight_bilateralGrid(L, R, RIGHT, 10, 10, 0, 5)
    assert e.value.status == asw.ERR_UNSUPPORTED_LAYOUT
    for rates in ((0, 10), (10, 0), (10, 1.0)):   # division by the rate / more than 101 bins per range axis
        with pytest.raises(asw.AswError) as e:
            ctx.computeAdaptiveWeight_bilateralGrid(L, R, LEFT, rates[0], rates[1], 0, 5)
        assert e.value.status == asw.ERR_BAD_ARGUMENT


# ---------------------------------------------------------------- left-right check (consumer of the RIGHT maps)
def test_left_right_check(ctx, oracle):
    L, R, gt = make_pair(60, 120, 12, seed=21)
    dl = ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT, 9, 0, 12)
    dr = ctx.stereoMatching(L, R, RIGHT, A.ADAPTIVE_WEIGHT, 9, 0, 12)
    for tau, inv in ((1.0, -1.0), (0.0, 0.0), (3.0, np.float32(np.nan))):
        got, bad = ctx.leftRightCheck(dl, dr, tau, inv)
        want, wbad = oracle.lr_check(dl, dr, tau, inv)
        assert np.array_equal(got, want, equal_nan=True) and bad == wbad
    got, bad = ctx.leftRightCheck(dl, dr, 1.0, -1.0)
    kept = got >= 0
    assert 0.5 < kept.mean() < 1.0                      # occlusions and border columns are rejected, the bulk survives
    assert (np.abs(got[kept] - gt[kept]) <= 1).mean() > 0.9   # and what survives is mostly right
    # NaN / out-of-image targets
    a = np.array([[np.nan, 5.0, 1.0, 0.0, np.inf, -1e30]], np.float32)
    b = np.array([[0.0, 9.0, 9.0, 0.0, np.inf, 0.0]], np.float32)
    got, bad = ctx.leftRightCheck(a, b, 1.0, -2.0)
    assert np.array_equal(got, [[-2.0, -2.0, -2.0, 0.0, -2.0, -2.0]]) and bad == 5
    assert np.array_equal(got, oracle.lr_check(a, b, 1.0, -2.0)[0])
    with pytest.raises(asw.AswError):
        ctx.leftRightCheck(a, b, -1.0)


# ---------------------------------------------------------------- VERDICT r01 item 4 (b), (c)
def test_gray_constant_sets_are_self_consistent_gpu_vs_oracle(oracle):
    """the 14-bit BGR2GRAY constants (OpenCV 4.1.0, default) and the 15-bit set of later releases behind ONE
    switch in the oracle and in the library; with either set the GPU path equals the oracle bit for bit."""
    L, R, _ = make_pair(30, 70, 10, seed=15, block=10)
    c = asw.Context(0)
    try:
        for bits in (15, 14):
            c.set_gray_bits(bits)
            oracle.set_gray_bits(bits)
            assert np.array_equal(c.bgr2gray(L), oracle.bgr2gray(L))
            d, v = c.computeAdaptiveWeight(L, R, 30, 20, LEFT, 7, 0, 10, return_cost_volume=True)
            rc, dw, vw = oracle.asw_classic(L, R, 30, 20, 0, 7, 0, 10, want_vol=True)
            assert rc == 0 and np.array_equal(v, vw) and np.array_equal(d, dw), bits
            assert np.array_equal(np.stack(c.getCostSAD(L, R, LEFT, 7, 0, 10)), oracle.cost_sad(L, R, 0, 7, 0, 10)[1]), bits
            assert np.array_equal(np.stack(c.computeNCC_costs(L, R, LEFT, 5, 0, 6, normalized=False)),
                                  oracle.cost_ncc(L, R, 0, 5, 0, 6, raw=True)[1], equal_nan=True), bits
            d, v = c.computeAdaptiveWeight_BLO1(L, R, LEFT, 0.015, 5, 0, 6, return_cost_volume=True)
            rc, dw, vw = oracle.asw_blo1(L, R, 0, 0.015, 5, 0, 6, want_vol=True)
            fin = np.isfinite(vw)
            assert rc == 0 and np.array_equal(v[fin], vw[fin]) and np.array_equal(d, dw), bits
        g14, g15 = oracle.bgr2gray(L), None
        oracle.set_gray_bits(15)
        g15 = oracle.bgr2gray(L)
        assert not np.array_equal(g14, g15)   # the switch does something
    finally:
        oracle.set_gray_bits(14)
        c.close()
    with pytest.raises(asw.AswError):
        asw.Context(0).set_gray_bits(16)


def test_guided_paths_vs_opencv_literal_box_sums(ctx, oracle):
    """orc_set_box_mode(1): the oracle's box filter in OpenCV's literal RowSum / ColumnSum sliding form (the reference's own
    arithmetic on finite data).  The kernels (vertical sliding + horizontal direct sums) must agree with it within the same
    tolerance as with the canonical window sums, and pick the same disparities."""
    L, R, _ = make_pair(120, 260, 24, seed=16)
    oracle.set_box_mode(1)
    try:
        for name, fo, fg in (("guided2", oracle.asw_guided2, ctx.computeAdaptiveWeight_GuidedF_2),
                             ("guided", oracle.asw_guided, ctx.computeAdaptiveWeight_GuidedF)):
            rc, dw, vw = fo(L, R, 0, 1e-6, 15, 0, 24, want_vol=True)
            d, v = fg(L, R, LEFT, 1e-6, 15, 0, 24, return_cost_volume=True)
            assert rc == 0 and np.all(np.abs(v - vw) <= 1e-4 * np.abs(vw) + 1e-7), name
            assert np.array_equal(d, dw), (name, int((d != dw).sum()))
        P = np.random.default_rng(17).random((120, 260), dtype=np.float32)
        for guide in (L, np.concatenate([L, R], axis=2)):
            rc, qw = oracle.guided_filter(guide, P, 15, 1e-6)
            assert rc == 0 and np.all(np.abs(ctx.getGuidedFilter(guide, P, 15, 1e-6) - qw) <= 1e-4 * np.abs(qw) + 1e-7)
        assert np.array_equal(np.stack(ctx.getCostSAD(L, R, LEFT, 15, 0, 24)), oracle.cost_sad(L, R, 0, 15, 0, 24)[1])  # integer sums: exact in any order
    finally:
        oracle.set_box_mode(0)
