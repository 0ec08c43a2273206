"""The xq form of the geodesic aggregation (k_geodesic_xq.hip: passes of 128 / 64 candidates + k_asw_geodesic for the rest,
winners merged per slice) against the oracle and against the one-kernel path, bit for bit (volume incl. NaN slots, WTA)."""
import os

import numpy as np
import pytest

import aswstereomatch_amd as asw
from aswstereomatch_amd.synth import make_pair

pytestmark = pytest.mark.gpu
A = asw.StereoMatchingAlgorithms
LEFT, RIGHT = asw.DISPARITY_LEFT, asw.DISPARITY_RIGHT


@pytest.fixture(scope="module")
def ctx():
    c = asw.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_old():
    c = asw.Context(0, env={"ASW_GEODESIC_XQ": "0"})  # the one-kernel form only (switches are read once, in asw_create)
    yield c
    c.close()


# (H, W, minD, numD): 64 only (one 4-wave pass), 64+tail, 128, 128+64+tail (KITTI's 193 candidates), 2x128, minD > 0,
# partial / single tiles, W < numD
SHAPES = [(4, 64, 0, 63), (6, 150, 0, 70), (3, 200, 0, 127), (5, 257, 0, 192), (2, 333, 3, 255), (9, 90, 17, 100), (1, 640, 0, 192),
          (7, 71, 0, 140)]


@pytest.mark.parametrize("dt", [LEFT, RIGHT])   # M.cpp:1467-1496 / 1498-1520
@pytest.mark.parametrize("H,W,minD,numD", SHAPES)
def test_geodesic_xq_matches_oracle_and_one_kernel_path(ctx, ctx_old, oracle, H, W, minD, numD, dt):
    L, R, _ = make_pair(H, W, min(numD, W // 2), seed=H * 977 + W, block=16)
    if H > 2:
        L[1:3, 20:40] = L[1, 20]   # flat patches: zero weights, 0/0 = NaN slots (App. B-9)
        R[1:3, 10:34] = R[1, 10]
    d, v = ctx.computeAdaptiveWeight_geodesic(L, R, dt, 15, minD, numD, return_cost_volume=True)
    rc, dw, vw = oracle.asw_geodesic(L, R, int(dt), 15, minD, numD, want_vol=True)
    assert rc == 0 and v.shape == vw.shape == (numD + 1, H, W)
    assert np.array_equal(v, vw, equal_nan=True), np.argwhere(~((v == vw) | (np.isnan(v) & np.isnan(vw))))[:5]
    assert np.array_equal(d, dw)
    d0, v0 = ctx_old.computeAdaptiveWeight_geodesic(L, R, dt, 15, minD, numD, return_cost_volume=True)
    assert np.array_equal(v, v0, equal_nan=True) and np.array_equal(d, d0)


def test_geodesic_xq_selector_resident_and_other_cases(ctx, ctx_old, oracle):
    L, R, _ = make_pair(6, 180, 60, seed=31, block=16)
    assert np.array_equal(ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT_GEODESIC, 15, 0, 128), oracle.stereo_matching(L, R, 0, 4, 15, 0, 128)[1])
    ctx.upload_pair(6, L, R)
    ctx.match_resident(6, LEFT, A.ADAPTIVE_WEIGHT_GEODESIC, 15, 0, 192, keep_volume=False)
    assert np.array_equal(ctx.download_disparity(6, (6, 180)), oracle.asw_geodesic(L, R, 0, 15, 0, 192)[1])
    for dt, win in ((RIGHT, 13), (LEFT, 13), (LEFT, 17)):   # stay on the one-kernel path
        d, v = ctx.computeAdaptiveWeight_geodesic(L, R, dt, win, 0, 100, return_cost_volume=True)
        rc, dw, vw = oracle.asw_geodesic(L, R, int(dt), win, 0, 100, want_vol=True)
        assert rc == 0 and np.array_equal(v, vw, equal_nan=True) and np.array_equal(d, dw), (dt, win)


@pytest.mark.parametrize("dt", [LEFT, RIGHT])
def test_geodesic_xq_mid_size_equals_one_kernel_path(ctx, ctx_old, dt):
    L, R, _ = make_pair(188, 621, 192, seed=32)
    d, v = ctx.computeAdaptiveWeight_geodesic(L, R, dt, 15, 0, 192, return_cost_volume=True)
    d0, v0 = ctx_old.computeAdaptiveWeight_geodesic(L, R, dt, 15, 0, 192, return_cost_volume=True)
    assert np.array_equal(v, v0, equal_nan=True) and np.array_equal(d, d0)
