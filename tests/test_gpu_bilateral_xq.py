"""The xq form of the classic bilateral kernel (k_bilateral_xq.hip: thread = 4 pixels x 4 right-image positions, candidates
[0, 128) + a tail launch of k_asw_bilateral) against the oracle and against the one-kernel path, bit for bit.  It serves
both directions at win = 15, numDisparity >= 127 (the reference's configuration at 1080p, M.cpp:58 / main.cpp:94)."""
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
    c = asw.Context(0, env={"ASW_BILATERAL_XQ": "0"})  # the one-kernel form only (switches are read once, in asw_create)
    yield c
    c.close()


# (H, W, minD, numD): one and several 64-pixel tiles, partial last tile, the tile whose windows reach the right border,
# W < numD (every position clamps to column 0 somewhere), minD > 0, tails of 4 / 5 / 16+ candidates
SHAPES = [(5, 64, 0, 128), (9, 200, 0, 128), (3, 257, 2, 128), (17, 130, 0, 127), (4, 333, 48, 131), (2, 71, 0, 160),
          (21, 96, 7, 128), (1, 640, 0, 128),
          # 64..127 candidates: the 4-wavefront form (64 candidates per pass; the reference's own call passes numDisparity 64)
          (5, 64, 0, 64), (9, 200, 0, 63), (3, 257, 2, 100), (4, 333, 48, 70), (2, 71, 0, 90), (17, 130, 0, 126), (1, 640, 5, 64)]


@pytest.mark.parametrize("H,W,minD,numD", SHAPES)
def test_xq_matches_oracle_and_one_kernel_path(ctx, ctx_old, oracle, H, W, minD, numD):
    L, R, _ = make_pair(H, W, min(numD, W // 2), seed=H * 1000 + W, block=16)
    d, v = ctx.computeAdaptiveWeight(L, R, 30, 20, LEFT, 15, minD, numD, return_cost_volume=True)
    rc, dw, vw = oracle.asw_classic(L, R, 30, 20, 0, 15, minD, numD, want_vol=True)
    assert rc == 0 and v.shape == vw.shape == (numD + 1, H, W)
    assert np.array_equal(v, vw, equal_nan=True), np.argwhere(v != vw)[:5]
    assert np.array_equal(d, dw)
    d0, v0 = ctx_old.computeAdaptiveWeight(L, R, 30, 20, LEFT, 15, minD, numD, return_cost_volume=True)
    assert np.array_equal(v, v0, equal_nan=True) and np.array_equal(d, d0)


# DISPARITY_RIGHT (M.cpp:1113-1142): positions run to the right, the border tile is the first of a row; minD > 0 only while the
# last tile still holds columns W-8..W-1 (x0_last + minD <= W - 1), else the one-kernel path serves the call
RIGHT_SHAPES = [(5, 64, 0, 128), (9, 200, 0, 128), (3, 257, 0, 128), (17, 130, 1, 127), (4, 333, 12, 131), (2, 71, 6, 160),
                (21, 96, 7, 128), (1, 640, 0, 128), (3, 200, 40, 128),
                (5, 64, 0, 64), (9, 200, 0, 63), (3, 257, 0, 100), (4, 333, 12, 70), (2, 71, 6, 90), (1, 640, 3, 64)]


@pytest.mark.parametrize("H,W,minD,numD", RIGHT_SHAPES)
def test_xq_right_matches_oracle_and_one_kernel_path(ctx, ctx_old, oracle, H, W, minD, numD):
    L, R, _ = make_pair(H, W, min(numD, W // 2), seed=H * 1000 + W + 7, block=16)
    d, v = ctx.computeAdaptiveWeight(L, R, 30, 20, RIGHT, 15, minD, numD, return_cost_volume=True)
    rc, dw, vw = oracle.asw_classic(L, R, 30, 20, 1, 15, minD, numD, want_vol=True)
    assert rc == 0 and v.shape == vw.shape == (numD + 1, H, W)
    assert np.array_equal(v, vw, equal_nan=True), np.argwhere(v != vw)[:5]
    assert np.array_equal(d, dw)
    d0, v0 = ctx_old.computeAdaptiveWeight(L, R, 30, 20, RIGHT, 15, minD, numD, return_cost_volume=True)
    assert np.array_equal(v, v0, equal_nan=True) and np.array_equal(d, d0)


def test_xq_right_mid_size_and_lr_check(ctx, ctx_old):
    L, R, _ = make_pair(135, 480, 128, seed=15)
    d, v = ctx.computeAdaptiveWeight(L, R, 30, 20, RIGHT, 15, 0, 128, return_cost_volume=True)
    d0, v0 = ctx_old.computeAdaptiveWeight(L, R, 30, 20, RIGHT, 15, 0, 128, return_cost_volume=True)
    assert np.array_equal(v, v0, equal_nan=True) and np.array_equal(d, d0)


def test_xq_flat_and_random_images(ctx, ctx_old, oracle):
    # constant images: all costs 0, every E = 0 -> d = minD everywhere; pure noise: ties and large gray steps
    for L, R in ((np.full((6, 150, 3), 80, np.uint8), np.full((6, 150, 3), 80, np.uint8)),
                 (np.random.default_rng(3).integers(0, 256, (6, 150, 3)).astype(np.uint8),
                  np.random.default_rng(4).integers(0, 256, (6, 150, 3)).astype(np.uint8))):
        d, v = ctx.computeAdaptiveWeight(L, R, 30, 20, LEFT, 15, 0, 128, return_cost_volume=True)
        rc, dw, vw = oracle.asw_classic(L, R, 30, 20, 0, 15, 0, 128, want_vol=True)
        assert rc == 0 and np.array_equal(v, vw, equal_nan=True) and np.array_equal(d, dw)


def test_xq_other_gammas_and_selector(ctx, ctx_old, oracle):
    L, R, _ = make_pair(7, 180, 60, seed=12, block=16)
    for gc, gg in ((30, 2), (7.5, 11.25), (255, 1)):
        d, v = ctx.computeAdaptiveWeight(L, R, gc, gg, LEFT, 15, 0, 130, return_cost_volume=True)
        rc, dw, vw = oracle.asw_classic(L, R, gc, gg, 0, 15, 0, 130, want_vol=True)
        assert rc == 0 and np.array_equal(v, vw, equal_nan=True) and np.array_equal(d, dw), (gc, gg)
    assert np.array_equal(ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT, 15, 0, 128), oracle.stereo_matching(L, R, 0, 2, 15, 0, 128)[1])
    # without the volume (the WTA-only call of the selector) and through the resident API
    ctx.upload_pair(5, L, R)
    ctx.match_resident(5, LEFT, A.ADAPTIVE_WEIGHT, 15, 0, 128, keep_volume=False)
    assert np.array_equal(ctx.download_disparity(5, (7, 180)), oracle.asw_classic(L, R, 30, 20, 0, 15, 0, 128)[1])


def test_other_windows_stay_on_the_one_kernel_path(ctx, ctx_old, oracle):
    L, R, _ = make_pair(6, 140, 40, seed=13, block=16)
    for dt, win in ((RIGHT, 13), (LEFT, 13), (LEFT, 17)):
        d, v = ctx.computeAdaptiveWeight(L, R, 30, 20, dt, win, 0, 128, return_cost_volume=True)
        rc, dw, vw = oracle.asw_classic(L, R, 30, 20, int(dt), win, 0, 128, want_vol=True)
        assert rc == 0 and np.array_equal(v, vw, equal_nan=True) and np.array_equal(d, dw), (dt, win)


def test_xq_mid_size_equals_one_kernel_path(ctx, ctx_old):
    # 270 x 480, D = 128: too long for the oracle in a unit test, so the two GPU paths check each other (the one-kernel path
    # is itself checked against the oracle at this size by the fuzz sweeps and at 1080p on row bands)
    L, R, _ = make_pair(270, 480, 128, seed=14)
    d, v = ctx.computeAdaptiveWeight(L, R, 30, 20, LEFT, 15, 0, 128, return_cost_volume=True)
    d0, v0 = ctx_old.computeAdaptiveWeight(L, R, 30, 20, LEFT, 15, 0, 128, return_cost_volume=True)
    assert np.array_equal(v, v0, equal_nan=True) and np.array_equal(d, d0)
