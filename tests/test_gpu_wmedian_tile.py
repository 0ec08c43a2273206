"""The 15x15 weighted median in its tile form (k_wmedian_tile.hip: the 22x22 neighbourhood of an 8x8 pixel block is sorted once
per slice, every pixel walks the sorted list) against the oracle (M.cpp:3228-3383) and against the per-pixel sort
(k_wmedian.hip, ASW_WMEDIAN_TILE=0): the aggregated volume is one of the input costs, so both comparisons are bit for bit."""
import os

import numpy as np
import pytest

import aswstereomatch_amd as asw
from aswstereomatch_amd.synth import make_pair

pytestmark = pytest.mark.gpu
A = asw.StereoMatchingAlgorithms
LEFT = asw.DISPARITY_LEFT


@pytest.fixture(scope="module")
def ctx():
    c = asw.Context(0)
    yield c
    c.close()


def _with_env(name, value, fn):
    """fn(context) on a context created under the switch (the library reads its switches once, in asw_create)"""
    c = asw.Context(0, env={name: value})
    try:
        return fn(c)
    finally:
        c.close()


# (H, W, minD, numD): blocks cut by the right / bottom image border, images smaller than one region (reflections of
# reflections), one-pixel rows and columns, minD > 0, numD not a multiple of the 8 slices a workgroup interleaves
SHAPES = [(8, 8, 0, 8), (9, 17, 0, 5), (16, 40, 1, 9), (23, 61, 0, 13), (1, 30, 0, 4), (30, 1, 0, 3), (5, 5, 2, 7), (33, 75, 3, 17)]


@pytest.mark.parametrize("H,W,minD,numD", SHAPES)
def test_tile_form_matches_oracle_and_per_pixel_sort(ctx, oracle, H, W, minD, numD):
    L, R, _ = make_pair(H, W, max(2, min(numD, W) // 2), seed=H * 131 + W, block=8)
    run = lambda c: c.computeAdaptiveWeight_WeightedMedian(L, R, LEFT, 15, 10, 10, minD, numD, return_cost_volume=True)
    d, v = run(ctx)
    rc, dw, vw = oracle.asw_wmedian(L, R, 0, 15, 10, 10, minD, numD, want_vol=True)
    assert rc == 0 and v.shape == vw.shape == (numD, H, W)
    assert np.array_equal(v, vw), np.argwhere(v != vw)[:5]
    assert np.array_equal(d, dw)
    d0, v0 = _with_env("ASW_WMEDIAN_TILE", "0", run)
    assert np.array_equal(v, v0) and np.array_equal(d, d0)


def test_tile_form_ties_and_flat_images(ctx, oracle):
    # identical images: every cost plane is full of equal costs, the multimap's insertion order decides every crossing;
    # constant images: all 225 costs AND weights equal
    L, _, _ = make_pair(20, 44, 4, seed=8, block=8)
    for L_, R_ in ((L, L.copy()), (np.full((12, 20, 3), 90, np.uint8), np.full((12, 20, 3), 90, np.uint8)),
                   ((L // 64) * 64, (np.roll(L, 2, axis=1) // 64) * 64)):
        d, v = ctx.computeAdaptiveWeight_WeightedMedian(L_, R_, LEFT, 15, 10, 10, 0, 6, return_cost_volume=True)
        rc, dw, vw = oracle.asw_wmedian(L_, R_, 0, 15, 10, 10, 0, 6, want_vol=True)
        assert rc == 0 and np.array_equal(v, vw) and np.array_equal(d, dw)


def test_tile_form_slice_chunks(ctx, oracle):
    # the slices are processed in chunks (sorted lists <= 2 GiB): chunk sizes that leave a short last chunk and wavefronts
    # without a slice
    L, R, _ = make_pair(19, 52, 10, seed=77, block=8)
    rc, dw, vw = oracle.asw_wmedian(L, R, 0, 15, 10, 10, 0, 21, want_vol=True)
    for chunk in ("1", "3", "8", "16", "100"):
        d, v = _with_env("ASW_WMEDIAN_TILE_CHUNK", chunk,
                         lambda c: c.computeAdaptiveWeight_WeightedMedian(L, R, LEFT, 15, 10, 10, 0, 21, return_cost_volume=True))
        assert np.array_equal(v, vw) and np.array_equal(d, dw), chunk


def test_tile_form_selector_and_mid_size(ctx, oracle):
    L, R, _ = make_pair(48, 160, 24, seed=5)
    assert np.array_equal(ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT_MEDIAN, 15, 0, 24), oracle.stereo_matching(L, R, 0, 10, 15, 0, 24)[1])
    # 188 x 621, D = 64: the two GPU forms check each other (the oracle needs ~10 s for it; the whole-frame C4 test has it)
    L, R, _ = make_pair(188, 621, 64, seed=6)
    run = lambda c: c.computeAdaptiveWeight_WeightedMedian(L, R, LEFT, 15, 10, 10, 0, 64, return_cost_volume=True)
    d, v = run(ctx)
    d0, v0 = _with_env("ASW_WMEDIAN_TILE", "0", run)
    assert np.array_equal(v, v0) and np.array_equal(d, d0)
