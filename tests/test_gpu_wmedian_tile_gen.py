"""The weighted median at windows 3x3 .. 13x13 and 17x17 .. 37x37 in its tile form (k_wmedian_tile_gen.hip: the (8 + win - 1)^2 neighbourhood of an
8x8 pixel block is sorted once per slice, every pixel walks the sorted list; the header's default window for this method is 35,
aswMethods.h:179-182) against the oracle (M.cpp:3228-3383) and against the per-pixel sort (k_wmedian_big, ASW_WMEDIAN_TILE=0).
The aggregated volume is one of the input costs, so every comparison is bit for bit.  Windows above 37 keep the per-pixel sort."""
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


@pytest.fixture(scope="module")
def ctx_sort():
    """per-pixel sort for every window (the library reads its switches once, in asw_create)"""
    c = asw.Context(0, env={"ASW_WMEDIAN_TILE": "0"})
    yield c
    c.close()


def _with_env(env, fn):
    c = asw.Context(0, env=env)
    try:
        return fn(c)
    finally:
        c.close()


# (H, W, win, minD, numD): all four list sizes (256 slots up to 9x9, 512 up to 13x13, 1024 up to 25x25, 2048 above), blocks cut by the right / bottom border,
# images smaller than the window (reflections of reflections), one-pixel rows and columns, minD > 0, slice counts that leave
# wavefronts without a slice
CASES = [(16, 24, 3, 0, 6), (9, 17, 5, 0, 5), (23, 61, 7, 1, 9), (12, 30, 9, 0, 4), (17, 41, 11, 0, 7), (8, 8, 13, 2, 5), (1, 30, 7, 0, 4),
         (30, 1, 9, 0, 3), (33, 75, 13, 3, 11), (16, 24, 17, 0, 6), (9, 17, 21, 0, 5), (23, 61, 25, 1, 9), (12, 30, 27, 0, 4), (17, 41, 35, 0, 7), (8, 8, 37, 2, 5),
         (1, 30, 19, 0, 4), (30, 1, 23, 0, 3), (33, 75, 35, 3, 11), (24, 50, 31, 0, 13)]


@pytest.mark.parametrize("H,W,win,minD,numD", CASES)
def test_general_tile_form_matches_oracle_and_per_pixel_sort(ctx, ctx_sort, oracle, H, W, win, minD, numD):
    L, R, _ = make_pair(H, W, max(2, min(numD, W) // 2), seed=H * 131 + W + win, block=8)
    run = lambda c: c.computeAdaptiveWeight_WeightedMedian(L, R, LEFT, win, 10, 10, minD, numD, return_cost_volume=True)
    d, v = run(ctx)
    rc, dw, vw = oracle.asw_wmedian(L, R, 0, win, 10, 10, minD, numD, want_vol=True)
    assert rc == 0 and v.shape == vw.shape == (numD, H, W)
    assert np.array_equal(v, vw), np.argwhere(v != vw)[:5]
    assert np.array_equal(d, dw)
    d0, v0 = run(ctx_sort)
    assert np.array_equal(v, v0) and np.array_equal(d, d0)


def test_general_tile_form_ties_and_flat_images(ctx, oracle):
    # identical images (every cost plane full of equal costs: the multimap's insertion order decides every crossing), constant
    # images (all costs AND weights equal), coarsely quantised images
    L, _, _ = make_pair(20, 44, 4, seed=8, block=8)
    for win in (5, 11, 21, 35):
        for L_, R_ in ((L, L.copy()), (np.full((12, 20, 3), 90, np.uint8), np.full((12, 20, 3), 90, np.uint8)),
                       ((L // 64) * 64, (np.roll(L, 2, axis=1) // 64) * 64)):
            d, v = ctx.computeAdaptiveWeight_WeightedMedian(L_, R_, LEFT, win, 10, 10, 0, 6, return_cost_volume=True)
            rc, dw, vw = oracle.asw_wmedian(L_, R_, 0, win, 10, 10, 0, 6, want_vol=True)
            assert rc == 0 and np.array_equal(v, vw) and np.array_equal(d, dw), win


def test_general_tile_form_chunks_and_rows_per_workgroup(ctx, oracle):
    # slices in chunks (sorted lists <= 2 GiB) that leave a short last chunk; every legal number of block rows per workgroup
    L, R, _ = make_pair(19, 52, 10, seed=77, block=8)
    for win in (7, 19, 29):
        rc, dw, vw = oracle.asw_wmedian(L, R, 0, win, 10, 10, 0, 13, want_vol=True)
        run = lambda c: c.computeAdaptiveWeight_WeightedMedian(L, R, LEFT, win, 10, 10, 0, 13, return_cost_volume=True)
        for env in ({"ASW_WMEDIAN_TILE_CHUNK": "1"}, {"ASW_WMEDIAN_TILE_CHUNK": "5"}, {"ASW_WMEDIAN_TILE_CHUNK": "100"},
                    {"ASW_WMEDIAN_GEN_ROWS": "1"}, {"ASW_WMEDIAN_GEN_ROWS": "2"}, {"ASW_WMEDIAN_GEN_ROWS": "4"}) + \
                (({"ASW_WMEDIAN_GEN_ROWS": "8"},) if win <= 19 else ()):
            d, v = _with_env(env, run)
            assert np.array_equal(v, vw) and np.array_equal(d, dw), (win, env)


def test_general_tile_form_mid_size_and_selector(ctx, ctx_sort, oracle):
    L, R, _ = make_pair(40, 120, 16, seed=5)
    assert np.array_equal(ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT_MEDIAN, 21, 0, 16), oracle.stereo_matching(L, R, 0, 10, 21, 0, 16)[1])
    # 94 x 311, D = 24 at the header's default window: the two GPU forms check each other
    L, R, _ = make_pair(94, 311, 24, seed=6)
    run = lambda c: c.computeAdaptiveWeight_WeightedMedian(L, R, LEFT, 35, 10, 10, 0, 24, return_cost_volume=True)
    d, v = run(ctx)
    d0, v0 = run(ctx_sort)
    assert np.array_equal(v, v0) and np.array_equal(d, d0)


def test_windows_above_37_keep_the_per_pixel_sort(ctx, ctx_sort, oracle):
    L, R, _ = make_pair(12, 26, 4, seed=9, block=8)
    run = lambda c: c.computeAdaptiveWeight_WeightedMedian(L, R, LEFT, 39, 10, 10, 0, 4, return_cost_volume=True)
    d, v = run(ctx)
    rc, dw, vw = oracle.asw_wmedian(L, R, 0, 39, 10, 10, 0, 4, want_vol=True)
    assert rc == 0 and np.array_equal(v, vw) and np.array_equal(d, dw)
    d0, v0 = run(ctx_sort)
    assert np.array_equal(v, v0)
