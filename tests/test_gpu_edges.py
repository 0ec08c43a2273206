"""Edge cases the reference's loops admit: tiny and ragged images, windows larger than the image, disparity
ranges larger than the width, minDisparity > 0, 1-pixel windows, constant images."""
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


def _pair(H, W, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, (H, W, 3)).astype(np.uint8), rng.integers(0, 256, (H, W, 3)).astype(np.uint8))


SHAPES = [(1, 1), (1, 7), (5, 1), (3, 5), (2, 65), (7, 63), (4, 64), (5, 129), (9, 200)]


@pytest.mark.parametrize("H,W", SHAPES)
def test_classic_small_shapes(ctx, oracle, H, W):
    L, R = _pair(H, W, H * 131 + W)
    for win, minD, numD, dt in [(3, 0, 4, 0), (7, 1, 9, 0), (15, 0, 3, 1), (5, 2, 70, 0)]:
        rc, dw, vw = oracle.asw_classic(L, R, 30, 20, dt, win, minD, numD, want_vol=True)
        d, v = ctx.computeAdaptiveWeight(L, R, 30, 20, dt, win, minD, numD, return_cost_volume=True)
        assert rc == 0 and np.array_equal(v, vw, equal_nan=True) and np.array_equal(d, dw), (win, minD, numD, dt)


def test_classic_window_one(ctx, oracle):
    # win = 1: zero taps -> E = 0/0 = NaN for every d -> never selected -> 0 (reference: uninitialised)
    L, R = _pair(6, 10, 1)
    rc, dw, vw = oracle.asw_classic(L, R, 30, 20, 0, 1, 0, 3, want_vol=True)
    d, v = ctx.computeAdaptiveWeight(L, R, 30, 20, LEFT, 1, 0, 3, return_cost_volume=True)
    assert np.isnan(vw).all() and np.isnan(v).all() and (d == 0).all() and np.array_equal(d, dw)


@pytest.mark.parametrize("H,W", SHAPES)
def test_geodesic_small_shapes(ctx, oracle, H, W):
    L, R = _pair(H, W, H * 17 + W)
    for win, minD, numD, dt in [(3, 0, 4, 0), (7, 1, 9, 1), (15, 0, 3, 0), (1, 0, 5, 0)]:
        rc, dw, vw = oracle.asw_geodesic(L, R, dt, win, minD, numD, want_vol=True)
        d, v = ctx.computeAdaptiveWeight_geodesic(L, R, dt, win, minD, numD, return_cost_volume=True)
        assert rc == 0 and np.array_equal(v, vw, equal_nan=True) and np.array_equal(d, dw), (win, minD, numD, dt)
    rc, ww = oracle.geodesic_dist(L, 5, 3)
    assert np.array_equal(ctx.getGeodesicDist(L, 5, 3), ww)


@pytest.mark.parametrize("H,W", SHAPES)
def test_costs_small_shapes(ctx, oracle, H, W):
    L, R = _pair(H, W, H * 7 + W)
    for dt in (0, 1):
        assert np.array_equal(np.stack(ctx.computeAD(L, R, dt, 1, 6)), oracle.compute_ad(L, R, dt, 1, 6)[1])
        assert np.array_equal(np.stack(ctx.getCostSAD(L, R, dt, 5, 0, 4)), oracle.cost_sad(L, R, dt, 5, 0, 4)[1])
    assert np.array_equal(np.stack(ctx.computeSimilarity(L, R, 0.4, 10, 50, LEFT, 2, 5)), oracle.compute_similarity(L, R, 0.4, 10, 50, 0, 2, 5)[1])
    assert np.array_equal(np.stack(ctx.computeSimilarity(L, R, 0.3, 7, 40, LEFT, 0, 3, winSize=5)),
                          oracle.compute_similarity(L, R, 0.3, 7, 40, 0, 0, 3, win=5)[1])


@pytest.mark.parametrize("H,W", [(1, 1), (3, 5), (2, 65), (7, 63), (5, 129), (20, 130)])
def test_guided_and_median_small_shapes(ctx, oracle, H, W):
    L, R = _pair(H, W, H * 3 + W)
    for win in (3, 5, 15):
        rc, dw, vw = oracle.asw_guided2(L, R, 0, 1e-6, win, 1, 5, want_vol=True)
        d, v = ctx.computeAdaptiveWeight_GuidedF_2(L, R, LEFT, 1e-6, win, 1, 5, return_cost_volume=True)
        assert rc == 0 and np.abs(v - vw).max() < 1e-4 and np.array_equal(d, dw), ("g2", win)
        for dt in (0, 1):
            rc, dw, vw = oracle.asw_guided(L, R, dt, 1e-6, win, 0, 4, want_vol=True)
            d, v = ctx.computeAdaptiveWeight_GuidedF(L, R, dt, 1e-6, win, 0, 4, return_cost_volume=True)
            assert rc == 0 and np.abs(v - vw).max() < 1e-4 and np.array_equal(d, dw), ("g", win, dt)
        rc, dw, vw = oracle.asw_wmedian(L, R, 0, win, 10, 10, 1, 5, want_vol=True)
        d, v = ctx.computeAdaptiveWeight_WeightedMedian(L, R, LEFT, win, 10, 10, 1, 5, return_cost_volume=True)
        assert rc == 0 and np.array_equal(v, vw) and np.array_equal(d, dw), ("wm", win)


@pytest.mark.parametrize("H,W,win,minD,numD,seed", [(12, 30, 17, 0, 5, 1), (20, 44, 21, 1, 6, 2), (16, 40, 23, 0, 4, 3), (14, 36, 35, 0, 3, 4),
                                                     (10, 24, 45, 0, 3, 5)])
def test_weighted_median_large_windows(ctx, oracle, H, W, win, minD, numD, seed):
    # windows above 15x15 take the general path (512 / 1024 / 2048 slots, 64-bit keys): same stable order, same pick
    L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=8)
    rc, dw, vw = oracle.asw_wmedian(L, R, 0, win, 10, 10, minD, numD, want_vol=True)
    d, v = ctx.computeAdaptiveWeight_WeightedMedian(L, R, LEFT, win, 10, 10, minD, numD, return_cost_volume=True)
    assert rc == 0 and np.array_equal(v, vw) and np.array_equal(d, dw)


def test_even_guided2_window_is_accepted_like_the_reference(ctx, oracle):
    # GuidedF_2 has no parity check on winSize: boxFilter(Size(6,6)) uses anchor 3 (M.cpp:2976-3006)
    L, R, _ = make_pair(20, 40, 6, seed=3, block=8)
    rc, dw, vw = oracle.asw_guided2(L, R, 0, 1e-6, 6, 0, 6, want_vol=True)
    d, v = ctx.computeAdaptiveWeight_GuidedF_2(L, R, LEFT, 1e-6, 6, 0, 6, return_cost_volume=True)
    assert rc == 0 and np.abs(v - vw).max() < 1e-4 and np.array_equal(d, dw)


def test_constant_images(ctx, oracle):
    L = np.full((10, 20, 3), 9, np.uint8)
    R = L.copy()
    for alg, fn in [(A.ADAPTIVE_WEIGHT, lambda: oracle.stereo_matching(L, R, 0, 2, 5, 0, 4)),
                    (A.ADAPTIVE_WEIGHT_GUIDED_FILTER_2, lambda: oracle.stereo_matching(L, R, 0, 8, 5, 0, 4)),
                    (A.ADAPTIVE_WEIGHT_GUIDED_FILTER, lambda: oracle.stereo_matching(L, R, 0, 7, 5, 0, 4)),
                    (A.ADAPTIVE_WEIGHT_MEDIAN, lambda: oracle.stereo_matching(L, R, 0, 10, 5, 0, 4))]:
        rc, want = fn()
        assert rc == 0 and np.array_equal(ctx.stereoMatching(L, R, LEFT, alg, 5, 0, 4), want), alg


def test_bad_arguments(ctx):
    L, R = _pair(8, 8, 1)
    for bad in [dict(numDisparity=0), dict(minDisparity=-1)]:
        with pytest.raises(asw.AswError) as e:
            ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT, 5, bad.get("minDisparity", 0), bad.get("numDisparity", 4))
        assert e.value.status == asw.ERR_BAD_ARGUMENT
    with pytest.raises(TypeError):
        ctx.stereoMatching(L.astype(np.float32), R, LEFT, A.ADAPTIVE_WEIGHT, 5, 0, 4)
    with pytest.raises(asw.AswError):   # weighted median above the 2048-slot general path
        ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT_MEDIAN, 47, 0, 4)
    # non-contiguous (strided) inputs are honoured through asw_image.step
    big = np.zeros((8, 16, 3), np.uint8)
    big[:, ::2] = L
    d1 = ctx.stereoMatching(big[:, ::2], R, LEFT, A.ADAPTIVE_WEIGHT, 5, 0, 4)
    assert np.array_equal(d1, ctx.stereoMatching(L, R, LEFT, A.ADAPTIVE_WEIGHT, 5, 0, 4))


@pytest.mark.parametrize("win", [21, 35])
def test_large_windows(ctx, oracle, win):
    # config C2 uses a 35x35 window for the classic method; the other methods accept the same sizes
    L, R, _ = make_pair(44, 100, 10, seed=win, block=16)
    rc, dw, vw = oracle.asw_geodesic(L, R, 0, win, 0, 9, want_vol=True)
    d, v = ctx.computeAdaptiveWeight_geodesic(L, R, LEFT, win, 0, 9, return_cost_volume=True)
    assert rc == 0 and np.array_equal(v, vw, equal_nan=True) and np.array_equal(d, dw)
    rc, dw, vw = oracle.asw_guided2(L, R, 0, 1e-6, win, 0, 6, want_vol=True)
    d, v = ctx.computeAdaptiveWeight_GuidedF_2(L, R, LEFT, 1e-6, win, 0, 6, return_cost_volume=True)
    assert rc == 0 and np.abs(v - vw).max() < 1e-4 and np.array_equal(d, dw)
    rc, dw, vw = oracle.asw_blo1(L, R, 0, 0.015, win, 0, 5, want_vol=True)
    d, v = ctx.computeAdaptiveWeight_BLO1(L, R, LEFT, 0.015, win, 0, 5, return_cost_volume=True)
    fin = np.isfinite(vw)
    assert rc == 0 and np.allclose(v[fin], vw[fin], rtol=1e-4, atol=0) and np.array_equal(d, dw)
    rc, ww = oracle.geodesic_dist(L[:20, :30], win, 3)
    assert np.array_equal(ctx.getGeodesicDist(L[:20, :30], win, 3), ww)


def test_one_context_per_thread_runs_concurrently(ctx):
    """SURVEY 8b threading: one ctx per thread, never shared.  Four host threads, each with its own context and its own
    method, interleave on the device; every result equals the single-threaded one."""
    import threading

    L, R, _ = make_pair(72, 160, 16, seed=33)
    algs = [A.ADAPTIVE_WEIGHT, A.ADAPTIVE_WEIGHT_GUIDED_FILTER_2, A.ADAPTIVE_WEIGHT_GEODESIC, A.ADAPTIVE_WEIGHT_BILATERAL_GRID]
    want = [ctx.stereoMatching(L, R, LEFT, a, 9, 0, 16) for a in algs]
    got = [None] * len(algs)
    errs = []

    def work(i):
        try:
            c = asw.Context(0)
            for _ in range(3):
                got[i] = c.stereoMatching(L, R, LEFT, algs[i], 9, 0, 16)
            c.close()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(algs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_batch_on_a_missing_device_fails_cleanly():
    L, R, _ = make_pair(16, 32, 4, seed=2)
    with pytest.raises(asw.AswError) as e:
        asw.stereoMatchingBatch([L], [R], LEFT, A.ADAPTIVE_WEIGHT, 5, 0, 4, device_ids=[99])
    assert e.value.status in (asw.ERR_HIP, asw.ERR_BAD_ARGUMENT)
    # and the scheduler still works afterwards
    out = asw.stereoMatchingBatch([L], [R], LEFT, A.ADAPTIVE_WEIGHT, 5, 0, 4, device_ids=[0])
    assert out[0].shape == (16, 32)


def test_window_one_on_a_fresh_context(oracle):
    """win = 1 has no taps at all; on a context whose tap table was never allocated the kernel once read it anyway
    (GPU memory fault found by tools/fuzz_parity.py --seed 99: direct8, 14 x 344, win 1 as the first bilateral call)."""
    L, R, _ = make_pair(14, 344, 12, seed=587025979, block=8)
    for which in ("direct8", "classic"):
        c = asw.Context(0)
        if which == "direct8":
            d, v = c.computeAdaptiveWeight_direct8(L, R, LEFT, 1, 0, 12, return_cost_volume=True)
            rc, dw, vw = oracle.asw_direct8(L, R, 0, 1, 0, 12, want_vol=True)
        else:
            d, v = c.computeAdaptiveWeight(L, R, 30, 20, LEFT, 1, 0, 12, return_cost_volume=True)
            rc, dw, vw = oracle.asw_classic(L, R, 30, 20, 0, 1, 0, 12, want_vol=True)
        c.close()
        assert rc == 0 and np.array_equal(d, dw) and np.array_equal(v, vw, equal_nan=True)


def test_nan_costs_stay_inside_their_windows(ctx, oracle):
    """Flat windows give 0/0 NCC costs (M.cpp:867-868).  A NaN must poison exactly the box windows that contain it (the CPU
    restatement's definition) -- not everything below it in the band or the next column of a lane pair, which is what plain
    sliding sums do.  Found by tools/fuzz_parity.py --fresh-every 1 --seed 123 (guided3, 1 x 247)."""
    L, R, _ = make_pair(40, 200, 12, seed=5, block=8)
    L[10:16, 50:60] = 77        # flat patches: every 3x3 / 5x5 window inside is constant -> NCC = 0/0
    R[10:16, 40:52] = 91
    L[30:33, 120:126] = 5
    for win, dt in ((3, LEFT), (5, LEFT), (3, RIGHT)):
        rc, dw, vw = oracle.asw_guided3(L, R, dt, 1e-6, win, 0, 12, want_vol=True)
        d, v = ctx.computeAdaptiveWeight_GuidedF_3(L, R, dt, 1e-6, win, 0, 12, return_cost_volume=True)
        assert rc == 0 and np.isnan(vw).any() and np.isfinite(vw).mean() > 0.5
        assert np.array_equal(np.isnan(v), np.isnan(vw)), (win, dt)
        fin = np.isfinite(vw)
        assert np.allclose(v[fin], vw[fin], rtol=1e-4, atol=1e-30)
        ok = fin.all(axis=0)
        assert np.array_equal(d[ok], dw[ok])
    # the public guided filter with NaN in the caller's P
    rng = np.random.default_rng(8)
    P = rng.random((40, 200), dtype=np.float32)
    P[7, 9] = np.nan
    P[20:22, 100:103] = np.nan
    for guide in (L, np.concatenate([L, R], axis=2)):
        rc, qw = oracle.guided_filter(guide, P, 7, 1e-6)
        q = ctx.getGuidedFilter(guide, P, 7, 1e-6)
        assert rc == 0 and np.array_equal(np.isnan(q), np.isnan(qw)) and 0 < np.isnan(qw).mean() < 0.2
        assert np.allclose(q[np.isfinite(qw)], qw[np.isfinite(qw)], rtol=1e-4, atol=1e-30)


@pytest.mark.parametrize("win", [33, 35])
def test_six_channel_guided_filter_with_the_reference_default_window(ctx, oracle, win):
    """M.h:166-176 declares winSize = 35 as the default of the GuidedF family; the 7-plane pass of the 6-channel guide once
    rejected boxes wider than 32 (found by the fuzz sweep with large windows)."""
    L, R, _ = make_pair(40, 120, 10, seed=win)
    rc, dw, vw = oracle.asw_guided(L, R, 0, 1e-6, win, 0, 10, want_vol=True)
    d, v = ctx.computeAdaptiveWeight_GuidedF(L, R, LEFT, 1e-6, win, 0, 10, return_cost_volume=True)
    assert rc == 0 and np.allclose(v, vw, rtol=1e-4, atol=1e-30) and np.array_equal(d, dw)
    rc, dw, vw = oracle.asw_guided3(L, R, 0, 1e-6, win, 0, 10, want_vol=True)
    d, v = ctx.computeAdaptiveWeight_GuidedF_3(L, R, LEFT, 1e-6, win, 0, 10, return_cost_volume=True)
    fin = np.isfinite(vw)
    assert rc == 0 and np.array_equal(np.isnan(v), np.isnan(vw)) and np.allclose(v[fin], vw[fin], rtol=1e-4, atol=1e-30)
    P = np.random.default_rng(win).random((40, 120), dtype=np.float32)
    guide = np.concatenate([L, R], axis=2)
    assert np.allclose(ctx.getGuidedFilter(guide, P, win, 1e-6), oracle.guided_filter(guide, P, win, 1e-6)[1], rtol=1e-4, atol=1e-30)


# ---------------------------------------------------------------- resident slots: no stale results
def test_downloads_never_return_another_frames_result(ctx, oracle):
    """asw_upload_pair into a slot drops the slot's previous disparity and volume: before the fix a download right after
    uploading a LARGER pair copied rows*cols*4 bytes out of the old, smaller allocation (out-of-bounds device read), and
    with an equal size it silently returned the previous frame's map."""
    small = make_pair(20, 40, 6, seed=1, block=8)[:2]
    large = make_pair(60, 130, 6, seed=2, block=8)[:2]
    same = make_pair(20, 40, 6, seed=3, block=8)[:2]
    ctx.upload_pair(11, *small)
    ctx.match_resident(11, LEFT, A.ADAPTIVE_WEIGHT, 5, 0, 6, keep_volume=True)
    d_small = ctx.download_disparity(11, (20, 40))
    assert ctx.download_volume(11, (7, 20, 40)).shape == (7, 20, 40)
    for nxt, shape in ((large, (60, 130)), (same, (20, 40))):
        ctx.upload_pair(11, *nxt)
        for call in (lambda: ctx.download_disparity(11, shape), lambda: ctx.download_disparity_u8(11, shape),
                     lambda: ctx.download_volume(11, (7,) + shape)):
            with pytest.raises(asw.AswError) as e:
                call()
            assert e.value.status == asw.ERR_NO_FRAME
        ctx.match_resident(11, LEFT, A.ADAPTIVE_WEIGHT, 5, 0, 6)
        rc, want, _ = oracle.asw_classic(nxt[0], nxt[1], 30, 20, 0, 5, 0, 6)
        assert rc == 0 and np.array_equal(ctx.download_disparity(11, shape), want)
        with pytest.raises(asw.AswError):  # this match kept no volume: the older one must not come back
            ctx.download_volume(11, (7,) + shape)
    assert not np.array_equal(d_small, ctx.download_disparity(11, (20, 40)))
    # a failed match (even window) leaves no result behind either, and the resident API raises instead of returning quietly
    with pytest.raises(asw.AswError) as e:
        ctx.match_resident(11, LEFT, A.ADAPTIVE_WEIGHT_GEODESIC, 6, 0, 6)
    assert e.value.status == asw.ERR_EVEN_WINDOW
    with pytest.raises(asw.AswError) as e:
        ctx.download_disparity(11, (20, 40))
    assert e.value.status == asw.ERR_NO_FRAME
    # a rejected upload empties the slot: nothing of the previous pair can be matched by accident
    with pytest.raises(asw.AswError) as e:
        ctx.upload_pair(11, same[0], same[1][:, :30])
    assert e.value.status == asw.ERR_SIZE_MISMATCH
    with pytest.raises(asw.AswError) as e:
        ctx.match_resident(11, LEFT, A.ADAPTIVE_WEIGHT, 5, 0, 6)
    assert e.value.status == asw.ERR_NO_FRAME


def test_host_buffer_entry_points_leave_resident_slots_alone(ctx, oracle):
    """asw_stereo_match & co. work on a private frame: upload_pair(0, A); stereoMatching(B); match_resident(0) computes on A."""
    A_pair = make_pair(24, 50, 6, seed=21, block=8)[:2]
    B_pair = make_pair(31, 77, 6, seed=22, block=8)[:2]
    ctx.upload_pair(0, *A_pair)
    dB = ctx.stereoMatching(B_pair[0], B_pair[1], LEFT, A.ADAPTIVE_WEIGHT, 5, 0, 6)
    ctx.computeAdaptiveWeight_GuidedF_2(B_pair[0], B_pair[1], LEFT, 1e-6, 5, 0, 6)
    ctx.match_resident(0, LEFT, A.ADAPTIVE_WEIGHT, 5, 0, 6)
    dA = ctx.download_disparity(0, (24, 50))
    assert np.array_equal(dA, oracle.asw_classic(A_pair[0], A_pair[1], 30, 20, 0, 5, 0, 6)[1])
    assert np.array_equal(dB, oracle.asw_classic(B_pair[0], B_pair[1], 30, 20, 0, 5, 0, 6)[1])


def test_short_cost_volume_buffer_is_refused(ctx):
    """cost_volume_floats: the C caller states its capacity; one plane short (the inclusive-range trap) is ASW_ERR_BAD_ARGUMENT."""
    import ctypes as C

    from aswstereomatch_amd import _image

    L, R = make_pair(12, 20, 4, seed=5, block=8)[:2]
    li, la = _image(L)
    ri, ra = _image(R)
    disp = np.zeros((12, 20), np.float32)
    di, _ = _image(disp, 5)
    guard = np.full(5 * 12 * 20 + 64, 7.0, np.float32)      # ADAPTIVE_WEIGHT with numD = 4 writes 5 planes
    pv = guard.ctypes.data_as(C.c_void_p)
    rc = ctx._lib.asw_stereo_match(ctx._h, C.byref(li), C.byref(ri), C.byref(di), 0, 2, 5, 0, 4, pv, 4 * 12 * 20)
    assert rc == asw.ERR_BAD_ARGUMENT and (guard == 7.0).all()
    rc = ctx._lib.asw_stereo_match(ctx._h, C.byref(li), C.byref(ri), C.byref(di), 0, 2, 5, 0, 4, pv, 5 * 12 * 20)
    assert rc == 0 and (guard[5 * 12 * 20:] == 7.0).all() and not (guard[:5 * 12 * 20] == 7.0).all()


def test_padded_host_rows_in_and_out(ctx, oracle):
    # asw_image.step > cols * channels on every host buffer (a cv::Mat ROI): rows are packed / unpacked on the host and travel as
    # one dense copy (asw_context.hip: copy_rows); widths whose rows are no multiple of 64 bytes are the case that matters
    import ctypes as C
    from aswstereomatch_amd import _image
    H, W = 9, 131
    L, R, _ = make_pair(H, W, 6, seed=321, block=8)
    Lp = np.full((H, W + 5, 3), 7, np.uint8)
    Rp = np.full((H, W + 9, 3), 200, np.uint8)
    Lp[:, :W], Rp[:, :W] = L, R
    want = oracle.asw_classic(L, R, 30, 20, 0, 7, 0, 6)[1]
    got = ctx.stereoMatching(Lp[:, :W], Rp[:, :W], LEFT, A.ADAPTIVE_WEIGHT, 7, 0, 6)   # padded inputs, dense output
    assert np.array_equal(got, want)
    ctx.upload_pair(3, Lp[:, :W], Rp[:, :W])
    ctx.match_resident(3, LEFT, A.ADAPTIVE_WEIGHT, 7, 0, 6, keep_volume=False)
    big = np.full((H, W + 3), -5.0, np.float32)
    di, _keep = _image(big[:, :W], 5)                      # padded float output
    assert di.step == (W + 3) * 4
    assert ctx._lib.asw_download_disparity(ctx._h, 3, C.byref(di)) == 0
    assert np.array_equal(big[:, :W], want) and (big[:, W:] == -5.0).all()
    u8big = np.full((H, W + 2), 9, np.uint8)
    ui, _keep2 = _image(u8big[:, :W])
    assert ctx._lib.asw_download_disparity_u8(ctx._h, 3, C.byref(ui), 0) == 0
    assert np.array_equal(u8big[:, :W], oracle.disparity_to_u8(want, False)) and (u8big[:, W:] == 9).all()


# ---------------------------------------------------------------- the library reads its switches once, in asw_create
def test_switches_are_read_once_at_context_creation(oracle):
    """ASW_BILATERAL_XQ & co. are measurement / test switches: asw_create reads them, no call does.  A context
    created under a switch keeps it after the environment changed back, and a context created without it ignores a later setenv;
    the results are the same either way (both kernel forms are bit-identical), only the launch count tells them apart."""
    import os

    L, R, _ = make_pair(6, 200, 100, seed=4, block=16)
    rc, want, _ = oracle.asw_classic(L, R, 30, 20, 0, 15, 0, 128)
    forced = asw.Context(0, env={"ASW_BILATERAL_XQ": "0"})       # the variable exists only while asw_create runs
    assert "ASW_BILATERAL_XQ" not in os.environ
    plain = asw.Context(0)
    os.environ["ASW_BILATERAL_XQ"] = "0"                          # too late for `plain`
    try:
        for c, launches in ((forced, 1), (plain, 4)):
            assert np.array_equal(c.computeAdaptiveWeight(L, R, 30, 20, LEFT, 15, 0, 128), want)
            assert c.timing()["aggregate_launches"] == launches
    finally:
        del os.environ["ASW_BILATERAL_XQ"]
        forced.close()
        plain.close()


def test_guided2_fused_walk_equals_two_pass_path(oracle):
    """ASW_GUIDED_FUSED=1 (k_guided_fused3: one walk does both box stages, no a/b volume; not the default -- slower) against the
    two-pass path: bit for bit, image borders included (a/b at a virtual row / column is evaluated on the mirrored window), and
    against the oracle within the volume tolerance."""
    base = asw.Context(0)
    fused = asw.Context(0, env={"ASW_GUIDED_FUSED": "1"})
    try:
        for (H, W, D, seed) in ((16, 40, 5, 1), (37, 130, 9, 2), (75, 231, 12, 3), (20, 301, 4, 4)):
            L, R, _ = make_pair(H, W, D, seed=seed)
            d0, v0 = base.computeAdaptiveWeight_GuidedF_2(L, R, LEFT, 1e-6, 15, 0, D, return_cost_volume=True)
            d1, v1 = fused.computeAdaptiveWeight_GuidedF_2(L, R, LEFT, 1e-6, 15, 0, D, return_cost_volume=True)
            assert np.array_equal(d0, d1) and np.allclose(v0, v1, rtol=1e-6, atol=0)
            rc, dw, vw = oracle.asw_guided2(L, R, 0, 1e-6, 15, 0, D, want_vol=True)
            assert rc == 0 and np.allclose(v1, vw, rtol=1e-4, atol=1e-6) and np.array_equal(d1, dw)
    finally:
        base.close()
        fused.close()


def test_guided6_pair_kernels_equal_box_walk_passes(oracle):
    """GuidedF at 15x15: the wavefront-pair kernels (k_ab6_pair / k_q6_pair: one wavefront per guide word, the dot products handed
    over through LDS in the reference's left-to-right order) against the k_box_walk passes (ASW_AB6_PAIR=0 ASW_Q6_PAIR=0): bit for
    bit in both directions, each pass switched separately, and against the oracle within the volume tolerance."""
    new = asw.Context(0)
    olds = [asw.Context(0, env=e) for e in ({"ASW_AB6_PAIR": "0", "ASW_Q6_PAIR": "0"}, {"ASW_AB6_PAIR": "0"}, {"ASW_Q6_PAIR": "0"})]
    try:
        for (H, W, D, dt, seed) in ((16, 40, 5, LEFT, 1), (37, 130, 9, RIGHT, 2), (75, 231, 12, LEFT, 3), (20, 301, 4, RIGHT, 4), (3, 17, 2, LEFT, 5)):
            L, R, _ = make_pair(H, W, D, seed=seed)
            d1, v1 = new.computeAdaptiveWeight_GuidedF(L, R, dt, 1e-6, 15, 0, D, return_cost_volume=True)
            for c in olds:
                d0, v0 = c.computeAdaptiveWeight_GuidedF(L, R, dt, 1e-6, 15, 0, D, return_cost_volume=True)
                assert np.array_equal(d0, d1) and np.array_equal(v0, v1)
            rc, dw, vw = oracle.asw_guided(L, R, int(dt), 1e-6, 15, 0, D, want_vol=True)
            assert rc == 0 and np.abs(v1 - vw).max() < 1e-4 and np.array_equal(d1, dw)
    finally:
        new.close()
        for c in olds:
            c.close()
