"""Randomised GPU-vs-oracle parity sweep (not part of the pytest suite: run it on the GPU box when kernels change).

    python tools/fuzz_parity.py --seconds 120 --seed 1

Every method of the selector is run through the C-ABI on random shapes / windows / disparity ranges / directions and
compared with the CPU oracle: WTA index bit-exact everywhere; cost volume bit-exact for the methods whose summation
order the kernels reproduce, within 1e-4 (relative) for the guided-filter family.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aswstereomatch_amd as asw  # noqa: E402
from aswstereomatch_amd.synth import make_pair  # noqa: E402
from oracle import asw_oracle as O  # noqa: E402

A = asw.StereoMatchingAlgorithms


def close(a, b):
    fin = np.isfinite(b)
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.allclose(a[fin], b[fin], rtol=1e-4, atol=1e-30)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-h", type=int, default=70)
    ap.add_argument("--max-w", type=int, default=260)
    ap.add_argument("--fresh-every", type=int, default=300, help="recreate the context every N cases: first-use paths (unallocated tables, scratch growth) get exercised in many orders")
    ap.add_argument("--flat", type=float, default=0.25, help="share of the cases whose images get constant rectangles (flat windows: 0/0 NCC costs, zero variances, ties)")
    ap.add_argument("--trace", action="store_true", help="print every case before it runs (to identify a faulting one)")
    ap.add_argument("--xq", type=float, default=0.0, help="share of the cases forced into the domain of the xq kernels (classic / geodesic, "
                    "both directions, win 15, 63..300 candidates, widths 64..420 incl. partial and border tiles)")
    ap.add_argument("--only", default="", help="comma-separated method names: every case is drawn from these only (e.g. guided2 with ASW_GUIDED_FUSED=1 in the environment)")
    ap.add_argument("--wmbig", type=float, default=0.0, help="share of the cases forced to the weighted median at 3x3 .. 13x13 and 17x17 .. 41x41 (general tile form, k_wmedian_tile_gen.hip; per-pixel sort above 37)")
    ap.add_argument("--wm15", type=float, default=0.0, help="share of the cases forced to the 15x15 weighted median (the tile form, k_wmedian_tile.hip)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = asw.Context(0)
    O.set_threads(O.usable_cores())
    t0 = time.time()
    n = fails = 0
    counts = {}
    last = t0
    while time.time() - t0 < args.seconds:
        H = int(rng.integers(1, args.max_h))
        W = int(rng.integers(1, args.max_w))
        win = int(rng.choice([1, 3, 5, 7, 9, 11, 15, 17, 21, 25, 33, 35]))
        minD = int(rng.choice([0, 0, 0, 1, 3, 17]))
        numD = int(rng.integers(1, 48))
        if rng.random() < 0.08:  # long candidate ranges cross the 16-wide chunk and z-split boundaries
            numD = int(rng.integers(48, 150))
            H, W = min(H, 24), min(W, 96)
        dt = int(rng.integers(0, 2))
        if win > 21:  # big windows: keep the CPU side in the millisecond range
            H, W, numD = min(H, 40), min(W, 120), min(numD, 12)
        seed = int(rng.integers(0, 1 << 30))
        L, R, _ = make_pair(H, W, max(2, numD // 2), seed=seed, block=int(rng.choice([4, 8, 16])))
        if rng.random() < args.flat:
            for img in (L, R):
                for _ in range(int(rng.integers(1, 4))):
                    y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
                    img[y0:y0 + int(rng.integers(1, 12)), x0:x0 + int(rng.integers(1, 40))] = rng.integers(0, 256, 3).astype(np.uint8)
        if rng.random() < 0.15:  # rows with padding: the C-ABI reads them through asw_image.step
            pad = int(rng.integers(1, 9))
            Lp = np.zeros((H, W + pad, 3), np.uint8)
            Rp = np.full((H, W + pad, 3), 255, np.uint8)
            Lp[:, :W], Rp[:, :W] = L, R
            L, R = Lp[:, :W], Rp[:, :W]   # non-contiguous views; the oracle wrappers copy them
        method = str(rng.choice(["classic", "direct8", "geodesic", "guided", "guided2", "guided3", "wmedian", "blo1", "ncc", "ncc_cost",
                                "ad_tad", "similarity", "sad", "geodist", "gfilter", "prep", "bilgrid", "lrcheck", "resident", "batch"]))
        if args.only:
            method = str(rng.choice(args.only.split(",")))
        if rng.random() < args.xq:
            method = str(rng.choice(["classic", "geodesic"]))
            H, W = int(rng.integers(1, 10)), int(rng.integers(64, 420))
            win, dt = 15, int(rng.integers(0, 2))   # both directions have an xq form
            minD = int(rng.choice([0, 0, 1, 5, 48, 49, 70])) if method == "classic" else int(rng.choice([0, 0, 2, 33, 130]))
            numD = int(rng.choice([63, 64, 65, 100, 126, 127, 128, 129, 191, 192, 255, int(rng.integers(63, 128)), int(rng.integers(63, 300))]))
            seed = int(rng.integers(0, 1 << 30))
            L, R, _ = make_pair(H, W, max(2, min(numD, W) // 2), seed=seed, block=int(rng.choice([4, 8, 16])))
            if rng.random() < 0.3:
                for img in (L, R):
                    y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
                    img[y0:y0 + int(rng.integers(1, 6)), x0:x0 + int(rng.integers(1, 60))] = rng.integers(0, 256, 3).astype(np.uint8)
        if rng.random() < args.wm15:
            method, win, dt = "wmedian", 15, 0
            numD = int(rng.integers(1, 40))
        if rng.random() < args.wmbig:
            method, win, dt = "wmedian", int(rng.choice([3, 5, 7, 9, 11, 13, 17, 19, 21, 23, 25, 27, 29, 31, 33, 35, 37, 39, 41])), 0
            H, W, numD = int(rng.integers(1, 36)), int(rng.integers(1, 90)), int(rng.integers(1, 20))
            minD = int(rng.choice([0, 0, 1, 4]))
            seed = int(rng.integers(0, 1 << 30))
            L, R, _ = make_pair(H, W, max(2, min(numD, W) // 2), seed=seed, block=int(rng.choice([4, 8, 16])))
            if rng.random() < 0.3:  # flat rectangles: ties
                for img in (L, R):
                    y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
                    img[y0:y0 + int(rng.integers(1, 12)), x0:x0 + int(rng.integers(1, 40))] = rng.integers(0, 256, 3).astype(np.uint8)
        tag = (method, H, W, win, minD, numD, dt, seed)
        if args.fresh_every > 0 and n > 0 and n % args.fresh_every == 0:
            ctx.close()
            ctx = asw.Context(0)
        if args.trace:
            print("case", n, tag, flush=True)
        try:
            if method == "classic":
                gc, gg = float(rng.choice([30, 5, 0.5, 100])), float(rng.choice([20, 2, 7.5, 60]))
                rc, dw, vw = O.asw_classic(L, R, gc, gg, dt, win, minD, numD, want_vol=True)
                d, v = ctx.computeAdaptiveWeight(L, R, gc, gg, dt, win, minD, numD, return_cost_volume=True)
                ok = np.array_equal(v, vw, equal_nan=True) and np.array_equal(d, dw)
            elif method == "direct8":
                rc, dw, vw = O.asw_direct8(L, R, 0, win, minD, numD, want_vol=True)
                d, v = ctx.computeAdaptiveWeight_direct8(L, R, 0, win, minD, numD, return_cost_volume=True)
                ok = np.array_equal(v, vw, equal_nan=True) and np.array_equal(d, dw)
            elif method == "geodesic":
                if win > 15:  # (win+2)^2 relaxations per pixel and pass on the CPU side: small frames
                    L, R = np.ascontiguousarray(L[:20, :64]), np.ascontiguousarray(R[:20, :64])
                    numD = min(numD, 8)
                rc, dw, vw = O.asw_geodesic(L, R, dt, win, minD, numD, want_vol=True)
                d, v = ctx.computeAdaptiveWeight_geodesic(L, R, dt, win, minD, numD, return_cost_volume=True)
                ok = np.array_equal(v, vw, equal_nan=True) and np.array_equal(d, dw)
            elif method == "guided":
                eps = float(rng.choice([1e-6, 1e-8, 1e-3]))
                rc, dw, vw = O.asw_guided(L, R, dt, eps, win, minD, numD, want_vol=True)
                d, v = ctx.computeAdaptiveWeight_GuidedF(L, R, dt, eps, win, minD, numD, return_cost_volume=True)
                ok = close(v, vw) and np.array_equal(d, dw)
            elif method == "guided2":
                eps = float(rng.choice([1e-6, 1e-8, 1e-3]))
                rc, dw, vw = O.asw_guided2(L, R, 0, eps, win, minD, numD, want_vol=True)
                d, v = ctx.computeAdaptiveWeight_GuidedF_2(L, R, 0, eps, win, minD, numD, return_cost_volume=True)
                ok = close(v, vw) and np.array_equal(d, dw)
            elif method == "guided3":
                rc, dw, vw = O.asw_guided3(L, R, dt, 1e-6, win, minD, numD, want_vol=True)
                d, v = ctx.computeAdaptiveWeight_GuidedF_3(L, R, dt, 1e-6, win, minD, numD, return_cost_volume=True)
                # flat windows make NaN costs: their WTA is build-defined on both sides, compare where finite
                ok = close(v, vw) and np.array_equal(d[np.isfinite(vw).all(axis=0)], dw[np.isfinite(vw).all(axis=0)])
            elif method == "bilgrid":
                sS = float(rng.choice([2.5, 4, 6, 7.5, 10, 16]))
                sR = float(rng.choice([3, 5, 10, 33.3, 40, 64, 128, 300]))
                if rng.integers(0, 2):  # flat regions: bins with enough pixels for the int counts to survive
                    L = (L // 64) * 64
                    R = (R // 64) * 64
                nD = min(numD, 12)
                rc, dw, vw = O.asw_bilgrid(L, R, 0, sS, sR, minD, nD, want_vol=True)
                d, v = ctx.computeAdaptiveWeight_bilateralGrid(L, R, 0, sS, sR, minD, nD, return_cost_volume=True)
                ok = np.array_equal(v, vw, equal_nan=True) and np.array_equal(d, dw)
            elif method == "resident":
                # the resident split (asw_upload_pair / asw_match_resident / asw_download_*) against the one-call entry point
                alg = int(rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 10, 11]))
                w2 = win if win % 2 else win + 1
                w2 = min(w2, 15)
                dt2 = dt if alg in (2, 4, 6, 7, 9, 11) else 0
                md = 0 if alg == 6 else minD
                nD = min(numD, 16)
                slot = int(rng.integers(0, 3))
                keep = bool(rng.integers(0, 2)) and not (alg == 11 and dt2 == 1)  # computeNCC's RIGHT branch never forms a cost
                want = ctx.stereoMatching(L, R, dt2, alg, w2, md, nD, return_cost_volume=keep)
                ctx.upload_pair(slot, L, R)
                ctx.match_resident(slot, dt2, alg, w2, md, nD, keep_volume=keep)
                got = ctx.download_disparity(slot, L.shape[:2])
                ok = np.array_equal(got, want[0] if keep else want)
                if keep and alg != 11:
                    ok = ok and np.array_equal(ctx.download_volume(slot, want[1].shape), want[1], equal_nan=True)
            elif method == "batch":
                # the frame scheduler: a few frames of different shapes, one to three scheduler threads on the device
                alg = int(rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 10, 11]))
                w2 = min(win if win % 2 else win + 1, 15)
                dt2 = dt if alg in (2, 4, 6, 7, 9, 11) else 0
                md = 0 if alg == 6 else minD
                nD = min(numD, 12)
                fr = []
                for q in range(int(rng.integers(1, 6))):
                    hq, wq = int(rng.integers(1, 60)), int(rng.integers(1, 160))
                    fr.append(make_pair(hq, wq, 4, seed=seed + q, block=8)[:2])
                devs = [0] * int(rng.integers(1, 4))
                outs = asw.stereoMatchingBatch([f[0] for f in fr], [f[1] for f in fr], dt2, alg, w2, md, nD, device_ids=devs)
                ok = all(np.array_equal(o, ctx.stereoMatching(f[0], f[1], dt2, alg, w2, md, nD)) for f, o in zip(fr, outs))
            elif method == "lrcheck":
                a = rng.integers(-2, numD + 3, (H, W)).astype(np.float32)
                b = rng.integers(-2, numD + 3, (H, W)).astype(np.float32)
                a[rng.random((H, W)) < 0.02] = np.nan
                tau = float(rng.choice([0.0, 1.0, 2.5]))
                want, wbad = O.lr_check(a, b, tau, -1.0)
                got, bad = ctx.leftRightCheck(a, b, tau, -1.0)
                ok = np.array_equal(got, want) and bad == wbad
            elif method == "wmedian":
                if win > 37:  # per-pixel sort (64-bit keys, 2048 slots): small frames only
                    H2, W2 = min(H, 14), min(W, 40)
                    L, R = np.ascontiguousarray(L[:H2, :W2]), np.ascontiguousarray(R[:H2, :W2])
                    numD = min(numD, 6)
                elif win > 17:  # general tile form; the oracle's multimap per (pixel, d) sets the size
                    H2, W2 = min(H, 24), min(W, 64)
                    L, R = np.ascontiguousarray(L[:H2, :W2]), np.ascontiguousarray(R[:H2, :W2])
                    numD = min(numD, 10)
                rs, rr = [(10, 10), (5, 20), (3, 3), (25, 2.5)][int(rng.integers(0, 4))]
                rc, dw, vw = O.asw_wmedian(L, R, 0, win, rs, rr, minD, numD, want_vol=True)
                d, v = ctx.computeAdaptiveWeight_WeightedMedian(L, R, 0, win, rs, rr, minD, numD, return_cost_volume=True)
                ok = np.array_equal(v, vw) and np.array_equal(d, dw)
            elif method == "blo1":
                rate = float(rng.choice([0.015, 0.05, 0.004, 0.2]))
                rc, dw, vw = O.asw_blo1(L, R, dt, rate, win, 0, numD, want_vol=True)
                d, v = ctx.computeAdaptiveWeight_BLO1(L, R, dt, rate, win, 0, numD, return_cost_volume=True)
                fin = np.isfinite(vw)
                ok = np.array_equal(v[fin], vw[fin]) and np.array_equal(np.isnan(v), np.isnan(vw)) and np.array_equal(d, dw)
            elif method == "ncc":
                rc, dw = O.ncc_disparity(L, R, dt, win, minD, numD)
                ok = np.array_equal(ctx.computeNCC(L, R, dt, win, minD, numD), dw)
            elif method == "ad_tad":
                T = int(rng.integers(-5, 261))
                if rng.integers(0, 3) == 0:  # 1-channel branch, M.cpp:264-291
                    L, R = np.ascontiguousarray(L[:, :, 1]), np.ascontiguousarray(R[:, :, 2])
                rc, vw = O.compute_ad(L, R, dt, minD, numD)
                rc, tw = O.compute_tad(L, R, dt, T, minD, numD)
                ok = (np.array_equal(np.stack(ctx.computeAD(L, R, dt, minD, numD)), vw) and
                      np.array_equal(np.stack(ctx.computeTAD(L, R, dt, T, minD, numD)), tw) and
                      np.array_equal(np.stack(ctx.computeSD(L // 8, R // 8, dt, minD, numD)), O.compute_sd(L // 8, R // 8, dt, minD, numD)[1]))
            elif method == "similarity":
                reg = float(rng.choice([0.4, 0.0, 1.0, 0.25]))
                thc = float(rng.choice([10, 0, 254.5, 300, -1, 3.7]))
                thg = float(rng.choice([50, 0, 0.5, 1e4, 12.3]))
                wpad = win if rng.integers(0, 3) == 0 and win % 2 == 1 else None   # the padded overload, M.cpp:651-668
                rc, vw = O.compute_similarity(L, R, reg, thc, thg, 0, minD, numD, win=wpad)
                ok = np.array_equal(np.stack(ctx.computeSimilarity(L, R, reg, thc, thg, 0, minD, numD, winSize=wpad)), vw)
            elif method == "sad":
                rc, vw = O.cost_sad(L, R, dt, win, minD, numD)
                ok = np.array_equal(np.stack(ctx.getCostSAD(L, R, dt, win, minD, numD)), vw)
            elif method == "geodist":
                if win > 15:
                    L = np.ascontiguousarray(L[:20, :64])
                rc, vw = O.geodesic_dist(L, win, 3)
                ok = np.array_equal(ctx.getGeodesicDist(L, win, 3), vw)
            elif method == "gfilter":
                P = rng.random((H, W), dtype=np.float32)
                guide = L if dt == 0 else np.concatenate([L, R], axis=2)
                rc, qw = O.guided_filter(guide, P, max(win, 1), 1e-6)
                q = ctx.getGuidedFilter(guide, P, max(win, 1), 1e-6)
                ok = close(q, qw)
            elif method == "prep":
                dw_, dh_ = int(rng.integers(1, 2 * W + 2)), int(rng.integers(1, 2 * H + 2))
                if rng.integers(0, 4) == 0 and H % 2 == 0 and W % 2 == 0:
                    dw_, dh_ = W // 2, H // 2   # the exact 2x downscale resize() does with INTER_AREA
                boost = bool(dt)
                ok = bool(ctx.preprocess_pair(7, L, R, (dw_, dh_), detail_boost=boost))
                gl, gr = ctx.download_pair(7, (dh_, dw_, 3))
                ok = ok and np.array_equal(gl, O.preprocess(L, (dw_, dh_), boost)) and np.array_equal(gr, O.preprocess(R, (dw_, dh_), boost))
                if ok and dw_ >= 4 and dh_ >= 2:  # match on the processed pair, 8-bit map as the driver writes it (main.cpp:97-98)
                    nrm = bool(rng.integers(0, 2))
                    ctx.match_resident(7, 0, 3, 5, 0, 6)
                    d8 = ctx.download_disparity_u8(7, (dh_, dw_), normalize=nrm)
                    ok = np.array_equal(d8, O.disparity_to_u8(ctx.download_disparity(7, (dh_, dw_)), nrm))
            else:
                rc, vw = O.cost_ncc(L, R, dt, win, minD, numD)
                ok = np.array_equal(np.stack(ctx.computeNCC_costs(L, R, dt, win, minD, numD)), vw, equal_nan=True)
        except Exception as e:  # noqa: BLE001
            ok = False
            print("EXC", tag, repr(e), flush=True)
        n += 1
        counts[method] = counts.get(method, 0) + 1
        if time.time() - last > 30:  # a line now and then: long silent runs look hung to the job runner
            last = time.time()
            print("progress: %d cases, %d failures, %.0f s" % (n, fails, last - t0), flush=True)
        if not ok:
            fails += 1
            print("MISMATCH", tag, flush=True)
    print("cases", n, "failures", fails, counts)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
