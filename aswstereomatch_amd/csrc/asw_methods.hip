// Method runners behind the selector stereoMatching (M.cpp:46-88): tables, scratch buffers and launch sequences of every
// method, with the literals the selector hard-codes and the reference's error behaviour.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "asw_internal.h"
#include "asw_host.h"

// ------------------------------------------------------------------------------------------
// classic bilateral ASW: host-side tables (tap list with the reference's two index conventions,
// weight LUT with the reference's expression) -- M.cpp:1044-1066, 1088-1102, SURVEY App. B-2
// ------------------------------------------------------------------------------------------
// mirror = 1: table for the x-mirrored problem (DISPARITY_RIGHT runs as DISPARITY_LEFT on mirrored, swapped images:
// M.cpp:1134-1138 is M.cpp:1104-1108 under x -> W-1-x), i.e. every x direction negated, tap ORDER unchanged.
// Tap table + weight LUT of the bilateral kernel.  kind 0: computeAdaptiveWeight (M.cpp:1041-1102);
// kind 1: computeAdaptiveWeight_direct8 (M.cpp:1195-1221, 1238-1259).
static int ensure_bilateral_tables(asw_ctx* ctx, int kind, int win, double gamma_c, double gamma_g, int mirror)
{
    BilateralTables& t = ctx->bil;
    if (t.kind == kind && t.win == win && t.gamma_c == gamma_c && t.gamma_g == gamma_g && t.mirror == mirror && t.taps.p)
        return ASW_OK;
    const int ks = win, h = ks / 2;
    std::vector<int> dxw, dyw, dxs, dys;  // weight direction (build order) / sample offset (consume order)
    if (kind == 0) {
        const int nt = ks * ks - 1;
        dxs.resize(nt); dys.resize(nt);
        for (int j = -h; j < h + 1; j++)          // build order of the weight maps, M.cpp:1044-1053
            for (int i = -h; i < h + 1; i++) {
                if (i == 0 && j == 0) continue;
                dxw.push_back(i); dyw.push_back(j);
            }
        for (int i = 0; i < nt; i++) {            // consume order of the samples, M.cpp:1088-1102
            int kx, ky;
            if (i > ks * ks / 2) { kx = (i + 1) / ks; ky = (i + 1) % ks; }
            else { kx = i / ks; ky = i % ks; }
            dxs[i] = -h + kx; dys[i] = -h + ky;
        }
    } else {
        for (int j = -h; j < h + 1; j++)          // M.cpp:1195-1201 == 1238-1245: same order, same test
            for (int i = -h; i < h + 1; i++) {
                if (i == 0 && j == 0) continue;
                if (i == j || i == 0 || j == 0 || (i + j) == ks - 1) { dxw.push_back(i); dyw.push_back(j); }
            }
        dxs = dxw; dys = dyw;                     // the sample is the neighbour the weight was built for
    }
    const int nt = (int)dxw.size();
    // distance classes: distinct values of i*i + j*j
    std::vector<int> cls_of_r2(2 * h * h + 1, -1);
    std::vector<int> r2s;
    for (int i = 0; i < nt; i++) {
        int r2 = dxw[i] * dxw[i] + dyw[i] * dyw[i];
        if (cls_of_r2[r2] < 0) { cls_of_r2[r2] = (int)r2s.size(); r2s.push_back(r2); }
    }
    // The kernel consumes taps in groups of 4: pad with taps of an all-zero weight class (0*w*c adds +0.0 to both sums).
    const int nt_pad = (nt + 3) / 4 * 4, zero_cls = (int)r2s.size();
    std::vector<float> lut((r2s.size() + 1) * 256, 0.0f);
    const double k = 3;  // M.cpp:1024, 1175
    for (size_t c = 0; c < r2s.size(); c++) {
        double delta_g = sqrt((double)r2s[c]);  // M.cpp:1054, 1205
        for (int dc = 0; dc < 256; dc++) {
            double delta_c = (double)dc;
            lut[c * 256 + dc] = (float)(k * exp(-(delta_c / gamma_c + delta_g / gamma_g)));  // M.cpp:1065, 1214
        }
    }
    std::vector<int4> taps(nt_pad);
    const int LW = bilateral_lds_row_stride(win);  // row stride of the kernel's LDS sample tile
    for (int i = 0; i < nt; i++) {
        const int sx = mirror ? -1 : 1;
        taps[i].x = dys[i] * LW + sx * dxs[i];  // sample cell, consume order (classic: transposed, App. B-2)
        taps[i].y = sx * dxw[i];                // weight direction, build order
        taps[i].z = dyw[i];
        taps[i].w = cls_of_r2[dxw[i] * dxw[i] + dyw[i] * dyw[i]] * 256;
    }
    for (int i = nt; i < nt_pad; i++) taps[i] = make_int4(0, 0, 0, zero_cls * 256);
    // Cell-indexed form of the same table for k_asw_bilateral_xq: window cell (kx, ky) -> the weight map that is applied to the
    // sample at that cell, {dx, dy of the direction the map was BUILT for, class * 256}.  Columns kx = -3..17 (units run up to
    // three columns behind the step counter); the all-zero class stands for "no tap": outside the window and the one cell the
    // reference's index arithmetic skips (kernel_x 7, kernel_y 8: M.cpp:1090-1099 jumps from i = 112 to the cell of i + 1).
    std::vector<int4> cells;
    if (kind == 0 && win == 15) {  // image coordinates, for either direction (the xq kernel does not mirror)
        cells.assign(21 * 15, make_int4(0, 0, zero_cls * 256, 0));
        for (int i = 0; i < nt; i++) {
            const int kx = dxs[i] + h, ky = dys[i] + h;
            cells[(kx + 3) * 15 + ky] = make_int4(dxw[i], dyw[i], cls_of_r2[dxw[i] * dxw[i] + dyw[i] * dyw[i]] * 256, 1);
        }
    }
    ASW_TRY(t.taps.ensure((taps.size() > 4 ? taps.size() : 4) * sizeof(int4)));  // never a null table, even for win = 1 (no taps)
    ASW_TRY(t.lut.ensure(lut.size() * sizeof(float)));
    if (!taps.empty())  // win = 1 has no taps at all (every E is 0/0)
        ASW_HIP_TRY(hipMemcpyAsync(t.taps.p, taps.data(), taps.size() * sizeof(int4), hipMemcpyHostToDevice, ctx->stream));
    ASW_HIP_TRY(hipMemcpyAsync(t.lut.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (!cells.empty()) {
        ASW_TRY(t.cells.ensure(cells.size() * sizeof(int4)));
        ASW_HIP_TRY(hipMemcpyAsync(t.cells.p, cells.data(), cells.size() * sizeof(int4), hipMemcpyHostToDevice, ctx->stream));
    }
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));  // host vectors die at return
    t.kind = kind; t.win = win; t.gamma_c = gamma_c; t.gamma_g = gamma_g; t.mirror = mirror; t.ntaps = nt_pad;
    t.ncls = (int)r2s.size() + 1;
    return ASW_OK;
}


// computeAdaptiveWeight (direct8 = false) and computeAdaptiveWeight_direct8 (direct8 = true: sparse support, its own
// gamma_g, DISPARITY_LEFT only -- the RIGHT branch of the reference indexes its weight vectors with a negative tap
// coordinate, M.cpp:1291-1295)
static int run_bilateral(asw_ctx* ctx, Frame* f, const MatchParams& mp, bool keep_volume, bool direct8 = false)
{
    if (mp.win % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // build decision: the reference has no guard
    if (mp.win < 1) return ASW_ERR_BAD_ARGUMENT;
    if (f->channels != 3) return ASW_ERR_UNSUPPORTED_LAYOUT;  // cvtColor(BGR2GRAY) asserts scn==3/4
    if (mp.disparity_type != ASW_DISPARITY_LEFT && mp.disparity_type != ASW_DISPARITY_RIGHT) return ASW_ERR_BAD_ARGUMENT;
    if (direct8 && mp.disparity_type != ASW_DISPARITY_LEFT) return ASW_ERR_UNSUPPORTED_LAYOUT;
    const int flip = mp.disparity_type == ASW_DISPARITY_RIGHT ? 1 : 0;
    if (mp.win > 127) return ASW_ERR_BAD_ARGUMENT;
    const int H = f->rows, W = f->cols, nD = mp.numD + 1;  // inclusive range, M.cpp:1021,1074
    if (direct8)
        ASW_TRY(ensure_bilateral_tables(ctx, 1, mp.win, 30.0, (double)(mp.win * 2 / 3), 0));  // M.cpp:1175: integer division
    else
        ASW_TRY(ensure_bilateral_tables(ctx, 0, mp.win, mp.gamma_c, mp.gamma_g, flip));
    DevBuf& gl = ctx->buf("grayL");
    DevBuf& gr = ctx->buf("grayR");
    ASW_TRY(gl.ensure((size_t)H * W));
    ASW_TRY(gr.ensure((size_t)H * W));
    ASW_TRY(f->disp.ensure((size_t)H * W * 4));
    f->vol_floats = 0;
    if (keep_volume) {
        ASW_TRY(f->vol.ensure((size_t)nD * H * W * 4));
        f->vol_floats = (size_t)nD * H * W;
    }
    ASW_TRY(launch_bgr2gray(ctx->stream, f->L.as<uint8_t>(), H, W, gl.as<uint8_t>(), ctx->gray_bits));
    ASW_TRY(launch_bgr2gray(ctx->stream, f->R.as<uint8_t>(), H, W, gr.as<uint8_t>(), ctx->gray_bits));
    BilateralLaunch a;
    a.gL = flip ? gr.as<uint8_t>() : gl.as<uint8_t>();  // RIGHT: reference image = right, read mirrored in the kernel
    a.gR = flip ? gl.as<uint8_t>() : gr.as<uint8_t>();
    a.flip = flip;
    a.H = H; a.W = W; a.win = mp.win; a.minD = mp.minD; a.nD = nD;
    a.taps = ctx->bil.taps.as<int4>(); a.lut = ctx->bil.lut.as<float>(); a.ntaps = ctx->bil.ntaps;
    a.vol = keep_volume ? f->vol.as<float>() : nullptr;
    a.disp = f->disp.as<float>();
    a.partE = nullptr; a.partD = nullptr; a.max_slices = 0;
    // Candidate ranges of the reference's own configuration (15x15, either direction) from 64 candidates up take the xq form of the
    // kernel for the first 128 (8 wavefronts per workgroup) or 64 (4 wavefronts: the reference's own call site passes
    // numDisparity 64 -> 65 candidates, aswStereoMatch.cpp:94) and this kernel for the tail.  The tile of the outermost workgroup must still hold the eight image
    // columns next to the border its positions clamp to: LEFT minD <= 48 (columns 0..7 in the first tile), RIGHT
    // x0_last + minD <= W - 1 (columns W-8..W-1 in the last).  ASW_BILATERAL_XQ=0 forces the one-kernel path (A/B, tests).
    const bool xq_fits = flip ? (W - 1) / 64 * 64 + mp.minD <= W - 1 : mp.minD <= 48;
    const int xq_waves = nD >= bilateral_xq_candidates(8) ? 8 : 4;
    const bool use_xq = !direct8 && mp.win == 15 && nD >= bilateral_xq_candidates(xq_waves) && mp.minD >= 0 && xq_fits && W >= 64 &&
                        ctx->tune.bilateral_xq != 0;
    if (use_xq) {
        DevBuf& pe = ctx->buf("bil_partE");
        DevBuf& pd = ctx->buf("bil_partD");
        const size_t plane = (size_t)H * W;
        ASW_TRY(pe.ensure(2 * plane * sizeof(double)));
        ASW_TRY(pd.ensure(2 * plane * sizeof(float)));
        a.partE = pe.as<double>(); a.partD = pd.as<float>(); a.max_slices = 2;
        a.c_begin = bilateral_xq_candidates(xq_waves);
        const bool tail = nD > a.c_begin;  // numDisparity = 127 / 63 ends exactly at the xq kernel's 128 / 64 candidates
        ASW_HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
        // fork: border tiles and the tail are independent of the interior launch (they write other pixels / another slice of the
        // per-slice winners); on side streams they overlap it instead of adding two latency-bound 0.5 ms launches to the frame
        ASW_HIP_TRY(hipEventRecord(ctx->aux_ev[0], ctx->stream));
        ASW_HIP_TRY(hipStreamWaitEvent(ctx->aux[0], ctx->aux_ev[0], 0));
        ASW_TRY(launch_bilateral_xq(ctx->stream, ctx->aux[0], xq_waves, a.gL, a.gR, H, W, mp.minD, ctx->bil.cells.as<int4>(), a.lut, a.vol,
                                    a.partE, a.partD, tail ? nullptr : a.disp, flip != 0));
        ASW_HIP_TRY(hipEventRecord(ctx->aux_ev[1], ctx->aux[0]));
        if (tail) {
            ASW_HIP_TRY(hipStreamWaitEvent(ctx->aux[1], ctx->aux_ev[0], 0));
            a.out_slice = 1;  // candidates [128 | 64, nD) -> slice 1; the xq launches fill slice 0
            ASW_TRY(launch_bilateral(ctx->aux[1], a));
            ASW_HIP_TRY(hipEventRecord(ctx->aux_ev[2], ctx->aux[1]));
            ASW_HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->aux_ev[2], 0));
        }
        ASW_HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->aux_ev[1], 0));  // join
        if (tail) ASW_TRY(launch_merge_slices(ctx->stream, a.partE, a.partD, 2, plane, a.disp));  // strict '<', ascending d
        ASW_HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
        ctx->timing.aggregate_launches = tail ? 4 : 2;
        return ASW_OK;
    }
    if ((size_t)H * W <= (size_t)1 << 20) {  // small frames only: scratch for the grid.z split of the disparity range
        const int max_slices = 8;
        DevBuf& pe = ctx->buf("bil_partE");
        DevBuf& pd = ctx->buf("bil_partD");
        ASW_TRY(pe.ensure((size_t)max_slices * H * W * sizeof(double)));
        ASW_TRY(pd.ensure((size_t)max_slices * H * W * sizeof(float)));
        a.partE = pe.as<double>(); a.partD = pd.as<float>(); a.max_slices = max_slices;
    }
    ASW_HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    ASW_TRY(launch_bilateral(ctx->stream, a));
    ASW_HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->timing.aggregate_launches = 1;
    return ASW_OK;
}


// ------------------------------------------------------------------------------------------
// guided-filter ASW: computeAdaptiveWeight_GuidedF_2 (M.cpp:2976-3050, TAD C+G cost, guide = left image)
// and computeAdaptiveWeight_GuidedF (M.cpp:2867-2963, SAD cost, 6-channel guide [L, R shifted by d])
// ------------------------------------------------------------------------------------------
int build_similarity_volume(asw_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int H, int W, int minD, int numD,
                                   double regularity, double thresC, double thresG, float* cost,
                                   uint32_t* ord_scratch, float2* scales)
{
    const int max_off = minD + numD - 1;
    DevBuf& gl = ctx->buf("scharrL");
    DevBuf& gr = ctx->buf("scharrR");
    ASW_TRY(gl.ensure((size_t)H * W * 3 * sizeof(short)));
    ASW_TRY(gr.ensure((size_t)H * (W + max_off) * 3 * sizeof(short)));
    ASW_TRY(launch_scharr_x(ctx->stream, dL, H, W, 0, gl.as<short>()));
    ASW_TRY(launch_scharr_x(ctx->stream, dR, H, W, max_off, gr.as<short>()));  // gradient of the PADDED right image
    return launch_similarity(ctx->stream, dL, dR, gl.as<short>(), gr.as<short>(), H, W, minD, numD, regularity, thresC, thresG,
                             cost, ord_scratch, scales);
}

// ------------------------------------------------------------------------------------------
// NCC cost (computeNCC / getInputImgNCC, M.cpp:767-1013): gray images, box means, window sums of squares, then k_ncc
// ------------------------------------------------------------------------------------------
int run_ncc_cost(asw_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int H, int W, int disparity_type, int win, int minD,
                        int numD, float* vol /* optional, un-normalised */, float* disp /* optional */, int nwta,
                        int channels)
{
    if (win % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // M.cpp:828-831, 939-942
    if (win < 1 || win > 63) return ASW_ERR_BAD_ARGUMENT;
    const bool right = disparity_type == ASW_DISPARITY_RIGHT;
    const int max_off = minD + numD - 1, Wp = W + max_off;
    const size_t plane = (size_t)H * W, pplane = (size_t)H * Wp;
    DevBuf& g0 = ctx->buf("ncc_gray_ref");
    DevBuf& g1 = ctx->buf("ncc_gray_oth");
    DevBuf& gp = ctx->buf("ncc_gray_pad");
    DevBuf& m0 = ctx->buf("ncc_mean_ref");
    DevBuf& m1 = ctx->buf("ncc_mean_oth");
    DevBuf& s0 = ctx->buf("ncc_ss_ref");
    DevBuf& s1 = ctx->buf("ncc_ss_oth");
    ASW_TRY(g0.ensure(plane)); ASW_TRY(g1.ensure(plane)); ASW_TRY(gp.ensure(pplane));
    ASW_TRY(m0.ensure(plane * 4)); ASW_TRY(m1.ensure(pplane * 4));
    ASW_TRY(s0.ensure(plane * 8)); ASW_TRY(s1.ensure(pplane * 8));
    // COLOR_RGB2GRAY on BGR data (M.cpp:835,840); reference image = left (LEFT) or right (RIGHT)
    if (channels == 3) {
        ASW_TRY(launch_rgb2gray(ctx->stream, right ? dR : dL, H, W, g0.as<uint8_t>(), ctx->gray_bits));
        ASW_TRY(launch_rgb2gray(ctx->stream, right ? dL : dR, H, W, g1.as<uint8_t>(), ctx->gray_bits));
    } else {  // single-channel input is used as it is (M.cpp:833-841: cvtColor only for 3 channels)
        ASW_HIP_TRY(hipMemcpyAsync(g0.p, right ? dR : dL, plane, hipMemcpyDeviceToDevice, ctx->stream));
        ASW_HIP_TRY(hipMemcpyAsync(g1.p, right ? dL : dR, plane, hipMemcpyDeviceToDevice, ctx->stream));
    }
    // the other image is padded by max_offset REFLECT columns: on the left (LEFT, M.cpp:852) / on the right (RIGHT, M.cpp:882)
    ASW_TRY(launch_pad_gray(ctx->stream, g1.as<uint8_t>(), H, W, right ? 0 : max_off, right ? max_off : 0, gp.as<uint8_t>()));
    ASW_TRY(launch_box_mean_u8(ctx->stream, g0.as<uint8_t>(), H, W, win, m0.as<float>()));   // M.cpp:785-786
    ASW_TRY(launch_box_mean_u8(ctx->stream, gp.as<uint8_t>(), H, Wp, win, m1.as<float>()));
    ASW_TRY(launch_ncc_selfsum(ctx->stream, g0.as<uint8_t>(), m0.as<float>(), H, W, win, s0.as<double>()));
    ASW_TRY(launch_ncc_selfsum(ctx->stream, gp.as<uint8_t>(), m1.as<float>(), H, Wp, win, s1.as<double>()));
    NccLaunch a;
    a.gref = g0.as<uint8_t>(); a.mref = m0.as<float>(); a.sref = s0.as<double>();
    a.goth = gp.as<uint8_t>(); a.moth = m1.as<float>(); a.soth = s1.as<double>();
    a.H = H; a.W = W; a.Wp = Wp; a.win = win; a.minD = minD; a.numD = numD; a.right = right ? 1 : 0; a.nwta = nwta;
    a.vol = vol; a.disp = disp;
    return launch_ncc(ctx->stream, a);
}

// computeNCC -> disparity (M.cpp:812-913): candidates minD .. max_offset-1 only, the SMALLEST cost wins (LEFT);
// DISPARITY_RIGHT compares `cost > DBL_MAX`: nothing is ever written -> zeros here (reference: uninitialised Mat).
static int run_ncc(asw_ctx* ctx, Frame* f, const MatchParams& mp, bool keep_volume)
{
    if (f->channels != 3) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (mp.disparity_type != ASW_DISPARITY_LEFT && mp.disparity_type != ASW_DISPARITY_RIGHT) return ASW_ERR_BAD_ARGUMENT;
    if (mp.win % 2 == 0) return ASW_ERR_EVEN_WINDOW;
    const int H = f->rows, W = f->cols;
    const size_t plane = (size_t)H * W;
    ASW_TRY(f->disp.ensure(plane * 4));
    f->vol_floats = 0;
    ASW_HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    if (mp.disparity_type == ASW_DISPARITY_RIGHT) {
        ASW_HIP_TRY(hipMemsetAsync(f->disp.p, 0, plane * 4, ctx->stream));
        if (keep_volume) return ASW_ERR_UNSUPPORTED_LAYOUT;
    } else {
        float* vol = nullptr;
        if (keep_volume) {  // raw (un-normalised) costs of all numD offsets, for inspection
            ASW_TRY(f->vol.ensure(plane * mp.numD * 4));
            f->vol_floats = plane * mp.numD;
            vol = f->vol.as<float>();
        }
        ASW_TRY(run_ncc_cost(ctx, f->L.as<uint8_t>(), f->R.as<uint8_t>(), H, W, mp.disparity_type, mp.win, mp.minD, mp.numD, vol,
                             f->disp.as<float>(), mp.numD - 1));
    }
    ASW_HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->timing.aggregate_launches = 1;
    return ASW_OK;
}

enum GuidedKind { GUIDED_SAD6 = 0 /* GuidedF */, GUIDED_SIM3 = 1 /* GuidedF_2 */, GUIDED_NCC = 2 /* GuidedF_3 */ };

static int run_guided(asw_ctx* ctx, Frame* f, const MatchParams& mp, bool keep_volume, int kind)
{
    const bool variant2 = kind == GUIDED_SIM3;
    // GuidedF_3 + DISPARITY_RIGHT: getGuidedFilter receives the plain right image (M.cpp:3110), a 3-channel guide
    const bool ncc = kind == GUIDED_NCC, plain3 = variant2 || (ncc && mp.disparity_type == ASW_DISPARITY_RIGHT);

    if (f->channels != 3) return ASW_ERR_UNSUPPORTED_LAYOUT;
    // GuidedF_2: RIGHT / gray branches of computeSimilarity throw in the reference (App. B-7).
    if (variant2 && mp.disparity_type != ASW_DISPARITY_LEFT) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (mp.disparity_type != ASW_DISPARITY_LEFT && mp.disparity_type != ASW_DISPARITY_RIGHT) return ASW_ERR_BAD_ARGUMENT;
    const bool right = mp.disparity_type == ASW_DISPARITY_RIGHT;
    if (!variant2 && mp.win % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // getCostSAD_d, M.cpp:2458-2462; computeNCC, M.cpp:939-942
    if (mp.win < 1 || mp.win > 128) return ASW_ERR_BAD_ARGUMENT;
    const int H = f->rows, W = f->cols, n = mp.numD, C = plain3 ? 3 : 6;
    const size_t plane = (size_t)H * W;
    DevBuf& raw = ctx->buf("g_raw");
    DevBuf& ord = ctx->buf("g_ord");
    DevBuf& psc = ctx->buf("g_pscales");
    DevBuf& gsc = ctx->buf("g_gscales");
    DevBuf& stats = ctx->buf("g_stats");
    DevBuf& ab = ctx->buf("g_ab");
    DevBuf& pxa = ctx->buf("bgrxL");
    DevBuf& pxb = ctx->buf("bgrxR");
    const int nstat = plain3 ? 1 : n;
    ASW_TRY(raw.ensure(plane * n * 4));
    DevBuf& parts = ctx->buf("g_parts");
    ASW_TRY(parts.ensure(similarity_parts_words(H, W, n) * 4));
    ASW_TRY(ord.ensure((size_t)(2 * n + 2) * 4));
    ASW_TRY(psc.ensure((size_t)n * sizeof(float2)));
    ASW_TRY(gsc.ensure((size_t)n * sizeof(float2)));
    ASW_TRY(stats.ensure(guided_stats_floats(C, nstat, H, W) * 4));
    // (GuidedF_2 on large frames runs the fused walk: no a/b volume -- 4.4 GB at 1080p D=128)
    const bool fused = guided_uses_fused(ctx->tune, C, plain3 ? 0 : 1, 0, ncc ? 1 : 0, H, W, n, mp.win);
    ASW_TRY(ab.ensure(fused ? 64 : guided_ab_floats(C, n, H, W, mp.win) * 4));
    ASW_TRY(pxa.ensure((plane + 4) * 4));  // + slack: the q pass reads the guide words of a lane's two columns as one pair, the last one may start at column W-1
    ASW_TRY(pxb.ensure((plane + 4) * 4));
    ASW_TRY(f->vol.ensure(plane * n * 4));  // q volume: always needed for the WTA pass
    ASW_TRY(f->disp.ensure(plane * 4));
    f->vol_floats = keep_volume ? plane * n : 0;
    const uint8_t* dL = f->L.as<uint8_t>();
    const uint8_t* dR = f->R.as<uint8_t>();

    GuidedLaunch a;
    // guide = [L, R shifted by -d] (LEFT, M.cpp:2907-2912) or [L shifted by +d, R] (RIGHT, M.cpp:2925-2929)
    a.shiftA = (!plain3 && right) ? 1 : 0; a.shiftB = (!plain3 && !right) ? -1 : 0; a.C = C; a.guide_per_slice = plain3 ? 0 : 1;
    a.nan_safe = ncc ? 1 : 0;  // SAD and TAD C+G costs are finite; an NCC cost is 0/0 where a window is flat
    const uint8_t* dGuide3 = variant2 ? dL : dR;  // the 3-channel guide: left image (GuidedF_2) / right image (GuidedF_3 RIGHT)
    ASW_TRY(launch_pack_words(ctx->stream, plain3 ? dGuide3 : dL, H, W, 3, 0, pxa.as<uint32_t>()));
    if (!plain3) ASW_TRY(launch_pack_words(ctx->stream, dR, H, W, 3, 0, pxb.as<uint32_t>()));
    a.guideA = pxa.as<uint32_t>(); a.guideB = plain3 ? nullptr : pxb.as<uint32_t>();
    if (ncc) {
        // costs_ds of computeNCC (M.cpp:3076): raw planes, then normalize(NORM_MINMAX) of every plane in place
        ASW_TRY(run_ncc_cost(ctx, dL, dR, H, W, mp.disparity_type, mp.win, mp.minD, n, raw.as<float>(), nullptr, 0));
        ASW_TRY(launch_slice_scales(ctx->stream, raw.as<float>(), n, plane, ord.as<uint32_t>(), psc.as<float2>()));
        ASW_TRY(launch_apply_scales(ctx->stream, raw.as<float>(), n, plane, psc.as<float2>()));
        if (plain3) {
            ASW_TRY(launch_u8_scale(ctx->stream, dGuide3, plane * 3, ord.as<uint32_t>() + 2 * n, gsc.as<float2>()));
        } else {
            DevBuf& colmm = ctx->buf("g_colmm");
            ASW_TRY(colmm.ensure((size_t)2 * W * sizeof(int)));
            ASW_TRY(launch_guide_scales_lr(ctx->stream, dL, dR, H, W, mp.minD, n, mp.disparity_type, ord.as<uint32_t>() + 2 * n,
                                           colmm.as<int>(), gsc.as<float2>()));
        }
    } else if (variant2) {
        ASW_TRY(build_similarity_volume(ctx, dL, dR, H, W, mp.minD, n, 0.4, 10, 50, raw.as<float>(), parts.as<uint32_t>(),
                                        psc.as<float2>()));  // M.cpp:2990 (+ the min/max of M.cpp:2775, fused)
        ASW_TRY(launch_u8_scale(ctx->stream, dL, plane * 3, ord.as<uint32_t>() + 2 * n, gsc.as<float2>()));
    } else {
        DevBuf& gl = ctx->buf("grayL");
        DevBuf& gr = ctx->buf("grayR");
        DevBuf& colmm = ctx->buf("g_colmm");
        ASW_TRY(gl.ensure(plane));
        ASW_TRY(gr.ensure(plane));
        ASW_TRY(colmm.ensure((size_t)2 * W * sizeof(int)));
        ASW_TRY(launch_bgr2gray(ctx->stream, dL, H, W, gl.as<uint8_t>(), ctx->gray_bits));
        ASW_TRY(launch_bgr2gray(ctx->stream, dR, H, W, gr.as<uint8_t>(), ctx->gray_bits));
        ASW_TRY(launch_cost_sad(ctx->stream, gl.as<uint8_t>(), gr.as<uint8_t>(), H, W, mp.disparity_type, mp.win, mp.minD, n,
                                raw.as<float>()));  // M.cpp:2884-2889
        ASW_TRY(launch_guide_scales_lr(ctx->stream, right ? dR : dL, right ? dL : dR, H, W, mp.minD, n, mp.disparity_type,
                                       ord.as<uint32_t>() + 2 * n, colmm.as<int>(), gsc.as<float2>()));
    }
    if (!variant2)
        ASW_TRY(launch_slice_scales(ctx->stream, raw.as<float>(), n, plane, ord.as<uint32_t>(), psc.as<float2>()));  // M.cpp:2775
    a.gscales = gsc.as<float2>(); a.P = raw.as<float>(); a.pscales = psc.as<float2>();
    a.H = H; a.W = W; a.n = n; a.r = mp.win; a.minD = mp.minD; a.eps = mp.eps;
    DevBuf& repb = ctx->buf("g_rep");
    ASW_TRY(repb.ensure((size_t)n * sizeof(int)));
    a.stats = stats.as<float>(); a.rep_scratch = repb.as<int>(); a.ab = ab.as<float>(); a.q = f->vol.as<float>();
    a.tune = &ctx->tune;
    ASW_HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    ASW_TRY(launch_guided(ctx->stream, a));
    ASW_TRY(launch_wta(ctx->stream, f->vol.as<float>(), n, H, W, mp.minD, f->disp.as<float>()));  // M.cpp:3032-3048
    ASW_HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->timing.aggregate_launches = plain3 ? (fused ? 3 : 4) : 5;  // statistics, [a/b, q | fused walk], WTA
    return ASW_OK;
}

// ------------------------------------------------------------------------------------------
// geodesic ASW: computeAdaptiveWeight_geodesic (M.cpp:1436-1534)
// ------------------------------------------------------------------------------------------
static int run_geodesic(asw_ctx* ctx, Frame* f, const MatchParams& mp, bool keep_volume)
{
    if (mp.win % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // M.cpp:1440-1443
    if (f->channels != 3) return ASW_ERR_UNSUPPORTED_LAYOUT;  // at<Vec3b>
    if (mp.disparity_type != ASW_DISPARITY_LEFT && mp.disparity_type != ASW_DISPARITY_RIGHT) return ASW_ERR_BAD_ARGUMENT;
    const int flip = mp.disparity_type == ASW_DISPARITY_RIGHT ? 1 : 0;  // M.cpp:1498-1520 == LEFT on the mirrored problem
    if (mp.win < 1 || mp.win > 35) return ASW_ERR_BAD_ARGUMENT;
    const int H = f->rows, W = f->cols, nD = mp.numD + 1;  // inclusive range, M.cpp:1447,1467
    const size_t plane = (size_t)H * W, cells = (size_t)mp.win * mp.win;
    DevBuf& pl = ctx->buf("bgrxL");
    DevBuf& pr = ctx->buf("bgrxR");
    DevBuf& wl = ctx->buf("geoWL");
    DevBuf& wr = ctx->buf("geoWR");
    ASW_TRY(pl.ensure(plane * 4));
    ASW_TRY(pr.ensure(plane * 4));
    ASW_TRY(wl.ensure(plane * cells * 2));
    ASW_TRY(wr.ensure(plane * cells * 2));
    ASW_TRY(f->disp.ensure(plane * 4));
    f->vol_floats = 0;
    if (keep_volume) {
        ASW_TRY(f->vol.ensure(plane * nD * 4));
        f->vol_floats = plane * nD;
    }
    ASW_TRY(launch_pack_bgrx(ctx->stream, f->L.as<uint8_t>(), H, W, pl.as<uint32_t>()));
    ASW_TRY(launch_pack_bgrx(ctx->stream, f->R.as<uint8_t>(), H, W, pr.as<uint32_t>()));
    ASW_HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    ASW_TRY(launch_geodesic_weights_u16(ctx->stream, pl.as<uint32_t>(), H, W, mp.win, 3, wl.as<uint16_t>()));  // M.cpp:1464
    ASW_TRY(launch_geodesic_weights_u16(ctx->stream, pr.as<uint32_t>(), H, W, mp.win, 3, wr.as<uint16_t>()));  // M.cpp:1465
    double* partE = nullptr;
    float* partD = nullptr;
    if (plane <= (size_t)1 << 20) {  // small frames only: scratch for the grid.z split of the disparity range
        DevBuf& pe = ctx->buf("bil_partE");
        DevBuf& pd = ctx->buf("bil_partD");
        ASW_TRY(pe.ensure((size_t)8 * plane * sizeof(double)));
        ASW_TRY(pd.ensure((size_t)8 * plane * sizeof(float)));
        partE = pe.as<double>(); partD = pd.as<float>();
    }
    // Long candidate ranges of the 15x15 case (either direction) run as passes of the xq kernel (128 / 64 candidates each) plus
    // k_asw_geodesic for what is left (< 64 candidates); every pass leaves its winners in one slice, merged at the end with
    // the reference's strict '<' in ascending d.  ASW_GEODESIC_XQ=0 forces the one-kernel path.
    if (mp.win == 15 && nD >= geodesic_xq_pass_candidates(4) && W >= 64 && mp.minD >= 0 && ctx->tune.geodesic_xq != 0) {
        // fixed image (the one the disparity map belongs to) / other image
        const uint32_t* pf = flip ? pr.as<uint32_t>() : pl.as<uint32_t>();
        const uint32_t* po = flip ? pl.as<uint32_t>() : pr.as<uint32_t>();
        const uint16_t* wf = flip ? wr.as<uint16_t>() : wl.as<uint16_t>();
        const uint16_t* wo = flip ? wl.as<uint16_t>() : wr.as<uint16_t>();
        const int nslices_max = nD / 64 + 2;
        DevBuf& pe = ctx->buf("bil_partE");
        DevBuf& pd = ctx->buf("bil_partD");
        ASW_TRY(pe.ensure((size_t)nslices_max * plane * sizeof(double)));
        ASW_TRY(pd.ensure((size_t)nslices_max * plane * sizeof(float)));
        double* sE = pe.as<double>();
        float* sD = pd.as<float>();
        float* vol = keep_volume ? f->vol.as<float>() : nullptr;
        ASW_HIP_TRY(hipEventRecord(ctx->aux_ev[0], ctx->stream));  // fork: border tiles and the tail run beside the passes
        ASW_HIP_TRY(hipStreamWaitEvent(ctx->aux[0], ctx->aux_ev[0], 0));
        ASW_HIP_TRY(hipStreamWaitEvent(ctx->aux[1], ctx->aux_ev[0], 0));
        int cb = 0, ns = 0;
        for (int nw = 8; nw >= 4; nw /= 2)
            while (nD - cb >= geodesic_xq_pass_candidates(nw)) {
                ASW_TRY(launch_geodesic_xq(ctx->stream, ctx->aux[0], nw, pf, po, wf, wo, H, W, mp.minD, cb, vol,
                                           sE + (size_t)ns * plane, sD + (size_t)ns * plane, flip != 0));
                cb += geodesic_xq_pass_candidates(nw);
                ns++;
            }
        if (cb < nD && nD - cb <= 4) {  // the reference's inclusive range: one candidate behind the passes at numDisparity = 192
            ASW_TRY(launch_asw_geodesic_few(ctx->aux[1], pf, po, wf, wo, H, W, mp.minD, cb, nD, flip != 0, vol, sE + (size_t)ns * plane,
                                            sD + (size_t)ns * plane));
            ns++;
        } else if (cb < nD) {
            ASW_TRY(launch_asw_geodesic(ctx->aux[1], pf, po, wf, wo, H, W, mp.win, mp.minD, nD, flip, vol, f->disp.as<float>(), sE, sD,
                                        cb, ns));
            ns++;
        }
        ASW_HIP_TRY(hipEventRecord(ctx->aux_ev[1], ctx->aux[0]));
        ASW_HIP_TRY(hipEventRecord(ctx->aux_ev[2], ctx->aux[1]));
        ASW_HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->aux_ev[1], 0));  // join
        ASW_HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->aux_ev[2], 0));
        ASW_TRY(launch_merge_slices(ctx->stream, sE, sD, ns, plane, f->disp.as<float>()));
    } else if (!flip)
        ASW_TRY(launch_asw_geodesic(ctx->stream, pl.as<uint32_t>(), pr.as<uint32_t>(), wl.as<uint16_t>(), wr.as<uint16_t>(), H, W,
                                    mp.win, mp.minD, nD, 0, keep_volume ? f->vol.as<float>() : nullptr, f->disp.as<float>(),
                                    partE, partD));
    else
        ASW_TRY(launch_asw_geodesic(ctx->stream, pr.as<uint32_t>(), pl.as<uint32_t>(), wr.as<uint16_t>(), wl.as<uint16_t>(), H, W,
                                    mp.win, mp.minD, nD, 1, keep_volume ? f->vol.as<float>() : nullptr, f->disp.as<float>(),
                                    partE, partD));
    ASW_HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->timing.aggregate_launches = 3;
    return ASW_OK;
}

// ------------------------------------------------------------------------------------------
// weighted-median ASW: computeAdaptiveWeight_WeightedMedian (M.cpp:3228-3383)
// ------------------------------------------------------------------------------------------
static int ensure_wmedian_tables(asw_ctx* ctx, int win, double rate_s, double rate_r)
{
    if (ctx->wm_rate_r != rate_r || !ctx->wm_lut2.p) {
        // computeColorWeightGau: exp((d0+d1+d2)/rateR*(-1)) == exp(addWeighted(d0+d1, a, d2, a)) with
        // a = (float)(-1/rateR) (M.cpp:3177-3179); cv::exp restated as expf
        std::vector<float> lut((size_t)511 * 256);
        const float al = (float)((1.0 / rate_r) * (-1.0));
        for (int m = 0; m < 511; m++)
            for (int c = 0; c < 256; c++) {
                float arg = (float)m * al + (float)c * al;
                lut[(size_t)m * 256 + c] = expf(arg);
            }
        ASW_TRY(ctx->wm_lut2.ensure(lut.size() * 4));
        ASW_HIP_TRY(hipMemcpyAsync(ctx->wm_lut2.p, lut.data(), lut.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->wm_rate_r = rate_r;
    }
    if (ctx->wm_rate_s != rate_s || ctx->wm_win != win || !ctx->wm_wd.p) {
        // computeSpaceWeightGau (M.cpp:3207-3226)
        const int h = win / 2;
        std::vector<float> wd((size_t)win * win);
        const float al = (float)((1.0 / rate_s) * (-1.0));
        for (int y = 0; y < win; y++)
            for (int x = 0; x < win; x++) {
                float v = (float)((x - h) * (x - h)) + (float)((y - h) * (y - h));
                wd[(size_t)x * win + y] = expf(v * al);
            }
        ASW_TRY(ctx->wm_wd.ensure(wd.size() * 4));
        ASW_HIP_TRY(hipMemcpyAsync(ctx->wm_wd.p, wd.data(), wd.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->wm_rate_s = rate_s;
        ctx->wm_win = win;
    }
    return ASW_OK;
}

static int run_wmedian(asw_ctx* ctx, Frame* f, const MatchParams& mp, bool keep_volume)
{
    if (mp.win % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // M.cpp:3238-3241
    if (f->channels != 3) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (mp.disparity_type != ASW_DISPARITY_LEFT) return ASW_ERR_UNSUPPORTED_LAYOUT;  // App. B-7 / B-13
    if (mp.win < 1 || mp.win > 45) return ASW_ERR_BAD_ARGUMENT;  // 256-slot fast network up to 15x15, 64-bit general path up to 2048 slots
    const int H = f->rows, W = f->cols, n = mp.numD, cells = mp.win * mp.win;
    const int max_off = mp.minD + mp.numD - 1, Wb = W + max_off;
    const size_t plane = (size_t)H * W;
    ASW_TRY(ensure_wmedian_tables(ctx, mp.win, mp.rate_s, mp.rate_r));
    DevBuf& raw = ctx->buf("g_raw");
    DevBuf& wl = ctx->buf("wmWL");
    DevBuf& wr = ctx->buf("wmWR");
    ASW_TRY(raw.ensure(plane * n * 4));
    ASW_TRY(wl.ensure(plane * cells * 4));
    ASW_TRY(wr.ensure((size_t)H * Wb * cells * 4));
    ASW_TRY(f->vol.ensure(plane * n * 4));
    ASW_TRY(f->disp.ensure(plane * 4));
    f->vol_floats = keep_volume ? plane * n : 0;
    const uint8_t* dL = f->L.as<uint8_t>();
    const uint8_t* dR = f->R.as<uint8_t>();
    ASW_TRY(build_similarity_volume(ctx, dL, dR, H, W, mp.minD, n, 0.4, 10, 50, raw.as<float>()));  // M.cpp:3250
    ASW_HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    ASW_TRY(launch_wm_weights(ctx->stream, dL, H, W, 0, mp.win, ctx->wm_lut2.as<float>(), ctx->wm_wd.as<float>(), wl.as<float>()));
    ASW_TRY(launch_wm_weights(ctx->stream, dR, H, W, max_off, mp.win, ctx->wm_lut2.as<float>(), nullptr, wr.as<float>()));
    // 15x15 (the reference's call site): the neighbourhood of an 8x8 pixel block is sorted once per slice and every pixel walks
    // it (k_wmedian_tile.hip); every other window up to 37x37 likewise (below); 1x1 and 39x39 .. 45x45 sort per pixel (k_wmedian.hip).  Slices go in chunks that keep the sorted lists
    // (3 KB per block and slice) below 2 GiB.  ASW_WMEDIAN_TILE=0 forces the per-pixel sort (A/B measurements, tests).
    if (mp.win == 15 && ctx->tune.wmedian_tile != 0) {
        const size_t per_slice = wmedian_tile_list_slots(H, W, 1);
        int chunk = (int)std::min<size_t>((size_t)n, std::max<size_t>(8, (((size_t)2 << 30) / 6 / per_slice) / 8 * 8));
        if (ctx->tune.wmedian_tile_chunk > 0) chunk = std::max(1, std::min(n, ctx->tune.wmedian_tile_chunk));  // tests: odd chunkings
        DevBuf& lc = ctx->buf("wmListC");
        DevBuf& lp = ctx->buf("wmListP");
        ASW_TRY(lc.ensure(per_slice * chunk * 4));
        ASW_TRY(lp.ensure(per_slice * chunk * 2));
        for (int d0 = 0; d0 < n; d0 += chunk)
            ASW_TRY(launch_wmedian_tile(ctx->stream, raw.as<float>(), wl.as<float>(), wr.as<float>(), H, W, n, max_off, d0,
                                        std::min(chunk, n - d0), lc.as<uint32_t>(), lp.as<uint16_t>(), f->vol.as<float>(),
                                        ctx->tune.wmedian_tile_split));
    } else if (wmedian_tile_gen_supported(mp.win) && ctx->tune.wmedian_tile != 0) {
        // 3x3 .. 13x13, 17x17 .. 37x37: the same scheme with the window a run-time parameter (k_wmedian_tile_gen.hip), 1.5 .. 12 KB of list per
        // block and slice
        const size_t per_slice = wmedian_tile_list_slots(H, W, 1) / 512 * (size_t)wmedian_tile_gen_slots(mp.win);
        int chunk = (int)std::min<size_t>((size_t)n, std::max<size_t>(8, (((size_t)2 << 30) / 6 / per_slice) / 8 * 8));
        if (ctx->tune.wmedian_tile_chunk > 0) chunk = std::max(1, std::min(n, ctx->tune.wmedian_tile_chunk));
        DevBuf& lc = ctx->buf("wmListC");
        DevBuf& lp = ctx->buf("wmListP");
        ASW_TRY(lc.ensure(per_slice * chunk * 4));
        ASW_TRY(lp.ensure(per_slice * chunk * 2));
        for (int d0 = 0; d0 < n; d0 += chunk)
            ASW_TRY(launch_wmedian_tile_gen(ctx->stream, raw.as<float>(), wl.as<float>(), wr.as<float>(), H, W, mp.win, n, max_off, d0,
                                            std::min(chunk, n - d0), lc.as<uint32_t>(), lp.as<uint16_t>(), f->vol.as<float>(),
                                            ctx->tune.wmedian_gen_rows));
    } else
        ASW_TRY(launch_wmedian(ctx->stream, raw.as<float>(), wl.as<float>(), wr.as<float>(), H, W, mp.win, n, max_off,
                               f->vol.as<float>()));
    ASW_TRY(launch_wta(ctx->stream, f->vol.as<float>(), n, H, W, mp.minD, f->disp.as<float>()));  // M.cpp:3365-3381
    ASW_HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->timing.aggregate_launches = 4;
    return ASW_OK;
}

// ------------------------------------------------------------------------------------------
// O(1)-bilateral ASW: computeAdaptiveWeight_BLO1 (M.cpp:2505-2725)
// ------------------------------------------------------------------------------------------
static int run_blo1(asw_ctx* ctx, Frame* f, const MatchParams& mp, bool keep_volume)
{
    if (mp.win % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // getCostSAD_d -> Mat(), M.cpp:2458-2462
    if (f->channels != 3 && f->channels != 1) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (mp.disparity_type != ASW_DISPARITY_LEFT && mp.disparity_type != ASW_DISPARITY_RIGHT) return ASW_ERR_BAD_ARGUMENT;
    // the reference indexes setsJB_ks_ds_x[key][offset] with the ABSOLUTE offset (M.cpp:2659): out of range unless 0
    if (mp.minD != 0) return ASW_ERR_BAD_ARGUMENT;
    if (mp.win < 1 || mp.win > 64) return ASW_ERR_BAD_ARGUMENT;
    const int step = (int)(256 * mp.blo_rate_r);  // M.cpp:2550
    if (step <= 0) return ASW_ERR_BAD_ARGUMENT;   // the reference's key loop would not terminate
    const int H = f->rows, W = f->cols, n = mp.numD;
    const size_t plane = (size_t)H * W;
    // keys 0, step, 2*step, ..., 255 (M.cpp:2551-2560) are implied by `step` in the kernel
    DevBuf& gl = ctx->buf("grayL");
    DevBuf& gr = ctx->buf("grayR");
    DevBuf& raw = ctx->buf("g_raw");
    ASW_TRY(gl.ensure(plane));
    ASW_TRY(gr.ensure(plane));
    ASW_TRY(raw.ensure(plane * n * 4));
    ASW_TRY(f->disp.ensure(plane * 4));
    f->vol_floats = 0;
    if (keep_volume) {
        ASW_TRY(f->vol.ensure(plane * n * 4));
        f->vol_floats = plane * n;
    }
    if (f->channels == 3) {  // M.cpp:2514-2521
        ASW_TRY(launch_bgr2gray(ctx->stream, f->L.as<uint8_t>(), H, W, gl.as<uint8_t>(), ctx->gray_bits));
        ASW_TRY(launch_bgr2gray(ctx->stream, f->R.as<uint8_t>(), H, W, gr.as<uint8_t>(), ctx->gray_bits));
    } else {
        ASW_HIP_TRY(hipMemcpyAsync(gl.p, f->L.p, plane, hipMemcpyDeviceToDevice, ctx->stream));
        ASW_HIP_TRY(hipMemcpyAsync(gr.p, f->R.p, plane, hipMemcpyDeviceToDevice, ctx->stream));
    }
    ASW_TRY(launch_cost_sad(ctx->stream, gl.as<uint8_t>(), gr.as<uint8_t>(), H, W, mp.disparity_type, mp.win, mp.minD, n,
                            raw.as<float>()));  // M.cpp:2529-2547
    ASW_HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    ASW_TRY(launch_blo1(ctx->stream, gl.as<uint8_t>(), gr.as<uint8_t>(), raw.as<float>(), step, H, W, mp.disparity_type, mp.win, n,
                        keep_volume ? f->vol.as<float>() : nullptr, f->disp.as<float>()));
    ASW_HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->timing.aggregate_launches = 1;
    return ASW_OK;
}

// computeAdaptiveWeight_bilateralGrid, M.cpp:2253-2430.  DISPARITY_LEFT only: the RIGHT branches read column `width` of the
// left image (min(x + offset, width), M.cpp:1929, 2356), one past the row.
static int run_bilgrid(asw_ctx* ctx, Frame* f, const MatchParams& mp, bool keep_volume)
{
    if (f->channels != 3 && f->channels != 1) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (mp.disparity_type != ASW_DISPARITY_LEFT) return ASW_ERR_UNSUPPORTED_LAYOUT;
    const int H = f->rows, W = f->cols, n = mp.numD + 1;  // offsets minD .. minD+numD inclusive
    int nx, ny, nz;
    ASW_TRY(bilgrid_dims(H, W, mp.grid_rate_s, mp.grid_rate_r, &nx, &ny, &nz));
    const size_t plane = (size_t)H * W;
    const size_t cells = (size_t)(nx + 1) * (ny + 1) * (nz + 1) * (nz + 1);
    if (cells > ((size_t)1 << 33)) return ASW_ERR_ALLOC;  // 8 G cells = 96 GB of grid
    DevBuf& gl = ctx->buf("grayL");
    DevBuf& gr = ctx->buf("grayR");
    DevBuf& gF = ctx->buf("grid_sum");
    DevBuf& gS = ctx->buf("grid_count");
    DevBuf& best = ctx->buf("grid_best");
    ASW_TRY(gl.ensure(plane));
    ASW_TRY(gr.ensure(plane));
    ASW_TRY(gF.ensure(cells * 8));
    ASW_TRY(gS.ensure(cells * 4));
    ASW_TRY(best.ensure(plane * 8));
    ASW_TRY(f->disp.ensure(plane * 4));
    f->vol_floats = 0;
    if (keep_volume) {
        ASW_TRY(f->vol.ensure(plane * n * 4));
        f->vol_floats = plane * n;
    }
    if (f->channels == 3) {  // M.cpp:2271-2278
        ASW_TRY(launch_bgr2gray(ctx->stream, f->L.as<uint8_t>(), H, W, gl.as<uint8_t>(), ctx->gray_bits));
        ASW_TRY(launch_bgr2gray(ctx->stream, f->R.as<uint8_t>(), H, W, gr.as<uint8_t>(), ctx->gray_bits));
    } else {
        ASW_HIP_TRY(hipMemcpyAsync(gl.p, f->L.p, plane, hipMemcpyDeviceToDevice, ctx->stream));
        ASW_HIP_TRY(hipMemcpyAsync(gr.p, f->R.p, plane, hipMemcpyDeviceToDevice, ctx->stream));
    }
    ASW_HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    ASW_TRY(launch_bilgrid(ctx->stream, gl.as<uint8_t>(), gr.as<uint8_t>(), H, W, mp.grid_rate_s, mp.grid_rate_r, mp.minD, mp.numD,
                           gF.as<double>(), gS.as<int>(), best.as<double>(), keep_volume ? f->vol.as<float>() : nullptr,
                           f->disp.as<float>()));
    ASW_HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->timing.aggregate_launches = 1;
    return ASW_OK;
}

int run_method(asw_ctx* ctx, Frame* f, int algorithm, const MatchParams& mp, bool keep_volume, bool sync)
{
    f->invalidate_results();  // whatever the slot's disparity / volume were, they are not this call's
    if (mp.numD <= 0 || mp.minD < 0) return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    ASW_HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
    int rc;
    switch (algorithm) {  // M.cpp:49-87
    case ASW_ALG_ADAPTIVE_WEIGHT: rc = run_bilateral(ctx, f, mp, keep_volume); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_8DIRECT: rc = run_bilateral(ctx, f, mp, keep_volume, true); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_GEODESIC: rc = run_geodesic(ctx, f, mp, keep_volume); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_BILATERAL_GRID: rc = run_bilgrid(ctx, f, mp, keep_volume); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_BLO1: rc = run_blo1(ctx, f, mp, keep_volume); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER: rc = run_guided(ctx, f, mp, keep_volume, GUIDED_SAD6); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER_2: rc = run_guided(ctx, f, mp, keep_volume, GUIDED_SIM3); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER_3: rc = run_guided(ctx, f, mp, keep_volume, GUIDED_NCC); break;
    case ASW_ALG_NCC: rc = run_ncc(ctx, f, mp, keep_volume); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_MEDIAN: rc = run_wmedian(ctx, f, mp, keep_volume); break;
    default: rc = ASW_ERR_UNSUPPORTED_METHOD; break;
    }
    if (rc != ASW_OK) {
        f->invalidate_results();
        return rc;
    }
    const size_t kept_floats = f->vol_floats;
    f->vol_floats = 0;  // restored together with has_disp once nothing can fail any more
    ASW_HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    if (!sync) {  // pipelined callers (batch scheduler) order and wait on the stream themselves
        f->has_disp = true; f->disp_rows = f->rows; f->disp_cols = f->cols; f->vol_floats = kept_floats;
        return ASW_OK;
    }
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    float t = 0;
    ASW_HIP_TRY(hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[1]));
    ctx->timing.total_ms = t;
    ASW_HIP_TRY(hipEventElapsedTime(&t, ctx->ev[2], ctx->ev[3]));
    ctx->timing.aggregate_ms = t;
    ctx->timing.cost_ms = ctx->timing.total_ms - ctx->timing.aggregate_ms;
    f->has_disp = true; f->disp_rows = f->rows; f->disp_cols = f->cols; f->vol_floats = kept_floats;
    return ASW_OK;
}

extern "C" int asw_match_resident(asw_ctx* ctx, int slot, int disparity_type, int algorithm, int win_size,
                                  int min_disparity, int num_disparity, int keep_volume)
{
    if (!ctx) return ASW_ERR_BAD_ARGUMENT;
    Frame* f = frame_slot(ctx, slot, false);
    if (!f || !f->valid) return ASW_ERR_NO_FRAME;
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    return run_method(ctx, f, algorithm, mp, keep_volume != 0);
}

// ---- internal hooks of the batch scheduler (batch.hip): device buffers of a slot, enqueue without waiting ----
int asw_internal_stage_slot(asw_ctx* ctx, int slot, int rows, int cols, int channels, Frame** out)
{
    Frame* f = frame_slot(ctx, slot, true);
    if (!f) return ASW_ERR_BAD_ARGUMENT;
    const size_t bytes = (size_t)rows * cols * channels;
    ASW_TRY(f->L.ensure(bytes));
    ASW_TRY(f->R.ensure(bytes));
    ASW_TRY(f->disp.ensure((size_t)rows * cols * 4));
    f->rows = rows; f->cols = cols; f->channels = channels; f->valid = true;
    f->invalidate_results();
    *out = f;
    return ASW_OK;
}

int asw_internal_enqueue_match(asw_ctx* ctx, int slot, int disparity_type, int algorithm, int win_size, int min_disparity,
                               int num_disparity)
{
    Frame* f = frame_slot(ctx, slot, false);
    if (!f || !f->valid) return ASW_ERR_NO_FRAME;
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    return run_method(ctx, f, algorithm, mp, false, false);
}

int asw_internal_check_pair(const asw_image* l, const asw_image* r, const asw_image* d)
{
    ASW_TRY(check_pair(l, r));
    return check_disp_out(d, l->rows, l->cols);
}

