// Exercises include/aswMethods_mi355x.hpp (the reference's C++ surface), on asw::Mat without OpenCV or -- built with
// -DASW_WITH_OPENCV -I tests/cpp/cv_stub -- through the header's cv::Mat branch (against a compile-only stand-in, see there):
//   shim_demo <H> <W> <left.raw> <right.raw> <alg> <win> <minD> <numD> <out_disp.raw> [<out_prefix>]
// Reads two 8UC3 images, calls stereoMatching() exactly like aswStereoMatch.cpp:94 does, writes the f32 map.
// With <out_prefix>: also getCostSAD_d (M.h:156) on a right view bordered as M.cpp:2878 does it -> <prefix>.sad, and the
// std::map form of getGeodesicDist (M.h:141) -> the window of pixel (x=3, y=2) in <prefix>.geo.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "aswMethods_mi355x.hpp"

static bool read_file(const char* path, void* dst, size_t n)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    size_t got = fread(dst, 1, n, f);
    fclose(f);
    return got == n;
}

int main(int argc, char** argv)
{
    if (argc != 10 && argc != 11) { fprintf(stderr, "usage\n"); return 2; }
    int H = atoi(argv[1]), W = atoi(argv[2]), alg = atoi(argv[5]), win = atoi(argv[6]), minD = atoi(argv[7]), numD = atoi(argv[8]);
    // Mats are made and measured through the header's own make() / view(), so the same source drives both Mat types
    AswMat L = asw::detail::make(H, W, ASW_8U, 3), R = asw::detail::make(H, W, ASW_8U, 3), disp;
    if (!read_file(argv[3], L.data, (size_t)H * W * 3) || !read_file(argv[4], R.data, (size_t)H * W * 3)) return 3;
    stereoMatching(L, R, disp, DISPARITY_LEFT, (StereoMatchingAlgorithms)alg, win, minD, numD);
    if (disp.empty()) { printf("empty\n"); return 0; }
    std::vector<AswMat> ad;
    computeAD(L, R, ad, DISPARITY_LEFT, minD, 4);
    std::vector<AswMat> sd;
    computeSD(L, R, sd, DISPARITY_LEFT, minD, 3);  // M.h:117-118
    int same = -1;
    if (alg == ADAPTIVE_WEIGHT_BILATERAL_GRID) {   // the per-method function with the selector's literals (M.cpp:67)
        AswMat g = computeAdaptiveWeight_bilateralGrid(L, R, DISPARITY_LEFT, 10, 10, minD, numD);
        same = !g.empty() && memcmp(g.data, disp.data, (size_t)H * W * 4) == 0;
    }
    size_t geo = 0;
    int sadd = -1;
    if (argc == 11) {
        const std::string prefix = argv[10];
        // copyMakeBorder(rightImg, rightImg_border, 0, 0, max_offset, 0, BORDER_REFLECT), M.cpp:2878
        const int max_off = minD + numD - 1;
        AswMat Rb = asw::detail::make(H, W + max_off, ASW_8U, 3);
        const size_t rb_step = asw::detail::view(Rb).step, r_step = asw::detail::view(R).step;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W + max_off; x++) {
                int sx = x - max_off;
                while (sx < 0 || sx >= W) sx = sx < 0 ? -sx - 1 : 2 * W - 1 - sx;  // fedcba|abcdefgh|hgfedcb
                memcpy(Rb.data + (size_t)y * rb_step + 3 * x, R.data + (size_t)y * r_step + 3 * sx, 3);
            }
        AswMat c = getCostSAD_d(L, Rb, minD + 1, DISPARITY_LEFT, win);  // as M.cpp:2884-2889 calls it
        AswMat none = getCostSAD_d(L, R, minD + 1, DISPARITY_LEFT, win);  // not bordered -> Mat() (M.cpp:2473-2476)
        sadd = (!c.empty() && c.rows == H && c.cols == W && none.empty()) ? 1 : 0;
        if (!c.empty()) {
            FILE* fs = fopen((prefix + ".sad").c_str(), "wb");
            fwrite(c.data, 1, (size_t)H * W * 4, fs);
            fclose(fs);
        }
        std::map<AswPoint, AswMat, MY_COMP_Point2i> wmap;
        getGeodesicDist(L, wmap, win, 3);
        geo = wmap.size();
        if (geo) {
            const AswMat& w = wmap[AswPoint(3, 2)];
            FILE* fg = fopen((prefix + ".geo").c_str(), "wb");
            fwrite(w.data, 1, (size_t)win * win * 4, fg);
            fclose(fg);
        }
    }
    printf("ok %d %d planes=%zu sd=%zu same=%d geo=%zu sadd=%d\n", disp.rows, disp.cols, ad.size(), sd.size(), same, geo, sadd);
    FILE* f = fopen(argv[9], "wb");
    fwrite(disp.data, 1, (size_t)H * W * 4, f);
    fclose(f);
    return 0;
}
