// Exercises include/aswMethods_mi355x.hpp (the reference's C++ surface) without OpenCV:
//   shim_demo <H> <W> <left.raw> <right.raw> <alg> <win> <minD> <numD> <out_disp.raw>
// Reads two 8UC3 images, calls stereoMatching() exactly like aswStereoMatch.cpp:94 does, writes the f32 map.
#include <cstdio>
#include <cstdlib>
#include <cstring>

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
    if (argc != 10) { fprintf(stderr, "usage\n"); return 2; }
    int H = atoi(argv[1]), W = atoi(argv[2]), alg = atoi(argv[5]), win = atoi(argv[6]), minD = atoi(argv[7]), numD = atoi(argv[8]);
    AswMat L(H, W, ASW_8U, 3), R(H, W, ASW_8U, 3), disp;
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
    printf("ok %d %d planes=%zu sd=%zu same=%d\n", disp.rows, disp.cols, ad.size(), sd.size(), same);
    FILE* f = fopen(argv[9], "wb");
    fwrite(disp.data, 1, (size_t)H * W * 4, f);
    fclose(f);
    return 0;
}
