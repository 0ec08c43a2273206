/* The drop-in boundary is plain C: this file is compiled with `gcc -std=c99` against include/asw_mi355x.h only.
 * usage: abi_demo rows cols left.raw right.raw algorithm win minD numD out_disp.raw out_u8.raw
 * Runs the driver sequence of the reference's main(): (pre-processed pair already at the matching size here) upload,
 * match on the device, download the f32 disparity and the normalised 8-bit one. */
#include <stdio.h>
#include <stdlib.h>

#include "asw_mi355x.h"

static unsigned char* slurp(const char* path, size_t n)
{
    unsigned char* p = (unsigned char*)malloc(n);
    FILE* f = fopen(path, "rb");
    if (!p || !f || fread(p, 1, n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    return p;
}

int main(int argc, char** argv)
{
    if (argc != 11) { fprintf(stderr, "usage\n"); return 2; }
    const int rows = atoi(argv[1]), cols = atoi(argv[2]), alg = atoi(argv[5]), win = atoi(argv[6]), minD = atoi(argv[7]),
              numD = atoi(argv[8]);
    const size_t n = (size_t)rows * cols;
    unsigned char* l = slurp(argv[3], n * 3);
    unsigned char* r = slurp(argv[4], n * 3);
    float* disp = (float*)malloc(n * sizeof(float));
    unsigned char* d8 = (unsigned char*)malloc(n);
    asw_ctx* ctx = NULL;
    int rc = asw_create(0, &ctx);
    if (rc != ASW_OK) { printf("create: %s\n", asw_status_string(rc)); return 1; }
    asw_image li = {l, rows, cols, 3, ASW_8U, (size_t)cols * 3}, ri = {r, rows, cols, 3, ASW_8U, (size_t)cols * 3};
    asw_image di = {disp, rows, cols, 1, ASW_32F, (size_t)cols * 4}, d8i = {d8, rows, cols, 1, ASW_8U, (size_t)cols};
    rc = asw_upload_pair(ctx, 0, &li, &ri);
    if (rc == ASW_OK) rc = asw_match_resident(ctx, 0, ASW_DISPARITY_LEFT, alg, win, minD, numD, 0);
    if (rc == ASW_OK) rc = asw_download_disparity(ctx, 0, &di);
    if (rc == ASW_OK) rc = asw_download_disparity_u8(ctx, 0, &d8i, 1);
    if (rc != ASW_OK) { printf("status %d: %s\n", rc, asw_status_string(rc)); asw_destroy(ctx); return rc == ASW_ERR_EVEN_WINDOW ? 0 : 1; }
    asw_timing t;
    asw_get_timing(ctx, &t);
    FILE* f = fopen(argv[9], "wb"); fwrite(disp, sizeof(float), n, f); fclose(f);
    f = fopen(argv[10], "wb"); fwrite(d8, 1, n, f); fclose(f);
    printf("ok %d %d kernels %.3f ms\n", rows, cols, t.total_ms);
    asw_destroy(ctx);
    free(l); free(r); free(disp); free(d8);
    return 0;
}
