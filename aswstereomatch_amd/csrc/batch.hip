// Batch scheduler: frames are independent, so an N-GPU node is used batch-wise --
// frame i goes to device device_ids[i % n_devices], one host thread + one context per device, no
// collective and no peer traffic.  Results land in the caller's per-frame output buffers.
#include <atomic>
#include <thread>
#include <vector>

#include "asw_internal.h"

extern "C" int asw_stereo_match_batch(int n_frames, const asw_image* lefts, const asw_image* rights, asw_image* disps,
                                      int disparity_type, int algorithm, int win_size, int min_disparity,
                                      int num_disparity, int n_devices, const int* device_ids)
{
    if (n_frames < 0 || n_devices <= 0 || (n_frames > 0 && (!lefts || !rights || !disps))) return ASW_ERR_BAD_ARGUMENT;
    if (n_frames == 0) return ASW_OK;
    std::vector<int> devs(n_devices);
    for (int k = 0; k < n_devices; k++) devs[k] = device_ids ? device_ids[k] : k;
    std::atomic<int> first_error(ASW_OK);
    auto worker = [&](int k) {
        asw_ctx* ctx = nullptr;
        int rc = asw_create(devs[k], &ctx);
        for (int i = k; i < n_frames && rc == ASW_OK; i += n_devices)
            rc = asw_stereo_match(ctx, &lefts[i], &rights[i], &disps[i], disparity_type, algorithm, win_size, min_disparity,
                                  num_disparity, nullptr);
        if (ctx) asw_destroy(ctx);
        if (rc != ASW_OK) {
            int expected = ASW_OK;
            first_error.compare_exchange_strong(expected, rc);
        }
    };
    const int nthreads = n_devices < n_frames ? n_devices : n_frames;
    std::vector<std::thread> threads;
    threads.reserve(nthreads);
    for (int k = 0; k < nthreads; k++) threads.emplace_back(worker, k);
    for (auto& t : threads) t.join();
    return first_error.load();
}
