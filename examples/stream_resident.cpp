// stream_resident.cpp -- the reference's per-frame loop (Examples/Demo/RealSenseD435i.cpp:199-321) on the
// device-resident C ABI: what a camera application does when it keeps its frames and features on the GPU.
// Per new frame:  ONE frame upload (+ pyramid) into the slot the frame before last occupied,
//                 gyro prediction on the device (pagk_gyro_predict_device) -> its outputs feed
//                 pagk_track_device directly (no pt_init / affine / status round trip),
//                 results back, Step-3 filter on the host (pagk_post_filter), survivors become the next keypoints.
// Prints the same lines as examples/track_sequence.cpp (which does the same loop through the PatchMatch /
// GyroAidedTracker shell, two frame uploads per pair), so the two can be diffed.
//
// Input: the sequence file of track_sequence.cpp followed by (n_frames-1) x 9 float KRKinv matrices
//        (mK * mRcl * mK.inv(), what GyroAidedTracker::SetRcl computes, src/gyro_aided_tracker.cpp:518).
// Build: g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include -I include examples/stream_resident.cpp \
//        -L <pkg> -l:libpagk_hip.so -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,<pkg> -Wl,-rpath,/opt/rocm/lib
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pagk.h"

#define CHECK_HIP(x)                                                                  \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            std::fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_));            \
            return 10;                                                                \
        }                                                                             \
    } while (0)
#define CHECK_PAGK(x)                                                                 \
    do {                                                                              \
        int rc_ = (x);                                                                \
        if (rc_ != PAGK_OK) {                                                         \
            std::fprintf(stderr, "%s -> %s (%s)\n", #x, pagk_strerror(rc_), pagk_last_error(ctx)); \
            return 11;                                                                \
        }                                                                             \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s sequence.bin [half_patch iterations pyramids]\n", argv[0]);
        return 2;
    }
    const int half = argc > 2 ? std::atoi(argv[2]) : 5, iters = argc > 3 ? std::atoi(argv[3]) : 10,
              pyr = argc > 4 ? std::atoi(argv[4]) : 3;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 3;
    int32_t hdr[4];
    float K[9], dist[4];
    if (std::fread(hdr, 4, 4, f) != 4 || std::fread(K, 4, 9, f) != 9 || std::fread(dist, 4, 4, f) != 4) return 4;
    const int nf = hdr[0], w = hdr[1], h = hdr[2], nk = hdr[3];
    std::vector<std::vector<unsigned char>> img(nf, std::vector<unsigned char>((size_t)w * h));
    for (auto &im : img)
        if (std::fread(im.data(), 1, im.size(), f) != im.size()) return 5;
    std::vector<float> kp((size_t)nk * 2), Rs((size_t)(nf - 1) * 9), KRK((size_t)(nf - 1) * 9);
    if (std::fread(kp.data(), 4, kp.size(), f) != kp.size() || std::fread(Rs.data(), 4, Rs.size(), f) != Rs.size() ||
        std::fread(KRK.data(), 4, KRK.size(), f) != KRK.size())
        return 6;
    std::fclose(f);

    pagk_ctx *ctx = nullptr;
    if (pagk_create(&ctx, 0) != PAGK_OK) {
        std::fprintf(stderr, "no HIP device\n");
        return 7;
    }
    // src/gyro_aided_tracker.cpp:276-282 + eType 4 (:402-408)
    pagk_params p;
    pagk_params_default(&p);
    p.half_patch = half, p.iterations = iters, p.pyramids = pyr;
    p.has_gyro_predict_initial = 1, p.consider_illumination = 1, p.consider_affine = 1, p.regularization_penalty = 0;
    p.fx = K[0], p.fy = K[4], p.cx = K[2], p.cy = K[5];
    p.n_dist_coef = 4;
    for (int k = 0; k < 4; k++) p.dist_coef[k] = dist[k];

    // device-resident per-feature arrays, sized for the initial keypoint count (the list only shrinks)
    float *d_keys, *d_pu, *d_pd, *d_aff, *d_ptun, *d_ptdist;
    uint8_t *d_st_in, *d_st_out;
    double *d_err, *d_dist;
    const size_t cap = (size_t)(nk > 0 ? nk : 1);
    CHECK_HIP(hipMalloc((void **)&d_keys, cap * 8));
    CHECK_HIP(hipMalloc((void **)&d_pu, cap * 8));
    CHECK_HIP(hipMalloc((void **)&d_pd, cap * 8));
    CHECK_HIP(hipMalloc((void **)&d_aff, cap * 16));
    CHECK_HIP(hipMalloc((void **)&d_ptun, cap * 8));
    CHECK_HIP(hipMalloc((void **)&d_ptdist, cap * 8));
    CHECK_HIP(hipMalloc((void **)&d_st_in, cap));
    CHECK_HIP(hipMalloc((void **)&d_st_out, cap));
    CHECK_HIP(hipMalloc((void **)&d_err, cap * 8));
    CHECK_HIP(hipMalloc((void **)&d_dist, cap * 8));
    pagk_outputs d_out{d_ptun, d_ptdist, d_st_out, d_err, d_dist, nullptr, nullptr};

    std::vector<float> keys = kp, pt_un(cap * 2), pt_dist(cap * 2), pp(cap * 2), ppu(cap * 2);
    std::vector<uint8_t> st_pm(cap), st(cap);
    std::vector<double> err(cap), dpred(cap);
    int n = nk;
    double checksum = 0;

    pagk_image im0{img[0].data(), w, h, (int64_t)w};
    CHECK_PAGK(pagk_frame_upload(ctx, 0, &im0, pyr));  // frame 0 is the first reference frame
    for (int k = 1; k < nf && n > 0; k++) {
        const int slot_ref = (k - 1) & 1, slot_cur = k & 1;
        pagk_image imk{img[k].data(), w, h, (int64_t)w};
        CHECK_PAGK(pagk_frame_upload(ctx, slot_cur, &imk, pyr));  // the only image transfer of this pair
        CHECK_HIP(hipMemcpy(d_keys, keys.data(), (size_t)n * 8, hipMemcpyHostToDevice));
        const float *Rk = &Rs[(size_t)(k - 1) * 9];
        CHECK_PAGK(pagk_gyro_predict_device(ctx, &p, w, h, &KRK[(size_t)(k - 1) * 9], Rk + 6, n, d_keys, d_pu, d_pd,
                                            d_st_in, d_aff));
        CHECK_PAGK(pagk_track_device(ctx, &p, slot_ref, slot_cur, n, d_keys, d_pu, d_aff, d_st_in, &d_out));
        CHECK_PAGK(pagk_sync(ctx));
        CHECK_HIP(hipMemcpy(pt_un.data(), d_ptun, (size_t)n * 8, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(pt_dist.data(), d_ptdist, (size_t)n * 8, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(st_pm.data(), d_st_out, (size_t)n, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(err.data(), d_err, (size_t)n * 8, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(dpred.data(), d_dist, (size_t)n * 8, hipMemcpyDeviceToHost));
        // Step 3 (src/gyro_aided_tracker.cpp:289-341): thresholds from the mean pixel error, final mask
        const int tracked = pagk_post_filter(n, half, st_pm.data(), err.data(), dpred.data(), pt_dist.data(),
                                             pt_un.data(), st.data(), pp.data(), ppu.data());
        if (tracked < 0) return 12;
        std::printf("pair %d tracked %d of %d\n", k, tracked, n);
        // Examples/Demo/RealSenseD435i.cpp:254-258: tracked points become the next frame's keypoints
        int m = 0;
        for (int i = 0; i < n; i++)
            if (st[i]) {
                keys[2 * m] = pt_un[2 * i], keys[2 * m + 1] = pt_un[2 * i + 1];
                checksum += pt_un[2 * i] + 2.0 * pt_un[2 * i + 1];
                m++;
            }
        n = m;
    }
    std::printf("survivors %d checksum %.6f\n", n, checksum);
    pagk_destroy(ctx);
    return 0;
}
