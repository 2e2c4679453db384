// multi_camera_batch.cpp -- BASELINE configs[4] ("batched multi-camera: concurrent streams, shared pyramid upload,
// hipGraph-captured iterate") on the plain C ABI, without Python: k cameras that share ONE GPU and are stepped together.
// The reference builds one PatchMatch per tracker (src/gyro_aided_tracker.cpp:276-283) and each of them runs
// CreatePyramids + OpticalFlowMultiLevel on the CPU; here every camera is one pagk_ctx, and per frame set the application
// issues TWO calls for all of them:
//      pagk_frame_set_device_batch   the k current frames' pyramids          (CreatePyramids, src/patch_match.cpp:61-76)
//      pagk_track_device_batch       the k trackers' OpticalFlowMultiLevel   (src/patch_match.cpp:79-142)
// recorded once into a hipGraph and replayed per frame set.  The program checks what INTEGRATION.md promises: every
// camera's results are the bits of its own pagk_track call (the drop-in on host buffers).
//
// Build: g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include -I include examples/multi_camera_batch.cpp
//        -L <pkg> -l:libpagk_hip.so -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,<pkg> -Wl,-rpath,/opt/rocm/lib
// Run:   multi_camera_batch [cameras = 4] [features per camera = 2500] [frames = 3]
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pagk.h"

#define CHECK_HIP(x)                                                       \
    do {                                                                   \
        hipError_t e_ = (x);                                               \
        if (e_ != hipSuccess) {                                            \
            std::fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); \
            return 10;                                                     \
        }                                                                  \
    } while (0)
#define CHECK_PAGK(c, x)                                                                            \
    do {                                                                                            \
        int rc_ = (x);                                                                              \
        if (rc_ != PAGK_OK) {                                                                       \
            std::fprintf(stderr, "%s -> %s (%s)\n", #x, pagk_strerror(rc_), pagk_last_error(c));   \
            return 11;                                                                              \
        }                                                                                           \
    } while (0)

namespace {

// a smooth texture with structure at several scales (something a KLT tracker can hold on to), sampled at (x, y)
double texture(double x, double y, int cam)
{
    const double p = 0.37 * cam;
    return 128.0 + 38.0 * std::sin(0.071 * x + 0.5 + p) * std::cos(0.053 * y - 0.3) + 30.0 * std::sin(0.193 * x - 0.231 * y + p) +
           22.0 * std::cos(0.317 * x + 0.289 * y) * std::sin(0.127 * y + 1.1 + p) + 14.0 * std::sin(0.611 * x + 0.2) * std::sin(0.577 * y - p);
}

void render(std::vector<uint8_t> &img, int w, int h, int cam, double sx, double sy, double gain, double offset)
{
    img.resize((size_t)w * h);
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            double v = gain * texture(c + sx, r + sy, cam) + offset;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            img[(size_t)r * w + c] = (uint8_t)std::lrint(v);
        }
}

struct Camera {
    pagk_ctx *ctx = nullptr;
    int w = 0, h = 0, n = 0;
    std::vector<uint8_t> ref;
    std::vector<std::vector<uint8_t>> cur;   // one per frame
    std::vector<float> pt_ref, pt_init;
    std::vector<uint8_t> status_in;
    uint8_t *d_ref = nullptr, *d_cur = nullptr, *d_status_in = nullptr, *d_status = nullptr;
    float *d_pt_ref = nullptr, *d_pt_init = nullptr, *d_pt_un = nullptr, *d_pt_dist = nullptr;
    double *d_pix_err = nullptr, *d_dist_pred = nullptr;
    int32_t *d_iters = nullptr;
};

}  // namespace

int main(int argc, char **argv)
{
    const int k = argc > 1 ? std::atoi(argv[1]) : 4, nfeat = argc > 2 ? std::atoi(argv[2]) : 2500, frames = argc > 3 ? std::atoi(argv[3]) : 3;
    if (k < 1 || k > 64 || nfeat < 1 || frames < 1) return 2;
    pagk_params p;
    pagk_params_default(&p);
    p.half_patch = 10, p.iterations = 30, p.pyramids = 3;   // BASELINE's patch, iteration cap and levels
    p.consider_affine = 0;
    p.fx = p.fy = 610.0f, p.cx = 640.0f, p.cy = 360.0f;    // the camera model of the distortion epilogue (:409-416)
    p.dist_coef[0] = -0.28f, p.dist_coef[1] = 0.07f, p.dist_coef[2] = 2e-4f, p.dist_coef[3] = 2e-5f, p.dist_coef[4] = 0.0f, p.n_dist_coef = 4;
    std::printf("pagk %d: %d cameras x %d features, %d frames\n", pagk_version(), k, nfeat, frames);

    hipStream_t stream;
    CHECK_HIP(hipSetDevice(0));
    CHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    const int sizes[4][2] = {{1280, 720}, {640, 480}, {752, 480}, {960, 540}};   // cameras need not be alike
    std::vector<Camera> cams((size_t)k);
    for (int j = 0; j < k; j++) {
        Camera &c = cams[(size_t)j];
        c.w = sizes[j % 4][0], c.h = sizes[j % 4][1], c.n = nfeat - 17 * j > 0 ? nfeat - 17 * j : 1;   // ragged feature counts
        if (pagk_create(&c.ctx, 0) != PAGK_OK) {
            std::fprintf(stderr, "pagk_create failed: no HIP device? (there is no CPU fallback)\n");
            return 3;
        }
        CHECK_PAGK(c.ctx, pagk_set_stream(c.ctx, stream));   // one stream for the batch: nothing to order across streams
        render(c.ref, c.w, c.h, j, 0.0, 0.0, 1.0, 0.0);
        c.cur.resize((size_t)frames);
        for (int f = 0; f < frames; f++) render(c.cur[(size_t)f], c.w, c.h, j, 1.3 + 0.9 * f, -0.8 - 0.4 * f, 1.04, 3.0);
        c.pt_ref.resize((size_t)c.n * 2), c.pt_init.resize((size_t)c.n * 2), c.status_in.assign((size_t)c.n, 1);
        for (int i = 0; i < c.n; i++) {   // keypoints on a jittered lattice, a third of them near the border
            const int cols = (int)std::ceil(std::sqrt((double)c.n * c.w / c.h)), gx = i % cols, gy = i / cols, rows = (c.n + cols - 1) / cols;
            c.pt_ref[2 * (size_t)i] = 6.0f + (c.w - 12.0f) * (gx + 0.5f) / cols + 0.37f * (float)((i * 7) % 5);
            c.pt_ref[2 * (size_t)i + 1] = 6.0f + (c.h - 12.0f) * (gy + 0.5f) / rows + 0.41f * (float)((i * 3) % 7);
            c.pt_init[2 * (size_t)i] = c.pt_ref[2 * (size_t)i] - 1.0f;       // a prediction about a pixel off
            c.pt_init[2 * (size_t)i + 1] = c.pt_ref[2 * (size_t)i + 1] + 0.5f;
            if (i % 11 == 0) c.status_in[(size_t)i] = 0;                       // and some the predictor rejected
        }
        const size_t px = (size_t)c.w * c.h, n = (size_t)c.n;
        CHECK_HIP(hipMalloc((void **)&c.d_ref, px));
        CHECK_HIP(hipMalloc((void **)&c.d_cur, px));
        CHECK_HIP(hipMalloc((void **)&c.d_pt_ref, n * 8));
        CHECK_HIP(hipMalloc((void **)&c.d_pt_init, n * 8));
        CHECK_HIP(hipMalloc((void **)&c.d_status_in, n));
        CHECK_HIP(hipMalloc((void **)&c.d_pt_un, n * 8));
        CHECK_HIP(hipMalloc((void **)&c.d_pt_dist, n * 8));
        CHECK_HIP(hipMalloc((void **)&c.d_status, n));
        CHECK_HIP(hipMalloc((void **)&c.d_pix_err, n * 8));
        CHECK_HIP(hipMalloc((void **)&c.d_dist_pred, n * 8));
        CHECK_HIP(hipMalloc((void **)&c.d_iters, n * 4));
        CHECK_HIP(hipMemcpy(c.d_ref, c.ref.data(), px, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(c.d_pt_ref, c.pt_ref.data(), n * 8, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(c.d_pt_init, c.pt_init.data(), n * 8, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(c.d_status_in, c.status_in.data(), n, hipMemcpyHostToDevice));
    }
    pagk_ctx *lead = cams[0].ctx;

    // the arrays the two batched calls take (host arrays of per-camera values / device pointers)
    std::vector<pagk_ctx *> ctxs;
    std::vector<int32_t> slot0((size_t)k, 0), slot1((size_t)k, 1), ns, ws, hs;
    std::vector<int64_t> steps;
    std::vector<const void *> im_ref, im_cur;
    std::vector<const float *> pt_ref, pt_init;
    std::vector<const uint8_t *> st_in;
    std::vector<pagk_outputs> outs;
    for (Camera &c : cams) {
        ctxs.push_back(c.ctx), ns.push_back(c.n), ws.push_back(c.w), hs.push_back(c.h), steps.push_back(c.w);
        im_ref.push_back(c.d_ref), im_cur.push_back(c.d_cur), pt_ref.push_back(c.d_pt_ref), pt_init.push_back(c.d_pt_init);
        st_in.push_back(c.d_status_in);
        outs.push_back(pagk_outputs{c.d_pt_un, c.d_pt_dist, c.d_status, c.d_pix_err, c.d_dist_pred, nullptr, c.d_iters});
    }
    // reference frames: once
    CHECK_PAGK(lead, pagk_frame_set_device_batch(ctxs.data(), k, slot0.data(), im_ref.data(), ws.data(), hs.data(), steps.data(), p.pyramids));

    auto step = [&]() -> int {   // one frame set: k pyramids, k trackers
        int rc = pagk_frame_set_device_batch(ctxs.data(), k, slot1.data(), im_cur.data(), ws.data(), hs.data(), steps.data(), p.pyramids);
        if (rc) return rc;
        return pagk_track_device_batch(ctxs.data(), k, &p, slot0.data(), slot1.data(), ns.data(), pt_ref.data(), pt_init.data(), nullptr,
                                       st_in.data(), outs.data());
    };
    int32_t graph = -1;
    long long tracked_total = 0, feats_total = 0;
    for (int f = 0; f < frames; f++) {
        for (Camera &c : cams)   // the new frames arrive (same device buffers every frame: the recorded graph stays valid)
            CHECK_HIP(hipMemcpyAsync(c.d_cur, c.cur[(size_t)f].data(), (size_t)c.w * c.h, hipMemcpyHostToDevice, stream));
        if (f == 0) {
            CHECK_PAGK(lead, step());                       // first frame set directly: allocations happen here
            CHECK_HIP(hipStreamSynchronize(stream));
            CHECK_PAGK(lead, pagk_graph_begin(lead));       // ... then record the step once
            int rc = step();
            int rc2 = pagk_graph_end(lead, &graph);
            CHECK_PAGK(lead, rc);
            CHECK_PAGK(lead, rc2);
        }
        CHECK_PAGK(lead, pagk_graph_launch(lead, graph));   // ... and replay it per frame set
        CHECK_HIP(hipStreamSynchronize(stream));
        for (Camera &c : cams) CHECK_PAGK(c.ctx, pagk_check_launch(c.ctx));
        // every camera against its own drop-in call on host buffers (pagk_track = the whole OpticalFlowMultiLevel)
        for (int j = 0; j < k; j++) {
            Camera &c = cams[(size_t)j];
            const size_t n = (size_t)c.n;
            std::vector<float> un(n * 2), di(n * 2), hun(n * 2), hdi(n * 2);
            std::vector<uint8_t> st(n), hst(n);
            std::vector<double> pe(n), dp(n), hpe(n), hdp(n);
            std::vector<int32_t> it(n), hit(n);
            CHECK_HIP(hipMemcpy(un.data(), c.d_pt_un, n * 8, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(di.data(), c.d_pt_dist, n * 8, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(st.data(), c.d_status, n, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(pe.data(), c.d_pix_err, n * 8, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(dp.data(), c.d_dist_pred, n * 8, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(it.data(), c.d_iters, n * 4, hipMemcpyDeviceToHost));
            pagk_ctx *own = nullptr;
            if (pagk_create(&own, 0) != PAGK_OK) return 3;
            const pagk_image ir{c.ref.data(), c.w, c.h, c.w}, ic{c.cur[(size_t)f].data(), c.w, c.h, c.w};
            const pagk_outputs ho{hun.data(), hdi.data(), hst.data(), hpe.data(), hdp.data(), nullptr, hit.data()};
            CHECK_PAGK(own, pagk_track(own, &p, &ir, &ic, c.n, c.pt_ref.data(), c.pt_init.data(), nullptr, c.status_in.data(), &ho));
            pagk_destroy(own);
            const bool same = !std::memcmp(un.data(), hun.data(), n * 8) && !std::memcmp(di.data(), hdi.data(), n * 8) &&
                              !std::memcmp(st.data(), hst.data(), n) && !std::memcmp(pe.data(), hpe.data(), n * 8) &&
                              !std::memcmp(dp.data(), hdp.data(), n * 8) && !std::memcmp(it.data(), hit.data(), n * 4);
            int tracked = 0, wanted = 0;
            double err = 0;
            for (size_t i = 0; i < n; i++) {
                wanted += c.status_in[i];
                if (!st[i]) continue;
                tracked++;
                // the scene moved by (-sx, -sy): cur(x) = ref(x + s)  =>  a point of ref at x is found in cur at x - s
                const double ex = un[2 * i] - (c.pt_ref[2 * i] - (1.3 + 0.9 * f)), ey = un[2 * i + 1] - (c.pt_ref[2 * i + 1] - (-0.8 - 0.4 * f));
                err += std::sqrt(ex * ex + ey * ey);
            }
            std::printf("frame %d camera %d (%dx%d, %d features): %d of %d tracked, mean |error| %.3f px, batched == own pagk_track: %s\n", f, j,
                        c.w, c.h, c.n, tracked, wanted, tracked ? err / tracked : 0.0, same ? "yes" : "NO");
            if (!same) return 20;
            tracked_total += tracked, feats_total += wanted;
        }
    }
    // steady state: the replayed step, timed
    const int reps = 50;
    for (int r = 0; r < 10; r++) CHECK_PAGK(lead, pagk_graph_launch(lead, graph));
    CHECK_HIP(hipStreamSynchronize(stream));
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) CHECK_PAGK(lead, pagk_graph_launch(lead, graph));
    CHECK_HIP(hipStreamSynchronize(stream));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    long long per_step = 0;
    for (Camera &c : cams)
        for (uint8_t s : c.status_in) per_step += s;
    std::printf("replayed step: %.3f ms for %lld features of %d cameras = %.1f M features/s (kernel variant %d)\n", ms, per_step, k,
                per_step / ms * 1e-3, pagk_last_variant(lead));
    std::printf("OK %lld of %lld tracked\n", tracked_total, feats_total);
    CHECK_PAGK(lead, pagk_graph_destroy(lead, graph));
    for (Camera &c : cams) pagk_destroy(c.ctx);
    (void)hipStreamDestroy(stream);
    return 0;
}
