// pagk_hip.hip -- C ABI (include/pagk.h) over the gfx950 kernels.  Host side of the boundary.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared  (see __graft_entry__.py)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/pagk.h"
#include "pagk_kernels.h"

using namespace pagk;

namespace {

constexpr int kSlots = 6;        // 0..3 for the caller, 4/5 = scratch pair of the host-buffer path
constexpr int kUserSlots = 4;

struct FrameSlot {
    int w = 0, h = 0, L = 0;
    int wrap0 = 1;                   // level-0 source was continuous (step == cols)
    int pad0 = 0;                    // min(step - cols, 2) of the level-0 source (free GetPixelValue, x == cols)
    uint8_t *u8[kMaxLevels] = {};    // u8[0] is our contiguous copy of level 0 (pitch = w)
    uint32_t *quad[kMaxLevels] = {};
    void *block = nullptr;           // one allocation for everything above
    size_t block_bytes = 0;
    bool valid = false;
};

struct FeatBuf {
    void *block = nullptr;
    void *host = nullptr;  // pinned mirror of `block` (host-buffer path: one copy in, one copy out)
    size_t bytes = 0;
    int cap = 0;
};

}  // namespace

struct pagk_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    FrameSlot slots[kSlots];
    FeatBuf feat;
    FeatBuf score;  // scratch of the host-buffer geometry scoring path
    void *quad_ws = nullptr;  // k_track_quad: iteration-invariant img1 samples, 4 * NCH * 64 floats per wave
    size_t quad_ws_bytes = 0;
    void *queue = nullptr;    // k_track_rows: the work-queue counter (256 B)
    int quad_capacity[3] = {0, 0, 0};  // resident waves of k_track_quad<2 / 4 / 7> (occupancy x CUs): the hand-over rule's "round"
    int rows_capacity[3] = {0, 0, 0};  // resident waves of k_track_rows<2 / 4 / 7> on this device (occupancy x CUs)
    int rows_waves_cap = 0;            // PAGK_ROWS_WAVES: upper bound of that grid (tests: a small grid, a long queue)
    void *susp = nullptr;     // continuation buffers: int count (256 B) | int list[n] | SuspState state[n]
    size_t susp_bytes = 0;
    int *susp_count_dev = nullptr;  // the hand-over count of the last launch that used one (in `susp` or in `lv`)
    void *lv = nullptr;       // one-level-per-wave launches: 8 sequences' counters (8 x 4096 B) | ready lists | float state[4 n]
    size_t lv_bytes = 0;
    // pagk_track_device_batch (lead context): the BatchStream array of a launch and its pinned source.  The copy to the
    // device is asynchronous, and a captured copy is replayed long after the call: a pair is therefore never rewritten
    // while something may still read it.  Direct launches take the pairs of a ring in turn (a pair is reused six calls
    // later, after waiting for the launch that used it); a capture takes pairs that pagk_graph_begin reserved for it and
    // that belong to the graph from then on (freed by pagk_graph_destroy).  Every pair holds the maximum of 64 streams.
    struct BatchDesc {
        void *host = nullptr, *dev = nullptr;
        hipEvent_t used = nullptr;   // recorded behind the launch that read `dev`
        bool recorded = false;
    };
    static constexpr int kBatchRing = 6, kBatchPerCapture = 4, kBatchMaxStreams = 64;
    BatchDesc batch_ring[kBatchRing];
    int batch_turn = 0;
    bool batch_seen = false;                 // this context has led a batched launch: captures reserve pairs
    std::vector<BatchDesc> cap_batch;        // reserved for the running capture
    int cap_batch_used = 0;
    hipEvent_t ev_batch = nullptr;  // orders a batched launch against the streams of the contexts it serves
    int *lv_error = nullptr;  // mapped host memory: a wave of such a launch gave up waiting (never expected; checked at syncs)
    int *lv_error_dev = nullptr;  // ... as the device addresses it
    hipStream_t aux_stream = nullptr;  // the live finisher's stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int susp_lone = 1;        // PAGK_SUSPEND_LONE=0: hand every feature over at the budget, not only the last of a wave
    int finisher_wgs = 16;    // PAGK_FINISHER_WGS: workgroups of the live finisher (0: sweep only)
    int finisher_polls = 20000;  // PAGK_FINISHER_POLLS: bounded wait of a finisher workgroup (~2 us per look: ~35 ms,
                                 // an order of magnitude beyond the longest launch the hand-over rule admits)
    int level_polls = 300000;  // PAGK_LEVEL_POLLS: bounded wait of a one-level-per-wave item for its ready-list entry (~1 s: far beyond any launch, also on a time-sliced device)
    int quad_budget = -1;     // iterations a feature may run in the four-features-per-wave kernel before it is handed to
                              // the latency kernel; 0: never; -1 (default): chosen per launch, see quad_budget_for().
                              // PAGK_QUAD_BUDGET overrides.
    int cus = 0;              // compute units of the device
    // hipGraph capture of the per-frame work (pagk_graph_*): while capturing, nothing may allocate and the
    // timing events are left out (an event recorded into a graph cannot be read back)
    bool capturing = false;
    static constexpr int kGraphs = 8;
    hipGraph_t graphs[kGraphs] = {};
    hipGraphExec_t graph_execs[kGraphs] = {};
    // A captured step that contains a launch with the LIVE finisher is replayed in segments (round 4): HIP replays the
    // parallel branches of one graph one after the other, so the finisher -- which must run BESIDE the throughput kernel --
    // is not a node: the capture is closed in front of that kernel, the finisher is remembered as a plain launch on the
    // auxiliary stream, the capture reopens.  graph_execs[k] is the LAST segment; pre_segs[k] what runs before it.
    struct GraphSeg {
        enum Kind { GRAPH, FINISHER, JOIN } kind = GRAPH;
        hipGraph_t g = nullptr;
        hipGraphExec_t ex = nullptr;
        const void *fn = nullptr;   // FINISHER: k_track_resume_live<..>, its arguments and launch shape
        TrackArgs args;
        int grid = 0;
        size_t lds = 0;
    };
    std::vector<GraphSeg> pre_segs[kGraphs];
    std::vector<BatchDesc> graph_batch[kGraphs];   // the descriptor pairs graph k's batched launches read
    std::vector<GraphSeg> cap_segs;   // segments closed so far in the capture that is open
    hipEvent_t ev_trk[2] = {}, ev_pyr[2] = {};
    bool trk_timed = false, pyr_timed = false;
    int kernel = 0;
    bool last_handover = false;  // the last tracking launch used the hand-over (pagk_last_handover)
    int concurrency = 1;    // pagk_set_concurrency: contexts like this one running at the same time on the device
    int last_variant = -1;  // variant the last tracking launch used (pagk_last_variant)
    // auto-selection thresholds, measured on MI355X at h = 10 (tools/sweep_n.py, profiles/r03_sweep_n.log): after round
    // 3's instruction diet the 4-wave DPP kernel is the fastest up to ~5000 features (it used to lose to the 2-wave MFMA
    // variant from 2500; that variant no longer wins at any size and is selected explicitly only), one wave per feature
    // wins between ~5000 and ~7000, four features per wave from there.  Later in round 3: four features per wave with
    // one pyramid level per wave (variant 7) is the fastest from ~6000 features for a context alone on the device
    // (profiles/r03_levels_sweep.log); contexts that share the device (pagk_set_concurrency) keep the sequence above.
    int mfma_min_features = 0x7fffffff;  // PAGK_MFMA_MIN
    int wave_min_features = 6000;        // PAGK_WAVE_MIN (5000 until the 4-wave kernel had its build for five workgroups per CU)
    int quad_min_features = 7000;        // PAGK_QUAD_MIN: four features per wave (pagk_quad_kernel.h)
    int block5_min_features = 2500;      // PAGK_BLOCK5_MIN: the 4-wave kernel in its five-workgroups-per-CU build (h = 10)
    int prio_k = 4;                      // PAGK_PRIO_K: iterations per pyramid level beyond which a 4-wave workgroup counts as behind (pagk_prio.h); 0 = rule off
    unsigned long long *prio_stats = nullptr;  // PAGK_PRIO_K=auto: device block [iterations, feature-levels (u64 each), K (int)]
    bool block5_window = true;           // ... also for launches that only five workgroups per CU hold in one round (off when PAGK_BLOCK5_MIN is set)
    int levels_min_features = 6000;      // PAGK_LEVELS_MIN: ... one level per wave (a context alone on the device)
    int levels_shift = 0;                // PAGK_LEVELS_XCD_SHIFT (tests): waves start with another XCD's ticket sequence
    bool levels_shared = false;          // PAGK_LEVELS_SHARED=1 (measurement): ... also for contexts that share the device
    bool unfused_pyramid = false;  // PAGK_UNFUSED_PYRAMID=1: level-by-level launches (cross-check)
    char err[256] = {0};
};

namespace {

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call,     \
                     hipGetErrorString(e_));                                                           \
            return e_ == hipErrorOutOfMemory ? PAGK_E_NOMEM : PAGK_E_HIP;                              \
        }                                                                                              \
    } while (0)

// Entry points that copy from / to host memory or synchronise cannot be part of a graph capture.
#define NOT_WHILE_CAPTURING(ctx, what)                                                                 \
    do {                                                                                               \
        if ((ctx)->capturing) {                                                                        \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s is not capturable: use the *_device entry points between pagk_graph_begin and pagk_graph_end", what); \
            return PAGK_E_ARG;                                                                         \
        }                                                                                              \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int level_dims(int w, int h, int L, int *lw, int *lh)
{
    lw[0] = w;
    lh[0] = h;
    for (int l = 1; l < L; l++) {
        // src/patch_match.cpp:69  cv::Size(cols * 0.5, rows * 0.5)
        lw[l] = (int)(lw[l - 1] * 0.5);
        lh[l] = (int)(lh[l - 1] * 0.5);
        if (lw[l] < 1 || lh[l] < 1) return PAGK_E_ARG;
    }
    return PAGK_OK;
}

// Is this context's work being recorded rather than executed?  Its own capture (pagk_graph_begin), or a capture of
// the stream it was switched to by somebody else -- another context of a batch (pagk_track_device_batch), the host
// application's own hipStreamBeginCapture.  Nothing may allocate then, and timing events are left out.
bool in_capture(pagk_ctx *ctx)
{
    if (ctx->capturing) return true;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(ctx->stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone;
}

// Close the open capture segment of pagk_graph_begin (what was recorded so far becomes one instantiated graph in
// ctx->cap_segs) and open the next one.  Used around a launch whose finisher must not be a graph node.
int capture_split(pagk_ctx *ctx)
{
    hipGraph_t g = nullptr;
    HIPCHK(ctx, hipStreamEndCapture(ctx->stream, &g));
    if (g) {
        pagk_ctx::GraphSeg seg;
        hipError_t e = hipGraphInstantiate(&seg.ex, g, nullptr, nullptr, 0);
        if (e != hipSuccess) {
            (void)hipGraphDestroy(g);
            ctx->capturing = false;
            snprintf(ctx->err, sizeof(ctx->err), "hipGraphInstantiate (segment) -> %s", hipGetErrorString(e));
            return PAGK_E_HIP;
        }
        seg.g = g;
        ctx->cap_segs.push_back(seg);
    }
    hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        ctx->capturing = false;
        snprintf(ctx->err, sizeof(ctx->err), "hipStreamBeginCapture (next segment) -> %s", hipGetErrorString(e));
        return PAGK_E_HIP;
    }
    return PAGK_OK;
}
void destroy_segs(std::vector<pagk_ctx::GraphSeg> &v)
{
    for (auto &sg : v) {
        if (sg.ex) (void)hipGraphExecDestroy(sg.ex);
        if (sg.g) (void)hipGraphDestroy(sg.g);
    }
    v.clear();
}

int batch_desc_alloc(pagk_ctx *ctx, pagk_ctx::BatchDesc &d)
{
    // (one size for both users: the BatchStream array of a tracking launch, the PyrBatchEntry array of a pyramid launch)
    static_assert(sizeof(PyrBatchEntry) <= sizeof(BatchStream), "a descriptor pair is sized by the larger record");
    const size_t bytes = (size_t)pagk_ctx::kBatchMaxStreams * sizeof(BatchStream);
    HIPCHK(ctx, hipMalloc(&d.dev, bytes));
    HIPCHK(ctx, hipHostMalloc(&d.host, bytes, hipHostMallocDefault));
    HIPCHK(ctx, hipEventCreateWithFlags(&d.used, hipEventDisableTiming));
    d.recorded = false;
    return PAGK_OK;
}
void batch_desc_free(pagk_ctx::BatchDesc &d)
{
    if (d.dev) (void)hipFree(d.dev);
    if (d.host) (void)hipHostFree(d.host);
    if (d.used) (void)hipEventDestroy(d.used);
    d = pagk_ctx::BatchDesc();
}
void batch_desc_free(std::vector<pagk_ctx::BatchDesc> &v)
{
    for (auto &d : v) batch_desc_free(d);
    v.clear();
}

// The descriptor pair of the batched launch `lead` is about to issue: the next of the ring (after waiting for the launch
// that used it last), or -- inside lead's own capture -- the next of those pagk_graph_begin reserved.  nullptr: *rc, lead->err.
pagk_ctx::BatchDesc *batch_desc_take(pagk_ctx *lead, int *rc)
{
    *rc = PAGK_OK;
    if (in_capture(lead)) {
        if (!lead->capturing || lead->cap_batch_used >= (int)lead->cap_batch.size()) {
            snprintf(lead->err, sizeof(lead->err), "a batched launch inside a graph capture needs descriptor buffers reserved by pagk_graph_begin of "
                     "ctxs[0]: issue the batched call once before capturing, capture through pagk_graph_begin(ctxs[0]), at most %d batched calls per capture",
                     pagk_ctx::kBatchPerCapture);
            *rc = PAGK_E_ARG;
            return nullptr;
        }
        return &lead->cap_batch[lead->cap_batch_used++];
    }
    pagk_ctx::BatchDesc *desc = &lead->batch_ring[lead->batch_turn];
    lead->batch_turn = (lead->batch_turn + 1) % pagk_ctx::kBatchRing;
    if (!desc->dev) {
        if ((*rc = batch_desc_alloc(lead, *desc)) != PAGK_OK) {
            batch_desc_free(*desc);
            return nullptr;
        }
    } else if (desc->recorded) {
        if (hipEventSynchronize(desc->used) != hipSuccess) {   // (the launch a ring's length ago)
            snprintf(lead->err, sizeof(lead->err), "hipEventSynchronize of a batch descriptor's last user failed");
            *rc = PAGK_E_HIP;
            return nullptr;
        }
    }
    lead->batch_seen = true;
    return desc;
}
// ... and behind that launch: from here on the pair is busy until the launch is over
int batch_desc_used(pagk_ctx *lead, pagk_ctx::BatchDesc *desc)
{
    if (in_capture(lead)) return PAGK_OK;   // (the graph owns it)
    HIPCHK(lead, hipEventRecord(desc->used, lead->stream));
    desc->recorded = true;
    return PAGK_OK;
}

int slot_reserve(pagk_ctx *ctx, FrameSlot &s, int w, int h, int L)
{
    int lw[kMaxLevels], lh[kMaxLevels];
    int rc = level_dims(w, h, L, lw, lh);
    if (rc) return rc;
    size_t total = 0, off_u8[kMaxLevels], off_q[kMaxLevels];
    for (int l = 0; l < L; l++) {
        off_u8[l] = total;
        total = align_up(total + (size_t)lw[l] * lh[l], 256);
    }
    for (int l = 0; l < L; l++) {
        off_q[l] = total;
        total = align_up(total + (size_t)lw[l] * lh[l] * 4, 256);
    }
    if (total > s.block_bytes) {
        if (in_capture(ctx)) {  // run the same calls once before pagk_graph_begin so that nothing allocates here
            snprintf(ctx->err, sizeof(ctx->err), "frame slot would have to be (re)allocated during graph capture");
            return PAGK_E_ARG;
        }
        if (s.block) HIPCHK(ctx, hipFree(s.block));
        s.block = nullptr;
        s.block_bytes = 0;
        HIPCHK(ctx, hipMalloc(&s.block, total));
        s.block_bytes = total;
    }
    for (int l = 0; l < L; l++) {
        s.u8[l] = static_cast<uint8_t *>(s.block) + off_u8[l];
        s.quad[l] = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(s.block) + off_q[l]);
    }
    s.w = w;
    s.h = h;
    s.L = L;
    return PAGK_OK;
}

// Arguments of the single-launch pyramid (k_pyramid_fused / the trailing blocks of k_track_block_pyr) for a
// reserved slot with at most 4 levels and even parents; returns the number of 256-thread blocks.
int make_pyr_args(const FrameSlot &s, const uint8_t *src0, int64_t pitch0, int wrap0, PyrArgs *out)
{
    int lw[kMaxLevels], lh[kMaxLevels];
    level_dims(s.w, s.h, s.L, lw, lh);
    PyrArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.src = src0;
    pa.pitch = pitch0;
    pa.wrap0 = wrap0;
    pa.n_levels = s.L;
    int nb = 0;
    for (int l = 0; l < 4; l++) {
        pa.first_block[l] = nb;
        if (l < s.L) {
            pa.cols[l] = lw[l];
            pa.rows[l] = lh[l];
            pa.u8[l] = s.u8[l];
            pa.quad[l] = s.quad[l];
            nb += (lw[l] * lh[l] + 255) / 256;
        }
    }
    pa.first_block[4] = nb;
    for (int l = s.L; l < 4; l++) pa.first_block[l] = nb;  // empty ranges for absent levels
    *out = pa;
    return nb;
}

bool pyramid_fusable(const FrameSlot &s)
{
    int lw[kMaxLevels], lh[kMaxLevels];
    level_dims(s.w, s.h, s.L, lw, lh);
    bool all_even = s.L <= 4;
    for (int l = 0; l + 1 < s.L; l++) all_even = all_even && !(lw[l] & 1) && !(lh[l] & 1);
    return all_even;
}

// CreatePyramids (src/patch_match.cpp:61-76) + tap packing, from a level-0 image that is
// already on the device at (src0, pitch0).
int slot_build(pagk_ctx *ctx, FrameSlot &s, const uint8_t *src0, int64_t pitch0, int wrap0)
{
    s.pad0 = wrap0 ? 0 : 2;  // callers that know the source's step refine this (frame_upload_any, pagk_frame_set_device)
    int lw[kMaxLevels], lh[kMaxLevels];
    level_dims(s.w, s.h, s.L, lw, lh);
    dim3 blk(32, 8);
    if (ctx->ev_pyr[0] && !in_capture(ctx)) HIPCHK(ctx, hipEventRecord(ctx->ev_pyr[0], ctx->stream));
    // the fused kernel re-derives every level from level 0 by nested 2x2 means: valid while every parent is
    // even in both dimensions (the exact-2x case of cv::resize); otherwise level by level
    bool all_even = true;
    for (int l = 0; l + 1 < s.L; l++) all_even = all_even && !(lw[l] & 1) && !(lh[l] & 1);
    if (s.L <= 4 && !ctx->unfused_pyramid && all_even) {
        PyrArgs pa;
        const int nb = make_pyr_args(s, src0, pitch0, wrap0, &pa);
        hipLaunchKernelGGL(k_pyramid_fused, dim3(nb), dim3(256), 0, ctx->stream, pa);
        HIPCHK(ctx, hipGetLastError());
        if (ctx->ev_pyr[1] && !in_capture(ctx)) HIPCHK(ctx, hipEventRecord(ctx->ev_pyr[1], ctx->stream));
        if (!in_capture(ctx)) ctx->pyr_timed = true;
        s.wrap0 = wrap0;
        s.valid = true;
        return PAGK_OK;
    }
    const uint8_t *src = src0;
    int64_t pitch = pitch0;
    for (int l = 0; l < s.L; l++) {
        if (l > 0) {
            dim3 grd((lw[l] + 31) / 32, (lh[l] + 7) / 8);
            if (!(lw[l - 1] & 1) && !(lh[l - 1] & 1))
                hipLaunchKernelGGL(k_pyr_down, grd, blk, 0, ctx->stream, src, pitch, lw[l], lh[l], s.u8[l]);
            else
                hipLaunchKernelGGL(k_pyr_down_linear, grd, blk, 0, ctx->stream, src, pitch, lw[l - 1], lh[l - 1], lw[l],
                                   lh[l], 1. / ((double)lw[l] / lw[l - 1]), 1. / ((double)lh[l] / lh[l - 1]), s.u8[l]);
            src = s.u8[l];
            pitch = lw[l];
        }
        dim3 grd((lw[l] + 31) / 32, (lh[l] + 7) / 8);
        hipLaunchKernelGGL(k_build_quads, grd, blk, 0, ctx->stream, src, pitch, lw[l], lh[l], l == 0 ? wrap0 : 1,
                           s.quad[l]);
    }
    HIPCHK(ctx, hipGetLastError());
    if (ctx->ev_pyr[1] && !in_capture(ctx)) HIPCHK(ctx, hipEventRecord(ctx->ev_pyr[1], ctx->stream));
    if (!in_capture(ctx)) ctx->pyr_timed = true;
    s.wrap0 = wrap0;
    s.valid = true;
    return PAGK_OK;
}

int check_params(const pagk_params *p)
{
    if (!p) return PAGK_E_ARG;
    if (p->half_patch < 1 || p->half_patch > PAGK_MAX_HALF_PATCH) return PAGK_E_ARG;
    if (p->iterations < 0 || p->pyramids < 1 || p->pyramids > PAGK_MAX_PYRAMIDS) return PAGK_E_ARG;
    if (p->inverse) return PAGK_E_UNSUPPORTED;        // src/patch_match.cpp:220 "not support yet"
    if (p->solver_variant & ~(SV_LOWER_SEQ | SV_UPPER_TREE | SV_NORM_SEQ | SV_LLT_RECIP | SV_PIVOT_TREE)) return PAGK_E_ARG;
    return PAGK_OK;
}

void fill_level(DevLevel &d, const FrameSlot &s, int l)
{
    int w = s.w, h = s.h;
    for (int k = 0; k < l; k++) {
        w = (int)(w * 0.5);
        h = (int)(h * 0.5);
    }
    d.quad = s.quad[l];
    d.cols = w;
    d.rows = h;
    d.fcols = (float)w;
    d.frows = (float)h;
    d.fcols_m1 = (float)(w - 1);
    d.frows_m1 = (float)(h - 1);
}

// pyr / pyr_blocks / pyr_done: optionally, another slot's pyramid to be built by trailing workgroups of the
// tracking launch (k_track_block_pyr).  Honoured when the 4-wave kernel is the one selected; *pyr_done tells the
// caller whether it was (otherwise the caller launches the pyramid itself).
// Hand-over budget of a four-features-per-wave launch of `waves` wavefronts.  The hand-over pays where the launch ends
// with an exposed tail -- between half a round and 1.25 rounds of resident waves (quad_capacity: the kernel's occupancy
// on this device x its CU count; 16 x 256 on MI355X): a handful of features
// with 3-5x the mean iteration count would otherwise each keep a wave alive long after the rest has finished
// (configs[3], 20000 features: -6 %; 8000: -4 %).  With a fuller second round the first round's stragglers are already
// hidden behind it and the finisher only displaces throughput waves (30000: +8 %), and a context that shares the device
// (pagk_set_concurrency) has other launches to fill its tail.  profiles/r02_ab_runs.md.
// Resident waves of k_track_quad<NCH> on this device: the kernel's own occupancy (registers, its 10000 B of LDS) times
// the CU count, asked once per context and patch size.
// A wave of a one-level-per-wave launch that gave up waiting for the level above (never expected: the wait is on a wave
// that started earlier) leaves results that must not be used.
int lv_check(pagk_ctx *ctx)
{
    if (ctx->lv_error && *static_cast<volatile int *>(ctx->lv_error) != 0) {
        *static_cast<volatile int *>(ctx->lv_error) = 0;
        snprintf(ctx->err, sizeof(ctx->err), "a wave of the level-by-level tracking launch gave up waiting for the level above");
        return PAGK_E_HIP;
    }
    return PAGK_OK;
}

// Library-owned device buffers that captured graphs point into (quad_ws, lv, susp) may only be reallocated while no
// instantiated graph of this context is alive: the caller destroys its graphs (pagk_graph_destroy), or runs the larger
// launch once BEFORE capturing, as include/pagk.h asks.
int no_live_graphs(pagk_ctx *ctx, const char *what)
{
    for (int k = 0; k < pagk_ctx::kGraphs; k++)
        if (ctx->graph_execs[k]) {
            snprintf(ctx->err, sizeof(ctx->err), "%s would have to grow while graph %d of this context is alive (its nodes hold the old "
                     "pointers): destroy the graph first, or run the largest launch once before capturing", what, k);
            return PAGK_E_ARG;
        }
    return PAGK_OK;
}

int quad_capacity(pagk_ctx *ctx, int half)
{
    const int slot = half == 5 ? 0 : (half == 7 ? 1 : 2);
    if (ctx->quad_capacity[slot] == 0) {
        int per_cu = 0;
        hipError_t e = half == 5   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_track_quad<2>, 64, 0)
                       : half == 7 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_track_quad<4>, 64, 0)
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_track_quad<7>, 64, 0);
        if (e != hipSuccess || per_cu <= 0) per_cu = 16;
        ctx->quad_capacity[slot] = per_cu * (ctx->cus > 0 ? ctx->cus : 256);
    }
    return ctx->quad_capacity[slot];
}

int quad_budget_for(pagk_ctx *ctx, int waves, int iterations, int levels, int half)
{
    if (ctx->quad_budget >= 0) return ctx->quad_budget;
    // a feature can run iterations x levels iterations at most: with the reference's own call site (10 x 3) nothing
    // runs long enough past the budget for the hand-over to pay for its list reset, finisher launch and sweep
    if (iterations * levels < 60) return 0;
    const long long cap = quad_capacity(ctx, half);  // one "round" of resident waves
    // (beyond 1.25 rounds it loses, also when restricted to the launch's drain phase -- "a feature may leave only once
    // every wave has been dispatched": +8 % at 30000 features, +5 % at 60000, profiles/r03_finisher_sweep_drain_rule.log)
    const bool exposed_tail = 100ll * waves > 45 * cap && 100ll * waves <= 125 * cap;
    // (not inside a graph capture: the replayed graph runs its two branches one after the other, measured, and a
    // finisher that starts after the throughput kernel is the plain sweep: +13 %)
    // (not inside a capture of the stream that is not ours -- the host application's own hipStreamBeginCapture: a replayed
    // graph runs a finisher branch AFTER the throughput kernel, +13 %.  pagk_graph_begin's capture is replayed in
    // segments with the finisher beside the kernel, like a direct launch)
    if (ctx->concurrency != 1 || !exposed_tail || (in_capture(ctx) && !ctx->capturing)) return 0;
    return 20;
}

// The same for a one-level-per-wave launch (variant 7) of `quads` four-feature groups.  Its waves are short, so what
// the hand-over removes is not an idle tail but the critical path itself: the launch cannot end before its slowest
// feature has run its levels one after the other at the throughput kernel's ~11 us per iteration.  Measured with
// budgets 0 / 14 / 20 / 26 (profiles/r03_levels_handover.log): 20 gains 30 % at 10000-12000 features of configs[3],
// 13-20 % at 16000-24000, 3-7 % on the easier 752x480 pair, nothing from 30000 on (no loss either), and costs 3-6 %
// at 6000; 14 hands over ten times as many features and loses everywhere.  On from 0.45 rounds of resident waves.
// Inside a graph capture (the replayed graph runs a finisher branch AFTER the throughput kernel) the hand-over is
// the sweep alone -- `*live` false: the stragglers leave the throughput waves early and are finished by the 4-wave
// kernel behind them.  That still pays while the launch is chain-bound (12000 features of configs[3] 577 us instead of
// 714, 20000 features 699 instead of 769) and costs from about 1.3 rounds on (30000 features: +7.5 %,
// profiles/r03_levels_sweep_only.log): in a capture, up to 1.25 rounds.
int levels_budget_for(pagk_ctx *ctx, int quads, int iterations, int levels, int half, bool *live)
{
    // a capture of the stream that is not pagk_graph_begin's (the host application's own): its replay runs a finisher
    // branch after the kernel, so there the hand-over is the sweep alone, up to 1.25 rounds.  Our own capture is replayed
    // in segments with the finisher beside the kernel (launch_track): the direct launch's rule.
    const bool foreign = in_capture(ctx) && !ctx->capturing;
    *live = !foreign || ctx->quad_budget >= 0;   // (a forced budget keeps the parallel branch: tests)
    if (ctx->quad_budget >= 0) return ctx->quad_budget;
    if (iterations * levels < 60 || ctx->concurrency != 1) return 0;
    const long long cap = quad_capacity(ctx, half);
    if (100ll * quads <= 45 * cap) return 0;
    if (foreign && 100ll * quads > 125 * cap) return 0;
    return 20;
}

// What a launch takes from pagk_params: the constructor's constants (src/patch_match.cpp:48-57) and the camera model.
void fill_param_args(TrackArgs &a, const pagk_params *p)
{
    a.half = p->half_patch;
    a.iterations = p->iterations;
    a.has_gyro = p->has_gyro_predict_initial;
    a.illum = p->consider_illumination;
    a.use_affine = p->consider_affine;
    a.penalty = p->regularization_penalty;
    a.calc_ncc = p->calculate_ncc;
    a.solver = p->solver_variant;
    float invlog = p->inv_log_max_dist != 0.0f ? p->inv_log_max_dist
                                               : pagk_inv_log_max_dist(p->alpha, p->max_distance);
    a.lam_invlog = p->lambda * invlog;           // :305  mLambda * mInvLogMaxDist (float)
    a.lam_invlog_alpha = a.lam_invlog * p->alpha; // :307  ... * mAlpha (float)
    a.alpha = p->alpha;
    // :57  1.0f / (2.0f*h + 1.0f) / (2.0f*h + 1.0f), float, stored in a double
    a.win_size_inv = (double)(1.0f / (2.0f * p->half_patch + 1.0f) / (2.0f * p->half_patch + 1.0f));
    a.distort_on = p->dist_coef[0] != 0.0f;  // :410
    a.fx = p->fx, a.fy = p->fy, a.cx = p->cx, a.cy = p->cy;
    a.fx_inv = (float)(1.0 / (double)p->fx);  // src/utils.cpp:53
    a.fy_inv = (float)(1.0 / (double)p->fy);
    a.k1 = p->dist_coef[0], a.k2 = p->dist_coef[1], a.p1 = p->dist_coef[2], a.p2 = p->dist_coef[3];
    a.k3 = p->n_dist_coef == 5 ? p->dist_coef[4] : 0.0f;
}

int launch_track(pagk_ctx *ctx, const pagk_params *p, const FrameSlot &sr, const FrameSlot &sc, int n,
                 const float *d_pt_ref, const float *d_pt_init, const float *d_affine, const uint8_t *d_status,
                 const pagk_outputs *o, const PyrArgs *pyr = nullptr, int pyr_blocks = 0, bool *pyr_done = nullptr)
{
    if (pyr_done) *pyr_done = false;
    TrackArgs a;
    memset(&a, 0, sizeof a);
    a.n_levels = p->pyramids;
    for (int l = 0; l < p->pyramids; l++) {
        fill_level(a.l1[l], sr, l);
        fill_level(a.l2[l], sc, l);
        // :66,:73  mvScales[i] = mvScales[i-1] * mPyramidScale  (float * double -> float)
        a.scales[l] = l == 0 ? 1.0f : (float)((double)a.scales[l - 1] * 0.5);
    }
    a.n = n;
    a.pt_ref = d_pt_ref;
    a.pt_init = d_pt_init;
    a.affine = d_affine;
    a.status_in = d_status;
    a.pt_un = o->pt_un;
    a.pt_dist = o->pt_dist;
    a.status = o->status;
    a.pix_err = o->pix_err;
    a.dist_pred = o->dist_pred;
    a.ncc = o->ncc;
    a.iters = o->iters;
#if defined(PAGK_STAMPS) || defined(PAGK_COUNT_REDO) || defined(PAGK_TIC)
    a.dbg = reinterpret_cast<unsigned long long *>(getenv("PAGK_DBG_PTR") ? strtoull(getenv("PAGK_DBG_PTR"), nullptr, 0) : 0ull);
#endif
    fill_param_args(a, p);
    a.prio_k = ctx->prio_k;
    a.prio_stats = ctx->prio_stats;
    a.prio_kbuf = ctx->prio_stats ? reinterpret_cast<const int *>(ctx->prio_stats + 2) : nullptr;

    if (ctx->ev_trk[0] && !in_capture(ctx)) HIPCHK(ctx, hipEventRecord(ctx->ev_trk[0], ctx->stream));
    if (n > 0) {
        const int Pm = (2 * a.half + 1) * (2 * a.half + 1);
        // MFMA variant: instantiated for the common patch sizes; chosen explicitly (kernel 2) or,
        // by default, when the launch has more features than can be resident at once
        const bool mfma_ok = a.half == 5 || a.half == 7 || a.half == 10;
        // four features per wave: no NCC epilogue of its own (calc_ncc launches run the one-wave-per-feature variant)
        // ... and with the four rows of a wave independent + a work queue (pagk_rows_kernel.h)
        ctx->last_handover = false;
        // the reference's defaults (no regularisation penalty, solver_variant 0) run kernels in which both are
        // compile-time facts (LEAN); everything else runs the generic instantiations
        const bool lean = !a.penalty && a.solver == 0;
        const long long n_sel = (long long)n * ctx->concurrency;  // what the automatic thresholds are applied to
#ifdef PAGK_ALL_VARIANTS
        const bool use_rows = mfma_ok && !a.calc_ncc && a.iterations >= 1 && ctx->kernel == 6;
#else
        const bool use_rows = false;   // (variant (e) is not in this build: pagk_set_kernel(ctx, 6) was refused)
#endif
        // four features per wave, one level per wave (needs more than one level to differ from the quad kernel)
        const bool use_levels = mfma_ok && !a.calc_ncc && p->pyramids >= 2 && ctx->lv_error &&
                                (ctx->kernel == 7 || (ctx->kernel == 0 && (ctx->concurrency == 1 || ctx->levels_shared) &&
                                                      n_sel >= ctx->levels_min_features));
        const bool quad_like = ctx->kernel == 5 || ctx->kernel == 6 || ctx->kernel == 7;
        const bool use_quad = !use_rows && !use_levels && mfma_ok && !a.calc_ncc && (quad_like || (ctx->kernel == 0 && n_sel >= ctx->quad_min_features));
        const bool use_wave = !use_quad && !use_rows && !use_levels && mfma_ok && (ctx->kernel == 3 || quad_like || (ctx->kernel == 0 && n_sel >= ctx->wave_min_features));
#ifdef PAGK_ALL_VARIANTS
        const bool use_mfma = !use_wave && mfma_ok && (ctx->kernel == 2 || (ctx->kernel == 0 && n_sel >= ctx->mfma_min_features));
#else
        const bool use_mfma = false;   // (variant (b) is not in this build)
#endif
        ctx->last_variant = ctx->kernel == 1 ? 1 : use_levels ? 7 : use_rows ? 6 : (use_quad ? 5 : (use_wave ? 3 : ((ctx->kernel == 4 && mfma_ok) ? 4 : (use_mfma ? 2 : 0))));
        if (ctx->kernel == 1) {
            hipLaunchKernelGGL(k_track_thread, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, a);
#ifdef PAGK_ALL_VARIANTS
        } else if (use_rows) {
            const int nch = (Pm + 63) / 64;
            const int slot = a.half == 5 ? 0 : (a.half == 7 ? 1 : 2);
            if (ctx->rows_capacity[slot] == 0) {
                int per_cu = 0, cus = 0;
                hipError_t oe = a.half == 5   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_track_rows<2>, 64, 0)
                                : a.half == 7 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_track_rows<4>, 64, 0)
                                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_track_rows<7>, 64, 0);
                HIPCHK(ctx, oe);
                HIPCHK(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
                ctx->rows_capacity[slot] = per_cu * cus > 0 ? per_cu * cus : 1024;
            }
            // a resident grid: every wave starts at once, rows pull the features past 4 * grid from the queue
            int waves = (n + 3) / 4 < ctx->rows_capacity[slot] ? (n + 3) / 4 : ctx->rows_capacity[slot];
            if (ctx->rows_waves_cap > 0 && waves > ctx->rows_waves_cap) waves = ctx->rows_waves_cap;
            const size_t need = (size_t)waves * 4 * nch * 64 * sizeof(float);
            if (need > ctx->quad_ws_bytes) {
                if (in_capture(ctx)) {
                    snprintf(ctx->err, sizeof(ctx->err), "the row kernel's workspace would have to be (re)allocated during graph capture");
                    return PAGK_E_ARG;
                }
                if (int gr = no_live_graphs(ctx, "the row kernel's workspace")) return gr;
                if (ctx->quad_ws) HIPCHK(ctx, hipFree(ctx->quad_ws));
                ctx->quad_ws = nullptr;
                ctx->quad_ws_bytes = 0;
                HIPCHK(ctx, hipMalloc(&ctx->quad_ws, need));
                ctx->quad_ws_bytes = need;
            }
            a.ws = static_cast<float *>(ctx->quad_ws);
            a.queue = static_cast<int *>(ctx->queue);
            HIPCHK(ctx, hipMemsetAsync(a.queue, 0, 4, ctx->stream));
            auto launch = [&](auto kern) -> hipError_t {
                hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, ctx->stream, a);
                return hipGetLastError();
            };
            hipError_t e = hipErrorInvalidValue;
            if (a.half == 5) e = launch(k_track_rows<2>);
            else if (a.half == 7) e = launch(k_track_rows<4>);
            else if (a.half == 10) e = launch(k_track_rows<7>);
            HIPCHK(ctx, e);
#endif
        } else if (use_quad || use_levels) {
            // four features per wave (pagk_quad_kernel.h): whole features per wave, or -- LEVELS -- one pyramid level per
            // wave (pyramids x ceil(n / 4) waves)
            const int nch = (Pm + 63) / 64, nq = (n + 3) / 4, waves = use_levels ? nq * p->pyramids : nq;
            const size_t need = (size_t)waves * 4 * nch * 64 * sizeof(float);
            const size_t ready_bytes = use_levels ? align_up((size_t)(p->pyramids - 1) * 8 * ((nq + 7) / 8) * 4, 256) : 0;
            // one-level-per-wave launches keep everything that must be zero before the launch in ONE block (one memset):
            // counters | ready lists | hand-over count | hand-over list; then the two state arrays
            const size_t susp_zero = 256 + align_up((size_t)n * 4, 256);
            const size_t need_lv = use_levels ? 32768 + ready_bytes + susp_zero + (size_t)n * 16 + (size_t)n * sizeof(SuspState) : 0;
            if (need > ctx->quad_ws_bytes || need_lv > ctx->lv_bytes) {
                if (in_capture(ctx)) {
                    snprintf(ctx->err, sizeof(ctx->err), "the quad kernel's workspace would have to be (re)allocated during graph capture");
                    return PAGK_E_ARG;
                }
                // an instantiated graph keeps the old pointers in its nodes (memset, kernel arguments): replaying it after
                // the buffers moved would write freed memory
                if (int gr = no_live_graphs(ctx, "the quad kernel's workspace")) return gr;
                if (need > ctx->quad_ws_bytes) {
                    if (ctx->quad_ws) HIPCHK(ctx, hipFree(ctx->quad_ws));
                    ctx->quad_ws = nullptr;
                    ctx->quad_ws_bytes = 0;
                    HIPCHK(ctx, hipMalloc(&ctx->quad_ws, need));
                    ctx->quad_ws_bytes = need;
                }
                if (need_lv > ctx->lv_bytes) {
                    if (ctx->lv) HIPCHK(ctx, hipFree(ctx->lv));
                    ctx->lv = nullptr;
                    ctx->lv_bytes = 0;
                    HIPCHK(ctx, hipMalloc(&ctx->lv, need_lv));
                    ctx->lv_bytes = need_lv;
                }
            }
            a.ws = static_cast<float *>(ctx->quad_ws);
            a.susp_polls = ctx->finisher_polls;
            if (use_levels) {
                uint8_t *lb = static_cast<uint8_t *>(ctx->lv);
                a.queue = reinterpret_cast<int *>(lb);
                a.lv_ready = reinterpret_cast<int *>(lb + 32768);
                a.lv_state = reinterpret_cast<float *>(lb + 32768 + ready_bytes + susp_zero);
                a.lv_error = ctx->lv_error_dev;
                a.lv_polls = ctx->level_polls;
                a.lv_shift = ctx->levels_shift;
            }
            // continuation buffers; the hand-over needs the 4-wave kernel's LDS (<= 48 KB at these patch sizes)
            bool live_ok = true;
            const int budget = use_levels ? levels_budget_for(ctx, nq, p->iterations, p->pyramids, a.half, &live_ok)
                                          : quad_budget_for(ctx, nq, p->iterations, p->pyramids, a.half);
            const bool handover = budget > 0;
            ctx->last_handover = handover;
            if (use_levels) {
                uint8_t *lb = static_cast<uint8_t *>(ctx->lv), *sb = lb + 32768 + ready_bytes;
                if (handover) {
                    a.iter_budget = budget;
                    a.susp_count = reinterpret_cast<int *>(sb);
                    a.susp_list = reinterpret_cast<int *>(sb + 256);
                    a.susp_state = reinterpret_cast<SuspState *>(sb + susp_zero + (size_t)n * 16);
                    a.susp_waves = nq;   // the waves that report their end: a quad's last-level wave
                    a.susp_lone = ctx->susp_lone;
                    ctx->susp_count_dev = a.susp_count;
                }
                HIPCHK(ctx, hipMemsetAsync(lb, 0, 32768 + ready_bytes + (handover ? 256 + (size_t)n * 4 : 0), ctx->stream));
            } else if (handover) {
                const size_t need_s = 256 + align_up((size_t)n * 4, 256) + (size_t)n * sizeof(SuspState);
                if (need_s > ctx->susp_bytes) {
                    if (in_capture(ctx)) {
                        snprintf(ctx->err, sizeof(ctx->err), "the continuation buffers would have to be (re)allocated during graph capture");
                        return PAGK_E_ARG;
                    }
                    if (int gr = no_live_graphs(ctx, "the continuation buffers")) return gr;
                    if (ctx->susp) HIPCHK(ctx, hipFree(ctx->susp));
                    ctx->susp = nullptr;
                    ctx->susp_bytes = 0;
                    HIPCHK(ctx, hipMalloc(&ctx->susp, need_s));
                    ctx->susp_bytes = need_s;
                }
                uint8_t *sb = static_cast<uint8_t *>(ctx->susp);
                a.iter_budget = budget;
                a.susp_count = reinterpret_cast<int *>(sb);
                a.susp_list = reinterpret_cast<int *>(sb + 256);
                a.susp_state = reinterpret_cast<SuspState *>(sb + 256 + align_up((size_t)n * 4, 256));
                a.susp_waves = nq;   // the waves that report their end: all of them
                a.susp_lone = ctx->susp_lone;
                ctx->susp_count_dev = a.susp_count;
                HIPCHK(ctx, hipMemsetAsync(sb, 0, 256 + (size_t)n * 4, ctx->stream));  // counters and list
            }
            // the live finisher runs beside the throughput kernel, on the context's auxiliary stream (inside a graph
            // capture the auxiliary stream joins the capture through the fork event: a parallel branch of the graph)
            const bool live = handover && live_ok && ctx->finisher_wgs > 0 && ctx->aux_stream;
            // inside pagk_graph_begin's capture the finisher is not a node of the graph but a launch of its own between
            // two segments of it (see pagk_ctx::GraphSeg); PAGK_GRAPH_BRANCH=1 keeps the old parallel-branch form (tests)
            const bool segmented = live && ctx->capturing && !getenv("PAGK_GRAPH_BRANCH");
            int fin_seg = -1;
            if (segmented) {
                if (int sr = capture_split(ctx)) return sr;
                ctx->cap_segs.emplace_back();
                ctx->cap_segs.back().kind = pagk_ctx::GraphSeg::FINISHER;   // (filled in below, once its arguments exist)
                fin_seg = (int)ctx->cap_segs.size() - 1;
            } else if (live) {
                HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
                HIPCHK(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
            }
            auto launch = [&](auto kern) -> hipError_t {
                hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, ctx->stream, a);
                return hipGetLastError();
            };
            hipError_t e = hipErrorInvalidValue;
            if (use_levels) {
                if (a.half == 5) e = lean ? launch(k_track_quad<2, true, true>) : launch(k_track_quad<2, false, true>);
                else if (a.half == 7) e = lean ? launch(k_track_quad<4, true, true>) : launch(k_track_quad<4, false, true>);
                else if (a.half == 10) e = lean ? launch(k_track_quad<7, true, true>) : launch(k_track_quad<7, false, true>);
            } else {
                if (a.half == 5) e = lean ? launch(k_track_quad<2, true>) : launch(k_track_quad<2>);        // P = 121: 2 chunks of 64 pixels
                else if (a.half == 7) e = lean ? launch(k_track_quad<4, true>) : launch(k_track_quad<4>);   // P = 225
                else if (a.half == 10) e = lean ? launch(k_track_quad<7, true>) : launch(k_track_quad<7>);  // P = 441
            }
            HIPCHK(ctx, e);
            TrackArgs af = a;   // what the latency kernels get
#ifdef PAGK_STAMPS
            if (af.dbg) af.dbg += (size_t)16 * (waves - nq);  // (diagnostic build: their records follow the throughput waves')
#endif
            if (live) {
                const size_t lds = track_block_lds_bytes(a.half);
                auto finisher = [&](auto kern) -> hipError_t {
                    if (segmented) {   // remembered, not launched: pagk_graph_launch issues it beside the kernel's segment
                        pagk_ctx::GraphSeg &fs = ctx->cap_segs[(size_t)fin_seg];
                        fs.fn = reinterpret_cast<const void *>(kern);
                        fs.args = af;
                        fs.grid = ctx->finisher_wgs;
                        fs.lds = lds;
                        return hipSuccess;
                    }
                    hipLaunchKernelGGL(kern, dim3(ctx->finisher_wgs), dim3(kBlock), lds, ctx->aux_stream, af);
                    return hipGetLastError();
                };
                if (a.half == 5) e = lean ? finisher(k_track_resume_live<1, 25, true>) : finisher(k_track_resume_live<1, 25>);
                else if (a.half == 7) e = lean ? finisher(k_track_resume_live<1, 1, true>) : finisher(k_track_resume_live<1, 1>);
                else e = lean ? finisher(k_track_resume_live<2, 25, true>) : finisher(k_track_resume_live<2, 25>);
                HIPCHK(ctx, e);
                if (segmented) {
                    // the kernel's own segment ends here; what follows (the sweep, whatever the caller records next) waits
                    // for the finisher at replay
                    if (int sr = capture_split(ctx)) return sr;
                    ctx->cap_segs.emplace_back();
                    ctx->cap_segs.back().kind = pagk_ctx::GraphSeg::JOIN;
                } else {
                    HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->aux_stream));
                    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
                }
            }
            if (handover) {
                // the sweep: the latency kernel finishes what is still waiting in the list (a fixed grid walks it)
                const size_t lds = track_block_lds_bytes(a.half);
                const int grid = live ? 64 : (n < 1024 ? n : 1024);
                auto resume = [&](auto kern) -> hipError_t {
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, ctx->stream, af);
                    return hipGetLastError();
                };
                if (a.half == 5) e = lean ? resume(k_track_resume<1, 25, true>) : resume(k_track_resume<1, 25>);
                else if (a.half == 7) e = lean ? resume(k_track_resume<1, 1, true>) : resume(k_track_resume<1, 1>);
                else e = lean ? resume(k_track_resume<2, 25, true>) : resume(k_track_resume<2, 25>);
                HIPCHK(ctx, e);
            }
        } else if (use_wave) {
            // one wavefront per feature (pagk_wave_kernel.h)
            const size_t lds = track_wave_lds_bytes(a.half);
            auto launch = [&](auto kern) -> hipError_t {
                hipLaunchKernelGGL(kern, dim3(n), dim3(64), lds, ctx->stream, a);
                return hipGetLastError();
            };
            hipError_t e = hipErrorInvalidValue;
            if (a.half == 5) e = lean ? launch(k_track_wave<2, 25, true>) : launch(k_track_wave<2, 25>);        // P = 121
            else if (a.half == 7) e = lean ? launch(k_track_wave<4, 1, true>) : launch(k_track_wave<4, 1>);    // P = 225
            else if (a.half == 10) e = lean ? launch(k_track_wave<7, 25, true>) : launch(k_track_wave<7, 25>);  // P = 441
            HIPCHK(ctx, e);
        } else if (ctx->kernel == 4 && mfma_ok) {
            // relaxed-order experiment (NOT parity-exact; never chosen automatically)
            const size_t lds = track_block_lds_bytes(a.half);
            auto launch = [&](auto kern) -> hipError_t {
                hipLaunchKernelGGL(kern, dim3(n), dim3(kBlock), lds, ctx->stream, a);
                return hipGetLastError();
            };
            hipError_t e = hipErrorInvalidValue;
            if (a.half == 5) e = launch(k_track_block<1, 25, 4, false, true>);
            else if (a.half == 7) e = launch(k_track_block<1, 1, 4, false, true>);
            else if (a.half == 10) e = launch(k_track_block<2, 25, 4, false, true>);
            HIPCHK(ctx, e);
#ifdef PAGK_ALL_VARIANTS
        } else if (use_mfma) {
            const size_t lds = track_mfma_lds_bytes(a.half);
            auto launch = [&](auto kern) -> hipError_t {
                hipLaunchKernelGGL(kern, dim3(n), dim3(128), lds, ctx->stream, a);
                return hipGetLastError();
            };
            hipError_t e = hipErrorInvalidValue;
            (void)Pm;
            if (a.half == 5) e = lean ? launch(k_track_block<1, 25, 2, true, false, true>) : launch(k_track_block<1, 25, 2, true>);        // P = 121
            else if (a.half == 7) e = lean ? launch(k_track_block<2, 1, 2, true, false, true>) : launch(k_track_block<2, 1, 2, true>);    // P = 225
            else if (a.half == 10) e = lean ? launch(k_track_block<4, 25, 2, true, false, true>) : launch(k_track_block<4, 25, 2, true>);  // P = 441
            HIPCHK(ctx, e);
#endif
        } else {
            const int P = (2 * a.half + 1) * (2 * a.half + 1);
            const int nr = (P + kBlock - 1) / kBlock, tail = P % 32;
            const size_t lds = track_block_lds_bytes(a.half);
            auto launch = [&](auto kern) -> hipError_t {
                // > 64 KB of dynamic LDS (h >= 14) needs the attribute; set it once per kernel and device
                static thread_local const void *configured[16] = {};
                const void *fn = reinterpret_cast<const void *>(kern);
                const int slot = ctx->device & 15;
                if (lds > 48 * 1024 && configured[slot] != fn) {
                    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    if (e != hipSuccess) return e;
                    configured[slot] = fn;
                }
                hipLaunchKernelGGL(kern, dim3(n), dim3(kBlock), lds, ctx->stream, a);
                return hipGetLastError();
            };
            // (NR, TAIL) for h = 1..15: P = (2h+1)^2 is an odd square, so P mod 32 is 1, 9, 17 or 25
            hipError_t e = hipErrorInvalidValue;
            if (pyr && pyr_blocks > 0 && mfma_ok && lds <= 48 * 1024) {
                auto launch_pyr = [&](auto kern) -> hipError_t {
                    hipLaunchKernelGGL(kern, dim3(n + pyr_blocks), dim3(kBlock), lds, ctx->stream, a, *pyr);
                    return hipGetLastError();
                };
                if (a.half == 5) e = lean ? launch_pyr(k_track_block_pyr<1, 25, true>) : launch_pyr(k_track_block_pyr<1, 25>);
                else if (a.half == 7) e = lean ? launch_pyr(k_track_block_pyr<1, 1, true>) : launch_pyr(k_track_block_pyr<1, 1>);
                else e = lean ? launch_pyr(k_track_block_pyr<2, 25, true>) : launch_pyr(k_track_block_pyr<2, 25>);
                HIPCHK(ctx, e);
                if (pyr_done) *pyr_done = true;
            } else
            switch (nr * 100 + tail) {
                case 101: e = lean ? launch(k_track_block<1, 1, 4, false, false, true>) : launch(k_track_block<1, 1>); break;    // h = 7
                case 109: e = lean ? launch(k_track_block<1, 9, 4, false, false, true>) : launch(k_track_block<1, 9>); break;    // h = 1, 6
                case 117: e = lean ? launch(k_track_block<1, 17, 4, false, false, true>) : launch(k_track_block<1, 17>); break;   // h = 3, 4
                case 125: e = lean ? launch(k_track_block<1, 25, 4, false, false, true>) : launch(k_track_block<1, 25>); break;   // h = 2, 5
                case 201: e = lean ? launch(k_track_block<2, 1, 4, false, false, true>) : launch(k_track_block<2, 1>); break;    // h = 8
                case 209: e = lean ? launch(k_track_block<2, 9, 4, false, false, true>) : launch(k_track_block<2, 9>); break;    // h = 9
                case 225:   // h = 10; several rounds of workgroups: the build for five workgroups per CU
                    // ... and a launch that five workgroups per CU hold at once but four do not (1025..1280 features on 256 CUs:
                    // one round instead of two, 104 -> 97 us at 1100 features; from 1300 on the pipelined kernel is
                    // equal or faster again, profiles/r04_block5_sweep_1100_3000.log)
                    e = lean ? ((n >= ctx->block5_min_features || (ctx->block5_window && n > 4 * ctx->cus && n <= 5 * ctx->cus))
                                    ? launch(k_track_block5<2, 25, true>)
                                    : launch(k_track_block<2, 25, 4, false, false, true>))
                             : launch(k_track_block<2, 25>);
                    break;
                case 317: e = lean ? launch(k_track_block<3, 17, 4, false, false, true>) : launch(k_track_block<3, 17>); break;   // h = 11, 12
                case 325: e = lean ? launch(k_track_block<3, 25, 4, false, false, true>) : launch(k_track_block<3, 25>); break;   // h = 13
                case 409: e = lean ? launch(k_track_block<4, 9, 4, false, false, true>) : launch(k_track_block<4, 9>); break;    // h = 14
                case 401: e = lean ? launch(k_track_block<4, 1, 4, false, false, true>) : launch(k_track_block<4, 1>); break;    // h = 15
                default: break;
            }
            if (!(pyr_done && *pyr_done)) HIPCHK(ctx, e);
        }
        HIPCHK(ctx, hipGetLastError());
    }
    if (ctx->ev_trk[1] && !in_capture(ctx)) HIPCHK(ctx, hipEventRecord(ctx->ev_trk[1], ctx->stream));
    if (!in_capture(ctx)) ctx->trk_timed = true;
    return PAGK_OK;
}

// device staging for the host-buffer path: one block holding every per-feature array
struct FeatPtrs {
    float *pt_ref, *pt_init, *affine;
    uint8_t *status_in;
    pagk_outputs out;
    size_t offs[11];   // byte offsets of the eleven arrays inside the block (and its pinned mirror)
    size_t in_bytes;   // [0, in_bytes) = the four input arrays; [in_bytes, total) = the outputs
    size_t total;
};

int feat_reserve(pagk_ctx *ctx, int n, FeatPtrs *fp)
{
    int cap = n < 1 ? 1 : n;
    // per feature: 8+8+16+1 in, 8+8+1+8+8+4+4 out; every array 256-aligned
    size_t sizes[11] = {8, 8, 16, 1, 8, 8, 1, 8, 8, 4, 4};
    size_t offs[11], total = 0;
    for (int k = 0; k < 11; k++) {
        offs[k] = total;
        total = align_up(total + sizes[k] * (size_t)cap, 256);
    }
    if (total > ctx->feat.bytes) {
        if (ctx->feat.block) HIPCHK(ctx, hipFree(ctx->feat.block));
        if (ctx->feat.host) HIPCHK(ctx, hipHostFree(ctx->feat.host));
        ctx->feat.block = nullptr;
        ctx->feat.host = nullptr;
        ctx->feat.bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->feat.block, total));
        HIPCHK(ctx, hipHostMalloc(&ctx->feat.host, total, hipHostMallocDefault));
        ctx->feat.bytes = total;
    }
    for (int k = 0; k < 11; k++) fp->offs[k] = offs[k];
    fp->in_bytes = offs[4];
    fp->total = total;
    uint8_t *b = static_cast<uint8_t *>(ctx->feat.block);
    fp->pt_ref = reinterpret_cast<float *>(b + offs[0]);
    fp->pt_init = reinterpret_cast<float *>(b + offs[1]);
    fp->affine = reinterpret_cast<float *>(b + offs[2]);
    fp->status_in = b + offs[3];
    fp->out.pt_un = reinterpret_cast<float *>(b + offs[4]);
    fp->out.pt_dist = reinterpret_cast<float *>(b + offs[5]);
    fp->out.status = b + offs[6];
    fp->out.pix_err = reinterpret_cast<double *>(b + offs[7]);
    fp->out.dist_pred = reinterpret_cast<double *>(b + offs[8]);
    fp->out.ncc = reinterpret_cast<float *>(b + offs[9]);
    fp->out.iters = reinterpret_cast<int32_t *>(b + offs[10]);
    return PAGK_OK;
}

int upload_level0(pagk_ctx *ctx, FrameSlot &s, const pagk_image *img)
{
    // straight from the caller's (pageable) memory: packing the rows into a pinned buffer first and shipping
    // one DMA measured the same (244 vs 238 us per pagk_track call, tools/host_path_time.py)
    HIPCHK(ctx, hipMemcpy2DAsync(s.u8[0], (size_t)s.w, img->data, (size_t)img->step, (size_t)s.w, (size_t)s.h,
                                 hipMemcpyHostToDevice, ctx->stream));
    return PAGK_OK;
}

int check_image(const pagk_image *im)
{
    if (!im || !im->data || im->width < 1 || im->height < 1 || im->step < im->width) return PAGK_E_ARG;
    // the samplers index with 24-bit multiplies and 32-bit element offsets
    if (im->width >= (1 << 24) || im->height >= (1 << 24) || (int64_t)im->width * im->height >= (1ll << 31)) return PAGK_E_ARG;
    return PAGK_OK;
}

int track_host_common(pagk_ctx *ctx, const pagk_params *p, int n, const float *pt_ref, const float *pt_init,
                      const float *affine, const uint8_t *status_in, const pagk_outputs *out, FrameSlot &sr,
                      FrameSlot &sc)
{
    if (n < 0 || !out || !out->pt_un || !out->status) return PAGK_E_ARG;
    if (n > 0 && (!pt_ref || !status_in)) return PAGK_E_ARG;
    if (n > 0 && p->has_gyro_predict_initial && !pt_init) return PAGK_E_ARG;
    if (n > 0 && p->consider_affine && !affine) return PAGK_E_ARG;
    FeatPtrs fp;
    int rc = feat_reserve(ctx, n, &fp);
    if (rc) return rc;
    uint8_t *hb = static_cast<uint8_t *>(ctx->feat.host), *db = static_cast<uint8_t *>(ctx->feat.block);
    if (n > 0) {
        // gather the (up to) four input arrays into the pinned mirror, ship them with ONE copy
        size_t nn = (size_t)n;
        memcpy(hb + fp.offs[0], pt_ref, nn * 8);
        if (pt_init) memcpy(hb + fp.offs[1], pt_init, nn * 8);
        if (affine) memcpy(hb + fp.offs[2], affine, nn * 16);
        memcpy(hb + fp.offs[3], status_in, nn);
        HIPCHK(ctx, hipMemcpyAsync(db, hb, fp.in_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    pagk_outputs dout = fp.out;
    if (!out->pt_dist) dout.pt_dist = nullptr;
    if (!out->pix_err) dout.pix_err = nullptr;
    if (!out->dist_pred) dout.dist_pred = nullptr;
    if (!out->ncc) dout.ncc = nullptr;
    if (!out->iters) dout.iters = nullptr;
    rc = launch_track(ctx, p, sr, sc, n, fp.pt_ref, pt_init ? fp.pt_init : nullptr, affine ? fp.affine : nullptr,
                      fp.status_in, &dout);
    if (rc) return rc;
    if (n > 0)  // every output array with ONE copy into the pinned mirror, scattered to the caller after the sync
        HIPCHK(ctx, hipMemcpyAsync(hb + fp.in_bytes, db + fp.in_bytes, fp.total - fp.in_bytes, hipMemcpyDeviceToHost,
                                   ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (int lr = lv_check(ctx)) return lr;
    if (n > 0) {
        size_t nn = (size_t)n;
        memcpy(out->pt_un, hb + fp.offs[4], nn * 8);
        if (out->pt_dist) memcpy(out->pt_dist, hb + fp.offs[5], nn * 8);
        memcpy(out->status, hb + fp.offs[6], nn);
        if (out->pix_err) memcpy(out->pix_err, hb + fp.offs[7], nn * 8);
        if (out->dist_pred) memcpy(out->dist_pred, hb + fp.offs[8], nn * 8);
        if (out->ncc) memcpy(out->ncc, hb + fp.offs[9], nn * 4);
        if (out->iters) memcpy(out->iters, hb + fp.offs[10], nn * 4);
    }
    return PAGK_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

int pagk_version(void) { return PAGK_VERSION; }

const char *pagk_strerror(int code)
{
    switch (code) {
        case PAGK_OK: return "ok";
        case PAGK_E_ARG: return "invalid argument";
        case PAGK_E_HIP: return "HIP runtime error";
        case PAGK_E_NOMEM: return "out of memory";
        case PAGK_E_UNSUPPORTED: return "unsupported mode";
        case PAGK_E_NODEVICE: return "no HIP device";
        case PAGK_E_NCCL: return "RCCL error";
        case PAGK_E_CAPACITY: return "output capacity too small";
        default: return "unknown error";
    }
}

const char *pagk_last_error(const pagk_ctx *ctx) { return ctx ? ctx->err : ""; }

// Reference call site src/gyro_aided_tracker.cpp:276-282 and eType 4 (:402-408).
void pagk_params_default(pagk_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->half_patch = 5;
    p->iterations = 10;
    p->pyramids = 3;
    p->has_gyro_predict_initial = 1;
    p->inverse = 0;
    p->consider_illumination = 1;
    p->consider_affine = 1;
    p->regularization_penalty = 0;
    p->calculate_ncc = 0;
    p->lambda = 1.0f;      // src/patch_match.cpp:48
    p->alpha = 0.5f;       // :49
    p->max_distance = 25;  // :50
    p->inv_log_max_dist = 0.0f;
    p->fx = p->fy = 1.0f;
    p->n_dist_coef = 4;
}

// src/patch_match.cpp:51  mInvLogMaxDist = 1.0 / (std::log(mAlpha * mMaxDistance + 1));
// float * int -> float, + 1 -> float, std::log(float) -> float, 1.0 / float -> double -> float member.
float pagk_inv_log_max_dist(float alpha, int32_t max_distance)
{
    float arg = alpha * (float)max_distance + 1;
    float lg = std::log(arg);
    return (float)(1.0 / (double)lg);
}

int pagk_create(pagk_ctx **out, int device)
{
    if (!out) return PAGK_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PAGK_E_NODEVICE;
    if (device < 0 || device >= ndev) return PAGK_E_ARG;
    pagk_ctx *ctx = new (std::nothrow) pagk_ctx();
    if (!ctx) return PAGK_E_NOMEM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return PAGK_E_HIP;
    }
    ctx->stream = ctx->own_stream;
    if (hipDeviceGetAttribute(&ctx->cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) ctx->cus = 0;
    if (hipMalloc(&ctx->queue, 256) != hipSuccess) {
        pagk_destroy(ctx);
        return PAGK_E_NOMEM;
    }
    {   // the error word of the one-level-per-wave launches lives in mapped host memory: read at a sync without a copy
        void *hp = nullptr;
        void *dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess) {
            memset(hp, 0, 64);
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess && dp) {
                ctx->lv_error = static_cast<int *>(hp);
                ctx->lv_error_dev = static_cast<int *>(dp);
            } else {
                (void)hipHostFree(hp);
            }
        }
    }
    // the selection thresholds were measured on 256 CUs; they are launch sizes relative to what the device holds at once,
    // so a device (or a partition of one) with another CU count gets them in proportion
    if (ctx->cus > 0 && ctx->cus != 256) {
        auto scaled = [&](int v) { return v >= 0x7fffffff / 2 ? v : (int)((long long)v * ctx->cus / 256); };
        ctx->wave_min_features = scaled(ctx->wave_min_features);
        ctx->quad_min_features = scaled(ctx->quad_min_features);
        ctx->levels_min_features = scaled(ctx->levels_min_features);
        ctx->block5_min_features = scaled(ctx->block5_min_features);
    }
    ctx->unfused_pyramid = getenv("PAGK_UNFUSED_PYRAMID") != nullptr;
    if (getenv("PAGK_MFMA_MIN")) ctx->mfma_min_features = atoi(getenv("PAGK_MFMA_MIN"));
    if (getenv("PAGK_WAVE_MIN")) ctx->wave_min_features = atoi(getenv("PAGK_WAVE_MIN"));
    if (getenv("PAGK_QUAD_MIN")) ctx->quad_min_features = atoi(getenv("PAGK_QUAD_MIN"));
    if (getenv("PAGK_BLOCK5_MIN")) ctx->block5_min_features = atoi(getenv("PAGK_BLOCK5_MIN")), ctx->block5_window = false;
    if (getenv("PAGK_LEVELS_MIN")) ctx->levels_min_features = atoi(getenv("PAGK_LEVELS_MIN"));
    if (getenv("PAGK_PRIO_K") && strcmp(getenv("PAGK_PRIO_K"), "auto") == 0) {
        // seeded with the BASELINE workloads' mean (3.5 iterations per feature and level: K = 4 until the context's own launches say otherwise)
        struct { unsigned long long st[2]; int k, pad; } seed = {{3500ull, 1000ull}, ctx->prio_k, 0};
        if (hipMalloc(reinterpret_cast<void **>(&ctx->prio_stats), sizeof seed) != hipSuccess ||
            hipMemcpy(ctx->prio_stats, &seed, sizeof seed, hipMemcpyHostToDevice) != hipSuccess) {
            if (ctx->prio_stats) (void)hipFree(ctx->prio_stats);
            ctx->prio_stats = nullptr;   // (the fixed threshold then)
            (void)hipGetLastError();
        }
    } else if (getenv("PAGK_PRIO_K")) {
        ctx->prio_k = atoi(getenv("PAGK_PRIO_K")) < 0 ? 0 : atoi(getenv("PAGK_PRIO_K"));
    }
    if (getenv("PAGK_LEVELS_XCD_SHIFT")) ctx->levels_shift = atoi(getenv("PAGK_LEVELS_XCD_SHIFT")) & 7;
    if (getenv("PAGK_LEVELS_SHARED")) ctx->levels_shared = atoi(getenv("PAGK_LEVELS_SHARED")) != 0;
    if (getenv("PAGK_QUAD_BUDGET")) ctx->quad_budget = atoi(getenv("PAGK_QUAD_BUDGET"));
    if (getenv("PAGK_ROWS_WAVES")) ctx->rows_waves_cap = atoi(getenv("PAGK_ROWS_WAVES"));
    if (getenv("PAGK_FINISHER_WGS")) ctx->finisher_wgs = atoi(getenv("PAGK_FINISHER_WGS"));
    if (getenv("PAGK_SUSPEND_LONE")) ctx->susp_lone = atoi(getenv("PAGK_SUSPEND_LONE"));
    if (getenv("PAGK_FINISHER_POLLS")) ctx->finisher_polls = atoi(getenv("PAGK_FINISHER_POLLS"));
    if (getenv("PAGK_LEVEL_POLLS")) ctx->level_polls = atoi(getenv("PAGK_LEVEL_POLLS"));
    if (hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_batch, hipEventDisableTiming) != hipSuccess) {
        pagk_destroy(ctx);
        return PAGK_E_HIP;
    }
    for (int k = 0; k < 2; k++) {
        if (hipEventCreate(&ctx->ev_trk[k]) != hipSuccess || hipEventCreate(&ctx->ev_pyr[k]) != hipSuccess) {
            pagk_destroy(ctx);
            return PAGK_E_HIP;
        }
    }
    *out = ctx;
    return PAGK_OK;
}

void pagk_destroy(pagk_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->aux_stream) {
        (void)hipStreamSynchronize(ctx->aux_stream);
        (void)hipStreamDestroy(ctx->aux_stream);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    for (int k = 0; k < pagk_ctx::kGraphs; k++) {
        if (ctx->graph_execs[k]) {
            (void)hipGraphExecDestroy(ctx->graph_execs[k]);
            (void)hipGraphDestroy(ctx->graphs[k]);
        }
        destroy_segs(ctx->pre_segs[k]);
    }
    destroy_segs(ctx->cap_segs);
    for (auto &s : ctx->slots)
        if (s.block) (void)hipFree(s.block);
    if (ctx->feat.block) (void)hipFree(ctx->feat.block);
    if (ctx->feat.host) (void)hipHostFree(ctx->feat.host);
    if (ctx->score.block) (void)hipFree(ctx->score.block);
    if (ctx->quad_ws) (void)hipFree(ctx->quad_ws);
    if (ctx->susp) (void)hipFree(ctx->susp);
    if (ctx->queue) (void)hipFree(ctx->queue);
    if (ctx->lv) (void)hipFree(ctx->lv);
    if (ctx->prio_stats) (void)hipFree(ctx->prio_stats);
    for (auto &d : ctx->batch_ring) batch_desc_free(d);
    batch_desc_free(ctx->cap_batch);
    for (int k = 0; k < pagk_ctx::kGraphs; k++) batch_desc_free(ctx->graph_batch[k]);
    if (ctx->ev_batch) (void)hipEventDestroy(ctx->ev_batch);
    if (ctx->lv_error) (void)hipHostFree(ctx->lv_error);
    for (int k = 0; k < 2; k++) {
        if (ctx->ev_trk[k]) (void)hipEventDestroy(ctx->ev_trk[k]);
        if (ctx->ev_pyr[k]) (void)hipEventDestroy(ctx->ev_pyr[k]);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int pagk_set_stream(pagk_ctx *ctx, void *hip_stream)
{
    if (!ctx) return PAGK_E_ARG;
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return PAGK_OK;
}

// Variants (b) (2-wave workgroup, f64 MFMA chain) and (e) (four independent rows per wave + work queue) win at no launch
// size any more ((b) since round 3's instruction diet of the 4-wave kernel, (e) only beyond 60000
// features) and nothing selects them automatically: they are compiled with -DPAGK_ALL_VARIANTS only (tools/ and the
// tests build that library when they want them), so the product's build and code object do not carry their fifteen
// instantiations.
int pagk_has_variant(int32_t which)
{
#ifdef PAGK_ALL_VARIANTS
    return which >= 0 && which <= 7;
#else
    return which >= 0 && which <= 7 && which != 2 && which != 6;
#endif
}

int pagk_set_kernel(pagk_ctx *ctx, int32_t which)
{
    if (!ctx || which < 0 || which > 7) return PAGK_E_ARG;
    if (!pagk_has_variant(which)) {
        snprintf(ctx->err, sizeof(ctx->err), "variant %d is not in this build of libpagk_hip.so (compile with -DPAGK_ALL_VARIANTS)", which);
        return PAGK_E_UNSUPPORTED;
    }
    ctx->kernel = which;
    return PAGK_OK;
}

int pagk_last_variant(const pagk_ctx *ctx) { return ctx ? ctx->last_variant : PAGK_E_ARG; }

int pagk_last_handover(pagk_ctx *ctx)
{
    if (!ctx) return PAGK_E_ARG;
    if (!ctx->last_handover || !ctx->susp_count_dev) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int count = 0;
    HIPCHK(ctx, hipMemcpyAsync(&count, ctx->susp_count_dev, sizeof count, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (int lr = lv_check(ctx)) return lr;
    return count;
}

// The K the next launch of the 4-wave kernels will use (csrc/pagk_prio.h): the fixed one, or what the context's statistics say.
int pagk_priority_threshold(pagk_ctx *ctx)
{
    if (!ctx) return PAGK_E_ARG;
    if (!ctx->prio_stats) return ctx->prio_k;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int k = 0;
    HIPCHK(ctx, hipMemcpyAsync(&k, ctx->prio_stats + 2, sizeof k, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (int lr = lv_check(ctx)) return lr;
    return k;
}

// The error word of the level-by-level launches without a synchronisation: for callers that synchronise the stream
// themselves (pagk_set_stream, a torch stream) and therefore never pass through pagk_sync.  Call it AFTER that
// synchronisation.  PAGK_OK, or PAGK_E_HIP once for a launch in which a wave gave up waiting.
int pagk_check_launch(pagk_ctx *ctx)
{
    if (!ctx) return PAGK_E_ARG;
    return lv_check(ctx);
}

int pagk_set_concurrency(pagk_ctx *ctx, int32_t streams)
{
    if (!ctx || streams < 1 || streams > 64) return PAGK_E_ARG;
    ctx->concurrency = streams;
    return PAGK_OK;
}

int pagk_sync(pagk_ctx *ctx)
{
    if (!ctx) return PAGK_E_ARG;
    NOT_WHILE_CAPTURING(ctx, "pagk_sync");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return lv_check(ctx);
}

int pagk_last_kernel_ms(pagk_ctx *ctx, float *track_ms, float *pyramid_ms)
{
    if (!ctx) return PAGK_E_ARG;
    NOT_WHILE_CAPTURING(ctx, "pagk_last_kernel_ms");
    if (track_ms) {
        *track_ms = 0.0f;
        if (ctx->trk_timed) {
            HIPCHK(ctx, hipEventSynchronize(ctx->ev_trk[1]));
            HIPCHK(ctx, hipEventElapsedTime(track_ms, ctx->ev_trk[0], ctx->ev_trk[1]));
        }
    }
    if (pyramid_ms) {
        *pyramid_ms = 0.0f;
        if (ctx->pyr_timed) {
            HIPCHK(ctx, hipEventSynchronize(ctx->ev_pyr[1]));
            HIPCHK(ctx, hipEventElapsedTime(pyramid_ms, ctx->ev_pyr[0], ctx->ev_pyr[1]));
        }
    }
    return PAGK_OK;
}

static int frame_upload_any(pagk_ctx *ctx, int32_t slot, const pagk_image *img, int32_t pyramids);

int pagk_frame_upload(pagk_ctx *ctx, int32_t slot, const pagk_image *img, int32_t pyramids)
{
    if (!ctx || slot < 0 || slot >= kUserSlots) return PAGK_E_ARG;
    NOT_WHILE_CAPTURING(ctx, "pagk_frame_upload");
    int rc = frame_upload_any(ctx, slot, img, pyramids);
    if (rc) return rc;
    // the copy reads img->data asynchronously when that memory is pinned (a camera ring buffer, a pinned tensor):
    // do not return before it has been read, or the caller could overwrite the frame while it is in flight
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PAGK_OK;
}

// The same for a frame in PINNED host memory (a camera driver's ring buffer, hipHostMalloc / hipHostRegister):
// asynchronous -- returns once the copy and the pyramid are enqueued on the context's stream -- and capturable: between
// pagk_graph_begin and pagk_graph_end the host -> device copy becomes a node of the graph, so a live loop replays
// [copy the frame the camera just wrote -> pyramid -> PatchMatch] with one pagk_graph_launch per frame.  The caller
// keeps the memory pinned, and unchanged from the call (or the replay) until that work has run.
int pagk_frame_upload_pinned(pagk_ctx *ctx, int32_t slot, const pagk_image *img, int32_t pyramids)
{
    if (!ctx || slot < 0 || slot >= kUserSlots) return PAGK_E_ARG;
    return frame_upload_any(ctx, slot, img, pyramids);
}

static int frame_upload_any(pagk_ctx *ctx, int32_t slot, const pagk_image *img, int32_t pyramids)
{
    if (!ctx || slot < 0 || slot >= kSlots || pyramids < 1 || pyramids > PAGK_MAX_PYRAMIDS) return PAGK_E_ARG;
    int rc = check_image(img);
    if (rc) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    FrameSlot &s = ctx->slots[slot];
    s.valid = false;
    if ((rc = slot_reserve(ctx, s, img->width, img->height, pyramids))) return rc;
    if ((rc = upload_level0(ctx, s, img))) return rc;
    rc = slot_build(ctx, s, s.u8[0], s.w, img->step == img->width);
    s.pad0 = (int)(img->step - img->width > 2 ? 2 : img->step - img->width);
    return rc;
}

int pagk_frame_set_device(pagk_ctx *ctx, int32_t slot, const void *d_data, int32_t width, int32_t height,
                          int64_t step, int32_t pyramids)
{
    if (!ctx || slot < 0 || slot >= kUserSlots || pyramids < 1 || pyramids > PAGK_MAX_PYRAMIDS) return PAGK_E_ARG;
    if (!d_data || width < 1 || height < 1 || step < width) return PAGK_E_ARG;
    if (width >= (1 << 24) || height >= (1 << 24) || (int64_t)width * height >= (1ll << 31)) return PAGK_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    FrameSlot &s = ctx->slots[slot];
    s.valid = false;
    int rc = slot_reserve(ctx, s, width, height, pyramids);
    if (rc) return rc;
    // level 0 is read in place: no copy
    rc = slot_build(ctx, s, static_cast<const uint8_t *>(d_data), step, step == width);
    s.pad0 = (int)(step - width > 2 ? 2 : step - width);
    return rc;
}

// pagk_frame_set_device for the frames of k contexts that share a device, as ONE launch: the CreatePyramids
// (src/patch_match.cpp:61-76) of k trackers stepped together.  A pyramid kernel is a few microseconds of work behind a
// launch: eight of them in a row cost 50-125 us of a 0.85 ms batched step (tools/batch_breakdown.py).  Per frame the same
// bytes as its own launch (the same block body on the same arguments).
int pagk_frame_set_device_batch(pagk_ctx *const *ctxs, int32_t k, const int32_t *slot, const void *const *d_data,
                                const int32_t *width, const int32_t *height, const int64_t *step, int32_t pyramids)
{
    if (!ctxs || k < 1 || k > pagk_ctx::kBatchMaxStreams || !slot || !d_data || !width || !height || !step) return PAGK_E_ARG;
    if (pyramids < 1 || pyramids > PAGK_MAX_PYRAMIDS) return PAGK_E_ARG;
    for (int j = 0; j < k; j++) {
        if (!ctxs[j] || slot[j] < 0 || slot[j] >= kUserSlots || !d_data[j] || width[j] < 1 || height[j] < 1 || step[j] < width[j]) return PAGK_E_ARG;
        if (width[j] >= (1 << 24) || height[j] >= (1 << 24) || (int64_t)width[j] * height[j] >= (1ll << 31)) return PAGK_E_ARG;
        for (int i = 0; i < j; i++)
            if (ctxs[i] == ctxs[j] && slot[i] == slot[j]) return PAGK_E_ARG;   // (one frame per slot)
    }
    pagk_ctx *lead = ctxs[0];
    for (int j = 1; j < k; j++)
        if (ctxs[j]->device != lead->device) {
            snprintf(lead->err, sizeof(lead->err), "pagk_frame_set_device_batch: context %d lives on device %d, the first on %d", j, ctxs[j]->device, lead->device);
            return PAGK_E_ARG;
        }
    HIPCHK(lead, hipSetDevice(lead->device));
    // the slots first (allocation, level pointers); then: can every frame go through the single-launch kernel?
    bool fused = k > 1;
    for (int j = 0; j < k; j++) {
        pagk_ctx *c = ctxs[j];
        FrameSlot &s = c->slots[slot[j]];
        s.valid = false;
        int rc = slot_reserve(c, s, width[j], height[j], pyramids);
        if (rc) {
            if (c != lead) snprintf(lead->err, sizeof(lead->err), "stream %d: %s", j, c->err);
            return rc;
        }
        fused = fused && pyramid_fusable(s) && !c->unfused_pyramid;
    }
    if (!fused) {   // odd parents, more than four levels, a single frame: every frame as its own launch(es) on its own context
        for (int j = 0; j < k; j++) {
            pagk_ctx *c = ctxs[j];
            FrameSlot &s = c->slots[slot[j]];
            int rc = slot_build(c, s, static_cast<const uint8_t *>(d_data[j]), step[j], step[j] == width[j]);
            s.pad0 = (int)(step[j] - width[j] > 2 ? 2 : step[j] - width[j]);
            if (rc) {
                if (c != lead) snprintf(lead->err, sizeof(lead->err), "stream %d: %s", j, c->err);
                return rc;
            }
        }
        return PAGK_OK;
    }
    int rc = PAGK_OK;
    pagk_ctx::BatchDesc *desc = batch_desc_take(lead, &rc);
    if (!desc) return rc;
    PyrBatchEntry *he = static_cast<PyrBatchEntry *>(desc->host);
    int nb = 0;
    for (int j = 0; j < k; j++) {
        FrameSlot &s = ctxs[j]->slots[slot[j]];
        memset(&he[j], 0, sizeof he[j]);
        he[j].block_base = nb;
        nb += make_pyr_args(s, static_cast<const uint8_t *>(d_data[j]), step[j], step[j] == width[j], &he[j].a);
    }
    // the launch overwrites slots the other contexts' streams may still be reading, and reads images they may be writing ...
    for (int j = 1; j < k; j++)
        if (ctxs[j]->stream != lead->stream) {
            HIPCHK(lead, hipEventRecord(ctxs[j]->ev_batch, ctxs[j]->stream));
            HIPCHK(lead, hipStreamWaitEvent(lead->stream, ctxs[j]->ev_batch, 0));
        }
    HIPCHK(lead, hipMemcpyAsync(desc->dev, he, (size_t)k * sizeof(PyrBatchEntry), hipMemcpyHostToDevice, lead->stream));
    if (lead->ev_pyr[0] && !in_capture(lead)) HIPCHK(lead, hipEventRecord(lead->ev_pyr[0], lead->stream));
    hipLaunchKernelGGL(k_pyramid_fused_batch, dim3(nb), dim3(256), 0, lead->stream, static_cast<const PyrBatchEntry *>(desc->dev), (int)k);
    HIPCHK(lead, hipGetLastError());
    if (lead->ev_pyr[1] && !in_capture(lead)) HIPCHK(lead, hipEventRecord(lead->ev_pyr[1], lead->stream));
    if (!in_capture(lead)) lead->pyr_timed = true;
    if ((rc = batch_desc_used(lead, desc)) != PAGK_OK) return rc;
    // ... and what those streams do next sees the pyramids
    bool others = false;
    for (int j = 1; j < k; j++) others = others || ctxs[j]->stream != lead->stream;
    if (others) {
        HIPCHK(lead, hipEventRecord(lead->ev_batch, lead->stream));
        for (int j = 1; j < k; j++)
            if (ctxs[j]->stream != lead->stream) HIPCHK(lead, hipStreamWaitEvent(ctxs[j]->stream, lead->ev_batch, 0));
    }
    for (int j = 0; j < k; j++) {
        FrameSlot &s = ctxs[j]->slots[slot[j]];
        s.wrap0 = step[j] == width[j];
        s.pad0 = (int)(step[j] - width[j] > 2 ? 2 : step[j] - width[j]);
        s.valid = true;
    }
    return PAGK_OK;
}

int pagk_frame_download_level(pagk_ctx *ctx, int32_t slot, int32_t level, uint8_t *dst, int32_t *width,
                              int32_t *height)
{
    if (!ctx || slot < 0 || slot >= kUserSlots || !dst) return PAGK_E_ARG;
    NOT_WHILE_CAPTURING(ctx, "pagk_frame_download_level");
    FrameSlot &s = ctx->slots[slot];
    if (!s.valid || level < 1 || level >= s.L) return PAGK_E_ARG;  // level 0 is the caller's own image
    int lw[kMaxLevels], lh[kMaxLevels];
    level_dims(s.w, s.h, s.L, lw, lh);
    HIPCHK(ctx, hipMemcpyAsync(dst, s.u8[level], (size_t)lw[level] * lh[level], hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (width) *width = lw[level];
    if (height) *height = lh[level];
    return PAGK_OK;
}

int pagk_track_device(pagk_ctx *ctx, const pagk_params *params, int32_t slot_ref, int32_t slot_cur, int32_t n,
                      const float *d_pt_ref_un, const float *d_pt_init_un, const float *d_affine,
                      const uint8_t *d_status_in, const pagk_outputs *d_out)
{
    if (!ctx) return PAGK_E_ARG;
    int rc = check_params(params);
    if (rc) return rc;
    if (slot_ref < 0 || slot_ref >= kUserSlots || slot_cur < 0 || slot_cur >= kUserSlots) return PAGK_E_ARG;
    FrameSlot &sr = ctx->slots[slot_ref], &sc = ctx->slots[slot_cur];
    if (!sr.valid || !sc.valid || sr.L < params->pyramids || sc.L < params->pyramids) return PAGK_E_ARG;
    if (sr.w != sc.w || sr.h != sc.h) return PAGK_E_ARG;
    if (n < 0 || !d_out || !d_out->pt_un || !d_out->status) return PAGK_E_ARG;
    if (n > 0 && (!d_pt_ref_un || !d_status_in)) return PAGK_E_ARG;
    if (n > 0 && params->has_gyro_predict_initial && !d_pt_init_un) return PAGK_E_ARG;
    if (n > 0 && params->consider_affine && !d_affine) return PAGK_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return launch_track(ctx, params, sr, sc, n, d_pt_ref_un, d_pt_init_un, d_affine, d_status_in, d_out);
}

// One launch for k camera streams that share this device (BASELINE configs[4], "batched multi-camera").  The reference
// builds one PatchMatch per tracker (src/gyro_aided_tracker.cpp:276-283); k trackers' calls are k independent feature
// sets over k image pairs.  As k launches they are k launches of a few thousand features -- the latency variants' size,
// or the throughput variant run as k concurrent grids that hold each other's slots; as ONE launch of variant 7 (four
// features per wave, one pyramid level per wave; a quad carries its stream) they are a launch of several rounds of
// resident waves, which is where that variant is at its rate.  Per stream: the bits of its own pagk_track_device.
int pagk_track_device_batch(pagk_ctx *const *ctxs, int32_t k, const pagk_params *params, const int32_t *slot_ref,
                            const int32_t *slot_cur, const int32_t *n, const float *const *d_pt_ref_un,
                            const float *const *d_pt_init_un, const float *const *d_affine,
                            const uint8_t *const *d_status_in, const pagk_outputs *d_out)
{
    if (!ctxs || k < 1 || k > 64 || !slot_ref || !slot_cur || !n || !d_pt_ref_un || !d_status_in || !d_out) return PAGK_E_ARG;
    for (int j = 0; j < k; j++)
        if (!ctxs[j]) return PAGK_E_ARG;
    pagk_ctx *lead = ctxs[0];
    int rc = check_params(params);
    if (rc) return rc;
    long long total_n = 0;
    int total_q = 0;
    for (int j = 0; j < k; j++) {
        pagk_ctx *c = ctxs[j];
        if (c->device != lead->device) {
            snprintf(lead->err, sizeof(lead->err), "pagk_track_device_batch: context %d lives on device %d, the first on %d", j, c->device, lead->device);
            return PAGK_E_ARG;
        }
        if (slot_ref[j] < 0 || slot_ref[j] >= kUserSlots || slot_cur[j] < 0 || slot_cur[j] >= kUserSlots) return PAGK_E_ARG;
        const FrameSlot &sr = c->slots[slot_ref[j]], &sc = c->slots[slot_cur[j]];
        if (!sr.valid || !sc.valid || sr.L < params->pyramids || sc.L < params->pyramids || sr.w != sc.w || sr.h != sc.h) return PAGK_E_ARG;
        if (n[j] < 0 || !d_out[j].pt_un || !d_out[j].status) return PAGK_E_ARG;
        if (n[j] > 0 && (!d_pt_ref_un[j] || !d_status_in[j])) return PAGK_E_ARG;
        if (n[j] > 0 && params->has_gyro_predict_initial && (!d_pt_init_un || !d_pt_init_un[j])) return PAGK_E_ARG;
        if (n[j] > 0 && params->consider_affine && (!d_affine || !d_affine[j])) return PAGK_E_ARG;
        total_n += n[j];
        total_q += (n[j] + 3) / 4;
    }
    HIPCHK(lead, hipSetDevice(lead->device));
    const bool mfma_ok = params->half_patch == 5 || params->half_patch == 7 || params->half_patch == 10;
    const bool batched = mfma_ok && !params->calculate_ncc && params->pyramids >= 2 && lead->lv_error && total_q > 0 &&
                         (lead->kernel == 7 || (lead->kernel == 0 && total_n >= lead->levels_min_features));
    if (!batched) {
        // every stream as its own launch on its own context (small batches, NCC launches, a forced variant)
        for (int j = 0; j < k; j++) {
            pagk_ctx *c = ctxs[j];
            rc = launch_track(c, params, c->slots[slot_ref[j]], c->slots[slot_cur[j]], n[j], d_pt_ref_un[j],
                              d_pt_init_un ? d_pt_init_un[j] : nullptr, d_affine ? d_affine[j] : nullptr, d_status_in[j], &d_out[j]);
            if (rc) {
                if (c != lead) snprintf(lead->err, sizeof(lead->err), "stream %d: %s", j, c->err);
                return rc;
            }
        }
        return PAGK_OK;
    }
    TrackArgs a;
    memset(&a, 0, sizeof a);
    a.n_levels = params->pyramids;
    for (int l = 0; l < params->pyramids; l++) a.scales[l] = l == 0 ? 1.0f : (float)((double)a.scales[l - 1] * 0.5);  // :66,:73
    fill_param_args(a, params);
    a.n = 4 * total_q;   // features numbered through the batch, each stream padded to whole quads
    a.batch_k = k;
    // the stream descriptors: pinned source -> device.  The copy is asynchronous and, inside a capture, a node that is
    // replayed later: the pair it uses is this launch's alone until the launch is over (ring) / the graph's (capture)
    pagk_ctx::BatchDesc *desc = batch_desc_take(lead, &rc);
    if (!desc) return rc;
    BatchStream *hb = static_cast<BatchStream *>(desc->host);
    int qb = 0;
    for (int j = 0; j < k; j++) {
        pagk_ctx *c = ctxs[j];
        BatchStream &B = hb[j];
        memset(&B, 0, sizeof B);
        for (int l = 0; l < params->pyramids; l++) {
            fill_level(B.l1[l], c->slots[slot_ref[j]], l);
            fill_level(B.l2[l], c->slots[slot_cur[j]], l);
        }
        B.pt_ref = d_pt_ref_un[j];
        B.pt_init = d_pt_init_un ? d_pt_init_un[j] : nullptr;
        B.affine = d_affine ? d_affine[j] : nullptr;
        B.status_in = d_status_in[j];
        B.pt_un = d_out[j].pt_un, B.pt_dist = d_out[j].pt_dist, B.status = d_out[j].status;
        B.pix_err = d_out[j].pix_err, B.dist_pred = d_out[j].dist_pred, B.ncc = d_out[j].ncc, B.iters = d_out[j].iters;
        B.n = n[j];
        B.quad_base = qb;
        qb += (n[j] + 3) / 4;
    }
    // the launch reads what the other contexts' streams produced (their pyramids, their prediction kernels' outputs) ...
    for (int j = 1; j < k; j++)
        if (ctxs[j]->stream != lead->stream) {
            HIPCHK(lead, hipEventRecord(ctxs[j]->ev_batch, ctxs[j]->stream));
            HIPCHK(lead, hipStreamWaitEvent(lead->stream, ctxs[j]->ev_batch, 0));
        }
    HIPCHK(lead, hipMemcpyAsync(desc->dev, hb, (size_t)k * sizeof(BatchStream), hipMemcpyHostToDevice, lead->stream));
    a.batch = static_cast<const BatchStream *>(desc->dev);
    // workspaces of a one-level-per-wave launch (launch_track), for the batch's quads; no hand-over
    const int Pm = (2 * a.half + 1) * (2 * a.half + 1), nch = (Pm + 63) / 64, nq = total_q, waves = nq * params->pyramids;
    const size_t need = (size_t)waves * 4 * nch * 64 * sizeof(float);
    const size_t ready_bytes = align_up((size_t)(params->pyramids - 1) * 8 * ((nq + 7) / 8) * 4, 256);
    const size_t susp_zero = 256 + align_up((size_t)a.n * 4, 256);
    const size_t need_lv = 32768 + ready_bytes + susp_zero + (size_t)a.n * 16 + (size_t)a.n * sizeof(SuspState);
    if (need > lead->quad_ws_bytes || need_lv > lead->lv_bytes) {
        if (in_capture(lead)) {
            snprintf(lead->err, sizeof(lead->err), "the quad kernel's workspace would have to be (re)allocated during graph capture");
            return PAGK_E_ARG;
        }
        if (int gr = no_live_graphs(lead, "the quad kernel's workspace")) return gr;
        if (need > lead->quad_ws_bytes) {
            if (lead->quad_ws) HIPCHK(lead, hipFree(lead->quad_ws));
            lead->quad_ws = nullptr, lead->quad_ws_bytes = 0;
            HIPCHK(lead, hipMalloc(&lead->quad_ws, need));
            lead->quad_ws_bytes = need;
        }
        if (need_lv > lead->lv_bytes) {
            if (lead->lv) HIPCHK(lead, hipFree(lead->lv));
            lead->lv = nullptr, lead->lv_bytes = 0;
            HIPCHK(lead, hipMalloc(&lead->lv, need_lv));
            lead->lv_bytes = need_lv;
        }
    }
    uint8_t *lb = static_cast<uint8_t *>(lead->lv);
    a.ws = static_cast<float *>(lead->quad_ws);
    a.queue = reinterpret_cast<int *>(lb);
    a.lv_ready = reinterpret_cast<int *>(lb + 32768);
    a.lv_state = reinterpret_cast<float *>(lb + 32768 + ready_bytes + susp_zero);
    a.lv_error = lead->lv_error_dev;
    a.lv_polls = lead->level_polls;
    a.lv_shift = lead->levels_shift;
    HIPCHK(lead, hipMemsetAsync(lb, 0, 32768 + ready_bytes, lead->stream));
    if (lead->ev_trk[0] && !in_capture(lead)) HIPCHK(lead, hipEventRecord(lead->ev_trk[0], lead->stream));
    const bool lean = !a.penalty && a.solver == 0;
    auto launch = [&](auto kern) -> hipError_t {
        hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, lead->stream, a);
        return hipGetLastError();
    };
    hipError_t e = hipErrorInvalidValue;
    if (a.half == 5) e = lean ? launch(k_track_quad<2, true, true, true>) : launch(k_track_quad<2, false, true, true>);
    else if (a.half == 7) e = lean ? launch(k_track_quad<4, true, true, true>) : launch(k_track_quad<4, false, true, true>);
    else e = lean ? launch(k_track_quad<7, true, true, true>) : launch(k_track_quad<7, false, true, true>);
    HIPCHK(lead, e);
    if ((rc = batch_desc_used(lead, desc)) != PAGK_OK) return rc;
    if (lead->ev_trk[1] && !in_capture(lead)) HIPCHK(lead, hipEventRecord(lead->ev_trk[1], lead->stream));
    if (!in_capture(lead)) lead->trk_timed = true;
    // ... and whatever those streams do next sees its results
    bool others = false;
    for (int j = 1; j < k; j++) others = others || ctxs[j]->stream != lead->stream;
    if (others) {
        HIPCHK(lead, hipEventRecord(lead->ev_batch, lead->stream));
        for (int j = 1; j < k; j++)
            if (ctxs[j]->stream != lead->stream) HIPCHK(lead, hipStreamWaitEvent(ctxs[j]->stream, lead->ev_batch, 0));
    }
    for (int j = 0; j < k; j++) {
        ctxs[j]->last_variant = 7;
        ctxs[j]->last_handover = false;
    }
    return PAGK_OK;
}

// Tracking of (slot_ref, slot_cur) with the pyramid of ANOTHER frame (device image d_next) built into slot_next
// by the same launch when the 4-wave kernel is selected; otherwise two launches.  Same results as
// pagk_frame_set_device(slot_next, ...) followed by pagk_track_device(...).
int pagk_track_device_fused(pagk_ctx *ctx, const pagk_params *params, int32_t slot_ref, int32_t slot_cur, int32_t n,
                            const float *d_pt_ref_un, const float *d_pt_init_un, const float *d_affine,
                            const uint8_t *d_status_in, const pagk_outputs *d_out, int32_t slot_next,
                            const void *d_next, int32_t width, int32_t height, int64_t step, int32_t pyramids)
{
    if (!ctx || !d_next || slot_next < 0 || slot_next >= kUserSlots || slot_next == slot_ref || slot_next == slot_cur ||
        width < 1 || height < 1 || step < width || pyramids < 1 || pyramids > PAGK_MAX_PYRAMIDS)
        return PAGK_E_ARG;
    if (slot_ref < 0 || slot_ref >= kUserSlots || slot_cur < 0 || slot_cur >= kUserSlots) return PAGK_E_ARG;
    int rc = check_params(params);
    if (rc) return rc;
    FrameSlot &sr = ctx->slots[slot_ref], &sc = ctx->slots[slot_cur], &sn = ctx->slots[slot_next];
    if (!sr.valid || !sc.valid || sr.w != sc.w || sr.h != sc.h || sr.L < params->pyramids || sc.L < params->pyramids)
        return PAGK_E_ARG;
    if (n < 0 || !d_out || !d_out->pt_un || !d_out->status) return PAGK_E_ARG;
    if (n > 0 && (!d_pt_ref_un || !d_status_in)) return PAGK_E_ARG;
    if (n > 0 && params->has_gyro_predict_initial && !d_pt_init_un) return PAGK_E_ARG;
    if (n > 0 && params->consider_affine && !d_affine) return PAGK_E_ARG;
    pagk_image probe{static_cast<const uint8_t *>(d_next), width, height, step};
    if ((rc = check_image(&probe))) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    sn.valid = false;
    if ((rc = slot_reserve(ctx, sn, width, height, pyramids))) return rc;
    const uint8_t *src0 = static_cast<const uint8_t *>(d_next);
    const int wrap0 = step == width;
    bool fused = false;
    if (n > 0 && pyramid_fusable(sn) && !ctx->unfused_pyramid) {
        PyrArgs pa;
        const int nb = make_pyr_args(sn, src0, step, wrap0, &pa);
        rc = launch_track(ctx, params, sr, sc, n, d_pt_ref_un, d_pt_init_un, d_affine, d_status_in, d_out, &pa, nb, &fused);
        if (rc) return rc;
        if (fused) {
            sn.wrap0 = wrap0;
            sn.pad0 = (int)(step - width > 2 ? 2 : step - width);
            sn.valid = true;
            return PAGK_OK;
        }
    } else {
        rc = launch_track(ctx, params, sr, sc, n, d_pt_ref_un, d_pt_init_un, d_affine, d_status_in, d_out);
        if (rc) return rc;
    }
    rc = slot_build(ctx, sn, src0, step, wrap0);  // the launch selected another variant: pyramid on its own
    sn.pad0 = (int)(step - width > 2 ? 2 : step - width);
    return rc;
}

int pagk_track(pagk_ctx *ctx, const pagk_params *params, const pagk_image *ref, const pagk_image *cur, int32_t n,
               const float *pt_ref_un, const float *pt_init_un, const float *affine, const uint8_t *status_in,
               const pagk_outputs *out)
{
    if (!ctx) return PAGK_E_ARG;
    NOT_WHILE_CAPTURING(ctx, "pagk_track");
    int rc = check_params(params);
    if (rc) return rc;
    if ((rc = check_image(ref)) || (rc = check_image(cur))) return rc;
    if (ref->width != cur->width || ref->height != cur->height) return PAGK_E_ARG;
    if ((rc = frame_upload_any(ctx, 4, ref, params->pyramids))) return rc;
    if ((rc = frame_upload_any(ctx, 5, cur, params->pyramids))) return rc;
    return track_host_common(ctx, params, n, pt_ref_un, pt_init_un, affine, status_in, out, ctx->slots[4],
                             ctx->slots[5]);
}

int pagk_track_pyr(pagk_ctx *ctx, const pagk_params *params, int32_t n_levels, const pagk_image *ref_levels,
                   const pagk_image *cur_levels, int32_t n, const float *pt_ref_un, const float *pt_init_un,
                   const float *affine, const uint8_t *status_in, const pagk_outputs *out)
{
    if (!ctx || !ref_levels || !cur_levels) return PAGK_E_ARG;
    NOT_WHILE_CAPTURING(ctx, "pagk_track_pyr");
    int rc = check_params(params);
    if (rc) return rc;
    if (n_levels != params->pyramids) return PAGK_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // caller-built pyramids: every level is uploaded and packed as is; level sizes must be
    // the ones CreatePyramids would produce (int(cols*0.5)), but parents may be odd.
    FrameSlot *sl[2] = {&ctx->slots[4], &ctx->slots[5]};
    const pagk_image *lv[2] = {ref_levels, cur_levels};
    for (int k = 0; k < 2; k++) {
        for (int l = 0; l < n_levels; l++)
            if ((rc = check_image(&lv[k][l]))) return rc;
        int w = lv[k][0].width, h = lv[k][0].height;
        for (int l = 1; l < n_levels; l++) {
            w = (int)(w * 0.5);
            h = (int)(h * 0.5);
            if (lv[k][l].width != w || lv[k][l].height != h) return PAGK_E_ARG;
        }
        if (lv[k][0].width != lv[0][0].width || lv[k][0].height != lv[0][0].height) return PAGK_E_ARG;
        FrameSlot &s = *sl[k];
        s.valid = false;
        // reserve without the even-parent restriction
        size_t total = 0, off_u8[kMaxLevels], off_q[kMaxLevels];
        for (int l = 0; l < n_levels; l++) {
            off_u8[l] = total;
            total = align_up(total + (size_t)lv[k][l].width * lv[k][l].height, 256);
        }
        for (int l = 0; l < n_levels; l++) {
            off_q[l] = total;
            total = align_up(total + (size_t)lv[k][l].width * lv[k][l].height * 4, 256);
        }
        if (total > s.block_bytes) {
            if (s.block) HIPCHK(ctx, hipFree(s.block));
            s.block = nullptr;
            s.block_bytes = 0;
            HIPCHK(ctx, hipMalloc(&s.block, total));
            s.block_bytes = total;
        }
        s.w = lv[k][0].width;
        s.h = lv[k][0].height;
        s.L = n_levels;
        dim3 blk(32, 8);
        for (int l = 0; l < n_levels; l++) {
            const pagk_image &im = lv[k][l];
            s.u8[l] = static_cast<uint8_t *>(s.block) + off_u8[l];
            s.quad[l] = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(s.block) + off_q[l]);
            HIPCHK(ctx, hipMemcpy2DAsync(s.u8[l], (size_t)im.width, im.data, (size_t)im.step, (size_t)im.width,
                                         (size_t)im.height, hipMemcpyHostToDevice, ctx->stream));
            dim3 grd((im.width + 31) / 32, (im.height + 7) / 8);
            hipLaunchKernelGGL(k_build_quads, grd, blk, 0, ctx->stream, (const uint8_t *)s.u8[l], (int64_t)im.width,
                               im.width, im.height, im.step == im.width ? 1 : 0, s.quad[l]);
        }
        HIPCHK(ctx, hipGetLastError());
        s.wrap0 = lv[k][0].step == lv[k][0].width;
        s.pad0 = (int)(lv[k][0].step - lv[k][0].width > 2 ? 2 : lv[k][0].step - lv[k][0].width);
        s.valid = true;
    }
    return track_host_common(ctx, params, n, pt_ref_un, pt_init_un, affine, status_in, out, ctx->slots[4],
                             ctx->slots[5]);
}

static int gyro_predict_any(pagk_ctx *ctx, const pagk_params *params, int32_t width, int32_t height,
                            const float *KRKinv, const float *r3, const float *d_rot, int32_t n,
                            const float *d_pt_ref_un, float *d_pt_predict_un, float *d_pt_predict, uint8_t *d_status,
                            float *d_affine)
{
    if (!ctx || !params || (!d_rot && (!KRKinv || !r3)) || n < 0 || width < 1 || height < 1) return PAGK_E_ARG;
    if (params->half_patch < 1 || params->half_patch > PAGK_MAX_HALF_PATCH) return PAGK_E_ARG;
    if (n > 0 && (!d_pt_ref_un || !d_pt_predict_un || !d_pt_predict || !d_status)) return PAGK_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PredictArgs a;
    memset(&a, 0, sizeof a);
    a.n = n, a.width = width, a.height = height;
    a.half = (float)params->half_patch;
    a.single_homography = params->predict_method == 2;  // ePredictMethod SINGLE_HOMOGRAPHY
    a.fx = params->fx, a.fy = params->fy, a.cx = params->cx, a.cy = params->cy;
    a.fx_inv = (float)(1.0 / (double)params->fx);  // src/gyro_aided_tracker.cpp:66
    a.fy_inv = (float)(1.0 / (double)params->fy);
    a.k1 = params->dist_coef[0], a.k2 = params->dist_coef[1], a.p1 = params->dist_coef[2], a.p2 = params->dist_coef[3];
    a.k3 = params->n_dist_coef == 5 ? params->dist_coef[4] : 0.0f;
    if (d_rot) {
        a.d_rot = d_rot;
    } else {
        for (int k = 0; k < 6; k++) a.K[k] = KRKinv[k];
        a.r31 = r3[0], a.r32 = r3[1], a.r33 = r3[2];
    }
    // (B B^T)^-1 for B = +-h corners (:73-78): B B^T = diag(4h^2); cv::Mat::inv of a 2x2 goes through
    // the double determinant
    const float m00 = 4.0f * a.half * a.half;
    const double det = (double)m00 * m00, dinv = det != 0 ? 1. / det : 0;
    a.inv00 = (float)(m00 * dinv);
    a.inv01 = (float)(-0.0f * dinv);
    a.pt_ref = d_pt_ref_un;
    a.pt_un = d_pt_predict_un;
    a.pt_dist = d_pt_predict;
    a.status = d_status;
    a.affine = d_affine;
    if (n > 0) {
        hipLaunchKernelGGL(k_gyro_predict, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, a);
        HIPCHK(ctx, hipGetLastError());
    }
    return PAGK_OK;
}

int pagk_gyro_predict_device(pagk_ctx *ctx, const pagk_params *params, int32_t width, int32_t height,
                             const float *KRKinv, const float *r3, int32_t n, const float *d_pt_ref_un,
                             float *d_pt_predict_un, float *d_pt_predict, uint8_t *d_status, float *d_affine)
{
    return gyro_predict_any(ctx, params, width, height, KRKinv, r3, nullptr, n, d_pt_ref_un, d_pt_predict_un,
                            d_pt_predict, d_status, d_affine);
}

int pagk_gyro_predict_device_rot(pagk_ctx *ctx, const pagk_params *params, int32_t width, int32_t height,
                                 const float *d_rot, int32_t n, const float *d_pt_ref_un, float *d_pt_predict_un,
                                 float *d_pt_predict, uint8_t *d_status, float *d_affine)
{
    if (!d_rot) return PAGK_E_ARG;
    return gyro_predict_any(ctx, params, width, height, nullptr, nullptr, d_rot, n, d_pt_ref_un, d_pt_predict_un,
                            d_pt_predict, d_status, d_affine);
}

// GyroAidedTracker::GyroPredictFeaturesAndOpticalFlowRefined, Step 3,
// src/gyro_aided_tracker.cpp:289-341.  O(n) host arithmetic on the gathered results.
int pagk_post_filter(int32_t n, int32_t half_patch, const uint8_t *status_pm, const double *pix_err,
                     const double *dist_pred, const float *pt_pm, const float *pt_pm_un, uint8_t *status_out,
                     float *pt_predict, float *pt_predict_un)
{
    if (n < 0 || (n > 0 && (!status_pm || !pix_err || !dist_pred || !status_out))) return PAGK_E_ARG;
    double sum = 0;
    int cnt = 0;
    for (int i = 0; i < n; ++i)
        if (status_pm[i]) {  // :298
            sum += pix_err[i];
            cnt++;
        }
    const double avg = sum / cnt;  // :305 (cnt == 0 -> NaN -> threshold falls back to h)
    const double th_pix = 4.0 * avg > half_patch ? 4.0 * avg : half_patch;  // :308
    const double th_dist = half_patch * 4.0;                                // :312
    int kept = 0;
    for (int i = 0; i < n; i++) {  // :318
        const bool ok = status_pm[i] && pix_err[i] < th_pix && dist_pred[i] < th_dist;
        status_out[i] = ok ? 1 : 0;
        if (!ok) continue;
        if (pt_predict && pt_pm) {
            pt_predict[2 * i] = pt_pm[2 * i];
            pt_predict[2 * i + 1] = pt_pm[2 * i + 1];
        }
        if (pt_predict_un && pt_pm_un) {
            pt_predict_un[2 * i] = pt_pm_un[2 * i];
            pt_predict_un[2 * i + 1] = pt_pm_un[2 * i + 1];
        }
        kept++;
    }
    return kept;
}


// ---- hipGraph capture of the per-frame work ---------------------------------------------------
// BASELINE configs[4] ("hipGraph-captured iterate"): a camera stream issues the same launches with the same
// device pointers every frame (new frame written into a fixed device buffer -> pyramid -> prediction ->
// tracking -> scoring), so they are recorded once and replayed with one hipGraphLaunch.
int pagk_graph_begin(pagk_ctx *ctx)
{
    if (!ctx || ctx->capturing) return PAGK_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    destroy_segs(ctx->cap_segs);
    batch_desc_free(ctx->cap_batch);
    ctx->cap_batch_used = 0;
    if (ctx->batch_seen) {   // a context that leads batched launches: the capture's own descriptor pairs (no allocation inside a capture)
        ctx->cap_batch.resize(pagk_ctx::kBatchPerCapture);
        for (auto &d : ctx->cap_batch)
            if (int ar = batch_desc_alloc(ctx, d)) {
                batch_desc_free(ctx->cap_batch);
                return ar;
            }
    }
    HIPCHK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    return PAGK_OK;
}

int pagk_graph_end(pagk_ctx *ctx, int32_t *graph_id)
{
    if (!ctx || !ctx->capturing || !graph_id) return PAGK_E_ARG;
    ctx->capturing = false;
    hipGraph_t g = nullptr;
    HIPCHK(ctx, hipStreamEndCapture(ctx->stream, &g));
    int id = -1;
    for (int k = 0; k < pagk_ctx::kGraphs; k++)
        if (!ctx->graph_execs[k]) {
            id = k;
            break;
        }
    if (id < 0 || !g) {
        if (g) (void)hipGraphDestroy(g);
        destroy_segs(ctx->cap_segs);
        batch_desc_free(ctx->cap_batch);
        snprintf(ctx->err, sizeof(ctx->err), id < 0 ? "all %d graph slots are in use" : "capture produced no graph (%d)", pagk_ctx::kGraphs);
        return PAGK_E_ARG;
    }
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        destroy_segs(ctx->cap_segs);
        batch_desc_free(ctx->cap_batch);
        snprintf(ctx->err, sizeof(ctx->err), "hipGraphInstantiate -> %s", hipGetErrorString(e));
        return PAGK_E_HIP;
    }
    ctx->graphs[id] = g;
    ctx->graph_execs[id] = ex;
    ctx->pre_segs[id] = std::move(ctx->cap_segs);   // (empty for a capture without a live finisher: one graph, as before)
    ctx->cap_segs.clear();
    batch_desc_free(ctx->graph_batch[id]);
    ctx->graph_batch[id] = std::move(ctx->cap_batch);   // (unused reserved pairs go with it: freed with the graph)
    ctx->cap_batch.clear();
    ctx->cap_batch_used = 0;
    *graph_id = id;
    return PAGK_OK;
}

int pagk_graph_launch(pagk_ctx *ctx, int32_t graph_id)
{
    if (!ctx || ctx->capturing || graph_id < 0 || graph_id >= pagk_ctx::kGraphs || !ctx->graph_execs[graph_id]) return PAGK_E_ARG;
    for (pagk_ctx::GraphSeg &sg : ctx->pre_segs[graph_id]) {
        if (sg.kind == pagk_ctx::GraphSeg::GRAPH) {
            HIPCHK(ctx, hipGraphLaunch(sg.ex, ctx->stream));
        } else if (sg.kind == pagk_ctx::GraphSeg::FINISHER) {
            // the live finisher beside the throughput kernel of the next segment: the fork of a direct launch
            HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
            void *kargs[] = {&sg.args};
            HIPCHK(ctx, hipLaunchKernel(sg.fn, dim3(sg.grid), dim3(kBlock), kargs, sg.lds, ctx->aux_stream));
            HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->aux_stream));
        } else {
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        }
    }
    HIPCHK(ctx, hipGraphLaunch(ctx->graph_execs[graph_id], ctx->stream));
    return PAGK_OK;
}

int pagk_graph_destroy(pagk_ctx *ctx, int32_t graph_id)
{
    if (!ctx || graph_id < 0 || graph_id >= pagk_ctx::kGraphs || !ctx->graph_execs[graph_id]) return PAGK_E_ARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->aux_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->aux_stream));
    (void)hipGraphExecDestroy(ctx->graph_execs[graph_id]);
    (void)hipGraphDestroy(ctx->graphs[graph_id]);
    destroy_segs(ctx->pre_segs[graph_id]);
    batch_desc_free(ctx->graph_batch[graph_id]);
    ctx->graph_execs[graph_id] = nullptr;
    ctx->graphs[graph_id] = nullptr;
    return lv_check(ctx);
}

// ---- geometry validation scoring ----------------------------------
int pagk_geometry_scores_device(pagk_ctx *ctx, const double *H21, const double *H12, const double *F21,
                                int32_t n, const float *d_pts1, const float *d_pts2, float sigma,
                                uint8_t *d_inliers_H, uint8_t *d_inliers_F, float *d_scores)
{
    if (!ctx || !H21 || !H12 || !F21 || n < 0 || !d_scores) return PAGK_E_ARG;
    if (n > 0 && (!d_pts1 || !d_pts2 || !d_inliers_H || !d_inliers_F)) return PAGK_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ScoreArgs a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < 9; k++) a.H21[k] = H21[k], a.H12[k] = H12[k], a.F21[k] = F21[k];
    a.pts1 = d_pts1, a.pts2 = d_pts2, a.n = n, a.sigma = sigma;
    a.inl_H = d_inliers_H, a.inl_F = d_inliers_F, a.scores = d_scores;
    hipLaunchKernelGGL(k_geometry_scores, dim3(2), dim3(256), 0, ctx->stream, a);  // n == 0: scores = 0
    HIPCHK(ctx, hipGetLastError());
    return PAGK_OK;
}

int pagk_geometry_scores(pagk_ctx *ctx, const double *H21, const double *H12, const double *F21, int32_t n,
                         const float *pts1, const float *pts2, float sigma, uint8_t *inliers_H,
                         uint8_t *inliers_F, float *score_H, float *score_F)
{
    if (!ctx || !H21 || !H12 || !F21 || n < 0 || !score_H || !score_F) return PAGK_E_ARG;
    NOT_WHILE_CAPTURING(ctx, "pagk_geometry_scores");
    if (n > 0 && (!pts1 || !pts2 || !inliers_H || !inliers_F)) return PAGK_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)(n < 1 ? 1 : n);
    const size_t o_p2 = align_up(nn * 8, 256), o_h = o_p2 + align_up(nn * 8, 256), o_f = o_h + align_up(nn, 256);
    const size_t o_s = o_f + align_up(nn, 256), total = o_s + 256;
    if (total > ctx->score.bytes) {
        if (ctx->score.block) HIPCHK(ctx, hipFree(ctx->score.block));
        ctx->score.block = nullptr;
        ctx->score.bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->score.block, total));
        ctx->score.bytes = total;
    }
    uint8_t *b = static_cast<uint8_t *>(ctx->score.block);
    float *d_p1 = reinterpret_cast<float *>(b), *d_p2 = reinterpret_cast<float *>(b + o_p2);
    float *d_s = reinterpret_cast<float *>(b + o_s);
    if (n > 0) {
        HIPCHK(ctx, hipMemcpyAsync(d_p1, pts1, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(d_p2, pts2, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    int rc = pagk_geometry_scores_device(ctx, H21, H12, F21, n, d_p1, d_p2, sigma, b + o_h, b + o_f, d_s);
    if (rc) return rc;
    float sc[2];
    if (n > 0) {
        HIPCHK(ctx, hipMemcpyAsync(inliers_H, b + o_h, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(inliers_F, b + o_f, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipMemcpyAsync(sc, d_s, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *score_H = sc[0];
    *score_F = sc[1];
    return PAGK_OK;
}

// src/gyro_aided_tracker.cpp:462-465: `float RH = ...; if (RH > 0.45)` compares in double.
int pagk_geometry_select(float score_H, float score_F)
{
    const float RH = score_H / (score_F + score_H);
    return RH > 0.45 ? 1 : 0;
}

int pagk_geometry_validation(pagk_ctx *ctx, const double *H21, const double *H12, const double *F21,
                             int32_t n, const float *pt_ref_un, const float *pt_predict_un,
                             uint8_t *status, float sigma, float *track_score)
{
    if (!ctx || !H21 || !H12 || !F21 || n < 0) return PAGK_E_ARG;
    if (n > 0 && (!pt_ref_un || !pt_predict_un || !status)) return PAGK_E_ARG;
    if (track_score) *track_score = 0;  // :447
    try {
    std::vector<int> idx;               // :433-440
    std::vector<float> p1, p2;
    for (int i = 0; i < n; i++)
        if (status[i]) {
            idx.push_back(i);
            p1.push_back(pt_ref_un[2 * i]), p1.push_back(pt_ref_un[2 * i + 1]);
            p2.push_back(pt_predict_un[2 * i]), p2.push_back(pt_predict_un[2 * i + 1]);
        }
    const int m = (int)idx.size();
    if (m <= 8) return 0;  // :445
    std::vector<uint8_t> inH((size_t)m), inF((size_t)m);
    float sH = 0, sF = 0;
    int rc = pagk_geometry_scores(ctx, H21, H12, F21, m, p1.data(), p2.data(), sigma, inH.data(), inF.data(), &sH, &sF);
    if (rc) return rc;
    const bool useH = pagk_geometry_select(sH, sF) != 0;  // :462-470
    const std::vector<uint8_t> &in = useH ? inH : inF;
    if (track_score) *track_score = useH ? sH : sF;
    int cnt_inlier = 0;
    for (int k = 0; k < m; k++) {  // :472-480
        if (!in[k])
            status[idx[k]] = 0;
        else
            cnt_inlier++;
    }
    return cnt_inlier;
    } catch (const std::bad_alloc &) {  // nothing crosses the C ABI
        return PAGK_E_NOMEM;
    }
}

// ---- NCC nearest-neighbour matching ---------------------------------
static int near_neighbors_launch(pagk_ctx *ctx, const FrameSlot &sr, const FrameSlot &sc, int32_t half_patch, int32_t n,
                                 const float *d_keys_ref, const float *d_pt_predict_un, const uint8_t *d_status,
                                 const float *d_affine, int32_t m, const float *d_keys_cur, const float *d_keys_cur_un,
                                 int32_t level, float radius_unit, int32_t use_ncc, int32_t pairs, int32_t cap,
                                 int32_t *d_count, int32_t *d_nbr_idx, float *d_nbr_dist, float *d_nbr_ncc)
{
    NeighborArgs a;
    memset(&a, 0, sizeof a);
    fill_level(a.ref0, sr, 0);
    fill_level(a.cur0, sc, 0);
    a.pad_ref = sr.pad0;
    a.pad_cur = sc.pad0;
    a.half = half_patch, a.n = n, a.m = m, a.cap = cap, a.level = level, a.use_ncc = use_ncc, a.pairs = pairs;
    a.radius = (float)level * radius_unit;  // src/gyro_aided_tracker.cpp:811  int * float
    a.keys_ref = d_keys_ref, a.pt_pred = d_pt_predict_un, a.affine = d_affine, a.status = d_status;
    a.keys_cur = d_keys_cur, a.keys_cur_un = d_keys_cur_un;
    a.count = d_count, a.nbr_idx = d_nbr_idx, a.nbr_dist = d_nbr_dist, a.nbr_ncc = d_nbr_ncc;
    if (n <= 0) return PAGK_OK;
    const int P = (2 * half_patch + 1) * (2 * half_patch + 1);
    const int nr = (P + 255) / 256, tail = P % 32;
    const size_t lds = neighbor_lds_bytes(half_patch, cap);
    if (lds > 64 * 1024) {
        snprintf(ctx->err, sizeof(ctx->err), "neighbour capacity %d needs %zu bytes of LDS (limit 64 KiB)", cap, lds);
        return PAGK_E_ARG;
    }
    auto launch = [&](auto kern) -> hipError_t {
        hipLaunchKernelGGL(kern, dim3(n), dim3(256), lds, ctx->stream, a);
        return hipGetLastError();
    };
    hipError_t e = hipErrorInvalidValue;
    switch (nr * 100 + tail) {  // (NR, TAIL) as for k_track_block
        case 101: e = launch(k_near_neighbors<1, 1>); break;
        case 109: e = launch(k_near_neighbors<1, 9>); break;
        case 117: e = launch(k_near_neighbors<1, 17>); break;
        case 125: e = launch(k_near_neighbors<1, 25>); break;
        case 201: e = launch(k_near_neighbors<2, 1>); break;
        case 209: e = launch(k_near_neighbors<2, 9>); break;
        case 225: e = launch(k_near_neighbors<2, 25>); break;
        case 317: e = launch(k_near_neighbors<3, 17>); break;
        case 325: e = launch(k_near_neighbors<3, 25>); break;
        case 409: e = launch(k_near_neighbors<4, 9>); break;
        case 401: e = launch(k_near_neighbors<4, 1>); break;
        default: break;
    }
    HIPCHK(ctx, e);
    return PAGK_OK;
}

int pagk_near_neighbors_device(pagk_ctx *ctx, int32_t slot_ref, int32_t slot_cur, int32_t half_patch, int32_t n,
                               const float *d_keys_ref, const float *d_pt_predict_un, const uint8_t *d_status,
                               const float *d_affine, int32_t m, const float *d_keys_cur, const float *d_keys_cur_un,
                               int32_t level, float radius_unit, int32_t use_ncc, int32_t cap, int32_t *d_count,
                               int32_t *d_nbr_idx, float *d_nbr_dist, float *d_nbr_ncc)
{
    if (!ctx || slot_ref < 0 || slot_ref >= kUserSlots || slot_cur < 0 || slot_cur >= kUserSlots) return PAGK_E_ARG;
    if (half_patch < 1 || half_patch > PAGK_MAX_HALF_PATCH || n < 0 || m < 0 || cap < 1 || level < 0) return PAGK_E_ARG;
    if (n > 0 && (!d_keys_ref || !d_pt_predict_un || !d_status || !d_count || !d_nbr_idx || !d_nbr_dist || !d_nbr_ncc))
        return PAGK_E_ARG;
    if (n > 0 && m > 0 && (!d_keys_cur || !d_keys_cur_un)) return PAGK_E_ARG;
    const FrameSlot &sr = ctx->slots[slot_ref], &sc = ctx->slots[slot_cur];
    if (!sr.valid || !sc.valid) return PAGK_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return near_neighbors_launch(ctx, sr, sc, half_patch, n, d_keys_ref, d_pt_predict_un, d_status, d_affine, m, d_keys_cur,
                                 d_keys_cur_un, level, radius_unit, use_ncc, 0, cap, d_count, d_nbr_idx, d_nbr_dist,
                                 d_nbr_ncc);
}

// host-buffer forms: both frames go through the scratch slots (level 0 only), the per-feature arrays through one
// device block; synchronous
static int neighbors_host(pagk_ctx *ctx, const pagk_image *ref, const pagk_image *cur, int32_t half_patch, int32_t n,
                          const float *keys_ref, const float *pt_predict_un, const uint8_t *status, const float *affine,
                          int32_t m, const float *keys_cur, const float *keys_cur_un, int32_t level, float radius_unit,
                          int32_t use_ncc, int32_t pairs, int32_t cap, int32_t *count, int32_t *nbr_idx, float *nbr_dist,
                          float *nbr_ncc)
{
    int rc;
    if ((rc = check_image(ref)) || (rc = check_image(cur))) return rc;
    if ((rc = frame_upload_any(ctx, 4, ref, 1)) || (rc = frame_upload_any(ctx, 5, cur, 1))) return rc;
    const size_t nn = (size_t)(n < 1 ? 1 : n), mm = (size_t)(m < 1 ? 1 : m), cc = (size_t)cap;
    size_t off[11], total = 0;
    const size_t sizes[11] = {nn * 8, nn * 8, nn, nn * 16, mm * 8, mm * 8, nn * 4, nn * cc * 4, nn * cc * 4, nn * cc * 4, 0};
    for (int k = 0; k < 11; k++) {
        off[k] = total;
        total = align_up(total + sizes[k], 256);
    }
    if (total > ctx->score.bytes) {
        if (ctx->score.block) HIPCHK(ctx, hipFree(ctx->score.block));
        ctx->score.block = nullptr;
        ctx->score.bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->score.block, total));
        ctx->score.bytes = total;
    }
    uint8_t *b = static_cast<uint8_t *>(ctx->score.block);
    auto up = [&](int k, const void *src, size_t bytes) -> int {
        if (src && bytes) HIPCHK(ctx, hipMemcpyAsync(b + off[k], src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return PAGK_OK;
    };
    if (n > 0) {
        if ((rc = up(0, keys_ref, (size_t)n * 8)) || (rc = up(1, pt_predict_un, (size_t)n * 8)) ||
            (rc = up(2, status, (size_t)n)) || (rc = up(3, affine, (size_t)n * 16)) || (rc = up(6, count, (size_t)n * 4)))
            return rc;
    }
    if (m > 0 && ((rc = up(4, keys_cur, (size_t)m * 8)) || (rc = up(5, keys_cur_un, (size_t)m * 8)))) return rc;
    // the lists of skipped features (status 0, or already filled at a smaller radius) keep the caller's content
    if (!pairs && n > 0 && ((rc = up(7, nbr_idx, (size_t)n * cc * 4)) || (rc = up(8, nbr_dist, (size_t)n * cc * 4)) ||
                            (rc = up(9, nbr_ncc, (size_t)n * cc * 4))))
        return rc;
    rc = near_neighbors_launch(ctx, ctx->slots[4], ctx->slots[5], half_patch, n, reinterpret_cast<float *>(b + off[0]),
                               reinterpret_cast<float *>(b + off[1]), b + off[2],
                               affine ? reinterpret_cast<float *>(b + off[3]) : nullptr, m,
                               reinterpret_cast<float *>(b + off[4]), reinterpret_cast<float *>(b + off[5]), level,
                               radius_unit, use_ncc, pairs, cap, reinterpret_cast<int32_t *>(b + off[6]),
                               reinterpret_cast<int32_t *>(b + off[7]), reinterpret_cast<float *>(b + off[8]),
                               reinterpret_cast<float *>(b + off[9]));
    if (rc) return rc;
    if (n > 0) {
        HIPCHK(ctx, hipMemcpyAsync(count, b + off[6], (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (nbr_idx) HIPCHK(ctx, hipMemcpyAsync(nbr_idx, b + off[7], (size_t)n * cc * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (nbr_dist) HIPCHK(ctx, hipMemcpyAsync(nbr_dist, b + off[8], (size_t)n * cc * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(nbr_ncc, b + off[9], (size_t)n * cc * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PAGK_OK;
}

int pagk_find_near_neighbors(pagk_ctx *ctx, const pagk_image *ref, const pagk_image *cur, int32_t half_patch, int32_t n,
                             const float *keys_ref, const float *pt_predict_un, const uint8_t *status,
                             const float *affine, int32_t m, const float *keys_cur, const float *keys_cur_un,
                             int32_t level, float radius_unit, int32_t use_ncc, int32_t cap, int32_t *count,
                             int32_t *nbr_idx, float *nbr_dist, float *nbr_ncc)
{
    if (!ctx) return PAGK_E_ARG;
    NOT_WHILE_CAPTURING(ctx, "pagk_find_near_neighbors");
    if (half_patch < 1 || half_patch > PAGK_MAX_HALF_PATCH || n < 0 || m < 0 || cap < 1 || level < 0) return PAGK_E_ARG;
    if (n > 0 && (!keys_ref || !pt_predict_un || !status || !count || !nbr_idx || !nbr_dist || !nbr_ncc)) return PAGK_E_ARG;
    if (n > 0 && m > 0 && (!keys_cur || !keys_cur_un)) return PAGK_E_ARG;
    int rc = neighbors_host(ctx, ref, cur, half_patch, n, keys_ref, pt_predict_un, status, affine, m, keys_cur, keys_cur_un,
                            level, radius_unit, use_ncc, 0, cap, count, nbr_idx, nbr_dist, nbr_ncc);
    if (rc) return rc;
    for (int i = 0; i < n; i++)
        if (count[i] > cap) {
            snprintf(ctx->err, sizeof(ctx->err), "feature %d has %d neighbours, capacity is %d", i, count[i], cap);
            return PAGK_E_CAPACITY;
        }
    return PAGK_OK;
}

int pagk_ncc_free(pagk_ctx *ctx, const pagk_image *ref, const pagk_image *cur, int32_t half_patch, int32_t n,
                  const float *pt_ref, const float *pt_cur, const float *affine, float *ncc)
{
    if (!ctx) return PAGK_E_ARG;
    NOT_WHILE_CAPTURING(ctx, "pagk_ncc_free");
    if (half_patch < 1 || half_patch > PAGK_MAX_HALF_PATCH || n < 0) return PAGK_E_ARG;
    if (n > 0 && (!pt_ref || !pt_cur || !ncc)) return PAGK_E_ARG;
    if (n == 0) return PAGK_OK;
    try {
        std::vector<uint8_t> st((size_t)n, 1);
        std::vector<int32_t> cnt((size_t)n, 0);
        // pairs mode: feature i's only candidate is point i of pt_cur; capacity 1
        return neighbors_host(ctx, ref, cur, half_patch, n, pt_ref, pt_ref, st.data(), affine, n, pt_cur, pt_cur, 0, 0.0f, 1, 1,
                              1, cnt.data(), nullptr, nullptr, ncc);
    } catch (const std::bad_alloc &) {
        return PAGK_E_NOMEM;
    }
}

// ---- diagnostics: the solve's arithmetic on arbitrary operands (pagk_selftest_kernel.h) --------------------------
namespace {
// n doubles-per-item arrays in, copied to the device as one block; returns device pointers through `d`
struct Scratch {
    void *p = nullptr;
    ~Scratch() { if (p) (void)hipFree(p); }
};
}  // namespace

int pagk_selftest_divide(pagk_ctx *ctx, int32_t n, const double *num, const double *den, double *q_plain,
                         double *q_prepared, double *root, double *root_lean)
{
    if (!ctx || n < 0) return PAGK_E_ARG;
    NOT_WHILE_CAPTURING(ctx, "pagk_selftest_divide");
    if (n == 0) return PAGK_OK;
    if (!num || !den || !q_plain || !q_prepared || !root || !root_lean) return PAGK_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch s;
    const size_t nb = (size_t)n * sizeof(double);
    HIPCHK(ctx, hipMalloc(&s.p, 6 * nb));
    double *d = static_cast<double *>(s.p);
    HIPCHK(ctx, hipMemcpyAsync(d, num, nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d + n, den, nb, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_selftest_divide, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, d, d + n, d + 2 * (size_t)n,
                       d + 3 * (size_t)n, d + 4 * (size_t)n, d + 5 * (size_t)n);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(q_plain, d + 2 * (size_t)n, nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(q_prepared, d + 3 * (size_t)n, nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(root, d + 4 * (size_t)n, nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(root_lean, d + 5 * (size_t)n, nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PAGK_OK;
}

int pagk_selftest_repeat_sum(pagk_ctx *ctx, int32_t n, const float *c, int32_t count, double *closed, double *loop)
{
    if (!ctx || n < 0 || count <= 57 || count > 480) return PAGK_E_ARG;
    NOT_WHILE_CAPTURING(ctx, "pagk_selftest_repeat_sum");
    if (n == 0) return PAGK_OK;
    if (!c || !closed || !loop) return PAGK_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch s;
    const size_t nn = (size_t)n;
    HIPCHK(ctx, hipMalloc(&s.p, nn * (2 * sizeof(double) + sizeof(float))));
    double *d_closed = static_cast<double *>(s.p), *d_loop = d_closed + nn;
    float *d_c = reinterpret_cast<float *>(d_loop + nn);
    HIPCHK(ctx, hipMemcpyAsync(d_c, c, nn * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_selftest_repeat_sum, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, d_c, count, d_closed, d_loop);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(closed, d_closed, nn * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(loop, d_loop, nn * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PAGK_OK;
}

int pagk_selftest_solve(pagk_ctx *ctx, int32_t n, const double *H, const double *b, uint32_t solver_variant,
                        double *x_serial, double *norm_serial, double *x_lanes, double *nsq_lanes)
{
    if (!ctx || n < 0) return PAGK_E_ARG;
    NOT_WHILE_CAPTURING(ctx, "pagk_selftest_solve");
    if (solver_variant & ~(SV_LOWER_SEQ | SV_UPPER_TREE | SV_NORM_SEQ | SV_LLT_RECIP | SV_PIVOT_TREE)) return PAGK_E_ARG;
    if (n == 0) return PAGK_OK;
    if (!H || !b || !x_serial || !norm_serial || !x_lanes || !nsq_lanes) return PAGK_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch s;
    const size_t nn = (size_t)n;
    HIPCHK(ctx, hipMalloc(&s.p, (16 + 4 + 4 + 1 + 4 + 1) * nn * sizeof(double)));
    double *dH = static_cast<double *>(s.p), *db = dH + 16 * nn, *dxs = db + 4 * nn, *dns = dxs + 4 * nn,
           *dxl = dns + nn, *dnl = dxl + 4 * nn;
    HIPCHK(ctx, hipMemcpyAsync(dH, H, 16 * nn * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(db, b, 4 * nn * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_selftest_solve, dim3((n + 15) / 16), dim3(64), 0, ctx->stream, n, dH, db, solver_variant, dxs, dns,
                       dxl, dnl);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(x_serial, dxs, 4 * nn * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(norm_serial, dns, nn * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(x_lanes, dxl, 4 * nn * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(nsq_lanes, dnl, nn * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PAGK_OK;
}

// GyroAidedTracker::MatchFeatures, src/gyro_aided_tracker.cpp:949-1008.  Host-side: a sequential pass whose
// decisions depend on the matches accepted so far.
int pagk_match_features(int32_t n, int32_t cap, const int32_t *count, const int32_t *nbr_idx, const float *nbr_dist,
                        const float *nbr_ncc, int32_t use_ncc, int32_t *match_query, int32_t *match_train,
                        float *match_dist, float *match_ncc)
{
    if (n < 0 || cap < 1) return PAGK_E_ARG;
    if (n > 0 && (!count || !nbr_idx || !nbr_dist || !nbr_ncc || !match_query || !match_train)) return PAGK_E_ARG;
    const float TH_NCC_HIGH = 0.6f, TH_NCC_LOW = 0.3f, TH_RATIO = 0.75f;  // :7-9
    struct M {
        int q, t;
        float d, c;
    };
    try {
        std::vector<M> matches;
        std::vector<int> found;  // sFoundInCurPts (:951): indices stay in it after their matches are erased
        for (int i = 0; i < n; i++) {
            const int c = count[i];
            if (c <= 0) continue;  // :955
            if (c > cap) return PAGK_E_CAPACITY;
            const float *ncc = nbr_ncc + (size_t)i * cap, *dist = nbr_dist + (size_t)i * cap;
            if (use_ncc) {  // :959-975
                if (!(ncc[0] > TH_NCC_HIGH)) {
                    if (c > 1) {
                        if (ncc[0] < TH_NCC_LOW) continue;
                        if (!(ncc[1] < ncc[0] * TH_RATIO)) continue;  // the two best are too similar
                    } else
                        continue;
                }
            } else if (c > 1 && !(dist[0] < dist[1] * TH_RATIO)) {  // :977-989
                continue;
            }
            const M m{i, nbr_idx[(size_t)i * cap], dist[0], ncc[0]};
            bool seen = false;
            for (int t : found) seen = seen || t == m.t;
            if (!seen) {  // :991-994
                matches.push_back(m);
                found.push_back(m.t);
            } else {      // :995-1005
                size_t w = 0;
                for (size_t r = 0; r < matches.size(); r++)
                    if (matches[r].t != m.t) matches[w++] = matches[r];
                matches.resize(w);
            }
        }
        for (size_t k = 0; k < matches.size(); k++) {
            match_query[k] = matches[k].q;
            match_train[k] = matches[k].t;
            if (match_dist) match_dist[k] = matches[k].d;
            if (match_ncc) match_ncc[k] = matches[k].c;
        }
        return (int)matches.size();
    } catch (const std::bad_alloc &) {
        return PAGK_E_NOMEM;
    }
}

}  // extern "C"

#include "pagk_multi.h"
