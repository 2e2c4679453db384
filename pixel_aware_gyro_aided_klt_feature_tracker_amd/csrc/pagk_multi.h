// pagk_multi.h -- the feature-sharded path behind the C ABI (include/pagk.h, "sharded over the GPUs of one node").
//
// The path shards by independent units: a feature's Gauss-Newton loop reads the two (read-only, replicated)
// pyramids and its own 33 bytes of input (reference src/patch_match.cpp:167-367, the cv::parallel_for_ of :103).
// Rank r owns the contiguous index block [r * ceil(n / G), ...), every GPU builds both pyramids itself, and the
// only exchange is ONE all-gather of the packed per-rank result slice (RCCL ncclAllGather over xGMI), after which
// every GPU holds every result and the tracker's global post-filter (src/gyro_aided_tracker.cpp:289-341) sees them
// in index order.
//
// Two ways to form the group: pagk_multi_create (one process drives all GPUs: ncclCommInitAll, one stream per
// device, the collective issued inside ncclGroupStart/End) and pagk_multi_create_rank (one process per GPU --
// torchrun -- each with one device; the unique id travels through the host application's own channel).
//
// RCCL is loaded lazily with dlopen: single-GPU users never pay for the 570 MB library, and libpagk_hip.so has
// no link-time dependency on it.  Included by pagk_hip.hip (needs pagk_ctx and the launch helpers).
#pragma once
#include <dlfcn.h>

#include <mutex>

namespace {

// the handful of RCCL entry points used, with the types of rccl.h restated (ncclComm_t is opaque, ncclResult_t
// and ncclDataType_t are C enums: ncclSuccess = 0, ncclUint8 = 1)
struct Id128 {
    char internal[128];
};
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;                             // ncclGetUniqueId(ncclUniqueId*): 128 bytes
    int (*CommInitAll)(void **, int, const int *) = nullptr;          // ncclCommInitAll
    int (*CommInitRank)(void **, int, Id128 /* ncclUniqueId by value */, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;                        // ncclCommCount(comm, &count) (optional)
};

void rccl_load_once(Rccl &r, char *err, size_t errn)
{
    const char *names[] = {getenv("PAGK_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *nm : names) {
        if (!nm || !*nm) continue;
        r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) {
        snprintf(err, errn, "cannot load librccl.so: %s", dlerror());
        return;
    }
    auto sym = [&](const char *n) { return dlsym(r.lib, n); };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    if (!r.GetUniqueId || !r.CommInitAll || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GroupStart ||
        !r.GroupEnd) {
        snprintf(err, errn, "librccl.so lacks a required entry point");
        dlclose(r.lib);
        r.lib = nullptr;
    }
}

// Loaded once per process (std::call_once: contexts live on several host threads).  A failure is remembered WITH its
// reason and every later call reports that reason again.
Rccl *rccl_load(char *err, size_t errn)
{
    static Rccl r;
    static char first_error[256] = {0};
    static std::once_flag once;
    std::call_once(once, [] { rccl_load_once(r, first_error, sizeof first_error); });
    if (r.lib) return &r;
    if (err && errn) snprintf(err, errn, "%s", first_error[0] ? first_error : "librccl.so could not be loaded");
    return nullptr;
}

constexpr int kShardFields = 7;
// bytes per feature of the packed slice, SetMatcher order (src/patch_match.cpp:370-388) + the diagnostic count
constexpr size_t kShardElem[kShardFields] = {8, 8, 1, 8, 8, 4, 4};  // pt_un pt_dist status pix_err dist_pred ncc iters

}  // namespace

struct pagk_multi {
    int world = 1;        // ranks of the group
    int n_local = 0;      // ranks driven by this process (world for pagk_multi_create, 1 for _create_rank)
    int first_rank = 0;   // global rank of local member 0
    std::vector<pagk_ctx *> ctx;
    std::vector<void *> comm;
    Rccl *rccl = nullptr;
    // staging of pagk_track_sharded, per local member
    struct Stage {
        void *d_in = nullptr, *h_in = nullptr;      // this rank's block of the four input arrays
        void *d_slice = nullptr, *d_all = nullptr;  // packed result slice, gathered slices of all ranks
        void *h_all = nullptr;                      // pinned copy of d_all (member 0 only)
        size_t in_bytes = 0, slice_bytes = 0, all_bytes = 0;
    };
    std::vector<Stage> stage;
    char err[256] = {0};
};

extern "C" {

// Contiguous blocks of ceil(n / world) features, the last ones possibly short or empty.
void pagk_shard_range(int32_t n, int32_t rank, int32_t world, int32_t *lo, int32_t *hi)
{
    const int m = world > 0 ? (n + world - 1) / world : n;
    int l = rank * m;
    if (l > n) l = n;
    int h = l + m;
    if (h > n) h = n;
    if (lo) *lo = l;
    if (hi) *hi = h;
}

// Layout of one rank's packed slice for m features: seven SoA blocks, each padded to 8 bytes.
size_t pagk_shard_layout(int32_t m, size_t offsets[7])
{
    size_t total = 0;
    const size_t mm = (size_t)(m < 1 ? 1 : m);
    for (int k = 0; k < kShardFields; k++) {
        if (offsets) offsets[k] = total;
        total += (kShardElem[k] * mm + 7) / 8 * 8;
    }
    return total;
}

const char *pagk_multi_last_error(const pagk_multi *pm) { return pm ? pm->err : ""; }
int32_t pagk_multi_world(const pagk_multi *pm) { return pm ? pm->world : 0; }
int32_t pagk_multi_local(const pagk_multi *pm) { return pm ? pm->n_local : 0; }
// What RCCL itself says about the group: ncclCommCount of local member 0's communicator (the ranks the all-gather
// really spans), as opposed to the number the group was created with.  < 0: unavailable.
int32_t pagk_multi_comm_count(const pagk_multi *pm)
{
    if (!pm || !pm->rccl || !pm->rccl->CommCount || pm->comm.empty() || !pm->comm[0]) return PAGK_E_NCCL;
    int count = -1;
    if (pm->rccl->CommCount(pm->comm[0], &count) != 0) return PAGK_E_NCCL;
    return count;
}
pagk_ctx *pagk_multi_ctx(pagk_multi *pm, int32_t local_index)
{
    return (pm && local_index >= 0 && local_index < pm->n_local) ? pm->ctx[local_index] : nullptr;
}

void pagk_multi_destroy(pagk_multi *pm)
{
    if (!pm) return;
    for (int k = 0; k < pm->n_local; k++) {
        if (pm->ctx[k]) {
            (void)hipSetDevice(pm->ctx[k]->device);
            (void)hipStreamSynchronize(pm->ctx[k]->stream);
        }
        if ((size_t)k < pm->stage.size()) {
            pagk_multi::Stage &s = pm->stage[k];
            if (s.d_in) (void)hipFree(s.d_in);
            if (s.h_in) (void)hipHostFree(s.h_in);
            if (s.d_slice) (void)hipFree(s.d_slice);
            if (s.d_all) (void)hipFree(s.d_all);
            if (s.h_all) (void)hipHostFree(s.h_all);
        }
        if ((size_t)k < pm->comm.size() && pm->comm[k] && pm->rccl) (void)pm->rccl->CommDestroy(pm->comm[k]);
        if (pm->ctx[k]) pagk_destroy(pm->ctx[k]);
    }
    delete pm;
}

#define NCCLCHK(pm, call)                                                                            \
    do {                                                                                             \
        int r_ = (call);                                                                             \
        if (r_ != 0) {                                                                               \
            snprintf((pm)->err, sizeof((pm)->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call,     \
                     (pm)->rccl->GetErrorString ? (pm)->rccl->GetErrorString(r_) : "RCCL error");    \
            return PAGK_E_NCCL;                                                                      \
        }                                                                                            \
    } while (0)

static int multi_alloc(pagk_multi **out, int n_local)
{
    pagk_multi *pm = new (std::nothrow) pagk_multi();
    if (!pm) return PAGK_E_NOMEM;
    try {
        pm->ctx.assign((size_t)n_local, nullptr);
        pm->comm.assign((size_t)n_local, nullptr);
        pm->stage.resize((size_t)n_local);
    } catch (const std::bad_alloc &) {
        delete pm;
        return PAGK_E_NOMEM;
    }
    pm->n_local = n_local;
    *out = pm;
    return PAGK_OK;
}

// One process, n_devices GPUs (a C++ GyroAidedTracker host): contexts on the listed devices and one RCCL
// communicator over them (ncclCommInitAll).  A single device is a valid group (the all-gather degenerates to a copy).
int pagk_multi_create(pagk_multi **out, const int32_t *devices, int32_t n_devices)
{
    if (!out || !devices || n_devices < 1 || n_devices > 64) return PAGK_E_ARG;
    *out = nullptr;
    for (int a = 0; a < n_devices; a++)
        for (int b = a + 1; b < n_devices; b++)
            if (devices[a] == devices[b]) return PAGK_E_ARG;  // a device can be one rank only
    pagk_multi *pm = nullptr;
    int rc = multi_alloc(&pm, n_devices);
    if (rc) return rc;
    pm->world = n_devices;
    for (int k = 0; k < n_devices; k++)
        if ((rc = pagk_create(&pm->ctx[k], devices[k]))) {
            pagk_multi_destroy(pm);
            return rc;
        }
    pm->rccl = rccl_load(pm->err, sizeof pm->err);
    if (!pm->rccl) {
        fprintf(stderr, "pagk_multi_create: %s\n", pm->err);
        pagk_multi_destroy(pm);
        return PAGK_E_NCCL;
    }
    std::vector<int> devs(devices, devices + n_devices);
    int r = pm->rccl->CommInitAll(pm->comm.data(), n_devices, devs.data());
    if (r != 0) {
        fprintf(stderr, "pagk_multi_create: ncclCommInitAll -> %s\n", pm->rccl->GetErrorString ? pm->rccl->GetErrorString(r) : "?");
        pagk_multi_destroy(pm);
        return PAGK_E_NCCL;
    }
    *out = pm;
    return PAGK_OK;
}

// One process per GPU: rank 0 obtains the 128-byte id, the host application hands it to every rank (MPI, a file,
// torch.distributed ...), each rank joins with its own device.
int pagk_multi_unique_id(uint8_t id[128])
{
    if (!id) return PAGK_E_ARG;
    char err[256] = {0};
    Rccl *r = rccl_load(err, sizeof err);
    if (!r) {
        fprintf(stderr, "pagk_multi_unique_id: %s\n", err);
        return PAGK_E_NCCL;
    }
    return r->GetUniqueId(id) == 0 ? PAGK_OK : PAGK_E_NCCL;
}

int pagk_multi_create_rank(pagk_multi **out, const uint8_t id[128], int32_t rank, int32_t world, int32_t device)
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return PAGK_E_ARG;
    *out = nullptr;
    pagk_multi *pm = nullptr;
    int rc = multi_alloc(&pm, 1);
    if (rc) return rc;
    pm->world = world;
    pm->first_rank = rank;
    if ((rc = pagk_create(&pm->ctx[0], device))) {
        pagk_multi_destroy(pm);
        return rc;
    }
    pm->rccl = rccl_load(pm->err, sizeof pm->err);
    if (!pm->rccl) {
        fprintf(stderr, "pagk_multi_create_rank: %s\n", pm->err);
        pagk_multi_destroy(pm);
        return PAGK_E_NCCL;
    }
    Id128 uid;
    memcpy(uid.internal, id, 128);
    int r = pm->rccl->CommInitRank(&pm->comm[0], world, uid, rank);
    if (r != 0) {
        fprintf(stderr, "pagk_multi_create_rank: ncclCommInitRank -> %s\n", pm->rccl->GetErrorString ? pm->rccl->GetErrorString(r) : "?");
        pagk_multi_destroy(pm);
        return PAGK_E_NCCL;
    }
    *out = pm;
    return PAGK_OK;
}

// The exchange itself: every local member contributes `bytes` bytes at d_send[k] and receives world * bytes at
// d_recv[k] (rank order), on its context's stream -- after whatever that stream already holds, before whatever is
// issued next.  Asynchronous.  hip_streams: NULL (the contexts' streams) or one stream per local member.
int pagk_multi_allgather(pagk_multi *pm, const void *const *d_send, void *const *d_recv, size_t bytes,
                         void *const *hip_streams)
{
    if (!pm || !d_send || !d_recv) return PAGK_E_ARG;
    if (pm->n_local > 1) NCCLCHK(pm, pm->rccl->GroupStart());
    for (int k = 0; k < pm->n_local; k++) {
        if (hipSetDevice(pm->ctx[k]->device) != hipSuccess) return PAGK_E_HIP;
        hipStream_t st = hip_streams && hip_streams[k] ? static_cast<hipStream_t>(hip_streams[k]) : pm->ctx[k]->stream;
        NCCLCHK(pm, pm->rccl->AllGather(d_send[k], d_recv[k], bytes, /* ncclUint8 */ 1, pm->comm[k], st));
    }
    if (pm->n_local > 1) NCCLCHK(pm, pm->rccl->GroupEnd());
    return PAGK_OK;
}

#define MHIPCHK(pm, call)                                                                            \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            snprintf((pm)->err, sizeof((pm)->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call,     \
                     hipGetErrorString(e_));                                                         \
            return e_ == hipErrorOutOfMemory ? PAGK_E_NOMEM : PAGK_E_HIP;                            \
        }                                                                                            \
    } while (0)

// PatchMatch::OpticalFlowMultiLevel() (src/patch_match.cpp:79-142) with its feature loop (:103) split over the
// group's GPUs.  Same arguments and the same results, bit for bit, as pagk_track.  Single-process groups
// (pagk_multi_create) only: every local member uploads both frames, builds both pyramids, tracks its block; the
// packed slices are all-gathered; the host reads the gathered result from member 0.  Synchronous.
int pagk_track_sharded(pagk_multi *pm, const pagk_params *params, const pagk_image *ref, const pagk_image *cur,
                       int32_t n, const float *pt_ref_un, const float *pt_init_un, const float *affine,
                       const uint8_t *status_in, const pagk_outputs *out)
{
    if (!pm || pm->n_local != pm->world) return PAGK_E_ARG;  // needs every rank in this process
    int rc = check_params(params);
    if (rc) return rc;
    if ((rc = check_image(ref)) || (rc = check_image(cur))) return rc;
    if (ref->width != cur->width || ref->height != cur->height) return PAGK_E_ARG;
    if (n < 0 || !out || !out->pt_un || !out->status) return PAGK_E_ARG;
    if (n > 0 && (!pt_ref_un || !status_in)) return PAGK_E_ARG;
    if (n > 0 && params->has_gyro_predict_initial && !pt_init_un) return PAGK_E_ARG;
    if (n > 0 && params->consider_affine && !affine) return PAGK_E_ARG;
    const int G = pm->world, m = (n + G - 1) / G;
    size_t off[7];
    const size_t slice = pagk_shard_layout(m, off);
    const size_t mm = (size_t)(m < 1 ? 1 : m);
    const size_t in_off[4] = {0, mm * 8, mm * 16, mm * 32};  // pt_ref | pt_init | affine | status_in
    const size_t in_bytes = mm * 33;
    // (an error on member k leaves members 0 .. k-1 with launches in flight that read this call's staging: wait for them)
    auto drain = [&](int upto) {
        for (int j = 0; j < upto && j < G; j++) {
            (void)hipSetDevice(pm->ctx[j]->device);
            (void)hipStreamSynchronize(pm->ctx[j]->stream);
        }
    };
    // 1. every member: frames, pyramids, its block of the inputs, the tracking launch
    for (int k = 0; k < G; k++) {
        pagk_ctx *c = pm->ctx[k];
        pagk_multi::Stage &s = pm->stage[k];
        MHIPCHK(pm, hipSetDevice(c->device));
        if (s.in_bytes < in_bytes) {
            if (s.d_in) MHIPCHK(pm, hipFree(s.d_in));
            if (s.h_in) MHIPCHK(pm, hipHostFree(s.h_in));
            s.d_in = s.h_in = nullptr, s.in_bytes = 0;
            MHIPCHK(pm, hipMalloc(&s.d_in, in_bytes));
            MHIPCHK(pm, hipHostMalloc(&s.h_in, in_bytes, hipHostMallocDefault));
            s.in_bytes = in_bytes;
        }
        if (s.slice_bytes < slice) {
            if (s.d_slice) MHIPCHK(pm, hipFree(s.d_slice));
            s.d_slice = nullptr, s.slice_bytes = 0;
            MHIPCHK(pm, hipMalloc(&s.d_slice, slice));
            MHIPCHK(pm, hipMemset(s.d_slice, 0, slice));
            s.slice_bytes = slice;
        }
        if (s.all_bytes < slice * G) {
            if (s.d_all) MHIPCHK(pm, hipFree(s.d_all));
            if (s.h_all) MHIPCHK(pm, hipHostFree(s.h_all));
            s.d_all = s.h_all = nullptr, s.all_bytes = 0;
            MHIPCHK(pm, hipMalloc(&s.d_all, slice * G));
            if (k == 0) MHIPCHK(pm, hipHostMalloc(&s.h_all, slice * G, hipHostMallocDefault));
            s.all_bytes = slice * G;
        }
        if ((rc = frame_upload_any(c, 4, ref, params->pyramids)) || (rc = frame_upload_any(c, 5, cur, params->pyramids))) {
            snprintf(pm->err, sizeof pm->err, "rank %d: %s", k, c->err);
            drain(k);
            return rc;
        }
        int lo, hi;
        pagk_shard_range(n, k, G, &lo, &hi);
        const int nk = hi - lo;
        uint8_t *hb = static_cast<uint8_t *>(s.h_in), *db = static_cast<uint8_t *>(s.d_in);
        if (nk > 0) {
            memcpy(hb + in_off[0], pt_ref_un + 2 * (size_t)lo, (size_t)nk * 8);
            if (pt_init_un) memcpy(hb + in_off[1], pt_init_un + 2 * (size_t)lo, (size_t)nk * 8);
            if (affine) memcpy(hb + in_off[2], affine + 4 * (size_t)lo, (size_t)nk * 16);
            memcpy(hb + in_off[3], status_in + lo, (size_t)nk);
            MHIPCHK(pm, hipMemcpyAsync(db, hb, in_bytes, hipMemcpyHostToDevice, c->stream));
        }
        uint8_t *sl = static_cast<uint8_t *>(s.d_slice);
        pagk_outputs o;
        o.pt_un = reinterpret_cast<float *>(sl + off[0]);
        o.pt_dist = reinterpret_cast<float *>(sl + off[1]);
        o.status = sl + off[2];
        o.pix_err = reinterpret_cast<double *>(sl + off[3]);
        o.dist_pred = reinterpret_cast<double *>(sl + off[4]);
        o.ncc = reinterpret_cast<float *>(sl + off[5]);
        o.iters = reinterpret_cast<int32_t *>(sl + off[6]);
        rc = launch_track(c, params, c->slots[4], c->slots[5], nk, reinterpret_cast<float *>(db + in_off[0]),
                          pt_init_un ? reinterpret_cast<float *>(db + in_off[1]) : nullptr,
                          affine ? reinterpret_cast<float *>(db + in_off[2]) : nullptr, db + in_off[3], &o);
        if (rc) {
            snprintf(pm->err, sizeof pm->err, "rank %d: %s", k, c->err);
            drain(k);
            return rc;
        }
    }
    // 2. the one exchange of the path
    std::vector<const void *> snd((size_t)G);
    std::vector<void *> rcv((size_t)G);
    for (int k = 0; k < G; k++) snd[(size_t)k] = pm->stage[k].d_slice, rcv[(size_t)k] = pm->stage[k].d_all;
    if ((rc = pagk_multi_allgather(pm, snd.data(), rcv.data(), slice, nullptr))) {
        drain(G);
        return rc;
    }
    // 3. member 0 hands the gathered slices to the host; everybody drains
    MHIPCHK(pm, hipSetDevice(pm->ctx[0]->device));
    MHIPCHK(pm, hipMemcpyAsync(pm->stage[0].h_all, pm->stage[0].d_all, slice * G, hipMemcpyDeviceToHost, pm->ctx[0]->stream));
    int lv_rc = PAGK_OK;
    for (int k = 0; k < G; k++) {
        MHIPCHK(pm, hipSetDevice(pm->ctx[k]->device));
        MHIPCHK(pm, hipStreamSynchronize(pm->ctx[k]->stream));
        // a level-by-level launch whose wait ran out reports here, at the synchronisation that ends it (include/pagk.h)
        if (int r = lv_check(pm->ctx[k])) {
            snprintf(pm->err, sizeof pm->err, "rank %d: %s", k, pm->ctx[k]->err);
            lv_rc = r;
        }
    }
    if (lv_rc) return lv_rc;
    const uint8_t *all = static_cast<const uint8_t *>(pm->stage[0].h_all);
    for (int k = 0; k < G; k++) {
        int lo, hi;
        pagk_shard_range(n, k, G, &lo, &hi);
        const size_t nk = (size_t)(hi - lo);
        if (!nk) continue;
        const uint8_t *sl = all + (size_t)k * slice;
        memcpy(out->pt_un + 2 * (size_t)lo, sl + off[0], nk * 8);
        if (out->pt_dist) memcpy(out->pt_dist + 2 * (size_t)lo, sl + off[1], nk * 8);
        memcpy(out->status + lo, sl + off[2], nk);
        if (out->pix_err) memcpy(out->pix_err + lo, sl + off[3], nk * 8);
        if (out->dist_pred) memcpy(out->dist_pred + lo, sl + off[4], nk * 8);
        if (out->ncc) memcpy(out->ncc + lo, sl + off[5], nk * 4);
        if (out->iters) memcpy(out->iters + lo, sl + off[6], nk * 4);
    }
    return PAGK_OK;
}

}  // extern "C"
