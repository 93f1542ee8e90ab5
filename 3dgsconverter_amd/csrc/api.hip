// api.hip -- C-ABI entry points of libgsx_hip.so (see include/gsx_hip.h).
#include <algorithm>
#include <mutex>

#include "gsx_common.h"
#include "sor_grid_params.h"

namespace gsx {

static thread_local std::string g_err;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return 0;
    if (p) {
        GSX_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;  // a little headroom: fewer re-allocations when N drifts
    GSX_HIP(hipMalloc(&p, want));
    cap = want;
    return 0;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

int timing_begin(gsx_ctx *ctx, int slot)
{
    if (!ctx->timing || !((ctx->timing_mask >> slot) & 1u)) return 0;
    TimingSlot &s = ctx->slots[slot];
    if (s.used + 2 > s.ev.size()) {
        for (int i = 0; i < 64; ++i) {
            hipEvent_t e;
            GSX_HIP(hipEventCreate(&e));
            s.ev.push_back(e);
        }
    }
    GSX_HIP(hipEventRecord(s.ev[s.used], ctx->stream));
    return 0;
}

int timing_end(gsx_ctx *ctx, int slot)
{
    if (!ctx->timing || !((ctx->timing_mask >> slot) & 1u)) return 0;
    TimingSlot &s = ctx->slots[slot];
    GSX_HIP(hipEventRecord(s.ev[s.used + 1], ctx->stream));
    s.used += 2;
    return 0;
}

static int timing_resolve(gsx_ctx *ctx, int slot)
{
    TimingSlot &s = ctx->slots[slot];
    for (size_t i = 0; i + 1 < s.used; i += 2) {
        float ms = 0.f;
        GSX_HIP(hipEventElapsedTime(&ms, s.ev[i], s.ev[i + 1]));
        s.total_ms += ms;
        s.launches += 1;
    }
    s.used = 0;
    return 0;
}

// kernels implemented in the other translation units
int launch_pack_points(gsx_ctx *, const float *, const float *, const float *, int64_t, int64_t, float4 *);  // flags non-finite input in ctx->devflags
int launch_knn_brute(gsx_ctx *, const float4 *, int64_t, int64_t, int64_t, const unsigned *, const unsigned *,
                     int64_t, int, float *);
int knn_tree_info(gsx_ctx *, gsx_sor_info *);
int launch_knn_tree(gsx_ctx *, const float *, const float *, const float *, int64_t, int64_t, int64_t, int64_t, int, float *, double *,
                    gsx_sor_info *, int64_t, int, int, bool);
int launch_knn_grid(gsx_ctx *, const float *, const float *, const float *, int64_t, int64_t, int64_t, int64_t, int,
                    float *, gsx_sor_info *, int share = 0, int nshares = 1);
int launch_sor_stats(gsx_ctx *, const float *, int64_t, double, float *);
int launch_sor_mask(gsx_ctx *, const float *, int64_t, const float *, uint8_t *);
int density_voxels_dev(gsx_ctx *, const float *, const float *, const float *, int64_t, int64_t, double, int64_t, int64_t,
                       int64_t *, int64_t *, int64_t *, int64_t *);
int density_hist_dev(gsx_ctx *, const float *, const float *, const float *, int64_t, int64_t, double, int64_t, int64_t *, int64_t *,
                     int64_t *);
int density_merge_dev(gsx_ctx *, const int64_t *, const int64_t *, int64_t, int64_t, int64_t, int64_t *, int64_t *, int64_t *, int64_t *);
int density_filter_dev(gsx_ctx *, const float *, const float *, const float *, int64_t, int64_t, double, int64_t, int, const float *,
                       uint8_t *, gsx_density_info *);
int density_mask_dev(gsx_ctx *, const float *, const float *, const float *, int64_t, int64_t, double, const int64_t *,
                     int64_t, uint8_t *);
int kmeans_lloyd_dev(gsx_ctx *, const float *, int64_t, int, int, int, float *, int32_t *);
int kmeans_lloyd_batch_dev(gsx_ctx *, const float *, const int64_t *, int, int, int, int, float *, int32_t *);
int quantize_dev(gsx_ctx *, const float *, int64_t, const float *, int, uint8_t *);

}  // namespace gsx

using namespace gsx;

extern "C" {

const char *gsx_version(void) { return "gsx-hip 0.1 (gfx950)"; }
const char *gsx_last_error(void) { return g_err.c_str(); }

int gsx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int gsx_device_uid(int device, char *out, int cap)
{
    if (!out || cap < 16) GSX_FAIL("gsx_device_uid: need a buffer of at least 16 bytes");
    if (hipDeviceGetPCIBusId(out, cap, device) != hipSuccess) {
        (void)hipGetLastError();
        GSX_FAIL("gsx_device_uid: no such device");
    }
    return 0;
}

int gsx_ctx_create(int device, gsx_ctx **out)
{
    if (!out) GSX_FAIL("gsx_ctx_create: null out");
    int n = gsx_device_count();
    if (n <= 0) GSX_FAIL("gsx_ctx_create: no HIP device visible");
    if (device < 0 || device >= n) GSX_FAIL("gsx_ctx_create: device %d out of range (%d visible)", device, n);
    GSX_HIP(hipSetDevice(device));
    gsx_ctx *c = new gsx_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    GSX_HIP(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        delete c;
        GSX_FAIL("gsx_ctx_create: device %d is %s; this library is built for gfx950 (MI355X) only", device,
                 prop.gcnArchName);
    }
    if (c->devflags.reserve(64) != 0 || hipMemset(c->devflags.p, 0, 64) != hipSuccess) {
        delete c;
        GSX_FAIL("gsx_ctx_create: cannot allocate the device error word");
    }
    *out = c;
    return 0;
}

int gsx_ctx_check(gsx_ctx *c)
{
    if (!c) GSX_FAIL("null ctx");
    GSX_HIP(hipSetDevice(c->device));
    unsigned flags = 0;
    GSX_HIP(hipMemcpyAsync(&flags, c->devflags.p, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    if (flags) {
        GSX_HIP(hipMemsetAsync(c->devflags.p, 0, 64, c->stream));
        if (flags & 1u) GSX_FAIL("sor: coordinates are not finite (NaN/inf): results were set to NaN");
        GSX_FAIL("device-side error flags 0x%x", flags);
    }
    return 0;
}

int gsx_comm_destroy(gsx_ctx *c);

void gsx_ctx_destroy(gsx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)gsx_comm_destroy(c);
    (void)hipStreamSynchronize(c->stream);
    for (auto &s : c->slots)
        for (auto e : s.ev) (void)hipEventDestroy(e);
    for (auto &w : c->ws) w.release_all();
    c->tree_ws.release_all();
    c->slab_ws.release_all();
    gsx::DevBuf *bufs[] = {&c->commscratch, &c->devflags, &c->vox_ids, &c->nzmask, &c->statspart, &c->scratch, &c->scratch2, &c->scratch3, &c->scratch4, &c->scratch5};
    for (auto b : bufs) b->release();
    if (c->owned_stream) (void)hipStreamDestroy(c->owned_stream);
    delete c;
}

int gsx_ctx_set_stream(gsx_ctx *c, void *s)
{
    if (!c) GSX_FAIL("null ctx");
    c->stream = reinterpret_cast<hipStream_t>(s);
    return 0;
}

int gsx_ctx_last_knn_algo(gsx_ctx *c) { return c ? c->last_knn_algo : -1; }

int gsx_ctx_own_stream(gsx_ctx *c)
{
    if (!c) GSX_FAIL("null ctx");
    GSX_HIP(hipSetDevice(c->device));
    if (!c->owned_stream) GSX_HIP(hipStreamCreateWithFlags(&c->owned_stream, hipStreamNonBlocking));
    c->stream = c->owned_stream;
    return 0;
}

int gsx_ctx_synchronize(gsx_ctx *c)
{
    if (!c) GSX_FAIL("null ctx");
    GSX_HIP(hipSetDevice(c->device));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int gsx_ctx_set_timing(gsx_ctx *c, int enable)
{
    if (!c) GSX_FAIL("null ctx");
    c->timing = enable != 0;
    return 0;
}

int gsx_ctx_reset_timing(gsx_ctx *c)
{
    if (!c) GSX_FAIL("null ctx");
    GSX_HIP(hipStreamSynchronize(c->stream));
    for (auto &s : c->slots) {
        s.used = 0;
        s.launches = 0;
        s.total_ms = 0.0;
    }
    return 0;
}

int gsx_ctx_get_timing(gsx_ctx *c, int slot, uint64_t *launches, double *total_ms)
{
    if (!c || slot < 0 || slot >= GSX_T_SLOTS) GSX_FAIL("gsx_ctx_get_timing: bad arguments");
    GSX_HIP(hipStreamSynchronize(c->stream));
    GSX_CHECK(timing_resolve(c, slot));
    if (launches) *launches = c->slots[slot].launches;
    if (total_ms) *total_ms = c->slots[slot].total_ms;
    return 0;
}

int gsx_ctx_set_param(gsx_ctx *c, const char *name, double value)
{
    if (!c || !name) GSX_FAIL("gsx_ctx_set_param: bad arguments");
    if (!strcmp(name, "grid_points_per_cell")) {
        if (value != 0.0 && !(value >= 1.0 && value <= 512.0)) GSX_FAIL("grid_points_per_cell must be 0 (auto) or in [1,512]");
        c->grid_points_per_cell = value;
    } else if (!strcmp(name, "brute_below")) {
        c->brute_below = (int64_t)value;
    } else if (!strcmp(name, "debug_skip")) {
#ifdef GSX_ABLATE
        c->debug_skip = (int)value;
#else
        if (value != 0.0) GSX_FAIL("debug_skip needs a profiling build of the library (-DGSX_ABLATE); this one computes exact results only");
#endif
    } else if (!strcmp(name, "kmeans_mfma")) {
        c->kmeans_mfma = value != 0.0;
    } else if (!strcmp(name, "km_exact_blocks")) {
        c->km_exact_blocks = value < 1 ? 1 : (value > 32 ? 32 : (int)value);
    } else if (!strcmp(name, "km_seg_rows")) {
        c->km_seg_rows = value >= 64 ? 64 : (value >= 32 ? 32 : 16);
    } else if (!strcmp(name, "km_group_mb")) {
        c->km_group_mb = value < 0 ? 0 : (int)value;
    } else if (!strcmp(name, "kmeans_cs")) {
        c->kmeans_cs = value != 0.0;
    } else if (!strcmp(name, "km_small_wgs")) {
        c->km_small_wgs = (int)value;
    } else if (!strcmp(name, "kmeans_cs_small")) {
        c->kmeans_cs_small = value != 0.0;
    } else if (!strcmp(name, "ring_fast")) {
        c->ring_fast = value != 0.0;
    } else if (!strcmp(name, "phase2_net")) {
        c->phase2_net = value != 0.0;
    } else if (!strcmp(name, "timing_mask")) {
        c->timing_mask = (unsigned)value;
    } else if (!strcmp(name, "adaptive")) {
        c->adaptive = (int)value;
    } else if (!strcmp(name, "tree")) {
        c->tree = (int)value;
    } else if (!strcmp(name, "tree_leaf_cap")) {
        c->tree_leaf_cap = (int)value;
    } else if (!strcmp(name, "tree_cand_limit")) {
        c->tree_cand_limit = (int)value;
    } else if (!strcmp(name, "tree_near_cell")) {
        c->tree_near_cell = value > 0.05 ? value : 0.5;
    } else if (!strcmp(name, "tree_scale")) {
        c->tree_scale = value;
    } else if (!strcmp(name, "defer_words")) {
        c->defer_words = (int)value;
    } else if (!strcmp(name, "filter_mfma")) {
        c->filter_mfma = value != 0.0;
    } else {
        GSX_FAIL("gsx_ctx_set_param: unknown parameter '%s'", name);
    }
    return 0;
}

int gsx_dev_malloc(gsx_ctx *c, size_t bytes, void **dptr)
{
    if (!c || !dptr) GSX_FAIL("gsx_dev_malloc: bad arguments");
    GSX_HIP(hipSetDevice(c->device));
    GSX_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return 0;
}

// page-locked host memory the DMA engines read at link rate (a staging buffer a host routine fills, e.g. gsx_host_gather_f32 writing
// the (n, 3) coordinates of a table straight into it: no page faults of a fresh allocation, no munmap afterwards)
int gsx_host_pinned_alloc(gsx_ctx *c, size_t bytes, void **hptr)
{
    if (!c || !hptr) GSX_FAIL("gsx_host_pinned_alloc: bad arguments");
    GSX_HIP(hipSetDevice(c->device));
    GSX_HIP(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return 0;
}

int gsx_host_pinned_free(gsx_ctx *c, void *hptr)
{
    if (!c) GSX_FAIL("null ctx");
    if (hptr) GSX_HIP(hipHostFree(hptr));
    return 0;
}

int gsx_dev_free(gsx_ctx *c, void *dptr)
{
    if (!c) GSX_FAIL("null ctx");
    if (dptr) GSX_HIP(hipFree(dptr));
    return 0;
}

int gsx_dev_upload(gsx_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c) GSX_FAIL("null ctx");
    GSX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int gsx_dev_download(gsx_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c) GSX_FAIL("null ctx");
    GSX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------ SOR
int gsx_sor_knn_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n_ref,
                    int64_t q_begin, int64_t q_count, int k, int algo, float *mean_out, gsx_sor_info *info)
{
    if (!c || !x || !y || !z || !mean_out) GSX_FAIL("gsx_sor_knn_dev: null argument");
    if (n_ref <= 0 || n_ref >= (1LL << 31) - 1024) GSX_FAIL("gsx_sor_knn_dev: n_ref=%lld out of range", (long long)n_ref);
    if (q_begin < 0 || q_count < 0 || q_begin + q_count > n_ref) GSX_FAIL("gsx_sor_knn_dev: query range out of bounds");
    if (k < 1 || k > 2047) GSX_FAIL("gsx_sor_knn_dev: k=%d not supported (1 <= k <= 2047)", k);
    if (stride < 1) GSX_FAIL("gsx_sor_knn_dev: bad stride");
    GSX_HIP(hipSetDevice(c->device));
    if (q_count == 0) return 0;
    if (algo == GSX_KNN_AUTO) algo = n_ref < c->brute_below ? GSX_KNN_BRUTE : GSX_KNN_GRID;
    if (k > 64) algo = GSX_KNN_GRID;   // the list-free any-k kernel lives on the grid path (csrc/sor_grid.hip: knn_anyk_kernel)
    if (algo == GSX_KNN_BRUTE) {
        GSX_CHECK(c->ws[0].packed.reserve(sizeof(float4) * (size_t)n_ref));
        GSX_CHECK(timing_begin(c, GSX_T_SOR_BIN));
        GSX_CHECK(launch_pack_points(c, x, y, z, stride, n_ref, c->ws[0].packed.as<float4>()));
        GSX_CHECK(timing_end(c, GSX_T_SOR_BIN));
        GSX_CHECK(timing_begin(c, GSX_T_SOR_KNN));
        GSX_CHECK(launch_knn_brute(c, c->ws[0].packed.as<float4>(), n_ref, q_begin, q_count, nullptr, nullptr, 0, k, mean_out));
        GSX_CHECK(timing_end(c, GSX_T_SOR_KNN));
        if (info) {
            memset(info, 0, sizeof(*info));
            info->algo = GSX_KNN_BRUTE;
            GSX_HIP(hipStreamSynchronize(c->stream));
        }
        return 0;
    }
    // (adaptive mode: launch_knn_grid itself hands clouds its grid cannot resolve to the tree path, csrc/sor_tree.hip)
    if (algo == GSX_KNN_GRID) return launch_knn_grid(c, x, y, z, stride, n_ref, q_begin, q_count, k, mean_out, info);
    if (algo == GSX_KNN_TREE) return launch_knn_tree(c, x, y, z, stride, n_ref, q_begin, q_count, k, mean_out, nullptr, info, INT32_MAX, 0, 1, false);
    GSX_FAIL("gsx_sor_knn_dev: unknown algo %d", algo);
}

int gsx_sor_knn_share_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                          int k, int algo, int share, int nshares, float *mean_out, gsx_sor_info *info)
{
    if (!c || !x || !y || !z || !mean_out) GSX_FAIL("gsx_sor_knn_share_dev: null argument");
    if (nshares < 1 || share < 0 || share >= nshares) GSX_FAIL("gsx_sor_knn_share_dev: share %d of %d", share, nshares);
    if (n <= 0 || n >= (1LL << 31) - 1024) GSX_FAIL("gsx_sor_knn_share_dev: n=%lld out of range", (long long)n);
    if (k < 1 || k > 64) GSX_FAIL("gsx_sor_knn_share_dev: k=%d not supported (1 <= k <= 64)", k);
    if (stride < 1) GSX_FAIL("gsx_sor_knn_share_dev: bad stride");
    GSX_HIP(hipSetDevice(c->device));
    // every query belongs to exactly one share; the other shares' entries stay +0.0f so that the
    // shares combine by a plain sum (x + 0 = x exactly)
    GSX_HIP(hipMemsetAsync(mean_out, 0, sizeof(float) * (size_t)n, c->stream));
    if (algo == GSX_KNN_AUTO) algo = n < c->brute_below ? GSX_KNN_BRUTE : GSX_KNN_GRID;
    if (algo == GSX_KNN_BRUTE) {  // no spatial structure: the share is an index range
        const int64_t q0 = n * share / nshares, q1 = n * (share + 1) / nshares;
        if (q1 == q0) return 0;
        return gsx_sor_knn_dev(c, x, y, z, stride, n, q0, q1 - q0, k, GSX_KNN_BRUTE, mean_out + q0, info);
    }
    if (algo == GSX_KNN_GRID) return launch_knn_grid(c, x, y, z, stride, n, 0, n, k, mean_out, info, share, nshares);
    if (algo == GSX_KNN_TREE) return launch_knn_tree(c, x, y, z, stride, n, 0, n, k, mean_out, nullptr, info, INT32_MAX, share, nshares, false);
    GSX_FAIL("gsx_sor_knn_share_dev: unknown algo %d", algo);
}

int gsx_sor_stats_dev(gsx_ctx *c, const float *md, int64_t n, double factor, float *stats_dev)
{
    if (!c || !md || !stats_dev) GSX_FAIL("gsx_sor_stats_dev: null argument");
    if (reinterpret_cast<uintptr_t>(md) & 15) GSX_FAIL("gsx_sor_stats_dev: mean_dists must be 16-byte aligned");
    GSX_HIP(hipSetDevice(c->device));
    GSX_CHECK(timing_begin(c, GSX_T_SOR_STATS));
    GSX_CHECK(launch_sor_stats(c, md, n, factor, stats_dev));
    GSX_CHECK(timing_end(c, GSX_T_SOR_STATS));
    return 0;
}

int gsx_sor_mask_dev(gsx_ctx *c, const float *md, int64_t n, const float *thr_dev, uint8_t *mask)
{
    if (!c || !md || !thr_dev || !mask) GSX_FAIL("gsx_sor_mask_dev: null argument");
    GSX_HIP(hipSetDevice(c->device));
    GSX_CHECK(timing_begin(c, GSX_T_SOR_STATS));
    GSX_CHECK(launch_sor_mask(c, md, n, thr_dev, mask));
    GSX_CHECK(timing_end(c, GSX_T_SOR_STATS));
    return 0;
}

// one cached context per process for the host-buffer entry points (callers are single-threaded,
// SURVEY.md 8(b)); guarded anyway
static std::mutex g_host_mu;
static gsx_ctx *g_host_ctx = nullptr;

static int host_ctx(gsx_ctx **out)
{
    if (!g_host_ctx) {
        GSX_CHECK(gsx_ctx_create(0, &g_host_ctx));
        g_host_ctx->adaptive = 1;  // the host entry points are synchronous anyway
    }
    GSX_HIP(hipSetDevice(g_host_ctx->device));
    *out = g_host_ctx;
    return 0;
}

// stage strided host xyz as three device columns (SoA in HBM)
static int upload_xyz(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                      float **dx, float **dy, float **dz, int64_t *dstride)
{
    if (stride == 1) {
        GSX_CHECK(c->scratch.reserve(sizeof(float) * 3 * (size_t)n));
        float *base = c->scratch.as<float>();
        GSX_HIP(hipMemcpyAsync(base, x, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
        GSX_HIP(hipMemcpyAsync(base + n, y, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
        GSX_HIP(hipMemcpyAsync(base + 2 * n, z, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
        *dx = base; *dy = base + n; *dz = base + 2 * n; *dstride = 1;
        return 0;
    }
    if (stride == 3 && y == x + 1 && z == x + 2) {  // the reference's (N,3) coords: one contiguous copy
        GSX_CHECK(c->scratch.reserve(sizeof(float) * 3 * (size_t)n));
        float *base = c->scratch.as<float>();
        GSX_HIP(hipMemcpyAsync(base, x, sizeof(float) * 3 * n, hipMemcpyHostToDevice, c->stream));
        *dx = base; *dy = base + 1; *dz = base + 2; *dstride = 3;
        return 0;
    }
    GSX_FAIL("xyz layout not supported by the host entry points (use three contiguous columns or (N,3) rows)");
}

int gsx_sor_filter(const float *x, const float *y, const float *z, int64_t stride, int64_t n, int k, double factor,
                   int algo, uint8_t *mask_out, float *mean_out, float *stats_out, gsx_sor_info *info)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!x || !y || !z || !mask_out) GSX_FAIL("gsx_sor_filter: null argument");
    if (n <= 0) GSX_FAIL("gsx_sor_filter: empty cloud");
    gsx_ctx *c;
    GSX_CHECK(host_ctx(&c));
    float *dx, *dy, *dz;
    int64_t ds;
    GSX_CHECK(upload_xyz(c, x, y, z, stride, n, &dx, &dy, &dz, &ds));
    GSX_CHECK(c->scratch2.reserve(sizeof(float) * (size_t)n));
    GSX_CHECK(c->scratch3.reserve(sizeof(float) * 4));
    GSX_CHECK(c->scratch4.reserve((size_t)n + 4));
    float *dmd = c->scratch2.as<float>();
    float *dstats = c->scratch3.as<float>();
    uint8_t *dmask = c->scratch4.as<uint8_t>();
    GSX_CHECK(gsx_sor_knn_dev(c, dx, dy, dz, ds, n, 0, n, k, algo, dmd, nullptr));
    GSX_CHECK(gsx_sor_stats_dev(c, dmd, n, factor, dstats));
    GSX_CHECK(gsx_sor_mask_dev(c, dmd, n, dstats + 2, dmask));
    GSX_HIP(hipMemcpyAsync(mask_out, dmask, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (mean_out) GSX_HIP(hipMemcpyAsync(mean_out, dmd, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (stats_out) GSX_HIP(hipMemcpyAsync(stats_out, dstats, sizeof(float) * 3, hipMemcpyDeviceToHost, c->stream));
    GSX_CHECK(gsx_ctx_check(c));  // synchronises; non-finite coordinates (either algorithm) are an error, like cKDTree's
    int used = k > 64 ? GSX_KNN_GRID : (algo == GSX_KNN_AUTO ? (n < c->brute_below ? GSX_KNN_BRUTE : GSX_KNN_GRID) : algo);
    if (used == GSX_KNN_GRID && c->last_knn_algo == GSX_KNN_TREE) used = GSX_KNN_TREE;   // adaptive mode chose the tree path
    if (info) {
        // re-query diagnostics without recomputing: only the grid path has device-side counters
        memset(info, 0, sizeof(*info));
        info->algo = used;
        if (used == GSX_KNN_GRID) {
            GridParams hgp;
            GSX_HIP(hipMemcpy(&hgp, c->ws[0].gridparams.p, sizeof(hgp), hipMemcpyDeviceToHost));
            info->grid_dim[0] = hgp.nx; info->grid_dim[1] = hgp.ny; info->grid_dim[2] = hgp.nz;
            info->cell_size = hgp.h;
            info->n_cells = hgp.ncells;
            info->n_bricks = hgp.nbricks;
            info->n_fallback = hgp.fail_count;
            info->n_exhaustive = hgp.exhaustive_count;
            info->n_deferred_bricks = hgp.deferred_count;
            info->n_refined = (int64_t)c->ws[0].refined_total;
        }
        if (used == GSX_KNN_TREE) GSX_CHECK(knn_tree_info(c, info));
    }
    return 0;
}

// ------------------------------------------------------------------ density
int gsx_density_voxels_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                           double voxel_size, int64_t min_points, int64_t dense_cap, int64_t *n_unique_out,
                           int64_t *n_dense_out, int64_t *dense_keys_out, int64_t *dense_counts_out)
{
    if (!c || !x || !y || !z || !n_unique_out || !n_dense_out || !dense_keys_out || !dense_counts_out)
        GSX_FAIL("gsx_density_voxels_dev: null argument");
    if (n <= 0) GSX_FAIL("gsx_density_voxels_dev: empty cloud");
    GSX_HIP(hipSetDevice(c->device));
    return density_voxels_dev(c, x, y, z, stride, n, voxel_size, min_points, dense_cap, n_unique_out, n_dense_out,
                              dense_keys_out, dense_counts_out);
}

int gsx_density_hist_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n, double voxel_size,
                         int64_t cap, int64_t *n_unique_out, int64_t *keys3_dev, int64_t *counts_dev)
{
    if (!c || !x || !y || !z || !n_unique_out || !keys3_dev || !counts_dev) GSX_FAIL("gsx_density_hist_dev: null argument");
    if (n <= 0) GSX_FAIL("gsx_density_hist_dev: empty cloud");
    GSX_HIP(hipSetDevice(c->device));
    return density_hist_dev(c, x, y, z, stride, n, voxel_size, cap, n_unique_out, keys3_dev, counts_dev);
}

int gsx_density_merge_dev(gsx_ctx *c, const int64_t *keys3_dev, const int64_t *counts_dev, int64_t m, int64_t min_points,
                          int64_t dense_cap, int64_t *n_unique_out, int64_t *n_dense_out, int64_t *dense_keys_out,
                          int64_t *dense_counts_out)
{
    if (!c || !n_unique_out || !n_dense_out || !dense_keys_out || !dense_counts_out || (m > 0 && (!keys3_dev || !counts_dev)))
        GSX_FAIL("gsx_density_merge_dev: null argument");
    GSX_HIP(hipSetDevice(c->device));
    return density_merge_dev(c, keys3_dev, counts_dev, m, min_points, dense_cap, n_unique_out, n_dense_out, dense_keys_out,
                             dense_counts_out);
}

int gsx_density_filter_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n, double voxel_size,
                           int64_t min_points, int keep_multicluster, const float *box6, uint8_t *mask_out_dev, gsx_density_info *info)
{
    if (!c || !x || !y || !z || !mask_out_dev || !info) GSX_FAIL("gsx_density_filter_dev: null argument");
    if (n <= 0) GSX_FAIL("gsx_density_filter_dev: empty cloud");
    GSX_HIP(hipSetDevice(c->device));
    return density_filter_dev(c, x, y, z, stride, n, voxel_size, min_points, keep_multicluster, box6, mask_out_dev, info);
}

int gsx_density_mask_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                         double voxel_size, const int64_t *kept_keys, int64_t n_kept, uint8_t *mask_out_dev)
{
    if (!c || !x || !y || !z || !mask_out_dev || (n_kept > 0 && !kept_keys)) GSX_FAIL("gsx_density_mask_dev: null argument");
    if (n <= 0) return 0;
    GSX_HIP(hipSetDevice(c->device));
    return density_mask_dev(c, x, y, z, stride, n, voxel_size, kept_keys, n_kept, mask_out_dev);
}

int gsx_density_voxels(const float *x, const float *y, const float *z, int64_t stride, int64_t n, double voxel_size,
                       int64_t min_points, int64_t dense_cap, int64_t *n_unique_out, int64_t *n_dense_out,
                       int64_t *dense_keys_out, int64_t *dense_counts_out)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!x || !y || !z) GSX_FAIL("gsx_density_voxels: null argument");
    if (n <= 0) GSX_FAIL("gsx_density_voxels: empty cloud");
    gsx_ctx *c;
    GSX_CHECK(host_ctx(&c));
    float *dx, *dy, *dz;
    int64_t ds;
    GSX_CHECK(upload_xyz(c, x, y, z, stride, n, &dx, &dy, &dz, &ds));
    return gsx_density_voxels_dev(c, dx, dy, dz, ds, n, voxel_size, min_points, dense_cap, n_unique_out, n_dense_out,
                                  dense_keys_out, dense_counts_out);
}

int gsx_density_mask(const float *x, const float *y, const float *z, int64_t stride, int64_t n, double voxel_size,
                     const int64_t *kept_keys, int64_t n_kept, uint8_t *mask_out)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!x || !y || !z || !mask_out) GSX_FAIL("gsx_density_mask: null argument");
    if (n <= 0) return 0;
    gsx_ctx *c;
    GSX_CHECK(host_ctx(&c));
    float *dx, *dy, *dz;
    int64_t ds;
    GSX_CHECK(upload_xyz(c, x, y, z, stride, n, &dx, &dy, &dz, &ds));
    GSX_CHECK(c->scratch4.reserve((size_t)n + 4));
    GSX_CHECK(gsx_density_mask_dev(c, dx, dy, dz, ds, n, voxel_size, kept_keys, n_kept, c->scratch4.as<uint8_t>()));
    GSX_HIP(hipMemcpyAsync(mask_out, c->scratch4.p, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------ K-Means
int gsx_kmeans_lloyd_dev(gsx_ctx *c, const float *data_dev, int64_t n, int d, int k, int max_iter,
                         float *centroids_dev, int32_t *labels_dev)
{
    if (!c || !data_dev || !centroids_dev || !labels_dev) GSX_FAIL("gsx_kmeans_lloyd_dev: null argument");
    if (max_iter < 0) GSX_FAIL("gsx_kmeans_lloyd_dev: negative max_iter");
    GSX_HIP(hipSetDevice(c->device));
    return kmeans_lloyd_dev(c, data_dev, n, d, k, max_iter, centroids_dev, labels_dev);
}

int gsx_kmeans_lloyd_batch_dev(gsx_ctx *c, const float *data_dev, const int64_t *row_off, int nprob, int d, int k, int max_iter,
                               float *centroids_dev, int32_t *labels_dev)
{
    if (!c || !data_dev || !row_off || !centroids_dev || !labels_dev) GSX_FAIL("gsx_kmeans_lloyd_batch_dev: null argument");
    if (max_iter < 0) GSX_FAIL("gsx_kmeans_lloyd_batch_dev: negative max_iter");
    GSX_HIP(hipSetDevice(c->device));
    return kmeans_lloyd_batch_dev(c, data_dev, row_off, nprob, d, k, max_iter, centroids_dev, labels_dev);
}

int gsx_kmeans_lloyd(const float *data, int64_t n, int d, int k, int max_iter, const float *init_centroids,
                     float *centroids_out, int32_t *labels_out)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!data || !init_centroids || !centroids_out || !labels_out) GSX_FAIL("gsx_kmeans_lloyd: null argument");
    if (n <= 0 || d <= 0 || k <= 0) GSX_FAIL("gsx_kmeans_lloyd: bad shape");
    gsx_ctx *c;
    GSX_CHECK(host_ctx(&c));
    const size_t nd = (size_t)n * d, kd = (size_t)k * d;
    GSX_CHECK(c->scratch.reserve(sizeof(float) * nd));
    GSX_CHECK(c->scratch2.reserve(sizeof(float) * kd));
    GSX_CHECK(c->scratch4.reserve(sizeof(int32_t) * (size_t)n));
    GSX_HIP(hipMemcpyAsync(c->scratch.p, data, sizeof(float) * nd, hipMemcpyHostToDevice, c->stream));
    GSX_HIP(hipMemcpyAsync(c->scratch2.p, init_centroids, sizeof(float) * kd, hipMemcpyHostToDevice, c->stream));
    GSX_HIP(hipMemsetAsync(c->scratch4.p, 0, sizeof(int32_t) * (size_t)n, c->stream));  // max_iter == 0: labels stay 0 (gpu_ops.py:183)
    GSX_CHECK(gsx_kmeans_lloyd_dev(c, c->scratch.as<float>(), n, d, k, max_iter, c->scratch2.as<float>(),
                                   c->scratch4.as<int32_t>()));
    GSX_HIP(hipMemcpyAsync(centroids_out, c->scratch2.p, sizeof(float) * kd, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipMemcpyAsync(labels_out, c->scratch4.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int gsx_kmeans_pp(const float *data, int64_t n, int d, int k, const double *uniforms, int n_local_trials, float *centroids_out)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!data || !uniforms || !centroids_out) GSX_FAIL("gsx_kmeans_pp: null argument");
    if (n <= 0 || d <= 0 || k <= 0) GSX_FAIL("gsx_kmeans_pp: bad shape");
    gsx_ctx *c;
    GSX_CHECK(host_ctx(&c));
    const size_t nd = (size_t)n * d, kd = (size_t)k * d;
    GSX_CHECK(c->scratch.reserve(sizeof(float) * nd));
    GSX_CHECK(c->scratch2.reserve(sizeof(float) * kd));
    GSX_HIP(hipMemcpyAsync(c->scratch.p, data, sizeof(float) * nd, hipMemcpyHostToDevice, c->stream));
    GSX_CHECK(gsx_kmeans_pp_dev(c, c->scratch.as<float>(), n, d, k, uniforms, n_local_trials, c->scratch2.as<float>()));
    GSX_HIP(hipMemcpyAsync(centroids_out, c->scratch2.p, sizeof(float) * kd, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int gsx_kmeans1d(const float *vals, int64_t n, int k, int iters, float *centroids_out, int32_t *labels_out, double *inertia3_out)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!vals || !centroids_out) GSX_FAIL("gsx_kmeans1d: null argument");
    if (n <= 0 || k <= 0) GSX_FAIL("gsx_kmeans1d: bad shape");
    gsx_ctx *c;
    GSX_CHECK(host_ctx(&c));
    GSX_CHECK(c->scratch.reserve(sizeof(float) * (size_t)n));
    GSX_CHECK(c->scratch2.reserve(sizeof(float) * (size_t)k));
    if (labels_out) GSX_CHECK(c->scratch4.reserve(sizeof(int32_t) * (size_t)n));
    GSX_HIP(hipMemcpyAsync(c->scratch.p, vals, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    GSX_CHECK(gsx_kmeans1d_dev(c, c->scratch.as<float>(), n, k, iters, 3, c->scratch2.as<float>(),
                               labels_out ? c->scratch4.as<int32_t>() : nullptr, inertia3_out));
    GSX_HIP(hipMemcpyAsync(centroids_out, c->scratch2.p, sizeof(float) * (size_t)k, hipMemcpyDeviceToHost, c->stream));
    if (labels_out) GSX_HIP(hipMemcpyAsync(labels_out, c->scratch4.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int gsx_quantize_sorted_codebook_dev(gsx_ctx *c, const float *vals_dev, int64_t n, const float *codebook_dev, int kcb,
                                     uint8_t *idx_out_dev)
{
    if (!c || !vals_dev || !codebook_dev || !idx_out_dev) GSX_FAIL("gsx_quantize_sorted_codebook_dev: null argument");
    GSX_HIP(hipSetDevice(c->device));
    return quantize_dev(c, vals_dev, n, codebook_dev, kcb, idx_out_dev);
}

int gsx_quantize_sorted_codebook(const float *vals, int64_t n, const float *codebook, int kcb, uint8_t *idx_out)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!vals || !codebook || !idx_out) GSX_FAIL("gsx_quantize_sorted_codebook: null argument");
    if (n <= 0) return 0;
    gsx_ctx *c;
    GSX_CHECK(host_ctx(&c));
    GSX_CHECK(c->scratch.reserve(sizeof(float) * (size_t)n));
    GSX_CHECK(c->scratch2.reserve(sizeof(float) * 256));
    GSX_CHECK(c->scratch4.reserve((size_t)n + 4));
    GSX_HIP(hipMemcpyAsync(c->scratch.p, vals, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    GSX_HIP(hipMemcpyAsync(c->scratch2.p, codebook, sizeof(float) * (size_t)(kcb > 0 && kcb <= 256 ? kcb : 0),
                           hipMemcpyHostToDevice, c->stream));
    GSX_CHECK(gsx_quantize_sorted_codebook_dev(c, c->scratch.as<float>(), n, c->scratch2.as<float>(), kcb,
                                               c->scratch4.as<uint8_t>()));
    GSX_HIP(hipMemcpyAsync(idx_out, c->scratch4.p, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------ SOG writer numeric core (host buffers)
int gsx_lexsort3_dev(gsx_ctx *c, const float *k0, const float *k1, const float *k2, int64_t stride, int64_t n, uint32_t *perm_out_dev);
int gsx_sog_quats_dev(gsx_ctx *c, const float *rot_rows_dev, int64_t n, uint8_t *out4_dev);

int gsx_lexsort3(const float *k0, const float *k1, const float *k2, int64_t n, uint32_t *perm_out)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!k0 || !k1 || !k2 || !perm_out) GSX_FAIL("gsx_lexsort3: null argument");
    if (n <= 0) return 0;
    gsx_ctx *c;
    GSX_CHECK(host_ctx(&c));
    GSX_CHECK(c->scratch.reserve(sizeof(float) * 3 * (size_t)n));
    GSX_CHECK(c->scratch4.reserve(sizeof(uint32_t) * (size_t)n));
    float *base = c->scratch.as<float>();
    const float *src[3] = {k0, k1, k2};
    for (int a = 0; a < 3; ++a) GSX_HIP(hipMemcpyAsync(base + (size_t)a * n, src[a], sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    GSX_CHECK(gsx_lexsort3_dev(c, base, base + n, base + 2 * (size_t)n, 1, n, c->scratch4.as<uint32_t>()));
    GSX_HIP(hipMemcpyAsync(perm_out, c->scratch4.p, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int gsx_rgb_from_sh(const float *f_dc, int64_t n, uint8_t *out, uint8_t *uncertain_out)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!f_dc || !out || !uncertain_out) GSX_FAIL("gsx_rgb_from_sh: null argument");
    if (n <= 0) return 0;
    gsx_ctx *c;
    GSX_CHECK(host_ctx(&c));
    GSX_CHECK(c->scratch.reserve(sizeof(float) * (size_t)n));
    GSX_CHECK(c->scratch4.reserve(2 * (size_t)n + 16));
    uint8_t *d_out = c->scratch4.as<uint8_t>(), *d_unc = d_out + (size_t)n;
    GSX_HIP(hipMemcpyAsync(c->scratch.p, f_dc, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    GSX_CHECK(gsx_rgb_from_sh_dev(c, c->scratch.as<float>(), n, d_out, d_unc));
    GSX_HIP(hipMemcpyAsync(out, d_out, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipMemcpyAsync(uncertain_out, d_unc, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int gsx_rgb_from_sh_list(const float *f_dc, int64_t n, uint8_t *out, uint32_t *list_out, int64_t cap, int64_t *count_out)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!f_dc || !out || !count_out || (cap > 0 && !list_out)) GSX_FAIL("gsx_rgb_from_sh_list: null argument");
    *count_out = 0;
    if (n <= 0) return 0;
    gsx_ctx *c;
    GSX_CHECK(host_ctx(&c));
    GSX_CHECK(c->scratch.reserve(sizeof(float) * (size_t)n));
    GSX_CHECK(c->scratch4.reserve((size_t)n + 16 + 16 + 4 * (size_t)cap));
    uint8_t *d_out = c->scratch4.as<uint8_t>();
    uint32_t *d_cnt = reinterpret_cast<uint32_t *>(c->scratch4.as<char>() + (((size_t)n + 15) & ~(size_t)15));
    uint32_t *d_list = d_cnt + 4;
    GSX_HIP(hipMemcpyAsync(c->scratch.p, f_dc, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    GSX_CHECK(gsx_rgb_from_sh_list_dev(c, c->scratch.as<float>(), n, d_out, d_list, cap, d_cnt));
    uint32_t hc = 0;
    GSX_HIP(hipMemcpyAsync(out, d_out, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipMemcpyAsync(&hc, d_cnt, 4, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    *count_out = hc;
    const size_t m = std::min<size_t>(hc, (size_t)cap);
    if (m) {
        GSX_HIP(hipMemcpyAsync(list_out, d_list, 4 * m, hipMemcpyDeviceToHost, c->stream));
        GSX_HIP(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int gsx_sog_positions(const float *v, int64_t n, float log_min, float log_max, uint16_t *out, uint8_t *uncertain_out)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!v || !out || !uncertain_out) GSX_FAIL("gsx_sog_positions: null argument");
    if (n <= 0) return 0;
    gsx_ctx *c;
    GSX_CHECK(host_ctx(&c));
    GSX_CHECK(c->scratch.reserve(sizeof(float) * (size_t)n));
    GSX_CHECK(c->scratch4.reserve(3 * (size_t)n + 16));
    uint16_t *d_out = c->scratch4.as<uint16_t>();
    uint8_t *d_unc = c->scratch4.as<uint8_t>() + 2 * (size_t)n;
    GSX_HIP(hipMemcpyAsync(c->scratch.p, v, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    GSX_CHECK(gsx_sog_positions_dev(c, c->scratch.as<float>(), n, log_min, log_max, d_out, d_unc));
    GSX_HIP(hipMemcpyAsync(out, d_out, 2 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipMemcpyAsync(uncertain_out, d_unc, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int gsx_sog_alpha(const float *opacity, int64_t n, uint8_t *out, uint8_t *uncertain_out)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!opacity || !out || !uncertain_out) GSX_FAIL("gsx_sog_alpha: null argument");
    if (n <= 0) return 0;
    gsx_ctx *c;
    GSX_CHECK(host_ctx(&c));
    GSX_CHECK(c->scratch.reserve(sizeof(float) * (size_t)n));
    GSX_CHECK(c->scratch4.reserve(2 * (size_t)n + 16));
    uint8_t *d_out = c->scratch4.as<uint8_t>(), *d_unc = d_out + (size_t)n;
    GSX_HIP(hipMemcpyAsync(c->scratch.p, opacity, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    GSX_CHECK(gsx_sog_alpha_dev(c, c->scratch.as<float>(), n, d_out, d_unc));
    GSX_HIP(hipMemcpyAsync(out, d_out, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipMemcpyAsync(uncertain_out, d_unc, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int gsx_sog_quats(const float *rot_rows, int64_t n, uint8_t *out4)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (!rot_rows || !out4) GSX_FAIL("gsx_sog_quats: null argument");
    if (n <= 0) return 0;
    gsx_ctx *c;
    GSX_CHECK(host_ctx(&c));
    GSX_CHECK(c->scratch.reserve(sizeof(float) * 4 * (size_t)n));
    GSX_CHECK(c->scratch4.reserve(4 * (size_t)n));
    GSX_HIP(hipMemcpyAsync(c->scratch.p, rot_rows, sizeof(float) * 4 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    GSX_CHECK(gsx_sog_quats_dev(c, c->scratch.as<float>(), n, c->scratch4.as<uint8_t>()));
    GSX_HIP(hipMemcpyAsync(out4, c->scratch4.p, 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
