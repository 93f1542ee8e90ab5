// dist_slab.hip -- multi-GPU SOR: RCCL wrappers + the device-side pieces of the slab exchange.
//
// The reference is a single process; its SOR treats queries as independent
// units over one reference set (data_processor.py:160-173).  Scale-out on one MI355X node
// (one process per GPU, splats sharded BY INDEX as a loader hands them out):
//
//   1. global bounding box            all-reduce(max) of 6 floats
//   2. slab planes along one axis     all-reduce(sum) of a 4096-bin histogram -> equal-count slabs
//   3. partition + all-to-all         every point goes to the rank that owns its slab, and as a
//                                     REFERENCE-ONLY copy to every rank whose slab lies within W of it
//                                     (12 B per point; no rank ever holds or bins the whole cloud)
//   4. exact KNN on (own + halo)      launch_knn_slab: the single-GPU pipeline, halo lanes dead
//   5. certificate                    a query's result is globally exact iff its (k+1)-th neighbour is
//                                     nearer than the slab's open faces pushed out by W
//   6. all-to-all of the mean distances back to the index owners (4 B per point)
//   7. numpy-exact statistics         numpy adds 8192-element pieces sequentially: piece sums are
//                                     computed where the elements live and all-gathered (KBs)
//
// RCCL is loaded with dlopen at gsx_comm_init, so single-GPU users never need it; the calls are
// the plain collectives (ncclAllReduce / ncclAllGather / grouped ncclSend + ncclRecv over xGMI).
// The orchestration (sizes, offsets, fallbacks) is host Python in 3dgsconverter_amd/dist.py.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>

#include "gsx_common.h"

namespace gsx {

// ---------------------------------------------------------------- RCCL through dlopen
typedef struct { char internal[128]; } rcclUniqueId;
typedef void *rcclComm_t;
enum { RCCL_INT8 = 0, RCCL_INT64 = 4, RCCL_FLOAT32 = 7 };  // ncclDataType_t
enum { RCCL_SUM = 0, RCCL_MAX = 2, RCCL_MIN = 3 };         // ncclRedOp_t

struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(rcclUniqueId *) = nullptr;
    int (*CommInitRank)(rcclComm_t *, int, rcclUniqueId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, rcclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};
static Rccl g_rccl;

static int rccl_load()
{
    if (g_rccl.h) return 0;
    // an already loaded librccl (torch bundles one under the same SONAME family) is reused by dlopen
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) GSX_FAIL("gsx_comm: cannot load librccl (%s)", dlerror());
#define GSX_SYM(field, name)                                                          \
    *reinterpret_cast<void **>(&g_rccl.field) = dlsym(h, name);                       \
    if (!g_rccl.field) GSX_FAIL("gsx_comm: librccl has no symbol %s", name)
    GSX_SYM(GetUniqueId, "ncclGetUniqueId");
    GSX_SYM(CommInitRank, "ncclCommInitRank");
    GSX_SYM(CommDestroy, "ncclCommDestroy");
    GSX_SYM(GetErrorString, "ncclGetErrorString");
    GSX_SYM(AllReduce, "ncclAllReduce");
    GSX_SYM(AllGather, "ncclAllGather");
    GSX_SYM(Send, "ncclSend");
    GSX_SYM(Recv, "ncclRecv");
    GSX_SYM(GroupStart, "ncclGroupStart");
    GSX_SYM(GroupEnd, "ncclGroupEnd");
#undef GSX_SYM
    g_rccl.h = h;
    return 0;
}

#define GSX_RCCL(call)                                                                                  \
    do {                                                                                                \
        int r__ = (call);                                                                               \
        if (r__ != 0) GSX_FAIL("%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r__) : "?"); \
    } while (0)

// ---------------------------------------------------------------- slab kernels
__device__ __forceinline__ void amax_f32(float *addr, float v)  // finite v; *addr starts at -inf
{
    if (v >= 0.0f) atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned *>(addr), __float_as_uint(v));
}

// out6 = max over points of (-x, -y, -z, x, y, z): ONE max all-reduce yields the global box.  out6[6] = 1 if any
// coordinate is not finite.
__global__ __launch_bounds__(256) void slab_bbox_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ z, int64_t stride, int64_t n,
                                                        float *__restrict__ out7)
{
    __shared__ float red[4][7];
    float m[7];
#pragma unroll
    for (int a = 0; a < 7; ++a) m[a] = a < 6 ? -__builtin_inff() : 0.0f;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * step) {   // 12 loads in flight per lane
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * step < n ? i0 + u * step : i0;
            v[u][0] = x[i * stride];
            v[u][1] = y[i * stride];
            v[u][2] = z[i * stride];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                m[a] = fmaxf(m[a], -v[u][a]);
                m[3 + a] = fmaxf(m[3 + a], v[u][a]);
                m[6] = (fabsf(v[u][a]) < __builtin_inff()) ? m[6] : 1.0f;
            }
    }
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m[a] = fmaxf(m[a], __shfl_xor(m[a], off));
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 7; ++a) red[threadIdx.x >> 6][a] = m[a];
    __syncthreads();
    if (threadIdx.x < 7) {
        const int a = threadIdx.x;
        const float v = fmaxf(fmaxf(red[0][a], red[1][a]), fmaxf(red[2][a], red[3][a]));
        if (v > -__builtin_inff()) amax_f32(&out7[a], v);
    }
}

constexpr int SLAB_BINS = 4096;

__device__ __forceinline__ int slab_bin(float v, float lo, float inv_w)
{
    const int b = (int)((v - lo) * inv_w);
    return min(max(b, 0), SLAB_BINS - 1);
}

// axis / range of the partition coordinate from the (all-reduced) bbox words: the longest edge, first one on ties
__device__ __forceinline__ void slab_axis(const float *__restrict__ b7, int &axis, float &lo, float &hi)
{
    const float e0 = b7[3] + b7[0], e1 = b7[4] + b7[1], e2 = b7[5] + b7[2];  // max - min = max + max(-x)
    axis = 0;
    if (e1 > e0) axis = 1;
    if (e2 > (e1 > e0 ? e1 : e0)) axis = 2;
    lo = -b7[axis];
    hi = b7[3 + axis];
}

// local histogram of the partition coordinate; range and axis are read from the device-resident global bbox, so no
// host round trip separates the bbox all-reduce from this pass
__global__ __launch_bounds__(256) void slab_hist_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ z, int64_t stride, int64_t n,
                                                        const float *__restrict__ b7, unsigned *__restrict__ hist)
{
    __shared__ unsigned h[SLAB_BINS];
    int axis;
    float lo, hi;
    slab_axis(b7, axis, lo, hi);
    const float inv_w = hi > lo ? (float)SLAB_BINS / (hi - lo) : 0.0f;
    const float *__restrict__ c = axis == 0 ? x : (axis == 1 ? y : z);
    for (int i = threadIdx.x; i < SLAB_BINS; i += 256) h[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&h[slab_bin(c[i * stride], lo, inv_w)], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < SLAB_BINS; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

constexpr int SLAB_MAX_RANKS = 16;
struct SlabPlan {
    int world;
    int axis;                        // 0/1/2: the partition axis (the longest edge of the global box)
    float lo, inv_w;                 // histogram binning of that axis
    int cut[SLAB_MAX_RANKS + 1];     // slab s OWNS bins [cut[s], cut[s+1])
    int halo_bins;                   // ... and RECEIVES (reference-only) bins [cut[s] - halo_bins, cut[s+1] + halo_bins):
                                     // membership by bin index, so every count follows from the histograms alone
    unsigned off[2 * SLAB_MAX_RANKS];  // first row of every slot in the send buffer (slot 2s: owned by s, 2s+1: halo copy for s)
};

// Per 2048-point tile (points stay in registers): count the rows per slot (slot 2s = owned by slab s, 2s+1 =
// reference-only copy for slab s) in LDS, reserve the tile's runs with one global atomic per non-empty slot, rank the
// rows inside their runs with a second LDS pass and write them.  A point may be a halo copy for any number of
// slabs (slabs thinner than the halo).
__global__ __launch_bounds__(256) void slab_partition_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                             const float *__restrict__ z, int64_t stride, int64_t n,
                                                             SlabPlan plan, unsigned *__restrict__ cursor /* [2*world], zeroed: rows handed out per slot */,
                                                             float *__restrict__ send /* rows of 3 floats */,
                                                             unsigned *__restrict__ send_src /* local index of each OWN row */)
{
    __shared__ unsigned s_cnt[2 * SLAB_MAX_RANKS];
    __shared__ unsigned s_base[2 * SLAB_MAX_RANKS];
    const int nslot = 2 * plan.world;
    const int64_t tile0 = (int64_t)blockIdx.x * 2048;
    if (tile0 >= n) return;
    if (threadIdx.x < nslot) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    float px[8], py[8], pz[8];
    int bin[8], owner[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int64_t i = tile0 + u * 256 + threadIdx.x;
        owner[u] = -1;
        if (i < n) {
            px[u] = x[i * stride];
            py[u] = y[i * stride];
            pz[u] = z[i * stride];
            bin[u] = slab_bin(plan.axis == 0 ? px[u] : (plan.axis == 1 ? py[u] : pz[u]), plan.lo, plan.inv_w);
            int s = 0;
#pragma unroll 1
            while (s + 1 < plan.world && bin[u] >= plan.cut[s + 1]) ++s;
            owner[u] = s;
            atomicAdd(&s_cnt[2 * s], 1u);
            for (int t = 0; t < plan.world; ++t)
                if (t != s && bin[u] >= plan.cut[t] - plan.halo_bins && bin[u] < plan.cut[t + 1] + plan.halo_bins)
                    atomicAdd(&s_cnt[2 * t + 1], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < nslot) {
        const unsigned c = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = c ? plan.off[threadIdx.x] + atomicAdd(&cursor[threadIdx.x], c) : 0u;
        s_cnt[threadIdx.x] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (owner[u] < 0) continue;
        const int64_t i = tile0 + u * 256 + threadIdx.x;
        {
            const size_t at = (size_t)s_base[2 * owner[u]] + atomicAdd(&s_cnt[2 * owner[u]], 1u);
            send[3 * at + 0] = px[u];
            send[3 * at + 1] = py[u];
            send[3 * at + 2] = pz[u];
            send_src[at] = (unsigned)i;
        }
        for (int t = 0; t < plan.world; ++t)
            if (t != owner[u] && bin[u] >= plan.cut[t] - plan.halo_bins && bin[u] < plan.cut[t + 1] + plan.halo_bins) {
                const size_t at = (size_t)s_base[2 * t + 1] + atomicAdd(&s_cnt[2 * t + 1], 1u);
                send[3 * at + 0] = px[u];
                send[3 * at + 1] = py[u];
                send[3 * at + 2] = pz[u];
            }
    }
}

// returned mean distances arrive in the order the points were sent: out[send_src[p]] = recv[p]
__global__ __launch_bounds__(256) void slab_unpermute_kernel(const float *__restrict__ recv, const unsigned *__restrict__ send_src,
                                                             int64_t n, float *__restrict__ out)
{
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x)
        out[send_src[p]] = recv[p];
}

// number of queries whose (k+1)-th neighbour might lie beyond what this rank holds: kth_d2 > (distance to the nearest
// OPEN face of the slab's halo)^2.  open_lo/open_hi = +-inf where the slab ends the cloud.
__global__ __launch_bounds__(256) void slab_certify_kernel(const float *__restrict__ c, int64_t stride, int64_t n_own,
                                                           const double *__restrict__ kth_d2, float open_lo, float open_hi,
                                                           unsigned *__restrict__ n_uncertain)
{
    unsigned bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_own; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = (double)c[i * stride];
        // margin: the halo membership test was made in f32 on the same coordinates -- exact; 1e-6 relative for the
        // plane arithmetic itself
        const double d = fmin(v - (double)open_lo, (double)open_hi - v) * (1.0 - 1e-6);
        bad += !(kth_d2[i] <= d * d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(n_uncertain, bad);
}

int launch_knn_slab(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride, int64_t n_own,
                    int64_t n_halo, int k, float *mean_out, double *kth_out);
int launch_sor_piece_sums(gsx_ctx *ctx, const float *a, int64_t n, const float *mean_dev, float *piece_out);
int launch_sor_stats_from_pieces(gsx_ctx *ctx, const float *pieces, int64_t npieces, int64_t n_total, int mode, double factor,
                                 float *stats_dev);

}  // namespace gsx

using namespace gsx;

struct gsx_comm {
    rcclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" {

int gsx_comm_unique_id(void *out128)
{
    if (!out128) GSX_FAIL("gsx_comm_unique_id: null argument");
    GSX_CHECK(rccl_load());
    rcclUniqueId id;
    GSX_RCCL(g_rccl.GetUniqueId(&id));
    memcpy(out128, &id, sizeof(id));
    return 0;
}

int gsx_comm_init(gsx_ctx *c, int rank, int world, const void *id128)
{
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) GSX_FAIL("gsx_comm_init: bad arguments");
    if (world > SLAB_MAX_RANKS) GSX_FAIL("gsx_comm_init: at most %d ranks (one node)", SLAB_MAX_RANKS);
    GSX_CHECK(rccl_load());
    GSX_HIP(hipSetDevice(c->device));
    if (c->comm) GSX_FAIL("gsx_comm_init: the context already has a communicator");
    gsx_comm *m = new gsx_comm();
    rcclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    int r = g_rccl.CommInitRank(&m->comm, world, id, rank);
    if (r != 0) {
        delete m;
        GSX_FAIL("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    }
    m->rank = rank;
    m->world = world;
    c->comm = m;
    return 0;
}

int gsx_comm_destroy(gsx_ctx *c)
{
    if (!c || !c->comm) return 0;
    gsx_comm *m = static_cast<gsx_comm *>(c->comm);
    if (m->comm) (void)g_rccl.CommDestroy(m->comm);
    delete m;
    c->comm = nullptr;
    return 0;
}

static int dtype_of(int elem_bytes, int *out)
{
    if (elem_bytes == 1) { *out = RCCL_INT8; return 0; }
    if (elem_bytes == 4) { *out = RCCL_FLOAT32; return 0; }
    if (elem_bytes == 8) { *out = RCCL_INT64; return 0; }
    GSX_FAIL("gsx_comm: element size %d not supported", elem_bytes);
}

/* in place; kind: 0 = f32 max, 1 = f32 sum, 2 = i64 sum */
int gsx_comm_all_reduce(gsx_ctx *c, void *buf_dev, int64_t count, int kind)
{
    if (!c || !c->comm || !buf_dev) GSX_FAIL("gsx_comm_all_reduce: no communicator / null buffer");
    gsx_comm *m = static_cast<gsx_comm *>(c->comm);
    const int dt = kind == 2 ? RCCL_INT64 : RCCL_FLOAT32, op = kind == 0 ? RCCL_MAX : RCCL_SUM;
    GSX_RCCL(g_rccl.AllReduce(buf_dev, buf_dev, (size_t)count, dt, op, m->comm, c->stream));
    return 0;
}

int gsx_comm_all_gather(gsx_ctx *c, const void *send_dev, void *recv_dev, int64_t bytes_per_rank)
{
    if (!c || !c->comm || !send_dev || !recv_dev) GSX_FAIL("gsx_comm_all_gather: no communicator / null buffer");
    gsx_comm *m = static_cast<gsx_comm *>(c->comm);
    GSX_RCCL(g_rccl.AllGather(send_dev, recv_dev, (size_t)bytes_per_rank, RCCL_INT8, m->comm, c->stream));
    return 0;
}

/* offsets and counts in ELEMENTS of elem_bytes, one entry per peer (host arrays); the local block is copied */
int gsx_comm_all_to_all_v(gsx_ctx *c, const void *send_dev, const int64_t *send_off, const int64_t *send_cnt, void *recv_dev,
                          const int64_t *recv_off, const int64_t *recv_cnt, int elem_bytes)
{
    if (!c || !c->comm || !send_off || !send_cnt || !recv_off || !recv_cnt)
        GSX_FAIL("gsx_comm_all_to_all_v: no communicator / null argument");
    gsx_comm *m = static_cast<gsx_comm *>(c->comm);
    int dt;
    GSX_CHECK(dtype_of(elem_bytes == 12 ? 4 : elem_bytes, &dt));
    const size_t mul = elem_bytes == 12 ? 3 : 1;   // rows of 3 floats travel as floats
    const size_t eb = elem_bytes;
    bool remote = false;
    for (int p = 0; p < m->world; ++p) remote |= p != m->rank && (send_cnt[p] > 0 || recv_cnt[p] > 0);
    if (remote) {   // (an empty group still costs RCCL bookkeeping)
        GSX_RCCL(g_rccl.GroupStart());
        for (int p = 0; p < m->world; ++p) {
            if (p == m->rank) continue;
            if (send_cnt[p] > 0)
                GSX_RCCL(g_rccl.Send(static_cast<const char *>(send_dev) + eb * (size_t)send_off[p], (size_t)send_cnt[p] * mul, dt, p,
                                     m->comm, c->stream));
            if (recv_cnt[p] > 0)
                GSX_RCCL(g_rccl.Recv(static_cast<char *>(recv_dev) + eb * (size_t)recv_off[p], (size_t)recv_cnt[p] * mul, dt, p,
                                     m->comm, c->stream));
        }
        GSX_RCCL(g_rccl.GroupEnd());
    }
    const int me = m->rank;
    if (send_cnt[me] != recv_cnt[me]) GSX_FAIL("gsx_comm_all_to_all_v: local block sizes differ");
    if (send_cnt[me] > 0)
        GSX_HIP(hipMemcpyAsync(static_cast<char *>(recv_dev) + eb * (size_t)recv_off[me],
                               static_cast<const char *>(send_dev) + eb * (size_t)send_off[me], eb * (size_t)send_cnt[me],
                               hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

/* ---- slab exchange: device-side pieces (usable without a communicator: world = 1 or an emulated exchange) ---- */
int gsx_slab_bbox_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n, float *out7_dev)
{
    if (!c || !x || !y || !z || !out7_dev || n < 0) GSX_FAIL("gsx_slab_bbox_dev: bad arguments");
    GSX_HIP(hipSetDevice(c->device));
    const float init[7] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, 0.0f};
    GSX_HIP(hipMemcpyAsync(out7_dev, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    if (n > 0) {
        // one result per workgroup goes through 7 same-address atomics: keep the grid at two workgroups per CU
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 2048), (int64_t)c->num_cu * 2));
        hipLaunchKernelGGL(slab_bbox_kernel, dim3(blocks), dim3(256), 0, c->stream, x, y, z, stride, n, out7_dev);
        GSX_HIP(hipGetLastError());
    }
    return 0;
}

int gsx_slab_hist_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n, const float *bbox7_dev,
                      uint32_t *hist4096_dev)
{
    if (!c || !x || !y || !z || !bbox7_dev || !hist4096_dev || n < 0) GSX_FAIL("gsx_slab_hist_dev: bad arguments");
    GSX_HIP(hipSetDevice(c->device));
    GSX_HIP(hipMemsetAsync(hist4096_dev, 0, sizeof(uint32_t) * SLAB_BINS, c->stream));
    if (n == 0) return 0;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 4096), (int64_t)c->num_cu * 4));
    hipLaunchKernelGGL(slab_hist_kernel, dim3(blocks), dim3(256), 0, c->stream, x, y, z, stride, n, bbox7_dev, hist4096_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

/*
 * plan (host): world, axis, [lo, hi] = the binned range (the all-reduced bbox words of that axis), cut[world+1], halo_bins.
 * Rows (3 floats) are written to send_dev from start_off[2*world] (HOST array: first row of every slot; cursor_dev[2*world]
 * is device scratch for the rows handed out so far, zeroed here): slot 2s = rows
 * owned by slab s, slot 2s+1 = reference-only copies for slab s; send_src_dev[row] = local index of every own row.
 * planes_out (host, 2*world floats, nullable): coordinates between which slab s is guaranteed to hold EVERY point.
 */
int gsx_slab_partition_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n, int world,
                           int axis, float lo, float hi, const int32_t *cut, int halo_bins, const uint32_t *start_off,
                           uint32_t *cursor_dev, float *send_dev, uint32_t *send_src_dev, float *planes_out)
{
    if (!c || !x || !y || !z || !cut || !start_off || world < 1 || world > SLAB_MAX_RANKS || axis < 0 || axis > 2 || halo_bins < 0)
        GSX_FAIL("gsx_slab_partition_dev: bad arguments");
    GSX_HIP(hipSetDevice(c->device));
    SlabPlan p;
    p.world = world;
    p.axis = axis;
    p.lo = lo;
    p.inv_w = hi > lo ? (float)SLAB_BINS / (hi - lo) : 0.0f;
    p.halo_bins = halo_bins;
    for (int s = 0; s <= world; ++s) p.cut[s] = cut[s];
    for (int s = 0; s < 2 * world; ++s) p.off[s] = start_off[s];
    if (planes_out) {
        // slab s holds every point whose bin is in [cut[s] - halo_bins, cut[s+1] + halo_bins).  bin(c) is a monotone f32
        // function of c whose steps sit within ~1e-3 of a bin of lo + b * bw: half a bin inside is safely inside.
        const double bw = hi > lo ? ((double)hi - (double)lo) / SLAB_BINS : 0.0;
        for (int s = 0; s < world; ++s) {
            const int b0 = cut[s] - halo_bins, b1 = cut[s + 1] + halo_bins;
            planes_out[2 * s] = (s == 0 || b0 <= 0) ? -INFINITY : (float)((double)lo + ((double)b0 + 0.5) * bw);
            planes_out[2 * s + 1] = (s == world - 1 || b1 >= SLAB_BINS) ? INFINITY : (float)((double)lo + ((double)b1 - 0.5) * bw);
        }
    }
    if (n <= 0) return 0;
    if (!cursor_dev || !send_dev || !send_src_dev) GSX_FAIL("gsx_slab_partition_dev: null buffer");
    GSX_HIP(hipMemsetAsync(cursor_dev, 0, sizeof(uint32_t) * 2 * world, c->stream));
    hipLaunchKernelGGL(slab_partition_kernel, dim3(div_up(n, 2048)), dim3(256), 0, c->stream, x, y, z, stride, n, p, cursor_dev,
                       send_dev, send_src_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

/* enqueue only: hipMemcpyAsync from pageable memory has consumed the host buffer when it returns */
int gsx_dev_upload_async(gsx_ctx *c, void *dst_dev, const void *src_host, size_t bytes)
{
    if (!c) GSX_FAIL("null ctx");
    if (bytes) GSX_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}

int gsx_dev_memset(gsx_ctx *c, void *dst_dev, int value, size_t bytes)
{
    if (!c) GSX_FAIL("null ctx");
    if (bytes) GSX_HIP(hipMemsetAsync(dst_dev, value, bytes, c->stream));
    return 0;
}

int gsx_dev_copy(gsx_ctx *c, void *dst_dev, const void *src_dev, size_t bytes)
{
    if (!c) GSX_FAIL("null ctx");
    if (bytes) GSX_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

/* rows: (n_own + n_halo) x 3 floats, own points first.  mean_out_dev: n_own floats, kth_d2_dev: n_own doubles */
int gsx_sor_knn_slab_dev(gsx_ctx *c, const float *rows_dev, int64_t n_own, int64_t n_halo, int k, float *mean_out_dev,
                         double *kth_d2_dev)
{
    if (!c || !rows_dev || !mean_out_dev || !kth_d2_dev) GSX_FAIL("gsx_sor_knn_slab_dev: null argument");
    if (n_own < 0 || n_halo < 0 || n_own + n_halo >= (1LL << 31) - 1024) GSX_FAIL("gsx_sor_knn_slab_dev: sizes out of range");
    if (k < 1 || k > 64) GSX_FAIL("gsx_sor_knn_slab_dev: k=%d not supported (1 <= k <= 64)", k);
    GSX_HIP(hipSetDevice(c->device));
    if (n_own == 0) return 0;
    return launch_knn_slab(c, rows_dev, rows_dev + 1, rows_dev + 2, 3, n_own, n_halo, k, mean_out_dev, kth_d2_dev);
}

int gsx_slab_certify_dev(gsx_ctx *c, const float *coord, int64_t stride, int64_t n_own, const double *kth_d2_dev, float open_lo,
                         float open_hi, uint32_t *n_uncertain_dev)
{
    if (!c || !coord || !kth_d2_dev || !n_uncertain_dev) GSX_FAIL("gsx_slab_certify_dev: null argument");
    GSX_HIP(hipSetDevice(c->device));
    GSX_HIP(hipMemsetAsync(n_uncertain_dev, 0, 2 * sizeof(uint32_t), c->stream));   // 8 bytes: the count is all-reduced as an int64
    if (n_own <= 0) return 0;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n_own, 1024), (int64_t)c->num_cu * 8));
    hipLaunchKernelGGL(slab_certify_kernel, dim3(blocks), dim3(256), 0, c->stream, coord, stride, n_own, kth_d2_dev, open_lo,
                       open_hi, n_uncertain_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

int gsx_slab_unpermute_dev(gsx_ctx *c, const float *recv_dev, const uint32_t *send_src_dev, int64_t n, float *out_dev)
{
    if (!c || !recv_dev || !send_src_dev || !out_dev) GSX_FAIL("gsx_slab_unpermute_dev: null argument");
    GSX_HIP(hipSetDevice(c->device));
    if (n <= 0) return 0;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 1024), (int64_t)c->num_cu * 8));
    hipLaunchKernelGGL(slab_unpermute_kernel, dim3(blocks), dim3(256), 0, c->stream, recv_dev, send_src_dev, n, out_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

/* numpy's 8192-element pieces of a[0, n): piece_out_dev[p] = pairwise f32 sum of a (mean_dev == NULL) or of (a - *mean_dev)^2 */
int gsx_sor_piece_sums_dev(gsx_ctx *c, const float *a_dev, int64_t n, const float *mean_dev, float *piece_out_dev)
{
    if (!c || !a_dev || !piece_out_dev || n <= 0) GSX_FAIL("gsx_sor_piece_sums_dev: bad arguments");
    if (reinterpret_cast<uintptr_t>(a_dev) & 15) GSX_FAIL("gsx_sor_piece_sums_dev: input must be 16-byte aligned");
    GSX_HIP(hipSetDevice(c->device));
    return launch_sor_piece_sums(c, a_dev, n, mean_dev, piece_out_dev);
}

/* mode 0: stats[0] = mean of the n_total elements whose piece sums are given; mode 1: stats[1] = std, stats[2] = threshold */
int gsx_sor_stats_from_pieces_dev(gsx_ctx *c, const float *pieces_dev, int64_t npieces, int64_t n_total, int mode,
                                  double threshold_factor, float *stats_dev)
{
    if (!c || !pieces_dev || !stats_dev || npieces <= 0 || n_total <= 0) GSX_FAIL("gsx_sor_stats_from_pieces_dev: bad arguments");
    GSX_HIP(hipSetDevice(c->device));
    return launch_sor_stats_from_pieces(c, pieces_dev, npieces, n_total, mode, threshold_factor, stats_dev);
}

}  // extern "C"
