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
#include <algorithm>
#include <cmath>

#include "gsx_common.h"
#include "sor_grid_params.h"

namespace gsx {

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
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 8 * step) {   // 8 loads in flight per lane
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = c[(i0 + u * step < n ? i0 + u * step : i0) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * step < n) atomicAdd(&h[slab_bin(v[u], lo, inv_w)], 1u);
    }
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
                                                             unsigned *__restrict__ send_src /* local index of each OWN row */,
                                                             int self_slot = -1, float *__restrict__ self_rows = nullptr,
                                                             int64_t self_shift = 0)
{
    // self_slot >= 0 (fused step): the rows this rank owns itself skip the send buffer and the wire and land where the
    // exchange would have put them: self_rows[3 * (send position + self_shift)]
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
            float *dst = 2 * owner[u] == self_slot ? self_rows + 3 * ((int64_t)at + self_shift) : send + 3 * at;
            dst[0] = px[u];
            dst[1] = py[u];
            dst[2] = pz[u];
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
                                                             int64_t n, float *__restrict__ out,
                                                             const unsigned *__restrict__ cnt_src = nullptr, unsigned *__restrict__ cnt_dst = nullptr,
                                                             int64_t self_lo = 0, int64_t self_hi = 0,
                                                             const float *__restrict__ self_src = nullptr /* indexed by p too */)
{
    // (fused step: this rank's certificate count rides behind its piece sums in the all-gather that follows)
    if (cnt_dst && blockIdx.x == 0 && threadIdx.x < 2) cnt_dst[threadIdx.x] = cnt_src[threadIdx.x];
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x)
        out[send_src[p]] = (p >= self_lo && p < self_hi) ? self_src[p] : recv[p];   // (own queries: straight from the KNN's output)
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

// ---------------------------------------------------------------- the fused step (gsx_sor_slab_step_dev)
// one launch instead of five memsets / tiny uploads: the box words, the histogram, the partition cursors and the
// certificate counter of a step start from here
__global__ __launch_bounds__(256) void slab_clear_kernel(float *__restrict__ b7, unsigned *__restrict__ hist,
                                                         unsigned *__restrict__ small_words /* cursor[32] | uncertain[2] */)
{
    for (int i = threadIdx.x; i < SLAB_BINS; i += 256) hist[i] = 0;
    if (threadIdx.x < 8) b7[threadIdx.x] = threadIdx.x < 6 ? -__builtin_inff() : 0.0f;
    if (threadIdx.x < 34) small_words[threadIdx.x] = 0;
}

// all-gathered piece sums sit at a stride of max_pieces per rank; numpy's sequential fold wants them back to back
struct PackPlan {
    int world;
    int stride;
    int count[SLAB_MAX_RANKS];
};
__global__ __launch_bounds__(256) void slab_pack_pieces_kernel(const float *__restrict__ all, PackPlan p, float *__restrict__ out,
                                                               unsigned long long *__restrict__ cert_total /* nullable */)
{
    if (cert_total && blockIdx.x == 0 && threadIdx.x == 0) {   // the ranks' certificate counts sit behind their piece sums
        unsigned long long t = 0;
        for (int q = 0; q < p.world; ++q) t += *reinterpret_cast<const unsigned long long *>(all + (size_t)q * p.stride + (p.stride - 2));
        *cert_total = t;
    }
    int o = 0;
    for (int q = 0; q < p.world; ++q) {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < p.count[q]; i += gridDim.x * 256) out[o + i] = all[(size_t)q * p.stride + i];
        o += p.count[q];
    }
}

int launch_sor_mask(gsx_ctx *ctx, const float *md, int64_t n, const float *thr_dev, uint8_t *mask);

// cell population of the KNN grid for n reference points (the rule of knn_grid_level in sor_grid.hip; Python restates
// it as dist_slab.pts_per_cell)
static double slab_pts_per_cell(int k, int64_t n)
{
    double m = std::max(2.0, 0.47 * (double)(k + 1));
    const double fill = n >= 4000000 ? 54.0 : 58.0;
    for (int cells = 8; cells >= 1; cells /= 2)
        if (m * cells > fill && m * cells <= 66.0) m = fill / cells;
    return m;
}

int launch_knn_slab(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride, int64_t n_own,
                    int64_t n_halo, int k, float *mean_out, double *kth_out, const SlabKnn *sk = nullptr);
int launch_sor_piece_sums(gsx_ctx *ctx, const float *a, int64_t n, const float *mean_dev, float *piece_out);
int launch_sor_stats_from_pieces(gsx_ctx *ctx, const float *pieces, int64_t npieces, int64_t n_total, int mode, double factor,
                                 float *stats_dev);

}  // namespace gsx

using namespace gsx;

extern "C" {

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

/* The slab plan of one step from what the step's host synchronisation brings back: words[0..7) = the all-reduced box
 * words (max of -x,-y,-z,x,y,z; non-finite flag), words[8 + 4096 q ...] = rank q's histogram of the longest axis.
 * Pure host arithmetic (no device, no communicator): every rank computes the identical plan from identical words.
 * Python restates it (dist_slab.plan_step) for the CPU tests of the choreography; tests/test_dist_cpu.py pins one
 * against the other. */
int gsx_slab_plan(const uint32_t *words, int world, int rank, int64_t n_local, int k, double halo_cells, gsx_slab_plan_t *out)
{
    if (!words || !out || world < 1 || world > SLAB_MAX_RANKS || rank < 0 || rank >= world || k < 1) GSX_FAIL("gsx_slab_plan: bad arguments");
    memset(out, 0, sizeof(*out));
    const int G = world;
    float hb[7];
    memcpy(hb, words, sizeof(hb));
    const uint32_t *hist = words + 8;
    out->world = G;
    out->rank = rank;
    out->n_local = n_local;
    // cumulative histograms: per rank and global
    std::vector<int64_t> cum((size_t)(G + 1) * (SLAB_BINS + 1), 0);
    int64_t *gc = cum.data() + (size_t)G * (SLAB_BINS + 1);
    for (int q = 0; q < G; ++q) {
        int64_t *c = cum.data() + (size_t)q * (SLAB_BINS + 1);
        for (int b = 0; b < SLAB_BINS; ++b) c[b + 1] = c[b] + (int64_t)hist[(size_t)q * SLAB_BINS + b];
        out->sizes[q] = c[SLAB_BINS];
    }
    for (int b = 0; b <= SLAB_BINS; ++b) {
        int64_t t = 0;
        for (int q = 0; q < G; ++q) t += cum[(size_t)q * (SLAB_BINS + 1) + b];
        gc[b] = t;
    }
    const int64_t n_total = gc[SLAB_BINS];
    out->n_total = n_total;
    if (n_total == 0) { out->status = GSX_SLAB_EMPTY; return 0; }
    bool finite = !(hb[6] > 0.0f);
    for (int a = 0; a < 6; ++a) finite = finite && std::isfinite(hb[a]);
    if (!finite) { out->status = GSX_SLAB_NONFINITE; return 0; }
    const float ext[3] = {hb[3] + hb[0], hb[4] + hb[1], hb[5] + hb[2]};   // float32, like the device (slab_axis)
    int axis = 0;
    if (ext[1] > ext[0]) axis = 1;
    if (ext[2] > std::max(ext[0], ext[1])) axis = 2;
    const float lo = -hb[axis], hi = hb[3 + axis];
    out->axis = axis;
    out->lo = lo;
    out->hi = hi;
    // equal-count cuts: slab s owns bins [cut[s], cut[s+1])
    out->cut[0] = 0;
    for (int s = 1; s < G; ++s) {
        const int64_t target = (n_total * s) / G;
        int b = (int)(std::lower_bound(gc, gc + SLAB_BINS + 1, target) - gc);
        out->cut[s] = std::min(std::max(b, out->cut[s - 1]), SLAB_BINS);
    }
    out->cut[G] = SLAB_BINS;
    if (out->sizes[rank] != n_local) GSX_FAIL("gsx_slab_plan: %lld rows given, %lld binned", (long long)n_local, (long long)out->sizes[rank]);
    if (G > 1) {
        int64_t mn = out->sizes[0];
        for (int q = 1; q < G; ++q) mn = std::min(mn, out->sizes[q]);
        if (mn < 8192) { out->status = GSX_SLAB_SMALL_SHARD; return 0; }
    }
    // halo width: halo_cells KNN cell edges of the global density, in whole bins
    double vol = 1.0;
    int nd = 0;
    for (int a = 0; a < 3; ++a)
        if ((double)ext[a] > 0.0) {
            vol *= (double)ext[a];
            ++nd;
        }
    const double per = nd ? vol * slab_pts_per_cell(k, n_total / G) / (double)n_total : 0.0;
    const double h_est = nd ? std::pow(per, 1.0 / (double)nd) : 0.0;
    const double bw = hi > lo ? ((double)hi - (double)lo) / SLAB_BINS : 0.0;
    const int halo_bins = bw > 0.0 ? (int)std::ceil(halo_cells * h_est / bw) + 1 : SLAB_BINS;
    out->halo_bins = halo_bins;
    // rows every source sends every slab (membership is decided by bin index: no counting pass)
    int64_t halo_sum = 0;
    for (int s = 0; s < G; ++s) {
        const int a = out->cut[s], b = out->cut[s + 1];
        const int ha = std::max(a - halo_bins, 0), hbn = std::min(b + halo_bins, SLAB_BINS);
        for (int q = 0; q < G; ++q) {
            const int64_t *c = cum.data() + (size_t)q * (SLAB_BINS + 1);
            const int64_t own = c[b] - c[a], halo = (c[hbn] - c[ha]) - own;
            halo_sum += halo;
            if (q == rank) { out->own_cnt[s] = own; out->halo_cnt[s] = halo; }
            if (s == rank) { out->in_own[q] = own; out->in_halo[q] = halo; }
        }
    }
    out->halo_total = halo_sum;
    if (G > 1 && (double)halo_sum > 0.75 * (double)(G - 1) * (double)n_total) { out->status = GSX_SLAB_NO_STRUCTURE; return 0; }
    int64_t o = 0, ho = n_local, ro = 0;
    for (int s = 0; s < G; ++s) { out->own_off[s] = o; o += out->own_cnt[s]; }
    for (int s = 0; s < G; ++s) { out->halo_off[s] = ho; ho += out->halo_cnt[s]; }
    out->n_send = ho;
    for (int q = 0; q < G; ++q) { out->r_own_off[q] = ro; ro += out->in_own[q]; }
    out->n_own = ro;
    for (int q = 0; q < G; ++q) { out->r_halo_off[q] = ro; ro += out->in_halo[q]; }
    out->n_halo = ro - out->n_own;
    // slab s holds every point whose bin is in [cut[s] - halo_bins, cut[s+1] + halo_bins).  bin(c) is a monotone f32
    // function of c whose steps sit within ~1e-3 of a bin of lo + b * bw: half a bin inside is safely inside.
    {
        const int b0 = out->cut[rank] - halo_bins, b1 = out->cut[rank + 1] + halo_bins;
        out->plane_lo = (rank == 0 || b0 <= 0) ? -INFINITY : (float)((double)lo + ((double)b0 + 0.5) * bw);
        out->plane_hi = (rank == G - 1 || b1 >= SLAB_BINS) ? INFINITY : (float)((double)lo + ((double)b1 - 0.5) * bw);
    }
    return 0;
}

/* One multi-GPU SOR step of this rank, start to finish (the choreography dist_slab.slab_sor spells out, with the
 * communicator gsx_comm_init gave the context; world = 1 without one): box -> all-reduce -> histogram -> all-gather ->
 * [the step's one host synchronisation: plan] -> partition -> rows to the slab owners (+ halo) -> exact KNN -> certificate
 * -> all-reduce -> means back -> un-permute -> numpy-exact statistics from all-gathered piece sums -> mask.
 * Returns 0 with out->status != 0 when the step declines (decided from gathered data: every rank declines together). */
static int slab_step_body(gsx_ctx *c, const float *rows_dev, int64_t n_local, int k, double threshold_factor, double halo_cells,
                          uint8_t *mask_out_dev, gsx_slab_step_t *out);

int gsx_sor_slab_step_dev(gsx_ctx *c, const float *rows_dev, int64_t n_local, int k, double threshold_factor, double halo_cells,
                          uint8_t *mask_out_dev, gsx_slab_step_t *out)
{
    // A rank that fails BETWEEN collectives (a reservation, the planner's row-count check, a launch) would leave its peers
    // blocked in the next one: every error exit tells them (hostwire: their barriers fail at once; RCCL: ncclCommAbort of
    // this rank's communicator, the peers' watchdog -- launch.comm_watchdog -- ends them).  ADVICE round 4.
    // Argument validation fails identically on every rank BEFORE any collective -- no peer can be left blocked, so it must
    // not cost the communicator: checked here, outside the abort-on-error wrapper.
    if (!c || !out || n_local < 0 || (n_local > 0 && !rows_dev)) GSX_FAIL("gsx_sor_slab_step_dev: bad arguments");
    if (k < 1 || k > 64) GSX_FAIL("gsx_sor_slab_step_dev: k=%d not supported (1 <= k <= 64)", k);
    if (n_local >= (1LL << 31) - 1024) GSX_FAIL("gsx_sor_slab_step_dev: shard too large");
    const int rc = slab_step_body(c, rows_dev, n_local, k, threshold_factor, halo_cells, mask_out_dev, out);
    if (rc != 0 && c->comm) gsx_comm_abort(c);
    return rc;
}

static int slab_step_body(gsx_ctx *c, const float *rows_dev, int64_t n_local, int k, double threshold_factor, double halo_cells,
                          uint8_t *mask_out_dev, gsx_slab_step_t *out)
{
    GSX_HIP(hipSetDevice(c->device));
    memset(out, 0, sizeof(*out));
    int G = 1, r = 0;
    GSX_CHECK(gsx_comm_rank(c, &r, &G));
    const bool wire = gsx_comm_transport(c) != 0;
    SlabWs &w = c->slab_ws;
    const float *x = rows_dev, *y = rows_dev + 1, *z = rows_dev + 2;
    // ---- 1./2. box, histogram (device-resident between them), gathered into plan_in
    const size_t plan_words = 8 + (size_t)SLAB_BINS * G;
    GSX_CHECK(w.plan_in.reserve(4 * plan_words));
    GSX_CHECK(w.hist.reserve(4 * SLAB_BINS));
    GSX_CHECK(w.small.reserve(1024));
    if (w.host_cap < 4 * plan_words) {
        if (w.host) (void)hipHostFree(w.host);
        w.host = nullptr;
        GSX_HIP(hipHostMalloc(&w.host, 4 * plan_words, hipHostMallocDefault));
        w.host_cap = 4 * plan_words;
    }
    float *b7 = w.plan_in.as<float>();
    unsigned *hist_all = w.plan_in.as<unsigned>() + 8;
    unsigned *hist_mine = G > 1 ? w.hist.as<unsigned>() : hist_all;
    unsigned *cursor = w.small.as<unsigned>();
    unsigned *unc = cursor + 32;
    float *stats = reinterpret_cast<float *>(cursor + 36);
    GSX_CHECK(timing_begin(c, GSX_T_SLAB_PREP));
    hipLaunchKernelGGL(slab_clear_kernel, dim3(1), dim3(256), 0, c->stream, b7, hist_mine, cursor);
    if (n_local > 0) {
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n_local, 2048), (int64_t)c->num_cu * 2));
        hipLaunchKernelGGL(slab_bbox_kernel, dim3(blocks), dim3(256), 0, c->stream, x, y, z, (int64_t)3, n_local, b7);
    }
    GSX_HIP(hipGetLastError());
    GSX_CHECK(timing_end(c, GSX_T_SLAB_PREP));
    if (G > 1) {
        GSX_CHECK(timing_begin(c, GSX_T_SLAB_COLL));
        GSX_CHECK(gsx_comm_all_reduce(c, b7, 7, GSX_COMM_F32_MAX));
        GSX_CHECK(timing_end(c, GSX_T_SLAB_COLL));
    }
    if (n_local > 0) {
        GSX_CHECK(timing_begin(c, GSX_T_SLAB_PREP));
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n_local, 4096), (int64_t)c->num_cu * 4));
        hipLaunchKernelGGL(slab_hist_kernel, dim3(blocks), dim3(256), 0, c->stream, x, y, z, (int64_t)3, n_local, b7, hist_mine);
        GSX_HIP(hipGetLastError());
        GSX_CHECK(timing_end(c, GSX_T_SLAB_PREP));
    }
    if (G > 1) {
        GSX_CHECK(timing_begin(c, GSX_T_SLAB_COLL));
        GSX_CHECK(gsx_comm_all_gather(c, hist_mine, hist_all, 4 * SLAB_BINS));
        GSX_CHECK(timing_end(c, GSX_T_SLAB_COLL));
    }
    GSX_HIP(hipMemcpyAsync(w.host, w.plan_in.p, 4 * plan_words, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));                                    // <- the step's host synchronisation
    gsx_slab_plan_t &p = out->plan;
    GSX_CHECK(gsx_slab_plan(static_cast<const uint32_t *>(w.host), G, r, n_local, k, halo_cells, &p));
    out->status = p.status;
    if (p.status != 0) return 0;
    // ---- 3. scatter into the send buffer, rows to the slab owners (own + reference-only halo) in ONE group
    const int64_t n_own = p.n_own, n_halo = p.n_halo;
    GSX_CHECK(w.send.reserve(12 * (size_t)std::max<int64_t>(p.n_send, 1)));
    GSX_CHECK(w.send_src.reserve(4 * (size_t)std::max<int64_t>(n_local, 1)));
    GSX_CHECK(w.slab.reserve(12 * (size_t)std::max<int64_t>(n_own + n_halo, 1)));
    SlabPlan sp;
    sp.world = G;
    sp.axis = p.axis;
    sp.lo = p.lo;
    sp.inv_w = p.hi > p.lo ? (float)SLAB_BINS / (p.hi - p.lo) : 0.0f;
    sp.halo_bins = p.halo_bins;
    for (int s = 0; s <= G; ++s) sp.cut[s] = p.cut[s];
    for (int s = 0; s < G; ++s) {
        sp.off[2 * s] = (unsigned)p.own_off[s];
        sp.off[2 * s + 1] = (unsigned)p.halo_off[s];
    }
    if (n_local > 0) {
        GSX_CHECK(timing_begin(c, GSX_T_SLAB_PREP));
        // (over a wire the rows this rank owns itself go straight into its slab: no self-copy of 1/G of the cloud)
        hipLaunchKernelGGL(slab_partition_kernel, dim3(div_up(n_local, 2048)), dim3(256), 0, c->stream, x, y, z, (int64_t)3, n_local, sp,
                           cursor, w.send.as<float>(), w.send_src.as<unsigned>(), wire ? 2 * r : -1, w.slab.as<float>(),
                           p.r_own_off[r] - p.own_off[r]);
        GSX_HIP(hipGetLastError());
        GSX_CHECK(timing_end(c, GSX_T_SLAB_PREP));
    }
    const float *slab_rows = w.slab.as<float>();
    if (wire) {
        int64_t so[2 * SLAB_MAX_RANKS], sc[2 * SLAB_MAX_RANKS], ro[2 * SLAB_MAX_RANKS], rc[2 * SLAB_MAX_RANKS];
        for (int q = 0; q < G; ++q) {
            so[q] = p.own_off[q]; sc[q] = p.own_cnt[q]; ro[q] = p.r_own_off[q]; rc[q] = p.in_own[q];
            so[G + q] = p.halo_off[q]; sc[G + q] = p.halo_cnt[q]; ro[G + q] = p.r_halo_off[q]; rc[G + q] = p.in_halo[q];
        }
        sc[r] = rc[r] = 0;   // already in place
        GSX_CHECK(timing_begin(c, GSX_T_SLAB_ROWS));
        GSX_CHECK(gsx_comm_all_to_all_segs(c, w.send.p, w.slab.p, 2, so, sc, ro, rc, 12));
        GSX_CHECK(timing_end(c, GSX_T_SLAB_ROWS));
    } else {
        slab_rows = w.send.as<float>();   // no communicator: the (permuted) send buffer IS the slab
    }
    // ---- 4./5. exact KNN on the slab; certificate (device-side count, summed over the ranks, read lazily by the caller)
    GSX_CHECK(w.md_slab.reserve(4 * (size_t)std::max<int64_t>(n_own, 1)));
    GSX_CHECK(w.kth.reserve(8 * (size_t)std::max<int64_t>(n_own, 1)));
    if (n_own > 0) {
        // the grid path's kernels count the uncertified queries themselves (no 8-byte k-th distance array, no kernel of its
        // own reading it back) and take the box from the all-reduced one: along the partition axis the rows lie between the
        // outer edges of the received bins (one bin of margin for the f32 binning arithmetic)
        SlabKnn sk;
        sk.cert_axis = p.axis;
        sk.cert_lo = p.plane_lo;
        sk.cert_hi = p.plane_hi;
        sk.cert_count = unc;
        const double bw = p.hi > p.lo ? ((double)p.hi - (double)p.lo) / SLAB_BINS : 0.0;
        const int b0 = p.cut[r] - p.halo_bins, b1 = p.cut[r + 1] + p.halo_bins;
        sk.box.b7 = b7;
        sk.box.axis = p.axis;
        sk.box.lo = (r == 0 || b0 <= 1) ? -INFINITY : std::nextafter((float)((double)p.lo + ((double)b0 - 1.0) * bw), -INFINITY);
        sk.box.hi = (r == G - 1 || b1 >= SLAB_BINS - 1) ? INFINITY : std::nextafter((float)((double)p.lo + ((double)b1 + 1.0) * bw), INFINITY);
        GSX_CHECK(launch_knn_slab(c, slab_rows, slab_rows + 1, slab_rows + 2, 3, n_own, n_halo, k, w.md_slab.as<float>(), w.kth.as<double>(), &sk));
        if (c->last_knn_algo == GSX_KNN_TREE) {   // an uneven slab took the Morton-tree path, which hands the k-th distances over
            const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n_own, 1024), (int64_t)c->num_cu * 8));
            hipLaunchKernelGGL(slab_certify_kernel, dim3(blocks), dim3(256), 0, c->stream, slab_rows + p.axis, (int64_t)3, n_own,
                               w.kth.as<double>(), p.plane_lo, p.plane_hi, unc);
            GSX_HIP(hipGetLastError());
        }
    }
    // (G > 1: the count is not all-reduced on its own -- it travels behind this rank's piece sums in the first all-gather of
    //  the statistics and is summed by the pack kernel: one small collective less per step)
    // ---- 6. mean distances back to the index owners, original order
    GSX_CHECK(w.md.reserve(4 * (size_t)(n_local + 8192 + 4)));
    const float *ret = w.md_slab.as<float>();
    if (wire) {
        GSX_CHECK(w.ret.reserve(4 * (size_t)std::max<int64_t>(n_local, 1)));
        int64_t sc[SLAB_MAX_RANKS], rc[SLAB_MAX_RANKS];
        for (int q = 0; q < G; ++q) {
            sc[q] = q == r ? 0 : p.in_own[q];    // the own queries' means are read where the KNN left them
            rc[q] = q == r ? 0 : p.own_cnt[q];
        }
        GSX_CHECK(timing_begin(c, GSX_T_SLAB_MEANS));
        GSX_CHECK(gsx_comm_all_to_all_segs(c, w.md_slab.p, w.ret.p, 1, p.r_own_off, sc, p.own_off, rc, 4));
        GSX_CHECK(timing_end(c, GSX_T_SLAB_MEANS));
        ret = w.ret.as<float>();
    }
    float *md = w.md.as<float>();
    // ---- 7. numpy-exact statistics from 8192-element piece sums (pieces are cut at the true GLOBAL offsets: the
    // < 8192 leading elements of a shard belong to the left neighbour's last piece and are handed over)
    int64_t starts[SLAB_MAX_RANKS + 1], heads[SLAB_MAX_RANKS + 1];
    starts[0] = 0;
    for (int q = 0; q < G; ++q) starts[q + 1] = starts[q] + p.sizes[q];
    for (int q = 0; q < G; ++q) heads[q] = (8192 - starts[q] % 8192) % 8192;
    heads[G] = 0;
    const int64_t head = heads[r], nxt_head = heads[r + 1];
    const int64_t n_mine = n_local - head + nxt_head;
    PackPlan pk;
    pk.world = G;
    int max_pieces = 0, total_pieces = 0;
    for (int q = 0; q < G; ++q) {
        const int64_t nq = p.sizes[q] - heads[q] + heads[q + 1];
        pk.count[q] = (int)((nq + 8191) / 8192);
        max_pieces = std::max(max_pieces, pk.count[q]);
        total_pieces += pk.count[q];
    }
    // piece buffer: the sums, then 2 words = this rank's certificate count (even offset: 8-byte aligned)
    const int stride = ((max_pieces + 1) & ~1) + 2;
    pk.stride = stride;
    GSX_CHECK(w.pieces.reserve(4 * (size_t)stride));
    GSX_CHECK(w.allpieces.reserve(4 * (size_t)stride * G));
    GSX_CHECK(w.packed.reserve(4 * (size_t)std::max(total_pieces, 1)));
    unsigned *unc_total = cursor + 34;
    if (n_local > 0) {   // (also hands this rank's certificate count to the tail of the piece buffer)
        GSX_CHECK(timing_begin(c, GSX_T_SLAB_PREP));
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n_local, 1024), (int64_t)c->num_cu * 8));
        const int64_t self_lo = wire ? p.own_off[r] : 0, self_hi = wire ? p.own_off[r] + p.own_cnt[r] : 0;
        hipLaunchKernelGGL(slab_unpermute_kernel, dim3(blocks), dim3(256), 0, c->stream, ret, w.send_src.as<unsigned>(), n_local, md,
                           unc, w.pieces.as<unsigned>() + (stride - 2), self_lo, self_hi,
                           w.md_slab.as<float>() + (p.r_own_off[r] - p.own_off[r]));
        GSX_HIP(hipGetLastError());
        GSX_CHECK(timing_end(c, GSX_T_SLAB_PREP));
    }
    const float *st_in;
    if (head % 4 == 0) {
        st_in = md + head;
    } else {   // piece sums read 16 bytes at a time: an aligned copy (4 B/point, device copy)
        GSX_CHECK(w.md_stats.reserve(4 * (size_t)(n_local + 8192 + 4)));
        GSX_HIP(hipMemcpyAsync(w.md_stats.p, md + head, 4 * (size_t)(n_local - head), hipMemcpyDeviceToDevice, c->stream));
        st_in = w.md_stats.as<float>();
    }
    if (G > 1) {
        int64_t so[SLAB_MAX_RANKS] = {0}, sc[SLAB_MAX_RANKS] = {0}, ro[SLAB_MAX_RANKS] = {0}, rc[SLAB_MAX_RANKS] = {0};
        if (r > 0) sc[r - 1] = head;
        if (r + 1 < G) rc[r + 1] = nxt_head;
        // (send from md[0..head), receive behind the own elements of md / of the aligned copy)
        GSX_CHECK(timing_begin(c, GSX_T_SLAB_MEANS));
        if (head % 4 == 0) {
            for (int q = 0; q < G; ++q) ro[q] = n_local;
            GSX_CHECK(gsx_comm_all_to_all_segs(c, md, md, 1, so, sc, ro, rc, 4));
        } else {
            for (int q = 0; q < G; ++q) ro[q] = n_local - head;
            GSX_CHECK(gsx_comm_all_to_all_segs(c, md, w.md_stats.p, 1, so, sc, ro, rc, 4));
        }
        GSX_CHECK(timing_end(c, GSX_T_SLAB_MEANS));
    }
    for (int mode = 0; mode < 2; ++mode) {
        GSX_CHECK(timing_begin(c, GSX_T_SOR_STATS));
        if (n_mine > 0) GSX_CHECK(launch_sor_piece_sums(c, st_in, n_mine, mode ? stats : nullptr, w.pieces.as<float>()));
        GSX_CHECK(timing_end(c, GSX_T_SOR_STATS));
        const float *pieces = w.pieces.as<float>();
        if (G > 1) {
            GSX_CHECK(timing_begin(c, GSX_T_SLAB_COLL));
            GSX_CHECK(gsx_comm_all_gather(c, w.pieces.p, w.allpieces.p, 4 * (int64_t)stride));
            GSX_CHECK(timing_end(c, GSX_T_SLAB_COLL));
            hipLaunchKernelGGL(slab_pack_pieces_kernel, dim3(4), dim3(256), 0, c->stream, w.allpieces.as<float>(), pk, w.packed.as<float>(),
                               mode == 0 ? reinterpret_cast<unsigned long long *>(unc_total) : nullptr);
            GSX_HIP(hipGetLastError());
            pieces = w.packed.as<float>();
        }
        GSX_CHECK(launch_sor_stats_from_pieces(c, pieces, total_pieces, p.n_total, mode, threshold_factor, stats));
    }
    // ---- 8. mask of the local index range
    uint8_t *mask = mask_out_dev;
    if (!mask) {
        GSX_CHECK(w.mask.reserve((size_t)n_local + 16));
        mask = w.mask.as<uint8_t>();
    }
    GSX_CHECK(launch_sor_mask(c, md, n_local, stats + 2, mask));
    out->mask_dev = mask;
    out->mean_dists_dev = md;
    out->stats_dev = stats;
    out->uncertain_dev = reinterpret_cast<const int64_t *>(G > 1 ? unc_total : unc);
    return 0;
}

}  // extern "C"
