// kmeans.hip -- Lloyd K-Means for the SOG writer's codebooks + sorted-codebook quantiser.
//
// Replaces the reference's Taichi kernels and driver:
//   gpu_ops.py:57-73   k_means_assign : brute-force argmin_c sum_d (x_d - c_d)^2 in f32,
//                                       dims accumulated in order, strict '<' (lowest index wins)
//   gpu_ops.py:75-96   k_means_update : zero, accumulate, divide; EMPTY cluster -> 0-vector
//   gpu_ops.py:178-191 driver         : exactly max_iter x (assign, update), returned labels
//                                       are one step older than the returned centroids
//   formats/sog.py:408-419 quantize_to_codebook (searchsorted + left-neighbour check)
// The data stays resident in HBM across iterations (the reference re-uploads it for every
// kernel call).
//
// assign (+ the accumulation half of update, fused): two points per lane with their D
// coordinates in registers; centroids are read with wave-uniform addresses, i.e. through the
// scalar cache straight into SGPR operands of packed-f32 VALU ops -- no per-lane centroid
// traffic.  2 VALU lane-ops per (point, centroid, dim): bound by FP32 VALU issue, not HBM
// (SURVEY.md 8(d)).  Cluster sums are float64 hardware atomics (order-insensitive to ~1e-16,
// unlike the reference's f32 atomics), in HBM or -- for small codebooks -- in LDS; one small
// kernel per iteration divides and re-zeroes.
#include "gsx_common.h"

namespace gsx {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Work split: a workgroup of NW waves owns 128 points (2 per lane: rows p and p+64 of the
// tile) and each wave scans 1/NW of the centroids; the per-wave winners are merged in
// centroid order with strict '<', which is the sequential scan's "lowest index wins".
// Splitting K instead of N keeps every SIMD busy for the small N of a SOG chunk (156 250 rows
// are only 1 221 tiles), and with NW=16 a tile is one CU-filling workgroup of short waves.
// The two points of a lane share each scalar-loaded centroid value through one v_pk_add_f32 +
// one v_pk_fma_f32 (dims still accumulated in order for each point).  The tile sits in LDS,
// so k_means_update's accumulation runs in the same launch.
constexpr int KM_TILE = 128;

template <int D, int NW, bool LACC>
__global__ __launch_bounds__(64 * NW) void kmeans_assign_kernel(const float *__restrict__ data, int64_t n,
                                                                 const float *__restrict__ cent, int k,
                                                                 int32_t *__restrict__ labels,
                                                                 double *__restrict__ sums, unsigned *__restrict__ counts)
{
    constexpr int DP = D | 1;  // odd LDS row stride: conflict-free per-lane row reads
    __shared__ float s_tile[KM_TILE * DP];
    __shared__ float s_best[NW][KM_TILE];
    __shared__ int s_idx[NW][KM_TILE];
    // LACC (k * D <= KM_LACC_MAX): the workgroup walks many tiles and keeps the cluster sums in
    // LDS (ds_add_f64), flushing once -- a small codebook over a large N would otherwise
    // serialise on k * D global atomic addresses (measured 15 ms / iteration at 30M x 1, k=256)
    extern __shared__ double s_acc[];  // [k * D] sums, then [k] counts
    unsigned *s_cnt = reinterpret_cast<unsigned *>(s_acc + (LACC ? k * D : 0));
    if (LACC) {
        for (int e = threadIdx.x; e < k * D; e += 64 * NW) s_acc[e] = 0.0;
        for (int e = threadIdx.x; e < k; e += 64 * NW) s_cnt[e] = 0u;
    }
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = (int)(((int64_t)k * w) / NW), c1 = (int)(((int64_t)k * (w + 1)) / NW);
    const int64_t tiles = (n + KM_TILE - 1) / KM_TILE;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t base = tile * KM_TILE;
        const int rows = (int)(n - base < KM_TILE ? n - base : KM_TILE);
        __syncthreads();  // previous tile fully consumed (and the LACC zeroing done)
        // the tile's rows are contiguous in the row-major input: one coalesced copy into LDS
        for (int e = threadIdx.x; e < rows * D; e += 64 * NW) {
            const int r = e / D;
            s_tile[r * DP + (e - r * D)] = data[base * D + e];
        }
        __syncthreads();
        const int ra = lane < rows ? lane : rows - 1;
        const int rb = lane + 64 < rows ? lane + 64 : rows - 1;
        f32x2 xv[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            xv[d].x = s_tile[ra * DP + d];
            xv[d].y = s_tile[rb * DP + d];
        }
        f32x2 best = {1e20f, 1e20f};  // gpu_ops.py:60
        int bia = -1, bib = -1;
        for (int c = c0; c < c1; ++c) {
            // wave-uniform address: s_load into SGPRs, broadcast to both halves of the packed
            // op by op_sel.  (Issuing a row's s_loads behind one wait by hand changed nothing:
            // four waves per SIMD already cover the scalar-cache latency.)
            const float *__restrict__ cc = cent + (int64_t)c * D;
            f32x2 dist = {0.0f, 0.0f};
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const f32x2 cv = {cc[d], cc[d]};
                const f32x2 diff = xv[d] - cv;
                dist = __builtin_elementwise_fma(diff, diff, dist);  // dims in order (gpu_ops.py:63-66)
            }
            if (dist.x < best.x) {
                best.x = dist.x;
                bia = c;
            }
            if (dist.y < best.y) {
                best.y = dist.y;
                bib = c;
            }
        }
        s_best[w][lane] = best.x;
        s_best[w][lane + 64] = best.y;
        s_idx[w][lane] = bia;
        s_idx[w][lane + 64] = bib;
        __syncthreads();
        if (threadIdx.x < KM_TILE) {
            float b = s_best[0][threadIdx.x];
            int bi = s_idx[0][threadIdx.x];
#pragma unroll
            for (int q = 1; q < NW; ++q) {
                const float v = s_best[q][threadIdx.x];
                if (v < b) {
                    b = v;
                    bi = s_idx[q][threadIdx.x];
                }
            }
            s_idx[0][threadIdx.x] = bi;
            if (threadIdx.x < rows) {
                // bi == -1: every distance was NaN or >= 1e20 (non-finite or huge input).  The reference
                // then indexes centroids[-1] (gpu_ops.py:85-88: undefined in Taichi); here the point keeps
                // label -1 and contributes to no cluster.
                labels[base + threadIdx.x] = bi;
                if (bi >= 0) atomicAdd(LACC ? &s_cnt[bi] : &counts[bi], 1u);
            }
        }
        __syncthreads();
        // k_means_update's accumulation (gpu_ops.py:83-89) while the tile is still in LDS
        for (int e = threadIdx.x; e < rows * D; e += 64 * NW) {
            const int r = e / D, d = e - r * D;
            const int lbl = s_idx[0][r];
            if (lbl < 0) continue;  // unassignable point (see above)
            const int64_t slot = (int64_t)lbl * D + d;
            unsafeAtomicAdd(LACC ? &s_acc[slot] : &sums[slot], (double)s_tile[r * DP + d]);
        }
    }
    if (LACC) {
        __syncthreads();
        for (int e = threadIdx.x; e < k; e += 64 * NW) {
            const unsigned cnt = s_cnt[e];
            if (cnt == 0u) continue;
            atomicAdd(&counts[e], cnt);
            for (int d = 0; d < D; ++d) unsafeAtomicAdd(&sums[(int64_t)e * D + d], s_acc[e * D + d]);
        }
    }
}

// any D (slow path): coordinates re-read from memory
__global__ __launch_bounds__(256) void kmeans_assign_generic_kernel(const float *__restrict__ data, int64_t n, int D,
                                                                    const float *__restrict__ cent, int k,
                                                                    int32_t *__restrict__ labels)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float best = 1e20f;
    int bi = -1;
    for (int c = 0; c < k; ++c) {
        float dist = 0.0f;
        for (int d = 0; d < D; ++d) {
            float diff = data[i * D + d] - cent[(int64_t)c * D + d];
            dist = __builtin_fmaf(diff, diff, dist);
        }
        if (dist < best) {
            best = dist;
            bi = c;
        }
    }
    labels[i] = bi;
}

__global__ __launch_bounds__(256) void kmeans_accumulate_kernel(const float *__restrict__ data, int64_t n, int D,
                                                                const int32_t *__restrict__ labels,
                                                                double *__restrict__ sums, unsigned *__restrict__ counts)
{
    // one lane per (point, dim) element: coalesced reads of the row-major data
    const int64_t total = n * D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / D;
        const int d = (int)(e - i * D);
        const int l = labels[i];
        if (l < 0) continue;  // unassignable point: label -1, no contribution
        unsafeAtomicAdd(&sums[(int64_t)l * D + d], (double)data[e]);
        if (d == 0) atomicAdd(&counts[l], 1u);
    }
}

// one workgroup per centroid (gpu_ops.py:91-96: inv = 1/float(cnt); centroid *= inv; an empty
// cluster keeps the 0 it was reset to); also re-zeroes the accumulators for the next iteration
// (gpu_ops.py:77-81) so that the fused assign+accumulate kernel needs no memset in between
__global__ __launch_bounds__(64) void kmeans_finalize_reset_kernel(double *__restrict__ sums, unsigned *__restrict__ counts,
                                                                   int D, float *__restrict__ cent)
{
    const int c = blockIdx.x;
    const unsigned cnt = counts[c];
    const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.0f;
    for (int d = threadIdx.x; d < D; d += 64) {
        const int64_t e = (int64_t)c * D + d;
        cent[e] = cnt > 0 ? (float)sums[e] * inv : 0.0f;
        sums[e] = 0.0;
    }
    __syncthreads();  // every lane has read counts[c]
    if (threadIdx.x == 0) counts[c] = 0u;
}

// ---- quantize_to_codebook (formats/sog.py:408-419) ----------------------------------------
__global__ __launch_bounds__(256) void quantize_kernel(const float *__restrict__ vals, int64_t n,
                                                       const float *__restrict__ cb, int kcb,
                                                       uint8_t *__restrict__ out)
{
    __shared__ float lcb[256];
    for (int i = threadIdx.x; i < kcb; i += 256) lcb[i] = cb[i];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = vals[i];
        int lo = 0, hi = kcb;  // np.searchsorted(cb, v, 'left'): first index with cb[idx] >= v
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (lcb[mid] < v) lo = mid + 1; else hi = mid;
        }
        int idx = min(lo, kcb - 1);
        int left = max(idx - 1, 0);
        float d_idx = fabsf(v - lcb[idx]);
        float d_left = fabsf(v - lcb[left]);
        if (d_left < d_idx) idx = left;  // strict: ties go to the right neighbour
        out[i] = (uint8_t)idx;
    }
}

constexpr int KM_LACC_MAX = 2048;  // k * D doubles of LDS accumulators (16 KiB + counts)

template <int D>
static void launch_assign_t(gsx_ctx *c, const float *data, int64_t n, const float *cent, int k, int32_t *labels,
                            double *sums, unsigned *counts)
{
    const int64_t tiles = div_up(n, KM_TILE);
    if constexpr (D <= 4) {
        if (k * D <= KM_LACC_MAX && tiles > (int64_t)c->num_cu * 8) {
            const size_t lds = sizeof(double) * (size_t)k * D + sizeof(unsigned) * (size_t)k;
            hipLaunchKernelGGL((kmeans_assign_kernel<D, 4, true>), dim3(c->num_cu * 8), dim3(256), lds, c->stream, data, n,
                               cent, k, labels, sums, counts);
            return;
        }
    }
    // few tiles and enough centroids per wave: split K sixteen ways (one workgroup fills a CU)
    if (tiles < (int64_t)c->num_cu * 32 && k >= 128)
        hipLaunchKernelGGL((kmeans_assign_kernel<D, 16, false>), dim3((unsigned)tiles), dim3(1024), 0, c->stream, data, n,
                           cent, k, labels, sums, counts);
    else
        hipLaunchKernelGGL((kmeans_assign_kernel<D, 4, false>), dim3((unsigned)tiles), dim3(256), 0, c->stream, data, n,
                           cent, k, labels, sums, counts);
}

// returns 0 and sets *fused when the templated kernel (assign + accumulate in one launch) ran
static int launch_assign(gsx_ctx *c, const float *data, int64_t n, int d, const float *cent, int k, int32_t *labels,
                         double *sums, unsigned *counts, bool *fused)
{
    *fused = true;
    switch (d) {
        case 1: launch_assign_t<1>(c, data, n, cent, k, labels, sums, counts); break;
        case 2: launch_assign_t<2>(c, data, n, cent, k, labels, sums, counts); break;
        case 3: launch_assign_t<3>(c, data, n, cent, k, labels, sums, counts); break;
        case 4: launch_assign_t<4>(c, data, n, cent, k, labels, sums, counts); break;
        case 9: launch_assign_t<9>(c, data, n, cent, k, labels, sums, counts); break;
        case 24: launch_assign_t<24>(c, data, n, cent, k, labels, sums, counts); break;
        case 45: launch_assign_t<45>(c, data, n, cent, k, labels, sums, counts); break;
        default:
            *fused = false;
            hipLaunchKernelGGL(kmeans_assign_generic_kernel, dim3(div_up(n, 256)), dim3(256), 0, c->stream, data, n, d,
                               cent, k, labels);
    }
    GSX_HIP(hipGetLastError());
    return 0;
}

// data_dev: n x d, cent_dev: k x d (in: init, out: result), labels_dev: n
int kmeans_lloyd_dev(gsx_ctx *c, const float *data_dev, int64_t n, int d, int k, int max_iter, float *cent_dev,
                     int32_t *labels_dev)
{
    if (n <= 0 || d <= 0 || k <= 0) GSX_FAIL("kmeans: bad shape n=%lld d=%d k=%d", (long long)n, d, k);
    const size_t kd = (size_t)k * d;
    GSX_CHECK(c->scratch3.reserve(sizeof(double) * kd + sizeof(unsigned) * (size_t)k + 64));
    double *sums = c->scratch3.as<double>();
    unsigned *counts = reinterpret_cast<unsigned *>(c->scratch3.as<char>() + sizeof(double) * kd);
    const int acc_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n * d, 256), (int64_t)c->num_cu * 16));
    GSX_HIP(hipMemsetAsync(sums, 0, sizeof(double) * kd + sizeof(unsigned) * (size_t)k, c->stream));
    for (int it = 0; it < max_iter; ++it) {
        bool fused = false;
        GSX_CHECK(timing_begin(c, GSX_T_KMEANS_ASSIGN));
        GSX_CHECK(launch_assign(c, data_dev, n, d, cent_dev, k, labels_dev, sums, counts, &fused));
        GSX_CHECK(timing_end(c, GSX_T_KMEANS_ASSIGN));
        GSX_CHECK(timing_begin(c, GSX_T_KMEANS_UPDATE));
        if (!fused)
            hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3(acc_blocks), dim3(256), 0, c->stream, data_dev, n, d,
                               labels_dev, sums, counts);
        hipLaunchKernelGGL(kmeans_finalize_reset_kernel, dim3(k), dim3(64), 0, c->stream, sums, counts, d, cent_dev);
        GSX_HIP(hipGetLastError());
        GSX_CHECK(timing_end(c, GSX_T_KMEANS_UPDATE));
    }
    return 0;
}

int quantize_dev(gsx_ctx *c, const float *vals_dev, int64_t n, const float *cb_dev, int kcb, uint8_t *out_dev)
{
    if (kcb < 1 || kcb > 256) GSX_FAIL("quantize: codebook size %d not in [1,256]", kcb);
    if (n <= 0) return 0;
    GSX_CHECK(timing_begin(c, GSX_T_QUANTIZE));
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 1024), (int64_t)c->num_cu * 8));
    hipLaunchKernelGGL(quantize_kernel, dim3(blocks), dim3(256), 0, c->stream, vals_dev, n, cb_dev, kcb, out_dev);
    GSX_HIP(hipGetLastError());
    GSX_CHECK(timing_end(c, GSX_T_QUANTIZE));
    return 0;
}

}  // namespace gsx
