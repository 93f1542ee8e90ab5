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
// assign: one lane per point, the point's D coordinates live in registers; centroids are
// read with wave-uniform addresses, i.e. through the scalar cache straight into SGPR
// operands of the VALU ops -- no LDS, no per-lane centroid traffic.  2 VALU ops per
// (point, centroid, dim): bound by FP32 VALU issue, not HBM.
// update: float64 hardware atomics (order-insensitive to ~1e-16, unlike the reference's f32
// atomics), then one pass over K x D.
#include "gsx_common.h"

namespace gsx {

template <int D>
__global__ __launch_bounds__(256) void kmeans_assign_kernel(const float *__restrict__ data, int64_t n,
                                                            const float *__restrict__ cent, int k,
                                                            int32_t *__restrict__ labels)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ii = i < n ? i : n - 1;
    float xv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xv[d] = data[ii * D + d];
    float best = 1e20f;  // gpu_ops.py:60
    int bi = -1;
    for (int c = 0; c < k; ++c) {
        const float *__restrict__ cc = cent + (int64_t)c * D;  // wave-uniform: scalar loads
        float dist = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float diff = xv[d] - cc[d];
            dist = __builtin_fmaf(diff, diff, dist);
        }
        if (dist < best) {
            best = dist;
            bi = c;
        }
    }
    if (i < n) labels[i] = bi;
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
        unsafeAtomicAdd(&sums[(int64_t)l * D + d], (double)data[e]);
        if (d == 0) atomicAdd(&counts[l], 1u);
    }
}

__global__ __launch_bounds__(256) void kmeans_finalize_kernel(const double *__restrict__ sums,
                                                              const unsigned *__restrict__ counts, int k, int D,
                                                              float *__restrict__ cent)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= k * D) return;
    const unsigned cnt = counts[e / D];
    // gpu_ops.py:91-96: inv = 1/float(cnt); centroid *= inv; an empty cluster keeps the 0 it was reset to
    cent[e] = cnt > 0 ? (float)sums[e] * (1.0f / (float)cnt) : 0.0f;
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

template <int D>
static void launch_assign_t(gsx_ctx *c, const float *data, int64_t n, const float *cent, int k, int32_t *labels)
{
    hipLaunchKernelGGL((kmeans_assign_kernel<D>), dim3(div_up(n, 256)), dim3(256), 0, c->stream, data, n, cent, k, labels);
}

static int launch_assign(gsx_ctx *c, const float *data, int64_t n, int d, const float *cent, int k, int32_t *labels)
{
    switch (d) {
        case 1: launch_assign_t<1>(c, data, n, cent, k, labels); break;
        case 2: launch_assign_t<2>(c, data, n, cent, k, labels); break;
        case 3: launch_assign_t<3>(c, data, n, cent, k, labels); break;
        case 4: launch_assign_t<4>(c, data, n, cent, k, labels); break;
        case 9: launch_assign_t<9>(c, data, n, cent, k, labels); break;
        case 24: launch_assign_t<24>(c, data, n, cent, k, labels); break;
        case 45: launch_assign_t<45>(c, data, n, cent, k, labels); break;
        default:
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
    for (int it = 0; it < max_iter; ++it) {
        GSX_CHECK(timing_begin(c, GSX_T_KMEANS_ASSIGN));
        GSX_CHECK(launch_assign(c, data_dev, n, d, cent_dev, k, labels_dev));
        GSX_CHECK(timing_end(c, GSX_T_KMEANS_ASSIGN));
        GSX_CHECK(timing_begin(c, GSX_T_KMEANS_UPDATE));
        GSX_HIP(hipMemsetAsync(sums, 0, sizeof(double) * kd + sizeof(unsigned) * (size_t)k, c->stream));
        hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3(acc_blocks), dim3(256), 0, c->stream, data_dev, n, d, labels_dev,
                           sums, counts);
        hipLaunchKernelGGL(kmeans_finalize_kernel, dim3(div_up((int64_t)kd, 256)), dim3(256), 0, c->stream, sums, counts, k,
                           d, cent_dev);
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
