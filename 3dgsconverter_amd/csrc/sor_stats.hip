// sor_stats.hip -- np.mean / np.std / threshold / mask of the SOR filter, bit-exact.
//
// Replaces data_processor.py:176-180 (identical code at gpu_ops.py:259-263):
//     global_mean = np.mean(md); global_std = np.std(md)
//     threshold   = global_mean + threshold_factor * global_std
//     mask        = md < threshold
// A survivor mask only matches the reference bit for bit if these scalars do, so the
// kernels reproduce numpy 2.2.6's float32 reduction exactly (probed, see oracle/gsx_oracle.c):
//   * the array is consumed in 8192-element buffer pieces, accumulated SEQUENTIALLY in f32;
//   * each piece is summed by numpy's pairwise routine: <=128-element leaves with 8
//     accumulators, recursive split at n/2 rounded down to a multiple of 8;
//   * mean = (float)((double)sum / n); var likewise from sum((x-mean)^2); std = sqrtf;
//   * threshold = mean + (float)factor * std in f32.
// One 128-lane workgroup sums one 8192-element piece: two lanes own each 128-element leaf (4 of
// numpy's 8 accumulators each, 16-byte loads), the 64 leaf sums combine in the balanced tree the
// recursion produces for 8192.  The ragged last
// piece is split into its (irregular) leaves by lane 0 and combined by the same recursion.
// HBM-bound and tiny: 4 B/splat per pass, 3 passes.
#include "gsx_common.h"

namespace gsx {

constexpr int NP_BUF = 8192;

template <bool SQ>
__device__ __forceinline__ float elem(float v, float mean)
{
    if (SQ) {
        float d = v - mean;
        return d * d;
    }
    return v;
}

// numpy leaf: n <= 128 elements
template <bool SQ>
__device__ float leaf_sum(const float *__restrict__ a, int n, float mean)
{
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; ++i) res += elem<SQ>(a[i], mean);
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = elem<SQ>(a[j], mean);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += elem<SQ>(a[i + j], mean);
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += elem<SQ>(a[i], mean);
    return res;
}

constexpr int CHUNK_THREADS = 128;  // one workgroup (2 waves) per 8192-element piece

template <bool SQ>
__global__ __launch_bounds__(CHUNK_THREADS) void chunk_sums_kernel(const float *__restrict__ a, int64_t n,
                                                                   const float *__restrict__ stats,
                                                                   float *__restrict__ chunk_sum)
{
    __shared__ int s_leaf_start[160];
    __shared__ int s_leaf_len[160];
    __shared__ float s_leaf_val[160];
    __shared__ float s_half[2];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int64_t c = blockIdx.x;
    const float mean = SQ ? stats[0] : 0.0f;
    const float *p = a + c * NP_BUF;
    const int len = (int)((n - c * NP_BUF) < NP_BUF ? (n - c * NP_BUF) : NP_BUF);

    if (len == NP_BUF) {  // block-uniform
        // 64 leaves of 128 elements; two lanes per leaf: lane (leaf, hh) owns accumulators 4hh..4hh+3
        // (numpy's r[0..7]) and walks the leaf's 16 rows with one 16-byte load per row.
        const int leaf = threadIdx.x >> 1, hh = threadIdx.x & 1;
        const float4 *q = reinterpret_cast<const float4 *>(p + leaf * 128 + hh * 4);
        float4 v = q[0];
        float r0 = elem<SQ>(v.x, mean), r1 = elem<SQ>(v.y, mean), r2 = elem<SQ>(v.z, mean), r3 = elem<SQ>(v.w, mean);
#pragma unroll
        for (int i = 1; i < 16; ++i) {
            v = q[2 * i];  // +8 floats per row
            r0 += elem<SQ>(v.x, mean); r1 += elem<SQ>(v.y, mean); r2 += elem<SQ>(v.z, mean); r3 += elem<SQ>(v.w, mean);
        }
        float s = (r0 + r1) + (r2 + r3);      // (r0+r1)+(r2+r3)  |  (r4+r5)+(r6+r7)
        s = s + __shfl_xor(s, 1);             // leaf sum (both lanes of the pair hold it)
        // balanced tree over adjacent leaves: 128 -> 256 -> ... -> 4096 inside the wave
#pragma unroll
        for (int off = 2; off < 64; off <<= 1) s = s + __shfl_xor(s, off);
        if (lane == 0) s_half[wv] = s;
        __syncthreads();
        if (threadIdx.x == 0) chunk_sum[c] = s_half[0] + s_half[1];  // 4096 + 4096
        return;
    }

    // ragged last piece (wave 0 only): enumerate the recursion's leaves in order (lane 0), sum the
    // leaves in parallel, then replay the recursion over the leaf sums (lane 0).
    if (wv != 0) return;
    int *ls = s_leaf_start, *ll = s_leaf_len;
    float *lv = s_leaf_val;
    int nleaf = 0;
    if (lane == 0) {
        int st_n[20], st_o[20], sp = 0;
        st_n[0] = len; st_o[0] = 0; sp = 1;
        while (sp > 0) {
            int nn = st_n[sp - 1], oo = st_o[sp - 1];
            --sp;
            if (nn <= 128) {
                ls[nleaf] = oo; ll[nleaf] = nn; ++nleaf;
            } else {
                int n2 = nn / 2;
                n2 -= n2 % 8;
                // push right first so the left half is expanded first (in-order leaves)
                st_n[sp] = nn - n2; st_o[sp] = oo + n2; ++sp;
                st_n[sp] = n2; st_o[sp] = oo; ++sp;
            }
        }
    }
    nleaf = __shfl(nleaf, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int l = lane; l < nleaf; l += 64) lv[l] = leaf_sum<SQ>(p + ls[l], ll[l], mean);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        // replay: post-order evaluation of S(n) = S(n2) + S(n - n2)
        float vals[24];
        int st_n[24];
        signed char st_s[24];
        int sp = 1, vp = 0, li = 0;
        st_n[0] = len; st_s[0] = 0;
        while (sp > 0) {
            int nn = st_n[sp - 1];
            int s = st_s[sp - 1];
            if (nn <= 128) {
                vals[vp++] = lv[li++];
                --sp;
            } else {
                int n2 = nn / 2;
                n2 -= n2 % 8;
                if (s == 0) { st_s[sp - 1] = 1; st_n[sp] = n2; st_s[sp] = 0; ++sp; }
                else if (s == 1) { st_s[sp - 1] = 2; st_n[sp] = nn - n2; st_s[sp] = 0; ++sp; }
                else { float rr = vals[vp - 2] + vals[vp - 1]; vp -= 2; vals[vp++] = rr; --sp; }
            }
        }
        chunk_sum[c] = vals[0];
    }
}

// mode 0: stats[0] = mean.  mode 1: stats[1] = std, stats[2] = threshold.
__global__ __launch_bounds__(256) void stats_finalize_kernel(const float *__restrict__ chunk_sum, int64_t n,
                                                             int mode, float factor, float *__restrict__ stats)
{
    __shared__ float buf[8192];
    const int64_t nchunks = (n + NP_BUF - 1) / NP_BUF;
    float acc = 0.0f;
    for (int64_t base = 0; base < nchunks; base += 8192) {
        const int m = (int)((nchunks - base) < 8192 ? (nchunks - base) : 8192);
        for (int i = threadIdx.x; i < m; i += blockDim.x) buf[i] = chunk_sum[base + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 0; i < m; ++i) acc += buf[i];  // numpy: sequential over buffer pieces
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const float q = (float)((double)acc / (double)n);  // f32 sum / np.intp count: float64 divide, cast back
    if (mode == 0) {
        stats[0] = q;
    } else {
        const float sd = __builtin_sqrtf(q);
        stats[1] = sd;
        stats[2] = stats[0] + factor * sd;
    }
}

__global__ __launch_bounds__(256) void sor_mask_kernel(const float *__restrict__ md, int64_t n,
                                                       const float *__restrict__ thr_p, uint8_t *__restrict__ mask)
{
    const float thr = *thr_p;
    const int64_t n4 = n / 4;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += step) {
        float4 v = reinterpret_cast<const float4 *>(md)[i];
        uchar4 o;
        o.x = v.x < thr; o.y = v.y < thr; o.z = v.z < thr; o.w = v.w < thr;
        reinterpret_cast<uchar4 *>(mask)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        int64_t i = n4 * 4 + threadIdx.x;
        mask[i] = md[i] < thr;
    }
}

int launch_sor_stats(gsx_ctx *ctx, const float *md, int64_t n, double factor, float *stats_dev)
{
    if (n <= 0) GSX_FAIL("sor_stats: empty input");
    const int64_t nchunks = (n + NP_BUF - 1) / NP_BUF;
    GSX_CHECK(ctx->statspart.reserve(sizeof(float) * (size_t)nchunks));
    float *cs = ctx->statspart.as<float>();
    const int blocks = (int)nchunks;
    const float tf = (float)factor;  // python float is a weak scalar: rounded to f32 first
    hipLaunchKernelGGL((chunk_sums_kernel<false>), dim3(blocks), dim3(CHUNK_THREADS), 0, ctx->stream, md, n, stats_dev, cs);
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(1), dim3(256), 0, ctx->stream, cs, n, 0, tf, stats_dev);
    hipLaunchKernelGGL((chunk_sums_kernel<true>), dim3(blocks), dim3(CHUNK_THREADS), 0, ctx->stream, md, n, stats_dev, cs);
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(1), dim3(256), 0, ctx->stream, cs, n, 1, tf, stats_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

int launch_sor_mask(gsx_ctx *ctx, const float *md, int64_t n, const float *thr_dev, uint8_t *mask)
{
    if (n <= 0) return 0;
    if ((reinterpret_cast<uintptr_t>(md) & 15) || (reinterpret_cast<uintptr_t>(mask) & 3))
        GSX_FAIL("sor_mask: mean_dists must be 16-byte and mask 4-byte aligned");
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n / 4 + 1, 256), (int64_t)ctx->num_cu * 8));
    hipLaunchKernelGGL(sor_mask_kernel, dim3(blocks), dim3(256), 0, ctx->stream, md, n, thr_dev, mask);
    GSX_HIP(hipGetLastError());
    return 0;
}

}  // namespace gsx
