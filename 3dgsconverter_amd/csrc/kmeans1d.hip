// kmeans1d.hip -- scalar (1-D) K-Means codebooks on the device, deterministic, k-means++-or-better quality.
//
// Replaces, for D = 1:
//   formats/sog.py:561        MiniBatchKMeans(n_clusters=256, n_init='auto').fit(centroids_flat)   (2.9 M scalars at 10M splats /
//                             compression_level 2 -- hard-wired to sklearn in the reference)
//   processing/gpu_ops.py:48-52  _kmeans_sklearn (k-means++-seeded MiniBatchKMeans) for the scalar codebooks of
//                             formats/sog.py:392-403,435-445 -- what the reference produces on a machine without Taichi
// Both are UNSEEDED in the reference, so no bit pattern exists to match: admissible = a codebook of at least that quality
// (tests assert inertia <= sklearn's on the same data; measured 0.2x - 0.9x).
//
// 1-D K-Means is a problem on a SORTED array: clusters are consecutive runs, a Lloyd step moves the K-1 run boundaries to the
// midpoints of adjacent centroids and the new centroid of a run is a difference of two prefix sums.  So:
//   1. radix sort of the values (rocPRIM, order-preserving u32 keys), float64 prefix sums of x and x^2 (two-level, fixed
//      order => deterministic);
//   2. ONE workgroup, thread j = centroid j, runs ALL Lloyd iterations in one launch: a boundary is a binary search
//      (22 dependent L2 loads at 2.9 M values), a centroid two prefix lookups: O(K log N) per iteration instead of O(N K);
//   3. initial centroids from the companding rule of optimal scalar quantisers (point density ~ p(x)^(1/3)): quantiles of
//      count^(1/3) over 1024 uniform bins of [min, max], and -- for heavy-tailed data where uniform bins cannot resolve the
//      core -- of width^(2/3) over 1024 equal-count bins; both are iterated and the lower inertia wins.  (Measured against
//      MiniBatchKMeans on Gaussian / bimodal / outlier / discrete data: 0.2 - 0.9 x its inertia after <= 50 iterations;
//      plain quantile starts need 100+.)
// Centroids come out ascending (callers sort them anyway: sog.py:403,444,562); an empty run keeps its previous centroid.
// Cost: sort + scan of N values (HBM-bound, 36 B/value) + ~0.3 ms of latency-bound iterations; the reference's sklearn call
// takes 0.3 - 1 s.
#include <rocprim/device/device_radix_sort.hpp>

#include "gsx_common.h"

namespace gsx {

constexpr int K1_TILE = 1024;   // values per prefix tile
constexpr int K1_BINS = 1024;   // bins of the companding histograms
constexpr int K1_MAXK = 1024;   // one thread per centroid

__device__ __forceinline__ unsigned k1_key(float v)     // ascending float order as unsigned order; NaN last
{
    if (v != v) return 0xffffffffu;
    if (v == 0.0f) v = 0.0f;
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float k1_unkey(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void k1_keys_kernel(const float *__restrict__ v, int64_t n, unsigned *__restrict__ keys, unsigned *__restrict__ flags)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float x = v[i];
        bad |= !(fabsf(x) <= 3.0e38f);
        keys[i] = k1_key(x);
    }
    if (bad) atomicOr(flags, 1u);
}

// sorted keys -> sorted values + exclusive float64 prefix sums INSIDE each 1024-value tile (x and x^2) + tile totals.
// Fixed summation order (4 consecutive values per thread, wave scan, 4 wave totals): bit-reproducible.
__global__ __launch_bounds__(256) void k1_tile_prefix_kernel(const unsigned *__restrict__ keys, int64_t n, float *__restrict__ xs,
                                                             double *__restrict__ p1, double *__restrict__ p2, double *__restrict__ tiles)
{
    __shared__ double s_w[2][4];
    const int64_t base = (int64_t)blockIdx.x * K1_TILE + threadIdx.x * 4;
    double x[4], a1[4], a2[4];
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float f = base + j < n ? k1_unkey(keys[base + j]) : 0.0f;
        if (base + j < n) xs[base + j] = f;
        x[j] = (double)f;
        a1[j] = t1;
        a2[j] = t2;
        t1 += x[j];
        t2 += x[j] * x[j];
    }
    // inclusive wave scan of the thread totals
    double i1 = t1, i2 = t2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double u1 = __shfl_up(i1, off), u2 = __shfl_up(i2, off);
        if (lane >= off) {
            i1 += u1;
            i2 += u2;
        }
    }
    if (lane == 63) {
        s_w[0][wv] = i1;
        s_w[1][wv] = i2;
    }
    __syncthreads();
    double w1 = 0.0, w2 = 0.0;
    for (int w = 0; w < wv; ++w) {
        w1 += s_w[0][w];
        w2 += s_w[1][w];
    }
    const double e1 = w1 + (i1 - t1), e2 = w2 + (i2 - t2);   // exclusive prefix of this thread inside the tile
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (base + j < n) {
            p1[base + j] = e1 + a1[j];
            p2[base + j] = e2 + a2[j];
        }
    if (threadIdx.x == 255) {
        tiles[2 * (int64_t)blockIdx.x] = w1 + i1;
        tiles[2 * (int64_t)blockIdx.x + 1] = w2 + i2;
    }
}

// exclusive scan of the tile totals, one workgroup, sequential over chunks of 256 tiles (deterministic)
__global__ __launch_bounds__(256) void k1_tile_scan_kernel(double *__restrict__ tiles, int64_t ntiles)
{
    __shared__ double s_w[2][4];
    __shared__ double s_carry[2];
    if (threadIdx.x == 0) s_carry[0] = s_carry[1] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t c0 = 0; c0 <= ntiles; c0 += 256) {   // entry [ntiles] receives the grand total
        const int64_t i = c0 + threadIdx.x;
        const double t1 = i < ntiles ? tiles[2 * i] : 0.0, t2 = i < ntiles ? tiles[2 * i + 1] : 0.0;
        double i1 = t1, i2 = t2;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double u1 = __shfl_up(i1, off), u2 = __shfl_up(i2, off);
            if (lane >= off) {
                i1 += u1;
                i2 += u2;
            }
        }
        if (lane == 63) {
            s_w[0][wv] = i1;
            s_w[1][wv] = i2;
        }
        __syncthreads();
        double w1 = s_carry[0], w2 = s_carry[1];
        for (int w = 0; w < wv; ++w) {
            w1 += s_w[0][w];
            w2 += s_w[1][w];
        }
        if (i <= ntiles) {
            tiles[2 * i] = w1 + (i1 - t1);
            tiles[2 * i + 1] = w2 + (i2 - t2);
        }
        __syncthreads();
        if (threadIdx.x == 255) {
            s_carry[0] = w1 + i1;
            s_carry[1] = w2 + i2;
        }
        __syncthreads();
    }
}

struct K1View {
    const float *xs;
    const double *p1, *p2, *tiles;
    int64_t n;
    // sum of xs[0..i) (and of squares), i in [0, n]
    __device__ __forceinline__ double s1(int64_t i) const { return i >= n ? tiles[2 * ((n + K1_TILE - 1) / K1_TILE)] : tiles[2 * (i / K1_TILE)] + p1[i]; }
    __device__ __forceinline__ double s2(int64_t i) const { return i >= n ? tiles[2 * ((n + K1_TILE - 1) / K1_TILE) + 1] : tiles[2 * (i / K1_TILE) + 1] + p2[i]; }
    // first index whose value is > t (t in float64: midpoints of two floats are exact there)
    __device__ __forceinline__ int64_t upper(double t) const
    {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((double)xs[mid] > t) hi = mid; else lo = mid + 1;
        }
        return lo;
    }
    // the same index, searched outwards from where the boundary was one Lloyd step ago (round 6: a boundary moves by a few values per
    // step once the centroids settle; the 22 dependent loads of the plain bisection were most of an iteration).  g < 0: no guess.
    __device__ __forceinline__ int64_t upper_from(double t, int64_t g) const
    {
        if (g < 0) return upper(t);
        int64_t lo = 0, hi = n;
        if (g >= n || (double)xs[g] > t) {          // the answer is at or below g
            hi = g < n ? g : n;
            for (int64_t step = 1;; step <<= 1) {
                const int64_t p = hi - step;
                if (p < 0) break;
                if ((double)xs[p] > t) hi = p; else { lo = p + 1; break; }
            }
        } else {                                    // xs[g] <= t: the answer is above g
            lo = g + 1;
            for (int64_t step = 1;; step <<= 1) {
                const int64_t p = lo + step - 1;
                if (p >= n) break;
                if ((double)xs[p] > t) { hi = p; break; }
                lo = p + 1;
            }
        }
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((double)xs[mid] > t) hi = mid; else lo = mid + 1;
        }
        return lo;
    }
};

__device__ __forceinline__ double k1_block_sum(double v, double *s_red /* [16] */)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x + 63) / 64; ++w) t += s_red[w];
    return t;
}

// One workgroup PER START, K threads each (rounded up to a wave).  mode_mask: bit 0 = uniform-bin start, bit 1 = equal-count-bin start.
// Round 6: the two starts used to run one after the other in ONE workgroup (a launch is a chain of dependent loads: latency, not
// throughput); now each has a workgroup of its own, parks its centroids in cent_tmp and the last one to arrive picks the winner by
// the old rule (start A unless B's inertia is strictly smaller).
__global__ __launch_bounds__(K1_MAXK) void k1_lloyd_kernel(K1View v, int k, int iters, int mode_mask, float *__restrict__ cent_out,
                                                           double *__restrict__ inertia_out /* [3]: chosen, start A, start B */,
                                                           float *__restrict__ cent_tmp /* [2][K1_MAXK] */, unsigned *__restrict__ ticket)
{
    __shared__ double s_w[K1_BINS + 1];
    __shared__ float s_c[K1_MAXK + 1];
    __shared__ double s_red[16];
    const int j = threadIdx.x;
    const int64_t n = v.n;
    const double lo = (double)v.xs[0], hi = (double)v.xs[n - 1];
    __shared__ unsigned s_last;
    {
        const int mode = mode_mask == 3 ? (int)blockIdx.x : (mode_mask >> 1);   // (one workgroup per set bit; mask 1 -> start A, 2 -> start B)
        // ---- companded start: weight of bin b, exclusive scan, centroid j at the (j + 1/2)/K quantile of the weights
        for (int b = j; b < K1_BINS; b += blockDim.x) {
            double w;
            if (mode == 0) {
                const double e0 = lo + (hi - lo) * ((double)b / K1_BINS), e1 = b + 1 == K1_BINS ? hi : lo + (hi - lo) * ((double)(b + 1) / K1_BINS);
                const int64_t i0 = b == 0 ? 0 : v.upper(e0), i1 = b + 1 == K1_BINS ? n : v.upper(e1);
                w = ::cbrt((double)(i1 - i0));
            } else {
                const double e0 = (double)v.xs[(int64_t)(((__int128)b * (n - 1)) / K1_BINS)];
                const double e1 = (double)v.xs[(int64_t)(((__int128)(b + 1) * (n - 1)) / K1_BINS)];
                const double width = e1 - e0;
                w = ::cbrt(width * width);
            }
            s_w[b + 1] = w;
        }
        if (j == 0) s_w[0] = 0.0;
        __syncthreads();
        if (j == 0)   // 1024 sequential float64 adds: 2 us, once per start; keeps the order fixed
            for (int b = 1; b <= K1_BINS; ++b) s_w[b] += s_w[b - 1];
        __syncthreads();
        const double wtot = s_w[K1_BINS];
        float c = (float)lo;
        if (j < k && wtot > 0.0) {
            const double t = ((double)j + 0.5) / (double)k * wtot;
            int b0 = 0, b1 = K1_BINS;   // last b with s_w[b] <= t
            while (b1 - b0 > 1) {
                const int m = (b0 + b1) >> 1;
                if (s_w[m] <= t) b0 = m; else b1 = m;
            }
            const double wb = s_w[b0 + 1] - s_w[b0];
            const double frac = wb > 0.0 ? (t - s_w[b0]) / wb : 0.5;
            double e0, e1;
            if (mode == 0) {
                e0 = lo + (hi - lo) * ((double)b0 / K1_BINS);
                e1 = b0 + 1 == K1_BINS ? hi : lo + (hi - lo) * ((double)(b0 + 1) / K1_BINS);
            } else {
                e0 = (double)v.xs[(int64_t)(((__int128)b0 * (n - 1)) / K1_BINS)];
                e1 = (double)v.xs[(int64_t)(((__int128)(b0 + 1) * (n - 1)) / K1_BINS)];
            }
            c = (float)(e0 + frac * (e1 - e0));
        }
        // ---- Lloyd iterations: thread j owns the run [b_j, b_{j+1}) of values nearest to centroid j (ties to the lower index)
        double my_inertia = 0.0;
        int64_t p_lo = -1, p_hi = -1;
        bool last = false;
        for (int it = 0; it <= iters; ++it) {
            __syncthreads();
            if (j < k) s_c[j] = c;
            __syncthreads();
            int64_t b_lo = 0, b_hi = n;
            if (j < k) {
                if (j > 0) b_lo = v.upper_from(((double)s_c[j - 1] + (double)c) * 0.5, p_lo);
                if (j + 1 < k) b_hi = v.upper_from(((double)c + (double)s_c[j + 1]) * 0.5, p_hi);
            }
            const int64_t cnt = j < k ? b_hi - b_lo : 0;
            // round 6: a step that moves no run boundary reproduces its centroids, and so would every later one: the remaining
            // iterations are skipped (same result bit for bit; a 50 000-value codebook settles in 20-40 of its 50 steps)
            if (!last && it < iters && __syncthreads_or((int)(b_lo != p_lo || b_hi != p_hi)) == 0) last = true;
            p_lo = b_lo;
            p_hi = b_hi;
            if (it == iters || last) {   // inertia of the final centroids: sum (x - c)^2 = S2 - 2 c S1 + cnt c^2
                if (cnt > 0) {
                    const double S1 = v.s1(b_hi) - v.s1(b_lo), S2 = v.s2(b_hi) - v.s2(b_lo), cd = (double)c;
                    my_inertia = S2 - 2.0 * cd * S1 + (double)cnt * cd * cd;
                }
                break;
            }
            if (cnt > 0) c = (float)((v.s1(b_hi) - v.s1(b_lo)) / (double)cnt);
        }
        const double inertia = k1_block_sum(my_inertia, s_red);
        if (j < k) __hip_atomic_store(&cent_tmp[mode * K1_MAXK + j], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (j == 0) __hip_atomic_store(&inertia_out[1 + mode], inertia, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    __syncthreads();
    if (j == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // the last workgroup to arrive: start A unless start B ran and its inertia is strictly smaller (the sequential rule)
    int win = (mode_mask & 1) ? 0 : 1;
    double best_inertia = __hip_atomic_load(&inertia_out[1 + win], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (mode_mask == 3) {
        const double ib = __hip_atomic_load(&inertia_out[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ib < best_inertia) {
            best_inertia = ib;
            win = 1;
        }
    }
    if (j < k) cent_out[j] = __hip_atomic_load(&cent_tmp[win * K1_MAXK + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (j == 0) {
        inertia_out[0] = best_inertia;
        *ticket = 0u;
    }
}

// label = nearest centroid, ties to the lower index (gpu_ops.py:66-70: strict `<` keeps the first minimum); centroids ascending
__global__ __launch_bounds__(256) void k1_labels_kernel(const float *__restrict__ v, int64_t n, const float *__restrict__ cent, int k,
                                                        int32_t *__restrict__ labels)
{
    __shared__ float s_c[K1_MAXK];
    for (int i = threadIdx.x; i < k; i += 256) s_c[i] = cent[i];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float x = v[i];
        int lo = 0, hi = k;   // first centroid >= x
        while (lo < hi) {
            const int m = (lo + hi) >> 1;
            if (s_c[m] < x) lo = m + 1; else hi = m;
        }
        int best = lo < k ? lo : k - 1;
        if (best > 0) {
            const double dl = (double)x - (double)s_c[best - 1], dr = (double)s_c[best] - (double)x;
            if (dl <= (dr < 0 ? -dr : dr)) --best;
        }
        while (best > 0 && s_c[best - 1] == s_c[best]) --best;   // duplicate centroids: the first one
        labels[i] = best;
    }
}

static int kmeans1d_dev(gsx_ctx *c, const float *vals, int64_t n, int k, int iters, int mode_mask, float *cent_dev, int32_t *labels_dev,
                        double *inertia_host)
{
    const int64_t ntiles = (n + K1_TILE - 1) / K1_TILE;
    size_t temp_bytes = 0;
    unsigned *nul = nullptr;
    if (rocprim::radix_sort_keys(nullptr, temp_bytes, nul, nul, (size_t)n, 0, 32, c->stream) != hipSuccess)
        GSX_FAIL("kmeans1d: rocprim size query failed");
    // layout: keys A | keys B | xs | p1 | p2 | tiles (2 x (ntiles + 1)) | flags + inertia | temp
    const size_t un = sizeof(unsigned) * (size_t)n, dn = sizeof(double) * (size_t)n;
    const size_t bytes = 3 * un + 2 * dn + sizeof(double) * 2 * (size_t)(ntiles + 2) + 256 + temp_bytes + 1024 + sizeof(float) * 2 * K1_MAXK + 256;
    GSX_CHECK(c->scratch5.reserve(bytes));
    char *p = c->scratch5.as<char>();
    auto take = [&](size_t b) {
        char *r = p;
        p += (b + 255) & ~(size_t)255;
        return r;
    };
    unsigned *ka = reinterpret_cast<unsigned *>(take(un)), *kb = reinterpret_cast<unsigned *>(take(un));
    float *xs = reinterpret_cast<float *>(take(un));
    double *p1 = reinterpret_cast<double *>(take(dn)), *p2 = reinterpret_cast<double *>(take(dn));
    double *tiles = reinterpret_cast<double *>(take(sizeof(double) * 2 * (size_t)(ntiles + 2)));
    unsigned *flags = reinterpret_cast<unsigned *>(take(64));     // [0]: non-finite input, [8]: k1_lloyd's arrival ticket
    double *inertia = reinterpret_cast<double *>(take(64));
    float *cent_tmp = reinterpret_cast<float *>(take(sizeof(float) * 2 * K1_MAXK));
    void *temp = take(temp_bytes);
    GSX_HIP(hipMemsetAsync(flags, 0, 64, c->stream));
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 1024), (int64_t)c->num_cu * 8));
    hipLaunchKernelGGL(k1_keys_kernel, dim3(blocks), dim3(256), 0, c->stream, vals, n, ka, flags);
    GSX_HIP(rocprim::radix_sort_keys(temp, temp_bytes, ka, kb, (size_t)n, 0, 32, c->stream));
    hipLaunchKernelGGL(k1_tile_prefix_kernel, dim3((unsigned)ntiles), dim3(256), 0, c->stream, kb, n, xs, p1, p2, tiles);
    hipLaunchKernelGGL(k1_tile_scan_kernel, dim3(1), dim3(256), 0, c->stream, tiles, ntiles);
    K1View v{xs, p1, p2, tiles, n};
    const int threads = std::min(K1_MAXK, ((k + 63) / 64) * 64);
    hipLaunchKernelGGL(k1_lloyd_kernel, dim3(mode_mask == 3 ? 2 : 1), dim3(threads), 0, c->stream, v, k, iters, mode_mask, cent_dev, inertia,
                       cent_tmp, flags + 8);
    if (labels_dev) hipLaunchKernelGGL(k1_labels_kernel, dim3(blocks), dim3(256), 0, c->stream, vals, n, cent_dev, k, labels_dev);
    GSX_HIP(hipGetLastError());
    unsigned hflag = 0;
    double hin[3] = {0, 0, 0};
    GSX_HIP(hipMemcpyAsync(&hflag, flags, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipMemcpyAsync(hin, inertia, sizeof(hin), hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    if (hflag) GSX_FAIL("kmeans1d: values are not finite (NaN/inf)");
    if (inertia_host) memcpy(inertia_host, hin, sizeof(hin));
    return 0;
}

}  // namespace gsx

using namespace gsx;

extern "C" {

int gsx_kmeans1d_dev(gsx_ctx *c, const float *vals_dev, int64_t n, int k, int iters, int start_mask, float *centroids_dev,
                     int32_t *labels_dev, double *inertia3_host)
{
    if (!c || !vals_dev || !centroids_dev) GSX_FAIL("gsx_kmeans1d_dev: null argument");
    if (n <= 0 || n >= (1LL << 31) || k <= 0 || k > K1_MAXK || iters < 0) GSX_FAIL("gsx_kmeans1d_dev: bad shape (1 <= k <= %d)", K1_MAXK);
    if ((start_mask & 3) == 0) start_mask = 3;
    GSX_HIP(hipSetDevice(c->device));
    return kmeans1d_dev(c, vals_dev, n, k, iters, start_mask & 3, centroids_dev, labels_dev, inertia3_host);
}

}  // extern "C"
