// kmeans_pp.hip -- k-means++ seeding on the device (Arthur & Vassilvitskii's D^2 sampling).
//
// The reference's CPU path for gpu_ops.kmeans is scikit-learn's MiniBatchKMeans (processing/gpu_ops.py:48-52), whose
// initial centroids are k-means++ draws; its Taichi path (:178-191) starts from uniformly random rows.  On data with
// separated clusters the two differ by large factors in inertia, so `kmeans(..., use_gpu=False)` -- the call that selects
// the sklearn path in the reference -- gets the same seeding rule here, followed by the full-batch Lloyd iterations of
// kmeans.hip.  Unseeded in the reference (sklearn draws from numpy's global stream): the k uniform numbers are drawn by
// the HOST from that same stream and passed in, so a seeded caller gets a reproducible result.
//
// Step t (t = 1 .. k-1), four launches -- scikit-learn's GREEDY k-means++ (_kmeans_plusplus: n_local_trials = 2 + int(ln k)
// candidates per centroid, the one that lowers the potential most wins; the single-trial rule misses small clusters):
//   kpp_update : mind2[i] = min(mind2[i], |x_i - c_{t-1}|^2) for every row (one thread per row, the row's floats are
//                consecutive in memory and stay in L1 across the dimension loop) + float64 sum of every 1024-row tile;
//   kpp_pick   : one thread per trial: the row where the running sum of mind2 crosses u * total
//                (searchsorted(cumsum(mind2), u * total));
//   kpp_eval   : one pass over the rows: distance to all L candidates at once, potential sum_i min(mind2[i], d_l[i]) per tile;
//   kpp_choose : argmin of the potentials; that candidate's row becomes centroid t.
// HBM/L2-bound: 2 x (4 d + 8) B per row per step; a SOG chunk (156 250 x 45, K = 1024) seeds in ~60 ms.
#include "gsx_common.h"

namespace gsx {

constexpr int KPP_TILE = 1024;

__global__ __launch_bounds__(256) void kpp_update_kernel(const float *__restrict__ data, int64_t n, int d, const float *__restrict__ cent,
                                                         int t /* index of the newest centroid */, float *__restrict__ mind2,
                                                         double *__restrict__ tile_sums)
{
    __shared__ double s_red[4];
    extern __shared__ float s_c[];   // the newest centroid (d floats)
    for (int i = threadIdx.x; i < d; i += 256) s_c[i] = cent[(int64_t)t * d + i];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * KPP_TILE;
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < KPP_TILE / 256; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        if (i < n) {
            const float *row = data + i * d;
            float s = 0.0f;
            for (int j = 0; j < d; ++j) {
                const float df = row[j] - s_c[j];
                s = fmaf(df, df, s);
            }
            const float m = t == 0 ? s : fminf(mind2[i], s);
            mind2[i] = m;
            acc += (double)m;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

constexpr int KPP_MAXL = 12;   // local trials per centroid (sklearn: 2 + int(ln k): 8 at k = 1024)
struct KppU {
    double u[KPP_MAXL];
};

// the row where the running sum of mind2 crosses u * total, for each of the L uniforms (one thread each; sequential float64
// sums in index order: the order searchsorted(cumsum) implies)
__global__ __launch_bounds__(64) void kpp_pick_kernel(int64_t n, const float *__restrict__ mind2, const double *__restrict__ tile_sums,
                                                      int64_t ntiles, KppU u, int L, int64_t *__restrict__ cand)
{
    const int l = threadIdx.x;
    if (l >= L) return;
    double tot = 0.0;
    for (int64_t b = 0; b < ntiles; ++b) tot += tile_sums[b];
    int64_t idx;
    if (!(tot > 0.0)) {
        idx = (int64_t)(u.u[l] * (double)n);          // every row coincides with a centroid already: any row
    } else {
        const double target = u.u[l] * tot;
        double run = 0.0;
        int64_t tile = ntiles - 1;
        for (int64_t b = 0; b < ntiles; ++b) {
            if (run + tile_sums[b] > target) {
                tile = b;
                break;
            }
            run += tile_sums[b];
        }
        const int64_t lo = tile * KPP_TILE, hi = lo + KPP_TILE < n ? lo + KPP_TILE : n;
        idx = hi - 1;
        for (int64_t i = lo; i < hi; ++i) {
            run += (double)mind2[i];
            if (run > target) {
                idx = i;
                break;
            }
        }
    }
    cand[l] = idx < 0 ? 0 : (idx >= n ? n - 1 : idx);
}

// potential of every candidate: sum_i min(mind2[i], |x_i - cand_l|^2), float64 per 1024-row tile -> pot[tile][l]
template <int LMAX>
__global__ __launch_bounds__(256) void kpp_eval_kernel(const float *__restrict__ data, int64_t n, int d, const int64_t *__restrict__ cand,
                                                       int L, const float *__restrict__ mind2, double *__restrict__ pot)
{
    extern __shared__ float s_c[];   // L candidate rows
    __shared__ double s_red[4][LMAX];
    for (int i = threadIdx.x; i < L * d; i += 256) s_c[i] = data[cand[i / d] * d + (i % d)];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * KPP_TILE;
    double acc[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) acc[l] = 0.0;
#pragma unroll
    for (int r = 0; r < KPP_TILE / 256; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        if (i < n) {
            const float *row = data + i * d;
            float s[LMAX];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) s[l] = 0.0f;
            for (int j = 0; j < d; ++j) {
                const float x = row[j];
#pragma unroll
                for (int l = 0; l < LMAX; ++l)
                    if (l < L) {
                        const float df = x - s_c[l * d + j];
                        s[l] = fmaf(df, df, s[l]);
                    }
            }
            const float m = mind2[i];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) acc[l] += (double)fminf(m, s[l]);
        }
    }
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        double v = acc[l];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][l] = v;
    }
    __syncthreads();
    if (threadIdx.x < L) {
        const int l = threadIdx.x;
        pot[(int64_t)blockIdx.x * LMAX + l] = (s_red[0][l] + s_red[1][l]) + (s_red[2][l] + s_red[3][l]);
    }
}

// best candidate = smallest potential (np.argmin: the first minimum); its row becomes centroid t
template <int LMAX>
__global__ __launch_bounds__(64) void kpp_choose_kernel(const float *__restrict__ data, int d, const int64_t *__restrict__ cand, int L,
                                                        const double *__restrict__ pot, int64_t ntiles, int t, float *__restrict__ cent)
{
    __shared__ double s_pot[LMAX];
    __shared__ int s_best;
    if ((int)threadIdx.x < L) {
        double tot = 0.0;
        for (int64_t b = 0; b < ntiles; ++b) tot += pot[b * LMAX + threadIdx.x];
        s_pot[threadIdx.x] = tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        for (int l = 1; l < L; ++l)
            if (s_pot[l] < s_pot[best]) best = l;
        s_best = best;
    }
    __syncthreads();
    const int64_t idx = cand[s_best];
    for (int j = threadIdx.x; j < d; j += 64) cent[(int64_t)t * d + j] = data[idx * d + j];
}

}  // namespace gsx

using namespace gsx;

extern "C" {

int gsx_kmeans_pp_dev(gsx_ctx *c, const float *data_dev, int64_t n, int d, int k, const double *uniforms_host, int n_local_trials,
                      float *centroids_dev)
{
    if (!c || !data_dev || !uniforms_host || !centroids_dev) GSX_FAIL("gsx_kmeans_pp_dev: null argument");
    if (n <= 0 || d <= 0 || k <= 0 || k > n || d > 4096) GSX_FAIL("gsx_kmeans_pp_dev: bad shape");
    int L = n_local_trials;
    if (L < 1 || L > KPP_MAXL) GSX_FAIL("gsx_kmeans_pp_dev: 1 <= n_local_trials <= %d", KPP_MAXL);
    if ((size_t)L * d > 12288) GSX_FAIL("gsx_kmeans_pp_dev: n_local_trials x d must fit 48 KiB of LDS");
    const int64_t nu = 1 + (int64_t)(k - 1) * L;
    for (int64_t t = 0; t < nu; ++t)
        if (!(uniforms_host[t] >= 0.0 && uniforms_host[t] < 1.0)) GSX_FAIL("gsx_kmeans_pp_dev: uniforms must lie in [0, 1)");
    GSX_HIP(hipSetDevice(c->device));
    const int64_t ntiles = (n + KPP_TILE - 1) / KPP_TILE;
    GSX_CHECK(c->scratch3.reserve(sizeof(float) * (size_t)n + sizeof(double) * (size_t)ntiles * (1 + KPP_MAXL) + 1024));
    float *mind2 = c->scratch3.as<float>();
    double *tiles = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(mind2 + n) + 255) & ~(uintptr_t)255);
    double *pot = tiles + ntiles;
    int64_t *cand = reinterpret_cast<int64_t *>(pot + ntiles * KPP_MAXL);
    // centroid 0: a uniformly random row
    int64_t first = (int64_t)(uniforms_host[0] * (double)n);
    if (first >= n) first = n - 1;
    GSX_HIP(hipMemcpyAsync(centroids_dev, data_dev + first * d, sizeof(float) * (size_t)d, hipMemcpyDeviceToDevice, c->stream));
    for (int t = 1; t < k; ++t) {
        hipLaunchKernelGGL(kpp_update_kernel, dim3((unsigned)ntiles), dim3(256), sizeof(float) * (size_t)d, c->stream, data_dev, n, d,
                           centroids_dev, t - 1, mind2, tiles);
        KppU u;
        for (int l = 0; l < KPP_MAXL; ++l) u.u[l] = l < L ? uniforms_host[1 + (int64_t)(t - 1) * L + l] : 0.0;
        hipLaunchKernelGGL(kpp_pick_kernel, dim3(1), dim3(64), 0, c->stream, n, mind2, tiles, ntiles, u, L, cand);
        hipLaunchKernelGGL(kpp_eval_kernel<KPP_MAXL>, dim3((unsigned)ntiles), dim3(256), sizeof(float) * (size_t)L * d, c->stream, data_dev, n, d,
                           cand, L, mind2, pot);
        hipLaunchKernelGGL(kpp_choose_kernel<KPP_MAXL>, dim3(1), dim3(64), 0, c->stream, data_dev, d, cand, L, pot, ntiles, t, centroids_dev);
    }
    GSX_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
