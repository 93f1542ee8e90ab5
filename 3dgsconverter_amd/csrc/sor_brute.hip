// sor_brute.hip -- LDS-tiled brute-force exact KNN mean distance (BASELINE.json configs[1]).
//
// Replaces the hot loop of data_processor.py:160-173 (cKDTree.query(k+1) + row mean) for
// small clouds, and is the last-resort exhaustive pass behind the grid kernels.
//
// Layout: reference points packed as float4 {x,y,z,bits(orig index)} in HBM.  A workgroup
// of 256 lanes owns 256*Q queries (Q per lane, in registers) and streams ALL reference
// points through a double-buffered 2 x 1024-point LDS tile (coalesced 16-B loads, one
// barrier per tile).  Every lane reads the same LDS address (broadcast ds_read_b128), so
// the wave walks the tile in lock step; the per-lane top-(k+1) list is only touched when
// some lane's f32 distance beats its current bound (wave-uniform branch), which becomes
// rare after the first few hundred candidates.
// Bound: FP32 VALU issue (7 lane-ops per pair), not HBM -- see DESIGN.md.
#include "gsx_common.h"
#include "knn_common.h"

namespace gsx {

constexpr int BRUTE_TILE = 1024;
constexpr int BRUTE_THREADS = 256;

__global__ void pack_points_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                   const float *__restrict__ z, int64_t stride, int64_t n,
                                   float4 *__restrict__ out, unsigned *__restrict__ devflags)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (; i < n; i += step) {
        float4 p;
        p.x = x[i * stride];
        p.y = y[i * stride];
        p.z = z[i * stride];
        p.w = __uint_as_float((unsigned)i);
        bad |= !(fabsf(p.x) < __builtin_inff() && fabsf(p.y) < __builtin_inff() && fabsf(p.z) < __builtin_inff());
        out[i] = p;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(devflags, 1u);  // NaN/inf: knn_brute writes NaN, gsx_ctx_check reports
}

template <int KCAP, int Q>
__global__ __launch_bounds__(BRUTE_THREADS) void knn_brute_kernel(
    const float4 *__restrict__ pts, int n_ref, int q_begin, int q_count,
    const unsigned *__restrict__ qlist, const unsigned *__restrict__ qlist_count, int k,
    float *__restrict__ mean_out, const unsigned *__restrict__ devflags)
{
    // SoA tile: one ds_read_b128 returns the same coordinate of 4 consecutive candidates
    // (3 LDS cycles per candidate per wave instead of 8 for a 12-byte AoS read).
    __shared__ __attribute__((aligned(16))) float tx[2][BRUTE_TILE];
    __shared__ __attribute__((aligned(16))) float ty[2][BRUTE_TILE];
    __shared__ __attribute__((aligned(16))) float tz[2][BRUTE_TILE];

    const int nq = qlist ? (int)*qlist_count : q_count;
    const int qbase = blockIdx.x * BRUTE_THREADS * Q;
    if (qbase >= nq) return;  // whole block leaves before any barrier

    const int kk = k + 1;
    float qx[Q], qy[Q], qz[Q], bound[Q];
    double qxd[Q], qyd[Q], qzd[Q];
    int qorig[Q];
    TopList<KCAP> lst[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        int qi = qbase + q * BRUTE_THREADS + (int)threadIdx.x;
        bool live = qi < nq;
        int orig = live ? (qlist ? (int)qlist[qi] : q_begin + qi) : -1;
        qorig[q] = orig;
        float4 p = pts[live ? orig : 0];
        qx[q] = p.x; qy[q] = p.y; qz[q] = p.z;
        qxd[q] = (double)p.x; qyd[q] = (double)p.y; qzd[q] = (double)p.z;
        lst[q].init();
        bound[q] = live ? __builtin_inff() : -1.0f;  // dead lanes never pass the filter
    }

    const int ntiles = (n_ref + BRUTE_TILE - 1) / BRUTE_TILE;
    const float INF = __builtin_inff();
    const int t4 = 4 * (int)threadIdx.x;  // this lane stages points t4..t4+3 of each tile

    float4 sx, sy, sz;  // staged coordinates of 4 consecutive points
    auto stage_load = [&](int tile_idx) {
        float v[3][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            int j = tile_idx * BRUTE_TILE + t4 + s;
            if (j < n_ref) {
                float4 p = pts[j];
                v[0][s] = p.x; v[1][s] = p.y; v[2][s] = p.z;
            } else {
                v[0][s] = INF; v[1][s] = INF; v[2][s] = INF;
            }
        }
        sx = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
        sy = make_float4(v[1][0], v[1][1], v[1][2], v[1][3]);
        sz = make_float4(v[2][0], v[2][1], v[2][2], v[2][3]);
    };
    auto stage_store = [&](int buf) {
        *reinterpret_cast<float4 *>(&tx[buf][t4]) = sx;
        *reinterpret_cast<float4 *>(&ty[buf][t4]) = sy;
        *reinterpret_cast<float4 *>(&tz[buf][t4]) = sz;
    };

    stage_load(0);
    stage_store(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < ntiles;
        if (more) stage_load(t + 1);  // HBM/L2 latency hides under the tile's compute
#pragma unroll 2
        for (int j = 0; j < BRUTE_TILE; j += 4) {
            // uniform addresses: LDS broadcast reads
            const float4 cx = *reinterpret_cast<const float4 *>(&tx[cur][j]);
            const float4 cy = *reinterpret_cast<const float4 *>(&ty[cur][j]);
            const float4 cz = *reinterpret_cast<const float4 *>(&tz[cur][j]);
            const float px[4] = {cx.x, cx.y, cx.z, cx.w};
            const float py[4] = {cy.x, cy.y, cy.z, cy.w};
            const float pz[4] = {cz.x, cz.y, cz.z, cz.w};
            float d2[4][Q];
            bool pass = false;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
#pragma unroll
                for (int u = 0; u < 4; ++u) d2[u][q] = dist2_f32(qx[q], qy[q], qz[q], px[u], py[u], pz[u]);
                float m = fminf(fminf(d2[0][q], d2[1][q]), fminf(d2[2][q], d2[3][q]));
                pass |= m <= bound[q];
            }
            if (__any(pass)) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        if (d2[u][q] <= bound[q]) {
                            double s = dist2_f64(qxd[q], qyd[q], qzd[q], px[u], py[u], pz[u]);
                            lst[q].insert(s);
                            bound[q] = bound_from(lst[q].kth(kk));
                        }
                    }
            }
        }
        if (more) stage_store(cur ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int q = 0; q < Q; ++q) {
        // non-finite input (flagged by pack_points): every distance involving it is meaningless -> NaN, loudly
        if (qorig[q] >= 0) mean_out[qorig[q] - q_begin] = (*devflags & 1u) ? __builtin_nanf("") : mean_from_list<KCAP>(lst[q], k);
    }
}

template <int KCAP, int Q>
static int launch_brute_t(gsx_ctx *ctx, const float4 *pts, int64_t n_ref, int64_t q_begin, int64_t q_count,
                          const unsigned *qlist, const unsigned *qlist_count, int64_t qlist_cap, int k,
                          float *mean_out)
{
    int64_t nq_max = qlist ? qlist_cap : q_count;
    if (nq_max <= 0) return 0;
    int blocks = div_up(nq_max, BRUTE_THREADS * Q);
    hipLaunchKernelGGL((knn_brute_kernel<KCAP, Q>), dim3(blocks), dim3(BRUTE_THREADS), 0, ctx->stream, pts,
                       (int)n_ref, (int)q_begin, (int)q_count, qlist, qlist_count, k, mean_out, ctx->devflags.as<unsigned>());
    GSX_HIP(hipGetLastError());
    return 0;
}

// qlist == nullptr: queries are the index range [q_begin, q_begin+q_count).
// qlist != nullptr: the first *qlist_count entries of qlist (device) are original indices.
int launch_knn_brute(gsx_ctx *ctx, const float4 *pts, int64_t n_ref, int64_t q_begin, int64_t q_count,
                     const unsigned *qlist, const unsigned *qlist_count, int64_t qlist_cap, int k,
                     float *mean_out)
{
    const int kk = k + 1;
    if (kk <= 9) return launch_brute_t<9, 2>(ctx, pts, n_ref, q_begin, q_count, qlist, qlist_count, qlist_cap, k, mean_out);
    if (kk <= 17) return launch_brute_t<17, 2>(ctx, pts, n_ref, q_begin, q_count, qlist, qlist_count, qlist_cap, k, mean_out);
    if (kk <= 33) return launch_brute_t<33, 2>(ctx, pts, n_ref, q_begin, q_count, qlist, qlist_count, qlist_cap, k, mean_out);
    if (kk <= 65) return launch_brute_t<65, 1>(ctx, pts, n_ref, q_begin, q_count, qlist, qlist_count, qlist_cap, k, mean_out);
    GSX_FAIL("sor: k=%d not supported (k must be <= 64)", k);
}

int launch_pack_points(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                       float4 *out)
{
    if (n <= 0) return 0;
    int blocks = (int)std::min<int64_t>(div_up(n, 256), 8192);
    hipLaunchKernelGGL(pack_points_kernel, dim3(blocks), dim3(256), 0, ctx->stream, x, y, z, stride, n, out,
                       ctx->devflags.as<unsigned>());
    GSX_HIP(hipGetLastError());
    return 0;
}

}  // namespace gsx
