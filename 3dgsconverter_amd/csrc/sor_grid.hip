// sor_grid.hip -- grid-binned EXACT k-nearest-neighbour mean distance.
//
// Replaces the hot loop of data_processor.py:160-173 (cKDTree build + query(k+1) + row
// mean) and supersedes the reference's approximate Taichi kernel gpu_ops.py:98-176 and
// its numpy hash-grid prep gpu_ops.py:203-237 (27 cells, K<=50, hash collisions; SURVEY
// F4/F5).  Results are bit-identical to the cKDTree path.
//
// Pipeline (all on one stream, no host round trip):
//   bbox_partial -> grid_params            per-axis min/max, cell edge h for ~m pts/cell
//   bucket_hist / bucket_scan / bucket_scatter / bucket_sort
//                                          two-level counting sort into float4 {x,y,z,orig}: every
//                                          per-point atomic is an LDS atomic (see the kernels)
//   knn_brick                              one WAVE per 2x2x2-cell brick (~56 queries)
//   knn_ring                               expanding-ring exact fallback for the few queries
//                                          whose (k+1)-th neighbour is farther than one cell
//
// knn_brick: the 64 lanes of a wave are 64 queries of one brick (2x2x2 cells for k <= 16, fewer
// cells for larger k).  The brick's neighbourhood (brick + 1 cell each way, 4x4x4 cells) is 16
// x-rows, each a CONTIGUOUS range of the sorted array, which the wave walks in lock step.
// Phase 1 only FILTERS: it builds a per-lane bit mask of the candidates
// whose squared distance is below r_safe^2 (32 candidates per word, words parked in LDS) -- for a
// batch whose words all fit the park with one v_mfma_f32_32x32x16_bf16 per 32 candidates x 32
// queries on bf16-split coordinates (see "MFMA phase-1 filter" below), otherwise with 6 f32 ops per
// candidate on scalar-cache loads and a sign-bit shift-in.  Phase 2 walks each lane's set bits
// (~4.19*m of ~512) with a private cursor, recomputes those distances in float64 exactly as
// cKDTree does and keeps the k+1 smallest in a register-resident sorted list.  A query is exact
// iff its (k+1)-th distance is <= r_safe = h*(1-1e-3): every point outside the searched cells is
// farther.
//
// Clouds the uniform grid cannot resolve (bounding box inflated by far outliers, clusters far
// denser than a cell) are handled by knn_grid_level's adaptive refinement: bricks that would be
// too expensive are deferred to a finer grid built on their neighbourhood, exactly.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "gsx_common.h"
#include "knn_common.h"
#include "knn_mfma.h"
#include "sor_grid_params.h"

namespace gsx {


constexpr int MAX_DIM = 1024;       // cells per axis (keeps the cell-index rounding bound, see r_safe)
constexpr int MAX_BUCKETS = 4096;         // LDS histogram bins of the coarse pass
constexpr int MAX_BUCKET_CELLS = 4096;    // LDS counters of the fine pass (16 KiB)
#ifndef GSX_BUCKET_POINTS
#define GSX_BUCKET_POINTS 4096
#endif
constexpr int BUCKET_POINTS = GSX_BUCKET_POINTS;   // target points per bucket (tuning builds: -DGSX_BUCKET_POINTS, with GSX_SORT_THREADS)
#ifndef GSX_BIN_TILE
#define GSX_BIN_TILE 8192
#endif
constexpr int BIN_TILE = GSX_BIN_TILE;    // points per workgroup tile in the coarse pass (4096 / 16384: no better)
constexpr int BRICK_THREADS = 256;  // 4 independent waves per workgroup
constexpr int HEAVY_RING_CANDIDATES = 1 << 16;
constexpr int HEAVY_CHUNK = 32768;  // points of the sorted array one wave of knn_heavy_scan covers (512 per lane: the
                                    // per-lane top list stops changing after the first few dozen, and the wave merge amortises)
#ifndef GSX_WCAP_BIG   // mask words parked per wave for the lists of more than 32 entries (round 5: 32 -- these kernels run two or three
#define GSX_WCAP_BIG 32   // waves per SIMD, LDS is not what limits them; a brick of more words is filtered by the float32 loop: k = 57 11.2 -> 7.8 ms)
#endif
#ifndef GSX_WCAP_MID   // ... and for 17 ... 32 entries (four waves per SIMD)
#define GSX_WCAP_MID 24
#endif
constexpr int WCAP = 24;            // mask words parked in LDS per wave between drains (6 KiB/wave)

// ---------------------------------------------------------------- bbox + grid params
struct GridParamArgs {   // what grid_params needs besides the partial boxes
    int n;
    double pts_per_cell;
    int cell_cap, debug_skip, share, nshares, defer_words;
    float parent_h;
    GridParams *gp;
    unsigned *devflags;
    double h_hint;   // > 0: cell edge suggested by the density probe of the parent level (never above the bbox-volume edge)
    // multi-GPU slab step (launch_knn_slab): the certificate's planes and counter (axis < 0: off) ...
    int cert_axis;
    float cert_lo, cert_hi;
    unsigned *cert_count;
};
__device__ void grid_params_body(const float *part, int nparts, const GridParamArgs &a);

// The LAST workgroup to arrive (ticket in GridParams) turns the partial boxes into the grid parameters: a separate
// one-wave launch cost ~9 us per step -- 2.5 % of the 1M-splat step.
__global__ __launch_bounds__(256) void bbox_partial_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ z, int64_t stride, int n,
                                                           float *__restrict__ part, GridParamArgs gpa)
{
    __shared__ float red[7][4];
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    float bad = 0.0f;  // fminf/fmaxf drop NaNs silently, so non-finite input is tracked separately
    const int step = gridDim.x * blockDim.x;
    for (int64_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * (int64_t)step) {   // 12 loads in flight per lane
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = (int64_t)(i0 + u * step < n ? i0 + u * step : i0) * stride;   // (a repeated point changes no extremum)
            v[u][0] = x[i];
            v[u][1] = y[i];
            v[u][2] = z[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                mn[a] = fminf(mn[a], v[u][a]);
                mx[a] = fmaxf(mx[a], v[u][a]);
                bad = (fabsf(v[u][a]) < __builtin_inff()) ? bad : 1.0f;
            }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bad = fmaxf(bad, __shfl_xor(bad, off));
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            red[a][w] = mn[a];
            red[3 + a][w] = mx[a];
        }
        red[6][w] = bad;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        float v = red[threadIdx.x][0];
        for (int i = 1; i < 4; ++i) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][i]) : fmaxf(v, red[threadIdx.x][i]);
        __hip_atomic_store(&part[blockIdx.x * 7 + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
    }
    __shared__ unsigned s_last;
    __builtin_amdgcn_s_waitcnt(0);   // the write-through stores above have completed before the ticket (no L2 write-back needed)
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&gpa.gp->ticket_bbox, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) gpa.gp->ticket_bbox = 0;
    if (threadIdx.x < 64) grid_params_body(part, (int)gridDim.x, gpa);
}

__device__ void grid_params_body(const float *part, int nparts, const GridParamArgs &a)
{
    const int n = a.n, cell_cap = a.cell_cap, debug_skip = a.debug_skip, share = a.share, nshares = a.nshares,
              defer_words = a.defer_words;
    const double pts_per_cell = a.pts_per_cell;
    const float parent_h = a.parent_h;
    GridParams *gp = a.gp;
    unsigned *devflags = a.devflags;
    const int lane = threadIdx.x;
    // all 7 x ceil(nparts/64) loads are independent: issue them together (reducing one value at a time
    // serialised 7 round trips and made this one-wave kernel 9 us, 2 % of the 1M-splat step)
    float v[7] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), -__builtin_inff(), -__builtin_inff(),
                  -__builtin_inff(), -__builtin_inff()};
    // (written by other workgroups: device-scope loads, which the compiler keeps in program order -- so eight
    // partial boxes are requested back to back before the first one is consumed: 2 round trips for 1024 parts, not 16)
    for (int i0 = lane; i0 < nparts; i0 += 64 * 8) {
        float t[8][7];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 64 * u < nparts ? i0 + 64 * u : i0;
#pragma unroll
            for (int a = 0; a < 7; ++a) t[u][a] = __hip_atomic_load(&part[i * 7 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < 7; ++a) v[a] = a < 3 ? fminf(v[a], t[u][a]) : fmaxf(v[a], t[u][a]);
    }
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(v[a], off);
            v[a] = a < 3 ? fminf(v[a], o) : fmaxf(v[a], o);
        }
    if (lane != 0) return;
    double e[3] = {(double)v[3] - (double)v[0], (double)v[4] - (double)v[1], (double)v[5] - (double)v[2]};
    double vol = 1.0, emax = 0.0;
    int nd = 0;
    for (int a = 0; a < 3; ++a) {
        if (e[a] > 0.0) { vol *= e[a]; ++nd; }
        emax = e[a] > emax ? e[a] : emax;
    }
    double h;
    if (nd == 0 || !(emax < 1e300)) {
        h = 1.0;
    } else {
        double per = vol * pts_per_cell / (double)(n > 0 ? n : 1);
        h = nd == 3 ? cbrt(per) : (nd == 2 ? sqrt(per) : per);
        if (a.h_hint > 0.0 && a.h_hint < h) h = a.h_hint;   // any edge is valid (exactness does not depend on it): section 5.5
        double hmin = emax / (double)(MAX_DIM - 1);
        if (!(h > hmin)) h = hmin;
    }
    int nx, ny, nz;
    int bg = 1, bny = 1, bnz = 1;
    long long padded = 0;
    float inv_h;
    for (int it = 0; it < 256; ++it) {
        inv_h = (float)(1.0 / h);
        // dims from the SAME f32 arithmetic the kernels use, so no point ever indexes past the grid
        nx = (int)((v[3] - v[0]) * inv_h) + 1;
        ny = (int)((v[4] - v[1]) * inv_h) + 1;
        nz = (int)((v[5] - v[2]) * inv_h) + 1;
        bool ok = nx <= MAX_DIM && ny <= MAX_DIM && nz <= MAX_DIM && nx > 0 && ny > 0 && nz > 0;
        if (ok) {
            // bucket = bg x bg x-rows, ~BUCKET_POINTS points; at most MAX_BUCKETS buckets, MAX_BUCKET_CELLS cells each
            double per_row = (double)nx * pts_per_cell;
            bg = (int)(sqrt((double)BUCKET_POINTS / (per_row > 1.0 ? per_row : 1.0)) + 0.5);
            if (bg < 1) bg = 1;
            if (bg > (ny > nz ? ny : nz)) bg = ny > nz ? ny : nz;  // never wider than the grid itself
            while (bg > 1 && (long long)bg * bg * nx > MAX_BUCKET_CELLS) --bg;
            for (;;) {
                bny = (ny + bg - 1) / bg;
                bnz = (nz + bg - 1) / bg;
                if ((long long)bny * bnz <= MAX_BUCKETS) break;
                ++bg;
            }
            padded = (long long)bny * bnz * bg * bg * nx;
            ok = (long long)bg * bg * nx <= MAX_BUCKET_CELLS && padded <= (long long)cell_cap;
        }
        if (ok) break;
        h *= 1.1;
    }
    bool bad = v[6] > 0.0f;  // NaN or inf anywhere in the cloud
    for (int a = 0; a < 6; ++a) bad |= !(fabsf(v[a]) < 3.0e38f);
    if (bad || !(nx > 0 && ny > 0 && nz > 0) || padded <= 0 || padded > (long long)cell_cap ||
        (long long)bg * bg * nx > MAX_BUCKET_CELLS) {
        // no grid can be built; an exhaustive search would be O(N^2): refuse instead (host raises)
        bad = true;
        nx = ny = nz = 1;
        bg = bny = bnz = 1;
        inv_h = 0.0f;
    }
    gp->ox = v[0]; gp->oy = v[1]; gp->oz = v[2];
    gp->inv_h = inv_h;
    gp->h = (float)h;
    gp->nx = nx; gp->ny = ny; gp->nz = nz;
    gp->bk_g = bg; gp->bk_ny = bny; gp->bk_nz = bnz;
    gp->bk_count = bny * bnz;
    gp->bk_cells = bg * bg * nx;
    gp->ncells = gp->bk_count * gp->bk_cells;
    // brick = the cells one wave owns.  Its ~pts_per_cell * cells queries should fill 64 lanes in ONE
    // batch (a second batch repeats the whole neighbourhood scan): 2x2x2 cells up to ~8 pts/cell
    // (k <= 16), then 2x2x1, 2x1x1, 1x1x1 as k -- hence the cell population -- grows.
    int bdx = 2, bdy = 2, bdz = 2;
    if (pts_per_cell * 8.0 > 66.0) bdz = 1;
    if (pts_per_cell * 4.0 > 66.0) bdy = 1;
    if (pts_per_cell * 2.0 > 66.0) bdx = 1;
    gp->bdx = bdx; gp->bdy = bdy; gp->bdz = bdz;
    gp->nbx = (nx + bdx - 1) / bdx; gp->nby = (ny + bdy - 1) / bdy; gp->nbz = (nz + bdz - 1) / bdz;
    gp->nbricks = bad ? 0 : gp->nbx * gp->nby * gp->nbz;
    // bricks are numbered z-major, so a contiguous share is a slab of the cloud
    gp->part_lo = (int)(((long long)gp->nbricks * share) / nshares);
    gp->part_hi = (int)(((long long)gp->nbricks * (share + 1)) / nshares);
    gp->bad_input = bad ? 1u : 0u;
    if (bad) atomicOr(devflags, 1u);  // reported by gsx_ctx_check; knn_ring fills the output with NaN
    gp->debug_skip = debug_skip;
    // r_safe: |p-q| <= H*h'*(1-1e-3) implies the cell coordinates differ by <= H per axis: the
    // f32 cell index floor(fl(fl(x-o)*inv_h)) is monotone and off by < dim*2^-22 <= 2.5e-4 cells.
    double hp = inv_h > 0.0f ? 1.0 / (double)inv_h : 0.0;
    double r1 = hp * (1.0 - 1e-3);
    gp->hprime = hp;
    gp->r1sq = r1 * r1;
    gp->tau1 = bound_from(r1 * r1);
    // refinement only pays while the cells keep shrinking (a cloud of identical points never would)
    gp->defer_words = (parent_h > 0.0f && !((float)h < 0.8f * parent_h)) ? 0 : defer_words;
    gp->fail_count = 0;
    gp->exhaustive_count = 0;
    gp->extra_count = 0;
    gp->heavy_limit = defer_words > 0 ? HEAVY_RING_CANDIDATES : 0;  // both need the host in the loop
    gp->heavy_count = 0;
    gp->ring2_count = 0;
    for (int a = 0; a < 3; ++a) {
        gp->qb_lo[a] = __builtin_inff();
        gp->qb_hi[a] = -__builtin_inff();
    }
    gp->deferred_count = 0;
    gp->sub_count = 0;
    gp->sub_queries = 0;
    gp->refined_count = 0;
    gp->cert_axis = a.cert_axis;
    gp->cert_lo = a.cert_lo;
    gp->cert_hi = a.cert_hi;
    gp->cert_count = a.cert_count;
    for (int i = 0; i < 8; ++i) {
        gp->brick_ctr[i * 32] = 0;
        gp->extra_ctr[i * 32] = 0;
        gp->ring_ctr[i * 32] = 0;
        gp->ringf_ctr[i * 32] = 0;
    }

}

// The box is known (multi-GPU slab step: the all-reduced global box, cut to the slab's bins along the partition axis):
// no pass over the rows, one wave writes the grid parameters.
__global__ __launch_bounds__(64) void grid_params_known_box_kernel(KnownBox kb, float *__restrict__ part, GridParamArgs gpa)
{
    if (threadIdx.x < 7) {
        const int a = threadIdx.x;
        float v = a < 3 ? -kb.b7[a] : kb.b7[a];          // minima | maxima | non-finite flag
        if (a == kb.axis) v = fmaxf(v, kb.lo);
        if (a == 3 + kb.axis) v = fminf(v, kb.hi);
        __hip_atomic_store(&part[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    grid_params_body(part, 1, gpa);
}

__device__ __forceinline__ int cell_coord(float v, float o, float inv_h, int dim)
{
    int c = (int)((v - o) * inv_h);
    return min(max(c, 0), dim - 1);
}

// Cell index space is BUCKET-MAJOR: all cells of bucket (by, bz) are contiguous, inside a bucket the
// x-rows follow each other, inside a row the cells run along x.  Every x-row of cells is therefore
// still one contiguous range of the sorted array (what knn_brick / knn_ring walk), and a bucket's
// cells can be laid out by the workgroup that sorts the bucket without any global scan.
__device__ __forceinline__ int row_base(const GridParams &g, int cy, int cz)
{
    const int by = cy / g.bk_g, bz = cz / g.bk_g;
    return (bz * g.bk_ny + by) * g.bk_cells + ((cz - bz * g.bk_g) * g.bk_g + (cy - by * g.bk_g)) * g.nx;
}

__device__ __forceinline__ int bucket_of(const GridParams &g, int cy, int cz)
{
    return (cz / g.bk_g) * g.bk_ny + cy / g.bk_g;
}

// ---------------------------------------------------------------- two-level counting sort by cell
// The points arrive in arbitrary order.  A one-level counting sort needs one returning device-scope
// atomic and one random 16-byte write per point (measured 0.69 ms per 10M points: the L2 atomic
// units, not HBM, are the limit).  Two levels keep every per-point atomic in LDS:
//   A1 bucket_hist    tile of 16k points -> LDS histogram over <= 4096 buckets -> one global
//                     atomic per (tile, non-empty bucket)
//   A0 bucket_scan    exclusive scan of the bucket sizes (one workgroup)
//   A2 bucket_scatter same tiles: reserve a run per (tile, bucket) with ONE returning atomic, rank
//                     inside the run with LDS atomics, write float4 {x,y,z,orig} runs
//   B  bucket_sort    one workgroup per bucket: LDS histogram over the bucket's cells, LDS scan ->
//                     cell_start (bucket base + local prefix, no global scan), LDS cursors -> final
//                     position; reads are contiguous, writes stay inside the bucket's ~64 KiB window.
__device__ void bucket_scan_body(int nb, unsigned *bk_cnt, unsigned *bk_start, unsigned *bk_cursor);

__global__ __launch_bounds__(256) void bucket_hist_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                          const float *__restrict__ z, int64_t stride, int first, int n,
                                                          GridParams *__restrict__ gp,
                                                          unsigned *__restrict__ bk_cnt, unsigned *__restrict__ bk_start,
                                                          unsigned *__restrict__ bk_cursor)
{
    __shared__ unsigned hist[MAX_BUCKETS];
    const GridParams g = *gp;
    if (g.bad_input) return;
    const int ntiles = (n + BIN_TILE - 1) / BIN_TILE;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        for (int i = threadIdx.x; i < g.bk_count; i += 256) hist[i] = 0;
        __syncthreads();
        const int lo = t * BIN_TILE, hi = min(n, lo + BIN_TILE);
        for (int i0 = lo + threadIdx.x; i0 < hi; i0 += 2048) {   // 16 independent loads in flight per lane
            float py[8], pz[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + 256 * u, hi - 1);
                const int64_t s = (int64_t)(first + i) * stride;
                py[u] = y[s];
                pz[u] = z[s];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + 256 * u < hi)
                    atomicAdd(&hist[bucket_of(g, cell_coord(py[u], g.oy, g.inv_h, g.ny), cell_coord(pz[u], g.oz, g.inv_h, g.nz))], 1u);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < g.bk_count; i += 256) {
            const unsigned c = hist[i];
            if (c) atomicAdd(&bk_cnt[i], c);
        }
        __syncthreads();
    }
    // the last workgroup to arrive scans the bucket sizes (formerly a one-workgroup launch of its own)
    __shared__ unsigned s_last;
    __builtin_amdgcn_s_waitcnt(0);   // this workgroup's device-scope atomics have completed
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&gp->ticket_hist, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) gp->ticket_hist = 0;
    bucket_scan_body(g.bk_count, bk_cnt, bk_start, bk_cursor);
}

__device__ __forceinline__ unsigned block_exclusive_scan_256(unsigned v, unsigned *total, unsigned *wsum /*[4]*/)
{
    // inclusive scan inside the wave
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned o = __shfl_up(inc, off);
        if ((int)(threadIdx.x & 63) >= off) inc += o;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wsum[w] = inc;
    __syncthreads();
    unsigned base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned s = wsum[i];
        if (i < w) base += s;
        tot += s;
    }
    *total = tot;
    __syncthreads();
    return base + inc - v;
}

// bk_start[0..bk_count] = exclusive scan of bk_cnt; bk_cursor = copy of bk_start; bk_cnt re-zeroed for the next call.
// Run by the 256 threads of bucket_hist's last workgroup; bk_cnt was accumulated with device-scope atomics (L2).
__device__ void bucket_scan_body(int nb, unsigned *bk_cnt, unsigned *bk_start, unsigned *bk_cursor)
{
    __shared__ unsigned wsum[4];
    static_assert(MAX_BUCKETS <= 256 * 16, "bucket_scan_body preloads 16 values per thread");
    unsigned vals[16];   // every bucket size requested before the first scan round (one round trip, not nb/256)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int i = 256 * u + (int)threadIdx.x;
        vals[u] = i < nb ? __hip_atomic_load(&bk_cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    unsigned carry = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int b = 256 * u;
        if (b >= nb) break;
        const int i = b + threadIdx.x;
        const unsigned v = vals[u];
        unsigned tot;
        const unsigned ex = block_exclusive_scan_256(v, &tot, wsum);
        if (i < nb) {
            bk_start[i] = carry + ex;
            bk_cursor[i] = carry + ex;
            bk_cnt[i] = 0;
        }
        carry += tot;
    }
    if (threadIdx.x == 0) bk_start[nb] = carry;
}

#ifndef GSX_SCATTER_THREADS
#define GSX_SCATTER_THREADS 1024
#endif
constexpr int SCATTER_THREADS = GSX_SCATTER_THREADS;
constexpr int SCATTER_PPT = BIN_TILE / SCATTER_THREADS;  // points per thread, kept in registers

// One 8192-point tile per workgroup of 1024 threads, every point read ONCE: coordinates, bucket and the rank the
// returning LDS atomic hands out stay in registers across the two barriers (count -> reserve the tile's run per bucket
// with one global atomic -> write).  (Round 1 swept the tile twice with 256 threads: 0.21 ms per 10M points.)
__global__ __launch_bounds__(SCATTER_THREADS) void bucket_scatter_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                             const float *__restrict__ z, int64_t stride, int first, int n,
                                                             const GridParams *__restrict__ gp,
                                                             unsigned *__restrict__ bk_cursor, float4 *__restrict__ out,
                                                             int ref_only_from)
{
    __shared__ unsigned hist[MAX_BUCKETS];   // per-bucket count of this tile
    __shared__ unsigned base[MAX_BUCKETS];   // start of this tile's run inside the bucket's region
    const GridParams g = *gp;
    if (g.bad_input) return;
    const int ntiles = (n + BIN_TILE - 1) / BIN_TILE;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        for (int i = threadIdx.x; i < g.bk_count; i += SCATTER_THREADS) hist[i] = 0;
        __syncthreads();
        const int lo = t * BIN_TILE, hi = min(n, lo + BIN_TILE);
        float px[SCATTER_PPT], py[SCATTER_PPT], pz[SCATTER_PPT];
        int bk[SCATTER_PPT];
        unsigned rk[SCATTER_PPT];
#pragma unroll
        for (int u = 0; u < SCATTER_PPT; ++u) {   // all loads of the tile in flight
            const int i = min(lo + u * SCATTER_THREADS + (int)threadIdx.x, hi - 1);
            const int64_t sidx = (int64_t)(first + i) * stride;
            px[u] = x[sidx];
            py[u] = y[sidx];
            pz[u] = z[sidx];
        }
#pragma unroll
        for (int u = 0; u < SCATTER_PPT; ++u) {
            bk[u] = bucket_of(g, cell_coord(py[u], g.oy, g.inv_h, g.ny), cell_coord(pz[u], g.oz, g.inv_h, g.nz));
            rk[u] = lo + u * SCATTER_THREADS + (int)threadIdx.x < hi ? atomicAdd(&hist[bk[u]], 1u) : 0u;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < g.bk_count; i += SCATTER_THREADS) {
            const unsigned c = hist[i];
            if (c) base[i] = atomicAdd(&bk_cursor[i], c);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SCATTER_PPT; ++u) {
            const int i = lo + u * SCATTER_THREADS + (int)threadIdx.x;
            if (i < hi)
                // points from ref_only_from on are REFERENCE-ONLY (the halo of a multi-GPU slab): bit 31 of
                // the index word keeps their lanes dead in knn_brick
                out[base[bk[u]] + rk[u]] = make_float4(px[u], py[u], pz[u],
                                                       __uint_as_float((unsigned)(first + i) | (i >= ref_only_from ? 0x80000000u : 0u)));
        }
        __syncthreads();
    }
}

#ifndef GSX_SORT_THREADS
#define GSX_SORT_THREADS 512
#endif
#ifndef GSX_SORT_PPT
#define GSX_SORT_PPT 16
#endif
constexpr int SORT_THREADS = GSX_SORT_THREADS;
constexpr int SORT_PPT = GSX_SORT_PPT;                // points per thread kept in registers
constexpr unsigned SORT_CAP = SORT_THREADS * SORT_PPT;  // buckets up to 8192 points are read ONCE

__global__ __launch_bounds__(SORT_THREADS) void bucket_sort_kernel(const GridParams *__restrict__ gp,
                                                          const unsigned *__restrict__ bk_start,
                                                          const float4 *__restrict__ in, float4 *__restrict__ out,
                                                          unsigned *__restrict__ cell_start, unsigned big_limit)
{
    __shared__ unsigned cnt[MAX_BUCKET_CELLS];
    __shared__ unsigned wsum[SORT_THREADS / 64];
    const GridParams g = *gp;
    const int b = blockIdx.x;
    if (g.bad_input || b >= g.bk_count) return;
    const int cells = g.bk_cells;
    const unsigned s0 = bk_start[b], s1 = bk_start[b + 1];
    if (b == g.bk_count - 1 && threadIdx.x == 0) cell_start[(size_t)g.bk_count * cells] = s1;
    if (s1 - s0 > big_limit) return;  // sorted by the multi-workgroup path below
    const int by = b % g.bk_ny, bz = b / g.bk_ny;
    for (int i = threadIdx.x; i < cells; i += SORT_THREADS) cnt[i] = 0;
    __syncthreads();
    auto local_cell = [&](const float4 p) {
        const int cx = cell_coord(p.x, g.ox, g.inv_h, g.nx);
        const int cy = cell_coord(p.y, g.oy, g.inv_h, g.ny);
        const int cz = cell_coord(p.z, g.oz, g.inv_h, g.nz);
        return ((cz - bz * g.bk_g) * g.bk_g + (cy - by * g.bk_g)) * g.nx + cx;
    };
    const bool fits = s1 - s0 <= SORT_CAP;  // block-uniform: the bucket's points, cells and ranks stay in registers
    float4 p[SORT_PPT];
    int cell[SORT_PPT];
    unsigned rk[SORT_PPT];
    if (fits) {
#pragma unroll
        for (int u = 0; u < SORT_PPT; ++u) p[u] = in[min(s0 + u * SORT_THREADS + threadIdx.x, s1 - 1)];
#pragma unroll
        for (int u = 0; u < SORT_PPT; ++u) {
            cell[u] = local_cell(p[u]);
            rk[u] = s0 + u * SORT_THREADS + threadIdx.x < s1 ? atomicAdd(&cnt[cell[u]], 1u) : 0u;
        }
    } else {
        for (unsigned i = s0 + threadIdx.x; i < s1; i += SORT_THREADS) atomicAdd(&cnt[local_cell(in[i])], 1u);
    }
    __syncthreads();
    // exclusive scan over the bucket's cells: each thread owns a contiguous segment
    const int seg = (cells + SORT_THREADS - 1) / SORT_THREADS;
    const int c0 = min(cells, (int)threadIdx.x * seg), c1 = min(cells, c0 + seg);
    unsigned sum = 0;
    for (int c = c0; c < c1; ++c) sum += cnt[c];
    // inclusive scan inside the wave, then across the SORT_THREADS / 64 waves
    unsigned inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned o = __shfl_up(inc, off);
        if ((int)(threadIdx.x & 63) >= off) inc += o;
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned run = inc - sum;
    for (int i = 0; i < wv; ++i) run += wsum[i];
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
        const unsigned v = cnt[c];
        cnt[c] = run;   // becomes the start (fits) / the cursor (two-pass) of the cell
        run += v;
    }
    __syncthreads();
    unsigned *cs = cell_start + (size_t)b * cells;
    for (int i = threadIdx.x; i < cells; i += SORT_THREADS) cs[i] = s0 + cnt[i];
    if (fits) {
#pragma unroll
        for (int u = 0; u < SORT_PPT; ++u)
            if (s0 + u * SORT_THREADS + threadIdx.x < s1) out[s0 + cnt[cell[u]] + rk[u]] = p[u];
    } else {
        __syncthreads();
        for (unsigned i = s0 + threadIdx.x; i < s1; i += SORT_THREADS) {
            const float4 q = in[i];
            out[s0 + atomicAdd(&cnt[local_cell(q)], 1u)] = q;
        }
    }
}

// ---- oversized buckets (adaptive mode only) -------------------------------------------------------
// bucket_sort is one workgroup per bucket, sized for ~4096 points.  When far outliers inflate the
// bounding box the whole scene lands in a handful of cells, i.e. ONE bucket holds millions of points
// and that workgroup alone took 4.5 of the 7.5 ms of a refined 1M-splat run (40 of 53 ms at 10M).
// Buckets above BIG_BUCKET points are instead sorted by all workgroups: every 8192-point chunk of the
// bucket-grouped array counts its points per cell into the (zeroed) cell_start entries of the
// bucket, one workgroup per big bucket turns the counts into cell starts + global cursors, and the
// chunks scatter into runs reserved with one returning atomic per (chunk, non-empty cell).
constexpr unsigned BIG_BUCKET = 32768;
constexpr int BIG_CHUNK = 8192;

// first bucket whose range contains position pos: largest b with bk_start[b] <= pos
__device__ __forceinline__ int bucket_containing(const unsigned *bk_start, int nb, unsigned pos)
{
    int lo = 0, hi = nb;  // bk_start[nb] = n
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (bk_start[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void big_bucket_kernel(const GridParams *__restrict__ gp, const unsigned *__restrict__ bk_start,
                                                         int n, const float4 *__restrict__ in, float4 *__restrict__ out,
                                                         unsigned *cell_start /* counts (pass 1) */,
                                                         unsigned *cursor /* global cursors (pass 2) */)
{
    __shared__ unsigned cnt[MAX_BUCKET_CELLS];
    __shared__ unsigned base[SCATTER ? MAX_BUCKET_CELLS : 1];
    const GridParams g = *gp;
    if (g.bad_input) return;
    const unsigned lo = (unsigned)blockIdx.x * BIG_CHUNK;
    if (lo >= (unsigned)n) return;
    const unsigned hi = min((unsigned)n, lo + BIG_CHUNK);
    const int cells = g.bk_cells;
    for (int b = bucket_containing(bk_start, g.bk_count, lo); b < g.bk_count; ++b) {  // block-uniform
        const unsigned s0 = bk_start[b], s1 = bk_start[b + 1];
        if (s0 >= hi) break;
        if (s1 - s0 <= BIG_BUCKET) continue;
        const unsigned p0 = max(lo, s0), p1 = min(hi, s1);
        const int by = b % g.bk_ny, bz = b / g.bk_ny;
        auto local_cell = [&](const float4 p) {
            const int cx = cell_coord(p.x, g.ox, g.inv_h, g.nx);
            const int cy = cell_coord(p.y, g.oy, g.inv_h, g.ny);
            const int cz = cell_coord(p.z, g.oz, g.inv_h, g.nz);
            return ((cz - bz * g.bk_g) * g.bk_g + (cy - by * g.bk_g)) * g.nx + cx;
        };
        for (int i = threadIdx.x; i < cells; i += 256) cnt[i] = 0;
        __syncthreads();
        for (unsigned i = p0 + threadIdx.x; i < p1; i += 256) atomicAdd(&cnt[local_cell(in[i])], 1u);
        __syncthreads();
        unsigned *gc = (SCATTER ? cursor : cell_start) + (size_t)b * cells;
        for (int i = threadIdx.x; i < cells; i += 256) {
            const unsigned c = cnt[i];
            if (c) {
                const unsigned r = atomicAdd(&gc[i], c);
                if (SCATTER) base[i] = r;
            }
            if (SCATTER) cnt[i] = 0;
        }
        __syncthreads();
        if (SCATTER) {
            for (unsigned i = p0 + threadIdx.x; i < p1; i += 256) {
                const float4 p = in[i];
                const int c = local_cell(p);
                out[base[c] + atomicAdd(&cnt[c], 1u)] = p;
            }
            __syncthreads();
        }
    }
}

// one workgroup per big bucket: counts in cell_start -> cell starts; cursors = copies
__global__ __launch_bounds__(256) void big_bucket_scan_kernel(const GridParams *__restrict__ gp,
                                                              const unsigned *__restrict__ bk_start,
                                                              unsigned *cell_start, unsigned *cursor)
{
    __shared__ unsigned wsum[4];
    const GridParams g = *gp;
    const int b = blockIdx.x;
    if (g.bad_input || b >= g.bk_count) return;
    const unsigned s0 = bk_start[b], s1 = bk_start[b + 1];
    if (s1 - s0 <= BIG_BUCKET) return;
    const int cells = g.bk_cells;
    unsigned *cs = cell_start + (size_t)b * cells, *cu = cursor + (size_t)b * cells;
    const int seg = (cells + 255) / 256;
    const int c0 = min(cells, (int)threadIdx.x * seg), c1 = min(cells, c0 + seg);
    unsigned sum = 0;
    for (int c = c0; c < c1; ++c) sum += cs[c];
    unsigned tot;
    unsigned run = s0 + block_exclusive_scan_256(sum, &tot, wsum);
    for (int c = c0; c < c1; ++c) {
        const unsigned v = cs[c];
        cs[c] = run;
        cu[c] = run;
        run += v;
    }
}

// ---------------------------------------------------------------- knn_brick
// One wave per brick.  refs/rstart: cell-sorted reference points and cell starts;
// qpts/qstart: cell-sorted QUERY points (same arrays when every reference is a query).
// Work item = (brick, batch of 64 queries).  EXTRA == false: one item per brick, its first batch; any
// further batch (a brick holding more than 64 queries: ~15-45 % of bricks on uniform data, thousands
// of batches for one brick inside a dense cluster) is APPENDED to a list instead of being looped over
// by the same wave.  EXTRA == true (second launch) spreads those items over the whole chip.
// waves per SIMD the register allocator must leave room for (the top-k list is 2*KCAP VGPRs)
// Round 3 (profiles/r03_variants.txt): phase 2 takes the candidates in blocks of 8 (TopNet::BS) -- the list, one block and
// four gathers in flight then need ~95 VGPRs and five waves per SIMD are resident for k <= 16 (four for k <= 32: round 4, -3 % at k = 25 / 32; five spill into the loops: +70 %).  The
// kernel is bound by latency as much as by issue: 3 -> 5 waves took knn_brick from 1.92 to 1.68 ms at 10M splats, the
// setup code's spills (outside the loops) notwithstanding; six waves push spills into the loops (2.39 ms).
// (tuning builds override these constants with -D; tools/build_variants.sh)
#ifndef GSX_NET_WAVES17
#define GSX_NET_WAVES17 5
#endif
#ifndef GSX_BRICK_UNROLL3   // phase 1's word pipeline with rotating names (round 5, A/B)
#define GSX_BRICK_UNROLL3 0
#endif
#ifndef GSX_NET_WAVES33
#define GSX_NET_WAVES33 4
#endif
#ifndef GSX_NET_HB
#define GSX_NET_HB 4
#endif
#ifndef GSX_NET_WAVES25
#define GSX_NET_WAVES25 4
#endif
#ifndef GSX_NET_WAVES49
#define GSX_NET_WAVES49 3
#endif
constexpr int NET_WAVES9 = 5, NET_WAVES17 = GSX_NET_WAVES17, NET_WAVES25 = GSX_NET_WAVES25, NET_WAVES33 = GSX_NET_WAVES33,
              NET_WAVES49 = GSX_NET_WAVES49;
constexpr int NET_HB = GSX_NET_HB;   // candidate gathers in flight per lane (8: no change)
constexpr int brick_min_waves(int kcap, bool mf, bool net)
{
    (void)mf;  // the MFMA filter's registers are not live together with the top-k list (single-drain path)
    // NET: list of KCAP-1 doubles + a 16-candidate block + 8 gathers in flight
    if (net) return kcap <= 9 ? NET_WAVES9 : (kcap <= 17 ? NET_WAVES17 : (kcap <= 25 ? NET_WAVES25 : (kcap <= 33 ? NET_WAVES33 : (kcap <= 49 ? NET_WAVES49 : 2))));   // (41 -> like 49, 57 -> like 65)
    return kcap <= 17 ? 5 : (kcap <= 26 ? 4 : (kcap <= 33 ? 3 : 2));
}

// NET: phase 2 selects with TopNet<KCAP-1> (sorting-network blocks, the query excluded by index) instead
// of the per-candidate bubble insert of TopList<KCAP>; needs KCAP-1 to be a power of two.
template <int KCAP, bool EXTRA, bool MF, bool NET>
__global__ __launch_bounds__(BRICK_THREADS, brick_min_waves(KCAP, MF, NET)) void knn_brick_kernel(
    GridParams *__restrict__ gp, const float4 *__restrict__ refs, const unsigned *__restrict__ rstart,
    const float4 *__restrict__ qpts, const unsigned *__restrict__ qstart, int k, int q_begin,
    float *__restrict__ mean_out, unsigned *__restrict__ faillist, uint2 *__restrict__ extra,
    unsigned *__restrict__ deferred, double *__restrict__ kth_out)
{
    constexpr int WCAP = KCAP > 33 ? GSX_WCAP_BIG : (KCAP > 17 ? GSX_WCAP_MID : gsx::WCAP);
    __shared__ unsigned s_mask[BRICK_THREADS / 64][WCAP][64];
    // per mask word: first candidate (index into refs); (c1, b2 - c1): candidates of the first row segment and the
    // offset of the second (MFMA words may span two rows).  Two narrow arrays: one 16-byte entry per word read
    // with per-lane indices cost 0.14 ms at 10M splats (LDS bank conflicts)
    __shared__ unsigned s_wbase[BRICK_THREADS / 64][WCAP];
    __shared__ uint2 s_wseg[MF ? BRICK_THREADS / 64 : 1][MF ? WCAP : 1];

    if (EXTRA && gp->extra_count == 0) return;  // the usual case: no brick shed a batch (an empty pass still cost 6.6 us)
    const int lane = lane_id();
    const int wv = uniform((int)(threadIdx.x >> 6));
    unsigned(*mask)[64] = s_mask[wv];
    unsigned *wbase = s_wbase[wv];
    uint2 *wseg = s_wseg[MF ? wv : 0];

    const int nx = gp->nx, ny = gp->ny, nz = gp->nz;
    const int nbx = gp->nbx, nby = gp->nby;
    const int bdx = gp->bdx, bdy = gp->bdy, bdz = gp->bdz;
    const int cry = bdy + 2;                  // candidate rows per z-slab
    const int ncrows = cry * (bdz + 2);       // <= 16 candidate x-rows
    const int nqrows = bdy * bdz;             // <= 4 query x-rows
    const double r1sq = gp->r1sq;
    const double hp = gp->hprime;
    const float tau1 = gp->tau1;
    const float g_ox = gp->ox, g_oy = gp->oy, g_oz = gp->oz, g_inv_h = gp->inv_h;
    const int bk_g = gp->bk_g, bk_ny = gp->bk_ny, bk_cells = gp->bk_cells;
    auto row_base_g = [&](int cy, int cz) {  // bucket-major cell index of cell (0, cy, cz)
        const int by_ = cy / bk_g, bz_ = cz / bk_g;
        return (bz_ * bk_ny + by_) * bk_cells + ((cz - bz_ * bk_g) * bk_g + (cy - by_ * bk_g)) * nx;
    };
#ifdef GSX_ABLATE  // profiling build only (results become wrong): -DGSX_ABLATE + gsx_ctx_set_param("debug_skip")
    const int dbg = gp->debug_skip;
#else
    constexpr int dbg = 0;
#endif
    const int kk = k + 1;
    const int kq = NET ? k : kk;  // position of the acceptance distance in the list (NET lists do not hold the query)

    WorkQueue wq;
    const int part_lo = gp->part_lo;
    const int dw = EXTRA ? 0 : gp->defer_words;  // loaded once: gp is written by atomics, the compiler would reload it per brick
    wq_init(wq, EXTRA ? gp->extra_ctr : gp->brick_ctr, EXTRA ? (int)gp->extra_count : gp->part_hi - part_lo,
            BRICK_THREADS / 64);
    for (;;) {
        const int item = uniform(wq_next(wq));
        if (item < 0) break;
        int b = item + part_lo, qb = 0;
        if (EXTRA) {
            const uint2 it = extra[item];
            b = uniform((int)it.x);
            qb = uniform((int)it.y);
        }
        const int bz = b / (nbx * nby);
        const int brem = b - bz * nbx * nby;
        const int by = brem / nbx;
        const int bx = brem - by * nbx;

        // lane r < 16: candidate row r (bdx+2 cells along x); lanes 16..19: query rows (bdx cells)
        int v_start = 0, v_len = 0;
        {
            // (recomputed per brick from a pinned lane id: as loop invariants these five values, the division magic behind
            //  them and a dozen more were kept in scratch across the whole kernel.  bdy is 1 or 2, cry = bdy + 2 is 3 or 4)
            const int ln = pinned_here(lane);
            const bool isq = ln >= 16;
            const int r = isq ? ln - 16 : ln;
            const int r_mod_bdy = bdy == 2 ? (r & 1) : 0, r_div_bdy = bdy == 2 ? (r >> 1) : r;
            const int r_div_cry = cry == 4 ? (r >> 2) : (r * 11) >> 5;       // r < 16
            const int r_mod_cry = r - r_div_cry * cry;
            const int yy = isq ? by * bdy + r_mod_bdy : by * bdy - 1 + r_mod_cry;
            const int zz = isq ? bz * bdz + r_div_bdy : bz * bdz - 1 + r_div_cry;
            const int xa = isq ? bx * bdx : max(bx * bdx - 1, 0);
            const int xb = isq ? min(bx * bdx + bdx - 1, nx - 1) : min(bx * bdx + bdx, nx - 1);
            const bool valid = (isq ? (lane < 20 && r < nqrows) : r < ncrows) && yy >= 0 && yy < ny && zz >= 0 && zz < nz;
            if (valid) {
                const unsigned *st = isq ? qstart : rstart;
                const int row = row_base_g(yy, zz);
                unsigned s = st[row + xa], e = st[row + xb + 1];
                v_start = (int)s;
                v_len = (int)(e - s);
            }
        }
        if (dbg & 16) continue;  // queue + row-table loads only
        int qoff[5];
        qoff[0] = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) qoff[r + 1] = qoff[r] + __builtin_amdgcn_readlane(v_len, 16 + r);
        const int nq = qoff[4];
        if (!EXTRA) {
            // Adaptive refinement: the cost of this brick is batches x candidate words.  A uniform grid
            // gives ~1.5 x 15; a brick inside (or next to) a cluster much denser than the cell size can
            // reach millions.  Such a brick is not searched here: it goes on the deferred list, and the
            // host re-runs the whole pipeline on the points of the deferred neighbourhoods with a grid
            // sized for THEM (launch_knn_grid, refine_level).
            if (dw > 0 && nq > 0) {
                int ncand = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) ncand += __builtin_amdgcn_readlane(v_len, r);
                const int words = (ncand + 31) >> 5, batches = (nq + 63) >> 6;
                if (words > dw || (long long)words * batches > 2LL * dw) {
                    if (lane == 0) deferred[atomicAdd(&gp->deferred_count, 1u)] = (unsigned)b;
                    continue;
                }
            }
        }
        // A wave loops over up to LOCAL_BATCHES batches of its brick itself (the neighbourhood is hot in
        // the scalar cache / L2 then); the batches beyond that -- a brick inside a dense cluster can
        // hold thousands -- become items of the second launch.
        constexpr int LOCAL_MAX = 3;
        const int LOCAL_BATCHES = nq > 64 * LOCAL_MAX ? 1 : LOCAL_MAX;  // heavy brick: keep one, spread the rest
        if (!EXTRA && nq > 64 * LOCAL_BATCHES) {
            const int nextra = (nq - 1) / 64 + 1 - LOCAL_BATCHES;
            unsigned slot = 0;
            if (lane == 0) slot = atomicAdd(&gp->extra_count, (unsigned)nextra);
            slot = (unsigned)uniform((int)slot);
            for (int e = lane; e < nextra; e += 64)
                extra[slot + e] = make_uint2((unsigned)b, (unsigned)(64 * (e + LOCAL_BATCHES)));
        }
        const int qb_end = EXTRA ? min(nq, qb + 64) : min(nq, 64 * LOCAL_BATCHES);

        for (; qb < qb_end; qb += 64) {
            // ---- this lane's query
            const int f = qb + lane;
            const bool live = f < nq;
            int qidx = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int s = __builtin_amdgcn_readlane(v_start, 16 + r);
                if (f >= qoff[r] && f < qoff[r + 1]) qidx = s + (f - qoff[r]);
            }
            const float4 qp = qpts[live ? qidx : 0];
            const float qx = qp.x, qy = qp.y, qz = qp.z;
            const bool is_query = live && !(__float_as_uint(qp.w) >> 31);  // reference-only points (slab halo) are never queries
            if (!__any(is_query)) continue;  // a batch of halo points only
            float tau = is_query ? tau1 : -1.0f;
            double racc_sq = r1sq;  // this lane's acceptance radius^2 (== the exact value its filter bound comes from)
            // Queries next to the cloud's bounding box see only part of their neighbourhood ball, so
            // their (k+1)-th neighbour is farther than one cell and they would all go to knn_ring.  But
            // where the brick's neighbourhood is cut by the GRID boundary nothing exists beyond that
            // face, so it does not limit the guaranteed radius: r_safe(q) = distance to the nearest
            // face of the searched box that has cells behind it (>= h').  The filter radius is widened
            // only as far as the missing ball volume requires (so phase 2 sees the usual ~30 candidates).
            {
                const int ulo[3] = {bx * bdx - 1, by * bdy - 1, bz * bdz - 1};
                const int uhi[3] = {bx * bdx + bdx, by * bdy + bdy, bz * bdz + bdz};
                const int dim[3] = {nx, ny, nz};
                const bool boundary = ulo[0] <= 0 || ulo[1] <= 0 || ulo[2] <= 0 || uhi[0] >= nx - 1 || uhi[1] >= ny - 1 ||
                                      uhi[2] >= nz - 1;
                // Only for a brick's FIRST batch: the later ones scan a sub-box of the neighbourhood (below) that reaches exactly one
                // cell beyond their queries' cells -- a radius above h' would certify against points that were never looked at.
                // (Found by the randomised sweep, round 3: a planar cloud, k = 64, two queries at the rim wrong in ~15 % of the
                // runs -- which queries land in a later batch depends on the order the binning's atomics leave in a cell.)
                if (boundary && qb == 0) {  // wave-uniform
                    // f32 is enough: the 1e-3*h' margin dwarfs its rounding (<= dims * 2^-23 * h' ~ 1e-4 h')
                    // (pinned: 8 % of the bricks get here; hoisted, its per-brick operands were computed and parked in
                    //  scratch for every brick)
                    const float hf = pinned_here((float)hp);
                    // (the faces again, from pinned brick coordinates: their float conversions stay in here too)
                    const int pbx = pinned_here_s(bx), pby = pinned_here_s(by), pbz = pinned_here_s(bz);
                    const int ulo[3] = {pbx * bdx - 1, pby * bdy - 1, pbz * bdz - 1};
                    const int uhi[3] = {pbx * bdx + bdx, pby * bdy + bdy, pbz * bdz + bdz};
                    const float rel[3] = {qx - g_ox, qy - g_oy, qz - g_oz};
                    const float r1 = hf * (1.0f - 1e-3f);
                    float rsafe = 3.0e38f, frac = 1.0f;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        // faces of the searched box with cells behind them (margin as in grid_params_kernel)
                        if (ulo[a] > 0) rsafe = fminf(rsafe, rel[a] - (float)ulo[a] * hf - 2e-3f * hf);
                        if (uhi[a] < dim[a] - 1) rsafe = fminf(rsafe, (float)(uhi[a] + 1) * hf - rel[a] - 2e-3f * hf);
                        // part of [q - r1, q + r1] that lies inside the grid along this axis
                        const float tl = fminf(fmaxf(rel[a], 0.0f), r1);
                        const float th = fminf(fmaxf((float)dim[a] * hf - rel[a], 0.0f), r1);
                        frac *= (tl + th) / (2.0f * r1);
                    }
                    float rq = r1 * cbrtf(1.0f / fmaxf(frac, 0.125f));
                    rq = fminf(rq, rsafe);
                    if (rq > r1 && is_query) {
                        racc_sq = (double)rq * (double)rq;
                        tau = bound_from(racc_sq);
                    }
                }
            }

            // Batches after a brick's first hold the LAST queries of the brick (typically its last cell
            // or two): their neighbourhood is a sub-box of the brick's, so only the rows / cells within
            // one cell of the batch's bounding box are scanned (still a superset of every live query's
            // 3x3x3 cells, so r_safe is unchanged).
            int rs_start = v_start, rs_len = v_len;
            if (qb > 0) {
                int lo[3], hi[3];
                const int c3[3] = {cell_coord(qx, g_ox, g_inv_h, nx), cell_coord(qy, g_oy, g_inv_h, ny),
                                   cell_coord(qz, g_oz, g_inv_h, nz)};
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    lo[a] = is_query ? c3[a] : 0x7fffffff;
                    hi[a] = is_query ? c3[a] : -1;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
                        lo[a] = min(lo[a], __shfl_xor(lo[a], off));
                        hi[a] = max(hi[a], __shfl_xor(hi[a], off));
                    }
                    lo[a] = uniform(lo[a]);
                    hi[a] = uniform(hi[a]);
                }
                rs_start = 0;
                rs_len = 0;
                if (lane < ncrows) {
                    const int ln = pinned_here(lane);
                    const int l_div_cry = cry == 4 ? (ln >> 2) : (ln * 11) >> 5;     // ln < 16
                    const int yy = by * bdy - 1 + (ln - l_div_cry * cry);
                    const int zz = bz * bdz - 1 + l_div_cry;
                    const int xa = max(max(bx * bdx - 1, 0), lo[0] - 1);
                    const int xb = min(min(bx * bdx + bdx, nx - 1), hi[0] + 1);
                    const bool need = yy >= max(lo[1] - 1, 0) && yy <= min(hi[1] + 1, ny - 1) &&
                                      zz >= max(lo[2] - 1, 0) && zz <= min(hi[2] + 1, nz - 1) && xa <= xb;
                    if (need) {
                        const int row = row_base_g(yy, zz);
                        const unsigned s0 = rstart[row + xa], e0 = rstart[row + xb + 1];
                        rs_start = (int)s0;
                        rs_len = (int)(e0 - s0);
                    }
                }
            }
            // initialised where phase 1 needs it live (see the MFMA single-drain path)
            typename std::conditional<NET, TopNet<NET ? KCAP - 1 : 8>, TopList<KCAP>>::type lst;
            bool lst_empty = true;  // wave-uniform: no block merged yet (NET)

            int widx = 0;
            unsigned nzw = 0;

            // ---- phase 2: walk this lane's set bits, exact f64 distance, sorted insert
            auto drain = [&]() __attribute__((always_inline)) {
                // widened here, not before phase 1: six registers that the filter loop does not have to carry
                const double qxd = (double)qx, qyd = (double)qy, qzd = (double)qz;
                wave_sync();  // wbase[] written by lane 0 is visible to every lane
                if (dbg & 32) nzw = 0;  // profiling: masks are built but never walked
                unsigned m = 0;
                int base = 0, c1 = 32, base2 = 0;
                if constexpr (NET) {
                    // Blocks of BS candidates per lane: the lane's next BS set bits are gathered (HB loads in
                    // flight), their exact distances fill a register block (+inf where the lane has run out, and
                    // for the query's own point), one sorting network orders the block and a bitonic merge folds
                    // it into the list.  The wave repeats while any lane has bits left.
                    using Net = TopNet<NET ? KCAP - 1 : 8>;
                    constexpr int BS = Net::BS, HB = NET_HB < BS ? NET_HB : BS;
                    const unsigned self_w = __float_as_uint(qp.w);
                    for (;;) {
                        double blk[BS];
#pragma unroll
                        for (int h0 = 0; h0 < BS; h0 += HB) {
                            float4 pt[HB];
                            bool ok[HB];
#pragma unroll
                            for (int j = 0; j < HB; ++j) {
                                if (m == 0 && nzw != 0) {
                                    const int w = __builtin_ctz(nzw);
                                    nzw &= nzw - 1;
                                    m = mask[w][lane];
                                    base = (int)wbase[w];
                                    if constexpr (MF) {
                                        const uint2 sg = wseg[w];
                                        c1 = (int)sg.x;
                                        base2 = (int)sg.y;
                                    }
                                }
                                ok[j] = m != 0;
                                // branch-free: a lane that has run out (m == 0) re-reads candidate 0 of its last
                                // word (a valid, cached address) and the result is discarded below
                                const int i = ok[j] ? __builtin_clz(m) : 0;
                                m &= ~(0x80000000u >> i);  // m == 0 stays 0
                                pt[j] = refs[(MF && i >= c1 ? base2 : base) + i];
                            }
#pragma unroll
                            for (int j = 0; j < HB; ++j) {
                                double d = dist2_f64(qxd, qyd, qzd, pt[j].x, pt[j].y, pt[j].z);
                                blk[h0 + j] = (ok[j] && __float_as_uint(pt[j].w) != self_w) ? d : __builtin_inf();
                            }
                        }
                        if (uniform((int)lst_empty)) lst.assign_block(blk); else lst.merge_block(blk);
                        lst_empty = false;
                        if (!__any(m != 0 || nzw != 0)) break;
                    }
                } else {
                // software pipeline: the gather of candidate n+1 is in flight while candidate n goes
                // through the float64 distance + the 2*KCAP-op sorted insert
                bool have_cur = false;
                float4 pc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (;;) {
                    if (m == 0 && nzw != 0) {
                        const int w = __builtin_ctz(nzw);
                        nzw &= nzw - 1;
                        m = mask[w][lane];
                        base = (int)wbase[w];
                        if constexpr (MF) {
                            const uint2 sg = wseg[w];
                            c1 = (int)sg.x;
                            base2 = (int)sg.y;
                        }
                    }
                    const bool have_next = m != 0;
                    float4 pn = pc;
                    if (have_next) {
                        const int i = __builtin_clz(m);
                        m &= ~(0x80000000u >> i);
                        pn = refs[(MF && i >= c1 ? base2 : base) + i];
                    }
                    if (have_cur && !(dbg & 1)) lst.insert(dist2_f64(qxd, qyd, qzd, pc.x, pc.y, pc.z));
                    if (!__any(have_next)) break;
                    pc = pn;
                    have_cur = have_next;
                }
                }
                tau = fminf(tau, bound_from(lst.kth(kq)));
                widx = 0;
                nzw = 0;
                wave_sync();  // all reads of mask/wbase done before they are overwritten
            };

            // ---- phase 1: lock-step filter over the 16 candidate rows
            // MFMA variant: only when ALL mask words of the batch fit the LDS park (the usual case).  Then
            // phase 2 runs exactly once, after phase 1, and the 2*KCAP registers of the top-k list are not
            // live while the 16 accumulators and the operands are: both phases fit 5 waves/SIMD without
            // spills (carrying the list through the MFMA loop cost 41-46 spilled dwords and made
            // phase 2 40 % slower).  A batch with more words takes the scalar filter.
            bool mf_done = false;
            if constexpr (MF) {
                // number of words the cutting below will produce (same rule: a word holds at most two row segments)
                int nwords = 0;
                {
                    int fill = 0, segs = 0;
                    for (int r = 0; r < ncrows; ++r) {
                        int len = __builtin_amdgcn_readlane(rs_len, r & 15);
                        while (len > 0) {
                            if (fill == 32 || segs == 2) {
                                ++nwords;
                                fill = 0;
                                segs = 0;
                            }
                            const int take = min(len, 32 - fill);
                            fill += take;
                            len -= take;
                            ++segs;
                        }
                    }
                    nwords += fill > 0;
                }
                if (nwords <= WCAP && !(dbg & 2)) {
                mf_done = true;
                // cell-unit coordinates relative to the brick centre; see the MFMA notes at the top
                // (pinned: computed here, per batch -- nine instructions -- instead of per brick with a round trip through scratch)
                const float hf = pinned_here((float)hp);
                const float ccx = g_ox + ((float)(bx * bdx) + 0.5f * (float)bdx) * hf;
                const float ccy = g_oy + ((float)(by * bdy) + 0.5f * (float)bdy) * hf;
                const float ccz = g_oz + ((float)(bz * bdz) + 0.5f * (float)bdz) * hf;
                const float uqx = (qx - ccx) * g_inv_h, uqy = (qy - ccy) * g_inv_h, uqz = (qz - ccz) * g_inv_h;
                const float nq2 = __builtin_fmaf(uqz, uqz, __builtin_fmaf(uqy, uqy, uqx * uqx));
                const bool upper = lane >= 32;
                // (tau * inv_h) * inv_h: no overflow for tiny cells; dead lanes (tau = -1) never pass
                const float s_q = tau >= 0.0f ? nq2 - ((tau * g_inv_h) * g_inv_h * (1.0f + 1e-6f) + MF_SLACK) : 1.0e30f;
                bf16x8 opa, opb;
                mf_query_operands(uqx, uqy, uqz, s_q, opa, opb);
                // The candidate rows are walked as ONE flat sequence cut into 32-candidate words; a word may
                // continue into the next non-empty row (two segments at most), otherwise the ~25 % of rows
                // with 33..40 candidates would each cost a second, almost empty, MFMA pair.  The points of
                // words n+1 and n+2 are requested before word n is processed (a per-lane 16-B load takes
                // ~1-2k cycles from L2/HBM, one word is ~400 cycles of work).
                const int nrows = ncrows;
                struct Word { int b1, c1, b2, c2, r, off; };  // wave-uniform; r/off = where the NEXT word starts
                auto next_word = [&](int r, int off) __attribute__((always_inline)) {
                    Word o{0, 0, 0, 0, r, off};
                    int len = r < nrows ? __builtin_amdgcn_readlane(rs_len, r & 15) : 0;
                    while (r < nrows && off >= len) {  // skip exhausted / empty rows
                        ++r;
                        off = 0;
                        len = r < nrows ? __builtin_amdgcn_readlane(rs_len, r & 15) : 0;
                    }
                    if (r < nrows) {
                        o.b1 = __builtin_amdgcn_readlane(rs_start, r & 15) + off;
                        o.c1 = min(32, len - off);
                        off += o.c1;
                        if (o.c1 < 32) {  // row finished inside the word: continue with the next non-empty row
                            do {
                                ++r;
                                off = 0;
                                len = r < nrows ? __builtin_amdgcn_readlane(rs_len, r & 15) : 0;
                            } while (r < nrows && len == 0);
                            if (r < nrows) {
                                o.b2 = __builtin_amdgcn_readlane(rs_start, r & 15);
                                o.c2 = min(32 - o.c1, len);
                                off = o.c2;
                            }
                        }
                    }
                    o.r = r;
                    o.off = off;
                    return o;
                };
                const int my_cand = mf_cand_of_row(lane & 31);
                auto fetch = [&](const Word &w) __attribute__((always_inline)) {
                    const int c = w.c1 + w.c2;
                    if (c == 0) return make_float4(0.f, 0.f, 0.f, 0.f);
                    const int t = min(my_cand, c - 1);  // slots past the end repeat the last candidate (masked out below)
                    return refs[t < w.c1 ? w.b1 + t : w.b2 + (t - w.c1)];
                };
                auto process = [&](const Word &w_cur, const float4 &p_cur) __attribute__((always_inline)) {
                    const bf16x8 cand = mf_candidate_operand((p_cur.x - ccx) * g_inv_h, (p_cur.y - ccy) * g_inv_h,
                                                             (p_cur.z - ccz) * g_inv_h, upper);
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    // lane l: tile a = rows-set (l>>5) of query l&31, tile b = of query 32 + (l&31)
                    const unsigned ma = mf_sign_bits(__builtin_amdgcn_mfma_f32_32x32x16_bf16(cand, opa, zero, 0, 0, 0));
                    const unsigned mb = mf_sign_bits(__builtin_amdgcn_mfma_f32_32x32x16_bf16(cand, opb, zero, 0, 0, 0));
                    auto sw = __builtin_amdgcn_permlane32_swap(ma, mb, false, false);
                    // now sw[0] = rows-set 0, sw[1] = rows-set 1 of THIS lane's query; drop the bits of
                    // the clamped duplicates past the end of a partial word
                    const unsigned m = ((sw[0] << 16) | (sw[1] & 0xffffu)) & (0xffffffffu << (32 - (w_cur.c1 + w_cur.c2)));
                    if (dbg & 64) atomicAdd(&gp->exhaustive_count, (unsigned)__builtin_popcount(m));  // profiling: candidates passed
                    mask[widx][lane] = m;
                    if (lane == 0) {
                        wbase[widx] = (unsigned)w_cur.b1;
                        wseg[widx] = make_uint2((unsigned)w_cur.c1, (unsigned)(w_cur.b2 - w_cur.c1));
                    }
                    nzw |= (m != 0 ? 1u : 0u) << widx;
                    ++widx;
                };
#if GSX_BRICK_UNROLL3
                // the three-deep word pipeline with ROTATING names instead of register moves (the ISA of the rolled loop spends
                // ~10 of its 105 VALU instructions per word on v_mov rotations of the two prefetched points and words)
                Word w0 = next_word(0, 0);
                Word w1 = next_word(w0.r, w0.off), w2;
                float4 p0 = fetch(w0), p1 = fetch(w1), p2;
                for (;;) {
                    if (!(w0.c1 > 0)) break;
                    w2 = next_word(w1.r, w1.off);
                    p2 = fetch(w2);
                    process(w0, p0);
                    if (!(w1.c1 > 0)) break;
                    w0 = next_word(w2.r, w2.off);
                    p0 = fetch(w0);
                    process(w1, p1);
                    if (!(w2.c1 > 0)) break;
                    w1 = next_word(w0.r, w0.off);
                    p1 = fetch(w1);
                    process(w2, p2);
                }
#else
                Word w_cur = next_word(0, 0);
                Word w_n1 = next_word(w_cur.r, w_cur.off);
                float4 p_cur = fetch(w_cur), p_n1 = fetch(w_n1);
                while (w_cur.c1 > 0) {
                    const Word w_n2 = next_word(w_n1.r, w_n1.off);
                    const float4 p_n2 = fetch(w_n2);
                    process(w_cur, p_cur);
                    w_cur = w_n1;
                    p_cur = p_n1;
                    w_n1 = w_n2;
                    p_n1 = p_n2;
                }
#endif
                lst.init();
                }
            }
            if (!mf_done) {
            lst.init();
            for (int r = 0; r < ((dbg & 2) ? 0 : ncrows); ++r) {
                const int gs = __builtin_amdgcn_readlane(rs_start, r);
                const int len = __builtin_amdgcn_readlane(rs_len, r);
                for (int w0 = 0; w0 < len; w0 += 32) {
                    if (widx == WCAP) drain();
                    const int c = min(32, len - w0);
                    const float4 *__restrict__ p = refs + gs + w0;
                    const float neg_tau = -tau;  // dead lanes: tau = -1 => t > 0 => bit 0
                    unsigned m = 0;
                    int i = 0;
                    // 4 candidates per scalar load (s_load_dwordx16), 8 in flight
                    struct Quad { float4 v[4]; };
                    const Quad *__restrict__ pq = reinterpret_cast<const Quad *>(p);
                    for (; i + 8 <= c; i += 8) {
                        const Quad a = pq[i / 4], b = pq[i / 4 + 1];  // wave-uniform addresses
#pragma unroll
                        for (int u = 0; u < 4; ++u) m = shift_in_lt(m, qx, qy, qz, a.v[u].x, a.v[u].y, a.v[u].z, neg_tau);
#pragma unroll
                        for (int u = 0; u < 4; ++u) m = shift_in_lt(m, qx, qy, qz, b.v[u].x, b.v[u].y, b.v[u].z, neg_tau);
                    }
                    for (; i < c; ++i) {
                        float4 P = p[i];
                        m = shift_in_lt(m, qx, qy, qz, P.x, P.y, P.z, neg_tau);
                    }
                    m <<= (32 - c);  // candidate i of this word <-> bit 31-i
                    if (dbg & 64) atomicAdd(&gp->exhaustive_count, (unsigned)__builtin_popcount(m));  // profiling: candidates passed
                    mask[widx][lane] = m;
                    if (lane == 0) {
                        wbase[widx] = (unsigned)(gs + w0);
                        if constexpr (MF) wseg[widx] = make_uint2(32u, 0u);
                    }
                    nzw |= (m != 0 ? 1u : 0u) << widx;
                    ++widx;
                }
            }
            }
            drain();

            // ---- exact iff the (k+1)-th distance lies inside the searched cells
            if (is_query) {
                const double kth_d2 = lst.kth(kq);
                if (dbg & 8) {
                    if (kth_d2 == 12345.0) mean_out[0] = 1.0f;  // keeps the list live, writes nothing
                } else if (dbg & 4) {
                    mean_out[(int)__float_as_uint(qp.w) - q_begin] = (float)kth_d2;
                } else if (kth_d2 <= racc_sq) {
                    kth_emit(gp, kth_out, (int)__float_as_uint(qp.w) - q_begin, kth_d2, qx, qy, qz);
                    if constexpr (NET) mean_out[(int)__float_as_uint(qp.w) - q_begin] = mean_from_net(lst, k);
                    else mean_out[(int)__float_as_uint(qp.w) - q_begin] = mean_from_list<KCAP>(lst, k);
                } else {
                    unsigned slot = atomicAdd(&gp->fail_count, 1u);
                    faillist[slot] = (unsigned)qidx;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- knn_ring (fallback)
// One wave per failed query: search the (2H+1)^3 cells around the query's cell, H = 2,4,8..;
// lanes split every row, keep private sorted lists, then the wave merges the 64 lists by
// repeated min-extraction.  Exact once the (k+1)-th distance <= H*r_safe or the ring covers
// the whole grid.
__device__ __forceinline__ double wave_min_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}

constexpr int RING_ROWS = 320;  // rows whose bounds are prefetched in parallel (covers H <= 8: 17x17)

template <int KCAP>
__global__ __launch_bounds__(BRICK_THREADS) void knn_ring_kernel(
    GridParams *__restrict__ gp, const float4 *__restrict__ refs, const unsigned *__restrict__ rstart,
    const float4 *__restrict__ qpts, const unsigned *__restrict__ faillist, int k, int q_begin,
    float *__restrict__ mean_out, double *__restrict__ kth_out, unsigned *__restrict__ heavylist, int out_count, int after_fast)
{
    __shared__ double s_out[BRICK_THREADS / 64][KCAP];
    __shared__ int s_rs[BRICK_THREADS / 64][RING_ROWS];       // first point of each row
    __shared__ int s_ro[BRICK_THREADS / 64][RING_ROWS + 1];   // flat offset of each row
    const int lane = lane_id();
    const int wv = uniform((int)(threadIdx.x >> 6));
    double *out = s_out[wv];
    int *rs = s_rs[wv], *ro = s_ro[wv];
    const GridParams g = *gp;
    if (g.bad_input) {  // non-finite coordinates: no kernel searched anything -- the whole output becomes NaN
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < out_count; i += gridDim.x * blockDim.x)
            mean_out[i] = __builtin_nanf("");
        return;
    }
    const int nfail = (int)(after_fast ? g.ring2_count : g.fail_count);   // after_fast: what knn_ring_fast could not finish
    if (nfail == 0) return;
    const int kk = k + 1;

    WorkQueue wq;
    wq_init(wq, gp->ring_ctr, nfail, BRICK_THREADS / 64);
    for (;;) {
        const int t = uniform(wq_next(wq));
        if (t < 0) break;
        const unsigned fq = faillist[t];
        const float4 qp = qpts[fq];
        const double qxd = (double)qp.x, qyd = (double)qp.y, qzd = (double)qp.z;
        const int cx = cell_coord(qp.x, g.ox, g.inv_h, g.nx);
        const int cy = cell_coord(qp.y, g.oy, g.inv_h, g.ny);
        const int cz = cell_coord(qp.z, g.oz, g.inv_h, g.nz);

        for (int H = 2;; H *= 2) {
            double a[KCAP];  // ascending, +inf padded: a[0] is this lane's smallest
#pragma unroll
            for (int i = 0; i < KCAP; ++i) a[i] = __builtin_inf();
            auto consider = [&](const float4 p) __attribute__((always_inline)) {
                double d = dist2_f64(qxd, qyd, qzd, p.x, p.y, p.z);
                if (d < a[KCAP - 1]) {
#pragma unroll
                    for (int i = 0; i < KCAP; ++i) {
                        double lo, hi;
                        asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a[i]), "v"(d));
                        asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a[i]), "v"(d));
                        a[i] = lo;
                        d = hi;
                    }
                }
            };
            const int x0 = max(cx - H, 0), x1 = min(cx + H, g.nx - 1);
            const int y0 = max(cy - H, 0), y1 = min(cy + H, g.ny - 1);
            const int z0 = max(cz - H, 0), z1 = min(cz + H, g.nz - 1);
            const int nyr = y1 - y0 + 1;
            const int nrows = nyr * (z1 - z0 + 1);
            if (nrows <= RING_ROWS) {
                // all row bounds in parallel (one lane per row), wave scan -> flat offsets in LDS
                int carry = 0;
                for (int r0 = 0; r0 < nrows; r0 += 64) {
                    const int r = r0 + lane;
                    int s = 0, len = 0;
                    if (r < nrows) {
                        const int zz = z0 + r / nyr, yy = y0 + r % nyr;
                        const int row = row_base(g, yy, zz);
                        s = (int)rstart[row + x0];
                        len = (int)rstart[row + x1 + 1] - s;
                    }
                    int inc = len;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        int o = __shfl_up(inc, off);
                        if (lane >= off) inc += o;
                    }
                    if (r < nrows) {
                        rs[r] = s;
                        ro[r] = carry + inc - len;
                    }
                    carry += __shfl(inc, 63);
                }
                const int total = carry;
                if (g.heavy_limit > 0 && total > g.heavy_limit) {  // wave-uniform
                    if (lane == 0) heavylist[atomicAdd(&gp->heavy_count, 1u)] = fq;
                    break;
                }
                if (lane == 0) ro[nrows] = total;
                wave_sync();
                // every lane walks the flat candidate range with stride 64, 4 independent loads in flight
                int row = 0;
                for (int f0 = lane; f0 < total; f0 += 256) {
                    float4 p[4];
                    bool ok[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int f = f0 + 64 * u;
                        ok[u] = f < total;
                        if (ok[u]) {
                            while (f >= ro[row + 1]) ++row;
                            p[u] = refs[rs[row] + (f - ro[row])];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (ok[u]) consider(p[u]);
                }
            } else {
                if (g.heavy_limit > 0) {
                    int total = 0;
                    for (int r = lane; r < nrows; r += 64) {
                        const int row = row_base(g, y0 + r % nyr, z0 + r / nyr);
                        total += (int)(rstart[row + x1 + 1] - rstart[row + x0]);
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
                    if (uniform(total) > g.heavy_limit) {
                        if (lane == 0) heavylist[atomicAdd(&gp->heavy_count, 1u)] = fq;
                        break;
                    }
                }
                for (int zz = z0; zz <= z1; ++zz)
                    for (int yy = y0; yy <= y1; ++yy) {
                        const int row = row_base(g, yy, zz);
                        const int s = (int)rstart[row + x0], e = (int)rstart[row + x1 + 1];
                        for (int j = s + lane; j < e; j += 64) consider(refs[j]);
                    }
            }
            // merge: kk rounds of wave-min over the list heads
            for (int r = 0; r < kk; ++r) {
                const double mn = wave_min_f64(a[0]);
                const unsigned long long eq = __ballot(a[0] == mn);
                const int win = (int)__builtin_ctzll(eq);
                if (lane == 0) out[r] = mn;
                if (lane == win) {
#pragma unroll
                    for (int i = 0; i + 1 < KCAP; ++i) a[i] = a[i + 1];
                    a[KCAP - 1] = __builtin_inf();
                }
            }
            wave_sync();
            const bool covers = x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.nx - 1 && y1 == g.ny - 1 && z1 == g.nz - 1;
            const double rH = (double)H * g.hprime * (1.0 - 1e-3);
            const double kth = out[kk - 1];
            if (covers || kth <= rH * rH) {
                // sqrt of the k+1 winners in parallel (one lane each), then lane 0 sums in numpy order
                for (int i = lane; i < kk; i += 64) out[i] = __dsqrt_rn(out[i]);
                wave_sync();
                if (lane == 0) {
                    if (covers && !(kth <= rH * rH)) atomicAdd(&gp->exhaustive_count, 1u);
                    double sum = pairwise_sum_le128([&](int i) { return out[1 + i]; }, k);
                    mean_out[(int)__float_as_uint(qp.w) - q_begin] = __double2float_rn(__ddiv_rn(sum, (double)k));
                    kth_emit(gp, kth_out, (int)__float_as_uint(qp.w) - q_begin, kth, qp.x, qp.y, qp.z);
                }
                wave_sync();
                break;
            }
            wave_sync();
        }
    }
}

// ---------------------------------------------------------------- knn_anyk (k > 64)
// The reference's cKDTree path takes any k (data_processor.py:171; the CLI's hidden --sor_k, main.py:278); the
// register-resident lists of the kernels above stop at k = 64.  Larger k takes this exact, list-free path: one wave per
// query; the (2H+1)^3 cells around the query's cell are walked as contiguous x-rows; the (k+1)-th smallest squared
// distance is found by an 8-pass radix SELECT on the float64 bit pattern (non-negative doubles order like their bits): a
// 256-bin LDS histogram per pass over the candidates that match the prefix so far, distances recomputed each pass -- no
// per-query storage that grows with the candidate count.  Exact iff that distance is within the guaranteed radius
// H h' (1 - 1e-3) (section 4.2's argument) or the block covers the grid; otherwise H grows.  The k+1 winners (ties at the
// k-th value are equal numbers: which one is taken does not matter) are then collected, sorted in LDS by a wave-wide
// bitonic network and summed by one lane in numpy's pairwise order (loops_utils.h.src: 8 accumulators up to 128
// elements, halves rounded down to a multiple of 8 above).  k + 1 <= 2048.  Slow (9 distance evaluations per candidate)
// and rare: nothing in the CLI's documented flags reaches it (--sor_intensity maps to k <= 50).
constexpr int ANYK_MAX = 2048;

template <int DEPTH, class F>
__device__ double pairwise_sum_np(F at, int lo, int n)   // numpy's pairwise_sum for any n (DEPTH halvings unrolled)
{
    if constexpr (DEPTH == 0) {
        return pairwise_sum_le128([&](int i) { return at(lo + i); }, n);
    } else {
        if (n <= 128) return pairwise_sum_le128([&](int i) { return at(lo + i); }, n);
        int n2 = n / 2;
        n2 -= n2 % 8;
        return __dadd_rn(pairwise_sum_np<DEPTH - 1>(at, lo, n2), pairwise_sum_np<DEPTH - 1>(at, lo + n2, n - n2));
    }
}

__global__ __launch_bounds__(BRICK_THREADS) void knn_anyk_kernel(GridParams *__restrict__ gp, const float4 *__restrict__ refs,
                                                                 const unsigned *__restrict__ rstart, const float4 *__restrict__ qpts,
                                                                 int nq, int k, int q_begin, float *__restrict__ mean_out,
                                                                 double *__restrict__ kth_out)
{
    __shared__ unsigned s_hist[BRICK_THREADS / 64][256];
    __shared__ double s_sel[BRICK_THREADS / 64][ANYK_MAX];
    __shared__ unsigned s_cnt[BRICK_THREADS / 64];
    const int lane = lane_id();
    const int wv = uniform((int)(threadIdx.x >> 6));
    unsigned *hist = s_hist[wv];
    double *sel = s_sel[wv];
    const GridParams g = *gp;
    const int kk = k + 1;
    const int nwaves = gridDim.x * (BRICK_THREADS / 64);
    if (g.bad_input) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) mean_out[i] = __builtin_nanf("");
        return;
    }
    for (int t = blockIdx.x * (BRICK_THREADS / 64) + wv; t < nq; t += nwaves) {
        const float4 qp = qpts[t];
        if (__float_as_uint(qp.w) >> 31) continue;   // reference-only point
        const double qxd = (double)qp.x, qyd = (double)qp.y, qzd = (double)qp.z;
        const int cx = cell_coord(qp.x, g.ox, g.inv_h, g.nx);
        const int cy = cell_coord(qp.y, g.oy, g.inv_h, g.ny);
        const int cz = cell_coord(qp.z, g.oz, g.inv_h, g.nz);
        for (int H = 1;; ++H) {
            const int x0 = max(cx - H, 0), x1 = min(cx + H, g.nx - 1);
            const int y0 = max(cy - H, 0), y1 = min(cy + H, g.ny - 1);
            const int z0 = max(cz - H, 0), z1 = min(cz + H, g.nz - 1);
            const bool covers = x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.nx - 1 && y1 == g.ny - 1 && z1 == g.nz - 1;
            // f(d2 bits) for every candidate of the block
            auto for_each = [&](auto f) __attribute__((always_inline)) {
                for (int zz = z0; zz <= z1; ++zz)
                    for (int yy = y0; yy <= y1; ++yy) {
                        const int row = row_base(g, yy, zz);
                        const int sb = (int)rstart[row + x0], se = (int)rstart[row + x1 + 1];
                        for (int j = sb + lane; j < se; j += 64) {
                            const float4 p = refs[j];
                            f((unsigned long long)__double_as_longlong(dist2_f64(qxd, qyd, qzd, p.x, p.y, p.z)));
                        }
                    }
            };
            // ---- radix select of the kk-th smallest
            unsigned long long prefix = 0ull, pmask = 0ull;
            int remaining = kk;
            bool enough = true;
            for (int pass = 0; pass < 8 && enough; ++pass) {
                const int shift = 56 - 8 * pass;
                for (int i = lane; i < 256; i += 64) hist[i] = 0u;
                wave_sync();
                for_each([&](unsigned long long b) {
                    if ((b & pmask) == prefix) atomicAdd(&hist[(unsigned)(b >> shift) & 255u], 1u);
                });
                wave_sync();
                // lane l owns bins 4l .. 4l+3
                const unsigned c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
                unsigned inc = c0 + c1 + c2 + c3;
                const unsigned mine = inc;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned o = __shfl_up(inc, off);
                    if (lane >= off) inc += o;
                }
                const unsigned total = __shfl(inc, 63);
                if (pass == 0 && (int)total < kk) {   // fewer than k+1 points in the block
                    enough = false;
                    break;
                }
                const unsigned before = inc - mine;
                const bool here = (int)before < remaining && remaining <= (int)(before + mine);
                const int owner = (int)__builtin_ctzll(__ballot(here));
                unsigned bin = 0, skip = 0;
                if (lane == owner) {
                    const unsigned c[4] = {c0, c1, c2, c3};
                    unsigned run = before;
                    for (int j = 0; j < 4; ++j) {
                        if ((int)(run + c[j]) >= remaining) {
                            bin = 4u * (unsigned)lane + (unsigned)j;
                            skip = run;
                            break;
                        }
                        run += c[j];
                    }
                }
                bin = (unsigned)__shfl((int)bin, owner);
                skip = (unsigned)__shfl((int)skip, owner);
                remaining -= (int)skip;
                prefix |= (unsigned long long)bin << shift;
                pmask |= 0xffull << shift;
                wave_sync();
            }
            if (!enough) {
                if (covers) {   // N < k + 1: cKDTree pads with inf -> the mean is inf
                    if (lane == 0) {
                        mean_out[(int)__float_as_uint(qp.w) - q_begin] = __builtin_inff();
                        kth_emit(gp, kth_out, (int)__float_as_uint(qp.w) - q_begin, __builtin_inf(), qp.x, qp.y, qp.z);
                    }
                    break;
                }
                continue;
            }
            const double kth = __longlong_as_double((long long)prefix);
            const double rH = (double)H * g.hprime * (1.0 - 1e-3);
            if (!(covers || kth <= rH * rH)) continue;
            // ---- collect the values strictly below the kk-th, fill up with the kk-th itself, sort, sum
            if (lane == 0) s_cnt[wv] = 0u;
            wave_sync();
            for_each([&](unsigned long long b) {
                if (b < prefix) sel[atomicAdd(&s_cnt[wv], 1u)] = __longlong_as_double((long long)b);
            });
            wave_sync();
            const int nlt = (int)s_cnt[wv];   // < kk by the definition of the kk-th smallest
            int P = 64;
            while (P < kk) P <<= 1;
            for (int i = nlt + lane; i < P; i += 64) sel[i] = i < kk ? kth : __builtin_inf();
            wave_sync();
            for (int size = 2; size <= P; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int i = lane; i < P / 2; i += 64) {
                        const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                        const bool up = (lo & size) == 0;
                        const double a = sel[lo], b = sel[hi];
                        if ((a > b) == up) {
                            sel[lo] = b;
                            sel[hi] = a;
                        }
                    }
                    wave_sync();
                }
            for (int i = lane; i < kk; i += 64) sel[i] = sqrt_rn_dist2(sel[i]);
            wave_sync();
            if (lane == 0) {
                const double sum = pairwise_sum_np<5>([&](int i) { return sel[1 + i]; }, 0, k);   // entry 0 is the query itself
                mean_out[(int)__float_as_uint(qp.w) - q_begin] = __double2float_rn(__ddiv_rn(sum, (double)k));
                kth_emit(gp, kth_out, (int)__float_as_uint(qp.w) - q_begin, kth, qp.x, qp.y, qp.z);   // (slab mode: the certificate is counted here)
                if (covers && !(kth <= rH * rH)) atomicAdd(&gp->exhaustive_count, 1u);
            }
            wave_sync();
            break;
        }
    }
}

// ---------------------------------------------------------------- knn_heavy (ring queries next to a huge cell)
// A far "floater" whose ring reaches a cell holding most of the cloud would make ONE wave scan
// millions of candidates (measured: 11 ms for 9 such queries at 1M splats).  knn_ring hands these
// queries over; here every wave takes (query, 32768-point chunk of the WHOLE sorted array), keeps
// the k+1 smallest squared distances of its chunk, and a second kernel merges the chunks of a
// query -- exhaustive, hence exact, and spread over the chip.
template <int KCAP>
struct LaneTop {
    double a[KCAP];  // ascending, +inf padded
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int i = 0; i < KCAP; ++i) a[i] = __builtin_inf();
    }
    __device__ __forceinline__ void consider(double d)
    {
        if (d < a[KCAP - 1]) {
#pragma unroll
            for (int i = 0; i < KCAP; ++i) {
                double lo, hi;
                asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a[i]), "v"(d));
                asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a[i]), "v"(d));
                a[i] = lo;
                d = hi;
            }
        }
    }
    // the kk smallest over the whole wave -> out[0..kk) (written by lane 0)
    __device__ __forceinline__ void wave_merge(int kk, int lane, double *out)
    {
        for (int r = 0; r < kk; ++r) {
            const double mn = wave_min_f64(a[0]);
            const unsigned long long eq = __ballot(a[0] == mn);
            const int win = (int)__builtin_ctzll(eq);
            if (lane == 0) out[r] = mn;
            if (lane == win) {
#pragma unroll
                for (int i = 0; i + 1 < KCAP; ++i) a[i] = a[i + 1];
                a[KCAP - 1] = __builtin_inf();
            }
        }
    }
};

template <int KCAP>
__global__ __launch_bounds__(BRICK_THREADS) void knn_heavy_scan_kernel(const float4 *__restrict__ refs, int n,
                                                                       const float4 *__restrict__ qpts,
                                                                       const unsigned *__restrict__ heavylist, int first,
                                                                       int count, int k, double *__restrict__ part)
{
    const int lane = lane_id();
    const int kk = k + 1;
    const int nchunks = (n + HEAVY_CHUNK - 1) / HEAVY_CHUNK;
    const long long items = (long long)count * nchunks;
    const long long wave0 = (long long)blockIdx.x * (BRICK_THREADS / 64) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (BRICK_THREADS / 64);
    for (long long it = wave0; it < items; it += nwaves) {
        const int qi = (int)(it / nchunks), c = (int)(it - (long long)qi * nchunks);
        const float4 qp = qpts[heavylist[first + qi]];
        const double qxd = (double)qp.x, qyd = (double)qp.y, qzd = (double)qp.z;
        LaneTop<KCAP> top;
        top.init();
        const int j0 = c * HEAVY_CHUNK, j1 = min(n, j0 + HEAVY_CHUNK);
        for (int j = j0 + lane; j < j1; j += 256) {
            float4 p[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ok[u] = j + 64 * u < j1;
                if (ok[u]) p[u] = refs[j + 64 * u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ok[u]) top.consider(dist2_f64(qxd, qyd, qzd, p[u].x, p[u].y, p[u].z));
        }
        top.wave_merge(kk, lane, part + (size_t)it * KCAP);
    }
}

template <int KCAP>
__global__ __launch_bounds__(BRICK_THREADS) void knn_heavy_merge_kernel(GridParams *__restrict__ gp, int n,
                                                                        const float4 *__restrict__ qpts,
                                                                        const unsigned *__restrict__ heavylist, int first,
                                                                        int count, int k, int q_begin,
                                                                        const double *__restrict__ part,
                                                                        float *__restrict__ mean_out,
                                                                        double *__restrict__ kth_out)
{
    __shared__ double s_out[BRICK_THREADS / 64][KCAP];
    const int lane = lane_id();
    const int wv = uniform((int)(threadIdx.x >> 6));
    double *out = s_out[wv];
    const int kk = k + 1;
    const int nchunks = (n + HEAVY_CHUNK - 1) / HEAVY_CHUNK;
    const int qi = blockIdx.x * (BRICK_THREADS / 64) + wv;
    if (qi >= count) return;
    LaneTop<KCAP> top;
    top.init();
    const double *mine = part + (size_t)qi * nchunks * KCAP;
    for (int e = lane; e < nchunks * KCAP; e += 64)
        if (e % KCAP < kk) top.consider(mine[e]);
    top.wave_merge(kk, lane, out);
    wave_sync();
    const double kth = out[kk - 1];
    for (int i = lane; i < kk; i += 64) out[i] = __dsqrt_rn(out[i]);
    wave_sync();
    if (lane == 0) {
        const float4 qp = qpts[heavylist[first + qi]];
        atomicAdd(&gp->exhaustive_count, 1u);
        double sum = pairwise_sum_le128([&](int i) { return out[1 + i]; }, k);
        mean_out[(int)__float_as_uint(qp.w) - q_begin] = __double2float_rn(__ddiv_rn(sum, (double)k));
        kth_emit(gp, kth_out, (int)__float_as_uint(qp.w) - q_begin, kth, qp.x, qp.y, qp.z);   // (slab mode: the certificate is counted here)
    }
}

// ---------------------------------------------------------------- host side
static int grid_blocks(const gsx_ctx *ctx, int64_t n, int per_thread = 1)
{
    int64_t want = (n + 256LL * per_thread - 1) / (256LL * per_thread);
    int64_t cap = (int64_t)ctx->num_cu * 16;
    return (int)std::max<int64_t>(1, std::min(want, cap));
}

// ---------------------------------------------------------------- adaptive refinement (level L -> L+1)
// Bucket-major index of cell (cx, cy, cz) -- the same mapping as row_base_g in knn_brick.
__device__ __forceinline__ int cell_index_of(const GridParams *gp, int cx, int cy, int cz)
{
    const int g = gp->bk_g;
    const int by_ = cy / g, bz_ = cz / g;
    return (bz_ * gp->bk_ny + by_) * gp->bk_cells + ((cz - bz_ * g) * g + (cy - by_ * g)) * gp->nx + cx;
}

// per component of deferred bricks: fresh counters and an empty query box
__global__ void reset_refine_kernel(GridParams *gp)
{
    gp->sub_count = 0;
    gp->sub_queries = 0;
    for (int a = 0; a < 3; ++a) {
        gp->qb_lo[a] = __builtin_inff();
        gp->qb_hi[a] = -__builtin_inff();
    }
}

// one 64-thread block per deferred brick: flag the cells of its neighbourhood (pass 0: value 1) and its
// own cells (pass 1: value 3 -- their points are the queries of the finer level)
__global__ __launch_bounds__(64) void mark_cells_kernel(const GridParams *__restrict__ gp, const unsigned *__restrict__ bricks,
                                                        unsigned count, uint8_t *__restrict__ cellflag, int own_pass)
{
    if (blockIdx.x >= count) return;
    const int b = (int)bricks[blockIdx.x];
    const int nbx = gp->nbx, nby = gp->nby;
    const int bz = b / (nbx * nby), brem = b - bz * nbx * nby, by = brem / nbx, bx = brem - by * nbx;
    const int t = threadIdx.x;
    const int dx = t & 3, dy = (t >> 2) & 3, dz = t >> 4;
    if (dx >= gp->bdx + 2 || dy >= gp->bdy + 2 || dz >= gp->bdz + 2) return;
    const int cx = bx * gp->bdx - 1 + dx, cy = by * gp->bdy - 1 + dy, cz = bz * gp->bdz - 1 + dz;
    if (cx < 0 || cy < 0 || cz < 0 || cx >= gp->nx || cy >= gp->ny || cz >= gp->nz) return;
    const bool own = dx >= 1 && dx <= gp->bdx && dy >= 1 && dy <= gp->bdy && dz >= 1 && dz <= gp->bdz;
    if (own_pass ? own : true) cellflag[cell_index_of(gp, cx, cy, cz)] = own_pass ? 3 : 1;
}

__device__ __forceinline__ void atomic_min_f32(float *addr, float v)  // finite v; *addr starts at +inf
{
    if (v >= 0.0f) atomicMin(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned *>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f32(float *addr, float v)  // finite v; *addr starts at -inf
{
    if (v >= 0.0f) atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned *>(addr), __float_as_uint(v));
}

// bounding box and number of the finer level's queries (points of the cells flagged 3).  Grid-stride,
// reduced per workgroup first: one set of same-address atomics per WAVE serialised in L2 and took
// 12 ms at 10M splats.
__global__ __launch_bounds__(256) void query_bbox_kernel(GridParams *__restrict__ gp, const float4 *__restrict__ refs, int n,
                                                         const uint8_t *__restrict__ cellflag)
{
    __shared__ float s_red[4][6];
    __shared__ unsigned s_cnt[4];
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    unsigned cnt = 0;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const float4 P = refs[j];
        const int cx = cell_coord(P.x, gp->ox, gp->inv_h, gp->nx), cy = cell_coord(P.y, gp->oy, gp->inv_h, gp->ny),
                  cz = cell_coord(P.z, gp->oz, gp->inv_h, gp->nz);
        if (cellflag[cell_index_of(gp, cx, cy, cz)] == 3) {
            lo[0] = fminf(lo[0], P.x); lo[1] = fminf(lo[1], P.y); lo[2] = fminf(lo[2], P.z);
            hi[0] = fmaxf(hi[0], P.x); hi[1] = fmaxf(hi[1], P.y); hi[2] = fmaxf(hi[2], P.z);
            ++cnt;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
        cnt += __shfl_xor(cnt, off);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            s_red[wv][a] = lo[a];
            s_red[wv][3 + a] = hi[a];
        }
        s_cnt[wv] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (c) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                atomic_min_f32(&gp->qb_lo[a], fminf(fminf(s_red[0][a], s_red[1][a]), fminf(s_red[2][a], s_red[3][a])));
                atomic_max_f32(&gp->qb_hi[a], fmaxf(fmaxf(s_red[0][3 + a], s_red[1][3 + a]), fmaxf(s_red[2][3 + a], s_red[3][3 + a])));
            }
            atomicAdd(&gp->sub_queries, c);
        }
    }
}

struct ClipBox {
    float lo[3], hi[3];  // reference points outside are not gathered
    float r_cert;        // every such point is farther than this from every query (0 = no clipping)
};

// gather the points of the flagged cells into a SoA sub-cloud.  One workgroup per 8192-point chunk:
// count, reserve the chunk's run with ONE atomic, then append in order (a returning atomic per wave
// on the same counter cost 1.8 ms at 10M splats).
constexpr int GATHER_CHUNK = 8192;
__global__ __launch_bounds__(256) void gather_sub_kernel(GridParams *__restrict__ gp, const float4 *__restrict__ refs, int n,
                                                         const uint8_t *__restrict__ cellflag, ClipBox clip,
                                                         float *__restrict__ sub, unsigned *__restrict__ sub_orig,
                                                         unsigned *__restrict__ sub_sorted)
{
    __shared__ unsigned s_wave[4];
    __shared__ unsigned s_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j0 = blockIdx.x * GATHER_CHUNK, j1 = min(n, j0 + GATHER_CHUNK);
    auto flag_of = [&](int j, float4 &P) -> unsigned {
        P = refs[j];
        const int cx = cell_coord(P.x, gp->ox, gp->inv_h, gp->nx), cy = cell_coord(P.y, gp->oy, gp->inv_h, gp->ny),
                  cz = cell_coord(P.z, gp->oz, gp->inv_h, gp->nz);
        unsigned flag = cellflag[cell_index_of(gp, cx, cy, cz)];
        if (flag == 1 && clip.r_cert > 0.0f &&
            !(P.x >= clip.lo[0] && P.x <= clip.hi[0] && P.y >= clip.lo[1] && P.y <= clip.hi[1] && P.z >= clip.lo[2] &&
              P.z <= clip.hi[2]))
            flag = 0;  // a reference point farther than r_cert from the box of the queries
        return flag;
    };
    // pass 1: how many points of this chunk are taken, per wave (each wave owns a contiguous quarter)
    const int q = (j1 - j0 + 3) / 4;
    const int w0 = j0 + wv * q, w1 = min(j1, w0 + q);
    unsigned mine = 0;
    for (int j = w0 + lane; j < w1; j += 64) {
        float4 P;
        mine += flag_of(j, P) != 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0) s_wave[wv] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        s_base = tot ? atomicAdd(&gp->sub_count, tot) : 0u;
    }
    __syncthreads();
    unsigned pos = s_base;
    for (int i = 0; i < wv; ++i) pos += s_wave[i];
    // pass 2: append
    for (int jb = w0; jb < w1; jb += 64) {
        const int j = jb + lane;
        float4 P = make_float4(0.f, 0.f, 0.f, 0.f);
        const unsigned flag = j < w1 ? flag_of(j, P) : 0u;
        const unsigned long long take = __ballot(flag != 0);
        if (flag != 0) {
            const unsigned at = pos + (unsigned)__builtin_popcountll(take & ((1ull << lane) - 1ull));
            sub[at] = P.x;
            sub[(size_t)n + at] = P.y;
            sub[2 * (size_t)n + at] = P.z;
            sub_orig[at] = __float_as_uint(P.w) | (flag == 3 ? 0x80000000u : 0u);
            sub_sorted[at] = (unsigned)j;
        }
        pos += (unsigned)__builtin_popcountll(take);
    }
}

// Density probe of a gathered sub-cloud (round 3).  The finer level used to size its cells from the sub-cloud's bounding-box
// VOLUME; a Gaussian blob's core is ~30x denser than its box average, so every level only peeled a shell (3-4 levels per
// blob).  Here: counts of a coarse G^3 grid over the box of the group's queries, then the point-weighted histogram of
// log2(count) -- the host reads 32 numbers and sizes the cells for the density that 85 % of the points do not exceed,
// divided by the factor a brick may be over-full before it is deferred again.  Affects speed only: any cell edge is exact.
struct ProbeBox {
    float lo[3], inv[3];   // bin = (int)((p - lo) * inv), inv = G / extent
    int g;
};
__global__ __launch_bounds__(256) void probe_count_kernel(const float *__restrict__ sub, int n_stride, const unsigned *__restrict__ n_sub_p,
                                                          ProbeBox pb, unsigned *__restrict__ counts)
{
    const int n_sub = (int)*n_sub_p;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_sub; i += gridDim.x * 256) {
        const float p[3] = {sub[i], sub[(size_t)n_stride + i], sub[2 * (size_t)n_stride + i]};
        int b[3];
        bool in = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float t = (p[a] - pb.lo[a]) * pb.inv[a];
            in = in && t >= 0.0f && t < (float)pb.g + 1.0f;
            b[a] = min(pb.g - 1, max(0, (int)t));
        }
        if (in) atomicAdd(&counts[(b[2] * pb.g + b[1]) * pb.g + b[0]], 1u);
    }
}
__global__ __launch_bounds__(256) void probe_hist_kernel(const float *__restrict__ sub, int n_stride, const unsigned *__restrict__ n_sub_p,
                                                         ProbeBox pb, const unsigned *__restrict__ counts, unsigned *__restrict__ hist32)
{
    __shared__ unsigned s_h[32];
    if (threadIdx.x < 32) s_h[threadIdx.x] = 0u;
    __syncthreads();
    const int n_sub = (int)*n_sub_p;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_sub; i += gridDim.x * 256) {
        const float p[3] = {sub[i], sub[(size_t)n_stride + i], sub[2 * (size_t)n_stride + i]};
        int b[3];
        bool in = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float t = (p[a] - pb.lo[a]) * pb.inv[a];
            in = in && t >= 0.0f && t < (float)pb.g + 1.0f;
            b[a] = min(pb.g - 1, max(0, (int)t));
        }
        if (in) {
            const unsigned c = counts[(b[2] * pb.g + b[1]) * pb.g + b[0]];
            atomicAdd(&s_h[31 - __builtin_clz(c | 1u)], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 32 && s_h[threadIdx.x]) atomicAdd(&hist32[threadIdx.x], s_h[threadIdx.x]);
}

// A finer-level result is the exact answer iff its (k+1)-th neighbour lies inside the region this
// level guarantees to have gathered: every point outside the brick's neighbourhood is farther than
// the distance to the nearest neighbourhood face that has cells behind it (same bound and margins
// as knn_brick's boundary-aware acceptance radius).  Anything else goes to this level's knn_ring.
__global__ __launch_bounds__(256) void merge_sub_kernel(GridParams *__restrict__ gp, const float *__restrict__ sub, int n,
                                                        const unsigned *__restrict__ sub_orig,
                                                        const unsigned *__restrict__ sub_sorted,
                                                        const float *__restrict__ submean, const double *__restrict__ subkth,
                                                        int q_begin, float r_cert, float *__restrict__ mean_out,
                                                        double *__restrict__ kth_out, unsigned *__restrict__ faillist)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= gp->sub_count) return;
    const unsigned so = sub_orig[i];
    if (!(so >> 31)) return;  // only gathered as a reference point
    const int orig = (int)(so & 0x7fffffffu);
    const float q[3] = {sub[i], sub[(size_t)n + i], sub[2 * (size_t)n + i]};
    const float o[3] = {gp->ox, gp->oy, gp->oz};
    const int dim[3] = {gp->nx, gp->ny, gp->nz};
    const int bd[3] = {gp->bdx, gp->bdy, gp->bdz};
    const float hf = (float)gp->hprime;
    float rsafe = r_cert > 0.0f ? r_cert : 3.0e38f;  // clipped references: see gather_sub_kernel
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int c = cell_coord(q[a], o[a], gp->inv_h, dim[a]);
        const int bb = c / bd[a];
        const int ulo = bb * bd[a] - 1, uhi = bb * bd[a] + bd[a];
        const float rel = q[a] - o[a];
        if (ulo > 0) rsafe = fminf(rsafe, rel - (float)ulo * hf - 2e-3f * hf);
        if (uhi < dim[a] - 1) rsafe = fminf(rsafe, (float)(uhi + 1) * hf - rel - 2e-3f * hf);
    }
    const double kth = subkth[i];
    if (rsafe > 0.0f && kth <= (double)rsafe * (double)rsafe) {
        mean_out[orig - q_begin] = submean[i];
        kth_emit(gp, kth_out, orig - q_begin, kth, q[0], q[1], q[2]);
    } else {
        faillist[atomicAdd(&gp->fail_count, 1u)] = sub_sorted[i];
    }
}

struct BrickLaunch {
    GridParams *gp;
    const float4 *refs;
    const unsigned *rstart;
    const float4 *qpts;
    const unsigned *qstart;
    int k;
    int64_t q_begin;
    float *mean_out;
    double *kth_out;
    unsigned *faillist;
    unsigned *ring2;      // what knn_ring_fast hands on to knn_ring
    uint2 *extra;
    unsigned *deferred;
    unsigned *heavylist;
    int64_t out_count;    // entries of mean_out (poisoned with NaN when the input is not finite)
    int64_t n_ref;        // reference points binned at this level (mean density for knn_ring_fast)
};

template <int KCAP, bool MF, bool NET>
static int launch_bricks(gsx_ctx *ctx, const BrickLaunch &a)
{
    // Work is assigned STATICALLY to waves, so every launched workgroup must be resident at once:
    // the grids are sized from the occupancy the built kernels actually get (a non-resident
    // workgroup would run its share only after a resident one has finished all of its own).
    static int occ_brick = 0, occ_extra = 0;
    if (!occ_brick) {
        GSX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_brick, knn_brick_kernel<KCAP, false, MF, NET>, BRICK_THREADS, 0));
        GSX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_extra, knn_brick_kernel<KCAP, true, MF, NET>, BRICK_THREADS, 0));
        occ_brick = std::max(1, std::min(occ_brick, 8));
        occ_extra = std::max(1, std::min(occ_extra, 8));
    }
    GSX_CHECK(timing_begin(ctx, GSX_T_SOR_KNN));
    hipLaunchKernelGGL((knn_brick_kernel<KCAP, false, MF, NET>), dim3(ctx->num_cu * occ_brick), dim3(BRICK_THREADS), 0, ctx->stream,
                       a.gp, a.refs, a.rstart, a.qpts, a.qstart, a.k, (int)a.q_begin, a.mean_out, a.faillist, a.extra,
                       a.deferred, a.kth_out);
    hipLaunchKernelGGL((knn_brick_kernel<KCAP, true, MF, NET>), dim3(ctx->num_cu * occ_extra), dim3(BRICK_THREADS), 0, ctx->stream,
                       a.gp, a.refs, a.rstart, a.qpts, a.qstart, a.k, (int)a.q_begin, a.mean_out, a.faillist, a.extra,
                       a.deferred, a.kth_out);
    GSX_HIP(hipGetLastError());
    GSX_CHECK(timing_end(ctx, GSX_T_SOR_KNN));
    return 0;
}

// ---------------------------------------------------------------- knn_ring_fast
// First attempt at the queries knn_brick could not certify, before knn_ring's per-lane lists.  A failed query's k-th
// neighbour is just beyond its brick's searched cells, i.e. 1..1.4 cell edges away.  So: ONE pass over the cells of the
// 5x5x5 block around the query's cell that a ball of radius r_t reaches (r_t chosen so that ~5(k+1) points are expected
// inside; second attempt with 1.9 cell edges, the most 2 cells certify), with a float32 filter at r_t (relative error of
// d32 <= 3e-7) that moves the few dozen candidates that matter to LDS.  A 64-bin histogram of
// their d32 isolates the ~k+2 that can be among the k+1 nearest; those are ranked exactly in float64 (rank = number of
// candidates before mine).  If at least k+1 candidates are inside r_t, the k+1 nearest are among them and every point
// within r_t was looked at: exact.  One wave per query, < 80 VGPRs (no per-lane lists): six waves per SIMD overlap
// the dependent loads (list entry -> query -> row bounds -> candidates) of different queries.  The queries it cannot
// finish (fewer than k+1 points within 1.9 cell edges, more than CAND inside r_t, heavy rings) go to a second list for
// knn_ring -- appended per wave in batches, because same-address atomics cost ~11 ns each.
constexpr int RINGF_INFLIGHT = 4;
constexpr double RINGF_TARGET = 5.0;   // candidates expected inside r_t, in units of k+1 (3.5: +0.008 ms, 7: +0.05 ms)
constexpr int RINGF_WAVES = 6;         // 5 waves (no scratch): no change; 4: +0.015 ms (profiles/r03_variants.txt)
constexpr int RINGF_ROWS = 25;   // 5 x 5 rows of 5 cells
constexpr int RINGF_LEFT = 32;   // leftovers a wave collects before it appends them to the second list
// histogram bin of a float32 squared distance d <= thr: two roundings of monotone operations, hence monotone in d
// (bins uniform in d hold ~sqrt(d/thr) * 1.5 cnt / 64 candidates: one or two where the (k+1)-th lies)
__device__ __forceinline__ int ringf_bin(float d32, float inv_thr)
{
    return min(63, (int)(d32 * inv_thr * 64.0f));
}
constexpr int ringf_cand(int kcap) { return kcap <= 17 ? 128 : (kcap <= 33 ? 256 : 512); }
template <int KCAP>
__global__ __launch_bounds__(BRICK_THREADS, KCAP <= 33 ? RINGF_WAVES : 3) void knn_ring_fast_kernel(
    GridParams *__restrict__ gp, const float4 *__restrict__ refs, const unsigned *__restrict__ rstart,
    const float4 *__restrict__ qpts, const unsigned *__restrict__ faillist, unsigned *__restrict__ ring2, int k, int q_begin,
    int64_t n_ref, float *__restrict__ mean_out, double *__restrict__ kth_out)
{
    constexpr int CAND = ringf_cand(KCAP);
    constexpr int NF = RINGF_INFLIGHT;
    __shared__ double s_out[BRICK_THREADS / 64][KCAP];
    __shared__ int s_rs[BRICK_THREADS / 64][RINGF_ROWS];
    __shared__ int s_ro[BRICK_THREADS / 64][RINGF_ROWS + 1];
    __shared__ float4 s_cand[BRICK_THREADS / 64][CAND];   // xyz + float32 squared distance
    __shared__ float4 s_fin[BRICK_THREADS / 64][64];      // the candidates that are ranked exactly
    __shared__ double s_cd[BRICK_THREADS / 64][64];
    __shared__ int s_hist[BRICK_THREADS / 64][64];
    __shared__ unsigned s_left[BRICK_THREADS / 64][RINGF_LEFT];
    const int lane = lane_id();
    const int wv = uniform((int)(threadIdx.x >> 6));
    double *out = s_out[wv];
    int *rs = s_rs[wv], *ro = s_ro[wv];
    float4 *cand = s_cand[wv], *fin = s_fin[wv];
    double *cd = s_cd[wv];
    int *hist = s_hist[wv];
    unsigned *left = s_left[wv];
    const GridParams g = *gp;
    if (g.bad_input) return;   // knn_ring fills the output with NaN
    const int nfail = (int)g.fail_count;
    const int kk = k + 1;
    // r_t: ~5(k+1) points expected inside at the cloud's mean density, at most 1.9 cell edges (2 cells certify 1.998)
    const double per_cell = (double)n_ref / (double)max(g.ncells, 1);
    const double rt_max = 1.9 * g.hprime;
    const double rt_first = __builtin_fmin(::cbrt(RINGF_TARGET * kk / (4.18879 * per_cell)) * g.hprime, rt_max);

    int nleft = 0;   // wave-uniform
    auto flush_left = [&]() __attribute__((always_inline)) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&gp->ring2_count, (unsigned)nleft);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        wave_sync();
        if (lane < nleft) ring2[base + lane] = left[lane];
        wave_sync();
        nleft = 0;
    };

    WorkQueue wq;
    wq_init(wq, gp->ringf_ctr, nfail, BRICK_THREADS / 64);
    // the next query's list entry and point are fetched while the current one is being worked on: the chain
    // entry -> point -> row bounds -> candidates is what a wave waits for
    int t_next = uniform(wq_next(wq));
    unsigned fq_next = t_next >= 0 ? faillist[t_next] : 0u;
    float4 qp_next = t_next >= 0 ? qpts[fq_next] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (;;) {
        if (t_next < 0) break;
        const float4 qp = qp_next;
        const unsigned fq = fq_next;
        t_next = uniform(wq_next(wq));
        fq_next = t_next >= 0 ? faillist[t_next] : 0u;   // consumed after this query's row bounds have arrived
        bool fetched_next = false;
        const double qxd = (double)qp.x, qyd = (double)qp.y, qzd = (double)qp.z;
        const int cx = cell_coord(qp.x, g.ox, g.inv_h, g.nx);
        const int cy = cell_coord(qp.y, g.oy, g.inv_h, g.ny);
        const int cz = cell_coord(qp.z, g.oz, g.inv_h, g.nz);
        const int x0 = max(cx - 2, 0), x1 = min(cx + 2, g.nx - 1);
        const int y0 = max(cy - 2, 0), y1 = min(cy + 2, g.ny - 1);
        const int z0 = max(cz - 2, 0), z1 = min(cz + 2, g.nz - 1);
        const int nyr = y1 - y0 + 1;
        const int nrows = nyr * (z1 - z0 + 1);
        const double ux = (qxd - (double)g.ox) * (double)g.inv_h, uy = (qyd - (double)g.oy) * (double)g.inv_h,
                     uz = (qzd - (double)g.oz) * (double)g.inv_h;
        bool solved = false;
        double rt = rt_first;
#pragma unroll 1
        for (int attempt = 0; attempt < 2 && !solved; ++attempt, rt = rt_max) {
            if (attempt == 1 && !(rt_first < rt_max)) break;
            const float thr = (float)(rt * rt) * (1.0f + 1e-6f);
            const float inv_thr = 1.0f / thr;
            const double rtc = rt * (double)g.inv_h;   // r_t in cell units (<= 1.9)
            // all row bounds in parallel (one lane per row), wave scan -> flat offsets.  Each row is trimmed to the cells
            // the ball of radius r_t can reach (the kernel is bound by the candidate fetch: 37 cells instead of 125): in
            // cell units a point of cell c lies in [c - 2.5e-4, c + 1 + 2.5e-4] (the f32 cell index is off by
            // < 2.5e-4 cells, section 4.2), so with a slack of 1e-3 per side no point within r_t is missed.
            int st = 0, len = 0;
            if (lane < nrows) {
                const int yy = y0 + lane % nyr, zz = z0 + lane / nyr;
                const double dy = uy < yy - 1e-3 ? yy - 1e-3 - uy : (uy > yy + 1.001 ? uy - (yy + 1.001) : 0.0);
                const double dz = uz < zz - 1e-3 ? zz - 1e-3 - uz : (uz > zz + 1.001 ? uz - (zz + 1.001) : 0.0);
                const double rem = rtc * rtc - dy * dy - dz * dz;
                if (rem >= 0.0) {
                    const double dxc = __builtin_sqrt(rem) + 1e-3;
                    const int xa = max(x0, (int)__builtin_floor(ux - dxc)), xb = min(x1, (int)__builtin_floor(ux + dxc));
                    if (xa <= xb) {
                        const int row = row_base(g, yy, zz);
                        st = (int)rstart[row + xa];
                        len = (int)rstart[row + xb + 1] - st;
                    }
                }
            }
            int inc = len;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                int o = __shfl_up(inc, off);
                if (lane >= off) inc += o;
            }
            if (lane < nrows) {
                rs[lane] = st;
                ro[lane] = inc - len;
            }
            const int total = __shfl(inc, nrows - 1);
            if (lane == 0) ro[nrows] = total;
            hist[lane] = 0;
            wave_sync();
            if (!fetched_next) {
                qp_next = t_next >= 0 ? qpts[fq_next] : qp;   // in flight together with this query's candidates
                fetched_next = true;
            }
            if (g.heavy_limit > 0 && total > g.heavy_limit) break;   // wave-uniform: knn_ring routes it to knn_heavy
            int cnt = 0, row = 0;
            for (int f0 = 0; f0 < total; f0 += 64 * NF) {   // wave-uniform trip count, NF loads in flight per lane
                float4 p[NF];
                bool in[NF];
#pragma unroll
                for (int u = 0; u < NF; ++u) {
                    const int f = f0 + 64 * u + lane;
                    in[u] = f < total;
                    if (in[u]) {
                        while (f >= ro[row + 1]) ++row;
                        p[u] = refs[rs[row] + (f - ro[row])];
                    }
                }
#pragma unroll
                for (int u = 0; u < NF; ++u) {
                    bool acc = false;
                    float d32 = 0.0f;
                    if (in[u]) {
                        const float dx = qp.x - p[u].x, dy = qp.y - p[u].y, dz = qp.z - p[u].z;
                        d32 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                        acc = d32 <= thr;
                    }
                    const unsigned long long bal = __ballot(acc);
                    if (acc) {
                        const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                        if (pos < CAND) {
                            p[u].w = d32;
                            cand[pos] = p[u];
                            atomicAdd(&hist[ringf_bin(d32, inv_thr)], 1);
                        }
                    }
                    cnt += (int)__builtin_popcountll(bal);
                }
            }
            wave_sync();
            if (cnt > CAND) break;      // too dense around this query: knn_ring
            if (cnt < kk) continue;     // too sparse: once more with the widest certifiable radius
            // ---- the k+1 nearest are among the candidates of the first histogram bins that hold k+1 of them (bins are
            // a monotone function of d32); thr2 = the largest d32 in those bins, and a candidate with
            // d32 <= thr2 (1 + 1e-6) cannot be excluded (float32 error 3e-7 on either side)
            int incl = hist[lane];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                int o = __shfl_up(incl, off);
                if (lane >= off) incl += o;
            }
            const int bstar = (int)__builtin_ctzll(__ballot(incl >= kk));   // exists: the bins hold cnt >= kk candidates
            float dmy[CAND / 64];
            float m32 = 0.0f;
#pragma unroll
            for (int u = 0; u < CAND / 64; ++u) {
                const int j = lane + 64 * u;
                dmy[u] = j < cnt ? cand[j].w : __builtin_inff();
                if (j < cnt && ringf_bin(dmy[u], inv_thr) <= bstar) m32 = fmaxf(m32, dmy[u]);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m32 = fmaxf(m32, __shfl_xor(m32, off));
            const float thr2 = m32 * (1.0f + 1e-6f);
            int cnt2 = 0;
#pragma unroll
            for (int u = 0; u < CAND / 64; ++u) {
                const bool sel = dmy[u] <= thr2;   // (+inf for the lanes beyond cnt)
                const unsigned long long bal = __ballot(sel);
                if (sel) {
                    const int pos = cnt2 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    if (pos < 64) fin[pos] = cand[lane + 64 * u];
                }
                cnt2 += (int)__builtin_popcountll(bal);
            }
            wave_sync();
            if (cnt2 > 64) break;   // a crowd of near-ties: knn_ring
            double dme = __builtin_inf();
            if (lane < cnt2) {
                const float4 pp = fin[lane];
                dme = dist2_f64(qxd, qyd, qzd, pp.x, pp.y, pp.z);
                cd[lane] = dme;
            }
            wave_sync();
            int rank = 0;
#pragma unroll 4
            for (int j = 0; j < cnt2; ++j) {   // exact: rank = candidates before mine (ties by position)
                const double dj = cd[j];
                rank += (dj < dme || (dj == dme && j < lane)) ? 1 : 0;
            }
            if (lane < cnt2 && rank < kk) out[rank] = dme;
            wave_sync();
            const double kth = out[kk - 1];
            if (!(kth <= rt * rt)) continue;   // (the float32 filter admits a hair more than r_t: then r_t certifies nothing)
            for (int i = lane; i < kk; i += 64) out[i] = __dsqrt_rn(out[i]);
            wave_sync();
            if (lane == 0) {   // out[0] = the query itself
                double sum = pairwise_sum_le128([&](int i) { return out[1 + i]; }, k);
                mean_out[(int)__float_as_uint(qp.w) - q_begin] = __double2float_rn(__ddiv_rn(sum, (double)k));
                kth_emit(gp, kth_out, (int)__float_as_uint(qp.w) - q_begin, kth, qp.x, qp.y, qp.z);
            }
            wave_sync();
            solved = true;
        }
        if (!fetched_next) qp_next = t_next >= 0 ? qpts[fq_next] : qp;
        if (!solved) {
            if (lane == 0) left[nleft] = fq;
            ++nleft;
            if (nleft == RINGF_LEFT) flush_left();
        }
    }
    if (nleft) flush_left();
}

template <int KCAP>
static int launch_ring(gsx_ctx *ctx, const BrickLaunch &a)
{
    static int occ_ring = 0, occ_fast = 0;
    if (!occ_ring) {
        GSX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_ring, knn_ring_kernel<KCAP>, BRICK_THREADS, 0));
        occ_ring = std::max(1, std::min(occ_ring, 8));
        GSX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_fast, knn_ring_fast_kernel<KCAP>, BRICK_THREADS, 0));
        occ_fast = std::max(1, std::min(occ_fast, 8));
    }
    GSX_CHECK(timing_begin(ctx, GSX_T_SOR_FALLBACK));
    const bool fast = ctx->ring_fast;
    if (fast)
        hipLaunchKernelGGL((knn_ring_fast_kernel<KCAP>), dim3(ctx->num_cu * occ_fast), dim3(BRICK_THREADS), 0, ctx->stream, a.gp, a.refs,
                           a.rstart, a.qpts, a.faillist, a.ring2, a.k, (int)a.q_begin, a.n_ref, a.mean_out, a.kth_out);
    hipLaunchKernelGGL((knn_ring_kernel<KCAP>), dim3(ctx->num_cu * occ_ring), dim3(BRICK_THREADS), 0, ctx->stream, a.gp, a.refs,
                       a.rstart, a.qpts, fast ? a.ring2 : a.faillist, a.k, (int)a.q_begin, a.mean_out, a.kth_out, a.heavylist, (int)a.out_count,
                       fast ? 1 : 0);
    GSX_HIP(hipGetLastError());
    GSX_CHECK(timing_end(ctx, GSX_T_SOR_FALLBACK));
    return 0;
}

template <int KCAP>
static int launch_heavy(gsx_ctx *ctx, KnnWs &w, const BrickLaunch &a, int64_t n_ref, unsigned heavy_count)
{
    const int nchunks = div_up(n_ref, HEAVY_CHUNK);
    const int batch = (int)std::max<int64_t>(1, std::min<int64_t>(heavy_count, (256LL << 20) / ((int64_t)nchunks * KCAP * 8)));
    GSX_CHECK(w.heavypart.reserve(sizeof(double) * (size_t)batch * nchunks * KCAP));
    GSX_CHECK(timing_begin(ctx, GSX_T_SOR_FALLBACK));
    for (unsigned first = 0; first < heavy_count; first += (unsigned)batch) {
        const int count = (int)std::min<unsigned>((unsigned)batch, heavy_count - first);
        const int64_t items = (int64_t)count * nchunks;
        const int blocks = (int)std::min<int64_t>(div_up(items, BRICK_THREADS / 64), (int64_t)ctx->num_cu * 8);
        hipLaunchKernelGGL((knn_heavy_scan_kernel<KCAP>), dim3(blocks), dim3(BRICK_THREADS), 0, ctx->stream, a.refs, (int)n_ref,
                           a.qpts, a.heavylist, (int)first, count, a.k, w.heavypart.as<double>());
        hipLaunchKernelGGL((knn_heavy_merge_kernel<KCAP>), dim3(div_up(count, BRICK_THREADS / 64)), dim3(BRICK_THREADS), 0,
                           ctx->stream, a.gp, (int)n_ref, a.qpts, a.heavylist, (int)first, count, a.k, (int)a.q_begin,
                           w.heavypart.as<double>(), a.mean_out, a.kth_out);
    }
    GSX_HIP(hipGetLastError());
    GSX_CHECK(timing_end(ctx, GSX_T_SOR_FALLBACK));
    return 0;
}

static int dispatch_heavy(gsx_ctx *ctx, KnnWs &w, const BrickLaunch &a, int64_t n_ref, unsigned heavy_count)
{
    const int kk = a.k + 1;
    if (kk <= 9) return launch_heavy<9>(ctx, w, a, n_ref, heavy_count);
    if (kk <= 17) return launch_heavy<17>(ctx, w, a, n_ref, heavy_count);
    if (kk <= 26) return launch_heavy<26>(ctx, w, a, n_ref, heavy_count);
    if (kk <= 33) return launch_heavy<33>(ctx, w, a, n_ref, heavy_count);
    if (kk <= 51) return launch_heavy<51>(ctx, w, a, n_ref, heavy_count);
    return launch_heavy<65>(ctx, w, a, n_ref, heavy_count);
}

// list-capacity buckets.  Sorting-network selection (default): the list holds k neighbours, capacities 8, 16, 24,
// 32, 40, 48, 56, 64 (template argument = capacity + 1; the multiples of 8 that are no power of two: round 4, TopNet's
// padded merge).  Bubble-insert selection (phase2_net = 0, kept for A/B): k + 1
// entries incl. the query; 26 and 51 are the CLI's default k=25 and its maximum k=50 (--sor_intensity 10).
static int dispatch_bricks(gsx_ctx *ctx, const BrickLaunch &a, bool mf, bool net)
{
    const int kk = a.k + 1;
    if (net) {
#define GSX_BRICKS(K) (mf ? launch_bricks<K, true, true>(ctx, a) : launch_bricks<K, false, true>(ctx, a))
        // (round 5: the multiples of 4 in between as well -- --sor_intensity 1 ... 10 asks for k = 10, 14, 18, 23, 27, 32, 36, 41, 45, 50
        //  and the default is 25: -5 % of the step at k = 25 with 28 entries instead of 32, profiles/r05_variants.txt)
        if (kk <= 9) return GSX_BRICKS(9);
        if (GSX_CAP4 && kk <= 13) return GSX_BRICKS(13);
        if (kk <= 17) return GSX_BRICKS(17);
        if (GSX_CAP4 && kk <= 21) return GSX_BRICKS(21);
        if (kk <= 25) return GSX_BRICKS(25);   // 24 and 48 entries: the reference CLI's k = 18 ... 24 and 33 ... 48
        if (GSX_CAP4 && kk <= 29) return GSX_BRICKS(29);
        if (kk <= 33) return GSX_BRICKS(33);
        if (GSX_CAP4 && kk <= 37) return GSX_BRICKS(37);
        if (kk <= 41) return GSX_BRICKS(41);   // (40 and 56 entries: -10 % at k = 36, -13 % at 40, -8 % at 56 against the next capacity)
        if (GSX_CAP4 && kk <= 45) return GSX_BRICKS(45);
        if (kk <= 49) return GSX_BRICKS(49);
        if (GSX_CAP4 && kk <= 53) return GSX_BRICKS(53);
        if (kk <= 57) return GSX_BRICKS(57);
        return GSX_BRICKS(65);
#undef GSX_BRICKS
    }
#define GSX_BRICKS(K) (mf ? launch_bricks<K, true, false>(ctx, a) : launch_bricks<K, false, false>(ctx, a))
    if (kk <= 9) return GSX_BRICKS(9);
    if (kk <= 17) return GSX_BRICKS(17);
    if (kk <= 26) return GSX_BRICKS(26);
    if (kk <= 33) return GSX_BRICKS(33);
    if (kk <= 51) return GSX_BRICKS(51);
    return GSX_BRICKS(65);
#undef GSX_BRICKS
}

static int dispatch_ring(gsx_ctx *ctx, const BrickLaunch &a)
{
    const int kk = a.k + 1;
    if (kk <= 9) return launch_ring<9>(ctx, a);
    if (kk <= 17) return launch_ring<17>(ctx, a);
    if (kk <= 26) return launch_ring<26>(ctx, a);
    if (kk <= 33) return launch_ring<33>(ctx, a);
    if (kk <= 51) return launch_ring<51>(ctx, a);
    return launch_ring<65>(ctx, a);
}

// Multi-GPU slab: points [0, n_own) are queries, [n_own, n_own + n_halo) reference-only; kth_out (nullable) receives
// every query's (k+1)-th squared distance so that the caller can certify it against the slab's open faces.
int launch_knn_slab(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride, int64_t n_own,
                    int64_t n_halo, int k, float *mean_out, double *kth_out, const SlabKnn *sk = nullptr);

// csrc/sor_tree.hip: the path for clouds this grid cannot resolve
int launch_knn_tree(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride, int64_t n_ref, int64_t q_begin,
                    int64_t q_count, int k, float *mean_out, double *kth_out, gsx_sor_info *info, int64_t ref_only_from, int share,
                    int nshares, bool guard);

int64_t grid_cell_cap(int64_t n_ref) { return std::max<int64_t>(n_ref / 2, 64) + 64; }

// Sort (x,y,z)[first, first+n) by cell of the grid in gp: `sorted` and `start` (cell_start) are outputs.
static int bin_points(gsx_ctx *ctx, KnnWs &w, const float *x, const float *y, const float *z, int64_t stride, int64_t first,
                      int64_t n, GridParams *gp, unsigned *start, float4 *sorted, int64_t cell_cap = 0, unsigned *cursor = nullptr,
                      int64_t ref_only_from = INT32_MAX, bool hist_done = false)
{
    const bool big_path = cursor != nullptr;  // adaptive mode: oversized buckets are sorted by all workgroups
    unsigned *bk_cnt = w.bkcnt.as<unsigned>();
    unsigned *bk_start = bk_cnt + MAX_BUCKETS;
    unsigned *bk_cursor = bk_start + MAX_BUCKETS + 1;
    float4 *tmp = w.bucketpts.as<float4>();
    const int tiles = (int)std::min<int64_t>(div_up(n, BIN_TILE), (int64_t)ctx->num_cu * 4);
    if (!hist_done)
        hipLaunchKernelGGL(bucket_hist_kernel, dim3(tiles), dim3(256), 0, ctx->stream, x, y, z, stride, (int)first, (int)n, gp,
                           bk_cnt, bk_start, bk_cursor);   // its last workgroup scans the bucket sizes
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(tiles), dim3(SCATTER_THREADS), 0, ctx->stream, x, y, z, stride, (int)first, (int)n,
                       gp, bk_cursor, tmp, (int)std::min<int64_t>(ref_only_from, INT32_MAX));
    if (big_path) GSX_HIP(hipMemsetAsync(start, 0, sizeof(unsigned) * (size_t)(cell_cap + 1), ctx->stream));  // counts of big buckets
    hipLaunchKernelGGL(bucket_sort_kernel, dim3(MAX_BUCKETS), dim3(SORT_THREADS), 0, ctx->stream, gp, bk_start, tmp, sorted, start,
                       big_path ? BIG_BUCKET : 0xffffffffu);
    if (big_path) {
        const int chunks = div_up(n, BIG_CHUNK);
        hipLaunchKernelGGL((big_bucket_kernel<false>), dim3(chunks), dim3(256), 0, ctx->stream, gp, bk_start, (int)n, tmp, sorted,
                           start, cursor);
        hipLaunchKernelGGL(big_bucket_scan_kernel, dim3(MAX_BUCKETS), dim3(256), 0, ctx->stream, gp, bk_start, start, cursor);
        hipLaunchKernelGGL((big_bucket_kernel<true>), dim3(chunks), dim3(256), 0, ctx->stream, gp, bk_start, (int)n, tmp, sorted,
                           start, cursor);
    }
    GSX_HIP(hipGetLastError());
    return 0;
}

// candidate words of a brick's neighbourhood above which it is deferred (a uniform cloud has ~15)

static int knn_grid_level(gsx_ctx *ctx, int level, const float *x, const float *y, const float *z, int64_t stride,
                          int64_t n_ref, int64_t q_begin, int64_t q_count, int k, float *mean_out, double *kth_out,
                          gsx_sor_info *info, int share, int nshares, bool adaptive, float parent_h,
                          int64_t ref_only_from = INT32_MAX, double h_hint = 0.0, const SlabKnn *sk = nullptr)
{
    KnnWs &w = ctx->ws[level];
    if (level == 0) ctx->last_knn_algo = GSX_KNN_GRID;
    w.refined_total = 0;
    const int kk = k + 1;
    const bool anyk = kk > 65;   // k > 64: the list-free exact path (knn_anyk_kernel) instead of knn_brick + ring kernels
    if (kk > ANYK_MAX) GSX_FAIL("sor: k=%d not supported (k must be <= %d)", k, ANYK_MAX - 1);
    // cells the grid may have.  Level 0: about one cell per two points (more would be empty cells to walk).  A refinement
    // level sized by the density probe covers a box that is mostly EMPTY by construction (a blob's tails, the few floaters
    // around a scene), so it gets room for 4 cells per point, within what the two-level sort can address.
    const int64_t cap = h_hint > 0.0 ? std::min<int64_t>(std::max<int64_t>(4 * n_ref, 64), (int64_t)MAX_BUCKETS * MAX_BUCKET_CELLS - 64) + 64
                                     : grid_cell_cap(n_ref);
    // slab mode (multi-GPU): every point is binned once, the halo [ref_only_from, n_ref) is flagged reference-only
    const bool slab = ref_only_from < n_ref;
    if (anyk && (nshares > 1 || slab)) GSX_FAIL("sor: k=%d > 64 is served by the single-GPU path only", k);
    if (anyk) adaptive = false;
    const bool all = (q_begin == 0 && q_count == n_ref) || slab;
    // (a share of the bricks -- the replicated multi-GPU exchange -- refines the deferred bricks of ITS share only)
    adaptive = adaptive && all && !slab && level + 1 < KNN_MAX_LEVELS;
    const int bbox_blocks = std::min(grid_blocks(ctx, n_ref, 8), ctx->num_cu * 4);
    // cell edge h is also the guaranteed search radius: the expected number of points within h is
    // 4.19 * m, and a query falls back to knn_ring when fewer than k+1 are.  m = 0.47 (k+1) puts
    // ~2 (k+1) points inside h (measured optimum at k = 8, 16, 25, 32: profiles/r01_sweep_m.log).
    // A brick's population cells*m should stay <= ~58 so that its queries fit ONE 64-lane batch
    // (at 64 on average 47 % of the bricks need a second one): k = 16 -> m = 7.25, not 8.
    double pts_per_cell = ctx->grid_points_per_cell;
    if (!(pts_per_cell > 0.0)) {
        pts_per_cell = std::max(2.0, 0.47 * (double)(k + 1));
        // a brick's queries must fit ONE 64-lane batch with a margin.  Fewer points per cell = fewer candidates per
        // query in knn_brick but more queries for the ring kernels, which are cheap per query once tens of thousands
        // of them keep every wave busy and latency-bound below that (profiles/r02_variants.txt: 10M k=16 is fastest at
        // 54 / 8 cells, 1M at 58 / 8, 50M k=32 at 52-54 / 4)
        const double fill = n_ref >= 4000000 ? 54.0 : 58.0;
        for (int cells = 8; cells >= 1; cells /= 2)
            if (pts_per_cell * cells > fill && pts_per_cell * cells <= 66.0) pts_per_cell = fill / cells;
        // k = 17 ... 20 (round 4; the reference CLI's --sor_intensity 3 is k = 18): 0.47 (k + 1) points per cell would
        // halve the brick to 2x2x1 cells -- 36 of 64 lanes busy, 12x instead of 8x its points as candidates.  Smaller cells
        // keep the 2x2x2 brick full at the price of ring queries (10M uniform, step ms: k = 18 3.16 -> 2.60 with 175 k ring
        // queries, k = 20 3.17 -> 2.90 with 186 k; from k = 23 on the ring queries cost more than the full lanes save)
        if (k >= 17 && k <= 20) pts_per_cell = std::min(pts_per_cell, fill / 8.0 + 0.375 * (double)std::max(0, k - 18));
        // ... and k = 35 ... 52 (--sor_intensity 7 ... 10: k = 36, 41, 45, 50) keeps 2x2x1 bricks instead of 2x1x1 (10M uniform, step ms:
        // k = 36 6.28 -> 4.52, k = 41 6.10 -> 5.02, k = 45 7.26 -> 5.73, k = 48 7.50 -> 6.3, k = 50 10.3 -> 8.3; k >= 56: the ring
        // queries win -- profiles/r04_variants.txt)
        if (k >= 26 && k <= 31) pts_per_cell = std::min(pts_per_cell, 12.0 * fill / 54.0);   // (k = 27: 3.49 -> 3.36, k = 30: 3.58 -> 3.42)
        if (k >= 35 && k <= 43) pts_per_cell = std::min(pts_per_cell, (k <= 38 ? 13.5 : 14.5) * fill / 54.0);
        // Round 5: with 32 parked words (1024 candidates) for the lists of more than 32 entries the cells of k >= 44 can be larger
        // again -- fewer ring queries (k = 50 at 15 points per cell: 595 k of them, 2.5 of 7.6 ms).  10M uniform, step ms, best of
        // a sweep 15 ... 26 (profiles/r05_variants.txt): k = 45 5.67 -> 5.41 (16), 47 6.01 -> 5.58 (16), 50 7.60 -> 6.37 (22), 52 8.86 ->
        // 6.14 (22), 55 7.17 -> 6.81 (22), 57 7.87 -> 7.22 (22), 60 7.87 -> 7.34 (24); k >= 63: 0.47 (k + 1) as before
        if (k >= 44 && k <= 62) pts_per_cell = std::min(pts_per_cell, k <= 49 ? 16.0 : (k <= 58 ? 22.0 : 24.0));
    }
    GSX_CHECK(w.packed.reserve(sizeof(float4) * (size_t)n_ref));
    GSX_CHECK(w.bucketpts.reserve(sizeof(float4) * (size_t)n_ref));
    GSX_CHECK(w.cellstart.reserve(sizeof(unsigned) * (size_t)(cap + 1)));
    if (!w.bkcnt.p) {  // bucket sizes | bucket starts | bucket cursors; sizes are re-zeroed by bucket_scan_kernel
        GSX_CHECK(w.bkcnt.reserve(sizeof(unsigned) * (3 * MAX_BUCKETS + 8)));
        GSX_HIP(hipMemsetAsync(w.bkcnt.p, 0, sizeof(unsigned) * (3 * MAX_BUCKETS + 8), ctx->stream));
    }
    if (!w.gridparams.p) {   // the arrival tickets inside must start at zero (they reset themselves afterwards)
        GSX_CHECK(w.gridparams.reserve(sizeof(GridParams)));
        GSX_HIP(hipMemsetAsync(w.gridparams.p, 0, sizeof(GridParams), ctx->stream));
    }
    GSX_CHECK(w.bboxpart.reserve(sizeof(float) * 7 * (size_t)bbox_blocks));
    GSX_CHECK(w.faillist.reserve(sizeof(unsigned) * 2 * (size_t)std::max<int64_t>(q_count, 1)));   // fail list | knn_ring_fast's leftovers
    GSX_CHECK(w.extraitems.reserve(sizeof(uint2) * (size_t)(std::max(q_count, slab ? n_ref : q_count) / 64 + 64)));
    if (adaptive) {
        GSX_CHECK(w.deferred.reserve(sizeof(unsigned) * (size_t)(cap + 64)));  // nbricks <= ncells <= cap
        GSX_CHECK(w.heavylist.reserve(sizeof(unsigned) * (size_t)std::max<int64_t>(q_count, 1)));
    }
    if (!all) {
        GSX_CHECK(w.qsorted.reserve(sizeof(float4) * (size_t)q_count));
        GSX_CHECK(w.qcellstart.reserve(sizeof(unsigned) * (size_t)(cap + 1)));
    }
    GridParams *gp = w.gridparams.as<GridParams>();
    float4 *refs = w.packed.as<float4>();
    unsigned *rstart = w.cellstart.as<unsigned>();

    GSX_CHECK(timing_begin(ctx, GSX_T_SOR_BIN));
    // (adaptive mode may still switch the deferral off below, once the histogram is known: defer_words is then cleared on the device)
    GridParamArgs gpa{(int)n_ref, pts_per_cell, (int)cap, ctx->debug_skip, share, nshares, adaptive ? ctx->defer_words : 0, parent_h, gp,
                      ctx->devflags.as<unsigned>(), h_hint, sk ? sk->cert_axis : -1, sk ? sk->cert_lo : 0.0f, sk ? sk->cert_hi : 0.0f,
                      sk ? sk->cert_count : nullptr};
    const bool adaptive_at_bbox = adaptive;
    if (sk && sk->box.b7)
        hipLaunchKernelGGL(grid_params_known_box_kernel, dim3(1), dim3(64), 0, ctx->stream, sk->box, w.bboxpart.as<float>(), gpa);
    else
        hipLaunchKernelGGL(bbox_partial_kernel, dim3(bbox_blocks), dim3(256), 0, ctx->stream, x, y, z, stride, (int)n_ref,
                           w.bboxpart.as<float>(), gpa);   // its last workgroup computes the grid parameters
    GSX_HIP(hipGetLastError());
    if (adaptive) GSX_CHECK(w.qcellstart.reserve(sizeof(unsigned) * (size_t)(cap + 1)));  // free in this mode: the cursors
    // Adaptive mode, first decision: can ONE grid resolve this cloud at all?  The coarse histogram of the two-level sort (one
    // bucket = a column of cells sized for ~4096 points of a uniform cloud) says so after 0.1 ms: a cloud whose fullest bucket
    // holds several times the average (a scene inside a box inflated by floaters: 2400x; Gaussian blobs: 30x+) goes to the
    // Morton-tree path (sor_tree.hip) at once -- no cell size fits it, and refining level by level costs a host round trip
    // and a re-binning per level (clustered 1M: 16.5 ms against 1.3 ms).
    // (slab mode -- one rank's slab + halo of the multi-GPU exchange -- never refines, but an uneven slab still goes to the tree:
    //  the grid would search its dense cells quadratically)
    bool tree_ok = (adaptive || (slab && ctx->adaptive)) && ctx->tree && level == 0 && kk <= 65 && n_ref > k;
    bool hist_done = false;
    if (tree_ok) {
        unsigned *bk_cnt = w.bkcnt.as<unsigned>();
        const int tiles = (int)std::min<int64_t>(div_up(n_ref, BIN_TILE), (int64_t)ctx->num_cu * 4);
        hipLaunchKernelGGL(bucket_hist_kernel, dim3(tiles), dim3(256), 0, ctx->stream, x, y, z, stride, 0, (int)n_ref, gp, bk_cnt,
                           bk_cnt + MAX_BUCKETS, bk_cnt + 2 * MAX_BUCKETS + 1);
        GSX_HIP(hipGetLastError());
        static thread_local std::vector<unsigned> starts(MAX_BUCKETS + 1);
        GSX_HIP(hipMemcpyAsync(starts.data(), bk_cnt + MAX_BUCKETS, sizeof(unsigned) * (MAX_BUCKETS + 1), hipMemcpyDeviceToHost, ctx->stream));
        GSX_HIP(hipStreamSynchronize(ctx->stream));
        unsigned mx = 0, nonzero = 0;
        for (int b = 0; b < MAX_BUCKETS; ++b) {
            const unsigned c = starts[b + 1] >= starts[b] ? starts[b + 1] - starts[b] : 0u;   // (entries past the last bucket repeat the total)
            mx = std::max(mx, c);
            nonzero += c > 0;
        }
        hist_done = true;
        if (nonzero > 0 && (double)mx > 3.0 * (double)n_ref / (double)nonzero) {
            GSX_CHECK(timing_end(ctx, GSX_T_SOR_BIN));
            if (getenv("GSX_TRACE_LEVELS"))
                fprintf(stderr, "[gsx] level 0: fullest bucket %u of %lld points in %u buckets -> tree path\n", mx, (long long)n_ref, nonzero);
            const int rc = launch_knn_tree(ctx, x, y, z, stride, n_ref, q_begin, q_count, k, mean_out, kth_out, info,
                                           slab ? ref_only_from : INT32_MAX, share, nshares, true);
            if (rc != GSX_TREE_UNSUITABLE) return rc;
            // (tens of thousands of points inside one cell of the tree's finest resolution: the refinement below re-scales)
            tree_ok = false;
            ctx->last_knn_algo = GSX_KNN_GRID;
            GSX_CHECK(timing_begin(ctx, GSX_T_SOR_BIN));
        }
        // an even histogram (fullest bucket within 1.5x of the average): no brick can be far over-full, so none is deferred and
        // the host does not have to look at the counters again -- the call costs ONE synchronisation, as before the probe existed
        if (nonzero > 0 && (double)mx <= 1.5 * (double)n_ref / (double)nonzero && nshares == 1) adaptive = false;
    }
    if (adaptive_at_bbox && !adaptive) {   // the grid parameters were written with deferral on
        GSX_HIP(hipMemsetAsync(&gp->defer_words, 0, sizeof(int), ctx->stream));
        GSX_HIP(hipMemsetAsync(&gp->heavy_limit, 0, sizeof(int), ctx->stream));
    }
    GSX_CHECK(bin_points(ctx, w, x, y, z, stride, 0, n_ref, gp, rstart, refs, cap, adaptive ? w.qcellstart.as<unsigned>() : nullptr,
                         slab ? ref_only_from : INT32_MAX, hist_done));
    const float4 *qpts = refs;
    const unsigned *qstart = rstart;
    if (!all) {
        GSX_CHECK(bin_points(ctx, w, x, y, z, stride, q_begin, q_count, gp, w.qcellstart.as<unsigned>(),
                             w.qsorted.as<float4>()));
        qpts = w.qsorted.as<float4>();
        qstart = w.qcellstart.as<unsigned>();
    }
    GSX_CHECK(timing_end(ctx, GSX_T_SOR_BIN));

    BrickLaunch a{gp, refs, rstart, qpts, qstart, k, q_begin, mean_out, kth_out, w.faillist.as<unsigned>(),
                  w.faillist.as<unsigned>() + std::max<int64_t>(q_count, 1),
                  w.extraitems.as<uint2>(), w.deferred.as<unsigned>(), w.heavylist.as<unsigned>(), q_count, n_ref};
    if (anyk) {
        GSX_CHECK(timing_begin(ctx, GSX_T_SOR_KNN));
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((q_count + 3) / 4, (int64_t)ctx->num_cu * 2));
        hipLaunchKernelGGL(knn_anyk_kernel, dim3(blocks), dim3(BRICK_THREADS), 0, ctx->stream, gp, refs, rstart, qpts, (int)q_count, k,
                           (int)q_begin, mean_out, kth_out);
        GSX_HIP(hipGetLastError());
        GSX_CHECK(timing_end(ctx, GSX_T_SOR_KNN));
        if (info) {
            GridParams h2;
            GSX_HIP(hipMemcpyAsync(&h2, gp, sizeof(GridParams), hipMemcpyDeviceToHost, ctx->stream));
            GSX_HIP(hipStreamSynchronize(ctx->stream));
            info->algo = GSX_KNN_GRID;
            info->grid_dim[0] = h2.nx;
            info->grid_dim[1] = h2.ny;
            info->grid_dim[2] = h2.nz;
            info->cell_size = h2.h;
            info->n_cells = (int64_t)h2.nx * h2.ny * h2.nz;
            info->n_bricks = 0;
            info->n_fallback = q_count;
            info->n_exhaustive = h2.exhaustive_count;
            info->n_deferred_bricks = 0;
            info->n_refined = 0;
        }
        return 0;
    }
    GSX_CHECK(dispatch_bricks(ctx, a, ctx->filter_mfma != 0, ctx->phase2_net != 0));

    const bool trace = getenv("GSX_TRACE_LEVELS") != nullptr;
    GridParams hgp;
    bool have_hgp = false;
    if (adaptive) {
        // the only host decision of the pipeline: did any brick ask for a finer grid?
        GSX_HIP(hipMemcpyAsync(&hgp, gp, sizeof(GridParams), hipMemcpyDeviceToHost, ctx->stream));
        GSX_HIP(hipStreamSynchronize(ctx->stream));
        have_hgp = true;
        if (getenv("GSX_TRACE_LEVELS"))
            fprintf(stderr, "[gsx] level %d: n=%lld h=%.5g dims=%dx%dx%d bricks=%d defer_words=%d deferred=%u extra=%u fail=%u\n", level,
                    (long long)n_ref, hgp.h, hgp.nx, hgp.ny, hgp.nz, hgp.nbricks, hgp.defer_words, hgp.deferred_count,
                    hgp.extra_count, hgp.fail_count);
        if (hgp.deferred_count > 0 && !hgp.bad_input && tree_ok && nshares == 1) {   // (a share sees only ITS bricks: the ranks of a
                                                                                    // replicated exchange could decide differently)
            // second decision: the histogram looked even, yet some bricks hold far more than their cells were sized for
            // (density varying inside the buckets).  The tree path takes the whole cloud over; this level's work is lost.
            if (getenv("GSX_TRACE_LEVELS")) fprintf(stderr, "[gsx] level 0: %u deferred bricks -> tree path\n", hgp.deferred_count);
            const int rc = launch_knn_tree(ctx, x, y, z, stride, n_ref, q_begin, q_count, k, mean_out, kth_out, info, INT32_MAX, 0, 1, true);
            if (rc != GSX_TREE_UNSUITABLE) return rc;
            ctx->last_knn_algo = GSX_KNN_GRID;
        }
        if (hgp.deferred_count > 0 && !hgp.bad_input) {
            const unsigned nd = hgp.deferred_count;
            GSX_CHECK(w.cellflag.reserve((size_t)hgp.ncells + 64));
            GSX_CHECK(w.subxyz.reserve(sizeof(float) * 3 * (size_t)n_ref));
            GSX_CHECK(w.submap.reserve(sizeof(unsigned) * 2 * (size_t)n_ref));
            uint8_t *flag = w.cellflag.as<uint8_t>();
            float *sub = w.subxyz.as<float>();
            unsigned *sub_orig = w.submap.as<unsigned>(), *sub_sorted = sub_orig + n_ref;
            // Deferred bricks that are far apart belong to different clusters with different densities:
            // each connected group (bricks within 2 of each other, i.e. overlapping neighbourhoods)
            // gets its own sub-cloud, hence its own bounding box and cell size.  At most 8 groups
            // (the smallest ones are merged into the last).
            std::vector<unsigned> dl(nd);
            GSX_HIP(hipMemcpyAsync(dl.data(), a.deferred, sizeof(unsigned) * nd, hipMemcpyDeviceToHost, ctx->stream));
            GSX_HIP(hipStreamSynchronize(ctx->stream));
            std::vector<std::vector<unsigned>> groups;
            {
                std::unordered_map<unsigned, int> where;  // brick id -> index in dl
                for (unsigned i = 0; i < nd; ++i) where[dl[i]] = (int)i;
                std::vector<int> comp(nd, -1);
                const int nbx = hgp.nbx, nby = hgp.nby, nbz = hgp.nbz;
                for (unsigned s0i = 0; s0i < nd; ++s0i) {
                    if (comp[s0i] >= 0) continue;
                    const int c = (int)groups.size();
                    groups.emplace_back();
                    std::vector<unsigned> stack{s0i};
                    comp[s0i] = c;
                    while (!stack.empty()) {
                        const unsigned i = stack.back();
                        stack.pop_back();
                        groups[c].push_back(dl[i]);
                        const int b = (int)dl[i];
                        const int bz = b / (nbx * nby), by = (b - bz * nbx * nby) / nbx, bx = b - bz * nbx * nby - by * nbx;
                        for (int dz = -2; dz <= 2; ++dz)
                            for (int dy = -2; dy <= 2; ++dy)
                                for (int dx = -2; dx <= 2; ++dx) {
                                    const int x2 = bx + dx, y2 = by + dy, z2 = bz + dz;
                                    if (x2 < 0 || y2 < 0 || z2 < 0 || x2 >= nbx || y2 >= nby || z2 >= nbz) continue;
                                    auto it = where.find((unsigned)((z2 * nby + y2) * nbx + x2));
                                    if (it != where.end() && comp[it->second] < 0) {
                                        comp[it->second] = c;
                                        stack.push_back((unsigned)it->second);
                                    }
                                }
                    }
                }
                std::sort(groups.begin(), groups.end(), [](const auto &l, const auto &r) { return l.size() > r.size(); });
                while (groups.size() > 8) {
                    groups[7].insert(groups[7].end(), groups.back().begin(), groups.back().end());
                    groups.pop_back();
                }
                unsigned o = 0;
                for (auto &g : groups) {
                    std::copy(g.begin(), g.end(), dl.begin() + o);
                    o += (unsigned)g.size();
                }
                if (groups.size() > 1) {  // one group: the device list is already it
                    GSX_HIP(hipMemcpyAsync(a.deferred, dl.data(), sizeof(unsigned) * nd, hipMemcpyHostToDevice, ctx->stream));
                    GSX_HIP(hipStreamSynchronize(ctx->stream));  // dl is a local
                }
            }
            unsigned g_first = 0;
            for (const auto &grp : groups) {
            const unsigned g_count = (unsigned)grp.size();
            const unsigned *g_list = a.deferred + g_first;
            g_first += g_count;
            GSX_CHECK(timing_begin(ctx, GSX_T_SOR_BIN));
            GSX_HIP(hipMemsetAsync(flag, 0, (size_t)hgp.ncells, ctx->stream));
            hipLaunchKernelGGL(reset_refine_kernel, dim3(1), dim3(1), 0, ctx->stream, gp);
            hipLaunchKernelGGL(mark_cells_kernel, dim3(g_count), dim3(64), 0, ctx->stream, gp, g_list, g_count, flag, 0);
            hipLaunchKernelGGL(mark_cells_kernel, dim3(g_count), dim3(64), 0, ctx->stream, gp, g_list, g_count, flag, 1);
            hipLaunchKernelGGL(query_bbox_kernel, dim3(std::min(div_up(n_ref, 256), ctx->num_cu * 4)), dim3(256), 0, ctx->stream, gp,
                               refs, (int)n_ref, flag);
            GSX_HIP(hipGetLastError());
            // The neighbourhood cells reach up to two of THIS level's cells beyond the queries; a stray
            // far point among them (the very outliers that inflated the bounding box) would inflate the
            // finer grid again.  Reference points farther than M from the queries' box are left out, and
            // the finer level's answers are accepted only up to M: M = 4 x the cell edge the finer grid
            // will get, i.e. ~5x its typical (k+1)-th neighbour distance.
            ClipBox clip{};
            unsigned hq_sub_queries = 0;
            int hq_nd3 = 0;
            double h_est = 0.0;   // cell edge the finer level would get from the box volume of the group's queries
            float hq_lo[3] = {0, 0, 0}, hq_hi[3] = {0, 0, 0};
            {
                GridParams hq;
                GSX_HIP(hipMemcpyAsync(&hq, gp, sizeof(GridParams), hipMemcpyDeviceToHost, ctx->stream));
                GSX_HIP(hipStreamSynchronize(ctx->stream));
                double vol = 1.0, amax = 0.0;
                int nd3 = 0;
                for (int ax = 0; ax < 3; ++ax) {
                    const double e = (double)hq.qb_hi[ax] - (double)hq.qb_lo[ax];
                    if (e > 0.0) { vol *= e; ++nd3; }
                    amax = std::max({amax, std::fabs((double)hq.qb_lo[ax]), std::fabs((double)hq.qb_hi[ax])});
                }
                hq_sub_queries = hq.sub_queries;
                hq_nd3 = nd3;
                for (int ax = 0; ax < 3; ++ax) {
                    hq_lo[ax] = hq.qb_lo[ax];
                    hq_hi[ax] = hq.qb_hi[ax];
                }
                const double per = hq.sub_queries > 0 ? vol * pts_per_cell / (double)hq.sub_queries : 0.0;
                h_est = nd3 == 3 ? std::cbrt(per) : (nd3 == 2 ? std::sqrt(per) : (nd3 == 1 ? per : 0.0));
                const double M = std::max(4.0 * h_est, 0.01 * (double)hgp.h);
                const double cert = M * (1.0 - 1e-3) - 4e-7 * amax;  // f32 rounding of lo - M / hi + M
                if (ctx->adaptive == 1 && hq.sub_queries > 0 && M < 2.0 * (double)hgp.h && cert > 0.0) {
                    for (int ax = 0; ax < 3; ++ax) {
                        clip.lo[ax] = std::nextafterf((float)((double)hq.qb_lo[ax] - M), -__builtin_inff());
                        clip.hi[ax] = std::nextafterf((float)((double)hq.qb_hi[ax] + M), __builtin_inff());
                    }
                    clip.r_cert = (float)cert;
                }
                if (getenv("GSX_TRACE_LEVELS"))
                    fprintf(stderr, "[gsx] level %d: group of %u bricks, %u deferred queries in [%g,%g]x[%g,%g]x[%g,%g], clip margin %g\n", level, g_count,
                            hq.sub_queries, hq.qb_lo[0], hq.qb_hi[0], hq.qb_lo[1], hq.qb_hi[1], hq.qb_lo[2], hq.qb_hi[2],
                            (double)clip.r_cert);
            }
            hipLaunchKernelGGL(gather_sub_kernel, dim3(div_up(n_ref, GATHER_CHUNK)), dim3(256), 0, ctx->stream, gp, refs, (int)n_ref,
                               flag, clip, sub, sub_orig, sub_sorted);
            GSX_HIP(hipGetLastError());
            // density probe of the gathered points (see probe_count_kernel): rides on the synchronisation that reads n_sub
            ProbeBox pb{};
            double bin_vol = 0.0;
            unsigned hist_h[32] = {0};
            unsigned *probe_hist = nullptr;
            if (ctx->adaptive == 1 && hq_sub_queries > 0 && hq_nd3 == 3) {
                int G = (int)std::lround(std::cbrt((double)hq_sub_queries / 32.0));
                G = std::max(8, std::min(64, G));
                GSX_CHECK(w.probe.reserve(sizeof(unsigned) * ((size_t)G * G * G + 32)));
                unsigned *counts = w.probe.as<unsigned>();
                probe_hist = counts + (size_t)G * G * G;
                GSX_HIP(hipMemsetAsync(counts, 0, sizeof(unsigned) * ((size_t)G * G * G + 32), ctx->stream));
                pb.g = G;
                bin_vol = 1.0;
                for (int ax = 0; ax < 3; ++ax) {
                    const double e = (double)hq_hi[ax] - (double)hq_lo[ax];
                    pb.lo[ax] = hq_lo[ax];
                    pb.inv[ax] = (float)((double)G / e);
                    bin_vol *= e / (double)G;
                }
                const int pblocks = std::min(div_up(n_ref, 1024), ctx->num_cu * 4);
                hipLaunchKernelGGL(probe_count_kernel, dim3(pblocks), dim3(256), 0, ctx->stream, sub, (int)n_ref, &gp->sub_count, pb, counts);
                hipLaunchKernelGGL(probe_hist_kernel, dim3(pblocks), dim3(256), 0, ctx->stream, sub, (int)n_ref, &gp->sub_count, pb, counts,
                                   probe_hist);
                GSX_HIP(hipGetLastError());
                GSX_HIP(hipMemcpyAsync(hist_h, probe_hist, sizeof(hist_h), hipMemcpyDeviceToHost, ctx->stream));
            }
            unsigned n_sub = 0;
            GSX_HIP(hipMemcpyAsync(&n_sub, &gp->sub_count, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
            GSX_HIP(hipStreamSynchronize(ctx->stream));
            GSX_CHECK(timing_end(ctx, GSX_T_SOR_BIN));
            double h_hint_sub = 0.0;
            constexpr double probe_q = 0.85, probe_shrink = 0.7;   // measured: quantiles 0.5 / 0.7 / 0.85 on four clouds
            if (probe_hist) {
                unsigned long long tot = 0, run = 0;
                for (int b = 0; b < 32; ++b) tot += hist_h[b];
                int bq = -1;
                for (int b = 0; b < 32 && tot; ++b) {
                    run += hist_h[b];
                    if ((double)run >= probe_q * (double)tot) {
                        bq = b;
                        break;
                    }
                }
                if (bq >= 0 && bin_vol > 0.0) {
                    // 85 % of the points sit in probe bins of at most ~1.5 * 2^bq points: cells sized for THAT density are
                    // over-full only for the densest 15 % (a Gaussian's core is 1.4x denser than its 85 % level: no second
                    // refinement), under-full for the tails, whose queries go to the ring kernels
                    const double rho = 1.5 * std::ldexp(1.0, bq) / bin_vol;
                    h_hint_sub = std::cbrt(pts_per_cell / rho);
                    // a grid of that edge over the group's box must fit the cell budget, otherwise the edge would be stretched
                    // to something in between and every brick of a uniform region would be over-full: keep the old sizing then
                    double cells = 1.0;
                    for (int ax = 0; ax < 3; ++ax) cells *= std::floor(((double)hq_hi[ax] - (double)hq_lo[ax] + 2.0 * (double)clip.r_cert) / h_hint_sub) + 1.0;
                    const double budget = (double)std::min<int64_t>(4 * (int64_t)n_sub, (int64_t)MAX_BUCKETS * MAX_BUCKET_CELLS - 64);
                    if (!(cells <= 0.8 * budget)) h_hint_sub = 0.0;
                    // a near-uniform group (the probed edge within 30 % of the box-volume edge) keeps the box-volume edge: the
                    // population per cell is tuned to a few per cent there (0.92x the edge = 13x the ring queries, measured)
                    if (h_hint_sub > probe_shrink * h_est) h_hint_sub = 0.0;
                }
            }
            if (getenv("GSX_TRACE_LEVELS")) fprintf(stderr, "[gsx] level %d: sub-cloud %u points, probed cell edge %g\n", level, n_sub, h_hint_sub);
            if (n_sub == 0) GSX_FAIL("sor: refinement gathered no points for %u deferred bricks", g_count);
            GSX_CHECK(w.submean.reserve(sizeof(float) * (size_t)n_sub));
            GSX_CHECK(w.subkth.reserve(sizeof(double) * (size_t)n_sub));
            GSX_CHECK(knn_grid_level(ctx, level + 1, sub, sub + n_ref, sub + 2 * n_ref, 1, (int64_t)n_sub, 0, (int64_t)n_sub, k,
                                     w.submean.as<float>(), w.subkth.as<double>(), nullptr, 0, 1, true, hgp.h, INT32_MAX, h_hint_sub));
            hipLaunchKernelGGL(merge_sub_kernel, dim3(div_up((int64_t)n_sub, 256)), dim3(256), 0, ctx->stream, gp, sub, (int)n_ref,
                               sub_orig, sub_sorted, w.submean.as<float>(), w.subkth.as<double>(), (int)q_begin, clip.r_cert,
                               mean_out, kth_out, a.faillist);
            GSX_HIP(hipGetLastError());
            w.refined_total += n_sub;
            }
        }
    }
    std::chrono::steady_clock::time_point t_ring0;
    if (trace) {
        GSX_HIP(hipStreamSynchronize(ctx->stream));
        t_ring0 = std::chrono::steady_clock::now();
    }
    // a ring can only meet a huge cell if some brick was too populated for this level, i.e. deferred;
    // otherwise knn_ring keeps every query (no hand-over list, no second host sync)
    const bool heavy_possible = adaptive && have_hgp && hgp.deferred_count > 0;
    if (adaptive && !heavy_possible) GSX_HIP(hipMemsetAsync(&gp->heavy_limit, 0, sizeof(int), ctx->stream));
    GSX_CHECK(dispatch_ring(ctx, a));
    if (heavy_possible) {
        unsigned nheavy = 0;
        GSX_HIP(hipMemcpyAsync(&nheavy, &gp->heavy_count, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        GSX_HIP(hipStreamSynchronize(ctx->stream));
        if (trace) fprintf(stderr, "[gsx] level %d: %u ring queries handed to knn_heavy\n", level, nheavy);
        if (nheavy > 0) GSX_CHECK(dispatch_heavy(ctx, w, a, n_ref, nheavy));
    }
    if (trace) {
        GSX_HIP(hipStreamSynchronize(ctx->stream));
        GridParams h3;
        GSX_HIP(hipMemcpy(&h3, gp, sizeof(GridParams), hipMemcpyDeviceToHost));
        fprintf(stderr, "[gsx] level %d: knn_ring %.3f ms for %u queries (%u exhaustive)\n", level,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_ring0).count(), h3.fail_count,
                h3.exhaustive_count);
    }

    if (info) {
        GridParams h2;
        GSX_HIP(hipMemcpyAsync(&h2, gp, sizeof(GridParams), hipMemcpyDeviceToHost, ctx->stream));
        GSX_HIP(hipStreamSynchronize(ctx->stream));
        if (h2.bad_input) return gsx_ctx_check(ctx);  // reports (and clears) the device flag
        info->algo = GSX_KNN_GRID;
        info->grid_dim[0] = h2.nx; info->grid_dim[1] = h2.ny; info->grid_dim[2] = h2.nz;
        info->cell_size = h2.h;
        info->n_cells = h2.ncells;
        info->n_bricks = h2.nbricks;
        info->n_fallback = h2.fail_count;
        info->n_exhaustive = h2.exhaustive_count;
        info->n_deferred_bricks = h2.deferred_count;
        info->n_refined = (int64_t)w.refined_total;
    } else if (have_hgp && hgp.bad_input && level == 0) {
        return gsx_ctx_check(ctx);
    }
    return 0;
}

int launch_knn_grid(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride, int64_t n_ref,
                    int64_t q_begin, int64_t q_count, int k, float *mean_out, gsx_sor_info *info, int share, int nshares)
{
    return knn_grid_level(ctx, 0, x, y, z, stride, n_ref, q_begin, q_count, k, mean_out, nullptr, info, share, nshares,
                          ctx->adaptive != 0, 0.0f);
}

// sk (optional): the certificate evaluated inside the grid path's kernels (kth_out is then only written by the tree path:
// ctx->last_knn_algo says which ran) and a box the caller already knows
int launch_knn_slab(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride, int64_t n_own,
                    int64_t n_halo, int k, float *mean_out, double *kth_out, const SlabKnn *sk)
{
    return knn_grid_level(ctx, 0, x, y, z, stride, n_own + n_halo, 0, n_own, k, mean_out, kth_out, nullptr, 0, 1, false, 0.0f,
                          n_own, 0.0, sk);
}

}  // namespace gsx
