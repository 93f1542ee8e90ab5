// sor_tree.hip -- EXACT k-nearest-neighbour mean distance on a Morton-ordered cloud: the path for clouds whose density varies
// by orders of magnitude (Gaussian blobs, a dense scene inside a box inflated by far floaters), which one uniform grid
// cannot resolve and which sor_grid.hip's level-by-level refinement serves with a host round trip per level.
//
// Replaces the same reference code as sor_grid.hip (data_processor.py:160-173: cKDTree build + query(k+1) + row mean; the
// approximate Taichi kernel gpu_ops.py:98-176) with the same bits in the output; it is the device-side counterpart of
// what cKDTree is for the reference: a space partition that follows the data.  No host synchronisation anywhere.
//
//   tree_bbox            bounding box -> origin, fine cell edge s = extent / 2^21                      (one pass)
//   tree_keys            63-bit Morton key of every point's fine cell                                  (one pass)
//   rocprim radix sort   (key, index) pairs
//   tree_gather, tree_samples
//                        float4 {x, y, z, index} in key order; the last key of every 32-key block (cache resident: the range
//                        look-ups below search it first)
//   tree_leaf_flags / tree_leaf_compact
//                        LEAVES: the largest nodes of the implicit binary radix tree (a node = all keys sharing the bits
//                        above bit level b: a box with sides 1:1:1, 2:1:1 or 2:2:1) that hold at most 64 points.  Found
//                        without building a tree: node(i, b) holds more than 64 points iff keys j and j+64 agree above b
//                        for some j in [i-64, i], so the smallest such b is a sliding-window minimum over the sorted keys.
//   knn_leaf             one WAVE per leaf: its <= 64 points are the queries (one per lane, contiguous in memory); the
//                        candidates are the points of the leaf's box grown by one CELL on every side, a cell being the cube
//                        of half the leaf's longest side: at most 4x4x4 cells, each a contiguous key range found by one
//                        lane's binary search.  From there on it is knn_brick (sor_grid.hip): bf16-split MFMA filter
//                        (phase 1), per-lane walk of the mask words with exact float64 distances and sorting-network
//                        selection (phase 2), numpy's summation order in the epilogue.  A query is exact iff its k-th
//                        neighbour is nearer than the nearest face of the searched box that has space behind it.
//   knn_tree_near        the queries knn_leaf could not certify (the rim of an object, leaves of ~5 points per cell): one wave
//                        per query, the ball of a radius to try covered by key-range cells, the points inside collected in
//                        LDS, the k nearest taken by rank.
//   knn_tree_query       what is left (a ball that reaches into a much denser region, sparse surroundings): nearest-first
//                        descent of the implicit octree below the cells that cover the ball, pruned by the running k-th
//                        distance; the list lives one entry per lane.
// Adaptive mode (sor_grid.hip: knn_grid_level) routes a cloud here when the coarse histogram of the grid's own sort is uneven.
//
// Exactness of the geometry: a point's fine cell is floor((x - o) * inv_s) evaluated in float64 -- monotone in x -- so all
// points of cells >= c along an axis lie at x >= o + c*s up to ~1e-15 relative; every plane distance used as a guarantee is
// reduced by TreeParams::slack (1e-14 of the cloud's magnitude), far more than those roundings.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "gsx_common.h"
#include "knn_common.h"
#include "knn_mfma.h"
#include "sor_grid_params.h"

namespace gsx {

constexpr int TB = 21;                 // key bits per axis
constexpr int LEAF_CAP = 64;           // points per leaf for k <= 16 = lanes of a wave; more for larger k (tree_leaf_cap_for)
constexpr int LEAF_CAP_MAX = 256;
constexpr int TREE_THREADS = 256;      // 4 independent waves per workgroup
#ifndef GSX_TWCAP
#define GSX_TWCAP 28
#endif
#ifndef GSX_LEAF_WAVES17
#define GSX_LEAF_WAVES17 5
#endif
#ifndef GSX_LEAF_HB   // candidate gathers in flight per lane in knn_leaf's phase 2
#define GSX_LEAF_HB 4
#endif
#ifndef GSX_TQ_SCAN_FULL   // knn_tree_query: points of a node that is scanned block-wise instead of split once the list is full
#define GSX_TQ_SCAN_FULL 4096   // measured 2048 .. 65536 on four clouds: one box load per lane covers the node
#endif
#ifndef GSX_TREE_ABL   // profiling builds only (results become wrong): 1 = no phase 2, 2 = no phase 1, 4 = no range look-ups, 16 = no word count, 32 = no cbrt
#define GSX_TREE_ABL 0
#endif
#ifndef GSX_TWCAP_BIG   // ... for lists of more than 32 entries (round 5: 32 -- fewer filter passes over the 96 / 128-point leaves, -2 ... -4 % at k = 36 ... 64)
#define GSX_TWCAP_BIG 32
#endif
static_assert(GSX_TWCAP <= 32 && GSX_TWCAP_BIG <= 32, "the non-empty-word mask of a batch is one 32-bit register");
constexpr int TWCAP = GSX_TWCAP;              // mask words parked in LDS per wave (7 KiB): 896 candidates per single-drain batch
constexpr int LEAF_TILE = 1024;        // points per workgroup of the leaf-flag kernels
constexpr int TREE_CAND_LIMIT = 4096;   // a leaf whose searched box holds more points (x leaf capacity / 64) hands its queries to knn_tree_near
                                        // (2048 until the filter ran in passes: blobs 10M k = 25 8.08 -> 7.85 ms, k = 50 16.5 -> 16.0; 8192: no better)
constexpr int KEY_BLOCK = 32;          // one key in 32 is copied to a small array (2.5 MB at 10M points: cache resident) that the
                                       // range look-ups search first; only the last 5 steps touch the 80 MB key array
constexpr int TQ_STACK = 256;          // pending nodes of a descent (best-first: the frontier of the ball, typically a few dozen)
constexpr int TQ_SCAN = 256;           // a node with at most this many points is scanned, not split
constexpr unsigned TREE_OVERFULL_LIMIT = 32768;   // points in one fine cell above which adaptive mode does not use this path (their
                                                   // search is quadratic in that number: the key resolution, extent / 2^21, is exhausted)
constexpr int TQ_DENSE = 4096;         // points of one cover cell knn_tree_near takes itself, block boxes first (more: the pruned descent;
                                       // 2^18 measured: 5x slower on blobs -- the ball fills the buffer and the query is handed on anyway)
#ifndef GSX_TQ_FLIGHT
#define GSX_TQ_FLIGHT 4
#endif
constexpr int TQ_FLIGHT = GSX_TQ_FLIGHT;   // 64-point blocks a descent has in flight while it scans a node (8 and 16 measured in round 5: no change --
                                           // the launch lasts as long as its slowest query, and that query's time is its ~70 splits, not its scans)
constexpr int TQ_CAND = 256;           // candidates inside the search ball a wave collects before it ranks them

struct TreeParams {
    double ox, oy, oz;     // origin = per-axis minimum
    double s, inv_s;       // fine cell edge, 1/s
    double slack;          // subtracted from every guaranteed plane distance
    int n;
    unsigned bad_input;    // non-finite coordinates: the kernels do nothing, knn_tree_query fills the output with NaN
    unsigned nleaves;
    unsigned fail_count;
    unsigned ticket_bbox;  // self-resetting arrival ticket of tree_bbox_kernel
    unsigned pad;
    unsigned fail2_count;  // queries knn_tree_near handed on to knn_tree_query
    unsigned max_leaf;     // points of the fullest leaf (> 64: more than 64 points in ONE fine cell -- the key resolution is exhausted)
    unsigned defer_why[8];   // knn_tree_near's reasons for handing a query on (trace output only): 0 too many cells, 1 dense cell,
                             // 2 buffer full, 3 / 4 fewer than k points inside the last radius tried / inside a known bound, 5-7 attempt
    unsigned probe_ticket;     // self-resetting: the last wave of tree_probe_kernel picks the scale
    unsigned probe_hist[32];   // density probe: samples by the fractional octave of their ideal leaf edge (tree_probe_kernel)
    unsigned leaf_ctr[8 * 32];
    unsigned fail_ctr[8 * 32];
    unsigned fail2_ctr[8 * 32];
};

__device__ __forceinline__ unsigned long long spread21(unsigned v)   // bit t -> bit 3t
{
    unsigned long long x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffULL;
    x = (x | x << 16) & 0x1f0000ff0000ffULL;
    x = (x | x << 8) & 0x100f00f00f00f00fULL;
    x = (x | x << 4) & 0x10c30c30c30c30c3ULL;
    x = (x | x << 2) & 0x1249249249249249ULL;
    return x;
}
__device__ __forceinline__ unsigned compact21(unsigned long long x)   // bit 3t -> bit t
{
    x &= 0x1249249249249249ULL;
    x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ULL;
    x = (x ^ (x >> 4)) & 0x100f00f00f00f00fULL;
    x = (x ^ (x >> 8)) & 0x1f0000ff0000ffULL;
    x = (x ^ (x >> 16)) & 0x1f00000000ffffULL;
    x = (x ^ (x >> 32)) & 0x1fffffULL;
    return (unsigned)x;
}
__device__ __forceinline__ unsigned long long morton63(unsigned ix, unsigned iy, unsigned iz)
{
    return spread21(ix) | (spread21(iy) << 1) | (spread21(iz) << 2);
}
// fine cell of a coordinate (monotone in v; v >= o for every point of the cloud)
__device__ __forceinline__ unsigned fine_cell(float v, double o, double inv_s)
{
    const double t = __dmul_rn(__dsub_rn((double)v, o), inv_s);
    const int c = (int)t;
    return (unsigned)min(max(c, 0), (1 << TB) - 1);
}

// ---------------------------------------------------------------- bounding box -> TreeParams
__global__ __launch_bounds__(256) void tree_bbox_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ z, int64_t stride, int n, float *__restrict__ part,
                                                        TreeParams *__restrict__ tp, unsigned *__restrict__ devflags, double scale)
{
    __shared__ float red[7][4];
    __shared__ unsigned s_last;
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    float bad = 0.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float v[3] = {x[(int64_t)i * stride], y[(int64_t)i * stride], z[(int64_t)i * stride]};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], v[a]);
            mx[a] = fmaxf(mx[a], v[a]);
            bad = (fabsf(v[a]) < __builtin_inff()) ? bad : 1.0f;
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
        __hip_atomic_store(&part[blockIdx.x * 7 + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&tp->ticket_bbox, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    // the last workgroup to arrive folds the partial boxes (its first wave) and writes the parameters
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    float v[7] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff(),
                  -__builtin_inff()};
    for (int i = lane; i < (int)gridDim.x; i += 64)
#pragma unroll
        for (int a = 0; a < 7; ++a) {
            const float t = __hip_atomic_load(&part[i * 7 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[a] = a < 3 ? fminf(v[a], t) : fmaxf(v[a], t);
        }
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(v[a], off);
            v[a] = a < 3 ? fminf(v[a], o) : fmaxf(v[a], o);
        }
    // zero the work counters of this call (every lane takes a few)
    for (int i = lane; i < 8 * 32; i += 64) {
        tp->leaf_ctr[i] = 0;
        tp->fail_ctr[i] = 0;
        tp->fail2_ctr[i] = 0;
    }
    if (lane < 32) tp->probe_hist[lane] = 0;
    if (lane != 0) return;
    tp->ticket_bbox = 0;
    const bool isbad = v[6] != 0.0f || n <= 0;
    double emax = 0.0, mag = 0.0;
    for (int a = 0; a < 3; ++a) {
        emax = fmax(emax, (double)v[3 + a] - (double)v[a]);
        mag = fmax(mag, fmax(fabs((double)v[a]), fabs((double)v[3 + a])));
    }
    // 2^21 fine cells along the longest axis; the 2^-18 head room keeps the maximum inside the last cell
    const double s = (emax > 0.0 && emax < 1e300) ? emax * (1.0 + 0x1p-18) * 0x1p-21 * scale : 1.0;
    tp->ox = isbad ? 0.0 : (double)v[0];
    tp->oy = isbad ? 0.0 : (double)v[1];
    tp->oz = isbad ? 0.0 : (double)v[2];
    tp->s = s;
    tp->inv_s = 1.0 / s;
    tp->slack = 1e-14 * (mag + emax);
    tp->n = n;
    tp->bad_input = v[6] != 0.0f ? 1u : 0u;
    tp->nleaves = 0;
    tp->fail_count = 0;
    tp->fail2_count = 0;
    tp->max_leaf = 0;
    for (int i = 0; i < 8; ++i) tp->defer_why[i] = 0;
    if (v[6] != 0.0f) atomicOr(devflags, 1u);
}

// ---------------------------------------------------------------- density probe: the scale of the key grid
// The leaves are nodes of a BINARY radix tree, so their boxes come in three shapes -- 1:1:1, 2:1:1, 2:2:1 -- and a cloud
// whose points mostly share one density gets ONE of them everywhere: which, is an accident of extent / 2^21.  Measured
// on the 10M scene + floaters (profiles/r04_tree_scale.txt; knn_leaf + fallback, ms): cubes of ~50 points 3.60, 2:2:1
// boxes of ~41 3.70, 2:1:1 boxes of ~36 (what extent / 2^21 happened to give) 4.38, 2:1:1 boxes of ~55 6.41 (18x their
// points as candidates: two drains per leaf).  The fine cell edge is free up to a factor of two (the keys keep 20 of
// their 21 bits per axis), so: estimate the local density at 4096 sample points (8th neighbour WITHIN the sample), express
// each as the fractional octave t of the cube edge that would hold 64 points, and stretch the grid by the 2^delta that
// puts the most samples where leaves are cheap.  A heuristic on the partition only: every result stays certified.
constexpr int PROBE_S = 4096, PROBE_M = 8, PROBE_BINS = 32;
constexpr int PROBE_MIN_N = 1 << 18;   // smaller clouds: the probe's three launches cost more than a better shape gains

__global__ __launch_bounds__(256) void tree_probe_gather_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                const float *__restrict__ z, int64_t stride, int n,
                                                                float4 *__restrict__ smp)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= PROBE_S) return;
    const int64_t i = ((int64_t)j * n / PROBE_S) * stride;
    smp[j] = make_float4(x[i], y[i], z[i], 0.0f);
}

// Where the samples' t concentrates (the mode of the histogram, +-1/8 octave holding >= 40 % of them: a cloud with one
// dominant density), the grid is stretched so that the mode lands on the cheapest leaf THIS k can use.  A leaf's searched
// box is the leaf grown by one cell c (half its longest side), and a query is certified iff its k-th neighbour is nearer
// than the box's faces, so the shape must leave  c / r_k = (n / (a k))^(1/3) >= ~1.12  (n points per leaf; a = 1.91 for a
// cube, 0.955 for 2:2:1, 0.477 for 2:1:1) -- below it the fallback kernels get every rim query: at k = 32 cubes of 50 points
// (ratio 0.94) send 1.1 M of 10 M queries there, 11.2 ms against 7.2 with 2:2:1 boxes (profiles/r04_tree_scale.txt).
// Candidates per query grow 8 : 12 : 18 with the shape, n stays clear of 64 (a node a few points above it splits into
// halves whose margin is too small) and of two drains (2:1:1, n > 48):
//   k <= 16: cubes of ~50 points (t = 0.12) | k <= 32: 2:2:1 boxes of ~48 (t = 0.805) | k <= 64: 2:1:1 boxes of ~38 (t = 0.59)
// The estimate reads ~0.065 octave high on the 10M scene (samples near the rim of the dense part see half-empty balls;
// E[log V_8] != log E): calibrated out.  No concentration (blobs whose density varies continuously: every t equally
// likely) -> nothing to choose.
constexpr float PROBE_BIAS_T = 0.065f;
__host__ __device__ constexpr float probe_target_t(int k) { return k <= 16 ? 0.12f : (k <= 32 ? 0.805f : 0.59f); }

__device__ __forceinline__ void tree_pick_scale(TreeParams *__restrict__ tp, int lane, const unsigned *hist, int k)   // one whole wave; k: per 64 points of leaf capacity
{
    const int l = lane & (PROBE_BINS - 1);
    const float h_mine = (float)hist[l];
    tp->probe_hist[l] = hist[l];   // (kept for the trace output)
    float win = 0.0f, mom = 0.0f, total = h_mine;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) total += __shfl_xor(total, off);
#pragma unroll
    for (int j = -4; j <= 4; ++j) {
        const float h = __shfl(h_mine, (l + j) & (PROBE_BINS - 1));
        win += h;
        mom += h * (float)j;
    }
    float best = win;
    int arg = l;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oa = __shfl_xor(arg, off);
        if (ob > best || (ob == best && oa < arg)) {
            best = ob;
            arg = oa;
        }
    }
    if (lane != arg) return;
    tp->probe_ticket = 0;
    if (total < PROBE_S / 8 || win < 0.40f * total) return;   // too few usable samples / no dominant density
    float delta = ((float)l + 0.5f + mom / win) / PROBE_BINS - PROBE_BIAS_T - probe_target_t(k);
    delta -= floorf(delta);
    const double s = tp->s * exp2((double)delta);
    tp->s = s;
    tp->inv_s = 1.0 / s;
}

// A workgroup stages all samples in LDS (64 KiB); each of its waves takes PROBE_PER_WAVE of them at once: every lane keeps the 4 nearest
// of its 64 candidates per sample, the wave pops the 8 nearest of those.  The last wave to finish turns the histogram into
// the scale.  (One sample per wave straight from L2 was a chain of 64 dependent loads: 90 us.)
constexpr int PROBE_PER_WAVE = 2;   // samples a wave takes at once (1: 37 us, 2: 28 us, 4: 31 us per launch)
__global__ __launch_bounds__(256) void tree_probe_kernel(const float4 *__restrict__ smp, int n, int k, float lg_cap, TreeParams *__restrict__ tp,
                                                         unsigned char *__restrict__ bins /* [PROBE_S]: histogram bin of every sample, 0xff = none */)
{
    __shared__ unsigned s_hist[PROBE_BINS];
    __shared__ unsigned s_last;
    __shared__ float4 s_smp[PROBE_S];
    {   // (all 16 loads of a thread in flight: rolled, the loop waited for each in turn -- 16 L2 round trips, 45 us)
        float4 t[PROBE_S / 256];
#pragma unroll
        for (int i = 0; i < PROBE_S / 256; ++i) t[i] = smp[i * 256 + threadIdx.x];
#pragma unroll
        for (int i = 0; i < PROBE_S / 256; ++i) s_smp[i * 256 + threadIdx.x] = t[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int q0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * PROBE_PER_WAVE;
    const float inf = __builtin_inff();
    float4 me[PROBE_PER_WAVE];
    float a[PROBE_PER_WAVE][4];
#pragma unroll
    for (int v = 0; v < PROBE_PER_WAVE; ++v) {
        me[v] = s_smp[q0 + v];
        a[v][0] = a[v][1] = a[v][2] = a[v][3] = inf;
    }
#pragma unroll 4
    for (int c = lane; c < PROBE_S; c += 64) {
        const float4 p = s_smp[c];
#pragma unroll
        for (int v = 0; v < PROBE_PER_WAVE; ++v) {
            const float dx = p.x - me[v].x, dy = p.y - me[v].y, dz = p.z - me[v].z;
            float d = dx * dx + dy * dy + dz * dz;
            d = (c == q0 + v || !(d >= 0.0f)) ? inf : d;
            a[v][3] = fminf(a[v][3], fmaxf(a[v][2], d));
            a[v][2] = fminf(a[v][2], fmaxf(a[v][1], d));
            a[v][1] = fminf(a[v][1], fmaxf(a[v][0], d));
            a[v][0] = fminf(a[v][0], d);
        }
    }
    const float lg_s = __log2f((float)tp->s);
    const float lg_c = __log2f((float)(PROBE_M - 1) / 4.18879f * ((float)n / PROBE_S));
    const bool bad = tp->bad_input != 0;
#pragma unroll
    for (int v = 0; v < PROBE_PER_WAVE; ++v) {
        float r2 = 0.0f;
        for (int round = 0; round < PROBE_M; ++round) {
            float m = a[v][0];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off));
            r2 = m;
            if (a[v][0] == m) {
                a[v][0] = a[v][1];
                a[v][1] = a[v][2];
                a[v][2] = a[v][3];
                a[v][3] = inf;
            }
        }
        if (lane == 0) {
            unsigned char bin = 0xff;
            if (!bad && r2 > 0.0f && r2 < inf) {   // (duplicates: no finite density)
                // rho = (m - 1) / (4/3 pi r^3) * n / S   (E[1/V_m] = rho / (m - 1));   the cube of edge s * 2^u holds 64 points:
                // u = log2(cbrt(64 / rho) / s) = (6 - log2 rho) / 3 - log2 s      (f32 logarithms: +-1e-6 octave)
                const float u = (lg_cap - lg_c + 1.5f * __log2f(r2)) * (1.0f / 3.0f) - lg_s;   // (lg_cap = 6: 64 points; a leaf holds 2^lg_cap)
                if (u > -64.0f && u < 64.0f) bin = (unsigned char)min(PROBE_BINS - 1, (int)((u - floorf(u)) * PROBE_BINS));
            }
            // (a byte per sample, counted by the last workgroup: 4096 atomics on one cache line took 45 us -- the chip
            //  serialises them at ~90 per us, see WorkQueue in sor_grid_params.h)
            __hip_atomic_store(&bins[q0 + v], bin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(&tp->probe_ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    if (threadIdx.x < PROBE_BINS) s_hist[threadIdx.x] = 0;
    __syncthreads();
    if (!s_last) return;
    for (int i = threadIdx.x; i < PROBE_S; i += 256) {
        const unsigned char b = __hip_atomic_load(&bins[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b < PROBE_BINS) atomicAdd(&s_hist[b], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) tree_pick_scale(tp, lane, s_hist, k);
}

__global__ __launch_bounds__(256) void tree_keys_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ z, int64_t stride, int n,
                                                        const TreeParams *__restrict__ tp, unsigned long long *__restrict__ keys,
                                                        unsigned *__restrict__ vals)
{
    const double ox = tp->ox, oy = tp->oy, oz = tp->oz, inv_s = tp->inv_s;
    const bool bad = tp->bad_input != 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned long long key = 0;
        if (!bad)
            key = morton63(fine_cell(x[(int64_t)i * stride], ox, inv_s), fine_cell(y[(int64_t)i * stride], oy, inv_s),
                           fine_cell(z[(int64_t)i * stride], oz, inv_s));
        keys[i] = key;
        vals[i] = (unsigned)i;
    }
}

// points in key order + the tight bounding box of every 64 consecutive ones (two float4: minima, maxima): what
// knn_tree_query prunes with below a node's own box, which is the box of the NODE -- half of it may be empty space beyond the
// face of an object.  A wave handles 64 consecutive points (256-thread workgroups, strides that are multiples of 256).
__global__ __launch_bounds__(256) void tree_gather_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                          const float *__restrict__ z, int64_t stride, int n,
                                                          const unsigned *__restrict__ order, float4 *__restrict__ refs,
                                                          float4 *__restrict__ boxes, int ref_only_from)
{
    const int npad = (n + 63) & ~63;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += gridDim.x * blockDim.x) {
        const unsigned v = order[i < n ? i : n - 1];   // (the lanes past the end repeat the last point: harmless for the box)
        const unsigned tag = v | ((int)v >= ref_only_from ? 0x80000000u : 0u);
        const float px = x[(int64_t)v * stride], py = y[(int64_t)v * stride], pz = z[(int64_t)v * stride];
        if (i < n) refs[i] = make_float4(px, py, pz, __uint_as_float(tag));
        float lo[3] = {px, py, pz}, hi[3] = {px, py, pz};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
                hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
            }
        if ((threadIdx.x & 63) == 0) {
            boxes[2 * (i >> 6)] = make_float4(lo[0], lo[1], lo[2], 0.f);
            boxes[2 * (i >> 6) + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
        }
    }
}

__global__ __launch_bounds__(256) void tree_samples_kernel(const unsigned long long *__restrict__ keys, int n,
                                                           unsigned long long *__restrict__ samples)
{
    const int nblk = (n + KEY_BLOCK - 1) / KEY_BLOCK;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nblk; j += gridDim.x * blockDim.x)
        samples[j] = keys[min(j * KEY_BLOCK + KEY_BLOCK - 1, n - 1)];   // the last key of block j
}

// ---------------------------------------------------------------- leaves
// flags[i] = bit level of the leaf of sorted point i | 0x80 if i is the leaf's first point; tilecnt[t] = leaves starting in tile t.
// cap = points a leaf may hold (64 ... LEAF_CAP_MAX).  The window minimum over cap + 1 entries is taken in two steps: the minima
// of all 64-entry windows by six doubling passes through LDS, then ceil((cap + 1) / 64) of those (overlapping where cap + 1 is
// no multiple of 64) -- the plain loop was 65 LDS reads per point at cap = 64 and would be 257 at 256.
__global__ __launch_bounds__(256) void tree_leaf_flags_kernel(const unsigned long long *__restrict__ keys, int n, int cap,
                                                              unsigned char *__restrict__ flags, unsigned *__restrict__ tilecnt)
{
    constexpr int SPAN = LEAF_TILE + LEAF_CAP_MAX + 64;
    __shared__ unsigned char buf[2][SPAN + 64];   // [t] belongs to sorted index t0 - cap + t
    __shared__ unsigned wsum[4];
    const int t0 = blockIdx.x * LEAF_TILE;
    const int span = LEAF_TILE + cap;   // entries that exist; the rest of the buffers reads as "no window"
    for (int t = threadIdx.x; t < SPAN + 64; t += 256) {
        const long long j = (long long)t0 - cap + t;
        int v = 64;   // no window of cap + 1 points starts here
        if (t < span && j >= 0 && j + cap < n) {
            const unsigned long long d = keys[j] ^ keys[j + cap];
            v = d ? 64 - __builtin_clzll(d) : 0;   // smallest bit level at which j and j + cap share a node
        }
        buf[0][t] = buf[1][t] = (unsigned char)v;
    }
    __syncthreads();
    int cur = 0;
#pragma unroll
    for (int st = 1; st < 64; st *= 2) {   // buf[cur][t] = min of the 2*st entries from t on
        for (int t = threadIdx.x; t < SPAN; t += 256) buf[cur ^ 1][t] = min(buf[cur][t], buf[cur][min(t + st, SPAN + 63)]);
        __syncthreads();
        cur ^= 1;
    }
    const unsigned char *w64 = buf[cur];   // minimum over [t, t + 64)
    unsigned heads = 0;
#pragma unroll
    for (int u = 0; u < LEAF_TILE / 256; ++u) {
        const int t = threadIdx.x + 256 * u;
        const int i = t0 + t;
        if (i < n) {
            int split = (int)w64[t + cap + 1 - 64];   // smallest bit level at which the node of i holds more than cap points
            for (int d = 0; d + 64 <= cap; d += 64) split = min(split, (int)w64[t + d]);
            const int bl = max(split - 1, 0);   // split == 0: more than cap points in one fine cell -- an over-full leaf
            const bool head = i == 0 || (keys[i] >> bl) != (keys[i - 1] >> bl);
            flags[i] = (unsigned char)(bl | (head ? 0x80 : 0));
            heads += head ? 1u : 0u;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) heads += __shfl_xor(heads, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = heads;
    __syncthreads();
    if (threadIdx.x == 0) tilecnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void tree_leaf_compact_kernel(const unsigned char *__restrict__ flags, int n,
                                                                const unsigned *__restrict__ tileoff, const unsigned *__restrict__ tilecnt,
                                                                unsigned *__restrict__ leafstart, unsigned char *__restrict__ leafbl,
                                                                TreeParams *__restrict__ tp)
{
    __shared__ unsigned wsum[4];
    const int t0 = blockIdx.x * LEAF_TILE;
    const int i0 = t0 + threadIdx.x * (LEAF_TILE / 256);   // 4 consecutive points per thread: leaves stay in key order
    unsigned char f[LEAF_TILE / 256];
    unsigned cnt = 0;
#pragma unroll
    for (int u = 0; u < LEAF_TILE / 256; ++u) {
        f[u] = i0 + u < n ? flags[i0 + u] : 0;
        cnt += f[u] >> 7;
    }
    unsigned inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(inc, off);
        if ((int)(threadIdx.x & 63) >= off) inc += o;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wsum[w] = inc;
    __syncthreads();
    unsigned base = tileoff[blockIdx.x];
    for (int i = 0; i < w; ++i) base += wsum[i];
    unsigned at = base + inc - cnt;
#pragma unroll
    for (int u = 0; u < LEAF_TILE / 256; ++u)
        if (f[u] & 0x80) {
            leafstart[at] = (unsigned)(i0 + u);
            leafbl[at] = f[u] & 0x7f;
            ++at;
        }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const unsigned total = tileoff[blockIdx.x] + tilecnt[blockIdx.x];
        leafstart[total] = (unsigned)n;
        tp->nleaves = total;
    }
}

// key range [a, b) of the cell `code` at octree level Lc (every lane its own cell; lanes with want == false get an empty range).
// Start: a binary lower bound, first over the sampled keys, then inside the 32-key block found.  End: galloped from the start --
// a cell holds a handful of points, so two or three probes next to the start replace a second 24-step search.  (The three
// KNN kernels are bound by the NUMBER of these scattered 8-byte loads, not by the length of the chain: a 4-ary search -- half
// the steps, 1.5x the loads -- made every one of them 5-10 % slower.)
__device__ __forceinline__ void cell_range(const unsigned long long *__restrict__ keys, const unsigned long long *__restrict__ samples,
                                           int n, int nblk, bool want, unsigned long long code, int Lc, unsigned &a, unsigned &b)
{
    const unsigned long long klo = code << (3 * Lc), khi = (code + 1) << (3 * Lc);   // khi may be 2^63: above every key
    unsigned a0 = 0, a1 = want ? (unsigned)nblk : 0u;
    while (__any(a0 < a1)) {
        if (a0 < a1) {
            const unsigned mid = (a0 + a1) >> 1;
            if (samples[mid] < klo) a0 = mid + 1u; else a1 = mid;
        }
    }
    // (block index nblk: every key is smaller, the bound is n)
    a0 *= KEY_BLOCK;
    a1 = want ? min(a0 + KEY_BLOCK, (unsigned)n) : a0;
    if (a0 > (unsigned)n) a0 = a1 = (unsigned)n;
    while (__any(a0 < a1)) {
        if (a0 < a1) {
            const unsigned mid = (a0 + a1) >> 1;
            if (keys[mid] < klo) a0 = mid + 1u; else a1 = mid;
        }
    }
    unsigned b0 = a0, b1 = a0, step = 4u;
    bool act = want && a0 < (unsigned)n;
    while (__any(act)) {
        if (act) {
            const unsigned p = b0 + step - 1u;
            if (p >= (unsigned)n) {
                b1 = (unsigned)n;
                act = false;
            } else if (keys[p] < khi) {
                b0 = p + 1u;
                step <<= 2;
            } else {
                b1 = p;
                act = false;
            }
        }
    }
    while (__any(b0 < b1)) {
        if (b0 < b1) {
            const unsigned mid = (b0 + b1) >> 1;
            if (keys[mid] < khi) b0 = mid + 1u; else b1 = mid;
        }
    }
    a = a0;
    b = b0;
}

// A wave-uniform value computed on the VALU (there is no scalar float64 unit) sits in a VGPR per lane; read back through
// v_readfirstlane it lives in scalar registers, whose spills cost a VGPR LANE each instead of a scratch dword per lane.
__device__ __forceinline__ double uniform_f64(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float uniform_f32(float v) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }

// points of the fullest over-full leaf (a leaf of more than 64 points is ONE fine cell): the host entry of adaptive mode reads it
__global__ __launch_bounds__(256) void tree_leaf_max_kernel(TreeParams *__restrict__ tp, const unsigned *__restrict__ leafstart, int cap)
{
    const int nl = (int)tp->nleaves;
    unsigned mx = 0;
    for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < nl; l += gridDim.x * blockDim.x) mx = max(mx, leafstart[l + 1] - leafstart[l]);
    if (mx > (unsigned)cap) atomicMax(&tp->max_leaf, mx);
}

// ---------------------------------------------------------------- knn_leaf
#ifndef GSX_LEAF_WAVES33
#define GSX_LEAF_WAVES33 4   // (round 4: -5 % at k = 25 against 3)
#endif
#ifndef GSX_LEAF_WAVES65
#define GSX_LEAF_WAVES65 2
#endif
#ifndef GSX_LEAF_WAVES25
#define GSX_LEAF_WAVES25 4
#endif
#ifndef GSX_LEAF_WAVES49
#define GSX_LEAF_WAVES49 3
#endif
constexpr int leaf_min_waves(int kcap)
{
    return kcap <= 17 ? GSX_LEAF_WAVES17 : (kcap <= 25 ? GSX_LEAF_WAVES25 : (kcap <= 33 ? GSX_LEAF_WAVES33 : (kcap <= 49 ? GSX_LEAF_WAVES49 : GSX_LEAF_WAVES65)));
}

template <int KCAP>
__global__ __launch_bounds__(TREE_THREADS, leaf_min_waves(KCAP)) void knn_leaf_kernel(
    TreeParams *__restrict__ tp, const unsigned long long *__restrict__ keys, const unsigned long long *__restrict__ samples,
    const float4 *__restrict__ refs, const unsigned *__restrict__ leafstart, const unsigned char *__restrict__ leafbl, int k, int cand_limit,
    int q_begin, int q_count, float rf_scale, float *__restrict__ mean_out, double *__restrict__ kth_out, unsigned *__restrict__ faillist,
    double *__restrict__ failbound, int share, int nshares)
{
    constexpr int L = KCAP - 1;
    using Net = TopNet<L>;
    constexpr int BS = Net::BS, HB = GSX_LEAF_HB < BS ? GSX_LEAF_HB : BS;
    constexpr int TWCAP = KCAP > 33 ? GSX_TWCAP_BIG : gsx::TWCAP;   // (the kernels of the longer lists run two or three waves per SIMD)
    __shared__ unsigned s_mask[TREE_THREADS / 64][TWCAP][64];
    // per mask word: the (pre-adjusted) first index of up to four key ranges and where in the word each one ends
    __shared__ unsigned s_wb[TREE_THREADS / 64][4][TWCAP];
    __shared__ unsigned s_wcut[TREE_THREADS / 64][TWCAP];
    // Per-leaf geometry, wave-uniform, PARKED IN LDS (round 5).  These values are computed on the VALU (there is no scalar
    // float64 unit), so each of them is a VGPR -- and at 96 VGPRs the register allocator kept them alive across the batch
    // loop by spilling them per LANE: 26 dwords x 64 lanes of scratch written per leaf, 2.2 GB per 10M-splat launch
    // (profiles/r04_tree_pmc_floaters_10m.txt).  One lane writes the record once per leaf; the uses read it back through a
    // volatile pointer (a broadcast ds_read at the point of use, nothing for the allocator to keep).
    struct LeafRec {
        double plane_lo[3], plane_hi[3];   // faces of the searched box with space behind them (-inf / +inf: none)
        double r_f, cell;
        float ccx, ccy, ccz, inv_h;
    };
    __shared__ LeafRec s_rec[TREE_THREADS / 64];

    if (tp->bad_input) return;
    const int lane = lane_id();
    const int wv = uniform((int)(threadIdx.x >> 6));
    unsigned(*mask)[64] = s_mask[wv];
    unsigned *wb0 = s_wb[wv][0], *wb1 = s_wb[wv][1], *wb2 = s_wb[wv][2], *wb3 = s_wb[wv][3];
    unsigned *wcut = s_wcut[wv];
    volatile LeafRec *rec = &s_rec[wv];

    const int n = tp->n;
    const int nblk = (n + KEY_BLOCK - 1) / KEY_BLOCK;
    const double ox = tp->ox, oy = tp->oy, oz = tp->oz, s = tp->s, slack = tp->slack;

    // multi-GPU (replicated exchange): the leaves [part_lo, part_hi) are this rank's share -- a slab of the Morton order
    const int part_lo = (int)(((long long)tp->nleaves * share) / nshares), part_hi = (int)(((long long)tp->nleaves * (share + 1)) / nshares);
    WorkQueue wq;
    wq_init(wq, tp->leaf_ctr, part_hi - part_lo, TREE_THREADS / 64);
    for (;;) {
        const int witem = uniform(wq_next(wq));
        if (witem < 0) break;
        const int item = witem + part_lo;
        const int ls = uniform((int)leafstart[item]), le = uniform((int)leafstart[item + 1]);
        const int bl = uniform((int)leafbl[item]);
        const int nq = le - ls;
        const unsigned long long key0 = keys[ls];
        // the leaf's box in fine cells: bits below bl are free -- x gets ceil(bl/3) of them, y ceil((bl-1)/3), z floor(bl/3)
        const int q3 = bl / 3, r3 = bl - 3 * q3;
        const int bxb = q3 + (r3 >= 1), byb = q3 + (r3 >= 2), bzb = q3;
        const unsigned long long nodekey = (key0 >> bl) << bl;
        const int fx = (int)compact21(nodekey), fy = (int)compact21(nodekey >> 1), fz = (int)compact21(nodekey >> 2);
        const int Lc = max(bxb - 1, 0);   // a CELL is the cube of 2^Lc fine cells: half the leaf's longest side
        const int ncx = 1 << (bxb - Lc), ncy = 1 << (byb - Lc), ncz = 1 << (bzb - Lc);   // cells of the leaf: 1 or 2 per axis
        const int CX0 = (fx >> Lc) - 1, CY0 = (fy >> Lc) - 1, CZ0 = (fz >> Lc) - 1;   // first cell of the searched box
        const int rx = ncx + 2, ry = ncy + 2, rz = ncz + 2;
        const int cmax = 1 << (TB - Lc);

        // ---- lane (ci, cj, cl) looks up the key range of cell (CX0 + ci, CY0 + cj, CZ0 + cl)
        int r_lo = 0, r_len = 0;
        {
            const int ci = lane & 3, cj = (lane >> 2) & 3, cl = lane >> 4;
            const int X = CX0 + ci, Y = CY0 + cj, Z = CZ0 + cl;
            const bool inleaf = ci >= 1 && ci <= ncx && cj >= 1 && cj <= ncy && cl >= 1 && cl <= ncz;
            const bool valid = ci < rx && cj < ry && cl < rz && X >= 0 && Y >= 0 && Z >= 0 && X < cmax && Y < cmax && Z < cmax && !inleaf;
            const unsigned long long code = valid ? morton63((unsigned)X, (unsigned)Y, (unsigned)Z) : 0ULL;
            unsigned a0 = 0, b0 = 0;
            if (GSX_TREE_ABL & 4) {
                a0 = (unsigned)ls;
                b0 = (unsigned)ls + ((lane & 7) == 0 ? 40u : 0u);   // 8 fake ranges of 40 candidates
                if (b0 > (unsigned)n) b0 = (unsigned)ls;
            } else {
                cell_range(keys, samples, n, nblk, valid, code, Lc, a0, b0);
            }
            r_lo = (int)a0;
            r_len = (int)(b0 - a0);
            if (lane == 21) {   // cell (1,1,1): stands for the leaf itself, whose range is known
                r_lo = ls;
                r_len = nq;
            }
            // the two cells of an x-pair are siblings (the leaf's first cell has an even x): one range
            if (ncx == 2) {
                const int nl = __shfl_down(r_len, 1);
                if (!inleaf && ci == 1) r_len += nl;
                if (!inleaf && ci == 2) r_len = 0;
            }
        }
        // non-empty ranges to the low lanes (lane order kept)
        const unsigned long long ne = __ballot(r_len > 0);
        const int nr = (int)__popcll(ne);
        int rs_start, rs_len;
        {
            const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ne >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ne, 0u));
            const int dst = r_len > 0 ? below : nr + (lane - below);
            rs_start = __builtin_amdgcn_ds_permute(dst << 2, r_lo);
            rs_len = __builtin_amdgcn_ds_permute(dst << 2, r_len);
        }
        const int ncand = uniform(wave_sum_i32(rs_len));

        // geometry shared by the batches of this leaf -> the wave's LDS record
        {
            const double cell = s * (double)(1 << Lc);   // cell edge (real units)
            const double vol = s * s * s * ldexp(1.0, bl);
            // radius inside which ~2 (k+1) points are expected at the leaf's own density (the uniform grid's cell edge), x rf_scale
            const double r_f = (GSX_TREE_ABL & 32) ? 1e30 : (double)rf_scale * cbrt(0.397 * (double)(k + 1) * vol / (double)max(nq, 1));
            wave_sync();   // the previous leaf's last reads of the record are done
            if (lane == 0) {
                const double od[3] = {ox, oy, oz};
                const int c0[3] = {CX0, CY0, CZ0};
                const int rr[3] = {rx, ry, rz};
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    rec->plane_lo[a] = c0[a] > 0 ? od[a] + (double)c0[a] * cell : -__builtin_inf();
                    rec->plane_hi[a] = c0[a] + rr[a] < cmax ? od[a] + (double)(c0[a] + rr[a]) * cell : __builtin_inf();
                }
                rec->r_f = r_f;
                rec->cell = cell;
                rec->ccx = (float)(ox + ((double)fx + 0.5 * (double)(1 << bxb)) * s);
                rec->ccy = (float)(oy + ((double)fy + 0.5 * (double)(1 << byb)) * s);
                rec->ccz = (float)(oz + ((double)fz + 0.5 * (double)(1 << bzb)) * s);
                rec->inv_h = (float)(1.0 / cell);
            }
            wave_sync();
        }
        const bool irregular = ncand > cand_limit;

        for (int qb = 0; qb < nq; qb += 64) {
            const int f = qb + lane;
            const bool live = f < nq;
            const int qidx = ls + (live ? f : 0);
            const float4 qp = refs[qidx];
            const float qx = qp.x, qy = qp.y, qz = qp.z;
            const unsigned self_w = __float_as_uint(qp.w);
            const int qorig = (int)(self_w & 0x7fffffffu) - q_begin;
            const bool is_query = live && !(self_w >> 31) && qorig >= 0 && qorig < q_count;
            if (!__any(is_query)) continue;
            if (irregular) {   // wave-uniform: too many candidates for one wave's lock-step scan
                // Typically a SLIVER: a node whose box is large because most of it lies outside the object whose rim it cuts
                // (<= 64 points in a thin slab along a face or an edge), while its one-cell margin reaches deep into the dense
                // inside.  The radius to try first comes from the density inside the tight box of the leaf's own points.
                float lo3[3] = {live ? qx : 3.0e38f, live ? qy : 3.0e38f, live ? qz : 3.0e38f};
                float hi3[3] = {live ? qx : -3.0e38f, live ? qy : -3.0e38f, live ? qz : -3.0e38f};
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    lo3[a] = wave_min_f32(lo3[a]);
                    hi3[a] = wave_max_f32(hi3[a]);
                }
                const double e3[3] = {(double)hi3[0] - (double)lo3[0], (double)hi3[1] - (double)lo3[1], (double)hi3[2] - (double)lo3[2]};
                const double emx = fmax(fmax(e3[0], e3[1]), e3[2]);
                const double vt = fmax(e3[0], 1e-3 * emx) * fmax(e3[1], 1e-3 * emx) * fmax(e3[2], 1e-3 * emx);
                double rt = 1.3 * cbrt(0.397 * (double)(k + 1) * vt / (double)min(nq - qb, 64));
                // ... which says nothing when the leaf holds a handful of points (a flyer alone in a node the size of the space
                // between it and the scene: tight box 0, radius to try half a fine cell, and the descent widened it pass by pass,
                // 2^-10 -> 2^6): the node was split down to THIS size because its parent held more than a leaf's worth of points,
                // so the neighbours are about a cell away
                if (min(nq - qb, 64) <= 4) rt = fmax(rt, 0.5 * rec->cell);
                {   // one run of list entries per leaf (knn_tree_near takes neighbouring entries together)
                    const unsigned long long fb = __ballot(is_query);
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(&tp->fail_count, (unsigned)__popcll(fb));
                    base = (unsigned)uniform((int)base);
                    if (is_query) {
                        const unsigned slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(fb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fb, 0u));
                        faillist[slot] = (unsigned)qidx;
                        failbound[slot] = -fmax(rt * rt, 0.25 * s * s);
                    }
                }
                continue;
            }
            // ---- acceptance radius: distance to the nearest face of the searched box with space behind it, capped at r_f.
            // Evaluated TWICE from the wave's LDS record (here for the filter bound, after phase 2 for the acceptance test)
            // instead of carried through both phases in two registers per lane, and from coordinates widened on the spot
            // (pinned: the float64 copies of qx, qy, qz would otherwise be shared with phase 2's and live -- i.e. spilled --
            // across the matrix-core loop).  Same inputs, same instructions: the same bits both times.
            auto accept_radius_sq = [&]() __attribute__((always_inline)) {
                double rs = rec->r_f;
                const double qd[3] = {(double)pinned_here(qx), (double)pinned_here(qy), (double)pinned_here(qz)};
#pragma unroll
                for (int a = 0; a < 3; ++a) {   // (no face: -inf / +inf, the distance is +inf and fmin keeps rs)
                    rs = fmin(rs, (qd[a] - rec->plane_lo[a]) - slack);
                    rs = fmin(rs, (rec->plane_hi[a] - qd[a]) - slack);
                }
                rs = fmax(rs, 0.0);
                return rs * rs;
            };
            float tau;
            {
                const double racc_sq = accept_radius_sq();
                tau = is_query ? bound_from(racc_sq) : -1.0f;
            }

            Net lst;
            bool lst_empty = true;
            int widx = 0;
            unsigned nzw = 0;

            // ---- phase 2 (as knn_brick's): walk this lane's set bits, exact float64 distances, network selection
            auto drain = [&]() __attribute__((always_inline)) {
                const double qxd = (double)pinned_here(qx), qyd = (double)pinned_here(qy), qzd = (double)pinned_here(qz);
                wave_sync();
                if (GSX_TREE_ABL & 1) nzw = 0;
                unsigned m = 0, cut = 0x202020u;
                int b0 = ls, b1 = ls, b2 = ls, b3 = ls;
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
                                b0 = (int)wb0[w];
                                b1 = (int)wb1[w];
                                b2 = (int)wb2[w];
                                b3 = (int)wb3[w];
                                cut = wcut[w];
                            }
                            ok[j] = m != 0;
                            const int i = ok[j] ? __builtin_clz(m) : 0;
                            m &= ~(0x80000000u >> i);
                            const int base = i < (int)(cut & 255u) ? b0 : (i < (int)((cut >> 8) & 255u) ? b1 : (i < (int)((cut >> 16) & 255u) ? b2 : b3));
                            pt[j] = refs[base + i];
                        }
#pragma unroll
                        for (int j = 0; j < HB; ++j) {
                            const double d = dist2_f64(qxd, qyd, qzd, pt[j].x, pt[j].y, pt[j].z);
                            blk[h0 + j] = (ok[j] && __float_as_uint(pt[j].w) != self_w) ? d : __builtin_inf();
                        }
                    }
                    if (uniform((int)lst_empty)) lst.assign_block(blk); else lst.merge_block(blk);
                    lst_empty = false;
                    if (!__any(m != 0 || nzw != 0)) break;
                }
                tau = fminf(tau, bound_from(lst.kth(k)));
                widx = 0;
                nzw = 0;
                wave_sync();
            };

            // ---- the candidate ranges are walked as ONE flat sequence cut into 32-candidate words; a word takes up to four
            // range pieces (a cell holds a handful of points: two pieces per word would leave the MFMA tiles half empty)
            struct Word { int b[4]; int e[4]; int r, off; };   // b[j] + t = index of candidate t for t in [e[j-1], e[j]); wave-uniform
            auto next_word = [&](int r, int off) __attribute__((always_inline)) {
                Word o;
                int fill = 0;
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {
                    int len = r < nr ? __builtin_amdgcn_readlane(rs_len, r & 63) : 0;
                    while (r < nr && off >= len) {
                        ++r;
                        off = 0;
                        len = r < nr ? __builtin_amdgcn_readlane(rs_len, r & 63) : 0;
                    }
                    int take = 0, st = 0;
                    if (r < nr && fill < 32) {
                        take = min(len - off, 32 - fill);
                        st = __builtin_amdgcn_readlane(rs_start, r & 63) + off;
                    }
                    o.b[sg] = st - fill;
                    fill += take;
                    off += take;
                    o.e[sg] = fill;
                }
                o.r = r;
                o.off = off;
                return o;
            };
            // ---- phase 1, matrix cores: cell-unit coordinates relative to the leaf centre (|u| <= 2); see knn_mfma.h.  One pass
            // fills the park (28 words = 896 candidates) from position (cr, coff) of the flat sequence on and says whether words
            // are left.  Nearly every 64-point leaf is ONE pass, and that pass runs with the list not yet alive (its registers
            // are the filter's); the larger leaves of k > 16 (tree_leaf_cap_for) and the boxes of odd shapes take further passes
            // -- drain, tighten the bound by the list's k-th distance, filter the next 896 candidates with the list alive.
            // (Round 4 filtered such boxes from scratch on the float32 VALU, scalar loads: 3x the time per candidate.)
            int cr = 0, coff = 0;
            auto mfma_fill = [&]() __attribute__((always_inline)) {
                const float ccx = uniform_f32(rec->ccx), ccy = uniform_f32(rec->ccy), ccz = uniform_f32(rec->ccz);
                const float g_inv_h = uniform_f32(rec->inv_h);
                const float uqx = (qx - ccx) * g_inv_h, uqy = (qy - ccy) * g_inv_h, uqz = (qz - ccz) * g_inv_h;
                const float nq2 = __builtin_fmaf(uqz, uqz, __builtin_fmaf(uqy, uqy, uqx * uqx));
                const bool upper = lane >= 32;
                const float s_q = tau >= 0.0f ? nq2 - ((tau * g_inv_h) * g_inv_h * (1.0f + 1e-6f) + MF_SLACK) : 1.0e30f;
                bf16x8 opa, opb;
                mf_query_operands(uqx, uqy, uqz, s_q, opa, opb);
                const int my_cand = mf_cand_of_row(lane & 31);
                auto fetch = [&](const Word &w) __attribute__((always_inline)) {
                    const int c = w.e[3];
                    if (c == 0) return make_float4(0.f, 0.f, 0.f, 0.f);
                    const int t = min(my_cand, c - 1);   // slots past the end repeat the last candidate (masked out below)
                    const int base = t < w.e[0] ? w.b[0] : (t < w.e[1] ? w.b[1] : (t < w.e[2] ? w.b[2] : w.b[3]));
                    return refs[base + t];
                };
                Word w_cur = next_word(cr, coff);
                Word w_n1 = next_word(w_cur.r, w_cur.off);
                float4 p_cur = fetch(w_cur), p_n1 = fetch(w_n1);
                while (w_cur.e[3] > 0 && widx < TWCAP && !(GSX_TREE_ABL & 2)) {
                    const Word w_n2 = next_word(w_n1.r, w_n1.off);
                    const float4 p_n2 = fetch(w_n2);
                    const bf16x8 cand = mf_candidate_operand((p_cur.x - ccx) * g_inv_h, (p_cur.y - ccy) * g_inv_h,
                                                             (p_cur.z - ccz) * g_inv_h, upper);
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    const unsigned ma = mf_sign_bits(__builtin_amdgcn_mfma_f32_32x32x16_bf16(cand, opa, zero, 0, 0, 0));
                    const unsigned mb = mf_sign_bits(__builtin_amdgcn_mfma_f32_32x32x16_bf16(cand, opb, zero, 0, 0, 0));
                    auto sw = __builtin_amdgcn_permlane32_swap(ma, mb, false, false);
                    const unsigned m = ((sw[0] << 16) | (sw[1] & 0xffffu)) & (0xffffffffu << (32 - w_cur.e[3]));
                    mask[widx][lane] = m;
                    if (lane == 0) {
                        wb0[widx] = (unsigned)w_cur.b[0];
                        wb1[widx] = (unsigned)w_cur.b[1];
                        wb2[widx] = (unsigned)w_cur.b[2];
                        wb3[widx] = (unsigned)w_cur.b[3];
                        wcut[widx] = (unsigned)w_cur.e[0] | ((unsigned)w_cur.e[1] << 8) | ((unsigned)w_cur.e[2] << 16);
                    }
                    nzw |= (m != 0 ? 1u : 0u) << widx;
                    ++widx;
                    cr = w_cur.r;      // the sequence continues behind the word just parked
                    coff = w_cur.off;
                    w_cur = w_n1;
                    p_cur = p_n1;
                    w_n1 = w_n2;
                    p_n1 = p_n2;
                }
                wave_sync();
                return w_cur.e[3] > 0 && !(GSX_TREE_ABL & 2);   // wave-uniform: words left over
            };
            bool more = mfma_fill();
            lst.init();
            for (;;) {
                drain();
                if (!uniform((int)more)) break;
                more = mfma_fill();
            }

            {
                const double kth_d2 = lst.kth(k);
                const double racc_sq = accept_radius_sq();
                const bool failed = is_query && !GSX_TREE_ABL && !(kth_d2 <= racc_sq);
                // the queries this leaf could not certify: ONE run of list entries (one atomic per leaf; knn_tree_near takes
                // neighbouring entries together)
                const unsigned long long fb = __ballot(failed);
                unsigned fbase = 0;
                if (fb != 0ull && lane == 0) fbase = atomicAdd(&tp->fail_count, (unsigned)__popcll(fb));
                fbase = (unsigned)uniform((int)fbase);
                if (!is_query) {
                } else if (GSX_TREE_ABL) {
                    mean_out[qorig] = (float)kth_d2;   // keeps the list live, never fails
                } else if (kth_d2 <= racc_sq) {
                    if (kth_out) kth_out[qorig] = kth_d2;
                    mean_out[qorig] = mean_from_net(lst, k);
                } else {
                    const unsigned slot = fbase + __builtin_amdgcn_mbcnt_hi((unsigned)(fb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fb, 0u));
                    faillist[slot] = (unsigned)qidx;
                    // the filter only let candidates inside the acceptance radius through, of which there were fewer than k (a full
                    // list would bound the k-th distance): the next ball to try is 1.3x wider -- 2.2x the volume
                    { const double cell = rec->cell; failbound[slot] = kth_d2 < __builtin_inf() ? kth_d2 : -fmax(1.69 * racc_sq, 0.25 * cell * cell); }
                }
            }
        }
    }
}

// ---------------------------------------------------------------- knn_tree_query (fallback)
struct TNode {
    unsigned lo, hi;            // range of the sorted array
    unsigned long long code;    // key >> 3*level of its points
    double mind2;               // lower bound of the squared distance from the query to any of its points
    int level;                  // octree level: a cube of 2^level fine cells
    int pad;
};

__device__ __forceinline__ double wave_min_f64_(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ double bcast_f64(double v, int src)   // wave-uniform src
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// first index in [lo, hi) whose key is >= K (hi if none): 64 probes per round, all lanes take part; wave-uniform arguments
__device__ __forceinline__ unsigned wave_lower_bound(const unsigned long long *__restrict__ keys, unsigned lo, unsigned hi,
                                                     unsigned long long K, int lane)
{
    while (hi - lo > 64u) {
        const unsigned size = hi - lo, stride = (size + 63u) >> 6;
        auto P = [&](unsigned t) { const unsigned p = lo + (t + 1u) * stride - 1u; return p < hi - 1u ? p : hi - 1u; };
        const bool pred = keys[P((unsigned)lane)] < K;
        const int cnt = (int)__popcll(__ballot(pred));
        const unsigned nlo = cnt > 0 ? P((unsigned)cnt - 1u) + 1u : lo;
        const unsigned nhi = cnt < 64 ? P((unsigned)cnt) : hi;
        lo = (unsigned)uniform((int)nlo);
        hi = (unsigned)uniform((int)nhi);
    }
    const unsigned p = lo + (unsigned)lane;
    const bool pred = p < hi && keys[p < hi ? p : 0u] < K;
    return lo + (unsigned)__popcll(__ballot(pred && hi > lo));
}

// ---------------------------------------------------------------- knn_tree_near (near misses)
// Nearly every query knn_leaf cannot certify has a FULL list: k candidates were found, the k-th just lies beyond the nearest
// face of the searched box.  That k-th distance bounds the true one from above, so the answer is inside a known ball -- no
// descent needed: the ball is covered by at most 6^3 cells whose edge is between a quarter and half of its diameter, each a
// contiguous key range (one lane, one search); the few dozen points inside the ball are collected in LDS and the k nearest
// taken by rank.  ~20 dependent memory round trips per query instead of knn_tree_query's 40-100.  Queries without a bound, and
// balls that reach into a much denser region, are handed on to knn_tree_query.
__global__ __launch_bounds__(TREE_THREADS, 6) void knn_tree_near_kernel(
    TreeParams *__restrict__ tp, const unsigned long long *__restrict__ keys, const unsigned long long *__restrict__ samples,
    const float4 *__restrict__ refs, const float4 *__restrict__ boxes, const unsigned *__restrict__ faillist,
    const double *__restrict__ failbound, int k, int q_begin, float *__restrict__ mean_out, double *__restrict__ kth_out,
    unsigned *__restrict__ faillist2, double *__restrict__ failbound2, double cell_frac)
{
    __shared__ double s_cand[TREE_THREADS / 64][TQ_CAND];
    __shared__ double s_out[TREE_THREADS / 64][64];
    __shared__ unsigned s_cnt[TREE_THREADS / 64];
    if (tp->bad_input) return;
    const int nfail = (int)tp->fail_count;
    if (nfail == 0) return;
    const int lane = lane_id();
    const int wv = uniform((int)(threadIdx.x >> 6));
    double *cand = s_cand[wv], *out = s_out[wv];
    unsigned *cnt = &s_cnt[wv];
    const int n = tp->n;
    const int nblk = (n + KEY_BLOCK - 1) / KEY_BLOCK;
    const double od[3] = {tp->ox, tp->oy, tp->oz};
    const double s = tp->s, inv_s = tp->inv_s, slack = tp->slack;

    WorkQueue wq;
    wq_init(wq, tp->fail_ctr, nfail, TREE_THREADS / 64);
    for (;;) {
        const int item = uniform(wq_next(wq));
        if (item < 0) break;
        const int qidx = uniform((int)faillist[item]);
        const double bound = failbound[item];
        bool defer = false;
        const float4 qp = refs[qidx];
        const unsigned self_w = __float_as_uint(qp.w);
        const double qd[3] = {(double)qp.x, (double)qp.y, (double)qp.z};
        // a known bound on the k-th distance, or (negative) the square of a radius to try: every point inside the ball is looked
        // at, so finding k of them within r - 2 slack certifies the answer.  A ball that holds fewer (the query sits at the rim
        // of an object: half of it is empty) is widened twice, 1.3x each, before the query is handed on.
        double r = bound >= 0.0 ? __dsqrt_rn(bound) * (1.0 + 1e-9) + 4.0 * slack : __dsqrt_rn(-bound);
        int M = 0;
        for (int attempt = 0;; ++attempt) {
            const double rin = fmax(r - 2.0 * slack, 0.0);
            const double T = bound >= 0.0 ? bound : rin * rin;
            int Lg = 0;   // cells of 2^Lg fine cells: the smallest with an edge of at least cell_frac x r ("tree_near_cell", 0.5)
            while (Lg < TB && s * (double)(1u << Lg) < cell_frac * r) ++Lg;
            const double g = s * (double)(1u << Lg);
            int c0[3], nc[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double tl = ((qd[a] - r) - od[a]) * inv_s - 1.0, th = ((qd[a] + r) - od[a]) * inv_s + 1.0;
                const unsigned l = (unsigned)fmin(fmax(tl, 0.0), (double)((1 << TB) - 1)) >> Lg;
                const unsigned h = (unsigned)fmin(fmax(th, 0.0), (double)((1 << TB) - 1)) >> Lg;
                c0[a] = uniform((int)l);
                nc[a] = uniform((int)(h - l) + 1);
            }
            const int total = nc[0] * nc[1] * nc[2];
            defer = total > 512;
            if (defer && lane == 0) atomicAdd(&tp->defer_why[0], 1u);
            if (lane == 0) *cnt = 0u;
            wave_sync();
            for (int cb = 0; cb < total && !defer; cb += 64) {
                const int c = cb + lane;
                const bool valid = c < total;
                const int iz = c / (nc[0] * nc[1]), rem = c - iz * nc[0] * nc[1], iy = rem / nc[0], ix = rem - iy * nc[0];
                const unsigned X[3] = {(unsigned)(c0[0] + ix), (unsigned)(c0[1] + iy), (unsigned)(c0[2] + iz)};
                double m2 = 0.0;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const double plo = od[a] + (double)X[a] * g, phi = od[a] + (double)(X[a] + 1u) * g;
                    const double dd = fmax(fmax((plo - qd[a]) - slack, (qd[a] - phi) - slack), 0.0);
                    m2 += dd * dd;
                }
                const bool keep = valid && m2 * (1.0 - 1e-14) <= T;
                unsigned a0, b0;
                cell_range(keys, samples, n, nblk, keep, keep ? morton63(X[0], X[1], X[2]) : 0ULL, Lg, a0, b0);
                const unsigned len = b0 - a0;
                if (__any(len > (unsigned)TQ_DENSE)) {   // a cell far denser than the ball's own neighbourhood: the pruning descent's job
                    defer = true;
                    if (lane == 0) atomicAdd(&tp->defer_why[1], 1u);
                    if (lane == 0) atomicAdd(&tp->defer_why[5 + min(attempt, 2)], 1u);
                    break;
                }
                if (len <= 64u)
                    for (unsigned j = 0; j < len; j += 4u) {   // a handful of points per cell: the lane that found it scans it,
                        float4 p[4];                           // four loads in flight (round 5: one at a time was a chain of
#pragma unroll                                                  // ~8 memory round trips per query)
                        for (unsigned u = 0; u < 4u; ++u) p[u] = refs[a0 + (j + u < len ? j + u : j)];
#pragma unroll
                        for (unsigned u = 0; u < 4u; ++u) {
                            const double d = dist2_f64(qd[0], qd[1], qd[2], p[u].x, p[u].y, p[u].z);
                            if (j + u < len && __float_as_uint(p[u].w) != self_w && d <= T) {
                                const unsigned at = atomicAdd(cnt, 1u);
                                if (at < (unsigned)TQ_CAND) cand[at] = d;
                            }
                        }
                    }
                // a cell of up to TQ_DENSE points (the query sits at the edge of something denser): the whole wave takes it, block
                // of 64 points by block, skipping the blocks whose tight box lies outside the ball
                for (unsigned long long dense = __ballot(len > 64u); dense; dense &= dense - 1) {
                    const int src = (int)__builtin_ctzll(dense);
                    const unsigned da = (unsigned)__shfl((int)a0, src), dl = (unsigned)__shfl((int)len, src);
                    const unsigned blk0 = da >> 6, blk1 = (da + dl - 1u) >> 6;
                    for (unsigned bb = blk0; bb <= blk1; bb += 64u) {
                        const unsigned myb = bb + (unsigned)lane;
                        bool need = myb <= blk1;
                        if (need) {
                            const float4 blo = boxes[2 * myb], bhi = boxes[2 * myb + 1];
                            const double dx = fmax(fmax((double)blo.x - qd[0], qd[0] - (double)bhi.x), 0.0);
                            const double dy = fmax(fmax((double)blo.y - qd[1], qd[1] - (double)bhi.y), 0.0);
                            const double dz = fmax(fmax((double)blo.z - qd[2], qd[2] - (double)bhi.z), 0.0);
                            need = (dx * dx + dy * dy + dz * dz) * (1.0 - 1e-14) <= T;
                        }
                        for (unsigned long long todo = __ballot(need); todo; todo &= todo - 1) {
                            const unsigned j = ((bb + (unsigned)__builtin_ctzll(todo)) << 6) + (unsigned)lane;
                            const bool in = j >= da && j < da + dl;
                            const float4 p = refs[in ? j : da];
                            const double d = dist2_f64(qd[0], qd[1], qd[2], p.x, p.y, p.z);
                            if (in && __float_as_uint(p.w) != self_w && d <= T) {
                                const unsigned at = atomicAdd(cnt, 1u);
                                if (at < (unsigned)TQ_CAND) cand[at] = d;
                            }
                        }
                    }
                }
            }
            wave_sync();
            M = uniform((int)*cnt);
            if (defer || M > TQ_CAND) {
                if (!defer && lane == 0) atomicAdd(&tp->defer_why[2], 1u);
                defer = true;
                break;
            }
            if (M >= k) break;
            if (bound >= 0.0 || attempt == 2) {
                if (lane == 0) atomicAdd(&tp->defer_why[bound >= 0.0 ? 4 : 3], 1u);
                defer = true;
                break;
            }
            r *= 1.3;
        }
        if (defer) {
            if (lane == 0) {
                const unsigned slot = atomicAdd(&tp->fail2_count, 1u);
                faillist2[slot] = (unsigned)qidx;
                failbound2[slot] = bound >= 0.0 ? bound : -4.0 * r * r;   // a radius that was tried: the descent starts at twice that
            }
            continue;
        }
        // the k smallest of the collected values, by rank
        double kth2 = 0.0;
        for (int i0 = 0; i0 < M; i0 += 64) {
            const int i = i0 + lane;
            const double ci = i < M ? cand[i] : __builtin_inf();
            int rank = 0;
            for (int j = 0; j < M; ++j) {
                const double cj = cand[j];
                rank += (cj < ci || (cj == ci && j < i)) ? 1 : 0;
            }
            if (i < M && rank < k) out[rank] = __dsqrt_rn(ci);
            const unsigned long long last = __ballot(i < M && rank == k - 1);
            if (last) kth2 = bcast_f64(ci, (int)__builtin_ctzll(last));
        }
        wave_sync();
        if (lane == 0) {
            const int qorig = (int)(self_w & 0x7fffffffu) - q_begin;
            const double sum = pairwise_sum_le128([&](int i) { return out[i]; }, k);
            mean_out[qorig] = __double2float_rn(__ddiv_rn(sum, (double)k));
            if (kth_out) kth_out[qorig] = kth2;
        }
        wave_sync();
    }
}

__global__ __launch_bounds__(TREE_THREADS, 3) void knn_tree_query_kernel(
    TreeParams *__restrict__ tp, const unsigned long long *__restrict__ keys, const unsigned long long *__restrict__ samples,
    const float4 *__restrict__ refs, const float4 *__restrict__ boxes,
    const unsigned *__restrict__ faillist, const double *__restrict__ failbound, int k, int q_begin, int out_count,
    float *__restrict__ mean_out, double *__restrict__ kth_out)
{
    __shared__ TNode s_stack[TREE_THREADS / 64][TQ_STACK];
    __shared__ double s_out[TREE_THREADS / 64][64];
    __shared__ double s_cand[TREE_THREADS / 64][TQ_CAND];
    if (tp->bad_input) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < out_count; i += gridDim.x * blockDim.x) mean_out[i] = __builtin_nanf("");
        return;
    }
    const int nfail = (int)tp->fail2_count;
    if (nfail == 0) return;
    const int lane = lane_id();
    const int wv = uniform((int)(threadIdx.x >> 6));
    TNode *stack = s_stack[wv];
    double *out = s_out[wv];
    double *cand = s_cand[wv];
    const int n = tp->n;
    const double od[3] = {tp->ox, tp->oy, tp->oz};
    const double s = tp->s, inv_s = tp->inv_s, slack = tp->slack;
    const int g = lane >> 3, u = lane & 7;

    WorkQueue wq;
    wq_init(wq, tp->fail2_ctr, nfail, TREE_THREADS / 64);
    for (;;) {
        const int item = uniform(wq_next(wq));
        if (item < 0) break;
        const int qidx = uniform((int)faillist[item]);
        const float4 qp = refs[qidx];
        const unsigned self_w = __float_as_uint(qp.w);
        const double qd[3] = {(double)qp.x, (double)qp.y, (double)qp.z};
        const double bound = failbound[item];
        double R = bound >= 0.0 ? __dsqrt_rn(bound) * (1.0 + 1e-9) + 4.0 * slack : __dsqrt_rn(-bound);
        bool known = bound >= 0.0;   // the bound is used for the first pass only (it cannot fail; if it did, plain doubling takes over)

        // Two ways through a pass.  COLLECT (first): every point inside the certifiable ball goes to an LDS buffer, and the k
        // nearest are picked by rank afterwards -- a query that just missed in knn_leaf has a bound a few neighbours wide, so
        // the buffer holds ~k..2k values and nothing is kept sorted on the way.  INSERT (when the buffer overflows: a ball that
        // reaches into a much denser region): the sorted list, one entry per lane, whose k-th entry prunes the descent.
        bool insert_mode = !known;   // a radius to try comes from knn_tree_near, which has done the collecting already
#ifdef GSX_TREE_PROFILE
        const unsigned long long t_begin = wall_clock64();
        int p_pass = 0, p_pop = 0, p_scan = 0, p_split = 0, p_scan_full = 0, p_split_full = 0, p_maxsp = 0, p_bigsplit = 0;
        long long p_pts = 0;
#endif
        for (;;) {   // one pass per search radius (a known bound needs exactly one)
#ifdef GSX_TREE_PROFILE
            ++p_pass;
#endif
            // The ball's box (one fine cell of margin against the cell rounding) is covered by at most 3^3 cells whose edge is at
            // least R + s: these are the roots of the descent -- NOT their common ancestor, which is the whole cloud whenever the
            // box straddles a high-level boundary of the octree (measured: most descents started from nodes of 10^5..10^7 points).
            int Lg = 0;
            while (Lg < TB && s * (double)(1u << Lg) < R + s) ++Lg;   // box width 2 R + 2 s <= 2 cells: it meets at most 3 per axis
            const bool whole = Lg >= TB;   // one cell: the whole cloud
            const double gedge = s * (double)(1u << Lg);
            int c0[3], nc[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double tl = ((qd[a] - R) - od[a]) * inv_s - 1.0, th = ((qd[a] + R) - od[a]) * inv_s + 1.0;
                const unsigned l = (unsigned)fmin(fmax(tl, 0.0), (double)((1 << TB) - 1)) >> Lg;
                const unsigned h = (unsigned)fmin(fmax(th, 0.0), (double)((1 << TB) - 1)) >> Lg;
                c0[a] = uniform((int)l);
                nc[a] = uniform((int)(h - l) + 1);
            }
            const int total = nc[0] * nc[1] * nc[2];   // <= 27: one lane each
            const double rc = whole ? __builtin_inf() : R;
            // points beyond the certified radius are of no use to this pass; a known bound on the k-th distance is tighter still
            const double rcert = fmin(R - 2.0 * slack, rc);
            double T0 = whole ? __builtin_inf() : rcert * rcert;
            if (known) T0 = fmin(T0, bound);
            known = false;
            int M = 0;              // collected candidates (wave-uniform)
            bool overflow = false;

            double best = __builtin_inf();   // lane j: j-th smallest squared distance so far (the query itself excluded)
            asm volatile("" : "+v"(best));   // opaque: see kv
            // entry k-1: the running k-th distance.  Read back from the list, not assigned the constant: this compiler puts a
            // wave-uniform double constant into s_mov_b64 with a 64-bit literal, which the instruction does not have (the
            // register then holds 0: measured)
            double kv = bcast_f64(best, k - 1);
            int sp = 0;
            {
                const int c = lane;
                const bool valid = c < total;
                const int iz = c / (nc[0] * nc[1]), rem = c - iz * nc[0] * nc[1], iy = rem / nc[0], ix = rem - iy * nc[0];
                const unsigned X[3] = {(unsigned)(c0[0] + ix), (unsigned)(c0[1] + iy), (unsigned)(c0[2] + iz)};
                double m2 = 0.0;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const double plo = od[a] + (double)X[a] * gedge, phi = od[a] + (double)(X[a] + 1u) * gedge;
                    const double dd = fmax(fmax((plo - qd[a]) - slack, (qd[a] - phi) - slack), 0.0);
                    m2 += dd * dd;
                }
                m2 *= (1.0 - 1e-14);
                bool keep = valid && !(m2 > T0);
                const unsigned long long code = keep ? morton63(X[0], X[1], X[2]) : 0ULL;
                unsigned rlo = 0, rhi = 0;
                if (whole) {
                    rhi = keep ? (unsigned)n : 0u;
                } else {
                    cell_range(keys, samples, n, (n + KEY_BLOCK - 1) / KEY_BLOCK, keep, code, Lg, rlo, rhi);
                }
                keep = keep && rhi > rlo;
                // the pending nodes are an unordered set: the nearest is looked for at every step
                const unsigned long long kb = __ballot(keep);
                if (keep) {
                    TNode nd;
                    nd.lo = rlo;
                    nd.hi = rhi;
                    nd.code = code;
                    nd.mind2 = m2;
                    nd.level = Lg;
                    nd.pad = 0;
                    stack[(int)__builtin_amdgcn_mbcnt_hi((unsigned)(kb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)kb, 0u))] = nd;
                }
                sp = (int)__popcll(kb);
            }
            wave_sync();
            while (sp > 0) {
                // BEST-FIRST: the pending node nearest to the query (a last-in-first-out stack dives to the bottom of the first
                // dense subtree it meets: a floater 60 units from a scene scanned 330 nodes / 175 000 points of the scene's face
                // before it ever looked at the floaters 40 units away that are its neighbours -- 2 ms for one query)
                int mi = -1;
                double mv = __builtin_inf();
                for (int i = lane; i < sp; i += 64) {
                    const double v = stack[i].mind2;
                    if (v < mv) {
                        mv = v;
                        mi = i;
                    }
                }
                const double gmin = wave_min_f64_(mv);
                const unsigned long long who = __ballot(mi >= 0 && mv == gmin);
                if (!who) break;   // (cannot happen: sp > 0 and no NaN)
                const int pick = __shfl(mi, (int)__builtin_ctzll(who));
                const TNode nd = stack[pick];   // same address in every lane
                --sp;
                wave_sync();
                if (lane == 0 && pick != sp) stack[pick] = stack[sp];
                wave_sync();
                const unsigned nlo = (unsigned)uniform((int)nd.lo), nhi = (unsigned)uniform((int)nd.hi);
                const int level = uniform(nd.level);
                const double md = bcast_f64(nd.mind2, 0);
#ifdef GSX_TREE_PROFILE
                ++p_pop;
#endif
                if (md > T0 || (insert_mode && md >= kv)) break;   // every other pending node is at least as far (kv == 0, k exact
                                                                   // duplicates of the query: nothing can be nearer)
                const unsigned cnt = nhi - nlo;
#ifdef GSX_TREE_PROFILE
                if (cnt <= ((insert_mode && kv < 1e300) ? (unsigned)GSX_TQ_SCAN_FULL : (unsigned)TQ_SCAN) || level == 0) { ++p_scan; p_pts += cnt; if (kv < 1e300) ++p_scan_full; } else { ++p_split; if (kv < 1e300) ++p_split_full; if (cnt > 100000u) ++p_bigsplit; }
                p_maxsp = max(p_maxsp, sp);
#endif
                // Once the list is full a node's box often undercuts the k-th distance while none of its points does: the box is the
                // NODE's, and a node that straddles the face of a dense object reaches far beyond its points (a query 26 units
                // above such a face: ~100 nodes within 6 units had boxes nearer than its k-th neighbour, 4 of them points).  So a
                // node of up to 4096 points is not split any further: the tight boxes of its 64-point blocks are tested (one load
                // per lane) and only the blocks that can hold a candidate are scanned.
                const unsigned scan_cap = (insert_mode && kv < 1e300) ? (unsigned)GSX_TQ_SCAN_FULL : (unsigned)TQ_SCAN;
                if (cnt <= scan_cap || level == 0 || sp > TQ_STACK - 8) {   // (a full set: scanning is slow but exact)
                    const unsigned blk0 = nlo >> 6, blk1 = (nhi - 1u) >> 6;   // cnt > 0
                    for (unsigned bb = blk0; bb <= blk1 && !(!insert_mode && M > TQ_CAND); bb += 64u) {
                        const unsigned myb = bb + (unsigned)lane;
                        bool need = myb <= blk1;
                        if (need && cnt > 256u) {   // (small nodes: every block is needed more often than not)
                            const float4 blo = boxes[2 * myb], bhi = boxes[2 * myb + 1];
                            const double dx = fmax(fmax((double)blo.x - qd[0], qd[0] - (double)bhi.x), 0.0);
                            const double dy = fmax(fmax((double)blo.y - qd[1], qd[1] - (double)bhi.y), 0.0);
                            const double dz = fmax(fmax((double)blo.z - qd[2], qd[2] - (double)bhi.z), 0.0);
                            const double m2b = (dx * dx + dy * dy + dz * dz) * (1.0 - 1e-14);
                            need = !(m2b > T0) && (!insert_mode || m2b < kv);
                        }
                        unsigned long long todo = __ballot(need);
                        // ---- scan: up to TQ_FLIGHT blocks of 64 points in flight; the nearest candidate below the running k-th distance first
                        while (todo) {
                            float4 p4[TQ_FLIGHT];
                            unsigned base4[TQ_FLIGHT];
#pragma unroll
                            for (int v = 0; v < TQ_FLIGHT; ++v) {
                                base4[v] = 0xffffffffu;
                                if (todo) {
                                    base4[v] = (bb + (unsigned)__builtin_ctzll(todo)) << 6;
                                    todo &= todo - 1;
                                }
                                const unsigned j = base4[v] + (unsigned)lane;
                                p4[v] = refs[(base4[v] != 0xffffffffu && j >= nlo && j < nhi) ? j : nlo];
                            }
#pragma unroll
                            for (int v = 0; v < TQ_FLIGHT; ++v) {
                                if (base4[v] == 0xffffffffu) break;   // wave-uniform
                                const unsigned j = base4[v] + (unsigned)lane;
                                const float4 p = p4[v];
                                const double d = dist2_f64(qd[0], qd[1], qd[2], p.x, p.y, p.z);
                                bool have = j >= nlo && j < nhi && __float_as_uint(p.w) != self_w && d <= T0;
                                if (!insert_mode) {
                                    const unsigned long long hb = __ballot(have);
                                    const int at = M + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hb, 0u));
                                    if (have && at < TQ_CAND) cand[at] = d;
                                    M += (int)__popcll(hb);
                                } else {
                                    for (;;) {
                                        const bool cnd = have && d < kv;
                                        if (!__any(cnd)) break;
                                        const double dm = wave_min_f64_(cnd ? d : __builtin_inf());
                                        const unsigned long long pick = __ballot(cnd && d == dm);
                                        const int src = (int)__builtin_ctzll(pick);
                                        if (lane == src) have = false;
                                        const int pos = (int)__popcll(__ballot(best <= dm));
                                        const double up = __shfl_up(best, 1);
                                        best = lane < pos ? best : (lane == pos ? dm : up);
                                        kv = bcast_f64(best, k - 1);
                                    }
                                }
                            }
                            if (!insert_mode && M > TQ_CAND) break;   // wave-uniform: the buffer is full
                        }
                    }
                    if (M > TQ_CAND) {   // wave-uniform
                        overflow = true;
                        break;
                    }
                } else {
                    // ---- split: the 7 inner boundaries of the node's 8 children, one 8-lane group each (8 probes per round)
                    const unsigned long long ccode = (nd.code << 3) | (unsigned long long)g;
                    const unsigned long long K = ccode << (3 * (level - 1));
                    unsigned slo = nlo, shi = g == 0 ? nlo : nhi;   // group 0: its child starts at nlo
                    for (;;) {
                        const unsigned size = shi - slo;
                        const bool big = size > 8u;
                        if (!__any(big)) break;
                        const unsigned stride = (size + 7u) >> 3;
                        auto P = [&](unsigned t) { const unsigned p = slo + (t + 1u) * stride - 1u; return p < shi - 1u ? p : shi - 1u; };
                        const unsigned p = big ? P((unsigned)u) : nlo;
                        const bool pred = big && keys[p] < K;
                        const unsigned long long bal = __ballot(pred);
                        const int c8 = __popc((unsigned)(bal >> (8 * g)) & 0xffu);
                        if (big) {
                            const unsigned a = c8 > 0 ? P((unsigned)c8 - 1u) + 1u : slo;
                            const unsigned b = c8 < 8 ? P((unsigned)c8) : shi;
                            slo = a;
                            shi = b;
                        }
                    }
                    unsigned pos;
                    {
                        const unsigned p = slo + (unsigned)u;
                        const bool pred = p < shi && keys[p < shi ? p : nlo] < K;
                        const unsigned long long bal = __ballot(pred);
                        pos = slo + (unsigned)__popc((unsigned)(bal >> (8 * g)) & 0xffu);
                    }
                    const unsigned nxt = (unsigned)__shfl_down((int)pos, 8);   // start of the next child
                    const unsigned clo = pos, chi = g == 7 ? nhi : nxt;
                    // lower bound of the distance to the child's box
                    const unsigned cc[3] = {compact21(ccode), compact21(ccode >> 1), compact21(ccode >> 2)};
                    const double side = s * (double)(1u << (level - 1));
                    double m2 = 0.0;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const double plo = od[a] + (double)cc[a] * side, phi = od[a] + (double)(cc[a] + 1u) * side;
                        const double dd = fmax(fmax((plo - qd[a]) - slack, (qd[a] - phi) - slack), 0.0);
                        m2 += dd * dd;
                    }
                    m2 *= (1.0 - 1e-14);
                    const bool keep = chi > clo && !(m2 > T0) && (!insert_mode || m2 < kv);
                    const unsigned long long kb = __ballot(keep && u == 0);
                    if (keep && u == 0) {
                        TNode ch;
                        ch.lo = clo;
                        ch.hi = chi;
                        ch.code = ccode;
                        ch.mind2 = m2;
                        ch.level = level - 1;
                        ch.pad = 0;
                        stack[sp + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(kb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)kb, 0u))] = ch;
                    }
                    sp += (int)__popcll(kb);
                    wave_sync();
                }
            }
            if (!insert_mode) {
                if (overflow) {   // same radius again, with the pruning list
                    insert_mode = true;
                    continue;
                }
                if (M >= k) {
                    // every collected value is <= T0 <= rcert^2: the k smallest are the answer.  rank = values before mine
                    wave_sync();
                    double kth2 = 0.0;
                    for (int i0 = 0; i0 < M; i0 += 64) {
                        const int i = i0 + lane;
                        const double ci = i < M ? cand[i] : __builtin_inf();
                        int rank = 0;
                        for (int j = 0; j < M; ++j) {
                            const double cj = cand[j];   // same address in every lane
                            rank += (cj < ci || (cj == ci && j < i)) ? 1 : 0;
                        }
                        if (i < M && rank < k) out[rank] = __dsqrt_rn(ci);
                        const unsigned long long last = __ballot(i < M && rank == k - 1);
                        if (last) kth2 = bcast_f64(ci, (int)__builtin_ctzll(last));
                    }
                    wave_sync();
                    if (lane == 0) {
                        const int qorig = (int)(self_w & 0x7fffffffu) - q_begin;
                        const double sum = pairwise_sum_le128([&](int i) { return out[i]; }, k);
                        mean_out[qorig] = __double2float_rn(__ddiv_rn(sum, (double)k));
                        if (kth_out) kth_out[qorig] = kth2;
                    }
                    wave_sync();
                    break;
                }
                R *= M == 0 ? 8.0 : (4 * M < k ? 4.0 : 2.0);   // fewer than k points inside the ball (a whole-cloud pass always has them: n > k); see below
                continue;
            }
            // INSERT mode: certified iff k neighbours were found inside the radius the pass covered
            if (whole || kv <= rcert * rcert) {
                if (lane < 64) out[lane] = __dsqrt_rn(best);
                wave_sync();
                if (lane == 0) {
                    const int qorig = (int)(self_w & 0x7fffffffu) - q_begin;
                    const double sum = pairwise_sum_le128([&](int i) { return out[i]; }, k);
                    mean_out[qorig] = __double2float_rn(__ddiv_rn(sum, (double)k));
                    if (kth_out) kth_out[qorig] = kv;
#ifdef GSX_TREE_PROFILE
                    const unsigned long long dt = wall_clock64() - t_begin;
                    if (dt > 20000ull) printf("slow descent: %.3f ms q=(%g %g %g) bound %g R %g kth %g passes %d pops %d scans %d (%d with a full list) splits %d (%d full, %d of > 100k points) points %lld max pending %d\n", dt * 1e-5, qd[0], qd[1], qd[2], bound, R, sqrt(kv), p_pass, p_pop, p_scan, p_scan_full, p_split, p_split_full, p_bigsplit, p_pts, p_maxsp);
#endif
                }
                wave_sync();
                break;
            }
            // Fewer than k points inside the ball: the next one is wider by what the count says.  (Plain doubling took a flyer
            // whose leaf holds nothing else -- a radius to try of half a fine cell -- through 17 passes, 2^-10 to 2^6: 0.3-0.7 ms
            // for ONE query, the duration of the whole launch.  A ball that is too wide costs little here: the descent is
            // nearest-first and pruned by the running k-th distance, the radius only certifies.)
            {
                const int found = (int)__popcll(__ballot(best < 1e300));
                R *= found == 0 ? 8.0 : (4 * found < k ? 4.0 : 2.0);
            }
        }
    }
}

// ---------------------------------------------------------------- host
static int tree_blocks(const gsx_ctx *ctx, int64_t n, int per_thread)
{
    const int64_t want = (n + 256LL * per_thread - 1) / (256LL * per_thread);
    return (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cu * 16));
}

// Points a leaf may hold.  knn_leaf certifies a query whose k-th neighbour is nearer than the faces of the leaf's box grown by
// half its longest side, so a leaf must be large against the ball of k points: with 64-point leaves 1 % of the queries of the
// six-blob cloud go to the per-query kernels at k = 16, 5 % at k = 25, 15 % at 36, 29 % at 50 (4-5 ns each against ~0.5 ns in
// knn_leaf: 15 of 25 ms at k = 50).  A leaf of more than 64 points is taken in batches of 64 queries against the same candidate
// set, whose size grows with the leaf -- knn_leaf's time grows by a third from 64 to 96 points.  Measured at 10M points, ms per
// step, capacity 64 / 96 / 128 / 192 (profiles/r05_variants.txt): blobs k = 25: 8.10 / 8.16 / 8.47 / 9.92, k = 36: 13.30 /
// 11.00 / 10.51 / 12.39, k = 50: 24.60 / 19.40 / 16.56 / 16.93; scene + floaters (one density, the probe picks the shape):
// k = 25: 6.05 / 6.68 / 6.98, k = 36: 8.76 / 9.01 / 8.93, k = 50: 12.28 / 13.11 / 12.39.
static int tree_leaf_cap_for(int k) { return k <= 28 ? LEAF_CAP : (k <= 34 ? 96 : 128); }

template <int KCAP>
static int launch_leaves(gsx_ctx *ctx, TreeWs &w, int k, int leaf_cap, int64_t q_begin, int64_t q_count, float *mean_out, double *kth_out, int share,
                         int nshares)
{
    static int occ = 0;
    if (!occ) {
        GSX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, knn_leaf_kernel<KCAP>, TREE_THREADS, 0));
        occ = std::max(1, std::min(occ, 8));
    }
    static const float rf_scale = getenv("GSX_TREE_RF") ? (float)atof(getenv("GSX_TREE_RF")) : 1.1f;   // (tuning: DESIGN.md 5.8)
    hipLaunchKernelGGL((knn_leaf_kernel<KCAP>), dim3(ctx->num_cu * occ), dim3(TREE_THREADS), 0, ctx->stream, w.params.as<TreeParams>(),
                       w.keys[1].as<unsigned long long>(), w.samples.as<unsigned long long>(), w.refs.as<float4>(), w.leafstart.as<unsigned>(),
                       w.leafbl.as<unsigned char>(), k, (ctx->tree_cand_limit > 0 ? ctx->tree_cand_limit : TREE_CAND_LIMIT) / LEAF_CAP * leaf_cap, (int)q_begin, (int)q_count, rf_scale, mean_out, kth_out,
                       w.faillist.as<unsigned>(), w.failbound.as<double>(), share, nshares);
    GSX_HIP(hipGetLastError());
    return 0;
}

int launch_knn_tree(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride, int64_t n_ref, int64_t q_begin,
                    int64_t q_count, int k, float *mean_out, double *kth_out, gsx_sor_info *info, int64_t ref_only_from, int share,
                    int nshares, bool guard)
{
    if (k < 1 || k > 64) GSX_FAIL("sor (tree): k=%d not supported (1 <= k <= 64)", k);
    if (n_ref < 1 || n_ref > (int64_t)INT32_MAX - 64) GSX_FAIL("sor (tree): n=%lld out of range", (long long)n_ref);
    if (n_ref <= k) GSX_FAIL("sor (tree): k=%d needs more than %lld points", k, (long long)n_ref);
    TreeWs &w = ctx->tree_ws;
    ctx->last_knn_algo = GSX_KNN_TREE;
    const int leaf_cap = ctx->tree_leaf_cap > 0 ? std::min(std::max(ctx->tree_leaf_cap, LEAF_CAP), LEAF_CAP_MAX) : tree_leaf_cap_for(k);
    const size_t n = (size_t)n_ref;
    const int ntiles = div_up(n_ref, LEAF_TILE);
    const int bbox_blocks = std::min(tree_blocks(ctx, n_ref, 8), ctx->num_cu * 4);
    for (int b = 0; b < 2; ++b) {
        GSX_CHECK(w.keys[b].reserve(sizeof(unsigned long long) * n));
        GSX_CHECK(w.vals[b].reserve(sizeof(unsigned) * n));
    }
    GSX_CHECK(w.refs.reserve(sizeof(float4) * n));
    GSX_CHECK(w.samples.reserve(sizeof(unsigned long long) * (n / KEY_BLOCK + 2)));
    GSX_CHECK(w.blockboxes.reserve(sizeof(float4) * 2 * (n / 64 + 2)));
    GSX_CHECK(w.flags.reserve(n));
    GSX_CHECK(w.tilecnt.reserve(sizeof(unsigned) * ((size_t)ntiles + 1)));
    GSX_CHECK(w.tileoff.reserve(sizeof(unsigned) * ((size_t)ntiles + 1)));
    GSX_CHECK(w.leafstart.reserve(sizeof(unsigned) * (n + 1)));
    GSX_CHECK(w.leafbl.reserve(n));
    GSX_CHECK(w.faillist.reserve(sizeof(unsigned) * 2 * n));    // knn_leaf's list | knn_tree_near's leftovers
    GSX_CHECK(w.failbound.reserve(sizeof(double) * 2 * n));
    GSX_CHECK(w.bboxpart.reserve(sizeof(float) * 7 * (size_t)bbox_blocks));
    if (!w.params.p) {
        GSX_CHECK(w.params.reserve(sizeof(TreeParams)));
        GSX_HIP(hipMemsetAsync(w.params.p, 0, sizeof(TreeParams), ctx->stream));
    }
    size_t t_sort = 0, t_scan = 0;
    unsigned long long *k0 = w.keys[0].as<unsigned long long>(), *k1 = w.keys[1].as<unsigned long long>();
    unsigned *v0 = w.vals[0].as<unsigned>(), *v1 = w.vals[1].as<unsigned>();
    unsigned *tilecnt = w.tilecnt.as<unsigned>(), *tileoff = w.tileoff.as<unsigned>();
    GSX_HIP(rocprim::radix_sort_pairs(nullptr, t_sort, k0, k1, v0, v1, n, 0, 63, ctx->stream));
    GSX_HIP(rocprim::exclusive_scan(nullptr, t_scan, tilecnt, tileoff, 0u, (size_t)ntiles, rocprim::plus<unsigned>(), ctx->stream));
    GSX_CHECK(w.temp.reserve(std::max(t_sort, t_scan)));
    TreeParams *tp = w.params.as<TreeParams>();

    GSX_CHECK(timing_begin(ctx, GSX_T_SOR_BIN));
    hipLaunchKernelGGL(tree_bbox_kernel, dim3(bbox_blocks), dim3(256), 0, ctx->stream, x, y, z, stride, (int)n_ref,
                       w.bboxpart.as<float>(), tp, ctx->devflags.as<unsigned>(), ctx->tree_scale > 0.0 ? ctx->tree_scale : 1.0);
    if (ctx->tree_scale == 0.0 && n_ref >= PROBE_MIN_N) {   // the shape of the leaves: see "density probe" above
        float4 *smp = w.keys[1].as<float4>();              // (free until the sort)
        hipLaunchKernelGGL(tree_probe_gather_kernel, dim3(PROBE_S / 256), dim3(256), 0, ctx->stream, x, y, z, stride, (int)n_ref, smp);
        hipLaunchKernelGGL(tree_probe_kernel, dim3(PROBE_S / (4 * PROBE_PER_WAVE)), dim3(256), 0, ctx->stream, smp, (int)n_ref,
                           std::max(1, k * LEAF_CAP / leaf_cap), log2f((float)leaf_cap), tp,
                           reinterpret_cast<unsigned char *>(smp + PROBE_S));
    }
    hipLaunchKernelGGL(tree_keys_kernel, dim3(tree_blocks(ctx, n_ref, 4)), dim3(256), 0, ctx->stream, x, y, z, stride, (int)n_ref, tp,
                       k0, v0);
    GSX_HIP(hipGetLastError());
    GSX_HIP(rocprim::radix_sort_pairs(w.temp.p, t_sort, k0, k1, v0, v1, n, 0, 63, ctx->stream));
    hipLaunchKernelGGL(tree_gather_kernel, dim3(tree_blocks(ctx, n_ref, 2)), dim3(256), 0, ctx->stream, x, y, z, stride, (int)n_ref,
                       v1, w.refs.as<float4>(), w.blockboxes.as<float4>(), (int)std::min<int64_t>(ref_only_from, INT32_MAX));
    hipLaunchKernelGGL(tree_samples_kernel, dim3(tree_blocks(ctx, n_ref / KEY_BLOCK + 1, 1)), dim3(256), 0, ctx->stream, k1, (int)n_ref,
                       w.samples.as<unsigned long long>());
    hipLaunchKernelGGL(tree_leaf_flags_kernel, dim3(ntiles), dim3(256), 0, ctx->stream, k1, (int)n_ref, leaf_cap,
                       w.flags.as<unsigned char>(), tilecnt);
    GSX_HIP(hipGetLastError());
    GSX_HIP(rocprim::exclusive_scan(w.temp.p, t_scan, tilecnt, tileoff, 0u, (size_t)ntiles, rocprim::plus<unsigned>(), ctx->stream));
    hipLaunchKernelGGL(tree_leaf_compact_kernel, dim3(ntiles), dim3(256), 0, ctx->stream, w.flags.as<unsigned char>(), (int)n_ref,
                       tileoff, tilecnt, w.leafstart.as<unsigned>(), w.leafbl.as<unsigned char>(), tp);
    GSX_HIP(hipGetLastError());
    if (guard) {
        // adaptive mode (the caller synchronises anyway): a cloud with tens of thousands of points inside ONE fine cell -- two
        // scales more than 2^21 apart -- would be searched quadratically there; the caller's grid refinement re-scales instead
        hipLaunchKernelGGL(tree_leaf_max_kernel, dim3(tree_blocks(ctx, n_ref / 32 + 1, 1)), dim3(256), 0, ctx->stream, tp,
                           w.leafstart.as<unsigned>(), leaf_cap);
        unsigned max_leaf = 0;
        GSX_HIP(hipMemcpyAsync(&max_leaf, &tp->max_leaf, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        GSX_HIP(hipStreamSynchronize(ctx->stream));
        if (max_leaf > TREE_OVERFULL_LIMIT) {
            GSX_CHECK(timing_end(ctx, GSX_T_SOR_BIN));
            if (getenv("GSX_TRACE_LEVELS")) fprintf(stderr, "[gsx] tree: %u points in one fine cell -> back to the grid\n", max_leaf);
            return GSX_TREE_UNSUITABLE;
        }
    }
    GSX_CHECK(timing_end(ctx, GSX_T_SOR_BIN));

    GSX_CHECK(timing_begin(ctx, GSX_T_SOR_KNN));
    const int kk = k + 1;
    if (kk <= 9) GSX_CHECK(launch_leaves<9>(ctx, w, k, leaf_cap, q_begin, q_count, mean_out, kth_out, share, nshares));
#define GSX_LEAVES(K) else if (kk <= K) GSX_CHECK(launch_leaves<K>(ctx, w, k, leaf_cap, q_begin, q_count, mean_out, kth_out, share, nshares));
#if GSX_CAP4   // (list capacities 12, 20, 28, ...: round 5, see dispatch_bricks)
    GSX_LEAVES(13) GSX_LEAVES(17) GSX_LEAVES(21) GSX_LEAVES(25) GSX_LEAVES(29) GSX_LEAVES(33) GSX_LEAVES(37) GSX_LEAVES(41) GSX_LEAVES(45)
    GSX_LEAVES(49) GSX_LEAVES(53) GSX_LEAVES(57)
#else
    GSX_LEAVES(17) GSX_LEAVES(25) GSX_LEAVES(33) GSX_LEAVES(41) GSX_LEAVES(49) GSX_LEAVES(57)
#endif
#undef GSX_LEAVES
    else GSX_CHECK(launch_leaves<65>(ctx, w, k, leaf_cap, q_begin, q_count, mean_out, kth_out, share, nshares));
    GSX_CHECK(timing_end(ctx, GSX_T_SOR_KNN));
    GSX_CHECK(timing_begin(ctx, GSX_T_SOR_FALLBACK));
    hipLaunchKernelGGL(knn_tree_near_kernel, dim3(ctx->num_cu * 6), dim3(TREE_THREADS), 0, ctx->stream, tp, k1,
                       w.samples.as<unsigned long long>(), w.refs.as<float4>(), w.blockboxes.as<float4>(), w.faillist.as<unsigned>(),
                       w.failbound.as<double>(), k, (int)q_begin, mean_out, kth_out, w.faillist.as<unsigned>() + n, w.failbound.as<double>() + n,
                       ctx->tree_near_cell);
    hipLaunchKernelGGL(knn_tree_query_kernel, dim3(ctx->num_cu * 3), dim3(TREE_THREADS), 0, ctx->stream, tp, k1,
                       w.samples.as<unsigned long long>(), w.refs.as<float4>(), w.blockboxes.as<float4>(),
                       w.faillist.as<unsigned>() + n, w.failbound.as<double>() + n, k, (int)q_begin, (int)q_count, mean_out, kth_out);
    GSX_HIP(hipGetLastError());
    GSX_CHECK(timing_end(ctx, GSX_T_SOR_FALLBACK));

    if (info || getenv("GSX_TRACE_LEVELS")) {
        TreeParams h;
        GSX_HIP(hipMemcpyAsync(&h, tp, sizeof(TreeParams), hipMemcpyDeviceToHost, ctx->stream));
        GSX_HIP(hipStreamSynchronize(ctx->stream));
        if (getenv("GSX_TRACE_LEVELS"))
            fprintf(stderr, "[gsx] tree: n=%d fine cell %g leaves=%u (%.1f points each) fallback queries=%u, of which descents=%u\n", h.n,
                    h.s, h.nleaves, h.nleaves ? (double)h.n / h.nleaves : 0.0, h.fail_count, h.fail2_count);
        if (getenv("GSX_TRACE_LEVELS")) {
            fprintf(stderr, "[gsx] probe:");
            for (int b = 0; b < 32; ++b) fprintf(stderr, " %u", h.probe_hist[b]);
            fprintf(stderr, "\n");
        }
        if (getenv("GSX_TRACE_LEVELS"))
            fprintf(stderr, "[gsx] near: cells>512 %u, dense cell %u (attempt 0/1/2: %u %u %u), M>cap %u, M<k %u, bound M<k %u\n", h.defer_why[0], h.defer_why[1],
                    h.defer_why[5], h.defer_why[6], h.defer_why[7], h.defer_why[2], h.defer_why[3], h.defer_why[4]);
        if (h.bad_input) return gsx_ctx_check(ctx);
        if (info) {
            info->algo = GSX_KNN_TREE;
            info->grid_dim[0] = info->grid_dim[1] = info->grid_dim[2] = 0;
            info->cell_size = (float)h.s;
            info->n_cells = 0;
            info->n_bricks = h.nleaves;
            info->n_fallback = h.fail_count;
            info->n_exhaustive = h.fail2_count;
            info->n_deferred_bricks = 0;
            info->n_refined = 0;
        }
    }
    return 0;
}

// diagnostics of the last tree run of this context (synchronous)
int knn_tree_info(gsx_ctx *ctx, gsx_sor_info *info)
{
    TreeParams h;
    if (!ctx->tree_ws.params.p) GSX_FAIL("sor (tree): no run to report on");
    GSX_HIP(hipMemcpy(&h, ctx->tree_ws.params.p, sizeof(TreeParams), hipMemcpyDeviceToHost));
    info->algo = GSX_KNN_TREE;
    info->cell_size = (float)h.s;
    info->n_bricks = h.nleaves;
    info->n_fallback = h.fail_count;
    info->n_exhaustive = h.fail2_count;
    return 0;
}

}  // namespace gsx
