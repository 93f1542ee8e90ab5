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
// One 128-lane workgroup sums one 8192-element piece: two lanes own each 128-element leaf (4 of numpy's 8 accumulators
// each, 16-byte loads), the 64 leaf sums combine in the balanced tree the recursion produces for 8192.  The ragged last piece is split into its (irregular) leaves by lane 0
// and combined by the same recursion.  HBM-bound and tiny: 4 B/splat per pass, 3 passes.
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

// Round 5: one WAVE per 8192-element piece, four pieces per 256-thread workgroup (GSX_CS_WAVE=1, default): no LDS, no
// barrier and a quarter of the workgroups / arrival tickets of the two-wave version (GSX_CS_WAVE=0: round 1-4, one
// 128-thread workgroup per piece).
#ifndef GSX_CS_WAVE
#define GSX_CS_WAVE 1
#endif
constexpr int CHUNK_THREADS = GSX_CS_WAVE ? 256 : 128;
constexpr int CHUNK_PIECES = GSX_CS_WAVE ? 4 : 1;   // pieces per workgroup

// numpy's split: left half = n/2 rounded down to a multiple of 8
__device__ __forceinline__ int np_split(int n)
{
    int n2 = n / 2;
    return n2 - (n2 % 8);
}

// leaf (lo, n) of the recursion over [0, len) that contains element `pos` -- registers only
__device__ __forceinline__ void leaf_of(int len, int pos, int &lo, int &n)
{
    lo = 0;
    n = len;
#pragma unroll 1
    while (n > 128) {
        const int n2 = np_split(n);
        if (pos < lo + n2) {
            n = n2;
        } else {
            lo += n2;
            n -= n2;
        }
    }
}

// S(n) = S(n2) + S(n - n2) replayed over the leaf sums (consumed in DFS order).  The recursion is at
// most 7 deep for n <= 8192, so it is expanded at compile time: no stack arrays, no scratch memory.
template <int DEPTH>
__device__ __forceinline__ float replay_tree(int n, const float *lv, int &li)
{
    if constexpr (DEPTH == 0) {
        return lv[li++];
    } else {
        if (n <= 128) return lv[li++];
        const int n2 = np_split(n);
        const float a = replay_tree<DEPTH - 1>(n2, lv, li);
        const float b = replay_tree<DEPTH - 1>(n - n2, lv, li);
        return a + b;
    }
}

// mode 0: stats[0] = mean.  mode 1: stats[1] = std, stats[2] = threshold.
// numpy adds the per-piece sums SEQUENTIALLY, so the chain is inherently serial: the calling workgroup (every thread of it
// must call) stages 1024 piece sums at a time in LDS with coalesced loads, thread 0 adds them in order from there (one
// ds_read_b128 per four dependent adds; round 2 pulled every value through v_readlane: ~15 cycles per piece sum, 20 us at 10M
// splats -- as long as the piece sums themselves took).
constexpr int FOLD_TILE = 1024;
__device__ __forceinline__ void stats_finalize_body(const float *__restrict__ chunk_sum, int64_t n, int64_t nchunks, int mode,
                                                    float factor, float *__restrict__ stats)
{
    __shared__ __attribute__((aligned(16))) float s_fold[FOLD_TILE];
    float acc = 0.0f;
    for (int64_t base = 0; base < nchunks; base += FOLD_TILE) {
        const int m = (int)((nchunks - base) < FOLD_TILE ? (nchunks - base) : FOLD_TILE);
        __syncthreads();
        // (written by other workgroups: device-scope loads)
        {
            float t[FOLD_TILE / 64];   // every load of the tile requested before the first is stored (blockDim >= 64)
#pragma unroll
            for (int u = 0; u < FOLD_TILE / 64; ++u) {
                const int i = (int)threadIdx.x + u * (int)blockDim.x;
                t[u] = i < m ? __hip_atomic_load(&chunk_sum[base + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < FOLD_TILE / 64; ++u) {
                const int i = (int)threadIdx.x + u * (int)blockDim.x;
                if (i < m) s_fold[i] = t[u];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int i = 0;
            for (; i + 64 <= m; i += 64) {   // 16 LDS reads in flight, then 64 dependent adds
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4 *>(&s_fold[i + 4 * u]);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    acc += v[u].x;
                    acc += v[u].y;
                    acc += v[u].z;
                    acc += v[u].w;
                }
            }
            for (; i < m; ++i) acc += s_fold[i];
        }
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

__global__ __launch_bounds__(64) void stats_finalize_kernel(const float *__restrict__ chunk_sum, int64_t n, int64_t nchunks,
                                                            int mode, float factor, float *__restrict__ stats)
{
    stats_finalize_body(chunk_sum, n, nchunks, mode, factor, stats);
}

// Single-GPU path: the last workgroup of chunk_sums_kernel to arrive folds the piece sums itself (ticket != NULL),
// which saves the two one-wave launches; the multi-GPU path exchanges the piece sums first and keeps them separate.
struct StatsFold {
    unsigned *ticket;
    float factor;
    float *stats_out;
};

template <bool SQ>
__device__ __forceinline__ void chunk_sum_body(const float *__restrict__ a, int64_t n, const float *__restrict__ stats,
                                               float *__restrict__ chunk_sum);
__host__ __device__ constexpr int64_t chunk_blocks(int64_t nchunks) { return (nchunks + CHUNK_PIECES - 1) / CHUNK_PIECES; }

template <bool SQ>
__global__ __launch_bounds__(CHUNK_THREADS) void chunk_sums_kernel(const float *__restrict__ a, int64_t n,
                                                                   const float *__restrict__ stats,
                                                                   float *__restrict__ chunk_sum, StatsFold fold)
{
    chunk_sum_body<SQ>(a, n, stats, chunk_sum);
    if (!fold.ticket) return;
    __shared__ unsigned s_last;
    __builtin_amdgcn_s_waitcnt(0);   // the write-through piece sum has completed before the ticket
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(fold.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) *fold.ticket = 0;
    stats_finalize_body(chunk_sum, n, (n + NP_BUF - 1) / NP_BUF, SQ ? 1 : 0, fold.factor, fold.stats_out);
}

template <bool SQ>
__device__ __forceinline__ void chunk_sum_body(const float *__restrict__ a, int64_t n, const float *__restrict__ stats,
                                               float *__restrict__ chunk_sum)
{
    __shared__ int s_leaf_start[160];
    __shared__ int s_leaf_len[160];
    __shared__ float s_leaf_val[160];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const float mean = SQ ? stats[0] : 0.0f;
    const int64_t nchunks = (n + NP_BUF - 1) / NP_BUF;
#if GSX_CS_WAVE
    // ---- wave wv of the workgroup owns piece 4 * blockIdx.x + wv.  A full piece = 64 leaves of 128 elements; two lanes per
    // leaf: lane (leaf, hh) owns accumulators 4hh..4hh+3 (numpy's r[0..7]) and walks the leaf's 16 rows with one 16-byte
    // load per row; the wave takes leaves 0..31, then 32..63 (16 loads in flight each time) and combines the two 4096-element
    // halves -- the balanced tree numpy's recursion produces for 8192.
    int64_t c = (int64_t)blockIdx.x * CHUNK_PIECES + wv;
    const bool ragged_block = (n % NP_BUF) != 0 && (int64_t)blockIdx.x == chunk_blocks(nchunks) - 1;   // block-uniform: it holds the ragged piece
    if (c < n / NP_BUF) {   // wave-uniform: a full piece
        const float *pp = a + c * NP_BUF;
        float half[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int leaf = hf * 32 + (lane >> 1), hh = lane & 1;
            const float4 *q = reinterpret_cast<const float4 *>(pp + leaf * 128 + hh * 4);
            float4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = q[2 * i];  // +8 floats per row
            float r0 = elem<SQ>(v[0].x, mean), r1 = elem<SQ>(v[0].y, mean), r2 = elem<SQ>(v[0].z, mean), r3 = elem<SQ>(v[0].w, mean);
#pragma unroll
            for (int i = 1; i < 16; ++i) {
                r0 += elem<SQ>(v[i].x, mean); r1 += elem<SQ>(v[i].y, mean); r2 += elem<SQ>(v[i].z, mean); r3 += elem<SQ>(v[i].w, mean);
            }
            float s = (r0 + r1) + (r2 + r3);      // (r0+r1)+(r2+r3)  |  (r4+r5)+(r6+r7)
            s = s + __shfl_xor(s, 1);             // leaf sum (both lanes of the pair hold it)
#pragma unroll
            for (int off = 2; off < 64; off <<= 1) s = s + __shfl_xor(s, off);   // 128 -> 256 -> ... -> 4096
            half[hf] = s;
        }
        if (lane == 0) __hip_atomic_store(&chunk_sum[c], half[0] + half[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // 4096 + 4096
    }
    if (!ragged_block) return;
    // the ragged last piece: the whole workgroup (the waves that summed full pieces above included)
    c = n / NP_BUF;
    const float *p = a + c * NP_BUF;
    const int len = (int)(n - c * NP_BUF);
#else
    __shared__ float s_half[2];
    const int64_t c = blockIdx.x;
    const float *p = a + c * NP_BUF;
    const int len = (int)((n - c * NP_BUF) < NP_BUF ? (n - c * NP_BUF) : NP_BUF);

    if (len == NP_BUF) {  // block-uniform
        // 64 leaves of 128 elements; two lanes per leaf: lane (leaf, hh) owns accumulators 4hh..4hh+3
        // (numpy's r[0..7]) and walks the leaf's 16 rows with one 16-byte load per row.  (Round 3 tried parking the piece
        // in LDS first with fully coalesced loads: 34 KB per workgroup cost more occupancy than the coalescing returned --
        // stats 0.099 -> 0.110 ms per 10M-splat step.)
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
        if (threadIdx.x == 0) __hip_atomic_store(&chunk_sum[c], s_half[0] + s_half[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // 4096 + 4096
        return;
    }
#endif

    // ragged last piece.  Every leaf of numpy's recursion is >= 57 elements, so it contains a multiple
    // of 32: each thread descends the recursion for two such positions (registers only) and the
    // thread whose position is the FIRST multiple of 32 inside a leaf owns it.  Owners are ranked in
    // position order (= the recursion's DFS order), 8 lanes per leaf sum the leaves, thread 0 replays
    // the tree over the leaf sums with a compile-time-expanded recursion.
    int *ls = s_leaf_start, *ll = s_leaf_len;
    float *lv = s_leaf_val;
    __shared__ int s_cnt[CHUNK_THREADS / 64];
    int my_lo[2], my_n[2];
    bool own[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int pos = 32 * ((int)threadIdx.x * 2 + u);   // positions in increasing thread order (threads >= 128: beyond any piece)
        own[u] = false;
        my_lo[u] = 0; my_n[u] = 0;
        if (pos < len || (pos == 0 && len > 0)) {
            leaf_of(len, pos, my_lo[u], my_n[u]);
            own[u] = pos - my_lo[u] < 32 && (pos == 0 || pos - 32 < my_lo[u]);
        }
    }
    const int mine = (int)own[0] + (int)own[1];
    // exclusive prefix of `mine` over the 128 threads (wave scan + one cross-wave add)
    int inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_cnt[wv] = inc;
    __syncthreads();
    int rank = inc - mine, nleaf = 0;
#pragma unroll
    for (int i = 0; i < CHUNK_THREADS / 64; ++i) {
        if (i < wv) rank += s_cnt[i];
        nleaf += s_cnt[i];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (own[u]) {
            ls[rank] = my_lo[u];
            ll[rank] = my_n[u];
            ++rank;
        }
    __syncthreads();
    const int j = threadIdx.x & 7;
    for (int l0 = 0; l0 < nleaf; l0 += CHUNK_THREADS / 8) {  // uniform trip count
        const int l = l0 + (int)(threadIdx.x >> 3);
        const bool live = l < nleaf;
        const int L = live ? ll[l] : 0;
        const float *q = p + (live ? ls[l] : 0);
        float res;
        if (L < 8) {
            res = 0.0f;
            for (int i = 0; i < L; ++i) res += elem<SQ>(q[i], mean);  // every lane of the group, same value
        } else {
            const int rows = L >> 3;
            float r = elem<SQ>(q[j], mean);
            for (int i = 1; i < rows; ++i) r += elem<SQ>(q[8 * i + j], mean);
            // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) -- shuffles stay inside the 8-lane group
            r = r + __shfl_xor(r, 1);
            r = r + __shfl_xor(r, 2);
            r = r + __shfl_xor(r, 4);
            res = r;
            for (int i = rows * 8; i < L; ++i) res += elem<SQ>(q[i], mean);
        }
        if (live && j == 0) lv[l] = res;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int li = 0;
        __hip_atomic_store(&chunk_sum[c], replay_tree<7>(len, lv, li), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    const int blocks = (int)chunk_blocks(nchunks);
    const float tf = (float)factor;  // python float is a weak scalar: rounded to f32 first
    const StatsFold fold{ctx->devflags.as<unsigned>() + 8, tf, stats_dev};   // word 8 of the flag block: arrival ticket
    hipLaunchKernelGGL((chunk_sums_kernel<false>), dim3(blocks), dim3(CHUNK_THREADS), 0, ctx->stream, md, n, stats_dev, cs, fold);
    hipLaunchKernelGGL((chunk_sums_kernel<true>), dim3(blocks), dim3(CHUNK_THREADS), 0, ctx->stream, md, n, stats_dev, cs, fold);
    GSX_HIP(hipGetLastError());
    return 0;
}

// Multi-GPU (dist_slab.hip): numpy's reduction is "pairwise inside an 8192-element piece, pieces added sequentially",
// so the piece sums can be computed wherever the elements live and combined anywhere, bit-exactly.
int launch_sor_piece_sums(gsx_ctx *ctx, const float *a, int64_t n, const float *mean_dev, float *piece_out)
{
    const int blocks = (int)chunk_blocks((n + NP_BUF - 1) / NP_BUF);
    if (mean_dev)
        hipLaunchKernelGGL((chunk_sums_kernel<true>), dim3(blocks), dim3(CHUNK_THREADS), 0, ctx->stream, a, n, mean_dev, piece_out, StatsFold{nullptr, 0.0f, nullptr});
    else
        hipLaunchKernelGGL((chunk_sums_kernel<false>), dim3(blocks), dim3(CHUNK_THREADS), 0, ctx->stream, a, n, mean_dev, piece_out, StatsFold{nullptr, 0.0f, nullptr});
    GSX_HIP(hipGetLastError());
    return 0;
}

int launch_sor_stats_from_pieces(gsx_ctx *ctx, const float *pieces, int64_t npieces, int64_t n_total, int mode, double factor,
                                 float *stats_dev)
{
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(1), dim3(64), 0, ctx->stream, pieces, n_total, npieces, mode, (float)factor,
                       stats_dev);
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
