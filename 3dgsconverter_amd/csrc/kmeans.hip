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
#include <vector>

#include "gsx_common.h"
#include "sog_math.h"

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

// ---- matrix-core assign (D = 9, 24, 45; K >= 64) -------------------------------------------------------------------
// argmin_c |x - c|^2 = argmin_c (|c|^2 - 2 x.c): the K x N inner products are a GEMM.  It runs on
// v_mfma_f32_32x32x16_bf16 as a FILTER, the same way knn_brick's phase 1 does (sor_grid.hip): every f32 value is split
// into two bf16 pieces (v = vh + vl, |v - vh - vl| <= 2^-18 |v|), x.c ~ xh.ch + xh.cl + xl.ch (three MFMAs per 16
// dimensions), |c|^2 rides in the padding slots of the K dimension as three bf16 pieces against 1.0.  Per point the THREE
// smallest approximate values are tracked with their indices.  If the two smallest are further apart than twice the
// error bound E below, the smallest IS the centroid the exact f32 kernel would pick.  Else, if the third is further
// than 2E from the first, only the first two can win and the reference's own arithmetic (gpu_ops.py:57-73: f32, dims
// in order, strict '<' = first minimum) is evaluated on those two rows in place.  Else (three-way near tie, duplicate
// centroids, NaN rows) the point goes to kmeans_assign_exact_list_kernel, which scans all K centroids that way.
// Labels are therefore IDENTICAL to kmeans_assign_kernel's.  Error budget of the approximate value against the exact-kernel's f32 distance minus |x|^2,
// with nx = |x|, nc = max_c |c|:
//   dropped xl.cl and split residuals      <= 2 * 3 * 2^-18 nx nc          = 2.3e-5 nx nc
//   f32 accumulation of <= 144 products    <= 144 * 2^-24 (2 nx nc + nc^2) = 8.6e-6 (2 nx nc + nc^2)
//   |c|^2 in f32 and its three bf16 pieces <= 2.8e-6 nc^2
//   the exact kernel's own fmaf chain      <= 47 * 2^-24 (nx + nc)^2       = 2.8e-6 (nx + nc)^2
//   index bits in the 4 low mantissa bits  <= 15 * 2^-24 (2 nx nc + nc^2)  = 9e-7 (2 nx nc + nc^2)
//   E := 2^-14 (nx nc + nc^2) + 2^-18 nx^2 covers the sum.
typedef __bf16 kbf16x8 __attribute__((ext_vector_type(8)));
typedef float kf32x16 __attribute__((ext_vector_type(16)));
typedef unsigned ku32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned km_cvt_pk_bf16(float lo, float hi)  // RNE, lo -> bits 0..15
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float km_bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float km_bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// eight f32 values -> their bf16 high pieces and the bf16 of the remainders, packed in MFMA operand order
__device__ __forceinline__ void km_split8(const float (&v)[8], ku32x4 &hi, ku32x4 &lo)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned h = km_cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
        hi[i] = h;
        lo[i] = km_cvt_pk_bf16(v[2 * i] - km_bf_lo(h), v[2 * i + 1] - km_bf_hi(h));
    }
}

// raw min / max: fminf / fmaxf make LLVM add a canonicalising v_max per operand (see knn_common.h); NaN operands are
// ignored by the hardware ops, which is what the tracking wants
__device__ __forceinline__ float km_min(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float km_max(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float km_min3(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ float km_med3(float a, float b, float c)
{
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// ---- many independent problems per launch (round 5: gsx_kmeans_lloyd_batch_dev) ------------------------------------------
// The SOG palette is 64 independent Lloyd problems of the same shape (formats/sog.py:536-552).  Run one after the other --
// or on a few concurrent streams -- their iterations are ~400 latency-bound launches; here every kernel of the matrix-core
// path takes the PROBLEM from blockIdx.y and rebases its pointers: rows / labels / list by the problem's row offset,
// centroids / sums by k*D, counts / starts / cursors by k, operand words and the meta block by their sizes.  One problem
// (the former launches) is the batch of one: off == nullptr, blockIdx.y == 0, every offset 0.
struct KmBatch {
    const int64_t *off;   // device: nprob + 1 row offsets (nullptr: ONE problem of n1 rows)
    int64_t n1;
};
__device__ __forceinline__ int64_t km_problem_rows(const KmBatch &b, int64_t &n)
{
    const int64_t r0 = b.off ? b.off[blockIdx.y] : 0;
    n = b.off ? b.off[blockIdx.y + 1] - r0 : b.n1;
    return r0;
}
constexpr int KM_META_WORDS = 32;   // per problem: 2 x 16 words, alternating by iteration

constexpr int km_dp(int d) { return (d + 3 + 15) / 16 * 16; }  // K-dimension incl. the three |c|^2 slots: 16, 32, 48

// Centroid operands for the whole iteration: for tile t (32 centroids), 16-dimension slice j, variant v (0 = high pieces
// + |c|^2 slots, 1 = low pieces) one 16-byte word per lane at ((t * NS + j) * 2 + v) * 64 + lane, lane l = row l & 31,
// k = 16 j + 8 (l >> 5) ... + 7.  Also *cmax2 = max |c|^2.
template <int D>
__global__ __launch_bounds__(64) void kmeans_centroid_operands_kernel(const float *__restrict__ cent, int k,
                                                                      ku32x4 *__restrict__ opnd, float *__restrict__ cmax2)
{
    constexpr int DP = km_dp(D), NS = DP / 16;
    cent += (size_t)blockIdx.y * k * D;                                  // problem blockIdx.y (KmBatch)
    opnd += (size_t)blockIdx.y * ((k + 31) / 32) * NS * 2 * 64;
    cmax2 += (size_t)blockIdx.y * KM_META_WORDS;
    const int t = blockIdx.x / NS, j = blockIdx.x % NS;
    const int lane = threadIdx.x;
    const int c = 32 * t + (lane & 31), k0 = 16 * j + 8 * (lane >> 5);
    float v[8];
    float n2 = 0.0f;
    if (c < k) {
        for (int d = 0; d < D; ++d) n2 = __builtin_fmaf(cent[(int64_t)c * D + d], cent[(int64_t)c * D + d], n2);
        if (j == 0 && lane < 32) atomicMax(reinterpret_cast<int *>(cmax2), __float_as_int(n2));  // n2 >= 0: int order = float order
    } else {
        n2 = 1.0e30f;  // padding rows never win
    }
    // three bf16 pieces of |c|^2
    const unsigned p1 = km_cvt_pk_bf16(n2, n2);
    const float r1 = n2 - km_bf_lo(p1);
    const unsigned p2 = km_cvt_pk_bf16(r1, r1);
    const float r2 = r1 - km_bf_lo(p2);
    bool norm_slot[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int d = k0 + i;
        norm_slot[i] = d >= D && d < D + 3;
        v[i] = (c < k && d < D) ? -2.0f * cent[(int64_t)c * D + d] : 0.0f;
    }
    ku32x4 hi, lo;
    km_split8(v, hi, lo);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (norm_slot[i]) {
            const int piece = k0 + i - D;
            const float val = piece == 0 ? km_bf_lo(p1) : (piece == 1 ? km_bf_lo(p2) : r2);
            const unsigned bits = km_cvt_pk_bf16(val, val) & 0xffffu;
            const int w = i >> 1, sh = (i & 1) * 16;
            hi[w] = (hi[w] & ~(0xffffu << sh)) | (bits << sh);
            lo[w] = lo[w] & ~(0xffffu << sh);
        }
    opnd[((size_t)(t * NS + j) * 2 + 0) * 64 + lane] = hi;
    opnd[((size_t)(t * NS + j) * 2 + 1) * 64 + lane] = lo;
}

// Round 4: the END of one iteration and the START of the next in one launch (formerly finalize_reset, a 64-byte memset and
// centroid_operands: three of the ~9 launches of a SOG chunk iteration, each a few latency-bound microseconds).  Block
// (tile t, slice j), lane l owns the 8 dimensions k0 .. k0+7 of centroid 32 t + (l & 31) -- every (centroid, dimension) has
// exactly one owner, which divides the float64 sum (gpu_ops.py:91-96), writes the centroid and turns it into operand words
// with kmeans_centroid_operands_kernel's arithmetic; |c|^2 is recomputed by every lane from all D sums of its centroid
// (the same expression, so the same bits).  The accumulators are NOT re-zeroed here (other blocks still read them): the
// label scatter zeroes the sums and the exact-list kernel the counts of the NEXT iteration.  meta_next[0] receives
// max |c|^2 (zeroed two iterations ago), meta_done (this iteration's list length, ticket, ...) is zeroed for the one after.
template <int D>
__global__ __launch_bounds__(64) void kmeans_finalize_operands_kernel(const double *__restrict__ sums, const unsigned *__restrict__ counts,
                                                                      int k, float *__restrict__ cent, ku32x4 *__restrict__ opnd,
                                                                      unsigned *__restrict__ meta_next, unsigned *__restrict__ meta_done)
{
    constexpr int DP = km_dp(D), NS = DP / 16;
    sums += (size_t)blockIdx.y * k * D;                                  // problem blockIdx.y (KmBatch)
    counts += (size_t)blockIdx.y * k;
    cent += (size_t)blockIdx.y * k * D;
    opnd += (size_t)blockIdx.y * ((k + 31) / 32) * NS * 2 * 64;
    meta_next += (size_t)blockIdx.y * KM_META_WORDS;
    meta_done += (size_t)blockIdx.y * KM_META_WORDS;
    const int t = blockIdx.x / NS, j = blockIdx.x % NS;
    const int lane = threadIdx.x;
    const int c = 32 * t + (lane & 31), k0 = 16 * j + 8 * (lane >> 5);
    if (blockIdx.x == 0 && lane < 16) meta_done[lane] = 0u;
    float v[8];
    float n2 = 0.0f;
    float inv = 0.0f;
    bool live = false;
    if (c < k) {
        const unsigned cnt = counts[c];
        live = cnt > 0;
        inv = live ? 1.0f / (float)cnt : 0.0f;
        double sv[D];   // all D loads in flight (a rolled loop made this one-wave-per-block kernel 15 us: latency)
#pragma unroll
        for (int d = 0; d < D; ++d) sv[d] = sums[(int64_t)c * D + d];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float cv = live ? (float)sv[d] * inv : 0.0f;   // empty cluster -> 0 (gpu_ops.py:78-96)
            n2 = __builtin_fmaf(cv, cv, n2);
        }
        if (j == 0 && lane < 32) atomicMax(reinterpret_cast<int *>(meta_next), __float_as_int(n2));  // n2 >= 0: int order = float order
    } else {
        n2 = 1.0e30f;  // padding rows never win
    }
    const unsigned p1 = km_cvt_pk_bf16(n2, n2);
    const float r1 = n2 - km_bf_lo(p1);
    const unsigned p2 = km_cvt_pk_bf16(r1, r1);
    const float r2 = r1 - km_bf_lo(p2);
    bool norm_slot[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int d = k0 + i;
        norm_slot[i] = d >= D && d < D + 3;
        float cv = 0.0f;
        if (c < k && d < D) {
            cv = live ? (float)sums[(int64_t)c * D + d] * inv : 0.0f;
            cent[(int64_t)c * D + d] = cv;
        }
        v[i] = -2.0f * cv;
    }
    ku32x4 hi, lo;
    km_split8(v, hi, lo);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (norm_slot[i]) {
            const int piece = k0 + i - D;
            const float val = piece == 0 ? km_bf_lo(p1) : (piece == 1 ? km_bf_lo(p2) : r2);
            const unsigned bits = km_cvt_pk_bf16(val, val) & 0xffffu;
            const int w = i >> 1, sh = (i & 1) * 16;
            hi[w] = (hi[w] & ~(0xffffu << sh)) | (bits << sh);
            lo[w] = lo[w] & ~(0xffffu << sh);
        }
    opnd[((size_t)(t * NS + j) * 2 + 0) * 64 + lane] = hi;
    opnd[((size_t)(t * NS + j) * 2 + 1) * 64 + lane] = lo;
}

constexpr int KM_MF_WAVES = 4;
constexpr int KM_MF_PT = 1;                          // 32-point tiles per wave (SOG chunk: 53 us with 1, 59 with 2, 61 with 3)
constexpr int KM_MF_TILE = KM_MF_WAVES * KM_MF_PT * 32;  // points per workgroup
constexpr int KM_PF = 2;                             // centroid tiles requested ahead (1 / 2 / 3: 59.1 / 59.2 / 59.2 us)
// (centroid operand tiles staged per workgroup in LDS instead of streamed per wave from L2: 89 vs 62 us -- a workgroup
//  barrier per tile; removed)

// labels only: the update runs as a sort-by-label segmented reduction (kmeans_update_* below), not as 45 float64
// atomics per point (7M per SOG chunk iteration: ~85 us of the fused VALU kernel's 340)
template <int D>
__global__ __launch_bounds__(64 * KM_MF_WAVES) void kmeans_assign_mfma_kernel(const float *__restrict__ data, int64_t n,
                                                                             const ku32x4 *__restrict__ opnd, int ktiles,
                                                                             const float *__restrict__ cmax2,
                                                                             int32_t *__restrict__ labels,
                                                                             unsigned *__restrict__ unc_list, unsigned *__restrict__ unc_count,
                                                                             KmBatch kb)
{
    constexpr int DP = km_dp(D), NS = DP / 16, LS = DP + 1;   // odd LDS row stride
    constexpr int AW = NS * 2 * 64;                           // operand words (16 B) of one centroid tile
    {
        const int64_t r0 = km_problem_rows(kb, n);             // problem blockIdx.y
        data += r0 * D;
        labels += r0;
        unc_list += r0;
        opnd += (size_t)blockIdx.y * ktiles * AW;
        cmax2 += (size_t)blockIdx.y * KM_META_WORDS;
        unc_count += (size_t)blockIdx.y * KM_META_WORDS;
    }
    __shared__ float s_tile[KM_MF_TILE * LS];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float nc2 = *cmax2, nc = __builtin_sqrtf(nc2);
    const int64_t tiles = (n + KM_MF_TILE - 1) / KM_MF_TILE;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t base = tile * KM_MF_TILE;
        const int rows = (int)(n - base < KM_MF_TILE ? n - base : KM_MF_TILE);
        __syncthreads();
        for (int e = threadIdx.x; e < rows * D; e += 64 * KM_MF_WAVES) {   // contiguous rows: one coalesced copy
            const int r = e / D;
            s_tile[r * LS + (e - r * D)] = data[base * D + e];
        }
        __syncthreads();
        // this wave's point operands: lane l = point (l & 31) of each of its tiles, dimensions 16 j + 8 (l >> 5) ... + 7
        ku32x4 bh[KM_MF_PT][NS], bl[KM_MF_PT][NS];
        float nx2[KM_MF_PT];
#pragma unroll
        for (int pt = 0; pt < KM_MF_PT; ++pt) {
            const int r = (wv * KM_MF_PT + pt) * 32 + (lane & 31);
            const int rr = r < rows ? r : rows - 1;
            float acc2 = 0.0f;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int d = 16 * j + 8 * (lane >> 5) + i;
                    v[i] = d < D ? s_tile[rr * LS + d] : 0.0f;
                    acc2 = __builtin_fmaf(v[i], v[i], acc2);
                }
                km_split8(v, bh[pt][j], bl[pt][j]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {   // 1.0 against the three |c|^2 pieces, in the high operand only
                    const int d = 16 * j + 8 * (lane >> 5) + i;
                    if (d >= D && d < D + 3) bh[pt][j][i >> 1] |= 0x3f80u << ((i & 1) * 16);
                }
            }
            nx2[pt] = acc2 + __shfl_xor(acc2, 32);
        }
        float best[KM_MF_PT], second[KM_MF_PT];
        int btile[KM_MF_PT];
#pragma unroll
        for (int pt = 0; pt < KM_MF_PT; ++pt) {
            best[pt] = __builtin_inff();
            second[pt] = __builtin_inff();
            btile[pt] = 0;
        }
        ku32x4 a[NS][2], a_pf[KM_PF][NS][2];   // a_pf[i]: tile t + 1 + i, requested KM_PF tiles ahead (an L2 round trip is
                                               // longer than one tile's ~600 cycles of MFMA work)
        {
#pragma unroll
            for (int j = 0; j < NS; ++j)
#pragma unroll
                for (int v = 0; v < 2; ++v) a[j][v] = opnd[(size_t)(j * 2 + v) * 64 + lane];
#pragma unroll
            for (int i = 0; i + 1 < KM_PF; ++i) {
                const int tt = i + 1 < ktiles ? i + 1 : ktiles - 1;
#pragma unroll
                for (int j = 0; j < NS; ++j)
#pragma unroll
                    for (int v = 0; v < 2; ++v) a_pf[i][j][v] = opnd[(size_t)tt * AW + (size_t)(j * 2 + v) * 64 + lane];
            }
        }
        for (int t = 0; t < ktiles; ++t) {
            {
                const int tn = t + KM_PF < ktiles ? t + KM_PF : ktiles - 1;
#pragma unroll
                for (int j = 0; j < NS; ++j)
#pragma unroll
                    for (int v = 0; v < 2; ++v) a_pf[KM_PF - 1][j][v] = opnd[(size_t)tn * AW + (size_t)(j * 2 + v) * 64 + lane];
            }
#pragma unroll
            for (int pt = 0; pt < KM_MF_PT; ++pt) {
                kf32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const kbf16x8 ah = __builtin_bit_cast(kbf16x8, a[j][0]), al = __builtin_bit_cast(kbf16x8, a[j][1]);
                    const kbf16x8 xh = __builtin_bit_cast(kbf16x8, bh[pt][j]), xl = __builtin_bit_cast(kbf16x8, bl[pt][j]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, acc, 0, 0, 0);
                }
                // lane l: column = its point, accumulator r = centroid row (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the tile.
                // The two smallest of the 16 values by a tournament on (lo <= hi) pairs -- 37 min/max ops instead of
                // 16 x (compare + select + 3 min/max); the accumulator number rides in the 4 low mantissa bits
                // (<= 15 ulp, inside E), the tile number is tracked once per tile.
                float lo[8], hi[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float a0 = __uint_as_float((__float_as_uint(acc[2 * q]) & ~0xfu) | (unsigned)(2 * q));
                    const float a1 = __uint_as_float((__float_as_uint(acc[2 * q + 1]) & ~0xfu) | (unsigned)(2 * q + 1));
                    lo[q] = km_min(a0, a1);
                    hi[q] = km_max(a0, a1);
                }
#pragma unroll
                for (int w = 4; w >= 1; w >>= 1)
#pragma unroll
                    for (int q = 0; q < w; ++q) {   // merge the sorted pairs q and q + w: two smallest of four
                        const float m = km_max(lo[q], lo[q + w]);
                        lo[q] = km_min(lo[q], lo[q + w]);
                        hi[q] = km_min3(m, hi[q], hi[q + w]);
                    }
                const float m = km_max(best[pt], lo[0]);
                btile[pt] = lo[0] < best[pt] ? t : btile[pt];
                best[pt] = km_min(best[pt], lo[0]);
                second[pt] = km_min3(m, second[pt], hi[0]);
            }
            {
#pragma unroll
                for (int j = 0; j < NS; ++j)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        a[j][v] = a_pf[0][j][v];
#pragma unroll
                        for (int i = 0; i + 1 < KM_PF; ++i) a_pf[i][j][v] = a_pf[i + 1][j][v];
                    }
            }
        }
        // the two half-waves hold the same points (different centroid rows): merge, certify, publish
#pragma unroll
        for (int pt = 0; pt < KM_MF_PT; ++pt) {
            const int rb = (int)(__float_as_uint(best[pt]) & 0xfu);   // accumulator number of this lane's best
            const int bidx = 32 * btile[pt] + (rb & 3) + 8 * (rb >> 2) + 4 * (lane >> 5);
            const float b2 = __shfl_xor(best[pt], 32), s2 = __shfl_xor(second[pt], 32);
            const int i2 = __shfl_xor(bidx, 32);
            const float mb = fminf(best[pt], b2);
            const float ms = fminf(fmaxf(best[pt], b2), fminf(second[pt], s2));
            const int mi = b2 < best[pt] ? i2 : bidx;
            const float nx = __builtin_sqrtf(nx2[pt]);
            const float E = 6.1035156e-5f * (nx * nc + nc2) + 3.8146973e-6f * nx2[pt];
            const bool sure = (ms - mb) > 2.0f * E;   // false for NaN / inf rows and exact ties
            const int r = (wv * KM_MF_PT + pt) * 32 + (lane & 31);
            if (lane < 32 && r < rows) {
                if (sure) labels[base + r] = mi;
                else unc_list[atomicAdd(unc_count, 1u)] = (unsigned)(base + r);
            }
        }
    }
}

#include "kmeans_cs.h"

// the points the matrix-core filter could not certify (near ties: ~0.3 % on SOG data): the reference's arithmetic
// (gpu_ops.py:57-73) over ALL centroids, one wave per point; lane = centroids lane, lane + 64, ...; the lowest index wins
// inside a lane by the strict '<' and across lanes by the merge.  D is a compile-time constant so that a lane's 45 loads
// per centroid are all in flight (a runtime loop was latency-bound: 90 us for a few hundred points).
constexpr int KM_EX_WAVES = 16;   // waves per workgroup of the exact-list kernel: each scans K / 16 centroids for the same 64 points
template <int D>
__global__ __launch_bounds__(64 * KM_EX_WAVES) void kmeans_assign_exact_list_kernel(const float *__restrict__ data,
                                                                       const float *__restrict__ cent, int k,
                                                                       const unsigned *__restrict__ list,
                                                                       const unsigned *__restrict__ list_count,
                                                                       int32_t *__restrict__ labels,
                                                                       unsigned *__restrict__ zero_counts /* nullable: k words to clear */,
                                                                       KmBatch kb)
{
    {
        int64_t n_;
        const int64_t r0 = km_problem_rows(kb, n_);            // problem blockIdx.y
        data += r0 * D;
        labels += r0;
        list += r0;
        cent += (size_t)blockIdx.y * k * D;
        list_count += (size_t)blockIdx.y * KM_META_WORDS;
        if (zero_counts) zero_counts += (size_t)blockIdx.y * k;
    }
    if (zero_counts)   // the label histogram that follows accumulates into these (round 4: nobody else re-zeroes them)
        for (int i = blockIdx.x * 64 * KM_EX_WAVES + threadIdx.x; i < k; i += gridDim.x * 64 * KM_EX_WAVES) zero_counts[i] = 0u;
    // Round 5: one LANE per point, the point's D coordinates in registers; a workgroup takes 64 points of the list, its sixteen
    // waves scan a sixteenth of the centroids each in index order -- every centroid row is a WAVE-UNIFORM address (scalar
    // loads, one fetch serves 64 points) -- with the reference's arithmetic (gpu_ops.py:57-73: diff, fma chain over d,
    // strict '<' keeps the first minimum); the parts meet in LDS, lower part first, strict '<' again: the lowest
    // index among equal distances wins, as in the sequential scan.  (Round 4 gave every point a workgroup of its own, which
    // read all K x D centroid values per point through the vector memory path: 184 KB per point at K = 1024, D = 45 --
    // 0.88 ms per iteration of the 10M-splat palette, a fifth of it, for 0.3 % of the points.)
    __shared__ float s_best[KM_EX_WAVES][64];
    __shared__ int s_bi[KM_EX_WAVES][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned cnt = *list_count;
    const int c_lo = (int)(((long long)k * wv) / KM_EX_WAVES), c_hi = (int)(((long long)k * (wv + 1)) / KM_EX_WAVES);
    for (unsigned g = blockIdx.x * 64u; g < cnt; g += gridDim.x * 64u) {
        const bool live = g + (unsigned)lane < cnt;
        const int64_t i = list[live ? g + (unsigned)lane : g];
        float x[D];
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = data[i * D + d];
        float best = 1e20f;  // gpu_ops.py:60
        int bi = -1;
        for (int c = c_lo; c < c_hi; ++c) {
            const float *__restrict__ cc = cent + (int64_t)c * D;   // wave-uniform
            float dist = 0.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float diff = x[d] - cc[d];
                dist = __builtin_fmaf(diff, diff, dist);
            }
            if (dist < best) {
                best = dist;
                bi = c;
            }
        }
        s_best[wv][lane] = best;
        s_bi[wv][lane] = bi;
        __syncthreads();
        if (wv == 0 && live) {
#pragma unroll
            for (int w = 1; w < KM_EX_WAVES; ++w) {
                const float ob = s_best[w][lane];
                const int oi = s_bi[w][lane];
                // the sequential scan keeps the FIRST minimum: a later part only wins with a strictly smaller distance
                const bool take = oi >= 0 && (bi < 0 || ob < best);
                best = take ? ob : best;
                bi = take ? oi : bi;
            }
            labels[i] = bi;
        }
        __syncthreads();
    }
}

// ---- update as a segmented reduction (k_means_update, gpu_ops.py:75-96) ---------------------------------------------
// labels -> counts[K] (LDS-aggregated histogram) -> exclusive scan -> point indices grouped by label -> one wave per
// centroid sums its rows (lanes = dimensions, rows read coalesced) in float64 and divides: no per-element atomics, no
// sums buffer, no separate finalize.  Points with label -1 (unassignable) belong to no cluster.
// starts[c] = exclusive scan of counts; cursor[c] = starts[c] -- by ONE workgroup of 256 threads (every thread must call)
__device__ __forceinline__ void kmeans_label_scan_body(const unsigned *__restrict__ counts, int k, unsigned *__restrict__ starts,
                                                       unsigned *__restrict__ cursor)
{
    __shared__ unsigned s_w[4];
    __shared__ unsigned s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int b = 0; b < k; b += 256) {
        const int i = b + threadIdx.x;
        // (written by other workgroups' atomics: device-scope loads)
        const unsigned v = i < k ? __hip_atomic_load(&counts[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        unsigned inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(inc, off);
            if ((int)(threadIdx.x & 63) >= off) inc += o;
        }
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
        __syncthreads();
        unsigned pre = s_carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) pre += s_w[w];
        if (i < k) {
            starts[i] = pre + inc - v;
            cursor[i] = pre + inc - v;
        }
        __syncthreads();
        if (threadIdx.x == 255) s_carry = pre + inc;
        __syncthreads();
    }
}

// histogram of the labels; the LAST workgroup to arrive (ticket, self-resetting) scans it -- formerly a one-workgroup
// launch of its own, 4.6 of the ~120 us of a SOG chunk iteration
__global__ __launch_bounds__(256) void kmeans_label_hist_kernel(const int32_t *__restrict__ labels, int64_t n, int k,
                                                                unsigned *__restrict__ counts, unsigned *__restrict__ ticket,
                                                                unsigned *__restrict__ starts, unsigned *__restrict__ cursor, KmBatch kb)
{
    extern __shared__ unsigned s_h[];
    {
        labels += km_problem_rows(kb, n);                      // problem blockIdx.y
        counts += (size_t)blockIdx.y * k;
        starts += (size_t)blockIdx.y * k;
        cursor += (size_t)blockIdx.y * k;
        ticket += (size_t)blockIdx.y * KM_META_WORDS;
    }
    const bool use_lds = k <= 8192;
    if (use_lds) {
        for (int i = threadIdx.x; i < k; i += 256) s_h[i] = 0;
        __syncthreads();
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int l = labels[i];
        if (l >= 0) atomicAdd(use_lds ? &s_h[l] : &counts[l], 1u);
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < k; i += 256)
            if (s_h[i]) atomicAdd(&counts[i], s_h[i]);
    }
    __shared__ unsigned s_last;
    __builtin_amdgcn_s_waitcnt(0);   // this workgroup's device-scope atomics have completed before the ticket
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) *ticket = 0;
    kmeans_label_scan_body(counts, k, starts, cursor);
}

// point indices grouped by label.  Per 2048-point tile: LDS ranks (returning atomics), ONE global atomic per (tile, label)
// reserves the run -- a global atomic per point serialises on the K cursors (35 us per SOG chunk)
__global__ __launch_bounds__(256) void kmeans_label_scatter_kernel(const int32_t *__restrict__ labels, int64_t n, int k,
                                                                   unsigned *__restrict__ cursor, unsigned *__restrict__ perm,
                                                                   double *__restrict__ zero_sums, int64_t n_sums, KmBatch kb)
{
    {
        const int64_t r0 = km_problem_rows(kb, n);             // problem blockIdx.y
        labels += r0;
        perm += r0;
        cursor += (size_t)blockIdx.y * k;
        if (zero_sums) zero_sums += (size_t)blockIdx.y * n_sums;
    }
    if (zero_sums)     // the segmented sums that follow accumulate into these (round 4: finalize no longer re-zeroes them)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_sums; i += (int64_t)gridDim.x * 256) zero_sums[i] = 0.0;
    extern __shared__ unsigned s_h[];   // [k] counts, then [k] run bases (k <= 8192)
    unsigned *s_b = s_h + k;
    const bool use_lds = k <= 8192;
    const int64_t ntiles = (n + 2047) / 2048;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int lab[8];
        unsigned rk[8];
        if (use_lds) {
            for (int i = threadIdx.x; i < k; i += 256) s_h[i] = 0;
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = t * 2048 + u * 256 + threadIdx.x;
            lab[u] = i < n ? labels[i] : -1;
            if (lab[u] >= 0) rk[u] = use_lds ? atomicAdd(&s_h[lab[u]], 1u) : atomicAdd(&cursor[lab[u]], 1u);
        }
        if (use_lds) {
            __syncthreads();
            for (int i = threadIdx.x; i < k; i += 256)
                if (s_h[i]) s_b[i] = atomicAdd(&cursor[i], s_h[i]);
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (lab[u] >= 0) perm[(use_lds ? s_b[lab[u]] : 0u) + rk[u]] = (unsigned)(t * 2048 + u * 256 + threadIdx.x);
        __syncthreads();
    }
}

// one workgroup per centroid: four waves take every fourth row (lanes = dimensions, rows read coalesced, eight in flight),
// float64 partial sums meet in LDS; also re-zeroes counts for the next iteration
__global__ __launch_bounds__(256) void kmeans_centroid_reduce_kernel(const float *__restrict__ data, int D,
                                                                     const unsigned *__restrict__ perm,
                                                                     const unsigned *__restrict__ starts,
                                                                     unsigned *__restrict__ counts, int k, float *__restrict__ cent)
{
    __shared__ double s_part[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x;
    const unsigned cnt = counts[c], s0 = starts[c];
    const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.0f;   // gpu_ops.py:93
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int d = d0 + lane;
        double acc = 0.0;   // float64 makes the order immaterial (the reference's f32 atomics are order-dependent)
        // wave g owns the rows g, g + 4, g + 8, ...: 64 of their indices are fetched at once (one per lane) and
        // broadcast, so the row loads do not wait on the permutation and 16 of them are in flight
        for (unsigned jb = g; jb < cnt; jb += 256) {
            const unsigned jmine = jb + 4u * (unsigned)lane;
            const unsigned my_row = jmine < cnt ? perm[s0 + jmine] : 0u;
            const int nrows = (int)min(64u, (cnt - jb + 3u) / 4u);
            for (int i0 = 0; i0 < nrows; i0 += 16) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const unsigned row = (unsigned)__shfl((int)my_row, (i0 + u) & 63);
                    v[u] = (i0 + u < nrows && d < D) ? data[(int64_t)row * D + d] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += (double)v[u];
            }
        }
        s_part[g][lane] = acc;
        __syncthreads();
        if (g == 0 && d < D) {
            const double sum = (s_part[0][lane] + s_part[1][lane]) + (s_part[2][lane] + s_part[3][lane]);
            cent[(int64_t)c * D + d] = cnt > 0 ? (float)sum * inv : 0.0f;   // empty cluster -> 0 (gpu_ops.py:78-96)
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[c] = 0u;
}

// Segmented-sum update (default): every wave takes 64 CONSECUTIVE entries of the sorted-by-label permutation -- rows of
// one or two clusters -- reads the rows coalesced (lanes = dimensions, 16 in flight), and adds its float64 partial sums
// to the cluster's accumulators when the label changes: perfectly balanced whatever the cluster sizes are (one workgroup
// per centroid, above, runs as long as its largest cluster and reads through one CU), ~2 x 45 float64 atomics per wave.
// kmeans_finalize_reset_kernel divides and re-zeroes.
template <int ROWS>   // consecutive entries of the permutation per wave: 64, 32 or 16 (fewer rows = more waves to hide the gathers behind)
__global__ __launch_bounds__(256) void kmeans_segment_sum_kernel(const float *__restrict__ data, int D,
                                                                 const unsigned *__restrict__ perm,
                                                                 const unsigned *__restrict__ starts,
                                                                 const unsigned *__restrict__ counts, int k,
                                                                 const int32_t *__restrict__ labels, double *__restrict__ sums, KmBatch kb)
{
    {
        int64_t n_;
        const int64_t r0 = km_problem_rows(kb, n_);            // problem blockIdx.y
        data += r0 * D;
        labels += r0;
        perm += r0;
        starts += (size_t)blockIdx.y * k;
        counts += (size_t)blockIdx.y * k;
        sums += (size_t)blockIdx.y * k * D;
    }
    const int64_t n = (int64_t)starts[k - 1] + counts[k - 1];   // rows with a label (unassignable rows are not in the permutation)
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w * ROWS < n; w += nw) {
        const int64_t j0 = w * ROWS;
        const int m = (int)((n - j0) < ROWS ? (n - j0) : ROWS);
        const unsigned my_row = lane < m ? perm[j0 + lane] : 0u;
        const int my_lab = lane < m ? labels[my_row] : -1;
        int cur = __builtin_amdgcn_readfirstlane(my_lab);
        double acc = 0.0;
        constexpr int HB = ROWS < 16 ? ROWS : 16;   // row loads in flight per lane (32: no faster)
        for (int i0 = 0; i0 < m; i0 += HB) {
            float v[HB];
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const unsigned row = (unsigned)__shfl((int)my_row, (i0 + u) & 63);
                v[u] = (i0 + u < m && lane < D) ? data[(int64_t)row * D + lane] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                if (i0 + u < m) {   // wave-uniform
                    const int lab = __shfl(my_lab, (i0 + u) & 63);
                    if (lab != cur) {
                        if (lane < D && cur >= 0) unsafeAtomicAdd(&sums[(int64_t)cur * D + lane], acc);
                        acc = 0.0;
                        cur = lab;
                    }
                    acc += (double)v[u];
                }
            }
        }
        if (lane < D && cur >= 0) unsafeAtomicAdd(&sums[(int64_t)cur * D + lane], acc);
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
        const int idx = sog_codebook_index(lcb, kcb, vals[i]);   // sog_math.h
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

// One Lloyd iteration (assign + update) of `nprob` problems of the same k and D in ONE set of launches (blockIdx.y = problem).
// n: rows of the largest problem (sizes the grids), n_total: rows of all of them (off_dev[nprob]); off_dev == nullptr: one problem.
template <int D>
static int launch_assign_mfma_t(gsx_ctx *c, const float *data, int64_t n, float *cent, int k, int32_t *labels, double *sums,
                                unsigned *counts, int it, int nprob = 1, const int64_t *off_dev = nullptr, int64_t n_total = -1)
{
    constexpr int NS = km_dp(D) / 16;
    if (n_total < 0) n_total = n;
    const unsigned P = (unsigned)nprob;
    const KmBatch kb{off_dev, n};
    const int ktiles = (k + 31) / 32;
    const size_t opnd_bytes = sizeof(ku32x4) * (size_t)ktiles * NS * 2 * 64 * P;
    // operand words | meta (per problem 2 x 16 words, alternating by iteration) | uncertain list / permutation (n_total words)
    // | starts (k per problem) | cursor (k per problem)
    GSX_CHECK(c->scratch5.reserve(opnd_bytes + sizeof(unsigned) * ((size_t)KM_META_WORDS * P + (size_t)n_total + 2 * (size_t)k * P + 16)));
    ku32x4 *opnd = c->scratch5.as<ku32x4>();
    unsigned *meta2 = reinterpret_cast<unsigned *>(c->scratch5.as<char>() + opnd_bytes);
    unsigned *meta = meta2 + 16 * (it & 1);            // [0] = max |c|^2 (float bits), [1] = list length, [2] = histogram ticket
    unsigned *meta_next = meta2 + 16 * ((it + 1) & 1);
    unsigned *list = meta2 + (size_t)KM_META_WORDS * P, *starts = list + n_total, *cursor = starts + (size_t)k * P;
    const bool fused_update = D <= 64;                 // (always: D is 9, 24 or 45)
    const int per = std::max(1, c->num_cu / nprob);    // workgroups per problem where one launch used to fill the chip
    if (it == 0 || !fused_update) {
        // the first iteration's operands come from the caller's centroids; every later one's were written by the
        // finalize + operands kernel at the end of the iteration before
        GSX_HIP(hipMemsetAsync(meta2, 0, sizeof(unsigned) * KM_META_WORDS * P, c->stream));
        hipLaunchKernelGGL((kmeans_centroid_operands_kernel<D>), dim3(ktiles * NS, P), dim3(64), 0, c->stream, cent, k, opnd,
                           reinterpret_cast<float *>(meta));
    }
    if (c->kmeans_cs && ktiles <= KM_CS_WAVES * KM_CS_CT) {
        // centroid-stationary: one 16-wave workgroup per CU keeps every centroid operand in registers (kmeans_cs.h); in a
        // batch a workgroup stays with ONE problem's centroids for all of its share of that problem's rows.  Smaller K
        // (the palette's 256 and 64 per chunk at --compression_level 4-9): fewer waves per workgroup, more workgroups.
        const float *metaf = reinterpret_cast<const float *>(meta);
        if (ktiles <= 2 && c->kmeans_cs_small) {
            const int blocks = (int)std::min<int64_t>(div_up(n, 64), (int64_t)per * (c->km_small_wgs > 0 ? c->km_small_wgs : 3));
            hipLaunchKernelGGL((kmeans_assign_mfma_cs_kernel<D, 2, 1, 2>), dim3(blocks, P), dim3(128), 0, c->stream, data, n, opnd, ktiles, metaf,
                               labels, list, meta + 1, kb);
        } else if (ktiles <= 8 && c->kmeans_cs_small) {
            const int blocks = (int)std::min<int64_t>(div_up(n, 64), (int64_t)per * (c->km_small_wgs > 0 ? c->km_small_wgs : 3));   // (141 VGPRs: three waves per SIMD)
            hipLaunchKernelGGL((kmeans_assign_mfma_cs_kernel<D, 4, 2, 2>), dim3(blocks, P), dim3(256), 0, c->stream, data, n, opnd, ktiles, metaf,
                               labels, list, meta + 1, kb);
        } else {
            const int blocks = (int)std::min<int64_t>(div_up(n, KM_CS_BLOCK), (int64_t)per);
            hipLaunchKernelGGL((kmeans_assign_mfma_cs_kernel<D>), dim3(blocks, P), dim3(64 * KM_CS_WAVES), 0, c->stream, data, n, opnd,
                               ktiles, metaf, labels, list, meta + 1, kb);
        }
    } else {
        const int64_t tiles = div_up(n, KM_MF_TILE);
        const int blocks = (int)std::min<int64_t>(tiles, (int64_t)per * 8);
        hipLaunchKernelGGL((kmeans_assign_mfma_kernel<D>), dim3(blocks, P), dim3(64 * KM_MF_WAVES), 0, c->stream, data, n, opnd, ktiles,
                           reinterpret_cast<const float *>(meta), labels, list, meta + 1, kb);
    }
    // (round 5: a workgroup takes 64 list entries at a time; the list holds ~0.3 % of the rows)
    const int eb = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 64 * 64), (int64_t)per * c->km_exact_blocks));
    hipLaunchKernelGGL((kmeans_assign_exact_list_kernel<D>), dim3(eb, P), dim3(64 * KM_EX_WAVES), 0, c->stream, data, cent, k, list,
                       meta + 1, labels, fused_update && it > 0 ? counts : nullptr, kb);
    GSX_HIP(hipGetLastError());
    GSX_CHECK(timing_end(c, GSX_T_KMEANS_ASSIGN));
    GSX_CHECK(timing_begin(c, GSX_T_KMEANS_UPDATE));
    // update (the list is dead now: its storage becomes the permutation)
    const int hb = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 2048), (int64_t)per * 4));
    hipLaunchKernelGGL(kmeans_label_hist_kernel, dim3(hb, P), dim3(256), k <= 8192 ? sizeof(unsigned) * (size_t)k : 0, c->stream, labels, n,
                       k, counts, meta + 2, starts, cursor, kb);   // (+ the scan of the counts, in each problem's last workgroup)
    hipLaunchKernelGGL(kmeans_label_scatter_kernel, dim3(hb, P), dim3(256), k <= 8192 ? 2 * sizeof(unsigned) * (size_t)k : 0, c->stream,
                       labels, n, k, cursor, list, fused_update && it > 0 ? sums : nullptr, (int64_t)k * D, kb);
    if (fused_update) {
        const int rows = c->km_seg_rows;
        const int sb = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 4 * rows), (int64_t)per * 8));
        if (rows == 64)
            hipLaunchKernelGGL(kmeans_segment_sum_kernel<64>, dim3(sb, P), dim3(256), 0, c->stream, data, D, list, starts, counts, k, labels, sums, kb);
        else if (rows == 32)
            hipLaunchKernelGGL(kmeans_segment_sum_kernel<32>, dim3(sb, P), dim3(256), 0, c->stream, data, D, list, starts, counts, k, labels, sums, kb);
        else
            hipLaunchKernelGGL(kmeans_segment_sum_kernel<16>, dim3(sb, P), dim3(256), 0, c->stream, data, D, list, starts, counts, k, labels, sums, kb);
        // finalize + the NEXT iteration's operands + the meta block of the one after, in one launch
        hipLaunchKernelGGL((kmeans_finalize_operands_kernel<D>), dim3(ktiles * NS, P), dim3(64), 0, c->stream, sums, counts, k, cent, opnd,
                           meta_next, meta);
    } else {
        hipLaunchKernelGGL(kmeans_centroid_reduce_kernel, dim3(k), dim3(256), 0, c->stream, data, D, list, starts, counts, k, cent);
    }
    GSX_HIP(hipGetLastError());
    GSX_CHECK(timing_end(c, GSX_T_KMEANS_UPDATE));
    return 0;
}

// returns 0 and sets *fused when the templated kernel (assign + accumulate in one launch) ran
static int launch_assign(gsx_ctx *c, const float *data, int64_t n, int d, float *cent, int k, int32_t *labels,
                         double *sums, unsigned *counts, bool *fused, bool *updated, int it)
{
    *fused = true;
    *updated = false;
    if (c->kmeans_mfma && k >= 64 && (d == 9 || d == 24 || d == 45)) {
        // matrix-core filter + exact certificate (identical labels), update by segmented reduction: the whole iteration
        *updated = true;
        if (d == 9) return launch_assign_mfma_t<9>(c, data, n, cent, k, labels, sums, counts, it);
        if (d == 24) return launch_assign_mfma_t<24>(c, data, n, cent, k, labels, sums, counts, it);
        return launch_assign_mfma_t<45>(c, data, n, cent, k, labels, sums, counts, it);
    }
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
        bool fused = false, updated = false;
        GSX_CHECK(timing_begin(c, GSX_T_KMEANS_ASSIGN));
        GSX_CHECK(launch_assign(c, data_dev, n, d, cent_dev, k, labels_dev, sums, counts, &fused, &updated, it));
        if (updated) continue;   // the matrix-core path ran assign AND update (sort-by-label reduction) and closed both timing slots
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

// nprob independent problems of the same d and k, rows concatenated: problem p = rows [off[p], off[p+1]) of data_dev / labels_dev,
// centroids p * k * d ... of cent_dev (in: init, out: result).  The matrix-core path runs every iteration of ALL problems in
// one set of launches (launch_assign_mfma_t); other shapes run the problems one after the other -- the same kernels, launch
// order and arithmetic per problem as kmeans_lloyd_dev either way.
int kmeans_lloyd_batch_dev(gsx_ctx *c, const float *data_dev, const int64_t *off_host, int nprob, int d, int k, int max_iter,
                           float *cent_dev, int32_t *labels_dev)
{
    if (nprob <= 0 || d <= 0 || k <= 0 || !off_host) GSX_FAIL("kmeans batch: bad shape nprob=%d d=%d k=%d", nprob, d, k);
    int64_t n_max = 0;
    for (int p = 0; p < nprob; ++p) {
        const int64_t rows = off_host[p + 1] - off_host[p];
        if (rows <= 0 || off_host[0] != 0) GSX_FAIL("kmeans batch: problem %d has %lld rows (offsets must start at 0 and increase)", p, (long long)rows);
        n_max = std::max(n_max, rows);
    }
    const int64_t n_total = off_host[nprob];
    const bool batched = c->kmeans_mfma && k >= 64 && (d == 9 || d == 24 || d == 45) && nprob > 1 && nprob <= 65535;
    if (!batched) {
        for (int p = 0; p < nprob; ++p)
            GSX_CHECK(kmeans_lloyd_dev(c, data_dev + off_host[p] * d, off_host[p + 1] - off_host[p], d, k, max_iter,
                                       cent_dev + (size_t)p * k * d, labels_dev + off_host[p]));
        return 0;
    }
    if (c->km_group_mb > 0) {
        // groups of consecutive problems whose rows fit km_group_mb: every iteration of a group runs before the next group
        // starts, so its rows are read from the Infinity Cache after the first pass
        const int64_t cap_rows = std::max<int64_t>(1, (int64_t)c->km_group_mb * (1 << 20) / (4 * (int64_t)d));
        int first = 0;
        while (first < nprob) {
            int last = first + 1;
            while (last < nprob && off_host[last + 1] - off_host[first] <= cap_rows) ++last;
            if (last - first < nprob) {   // (a single group = the plain batch below)
                std::vector<int64_t> sub((size_t)(last - first + 1));
                for (int p = first; p <= last; ++p) sub[(size_t)(p - first)] = off_host[p] - off_host[first];
                const int keep = c->km_group_mb;
                c->km_group_mb = 0;
                const int rc = kmeans_lloyd_batch_dev(c, data_dev + off_host[first] * d, sub.data(), last - first, d, k, max_iter,
                                                      cent_dev + (size_t)first * k * d, labels_dev + off_host[first]);
                c->km_group_mb = keep;
                if (rc != 0) return rc;
                first = last;
                continue;
            }
            break;
        }
        if (first >= nprob) return 0;
    }
    const size_t kd = (size_t)k * d * nprob, kk = (size_t)k * nprob;
    const size_t off_at = (sizeof(double) * kd + sizeof(unsigned) * kk + 63) & ~(size_t)63;
    GSX_CHECK(c->scratch3.reserve(off_at + sizeof(int64_t) * (size_t)(nprob + 1) + 64));
    double *sums = c->scratch3.as<double>();
    unsigned *counts = reinterpret_cast<unsigned *>(c->scratch3.as<char>() + sizeof(double) * kd);
    int64_t *off_dev = reinterpret_cast<int64_t *>(c->scratch3.as<char>() + off_at);
    GSX_HIP(hipMemcpyAsync(off_dev, off_host, sizeof(int64_t) * (size_t)(nprob + 1), hipMemcpyHostToDevice, c->stream));
    GSX_HIP(hipMemsetAsync(sums, 0, sizeof(double) * kd + sizeof(unsigned) * kk, c->stream));
    for (int it = 0; it < max_iter; ++it) {
        GSX_CHECK(timing_begin(c, GSX_T_KMEANS_ASSIGN));
        if (d == 9) GSX_CHECK(launch_assign_mfma_t<9>(c, data_dev, n_max, cent_dev, k, labels_dev, sums, counts, it, nprob, off_dev, n_total));
        else if (d == 24) GSX_CHECK(launch_assign_mfma_t<24>(c, data_dev, n_max, cent_dev, k, labels_dev, sums, counts, it, nprob, off_dev, n_total));
        else GSX_CHECK(launch_assign_mfma_t<45>(c, data_dev, n_max, cent_dev, k, labels_dev, sums, counts, it, nprob, off_dev, n_total));
    }
    GSX_HIP(hipStreamSynchronize(c->stream));   // off_host (the caller's, possibly pageable) must outlive its async copy
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
