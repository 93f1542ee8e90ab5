// knn_common.h -- device arithmetic shared by the three exact-KNN kernels
// (brute force, grid brick, expanding-ring fallback).
//
// The quantity every kernel must reproduce BIT FOR BIT is what the reference's CPU
// path computes (data_processor.py:160-173 through scipy cKDTree + numpy):
//     s      = ((0 + dx*dx) + dy*dy) + dz*dz        float64, inputs widened from f32,
//                                                   no FMA (ckdtree distance.h, m = 3)
//     d[0..k] = sqrt of the k+1 smallest s, ascending (d[0] is the query itself)
//     mean   = (float)( pairwise8(d[1..k]) / k )    numpy pairwise order, f64 divide
// Candidate SELECTION may use f32 (with a conservative margin); every value that
// reaches the output is recomputed in f64 in exactly that order.
#pragma once

#include <hip/hip_runtime.h>

namespace gsx {

// Relative slack on f32 squared distances.  fl32 error of ((dx*dx)+dy*dy)+dz*dz with
// fmaf is <= ~5 * 2^-24 = 3e-7 relative (all terms positive); a candidate whose exact
// s is <= T has f32 d2 <= T*(1+3e-7), so testing d2 <= up32(T)*(1+2e-6) never drops it.
__device__ constexpr float F32_SLACK = 1.0f + 2.0e-6f;
__device__ constexpr float F32_TINY = 1.0e-37f;  // absolute slack for denormal-range d2

__device__ __forceinline__ float dist2_f32(float qx, float qy, float qz, float px, float py, float pz)
{
    float dx = qx - px, dy = qy - py, dz = qz - pz;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// exact cKDTree arithmetic; __dmul_rn/__dadd_rn never contract into FMA
__device__ __forceinline__ double dist2_f64(double qx, double qy, double qz, float px, float py, float pz)
{
    double dx = __dsub_rn(qx, (double)px);
    double dy = __dsub_rn(qy, (double)py);
    double dz = __dsub_rn(qz, (double)pz);
    double s = __dmul_rn(dx, dx);
    s = __dadd_rn(s, __dmul_rn(dy, dy));
    s = __dadd_rn(s, __dmul_rn(dz, dz));
    return s;
}

// f32 filter bound for an exact f64 threshold T (T >= 0, may be +inf)
__device__ __forceinline__ float bound_from(double T)
{
    return __double2float_ru(T) * F32_SLACK + F32_TINY;
}

// Ascending list of the KCAP smallest exact squared distances seen so far (+inf padded), all in
// registers and only ever indexed with compile-time constants.  KCAP >= kk = k+1; entries
// beyond kk are simply further neighbours that are never read.
template <int KCAP>
struct TopList {
    double a[KCAP];

    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int i = 0; i < KCAP; ++i) a[i] = __builtin_inf();
    }

    // the kk-th smallest (kk is wave-uniform: the selects are driven by scalar compares)
    __device__ __forceinline__ double kth(int kk) const
    {
        double r = a[KCAP - 1];
#pragma unroll
        for (int i = 0; i < KCAP - 1; ++i) {
            r = (kk - 1 == i) ? a[i] : r;
            // opaque to the optimiser: without it LLVM folds the select chain into a[kk-1], a
            // DYNAMIC index that drags the whole list out of VGPRs into scratch memory
            asm("" : "+v"(r));
        }
        return r;
    }

    // branch-free sorted insert that drops the largest: 2 f64 VALU ops per slot.
    // Raw v_min_f64 / v_max_f64: fmin()/fmax() make LLVM add a canonicalising
    // v_max_f64 x,x,x per operand (sNaN quieting) -- +50 % on this chain; no value
    // here is ever NaN.
    __device__ __forceinline__ void insert(double s)
    {
#pragma unroll
        for (int i = 0; i < KCAP; ++i) {
            double lo, hi;
            asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a[i]), "v"(s));
            asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a[i]), "v"(s));
            a[i] = lo;
            s = hi;
        }
    }
};

// ---- selection by sorting network (knn_brick phase 2) ----------------------------------------------
// The bubble insert above costs 2*KCAP float64 ops per candidate, and a wave runs it max-over-lanes
// times (~44 for ~30 candidates per lane at k = 16): 1500 of the ~4600 VALU instructions of a batch.
// TopNet keeps the L smallest squared distances of the query's NEIGHBOURS -- the query itself is
// excluded by its index, so L = k for the power-of-two k the headline configs use -- and takes
// candidates in blocks of BS = min(L, 8): a block is sorted by Batcher's odd-even merge sort
// (19 compare-exchanges for 8), merged into the list with the bitonic rule
//     c[L-BS+i] = min(a[L-BS+i], b[BS-1-i])   ->   c holds the L smallest and is bitonic
// and log2(L) half-cleaner stages sort c again: (19 + 32) CE + 8 min = 110 ops per 8 candidates
// at L = 16, i.e. 14 ops per candidate instead of the bubble insert's 34.  (The merge rule with BS < L was
// checked with the 0-1 principle for (L, BS) in {16,32,64} x {4,8,16}.)  All indices are compile-time
// constants: the list and the block stay in VGPRs.
__device__ __forceinline__ void ce_f64(double &lo, double &hi)  // compare-exchange, no NaNs ever
{
    double a, b;
    asm("v_min_f64 %0, %1, %2" : "=v"(a) : "v"(lo), "v"(hi));
    asm("v_max_f64 %0, %1, %2" : "=v"(b) : "v"(lo), "v"(hi));
    lo = a;
    hi = b;
}

// Batcher's odd-even merge sort, ascending, N a power of two; fully unrolled (lo/r/N are constants)
template <int N, int LO, int R, int SPAN>
__device__ __forceinline__ void oe_merge(double *v)
{
    constexpr int STEP = R * 2;
    if constexpr (STEP < SPAN) {
        oe_merge<N, LO, STEP, SPAN>(v);
        oe_merge<N, LO + R, STEP, SPAN>(v);
#pragma unroll
        for (int i = LO + R; i < LO + SPAN - R; i += STEP) ce_f64(v[i], v[i + R]);
    } else {
        ce_f64(v[LO], v[LO + R]);
    }
}
template <int N, int LO, int SPAN>
__device__ __forceinline__ void oe_sort(double *v)
{
    if constexpr (SPAN > 1) {
        oe_sort<N, LO, SPAN / 2>(v);
        oe_sort<N, LO + SPAN / 2, SPAN / 2>(v);
        oe_merge<N, LO, 1, SPAN>(v);
    }
}

#ifndef GSX_CAP4   // list capacities 12, 20, 28, 36, 44, 52 in knn_brick and knn_leaf besides the multiples of 8 (round 5; 0 for A/B runs)
#define GSX_CAP4 1
#endif
template <int L>
struct TopNet {
    static_assert(L % 4 == 0 && L >= 8 && L <= 64, "list length: a multiple of 4");
    // Lengths that are not a power of two (24, 40, 48, 56: round 4 -- the reference's CLI asks for k = 18 ... 50, and a 64-entry list
    // for k = 36 is 2.2x the time of a 32-entry one) merge through the next power of two P with P - L entries of -inf
    // imagined IN FRONT of the list: [-inf ..., sorted head, bitonic tail] is still bitonic, and a compare-exchange whose
    // lower partner is -inf does nothing, so it is not emitted: 52 CE at L = 24 (80 at 32), 128 at L = 48 (192 at 64).
    // Checked with the 0-1 principle for every multiple of 4 up to 64 (tests/test_ring_fast_logic.py::test_padded_bitonic_merge);
    // round 5 instantiates 12, 20, 28 (64 CE: the reference's default k = 25 and --sor_intensity 5, k = 27), 36, 44 and 52 as well.
    static constexpr int P = L <= 8 ? 8 : (L <= 16 ? 16 : (L <= 32 ? 32 : 64)), OFF = P - L;
    // candidates per block.  8, not 16 (round 3): the same ~13 network ops per candidate, half the padding in a lane's
    // last block, 16 fewer live VGPRs -> two more waves per SIMD; 4 costs more network ops than it saves
#ifndef GSX_NET_BS
#define GSX_NET_BS 8
#endif
    static constexpr int BS = L < GSX_NET_BS ? L : GSX_NET_BS;
    double a[L];  // ascending, +inf padded

    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int i = 0; i < L; ++i) a[i] = __builtin_inf();
    }
    // j-th smallest (1-based, wave-uniform j <= L); same select chain as TopList::kth
    __device__ __forceinline__ double kth(int j) const
    {
        if (j == L) return a[L - 1];  // wave-uniform: the power-of-two k of the headline configs skips the select chain
        double r = a[L - 1];
#pragma unroll
        for (int i = 0; i < L - 1; ++i) {
            r = (j - 1 == i) ? a[i] : r;
            asm("" : "+v"(r));
        }
        return r;
    }
    // the list is still all +inf: the sorted block IS the list
    __device__ __forceinline__ void assign_block(double (&b)[BS])
    {
        oe_sort<BS, 0, BS>(b);
#pragma unroll
        for (int i = 0; i < BS; ++i) a[i] = b[i];
    }
    __device__ __forceinline__ void merge_block(double (&b)[BS])
    {
        oe_sort<BS, 0, BS>(b);
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            double c;
            asm("v_min_f64 %0, %1, %2" : "=v"(c) : "v"(a[L - BS + i]), "v"(b[BS - 1 - i]));
            a[L - BS + i] = c;
        }
#pragma unroll
        for (int half = P / 2; half >= 1; half /= 2)
#pragma unroll
            for (int p = OFF; p < P; ++p)   // (positions below OFF hold -inf)
                if ((p & half) == 0) ce_f64(a[p - OFF], a[p + half - OFF]);
    }
};

// Correctly rounded float64 square root for the squared distances of knn_brick's epilogue: the compiler's own
// lowering of sqrt(double) (v_rsq_f64 seed, two Goldschmidt steps, two residual corrections) WITHOUT its range
// scaling (inputs below 2^-767 are scaled by 2^256: squared differences of float32 coordinates are 0 or >= 2^-298)
// and without the class test for 0 / inf: the seed is taken of max(x, 1e-300), so x = 0 (duplicate points) runs
// through as g = 0 * y = 0 and every correction term stays 0.  Same instruction sequence otherwise, hence the same
// bits (the GPU suite compares 10M mean distances with cKDTree's).  11 instead of ~18 instructions per root.
__device__ __forceinline__ double sqrt_rn_dist2(double x)
{
    double xs;
    asm("v_max_f64 %0, %1, %2" : "=v"(xs) : "v"(x), "v"(1e-300));
    const double y = __builtin_amdgcn_rsq(xs);
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

// epilogue for TopNet: entries 0..k-1 are the neighbours (the query was never inserted)
template <int L>
__device__ __forceinline__ float mean_from_net(const TopNet<L> &lst, int k)
{
    double b[L];
#pragma unroll
    for (int j = 0; j < L; ++j) b[j] = sqrt_rn_dist2(lst.a[j]);
    double res;
    if (L % 8 == 0 && k == L) {  // wave-uniform; the headline k = 16 / 32: numpy's 8 accumulators over a MULTIPLE of 8, no scalar
                                 // branches (L = 12, 20, 28 ...: numpy adds the last 4 sequentially -> the general branch)
        double r[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) r[t] = b[t];
#pragma unroll
        for (int j = 8; j < L; ++j) r[j & 7] = __dadd_rn(r[j & 7], b[j]);
        res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                        __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
    } else if (k < 8) {
        res = 0.0;
#pragma unroll
        for (int j = 0; j < 7 && j < L; ++j)
            if (j < k) res = __dadd_rn(res, b[j]);
    } else {
        double r[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) r[t] = b[t];
        const int nfull = k - (k % 8);
#pragma unroll
        for (int j = 8; j < L; ++j)
            if (j < nfull) r[j & 7] = __dadd_rn(r[j & 7], b[j]);
        res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                        __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
#pragma unroll
        for (int j = 8; j < L; ++j)
            if (j >= nfull && j < k) res = __dadd_rn(res, b[j]);
    }
    return __double2float_rn(__ddiv_rn(res, (double)k));
}

// numpy pairwise sum of n <= 128 doubles read through a functor (loops_utils.h.src);
// used where the values sit in LDS (dynamic indexing is free there)
template <class F>
__device__ __forceinline__ double pairwise_sum_le128(F at, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res = __dadd_rn(res, at(i));
        return res;
    }
    double r0 = at(0), r1 = at(1), r2 = at(2), r3 = at(3), r4 = at(4), r5 = at(5), r6 = at(6), r7 = at(7);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 = __dadd_rn(r0, at(i + 0));
        r1 = __dadd_rn(r1, at(i + 1));
        r2 = __dadd_rn(r2, at(i + 2));
        r3 = __dadd_rn(r3, at(i + 3));
        r4 = __dadd_rn(r4, at(i + 4));
        r5 = __dadd_rn(r5, at(i + 5));
        r6 = __dadd_rn(r6, at(i + 6));
        r7 = __dadd_rn(r7, at(i + 7));
    }
    double res = __dadd_rn(__dadd_rn(__dadd_rn(r0, r1), __dadd_rn(r2, r3)),
                           __dadd_rn(__dadd_rn(r4, r5), __dadd_rn(r6, r7)));
    for (; i < n; ++i) res = __dadd_rn(res, at(i));
    return res;
}

// Epilogue: list -> (float) mean of sqrt(entries 1..k) in numpy's pairwise order (entry 0 is the
// query itself).  Everything is unrolled with compile-time register indices; k is wave-uniform,
// so the `j < ...` predicates are scalar branches.  No scratch memory.
template <int KCAP>
__device__ __forceinline__ float mean_from_list(const TopList<KCAP> &lst, int k)
{
    double b[KCAP - 1];
#pragma unroll
    for (int j = 0; j < KCAP - 1; ++j) b[j] = __dsqrt_rn(lst.a[1 + j]);
    double res;
    if (k < 8) {
        res = 0.0;
#pragma unroll
        for (int j = 0; j < 7 && j < KCAP - 1; ++j)
            if (j < k) res = __dadd_rn(res, b[j]);
    } else {
        double r[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) r[t] = b[t < KCAP - 1 ? t : 0];
        const int nfull = k - (k % 8);
#pragma unroll
        for (int j = 8; j < KCAP - 1; ++j)
            if (j < nfull) r[j & 7] = __dadd_rn(r[j & 7], b[j]);
        res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                        __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
#pragma unroll
        for (int j = 8; j < KCAP - 1; ++j)
            if (j >= nfull && j < k) res = __dadd_rn(res, b[j]);
    }
    return __double2float_rn(__ddiv_rn(res, (double)k));
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ---- whole-wave reductions without address registers (round 5) --------------------------------------------------------
// __shfl_xor(v, off) is ds_bpermute with a per-lane ADDRESS ((lane ^ off) << 2).  The compiler computes the six address
// registers of a butterfly once and keeps them for every later butterfly of the kernel -- in knn_leaf that meant six
// dwords per lane spilled to scratch per leaf.  ds_swizzle (bit-mask mode: lane ^ OFF inside each half-wave, the pattern is an
// immediate) and v_permlane32_swap (the two half-waves) need no address at all.  Every lane gets the result.
template <int OFF>
__device__ __forceinline__ int swz_xor(int v)   // v of lane (l ^ OFF), OFF < 32
{
    return __builtin_amdgcn_ds_swizzle(v, (OFF << 10) | 0x1f);
}
template <class Op>
__device__ __forceinline__ int wave_reduce_bits(int v, Op op)   // op: (int, int) -> int on the value's bit pattern
{
    v = op(v, swz_xor<1>(v));
    v = op(v, swz_xor<2>(v));
    v = op(v, swz_xor<4>(v));
    v = op(v, swz_xor<8>(v));
    v = op(v, swz_xor<16>(v));
    // (x, y) = (v, v) -> x' = [v.lo | v.lo], y' = [v.hi | v.hi]: lane l holds v[l mod 32] and v[l mod 32 + 32]
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return op((int)r[0], (int)r[1]);
}
__device__ __forceinline__ int wave_sum_i32(int v) { return wave_reduce_bits(v, [](int a, int b) { return a + b; }); }
__device__ __forceinline__ int wave_min_i32(int v) { return wave_reduce_bits(v, [](int a, int b) { return a < b ? a : b; }); }
__device__ __forceinline__ int wave_max_i32(int v) { return wave_reduce_bits(v, [](int a, int b) { return a > b ? a : b; }); }
__device__ __forceinline__ float wave_min_f32(float v)
{
    return __int_as_float(wave_reduce_bits(__float_as_int(v), [](int a, int b) { return __float_as_int(fminf(__int_as_float(a), __int_as_float(b))); }));
}
__device__ __forceinline__ float wave_max_f32(float v)
{
    return __int_as_float(wave_reduce_bits(__float_as_int(v), [](int a, int b) { return __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b))); }));
}

// wave-uniform value known to the compiler as scalar
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Values the optimiser must not hoist out of a loop or a branch: whatever is computed from the result stays where the
// call is (knn_brick: loop-invariant setup arithmetic hoisted to the kernel entry and parked in scratch cost 0.8 GB of
// spill traffic per 10M-splat launch; recomputing it is a handful of instructions)
__device__ __forceinline__ int pinned_here(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ float pinned_here(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int pinned_here_s(int v) { asm volatile("" : "+s"(v)); return v; }

}  // namespace gsx
