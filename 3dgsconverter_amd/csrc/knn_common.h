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

// wave-uniform value known to the compiler as scalar
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

}  // namespace gsx
