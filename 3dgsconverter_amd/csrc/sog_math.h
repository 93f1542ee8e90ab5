// sog_math.h -- per-element arithmetic of the SOG writer's numeric core (formats/sog.py:264-459), shared by the
// column kernels of sog.hip (host-array entry points, one stage per call) and the table kernels of sog_table.hip (the
// whole writer on a device-resident splat table).  One implementation, so both paths emit the same bytes.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace gsx {

// float32 -> uint32 whose unsigned order is numpy's sort order: -0.0 == +0.0, every NaN last
__device__ __forceinline__ unsigned sort_key(float v)
{
    if (v != v) return 0xffffffffu;
    if (v == 0.0f) v = 0.0f;  // -0.0 -> +0.0
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
// inverse (a NaN key comes back as a NaN, -0.0 as +0.0)
__device__ __host__ __forceinline__ float sort_unkey(unsigned k)
{
    const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}

// sog.py:315-386 for one splat: normalise, positive hemisphere of the largest component, * sqrt(2), three bytes + 252 + argmax
__device__ __forceinline__ uchar4 sog_quat_pack(float4 q4)
{
    float q[4] = {q4.x, q4.y, q4.z, q4.w};
    // np.linalg.norm(q, axis=1): sqrt(add.reduce(q*q)) in float32, four elements summed left to right
    float s = __fmul_rn(q[0], q[0]);
    s = __fadd_rn(s, __fmul_rn(q[1], q[1]));
    s = __fadd_rn(s, __fmul_rn(q[2], q[2]));
    s = __fadd_rn(s, __fmul_rn(q[3], q[3]));
    const float nrm = __builtin_sqrtf(s);   // correctly rounded (HIP's __fsqrt_rn is the NATIVE v_sqrt_f32 here: 1 ulp off ~ once per 10^5 rows)
    int mi = 0;
    float ma = -1.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        q[c] = __fdiv_rn(q[c], nrm);                // qn = q / norm
        const float a = fabsf(q[c]);
        if (a > ma) {                               // np.abs(qn).argmax(axis=1): first maximum
            ma = a;
            mi = c;
        }
    }
    const float mv = q[mi];
    const float sg = mv > 0.0f ? 1.0f : (mv < 0.0f ? -1.0f : 0.0f);   // np.sign(max_val)
    unsigned char b[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = __fmul_rn(q[c], sg);                                  // qn *= sign_flip
        v = (float)((double)v * 1.4142135623730951);                    // qn *= np.sqrt(2.0): float64 scalar, cast back
        float t = __fadd_rn(__fmul_rn(v, 0.5f), 0.5f);                  // quantize_vec: (v*0.5 + 0.5) * 255.0, float32
        t = __fmul_rn(t, 255.0f);
        t = fminf(fmaxf(t, 0.0f), 255.0f);                              // np.clip
        b[c] = (unsigned char)t;                                        // astype(uint8): truncation
    }
    // the three components that are not the maximum, in index order
    const int i0 = mi == 0 ? 1 : 0, i1 = mi <= 1 ? 2 : 1, i2 = mi == 3 ? 2 : 3;
    return make_uchar4(b[i0], b[i1], b[i2], (unsigned char)(252 + mi));
}

// float32 value `steps` ulps above (steps > 0) or below a finite float, crossing zero correctly
__device__ __forceinline__ float ulp_step(float a, int steps)
{
    int b = (int)__float_as_uint(a);
    b = b < 0 ? (int)0x80000000u - b : b;   // ordered integer: monotone in the float value
    b += steps;
    b = b < 0 ? (int)0x80000000u - b : b;
    return __uint_as_float((unsigned)b);
}

// numpy's float32 SIMD routines: log max error 3.83 ulp, exp 2.52 ulp (their documented bounds; measured here on 28M values
// each: 3.01 and 2.52).  The brackets add the half ulp of rounding the float64 value to float32 and a margin.
constexpr int SOG_ULPS_LOG = 5, SOG_ULPS_EXP = 4;
// (no absolute slack near 0: for |v| + 1 within a few ulp of 1 numpy's log keeps its RELATIVE accuracy -- measured 1.2 ulp
//  at |v| ~ 1e-6 -- and an absolute term would flag every texel of a scene a few micro-units across)

// sog.py:279-309 for one value: v -> sign(v) log(|v| + 1) -> (l - mn) / (mx - mn) * 65535 -> clip -> u16.
// *ok = both ends of the bracket of numpy's possible float32 logarithm give the same texel
__device__ __forceinline__ unsigned sog_position_texel(float x, float mn, float range, bool *ok)
{
    const float t = __fadd_rn(fabsf(x), 1.0f);                       // np.abs(v) + 1.0 in float32
    const float sg = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);    // np.sign
    const double lt = (double)sg * ::log((double)t);
    const float a = (float)lt;
    float lo = ulp_step(a, -SOG_ULPS_LOG), hi = ulp_step(a, SOG_ULPS_LOG);
    if (sg == 0.0f) lo = hi = 0.0f;                                   // 0 * log(1) is exactly 0 whatever log returns
    unsigned q[2];
    const float e[2] = {lo, hi};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float r = __fdiv_rn(__fsub_rn(e[s], mn), range);
        r = __fmul_rn(r, 65535.0f);
        r = fminf(fmaxf(r, 0.0f), 65535.0f);
        q[s] = (unsigned)r;
    }
    *ok = q[0] == q[1] && (x == x) && fabsf(x) <= 3.0e38f && range > 0.0f;
    return q[0];
}

// sog.py:457-459 for one value: 1 / (1 + exp(-o)) * 255 -> clip -> u8
__device__ __forceinline__ unsigned sog_alpha_texel(float x, bool *ok_out)
{
    const double et = ::exp(-(double)x);
    bool ok = (x == x) && fabsf(x) < 80.0f;                          // outside: exp over/underflows in float32 -> host
    const float a = ok ? (float)et : 1.0f;
    const float e[2] = {ulp_step(a, -SOG_ULPS_EXP), ulp_step(a, SOG_ULPS_EXP)};
    unsigned q[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float r = __fdiv_rn(1.0f, __fadd_rn(1.0f, e[s]));
        r = __fmul_rn(r, 255.0f);
        r = fminf(fmaxf(r, 0.0f), 255.0f);
        q[s] = (unsigned)r;
    }
    *ok_out = ok && q[0] == q[1];
    return q[0];
}

// quantize_to_codebook, sog.py:408-419: np.searchsorted(cb, v) ('left'), clipped, left neighbour when STRICTLY nearer.
// cb: kcb ascending float32 entries (LDS or global)
__device__ __forceinline__ int sog_codebook_index(const float *cb, int kcb, float v)
{
    int lo = 0, hi = kcb;  // first index with cb[idx] >= v
    if (v != v) lo = kcb;  // numpy orders NaN after every number
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cb[mid] < v) lo = mid + 1; else hi = mid;
    }
    int idx = min(lo, kcb - 1);
    const int left = max(idx - 1, 0);
    const float d_idx = fabsf(v - cb[idx]);
    const float d_left = fabsf(v - cb[left]);
    if (d_left < d_idx) idx = left;  // strict: ties go to the right neighbour
    return idx;
}

}  // namespace gsx
