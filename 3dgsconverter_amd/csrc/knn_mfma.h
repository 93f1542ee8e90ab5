// knn_mfma.h -- phase-1 filter arithmetic shared by the two wave-per-brick exact-KNN kernels (knn_brick of the uniform
// grid, sor_grid.hip; knn_leaf of the Morton tree, sor_tree.hip): the float32 sign-bit filter step and the bf16-split
// matrix-core filter (v_mfma_f32_32x32x16_bf16).  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

namespace gsx {

// Phase-1 filter step: shift the predicate "squared distance < tau" into the lane's bit mask.
// d2 - tau is evaluated as one fma chain and its SIGN BIT is the predicate, so the shift-in is a
// single v_alignbit_b32: ({m, t} >> 31) = (m << 1) | sign(t).  (v_cmp + v_addc cost 2 x 4.4
// cycles on gfx950 -- tools/ubench/valu_rates.hip -- i.e. 38 % of the whole filter step.)
// Rounding: the three fma roundings perturb t by <= 3 * 2^-24 * max(tau, d2), far inside the
// 2e-6 relative slack already built into tau (knn_common.h F32_SLACK).
__device__ __forceinline__ unsigned shift_in_lt(unsigned m, float qx, float qy, float qz, float px, float py,
                                                float pz, float neg_tau)
{
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    const float t = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, neg_tau)));
    return __builtin_amdgcn_alignbit(m, __float_as_uint(t), 31);
}

// ---- MFMA phase-1 filter (knn_brick<.., MF=true>) ------------------------------------------------
// d2 - tau for 32 candidates x 32 queries is ONE v_mfma_f32_32x32x16_bf16: with coordinates taken
// relative to the brick centre in CELL units (|u| <= ~2), every f32 value is split into bf16 pieces
// (v = vh + vl, |v - vh - vl| <= 2^-18 |v|) and the K = 16 slots hold
//     -2 p.q  ~  sum_c  ph_c*(-2 qh_c) + ph_c*(-2 ql_c) + pl_c*(-2 qh_c)          (9 slots)
//     |p|^2   =  n1 + n2 + n3 (three bf16 pieces) times 1                          (3 slots)
//     |q|^2 - tau - slack = s1 + s2 + s3, times 1                                  (3 slots, 1 spare)
// bf16 x bf16 products are exact in the f32 accumulator.  Error of the accumulated value against
// the true (d2 - tau), in cell units^2: dropped pl*ql and split residuals <= 2*3*3*2^-18*|p||q|
// <= 1.4e-4, recentring + norm roundings <= 1e-5, 16 f32 accumulations of partial sums <= ~30:
// <= 1e-4 even at 4 ulp each -- together < 3e-4.  MF_SLACK = 1e-3 (cell units^2, i.e. 0.05 % of the
// radius at r ~ 1 cell) makes the filter conservative: every candidate with d2 <= tau sets its bit;
// the few extra ones are discarded by the exact float64 phase 2 as before.
// Lane layout (measured, tools/ubench/mfma_layout.hip): A/B lane l holds row/col l&31, k =
// 8*(l>>5)+0..7; D lane l holds col l&31, rows (r&3) + 8*(r>>2) + 4*(l>>5) for r in [0,16).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr float MF_SLACK = 1e-3f;

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi)  // RNE, lo -> bits 0..15
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// The sign bits come out of the accumulators in the order pos -> tile row (pos&3) + 8*((pos>>2)&3) +
// 4*(pos>>4) (pos 0 = most significant mask bit).  Tile row r is therefore loaded with candidate
// mf_cand_of_row(r), the inverse permutation, and mask bit 31-i means candidate i of the word exactly as
// in the scalar filter.
__host__ __device__ constexpr int mf_cand_of_row(int r) { return (r & 3) | (((r >> 3) & 3) << 2) | (((r >> 2) & 1) << 4); }

// the K-slices of one candidate (cell-unit coordinates relative to the brick centre)
__device__ __forceinline__ bf16x8 mf_candidate_operand(float ux, float uy, float uz, bool upper)
{
    const unsigned l0 = cvt_pk_bf16(ux, ux);       // (ph_x, ph_x)
    const float lx = ux - bf_lo(l0);
    const unsigned l1 = cvt_pk_bf16(lx, uy);       // (pl_x, ph_y)
    const float ly = uy - bf_hi(l1);
    const unsigned l2 = cvt_pk_bf16(uy, ly);       // (ph_y, pl_y)
    const unsigned l3 = cvt_pk_bf16(uz, uz);       // (ph_z, ph_z)
    const float lz = uz - bf_lo(l3);
    const float n = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
    const unsigned u0 = cvt_pk_bf16(lz, n);        // (pl_z, n1)
    const float r1 = n - bf_hi(u0);
    const unsigned t = cvt_pk_bf16(r1, r1);
    const float r2 = r1 - bf_lo(t);
    const unsigned u1 = cvt_pk_bf16(r1, r2);       // (n2, n3)
    u32x4 w;
    w.x = upper ? u0 : l0;
    w.y = upper ? u1 : l1;
    w.z = upper ? 0x3f803f80u : l2;                // (1, 1)
    w.w = upper ? 0x00003f80u : l3;                // (1, 0)
    return __builtin_bit_cast(bf16x8, w);
}

// the query's K-slices for both tiles: out_a = operand of the tile whose columns are queries
// 0..31 (lanes 0..31), out_b = queries 32..63.  s = |q|^2 - tau - slack (+1e30 for a dead lane).
__device__ __forceinline__ void mf_query_operands(float ux, float uy, float uz, float s, bf16x8 &out_a, bf16x8 &out_b)
{
    const unsigned hx = cvt_pk_bf16(ux, uy);       // (qh_x, qh_y)
    const unsigned hz = cvt_pk_bf16(uz, uz);
    const float hxf = bf_lo(hx), hyf = bf_hi(hx), hzf = bf_lo(hz);
    const float lx = ux - hxf, ly = uy - hyf, lz = uz - hzf;
    u32x4 lo, up;
    lo.x = cvt_pk_bf16(-2.0f * hxf, -2.0f * lx);   // k0 k1
    lo.y = cvt_pk_bf16(-2.0f * hxf, -2.0f * hyf);  // k2 k3
    lo.z = cvt_pk_bf16(-2.0f * ly, -2.0f * hyf);   // k4 k5
    lo.w = cvt_pk_bf16(-2.0f * hzf, -2.0f * lz);   // k6 k7
    const unsigned t1 = cvt_pk_bf16(s, s);
    const float r1 = s - bf_lo(t1);
    const unsigned t2 = cvt_pk_bf16(r1, r1);
    const float r2 = r1 - bf_lo(t2);
    up.x = cvt_pk_bf16(-2.0f * hzf, 1.0f);         // k8 k9
    up.y = 0x3f803f80u;                            // k10 k11
    up.z = cvt_pk_bf16(s, r1);                     // k12 k13 = (s1, s2)
    up.w = cvt_pk_bf16(r2, 0.0f);                  // k14 k15 = (s3, 0)
    // tile A: lanes < 32 supply their own k0..7, lanes >= 32 the k8..15 of query (lane - 32);
    // tile B: lanes < 32 the k0..7 of query (lane + 32), lanes >= 32 their own k8..15.
    // v_permlane32_swap(x, y) exchanges x[32..63] with y[0..31]: one swap per dword makes both.
    u32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        auto r = __builtin_amdgcn_permlane32_swap(lo[i], up[i], false, false);
        a[i] = r[0];
        b[i] = r[1];
    }
    out_a = __builtin_bit_cast(bf16x8, a);
    out_b = __builtin_bit_cast(bf16x8, b);
}

// sign bits of the 16 accumulators -> bits 15..0 (register 0 first)
__device__ __forceinline__ unsigned mf_sign_bits(const f32x16 &acc)
{
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) m = __builtin_amdgcn_alignbit(m, __float_as_uint(acc[r]), 31);
    return m;
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace gsx
