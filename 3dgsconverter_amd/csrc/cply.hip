// cply.hip -- numeric core of the compressed-PLY writer: Morton ordering, 256-splat chunk
// bounds, and the 11-10-11 / 2-10-10-10 / 8-8-8-8 / u8 quantisers.
//
// Replaces, in gsconverter/formats/compressed_ply.py:
//   _sort_morton_order            :245-291  recursive 10-bit-per-axis Morton sort (groups of equal code > 256 are re-sorted
//                                           inside their own bounding box)
//   write, chunk loop             :205-241  per-chunk min/max of position, clipped log-scale and linearised colour
//   _normalize_and_pack_11_10_11  :293-302
//   _normalize_and_pack_8888      :304-313
//   _pack_quaternions             :315-341
//   SH AC quantisation            :236-241
// The reference walks the chunks in a Python loop (39 063 iterations of ~70 numpy calls at 10M splats).  Here one
// workgroup owns one chunk.  Every float32 operation is the one numpy performs, in numpy's order (no contraction:
// the library is built with -ffp-contract=off; divisions and square roots are the correctly rounded ones), so the
// packed words are the reference's bit for bit GIVEN THE SAME ORDER.  The order itself: np.argsort (introsort /
// x86-simd-sort, not stable) leaves splats with EQUAL Morton code in an order that depends on numpy's build and the
// CPU; this implementation is stable (ties keep ascending input index).  The sequence of Morton codes is identical;
// only the order inside runs of equal code (<= 256 splats, or coincident points) can differ.
// Coordinates must be finite (NaN -> uint32 casts are undefined in the reference too).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

#include "gsx_common.h"
#include "sog_math.h"

namespace gsx {

constexpr unsigned CPLY_CHUNK = 256;   // compressed_ply.py:14 CHUNK_SIZE

// monotone float32 -> uint32 (atomicMin / atomicMax on the image order the floats)
__device__ __forceinline__ unsigned f2key(float f)
{
    const unsigned b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k)
{
    return __uint_as_float((k >> 31) ? (k ^ 0x80000000u) : ~k);
}

// ---------------------------------------------------------------- Morton order
// One recursion level of _sort_morton_order for ALL the groups that are still being refined at once.  The M "active"
// splats are described by dense arrays: pos[j] = position in the order array, seg[j] = position of the first splat of
// its group (groups are contiguous in the order array, and contiguous in the dense arrays).  The bounding box of a group
// lives at the dense index of the group's first splat, hd = j - (pos[j] - seg[j]).
__global__ __launch_bounds__(256) void mo_iota_kernel(unsigned *__restrict__ order, unsigned *__restrict__ pos,
                                                      unsigned *__restrict__ seg, unsigned n)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    order[i] = i;
    pos[i] = i;
    seg[i] = 0;
}

__global__ __launch_bounds__(256) void mo_fill_kernel(unsigned *__restrict__ p, size_t count, unsigned v)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < count) p[i] = v;
}

__global__ __launch_bounds__(256) void mo_bbox_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                      const float *__restrict__ z, int64_t stride,
                                                      const unsigned *__restrict__ order, const unsigned *__restrict__ pos,
                                                      const unsigned *__restrict__ seg, unsigned m, unsigned *__restrict__ bb)
{
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    const bool live = j < m;
    unsigned hd = 0xffffffffu;
    unsigned k[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    if (live) {
        const unsigned p = pos[j];
        hd = j - (p - seg[j]);
        const int64_t i = (int64_t)order[p] * stride;
        k[0] = k[3] = f2key(x[i]);
        k[1] = k[4] = f2key(y[i]);
        k[2] = k[5] = f2key(z[i]);
    }
    const unsigned hd0 = __builtin_amdgcn_readfirstlane(hd);
    if (__all(hd == hd0 || !live) && hd0 != 0xffffffffu) {   // the whole wave is inside one group (the common case)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                k[a] = min(k[a], (unsigned)__shfl_xor((int)k[a], off));
                k[a + 3] = max(k[a + 3], (unsigned)__shfl_xor((int)k[a + 3], off));
            }
        }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                atomicMin(&bb[(size_t)a * m + hd0], k[a]);
                atomicMax(&bb[(size_t)(a + 3) * m + hd0], k[a + 3]);
            }
        }
    } else if (live) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&bb[(size_t)a * m + hd], k[a]);
            atomicMax(&bb[(size_t)(a + 3) * m + hd], k[a + 3]);
        }
    }
}

__device__ __forceinline__ unsigned part_1_by_2(unsigned n)   // compressed_ply.py:249-255
{
    n &= 0x000003ffu;
    n = (n ^ (n << 16)) & 0xff0000ffu;
    n = (n ^ (n << 8)) & 0x0300f00fu;
    n = (n ^ (n << 4)) & 0x030c30c3u;
    n = (n ^ (n << 2)) & 0x09249249u;
    return n;
}

__device__ __forceinline__ unsigned axis_cell(float c, float lo, float len)
{
    const float mul = len > 0.0f ? __fdiv_rn(1024.0f, len) : 0.0f;   // 1024.0 / xlen if xlen > 0 else 0   (:268-270)
    float v = __fmul_rn(__fsub_rn(c, lo), mul);
    v = fminf(fmaxf(v, 0.0f), 1023.0f);                              // np.clip(.., 0, 1023)
    return (unsigned)v;                                               // astype(np.uint32): truncation
}

struct SegBox {
    float lo[3], len[3];
    bool degenerate;
};

__device__ __forceinline__ SegBox seg_box(const unsigned *__restrict__ bb, unsigned m, unsigned hd)
{
    SegBox b;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = key2f(bb[(size_t)a * m + hd]);
        b.len[a] = __fsub_rn(key2f(bb[(size_t)(a + 3) * m + hd]), b.lo[a]);
    }
    b.degenerate = b.len[0] == 0.0f && b.len[1] == 0.0f && b.len[2] == 0.0f;   // :265 -> the group is left as it is
    return b;
}

__global__ __launch_bounds__(256) void mo_codes_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                       const float *__restrict__ z, int64_t stride,
                                                       const unsigned *__restrict__ order, const unsigned *__restrict__ pos,
                                                       const unsigned *__restrict__ seg, unsigned m,
                                                       const unsigned *__restrict__ bb, unsigned long long *__restrict__ key,
                                                       unsigned *__restrict__ val)
{
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    if (j >= m) return;
    const unsigned p = pos[j], s = seg[j];
    const SegBox b = seg_box(bb, m, j - (p - s));
    const unsigned src = order[p];
    const int64_t i = (int64_t)src * stride;
    unsigned code = 0;
    if (!b.degenerate)
        code = (part_1_by_2(axis_cell(z[i], b.lo[2], b.len[2])) << 2) | (part_1_by_2(axis_cell(y[i], b.lo[1], b.len[1])) << 1) |
               part_1_by_2(axis_cell(x[i], b.lo[0], b.len[0]));
    key[j] = ((unsigned long long)s << 32) | code;   // groups stay where they are, splats move inside their group
    val[j] = src;
}

__global__ __launch_bounds__(256) void mo_writeback_kernel(unsigned *__restrict__ order, const unsigned *__restrict__ pos,
                                                           const unsigned *__restrict__ val, const unsigned long long *__restrict__ key,
                                                           unsigned m, unsigned *__restrict__ head)
{
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    if (j >= m) return;
    order[pos[j]] = val[j];
    head[j] = (j == 0 || key[j] != key[j - 1]) ? j : 0u;   // inclusive max-scan -> dense index of the run's first splat
}

// a run of equal (group, code) longer than 256 splats becomes a group of the next level (:283-285), unless its parent
// had no extent at all (then it was not sorted and is not looked at again)
__global__ __launch_bounds__(256) void mo_active_kernel(const unsigned *__restrict__ pos, const unsigned *__restrict__ seg,
                                                        const unsigned long long *__restrict__ key, const unsigned *__restrict__ headpos,
                                                        const unsigned *__restrict__ bb, unsigned m, uint8_t *__restrict__ flag,
                                                        unsigned *__restrict__ newseg)
{
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    if (j >= m) return;
    const unsigned h = headpos[j], p = pos[j];
    bool act = h + CPLY_CHUNK < m && key[h + CPLY_CHUNK] == key[j];
    if (act) act = !seg_box(bb, m, j - (p - seg[j])).degenerate;
    flag[j] = act;
    newseg[j] = p - (j - h);
}

struct Bump {
    char *p;
    size_t left;
    template <class T>
    T *take(size_t count)
    {
        const size_t bytes = (sizeof(T) * count + 255) & ~(size_t)255;
        if (bytes > left) return nullptr;
        T *r = reinterpret_cast<T *>(p);
        p += bytes;
        left -= bytes;
        return r;
    }
};

static int morton_order_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n64,
                            unsigned *order, int *levels_out)
{
    const unsigned n = (unsigned)n64;
    size_t t_sort = 0, t_scan = 0, t_sel = 0;
    {
        unsigned long long *k = nullptr;
        unsigned *v = nullptr;
        uint8_t *f = nullptr;
        GSX_HIP(rocprim::radix_sort_pairs(nullptr, t_sort, k, k, v, v, (size_t)n, 0, 64, c->stream));
        GSX_HIP(rocprim::inclusive_scan(nullptr, t_scan, v, v, (size_t)n, rocprim::maximum<unsigned>(), c->stream));
        GSX_HIP(rocprim::select(nullptr, t_sel, v, f, v, v, (size_t)n, c->stream));
    }
    const size_t temp_bytes = std::max(t_sort, std::max(t_scan, t_sel)) + 256;
    const size_t per = (size_t)n + 64;
    // pos/seg (x2, ping-pong), bbox 6, key x2 (u64), val x2, flag, count
    const size_t total = 256 * 16 + temp_bytes + per * (4 * 4 + 6 * 4 + 2 * 8 + 2 * 4 + 1) + 1024;
    GSX_CHECK(c->scratch5.reserve(total));
    Bump bump{c->scratch5.as<char>(), total};
    unsigned *posA = bump.take<unsigned>(per), *segA = bump.take<unsigned>(per);
    unsigned *posB = bump.take<unsigned>(per), *segB = bump.take<unsigned>(per);
    unsigned *bb = bump.take<unsigned>(6 * per);
    unsigned long long *keyA = bump.take<unsigned long long>(per), *keyB = bump.take<unsigned long long>(per);
    unsigned *valA = bump.take<unsigned>(per), *valB = bump.take<unsigned>(per);
    uint8_t *flag = bump.take<uint8_t>(per);
    unsigned *count = bump.take<unsigned>(64);
    void *temp = bump.take<char>(temp_bytes);
    if (!temp) GSX_FAIL("morton_order: scratch layout overflow");

    hipLaunchKernelGGL(mo_iota_kernel, dim3(div_up((int64_t)n, 256)), dim3(256), 0, c->stream, order, posA, segA, n);
    unsigned m = n;
    int level = 0;
    int index_bits = 1;
    while ((1ull << index_bits) < (unsigned long long)n) ++index_bits;
    while (m > 1) {
        if (level >= 64) GSX_FAIL("morton_order: more than 64 refinement levels (non-finite coordinates?)");
        const dim3 grid((unsigned)div_up((int64_t)m, 256));
        hipLaunchKernelGGL(mo_fill_kernel, dim3((unsigned)div_up((int64_t)3 * m, 256)), dim3(256), 0, c->stream, bb, (size_t)3 * m, 0xffffffffu);
        hipLaunchKernelGGL(mo_fill_kernel, dim3((unsigned)div_up((int64_t)3 * m, 256)), dim3(256), 0, c->stream, bb + (size_t)3 * m, (size_t)3 * m, 0u);
        hipLaunchKernelGGL(mo_bbox_kernel, grid, dim3(256), 0, c->stream, x, y, z, stride, order, posA, segA, m, bb);
        hipLaunchKernelGGL(mo_codes_kernel, grid, dim3(256), 0, c->stream, x, y, z, stride, order, posA, segA, m, bb, keyA, valA);
        size_t tb = temp_bytes;
        GSX_HIP(rocprim::radix_sort_pairs(temp, tb, keyA, keyB, valA, valB, (size_t)m, 0, level == 0 ? 30 : 32 + index_bits, c->stream));
        hipLaunchKernelGGL(mo_writeback_kernel, grid, dim3(256), 0, c->stream, order, posA, valB, keyB, m, valA);
        tb = temp_bytes;
        GSX_HIP(rocprim::inclusive_scan(temp, tb, valA, valB, (size_t)m, rocprim::maximum<unsigned>(), c->stream));
        hipLaunchKernelGGL(mo_active_kernel, grid, dim3(256), 0, c->stream, posA, segA, keyB, valB, bb, m, flag, valA);
        tb = temp_bytes;
        GSX_HIP(rocprim::select(temp, tb, posA, flag, posB, count, (size_t)m, c->stream));
        tb = temp_bytes;
        GSX_HIP(rocprim::select(temp, tb, valA, flag, segB, count + 1, (size_t)m, c->stream));
        GSX_HIP(hipGetLastError());
        unsigned h = 0;
        GSX_HIP(hipMemcpyAsync(&h, count, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        GSX_HIP(hipStreamSynchronize(c->stream));
        std::swap(posA, posB);
        std::swap(segA, segB);
        m = h;
        ++level;
    }
    if (levels_out) *levels_out = level;
    return 0;
}

// ---------------------------------------------------------------- chunk bounds + packers
struct CplyCols {
    const float *col[14];   // x y z | scale_0..2 | f_dc_0..2 | alpha (sigmoid of opacity, computed by numpy) | rot_0..3
    int64_t stride[14];     // elements between consecutive splats of column a: 1 = a contiguous column, row_bytes / 4 = the field
                            // inside raw rows resident in HBM (round 6: the 13 fields of one splat then share two cache lines)
};

__device__ __forceinline__ unsigned quant_unit(float v, float lo, float hi, float t)
{
    // normalize (:294-298 / :305-308): zero when the range is below 1e-5, else clip(floor((v-lo)/(hi-lo) * t + 0.5), 0, t)
    const float range = __fsub_rn(hi, lo);
    if (range < 1e-5f) return 0u;
    const float norm = __fdiv_rn(__fsub_rn(v, lo), range);
    const float q = floorf(__fadd_rn(__fmul_rn(norm, t), 0.5f));
    return (unsigned)fminf(fmaxf(q, 0.0f), t);
}

__device__ __forceinline__ unsigned pack_quat(float q0, float q1, float q2, float q3)   // :315-341
{
    float q[4] = {q0, q1, q2, q3};
    // np.linalg.norm(axis=-1): sqrt(add.reduce(q*q)) -- numpy adds the four squares left to right
    const float ss = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(q0, q0), __fmul_rn(q1, q1)), __fmul_rn(q2, q2)), __fmul_rn(q3, q3));
    const float d = __fadd_rn(__builtin_sqrtf(ss) /* correctly rounded; __fsqrt_rn is the native approximation */, 1e-10f);
    int largest = 0;
    float best = -1.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        q[a] = __fdiv_rn(q[a], d);
        const float v = fabsf(q[a]);
        if (v > best) {   // np.argmax: first occurrence of the maximum
            best = v;
            largest = a;
        }
    }
    const float lv = q[largest];
    const float sg = lv > 0.0f ? 1.0f : (lv < 0.0f ? -1.0f : 0.0f);   // np.sign
    unsigned res = (unsigned)largest;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float v = __fmul_rn(q[a], sg);
        // pack_unorm: clip(floor((v * SQRT2_2 + 0.5) * 1023 + 0.5), 0, 1023); the Python-float constant is a weak scalar
        const float t = __fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(v, 0.70710678118654757f), 0.5f), 1023.0f), 0.5f);
        const unsigned pc = (unsigned)fminf(fmaxf(floorf(t), 0.0f), 1023.0f);
        if (a != largest) res = (res << 10) | pc;
    }
    return res;
}

// alpha byte of the packed colour from the OPACITY (round 6): the reference evaluates `1.0 / (1.0 + np.exp(-x))` with numpy's float32
// SIMD exp (not correctly rounded: <= 2.52 ulp), then floor(a * 255 + 0.5) (:200-203, :312).  As for the SOG textures (sog_math.h):
// exp in float64, both ends of the bracket of numpy's possible float32 result through numpy's exact float32 sequence; equal codes
// -> certain, else the splat goes on a short list and the host patches the byte with numpy's own expression.
__device__ __forceinline__ unsigned cply_alpha_code(float x, bool *ok_out)
{
    const double et = ::exp(-(double)x);
    bool ok = (x == x) && fabsf(x) < 80.0f;
    const float a = ok ? (float)et : 1.0f;
    const float e[2] = {ulp_step(a, -SOG_ULPS_EXP), ulp_step(a, SOG_ULPS_EXP)};
    unsigned q[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float r = __fdiv_rn(1.0f, __fadd_rn(1.0f, e[s]));
        q[s] = (unsigned)fminf(fmaxf(floorf(__fadd_rn(__fmul_rn(r, 255.0f), 0.5f)), 0.0f), 255.0f);
    }
    *ok_out = ok && q[0] == q[1];
    return q[0];
}

__global__ __launch_bounds__(256) void cply_pack_kernel(CplyCols c, const unsigned *__restrict__ order, int64_t n,
                                                        float *__restrict__ chunk_out, uint4 *__restrict__ vertex_out,
                                                        uint2 *__restrict__ unc_list /* null: column 9 is numpy's alpha */, unsigned unc_cap,
                                                        unsigned *__restrict__ unc_count)
{
    __shared__ float s_part[4][18];
    __shared__ float s_box[18];
    const int64_t i = (int64_t)blockIdx.x * CPLY_CHUNK + threadIdx.x;
    const bool live = i < n;
    float v[9], al = 0.0f, rq[4] = {0, 0, 0, 0};
    if (live) {
        const size_t s = order ? order[i] : (size_t)i;
#pragma unroll
        for (int a = 0; a < 3; ++a) v[a] = c.col[a][s * c.stride[a]];
#pragma unroll
        for (int a = 3; a < 6; ++a) v[a] = fminf(fmaxf(c.col[a][s * c.stride[a]], -20.0f), 20.0f);            // np.clip(scale, -20, 20)  (:212-214)
#pragma unroll
        for (int a = 6; a < 9; ++a) v[a] = __fadd_rn(__fmul_rn(c.col[a][s * c.stride[a]], 0.28209479177387814f), 0.5f);   // f_dc * SH_C0 + 0.5  (:195-198)
        al = c.col[9][s * c.stride[9]];
#pragma unroll
        for (int a = 0; a < 4; ++a) rq[a] = c.col[10 + a][s * c.stride[10 + a]];
    }
    float lo[9], hi[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        lo[a] = live ? v[a] : __builtin_inff();
        hi[a] = live ? v[a] : -__builtin_inff();
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 9; ++a) {
            s_part[threadIdx.x >> 6][a] = lo[a];
            s_part[threadIdx.x >> 6][9 + a] = hi[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 18) {
        const int a = threadIdx.x;
        float r = s_part[0][a];
        for (int w = 1; w < 4; ++w) r = a < 9 ? fminf(r, s_part[w][a]) : fmaxf(r, s_part[w][a]);
        s_box[a] = r;
        // chunk record (:167-174): min xyz, max xyz, min scale, max scale, min rgb, max rgb
        const int g = a < 9 ? a / 3 : (a - 9) / 3, e = a < 9 ? a % 3 : (a - 9) % 3;
        chunk_out[(size_t)blockIdx.x * 18 + g * 6 + (a < 9 ? 0 : 3) + e] = r;
    }
    __syncthreads();
    if (!live) return;
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        lo[a] = s_box[a];
        hi[a] = s_box[9 + a];
    }
    uint4 o;
    o.x = (quant_unit(v[0], lo[0], hi[0], 2047.0f) << 21) | (quant_unit(v[1], lo[1], hi[1], 1023.0f) << 11) | quant_unit(v[2], lo[2], hi[2], 2047.0f);
    o.y = pack_quat(rq[0], rq[1], rq[2], rq[3]);
    o.z = (quant_unit(v[3], lo[3], hi[3], 2047.0f) << 21) | (quant_unit(v[4], lo[4], hi[4], 1023.0f) << 11) | quant_unit(v[5], lo[5], hi[5], 2047.0f);
    unsigned na;
    if (unc_list) {   // column 9 is the opacity itself
        bool ok;
        na = cply_alpha_code(al, &ok);
        if (!ok) {
            const unsigned p = atomicAdd(unc_count, 1u);
            if (p < unc_cap) unc_list[p] = make_uint2((unsigned)i, __float_as_uint(al));
        }
    } else {
        na = (unsigned)fminf(fmaxf(floorf(__fadd_rn(__fmul_rn(al, 255.0f), 0.5f)), 0.0f), 255.0f);   // :312
    }
    o.w = (quant_unit(v[6], lo[6], hi[6], 255.0f) << 24) | (quant_unit(v[7], lo[7], hi[7], 255.0f) << 16) |
          (quant_unit(v[8], lo[8], hi[8], 255.0f) << 8) | na;
    vertex_out[i] = o;
}

// SH AC (:236-241): u8( clip((v / 8.0 + 0.5) * 256, 0, 255) ), m columns of the original table -> (n, m) bytes in the new order.
// Coefficient c of splat s is cols[c * col_stride + s * elem_stride]: contiguous columns (col_stride >= n, elem_stride 1) or the
// consecutive f_rest fields of raw rows (col_stride 1, elem_stride = row_bytes / 4).  Round 6: lane = coefficient, a wave walks
// its 64 splats (one 180-byte read per splat in row mode), the block's 256 x m bytes leave LDS as whole dwords -- rounds 3-5
// had every thread store its 45 bytes one by one, 45 bytes apart from its neighbour's.
__global__ __launch_bounds__(256) void cply_sh_kernel(const float *__restrict__ cols, int m, int64_t col_stride, int64_t elem_stride,
                                                      const unsigned *__restrict__ order, int64_t n, uint8_t *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_out[256 * 48];
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    const int rows_here = (int)min((int64_t)256, n - i0);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int r = wv * 64; r < min(rows_here, wv * 64 + 64); ++r) {
        const size_t s = order ? order[i0 + r] : (size_t)(i0 + r);   // wave-uniform
        if (lane < m) {
            const float t = __fmul_rn(__fadd_rn(__fmul_rn(cols[(size_t)lane * col_stride + s * elem_stride], 0.125f), 0.5f), 256.0f);
            s_out[r * m + lane] = (uint8_t)fminf(fmaxf(t, 0.0f), 255.0f);
        }
    }
    __syncthreads();
    const int total = rows_here * m;
    uint8_t *dst = out + (size_t)i0 * m;   // 256 m bytes per block: dword aligned for every m
    for (int e = threadIdx.x * 4; e < (total & ~3); e += 256 * 4) *reinterpret_cast<unsigned *>(dst + e) = *reinterpret_cast<const unsigned *>(s_out + e);
    if (threadIdx.x < (total & 3)) dst[(total & ~3) + threadIdx.x] = s_out[(total & ~3) + threadIdx.x];
}

}  // namespace gsx

using namespace gsx;

// Rows whose size is not a multiple of 4 (the table widened by three u1 colour fields, data_processor.py:262-274: 251 bytes) copied
// to rows of `dst_pitch` bytes (a multiple of 4): fields at 4-byte offsets inside a row then sit on the 4-byte grid of the device
// buffer and the strided packers below read them with plain float loads.  One thread per destination dword; the two source
// dwords it straddles are shared with its neighbours through the L1 / L2.
__global__ __launch_bounds__(256) void rows_repack_kernel(const unsigned *__restrict__ src, int64_t row_bytes, int64_t n, unsigned *__restrict__ dst,
                                                          int dst_dwords)
{
    const int64_t total = n * dst_dwords;
    const int row_dwords = (int)((row_bytes + 3) >> 2);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / dst_dwords;
        const int j = (int)(e - r * dst_dwords);
        unsigned v = 0u;
        if (j < row_dwords) {
            const int64_t b = r * row_bytes + 4 * (int64_t)j;          // first source byte of this dword
            const unsigned lo = src[b >> 2], hi = src[(b >> 2) + 1];   // (one dword past the table at most: see the header)
            v = __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(b & 3));
            const int valid = (int)min((int64_t)4, row_bytes - 4 * (int64_t)j);   // the row's last dword: the bytes behind it are the NEXT row's
            if (valid < 4) v &= (1u << (8 * valid)) - 1u;
        }
        dst[e] = v;
    }
}

// compressed_ply.py:139-150 (and sog.py:476-486): which of m consecutive float fields of the rows hold a value != 0 (NaN counts, -0.0
// does not: numpy's `data[f] != 0`).  Lane = field, a wave walks rows with 8 loads in flight; one OR per wave at the end.
__global__ __launch_bounds__(256) void fields_nonzero_kernel(const float *__restrict__ base, int64_t row_stride, int64_t n, int m,
                                                             unsigned long long *__restrict__ mask_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
    bool any = false;
    if (lane < m) {
        const float *col = base + lane;
        int64_t r = wave * 8;
        for (; r + 8 <= n; r += nwaves * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[(r + u) * row_stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) any |= v[u] != 0.0f;
        }
        for (; r < n; ++r) any |= col[r * row_stride] != 0.0f;   // (the last, partial group of eight rows: one wave gets here with r < n)
    }
    const unsigned long long bits = __ballot(any);
    if (lane == 0 && bits) atomicOr(mask_out, bits);
}

extern "C" {

int gsx_morton_order_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                         uint32_t *order_out_dev, int *levels_out)
{
    if (!c || (n > 0 && (!x || !y || !z || !order_out_dev))) GSX_FAIL("gsx_morton_order_dev: null argument");
    if (n < 0 || n >= (1LL << 31) || stride < 1) GSX_FAIL("gsx_morton_order_dev: bad size");
    GSX_HIP(hipSetDevice(c->device));
    if (levels_out) *levels_out = 0;
    if (n == 0) return 0;
    return morton_order_dev(c, x, y, z, stride, n, order_out_dev, levels_out);
}

int gsx_cply_pack_dev(gsx_ctx *c, const float *const *cols14_dev, const uint32_t *order_dev, int64_t n, float *chunk_out_dev,
                      uint32_t *vertex_out_dev)
{
    return gsx_cply_pack_strided_dev(c, cols14_dev, nullptr, order_dev, n, chunk_out_dev, vertex_out_dev);
}

int gsx_cply_pack_strided_dev(gsx_ctx *c, const float *const *cols14_dev, const int64_t *strides14, const uint32_t *order_dev, int64_t n,
                              float *chunk_out_dev, uint32_t *vertex_out_dev)
{
    return gsx_cply_pack_opacity_dev(c, cols14_dev, strides14, order_dev, n, chunk_out_dev, vertex_out_dev, nullptr, 0, nullptr);
}

int gsx_cply_pack_opacity_dev(gsx_ctx *c, const float *const *cols14_dev, const int64_t *strides14, const uint32_t *order_dev, int64_t n,
                              float *chunk_out_dev, uint32_t *vertex_out_dev, uint32_t *list_dev, int64_t cap, uint32_t *count_dev)
{
    if ((list_dev != nullptr) != (count_dev != nullptr) || cap < 0 || cap > 0xffffffffLL) GSX_FAIL("gsx_cply_pack_opacity_dev: list and counter come together");
    if (!c || !cols14_dev || (n > 0 && (!chunk_out_dev || !vertex_out_dev))) GSX_FAIL("gsx_cply_pack_dev: null argument");
    if (n < 0 || n >= (1LL << 32)) GSX_FAIL("gsx_cply_pack_dev: bad size");
    if (reinterpret_cast<uintptr_t>(vertex_out_dev) & 15) GSX_FAIL("gsx_cply_pack_dev: vertex output must be 16-byte aligned");
    GSX_HIP(hipSetDevice(c->device));
    if (n == 0) return 0;
    CplyCols cc;
    for (int a = 0; a < 14; ++a) {
        if (!cols14_dev[a]) GSX_FAIL("gsx_cply_pack_dev: column %d is null", a);
        cc.col[a] = cols14_dev[a];
        cc.stride[a] = strides14 ? strides14[a] : 1;
        if (cc.stride[a] < 1) GSX_FAIL("gsx_cply_pack_strided_dev: stride %d must be >= 1", a);
    }
    if (count_dev) GSX_HIP(hipMemsetAsync(count_dev, 0, 4, c->stream));
    hipLaunchKernelGGL(cply_pack_kernel, dim3((unsigned)div_up(n, (int64_t)CPLY_CHUNK)), dim3(256), 0, c->stream, cc, order_dev, n,
                       chunk_out_dev, reinterpret_cast<uint4 *>(vertex_out_dev), reinterpret_cast<uint2 *>(list_dev), (unsigned)cap, count_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

int gsx_cply_sh_dev(gsx_ctx *c, const float *cols_dev, int m, int64_t col_stride, const uint32_t *order_dev, int64_t n,
                    uint8_t *out_dev)
{
    if (n > 0 && col_stride < n) GSX_FAIL("gsx_cply_sh_dev: bad size");
    return gsx_cply_sh_strided_dev(c, cols_dev, m, col_stride, 1, order_dev, n, out_dev);
}

int gsx_cply_sh_strided_dev(gsx_ctx *c, const float *cols_dev, int m, int64_t col_stride, int64_t elem_stride, const uint32_t *order_dev,
                            int64_t n, uint8_t *out_dev)
{
    if (!c || (n > 0 && m > 0 && (!cols_dev || !out_dev))) GSX_FAIL("gsx_cply_sh_dev: null argument");
    if (n < 0 || n >= (1LL << 32) || m < 0 || m > 45 || col_stride < 1 || elem_stride < 1) GSX_FAIL("gsx_cply_sh_dev: bad size");
    if (reinterpret_cast<uintptr_t>(out_dev) & 3) GSX_FAIL("gsx_cply_sh_dev: output must be 4-byte aligned");
    GSX_HIP(hipSetDevice(c->device));
    if (n == 0 || m == 0) return 0;
    hipLaunchKernelGGL(cply_sh_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, c->stream, cols_dev, m, col_stride, elem_stride, order_dev, n,
                       out_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

int gsx_rows_repack_dev(gsx_ctx *c, const void *rows_dev, int64_t row_bytes, int64_t n, void *out_dev, int64_t out_pitch)
{
    if (!c || (n > 0 && (!rows_dev || !out_dev))) GSX_FAIL("gsx_rows_repack_dev: null argument");
    if (n < 0 || row_bytes < 1 || out_pitch < row_bytes || (out_pitch & 3) || out_pitch > (1 << 20)) GSX_FAIL("gsx_rows_repack_dev: bad size");
    if ((reinterpret_cast<uintptr_t>(rows_dev) & 3) || (reinterpret_cast<uintptr_t>(out_dev) & 3)) GSX_FAIL("gsx_rows_repack_dev: buffers must be 4-byte aligned");
    GSX_HIP(hipSetDevice(c->device));
    if (n == 0) return 0;
    const int dd = (int)(out_pitch / 4);
    const int64_t total = n * dd;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(div_up(total, (int64_t)1024), (int64_t)c->num_cu * 16));
    hipLaunchKernelGGL(rows_repack_kernel, dim3(blocks), dim3(256), 0, c->stream, static_cast<const unsigned *>(rows_dev), row_bytes, n,
                       static_cast<unsigned *>(out_dev), dd);
    GSX_HIP(hipGetLastError());
    return 0;
}

int gsx_fields_nonzero_dev(gsx_ctx *c, const float *first_field_dev, int64_t row_stride, int64_t n, int m, uint64_t *mask_out)
{
    if (!c || !mask_out || (n > 0 && m > 0 && !first_field_dev)) GSX_FAIL("gsx_fields_nonzero_dev: null argument");
    if (n < 0 || m < 0 || m > 64 || row_stride < m) GSX_FAIL("gsx_fields_nonzero_dev: bad size");
    GSX_HIP(hipSetDevice(c->device));
    *mask_out = 0;
    if (n == 0 || m == 0) return 0;
    GSX_CHECK(c->nzmask.reserve(16));
    unsigned long long *d_mask = c->nzmask.as<unsigned long long>();
    GSX_HIP(hipMemsetAsync(d_mask, 0, 8, c->stream));
    const int64_t groups = div_up(n, (int64_t)8);
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(div_up(groups, (int64_t)4), (int64_t)c->num_cu * 16));
    hipLaunchKernelGGL(fields_nonzero_kernel, dim3(blocks), dim3(256), 0, c->stream, first_field_dev, row_stride, n, m, d_mask);
    GSX_HIP(hipGetLastError());
    GSX_HIP(hipMemcpyAsync(mask_out, d_mask, 8, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
