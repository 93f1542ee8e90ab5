// chain.hip -- device-resident filter chain: stable compaction of the xyz rows between filters.
//
// The reference applies its filters one after the other on the host table (converter.py:196-236: bbox, alpha, density,
// SOR), each ending in `self.data = vertices[mask]` (data_processor.py:114,149).  Keeping the coordinates in HBM
// across density -> SOR (BASELINE.json configs[2]) removes the second gather + upload and lets the host compact its
// 248-byte rows ONCE, by the composed survivor list: SURVEY.md 8(f) rank 1, device half.
//   rows_out / orig_out = the rows with mask != 0, order kept; orig = index into the table the chain started from
#include <algorithm>

#include "gsx_common.h"

namespace gsx {

constexpr int CMP_TILE = 2048;

__global__ __launch_bounds__(256) void compact_count_kernel(const uint8_t *__restrict__ mask, int64_t n, unsigned *__restrict__ tile_cnt)
{
    __shared__ unsigned s[4];
    const int64_t base = (int64_t)blockIdx.x * CMP_TILE;
    unsigned c = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        c += (i < n && mask[i]) ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// exclusive scan of the tile counts in place (one workgroup); total -> *total_out
__global__ __launch_bounds__(1024) void compact_scan_kernel(unsigned *__restrict__ tile_cnt, int ntiles, unsigned *__restrict__ total_out)
{
    __shared__ unsigned s_w[16];
    __shared__ unsigned s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int b = 0; b < ntiles; b += 1024) {
        const int i = b + threadIdx.x;
        const unsigned v = i < ntiles ? tile_cnt[i] : 0u;
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
        if (i < ntiles) tile_cnt[i] = pre + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = pre + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = s_carry;
}

__global__ __launch_bounds__(256) void compact_write_kernel(const float *__restrict__ rows, const unsigned *__restrict__ orig,
                                                            const uint8_t *__restrict__ mask, int64_t n,
                                                            const unsigned *__restrict__ tile_off, float *__restrict__ rows_out,
                                                            unsigned *__restrict__ orig_out)
{
    __shared__ unsigned s_w[4];
    const int64_t base = (int64_t)blockIdx.x * CMP_TILE;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned run = tile_off[blockIdx.x];
    // element order inside the tile: u-major (u * 256 + thread), so the output keeps the input order
    for (int u = 0; u < 8; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        const bool keep = i < n && mask[i];
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_w[wv] = (unsigned)__builtin_popcountll(bal);
        __syncthreads();
        unsigned pos = run + (unsigned)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
        for (int w = 0; w < wv; ++w) pos += s_w[w];
        if (keep) {
            rows_out[3 * (size_t)pos + 0] = rows[3 * i + 0];
            rows_out[3 * (size_t)pos + 1] = rows[3 * i + 1];
            rows_out[3 * (size_t)pos + 2] = rows[3 * i + 2];
            orig_out[pos] = orig ? orig[i] : (unsigned)i;
        }
        run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}

// The two O(N) row filters that precede density / SOR in the reference's orchestrator (converter.py:196-203), as masks over
// the device-resident rows -- SURVEY.md 8(f) rank 4.  Both compare in f64: a float32 compared with a float32-rounded bound
// gives the same answer in either width, and numpy itself promotes to f64 when the bound is a np.float64 scalar (the
// alpha filter's logit threshold, data_processor.py:207-210).
__global__ __launch_bounds__(256) void mask_bbox_kernel(const float *__restrict__ rows, int64_t n, double lox, double loy, double loz,
                                                        double hix, double hiy, double hiz, uint8_t *__restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double x = rows[3 * i], y = rows[3 * i + 1], z = rows[3 * i + 2];
    mask[i] = (x >= lox) & (x <= hix) & (y >= loy) & (y <= hiy) & (z >= loz) & (z <= hiz);   // data_processor.py:217-224
}

__global__ __launch_bounds__(256) void mask_ge_kernel(const float *__restrict__ vals, const unsigned *__restrict__ orig, int64_t n,
                                                      double thr, uint8_t *__restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    mask[i] = (double)vals[orig ? orig[i] : (unsigned)i] >= thr;   // data_processor.py:210
}

// data_processor.py:316-343 (_compute_rgb_from_sh) for one channel: u8((clip(0.5 + f_dc * C0, 0, 1) ** (1/2.2)) * 255).
// The linear part is numpy's float32 arithmetic exactly (weak Python scalars: C0 and the exponent are rounded to float32);
// np.power on float32 is a libm / SVML routine whose bits a device cannot reproduce, so -- as in csrc/sog.hip -- the power
// is evaluated in float64, numpy's possible float32 result is bracketed by +-3 ulp, both ends go through the float32
// `* 255` and the truncation, and the element is flagged for the host when they disagree (about 1e-4 of the values).
__device__ __forceinline__ float chain_ulp_step(float a, int steps)
{
    int b = (int)__float_as_uint(a);
    b = b < 0 ? (int)0x80000000u - b : b;
    b += steps;
    b = b < 0 ? (int)0x80000000u - b : b;
    return __uint_as_float((unsigned)b);
}

// LIST: the uncertain elements as a compact list of their indices (wave-aggregated append; *count keeps counting past cap) instead of a
// flag byte per element -- round 6: the 30 MB of flags of a 10M-splat table cost 2 ms to bring back and 6.5 ms to scan on the host
template <bool LIST>
__global__ __launch_bounds__(256) void rgb_from_sh_kernel(const float *__restrict__ f_dc, int64_t n, uint8_t *__restrict__ out,
                                                          uint8_t *__restrict__ uncertain, unsigned *__restrict__ list, unsigned cap,
                                                          unsigned *__restrict__ count)
{
    const float c0 = 0.28209479177387814f;            // np.float32(SH_C0)
    const double e = (double)(float)(1.0 / 2.2);       // the exponent numpy uses: float32(1.0 / 2.2)
    const int64_t span = LIST ? ((n + 255) / 256) * 256 : n;   // (LIST: whole waves stay together for the ballot)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < span; i += (int64_t)gridDim.x * 256) {
        if (LIST && i >= n) {
            (void)__ballot(false);
            continue;
        }
        const float v = f_dc[i];
        float lin = __fadd_rn(0.5f, __fmul_rn(v, c0));
        lin = fminf(fmaxf(lin, 0.0f), 1.0f);           // np.clip (NaN propagates: flagged below)
        bool ok = v == v;
        unsigned q[2];
        if (lin == 0.0f || lin == 1.0f) {              // pow(0, e) = 0 and pow(1, e) = 1 exactly in any libm
            q[0] = q[1] = lin == 0.0f ? 0u : 255u;
        } else {
            const float a = (float)::pow((double)lin, e);
            // numpy's float32 power: libm powf (<= 1 ulp, measured 0.999) or, on AVX-512 builds, SVML whose low-accuracy
            // variants are bounded at 4 ulp -- bracket +-5 like the log of sog.hip
            const float b[2] = {chain_ulp_step(a, -5), chain_ulp_step(a, 5)};
#pragma unroll
            for (int s = 0; s < 2; ++s) q[s] = (unsigned)__fmul_rn(fminf(b[s], 1.0f), 255.0f);
        }
        ok = ok && q[0] == q[1];
        out[i] = (uint8_t)q[0];
        if constexpr (LIST) {
            const unsigned long long m = __ballot(!ok);
            if (m != 0ull) {
                const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(m));
                base = (unsigned)__shfl((int)base, leader);
                if (!ok) {
                    const unsigned pslot = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                    if (pslot < cap) list[pslot] = (unsigned)i;
                }
            }
        } else {
            uncertain[i] = ok ? 0 : 1;
        }
    }
}

}  // namespace gsx

using namespace gsx;

extern "C" int gsx_rgb_from_sh_dev(gsx_ctx *c, const float *f_dc_dev, int64_t n, uint8_t *out_dev, uint8_t *uncertain_dev)
{
    if (!c || (n > 0 && (!f_dc_dev || !out_dev || !uncertain_dev))) GSX_FAIL("gsx_rgb_from_sh_dev: null argument");
    GSX_HIP(hipSetDevice(c->device));
    if (n <= 0) return 0;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 1024), (int64_t)c->num_cu * 8));
    hipLaunchKernelGGL(rgb_from_sh_kernel<false>, dim3(blocks), dim3(256), 0, c->stream, f_dc_dev, n, out_dev, uncertain_dev, nullptr, 0u, nullptr);
    GSX_HIP(hipGetLastError());
    return 0;
}

extern "C" int gsx_rgb_from_sh_list_dev(gsx_ctx *c, const float *f_dc_dev, int64_t n, uint8_t *out_dev, uint32_t *list_dev, int64_t cap,
                                        uint32_t *count_dev)
{
    if (!c || !count_dev || (n > 0 && (!f_dc_dev || !out_dev)) || (cap > 0 && !list_dev)) GSX_FAIL("gsx_rgb_from_sh_list_dev: null argument");
    if (n < 0 || n >= (1LL << 32) || cap < 0 || cap > 0xffffffffLL) GSX_FAIL("gsx_rgb_from_sh_list_dev: bad size");
    GSX_HIP(hipSetDevice(c->device));
    GSX_HIP(hipMemsetAsync(count_dev, 0, 4, c->stream));
    if (n == 0) return 0;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 1024), (int64_t)c->num_cu * 8));
    hipLaunchKernelGGL(rgb_from_sh_kernel<true>, dim3(blocks), dim3(256), 0, c->stream, f_dc_dev, n, out_dev, nullptr, list_dev, (unsigned)cap, count_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

extern "C" int gsx_mask_bbox_dev(gsx_ctx *c, const float *rows_dev, int64_t n, const double *bounds6, uint8_t *mask_dev)
{
    if (!c || !bounds6 || (n > 0 && (!rows_dev || !mask_dev))) GSX_FAIL("gsx_mask_bbox_dev: null argument");
    if (n < 0 || n >= (1LL << 32)) GSX_FAIL("gsx_mask_bbox_dev: n out of range");
    GSX_HIP(hipSetDevice(c->device));
    if (n == 0) return 0;
    hipLaunchKernelGGL(mask_bbox_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, c->stream, rows_dev, n, bounds6[0], bounds6[1],
                       bounds6[2], bounds6[3], bounds6[4], bounds6[5], mask_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

extern "C" int gsx_mask_ge_dev(gsx_ctx *c, const float *vals_dev, const uint32_t *orig_dev, int64_t n, double threshold,
                               uint8_t *mask_dev)
{
    if (!c || (n > 0 && (!vals_dev || !mask_dev))) GSX_FAIL("gsx_mask_ge_dev: null argument");
    if (n < 0 || n >= (1LL << 32)) GSX_FAIL("gsx_mask_ge_dev: n out of range");
    GSX_HIP(hipSetDevice(c->device));
    if (n == 0) return 0;
    hipLaunchKernelGGL(mask_ge_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, c->stream, vals_dev, orig_dev, n, threshold, mask_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

extern "C" int gsx_compact_rows_dev(gsx_ctx *c, const float *rows_dev, const uint32_t *orig_dev, const uint8_t *mask_dev, int64_t n,
                                    float *rows_out_dev, uint32_t *orig_out_dev, int64_t *n_out)
{
    if (!c || !rows_dev || !mask_dev || !rows_out_dev || !orig_out_dev || !n_out) GSX_FAIL("gsx_compact_rows_dev: null argument");
    if (n < 0 || n >= (1LL << 32)) GSX_FAIL("gsx_compact_rows_dev: n out of range");
    GSX_HIP(hipSetDevice(c->device));
    *n_out = 0;
    if (n == 0) return 0;
    const int ntiles = div_up(n, CMP_TILE);
    GSX_CHECK(c->statspart.reserve(sizeof(unsigned) * ((size_t)ntiles + 4)));
    unsigned *tiles = c->statspart.as<unsigned>();
    unsigned *total = tiles + ntiles;
    hipLaunchKernelGGL(compact_count_kernel, dim3(ntiles), dim3(256), 0, c->stream, mask_dev, n, tiles);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, c->stream, tiles, ntiles, total);
    hipLaunchKernelGGL(compact_write_kernel, dim3(ntiles), dim3(256), 0, c->stream, rows_dev, orig_dev, mask_dev, n, tiles, rows_out_dev,
                       orig_out_dev);
    GSX_HIP(hipGetLastError());
    unsigned h = 0;
    GSX_HIP(hipMemcpyAsync(&h, total, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    *n_out = h;
    return 0;
}
