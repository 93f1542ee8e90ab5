// sog.hip -- numeric core of the SOG writer next to the K-Means codebooks.
//
//   formats/sog.py:264      indices = np.lexsort((z, y, x))        -> gsx_lexsort3: three stable radix-sort passes
//   formats/sog.py:315-386  quaternion normalise + smallest-three  -> gsx_sog_quats: one elementwise pass, byte-exact
//
//   formats/sog.py:279-309  sign(v) log(|v| + 1) -> u16 texels       -> gsx_sog_positions: float64 log + rounding certificate
//   formats/sog.py:457-459  255 / (1 + exp(-opacity)) -> u8 texels   -> gsx_sog_alpha: float64 exp + rounding certificate
//
// The last two go through numpy's float32 `log` / `exp`, SIMD routines up to ~4 ulp off the correctly rounded value
// (measured in round 2: 16 % / 39 % of the values differ from it), so a device log/exp cannot reproduce their BITS.  But the
// outputs are u16 / u8 quantised and every step after the transcendental is a monotone float32 operation: the kernels
// evaluate the transcendental in float64, bracket numpy's possible float32 result (+-5 ulp for log, +-4 for exp),
// push BOTH ends of the bracket through numpy's exact float32 sequence (sub, div, mul by 65535, clip, truncate) and emit
// the texel when the two agree -- otherwise the element is flagged and the host evaluates numpy's own expression for it
// (~1.4 % of the positions of a typical scene, 1e-4 of the opacities).  Byte-identical textures.
// HBM-bound: 16 B in, 4 B out per splat (quats); 3 x (8 B + 8 B) per splat (sort); 4 B in, 3 B out (positions), 4 in 2 out (alpha).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "gsx_common.h"
#include "sog_math.h"

namespace gsx {

__global__ __launch_bounds__(256) void lexsort_keys_kernel(const float *__restrict__ col, int64_t stride,
                                                           const unsigned *__restrict__ perm /* null: identity */, int64_t n,
                                                           unsigned *__restrict__ keys, unsigned *__restrict__ vals)
{
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
        const unsigned i = perm ? perm[j] : (unsigned)j;
        keys[j] = sort_key(col[(int64_t)i * stride]);
        vals[j] = i;
    }
}

// sog.py:315-386.  rot: (n,4) float32 rows (rot_0..rot_3); out: 4 bytes per splat (c0, c1, c2, 252 + max_idx)
__global__ __launch_bounds__(256) void sog_quats_kernel(const float *__restrict__ rot, int64_t n, uchar4 *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = sog_quat_pack(reinterpret_cast<const float4 *>(rot)[i]);
}

// sog.py:279-309 for one axis (sog_math.h: sog_position_texel)
__global__ __launch_bounds__(256) void sog_positions_kernel(const float *__restrict__ v, int64_t n, float mn, float mx,
                                                            uint16_t *__restrict__ out, uint8_t *__restrict__ uncertain)
{
    const float range = __fsub_rn(mx, mn);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        bool ok;
        out[i] = (uint16_t)sog_position_texel(v[i], mn, range, &ok);
        uncertain[i] = ok ? 0 : 1;
    }
}

// sog.py:457-459: 1 / (1 + exp(-o)) * 255 -> clip -> u8 (sog_math.h: sog_alpha_texel)
__global__ __launch_bounds__(256) void sog_alpha_kernel(const float *__restrict__ o, int64_t n, uint8_t *__restrict__ out,
                                                        uint8_t *__restrict__ uncertain)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        bool ok;
        out[i] = (uint8_t)sog_alpha_texel(o[i], &ok);
        uncertain[i] = ok ? 0 : 1;
    }
}

static int lexsort3_dev(gsx_ctx *c, const float *k0, const float *k1, const float *k2, int64_t stride, int64_t n, uint32_t *perm_out)
{
    // buffers: keys A/B, vals A/B (4 x n x u32) + rocprim temporary storage
    size_t temp_bytes = 0;
    unsigned *nul = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, temp_bytes, nul, nul, nul, nul, (size_t)n, 0, 32, c->stream) != hipSuccess)
        GSX_FAIL("lexsort: rocprim size query failed");
    const size_t col = sizeof(unsigned) * (size_t)n;
    GSX_CHECK(c->scratch5.reserve(4 * col + temp_bytes + 256));
    unsigned *ka = c->scratch5.as<unsigned>(), *kb = ka + n, *va = kb + n, *vb = va + n;
    void *temp = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(vb + n) + 255) & ~(uintptr_t)255);
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 1024), (int64_t)c->num_cu * 8));
    const float *cols[3] = {k0, k1, k2};   // least significant first, like np.lexsort's argument order
    const unsigned *perm = nullptr;
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(lexsort_keys_kernel, dim3(blocks), dim3(256), 0, c->stream, cols[pass], stride, perm, n, ka, va);
        unsigned *vo = pass == 2 ? perm_out : vb;
        GSX_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, ka, kb, va, vo, (size_t)n, 0, 32, c->stream));   // stable
        perm = vo;
    }
    GSX_HIP(hipGetLastError());
    return 0;
}

}  // namespace gsx

using namespace gsx;

extern "C" {

int gsx_lexsort3_dev(gsx_ctx *c, const float *k0, const float *k1, const float *k2, int64_t stride, int64_t n, uint32_t *perm_out_dev)
{
    if (!c || !k0 || !k1 || !k2 || !perm_out_dev) GSX_FAIL("gsx_lexsort3_dev: null argument");
    if (n < 0 || n >= (1LL << 32) || stride < 1) GSX_FAIL("gsx_lexsort3_dev: bad size");
    GSX_HIP(hipSetDevice(c->device));
    if (n == 0) return 0;
    return lexsort3_dev(c, k0, k1, k2, stride, n, perm_out_dev);
}

int gsx_sog_quats_dev(gsx_ctx *c, const float *rot_rows_dev, int64_t n, uint8_t *out4_dev)
{
    if (!c || !rot_rows_dev || !out4_dev) GSX_FAIL("gsx_sog_quats_dev: null argument");
    if ((reinterpret_cast<uintptr_t>(rot_rows_dev) & 15) || (reinterpret_cast<uintptr_t>(out4_dev) & 3))
        GSX_FAIL("gsx_sog_quats_dev: rows must be 16-byte and output 4-byte aligned");
    GSX_HIP(hipSetDevice(c->device));
    if (n <= 0) return 0;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 1024), (int64_t)c->num_cu * 8));
    hipLaunchKernelGGL(sog_quats_kernel, dim3(blocks), dim3(256), 0, c->stream, rot_rows_dev, n, reinterpret_cast<uchar4 *>(out4_dev));
    GSX_HIP(hipGetLastError());
    return 0;
}

int gsx_sog_positions_dev(gsx_ctx *c, const float *v_dev, int64_t n, float log_min, float log_max, uint16_t *out_dev, uint8_t *uncertain_dev)
{
    if (!c || !v_dev || !out_dev || !uncertain_dev) GSX_FAIL("gsx_sog_positions_dev: null argument");
    GSX_HIP(hipSetDevice(c->device));
    if (n <= 0) return 0;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 1024), (int64_t)c->num_cu * 8));
    hipLaunchKernelGGL(sog_positions_kernel, dim3(blocks), dim3(256), 0, c->stream, v_dev, n, log_min, log_max, out_dev, uncertain_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

int gsx_sog_alpha_dev(gsx_ctx *c, const float *opacity_dev, int64_t n, uint8_t *out_dev, uint8_t *uncertain_dev)
{
    if (!c || !opacity_dev || !out_dev || !uncertain_dev) GSX_FAIL("gsx_sog_alpha_dev: null argument");
    GSX_HIP(hipSetDevice(c->device));
    if (n <= 0) return 0;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 1024), (int64_t)c->num_cu * 8));
    hipLaunchKernelGGL(sog_alpha_kernel, dim3(blocks), dim3(256), 0, c->stream, opacity_dev, n, out_dev, uncertain_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
