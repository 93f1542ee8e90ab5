// sog_table.hip -- the SOG writer's numeric core on a DEVICE-RESIDENT splat table.
//
// The reference's SogFormat.write (formats/sog.py:249-600) is straight-line numpy over the structured table:
//   :264-265  indices = np.lexsort((z, y, x)); data_s = data[indices]       (a 2.5 GB fancy-index copy at 10M splats)
//   :279-312  log-transformed u16 positions -> means_l / means_u texels
//   :315-386  quaternion column_stack + smallest-three packing -> quats texels
//   :392-431  concatenate(scale_0..2), 50 000-sample 1-D codebook, quantise -> scales texels
//   :433-459  the same for f_dc_0..2 + sigmoid(opacity) -> sh0 texels
//   :461-552  SH band detection, 45-column column_stack, chunked palette K-Means -> labels texels
// Here the raw rows are uploaded ONCE and everything between the table and the RGBA texel arrays happens in HBM:
//   sog_scan      one pass over the rows in table order: float32 sort keys of x, y, z, their extremes, which f_rest_i hold a
//                 non-zero value (the band detection of :476-486)
//   sog_order     the lexsort as three stable radix passes over the key columns (rocPRIM)
//   sog_gather    `data[indices]` as ONE pass: a wave reads each permuted 248-byte row whole, the tile leaves LDS as the
//                 sorted columns the stages below read and as the (n, D) row-major SH matrix the K-Means kernels take
//   sog_*_texels  positions / quaternions / codebook indices / alpha / palette labels written straight into the 4-byte
//                 texel layout of the WebP images (padding texels included); the values next to a rounding boundary of
//                 numpy's float32 log / exp go into a COMPACT list of (index * 4 + channel, value) pairs the host evaluates with numpy
//                 (sog_math.h has the certificate) instead of a flag per element
// Only texels (4 B per splat and image) and the short lists travel back.  All kernels are HBM-bound streaming passes;
// algorithmic bytes per splat: scan R + 12 (R = row bytes), order 3 x ~40, gather R + 4 + 56 + 4 D, texels 12 + 8 / 16 + 4 /
// 12 + 4 + 4 / 4 + 4.
#include <algorithm>

#include <rocprim/device/device_radix_sort.hpp>

#include "gsx_common.h"
#include "sog_math.h"

namespace gsx {

constexpr int SOGT_ROWS = 128;        // rows per LDS tile (31 KiB of 248-byte rows: five workgroups per CU)
constexpr int SOGT_FIELDS = GSX_SOG_FIELDS;
constexpr int SOGT_MAX_ROW_DWORDS = 128;   // 512-byte rows: a tile is 64 KiB of LDS

struct SogLayoutDev {
    int row_dwords, n_rest;
    int off[SOGT_FIELDS];   // dword offset of x y z | rot_0..3 | scale_0..2 | f_dc_0..2 | opacity | f_rest_0..44 inside a row
    int row_bytes;          // BYTES layouts (rows that are not a multiple of 4 bytes -- the table widened by three u1 colour fields, 251 bytes:
    int bytes;              //  data_processor.py:262-274 -- or fields at odd offsets): off[] are BYTE offsets, bytes = 1
};

// A field out of a tile in LDS.  Aligned layouts: `row` = dword index of the row, `off` = dword offset.  BYTES layouts: `row` = BYTE
// position of the row's first byte inside the tile (the tile starts at the 4-byte boundary below its first row), `off` = byte offset;
// the value straddles two dwords (v_alignbyte_b32; the tile has one spare dword behind its last row).
template <bool BYTES>
__device__ __forceinline__ unsigned tile_field(const unsigned *s, int row, int off)
{
    if constexpr (!BYTES) {
        return s[row + off];
    } else {
        const int b = row + off;
        return __builtin_amdgcn_alignbyte(s[(b >> 2) + 1], s[b >> 2], (unsigned)(b & 3));
    }
}

struct SogScanDev {
    unsigned kmin[3], kmax[3];
    unsigned pad[2];
    unsigned long long rest_nonzero;
};

__device__ __forceinline__ void load_tile(unsigned *s_rows, const unsigned *src, int total_dwords)
{
    // a tile of consecutive rows is one span of memory: 16 bytes per lane while the span allows it
    if (((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (total_dwords & 3) == 0) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(s_rows);
        for (int e = threadIdx.x; e < total_dwords / 4; e += 256) d4[e] = s4[e];
    } else {
        for (int e = threadIdx.x; e < total_dwords; e += 256) s_rows[e] = src[e];
    }
}

// table order: keys of x, y, z (three columns of n), extremes of the keys, non-zero mask of the f_rest columns
template <bool BYTES>
__global__ __launch_bounds__(256) void sog_scan_kernel(const unsigned *__restrict__ rows, SogLayoutDev L, int64_t n,
                                                       unsigned *__restrict__ keys, SogScanDev *__restrict__ out)
{
    extern __shared__ unsigned s_rows[];
    __shared__ int s_off[SOGT_FIELDS];
    if (threadIdx.x < SOGT_FIELDS) s_off[threadIdx.x] = L.off[threadIdx.x];
    unsigned kmin[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, kmax[3] = {0u, 0u, 0u};
    unsigned long long nz = 0ull;
    const int64_t ntiles = (n + SOGT_ROWS - 1) / SOGT_ROWS;
    const int rd = L.row_dwords;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * SOGT_ROWS;
        const int rows_here = (int)min((int64_t)SOGT_ROWS, n - r0);
        __syncthreads();   // the previous tile has been read
        int shift = 0;         // BYTES: the tile's first row starts `shift` bytes behind a 4-byte boundary
        if constexpr (BYTES) {
            const int64_t g0 = r0 * L.row_bytes;
            shift = (int)(g0 & 3);
            load_tile(s_rows, reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(rows) + (g0 - shift)),
                      (shift + rows_here * L.row_bytes + 3) >> 2);
        } else {
            load_tile(s_rows, rows + r0 * rd, rows_here * rd);
        }
        __syncthreads();
        const int r = threadIdx.x & (SOGT_ROWS - 1), half = threadIdx.x >> 7;
        if (r < rows_here) {
            const int rw = BYTES ? shift + r * L.row_bytes : r * rd;
            if (half == 0) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const unsigned key = sort_key(__uint_as_float(tile_field<BYTES>(s_rows, rw, s_off[a])));
                    keys[(int64_t)a * n + r0 + r] = key;
                    kmin[a] = min(kmin[a], key);
                    kmax[a] = max(kmax[a], key);
                }
            }
            for (int f = half; f < L.n_rest; f += 2)   // `data_s[fn] != 0` (:484): +-0.0 are zero, a NaN is not
                if ((tile_field<BYTES>(s_rows, rw, s_off[14 + f]) << 1) != 0u) nz |= 1ull << f;
        }
    }
    // wave reduction, one atomic per wave and word
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            kmin[a] = min(kmin[a], (unsigned)__shfl_xor((int)kmin[a], o));
            kmax[a] = max(kmax[a], (unsigned)__shfl_xor((int)kmax[a], o));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nz |= (unsigned long long)__shfl_xor((long long)nz, o);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&out->kmin[a], kmin[a]);
            atomicMax(&out->kmax[a], kmax[a]);
        }
        if (nz) atomicOr(&out->rest_nonzero, nz);
    }
}

__global__ void sog_scan_init_kernel(SogScanDev *out)
{
    if (threadIdx.x < 3) {
        out->kmin[threadIdx.x] = 0xffffffffu;
        out->kmax[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0) out->rest_nonzero = 0ull;
}

// values within a threshold of an axis' extremes, as compact lists: list 2a = v <= lo[a], list 2a + 1 = v >= hi[a]
struct SogExtremeArgs {
    float lo[3], hi[3];
};
__global__ __launch_bounds__(256) void sog_extremes_kernel(const unsigned *__restrict__ keys, int64_t n, SogExtremeArgs t, int cap,
                                                           float *__restrict__ vals /* [6][cap] */, unsigned *__restrict__ counts /* [6] */)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = sort_unkey(keys[(int64_t)a * n + i]);
            if (v <= t.lo[a]) {
                const unsigned p = atomicAdd(&counts[2 * a], 1u);
                if (p < (unsigned)cap) vals[(int64_t)(2 * a) * cap + p] = v;
            }
            if (v >= t.hi[a]) {
                const unsigned p = atomicAdd(&counts[2 * a + 1], 1u);
                if (p < (unsigned)cap) vals[(int64_t)(2 * a + 1) * cap + p] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void sog_order_keys_kernel(const unsigned *__restrict__ col, const unsigned *__restrict__ perm /* null: identity */,
                                                             int64_t n, unsigned *__restrict__ keys, unsigned *__restrict__ vals)
{
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
        const unsigned i = perm ? perm[j] : (unsigned)j;
        keys[j] = col[i];
        vals[j] = i;
    }
}

// `data_s = data[indices]` (:265) for the columns the writer reads: output row j = table row perm[j]
template <int D, bool BYTES>
__global__ __launch_bounds__(256) void sog_gather_kernel(const unsigned *__restrict__ rows, SogLayoutDev L, const unsigned *__restrict__ perm,
                                                         int64_t n, float *__restrict__ pos, float4 *__restrict__ rot, float *__restrict__ scale,
                                                         float *__restrict__ dc, float *__restrict__ opacity, float *__restrict__ sh)
{
    extern __shared__ unsigned s_rows[];
    __shared__ int s_off[SOGT_FIELDS];
    if (threadIdx.x < SOGT_FIELDS) s_off[threadIdx.x] = L.off[threadIdx.x];
    // (BYTES: a row of the tile occupies `rd` dwords -- the row, up to 3 bytes in front of it back to a 4-byte boundary, rounded up, an ODD
    //  count so that the lanes' rows fall into different LDS banks -- and s_shift[r] is where inside them the row starts)
    __shared__ unsigned char s_shift[SOGT_ROWS];
    const int rd = L.row_dwords;
    const int64_t j0 = (int64_t)blockIdx.x * SOGT_ROWS;
    const int rows_here = (int)min((int64_t)SOGT_ROWS, n - j0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // a wave reads one permuted row per instruction: 248 contiguous bytes
    for (int r = wave; r < rows_here; r += 4) {
        const unsigned srow = __builtin_amdgcn_readfirstlane(perm[j0 + r]);
        if constexpr (BYTES) {
            const int64_t g = (int64_t)srow * L.row_bytes;
            const int sh_b = (int)(g & 3), nd = (sh_b + L.row_bytes + 3) >> 2;
            const unsigned *src = reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(rows) + (g - sh_b));
            for (int e = lane; e < nd; e += 64) s_rows[r * rd + e] = src[e];
            if (lane == 0) s_shift[r] = (unsigned char)sh_b;
        } else {
            const unsigned *src = rows + (int64_t)srow * rd;
            for (int e = lane; e < rd; e += 64) s_rows[r * rd + e] = src[e];
        }
    }
    __syncthreads();
    const int r = threadIdx.x & (SOGT_ROWS - 1), half = threadIdx.x >> 7;
    if (r < rows_here) {
        const int rw = BYTES ? r * rd * 4 + s_shift[r] : r * rd;
        const int64_t j = j0 + r;
        if (half == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) pos[(int64_t)a * n + j] = __uint_as_float(tile_field<BYTES>(s_rows, rw, s_off[a]));
            rot[j] = make_float4(__uint_as_float(tile_field<BYTES>(s_rows, rw, s_off[3])), __uint_as_float(tile_field<BYTES>(s_rows, rw, s_off[4])),
                                 __uint_as_float(tile_field<BYTES>(s_rows, rw, s_off[5])), __uint_as_float(tile_field<BYTES>(s_rows, rw, s_off[6])));
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                scale[(int64_t)a * n + j] = __uint_as_float(tile_field<BYTES>(s_rows, rw, s_off[7 + a]));
                dc[(int64_t)a * n + j] = __uint_as_float(tile_field<BYTES>(s_rows, rw, s_off[10 + a]));
            }
            opacity[j] = __uint_as_float(tile_field<BYTES>(s_rows, rw, s_off[13]));
        }
    }
    if constexpr (D > 0) {
        // the tile's SH rows are one span of the (n, D) matrix
        float *dst = sh + j0 * D;
        for (int e = threadIdx.x; e < rows_here * D; e += 256) {
            const int rr = e / D, c = e - rr * D;
            dst[e] = __uint_as_float(tile_field<BYTES>(s_rows, BYTES ? rr * rd * 4 + s_shift[rr] : rr * rd, s_off[14 + c]));
        }
    }
}

// wave-aggregated append of (tag, value bits) pairs for the lanes with `flag`; entries beyond cap are dropped, the counter keeps counting
__device__ __forceinline__ void list_append(bool flag, unsigned tag, float value, uint2 *__restrict__ list, unsigned cap, unsigned *__restrict__ count)
{
    const unsigned long long m = __ballot(flag);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(m));
    base = (unsigned)__shfl((int)base, leader);
    if (flag) {
        const unsigned p = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (p < cap) list[p] = make_uint2(tag, __float_as_uint(value));
    }
}

struct SogMeansArgs {
    float mn[3], mx[3];
    float v_mn[3], v_mx[3];   // the coordinate values whose transform IS mn / mx (numpy evaluated them): texels 0 and 65535, certain
};
// :279-312: texel i of means_l / means_u = (low / high byte of the u16 x, y, z, 255); padding texels 255
__global__ __launch_bounds__(256) void sog_means_texels_kernel(const float *__restrict__ pos, int64_t n, int64_t texels, SogMeansArgs a,
                                                               uchar4 *__restrict__ lo, uchar4 *__restrict__ hi, uint2 *__restrict__ list,
                                                               unsigned cap, unsigned *__restrict__ count)
{
    const float range[3] = {__fsub_rn(a.mx[0], a.mn[0]), __fsub_rn(a.mx[1], a.mn[1]), __fsub_rn(a.mx[2], a.mn[2])};
    const int64_t span = ((texels + 255) / 256) * 256;   // whole waves take part in the ballots
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < span; i += (int64_t)gridDim.x * 256) {
        unsigned q[3] = {65535u, 65535u, 65535u};
        bool ok[3] = {true, true, true};
        float v[3] = {0.0f, 0.0f, 0.0f};
        if (i < n) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                v[c] = pos[(int64_t)c * n + i];
                q[c] = sog_position_texel(v[c], a.mn[c], range[c], &ok[c]);
                // (l - mn) / range is exactly 0 / exactly 1 for the values numpy's min / max came from; their bracket straddles
                // the clip at 65535, which would put every copy of a repeated maximum on the list
                if (range[c] > 0.0f && v[c] == a.v_mn[c]) {
                    q[c] = 0u;
                    ok[c] = true;
                } else if (range[c] > 0.0f && v[c] == a.v_mx[c]) {
                    q[c] = 65535u;
                    ok[c] = true;
                }
            }
        }
        if (i < texels) {
            lo[i] = make_uchar4((unsigned char)(q[0] & 0xff), (unsigned char)(q[1] & 0xff), (unsigned char)(q[2] & 0xff), 255);
            hi[i] = make_uchar4((unsigned char)(q[0] >> 8), (unsigned char)(q[1] >> 8), (unsigned char)(q[2] >> 8), 255);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) list_append(!ok[c], (unsigned)i * 4u + (unsigned)c, v[c], list, cap, count);
    }
}

// :315-386; padding texels 255
__global__ __launch_bounds__(256) void sog_quats_texels_kernel(const float4 *__restrict__ rot, int64_t n, int64_t texels, uchar4 *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < texels; i += (int64_t)gridDim.x * 256)
        out[i] = i < n ? sog_quat_pack(rot[i]) : make_uchar4(255, 255, 255, 255);
}

// :408-431 / :446-459: texel i = (codebook index of the three columns, 255 or the sigmoid of the opacity); padding texels 0
__global__ __launch_bounds__(256) void sog_codes_texels_kernel(const float *__restrict__ cols, int64_t n, int64_t texels, const float *__restrict__ cb,
                                                               int kcb, const float *__restrict__ opacity /* null: alpha 255 */,
                                                               uchar4 *__restrict__ out, uint2 *__restrict__ list, unsigned cap,
                                                               unsigned *__restrict__ count)
{
    __shared__ float lcb[256];
    for (int i = threadIdx.x; i < kcb; i += 256) lcb[i] = cb[i];
    __syncthreads();
    const int64_t span = ((texels + 255) / 256) * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < span; i += (int64_t)gridDim.x * 256) {
        uchar4 t = make_uchar4(0, 0, 0, 0);
        bool ok = true;
        const float o = (opacity && i < n) ? opacity[i] : 0.0f;
        if (i < n) {
            // a codebook of ONE entry: `return np.zeros_like(vals, dtype=np.uint8)` (:410) -- what the search gives as well
            t.x = (unsigned char)sog_codebook_index(lcb, kcb, cols[i]);
            t.y = (unsigned char)sog_codebook_index(lcb, kcb, cols[n + i]);
            t.z = (unsigned char)sog_codebook_index(lcb, kcb, cols[2 * n + i]);
            t.w = opacity ? (unsigned char)sog_alpha_texel(o, &ok) : (unsigned char)255;
        }
        if (i < texels) out[i] = t;
        if (opacity) list_append(!ok, (unsigned)i * 4u + 3u, o, list, cap, count);
    }
}

// :546-552 + :600-606: palette label = chunk-local label + chunk * k -> (low byte, high byte, 0, 255); padding texels 0
__global__ __launch_bounds__(256) void sog_labels_texels_kernel(const int32_t *__restrict__ labels, int64_t n, int64_t texels, int64_t chunk_rows,
                                                                int k, uchar4 *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < texels; i += (int64_t)gridDim.x * 256) {
        uchar4 t = make_uchar4(0, 0, 0, 0);
        if (i < n) {
            const unsigned l = (unsigned)((int64_t)labels[i] + (i / chunk_rows) * k) & 0xffffu;   // .astype(np.uint16)
            t = make_uchar4((unsigned char)(l & 0xff), (unsigned char)(l >> 8), 0, 255);
        }
        out[i] = t;
    }
}

// dst row i = src row idx[i] (rows of row_floats float32): the codebooks' 50 000-sample (:397-400) and the palette's initial centroids
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src, int row_floats, const int64_t *__restrict__ idx, int64_t m,
                                                          float *__restrict__ dst)
{
    const int64_t total = m * row_floats;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e / row_floats;
        const int c = (int)(e - i * row_floats);
        dst[e] = src[idx[i] * row_floats + c];
    }
}

static int layout_to_dev(const gsx_sog_layout *l, int n_rest, SogLayoutDev *out, const char *who)
{
    if (!l) GSX_FAIL("%s: null layout", who);
    if (l->row_bytes < 4 || l->row_bytes > 4 * SOGT_MAX_ROW_DWORDS)
        GSX_FAIL("%s: rows of %lld bytes (4 ... %d are supported)", who, (long long)l->row_bytes, 4 * SOGT_MAX_ROW_DWORDS);
    if (n_rest < 0 || n_rest > 45) GSX_FAIL("%s: 0 <= n_rest <= 45", who);
    bool bytes = (l->row_bytes & 3) != 0;
    for (int f = 0; f < 14 + n_rest; ++f) {
        const int o = l->offset[f];
        if (o < 0 || o + 4 > l->row_bytes) GSX_FAIL("%s: field %d at byte offset %d of a %lld-byte row", who, f, o, (long long)l->row_bytes);
        bytes = bytes || (o & 3) != 0;
    }
    if (bytes && l->row_bytes > 500) GSX_FAIL("%s: rows of %lld bytes with fields off the 4-byte grid (up to 500 bytes are supported)", who, (long long)l->row_bytes);
    out->bytes = bytes ? 1 : 0;
    out->row_bytes = (int)l->row_bytes;
    // aligned: dwords per row.  BYTES: dwords a row occupies in the gather kernel's tile (see there), odd
    out->row_dwords = bytes ? (int)(((l->row_bytes + 6) >> 2) | 1) : (int)(l->row_bytes / 4);
    out->n_rest = n_rest;
    for (int f = 0; f < SOGT_FIELDS; ++f) out->off[f] = f < 14 + n_rest ? (bytes ? l->offset[f] : l->offset[f] / 4) : 0;
    return 0;
}

static inline int stream_blocks(const gsx_ctx *c, int64_t n)
{
    return (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 1024), (int64_t)c->num_cu * 8));
}

}  // namespace gsx

using namespace gsx;

extern "C" {

int gsx_sog_scan_dev(gsx_ctx *c, const void *rows_dev, const gsx_sog_layout *layout, int64_t n, uint32_t *keys3_dev, gsx_sog_scan *out)
{
    if (!c || !rows_dev || !keys3_dev || !out) GSX_FAIL("gsx_sog_scan_dev: null argument");
    if (n <= 0 || n >= (1LL << 30)) GSX_FAIL("gsx_sog_scan_dev: 1 <= n < 2^30");
    SogLayoutDev L;
    GSX_CHECK(layout_to_dev(layout, layout ? layout->n_rest : 0, &L, "gsx_sog_scan_dev"));
    if (reinterpret_cast<uintptr_t>(rows_dev) & 3) GSX_FAIL("gsx_sog_scan_dev: rows must be 4-byte aligned");
    GSX_HIP(hipSetDevice(c->device));
    GSX_CHECK(c->scratch3.reserve(256));
    SogScanDev *d = c->scratch3.as<SogScanDev>();
    hipLaunchKernelGGL(sog_scan_init_kernel, dim3(1), dim3(64), 0, c->stream, d);
    const int64_t ntiles = (n + SOGT_ROWS - 1) / SOGT_ROWS;
    const int blocks = (int)std::min<int64_t>(ntiles, (int64_t)c->num_cu * 5);
    if (L.bytes) {   // the tile = SOGT_ROWS consecutive rows from the 4-byte boundary below the first one, + the spare dword of tile_field
        const size_t lds = ((size_t)SOGT_ROWS * L.row_bytes + 3 + 3) / 4 * 4 + 4;
        hipLaunchKernelGGL(sog_scan_kernel<true>, dim3(blocks), dim3(256), lds, c->stream, reinterpret_cast<const unsigned *>(rows_dev), L, n, keys3_dev, d);
    } else {
        const size_t lds = sizeof(unsigned) * (size_t)SOGT_ROWS * L.row_dwords;
        hipLaunchKernelGGL(sog_scan_kernel<false>, dim3(blocks), dim3(256), lds, c->stream, reinterpret_cast<const unsigned *>(rows_dev), L, n, keys3_dev, d);
    }
    GSX_HIP(hipGetLastError());
    SogScanDev h;
    GSX_HIP(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    memset(out, 0, sizeof(*out));
    for (int a = 0; a < 3; ++a) {
        out->vmin[a] = sort_unkey(h.kmin[a]);
        out->vmax[a] = sort_unkey(h.kmax[a]);
        // NaN keys are the largest; +-inf are the extremes of the finite order
        if (h.kmax[a] == 0xffffffffu || !(fabsf(out->vmin[a]) <= 3.0e38f) || !(fabsf(out->vmax[a]) <= 3.0e38f)) out->nonfinite |= 1u << a;
    }
    out->rest_nonzero = h.rest_nonzero;
    return 0;
}

int gsx_sog_extremes_dev(gsx_ctx *c, const uint32_t *keys3_dev, int64_t n, const float *lo3, const float *hi3, int cap, float *vals_out,
                         int64_t *counts6_out)
{
    if (!c || !keys3_dev || !lo3 || !hi3 || !vals_out || !counts6_out) GSX_FAIL("gsx_sog_extremes_dev: null argument");
    if (n <= 0 || cap < 1 || cap > (1 << 24)) GSX_FAIL("gsx_sog_extremes_dev: bad size");
    GSX_HIP(hipSetDevice(c->device));
    const size_t list_bytes = sizeof(float) * 6 * (size_t)cap;
    GSX_CHECK(c->scratch3.reserve(256 + list_bytes));
    unsigned *counts = c->scratch3.as<unsigned>();
    float *vals = reinterpret_cast<float *>(c->scratch3.as<char>() + 256);
    GSX_HIP(hipMemsetAsync(counts, 0, 32, c->stream));
    SogExtremeArgs t;
    for (int a = 0; a < 3; ++a) {
        t.lo[a] = lo3[a];
        t.hi[a] = hi3[a];
    }
    hipLaunchKernelGGL(sog_extremes_kernel, dim3(stream_blocks(c, n)), dim3(256), 0, c->stream, keys3_dev, n, t, cap, vals, counts);
    GSX_HIP(hipGetLastError());
    unsigned hc[6];
    GSX_HIP(hipMemcpyAsync(hc, counts, sizeof(hc), hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    for (int l = 0; l < 6; ++l) {
        counts6_out[l] = hc[l];
        const size_t m = std::min<size_t>(hc[l], (size_t)cap);
        if (m) GSX_HIP(hipMemcpyAsync(vals_out + (size_t)l * cap, vals + (size_t)l * cap, sizeof(float) * m, hipMemcpyDeviceToHost, c->stream));
    }
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int gsx_sog_order_dev(gsx_ctx *c, const uint32_t *keys3_dev, int64_t n, uint32_t *perm_out_dev)
{
    if (!c || !keys3_dev || !perm_out_dev) GSX_FAIL("gsx_sog_order_dev: null argument");
    if (n <= 0 || n >= (1LL << 30)) GSX_FAIL("gsx_sog_order_dev: 1 <= n < 2^30");
    GSX_HIP(hipSetDevice(c->device));
    size_t temp_bytes = 0;
    unsigned *nul = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, temp_bytes, nul, nul, nul, nul, (size_t)n, 0, 32, c->stream) != hipSuccess)
        GSX_FAIL("gsx_sog_order_dev: rocprim size query failed");
    const size_t col = sizeof(unsigned) * (size_t)n;
    GSX_CHECK(c->scratch5.reserve(4 * col + temp_bytes + 256));
    unsigned *ka = c->scratch5.as<unsigned>(), *kb = ka + n, *va = kb + n, *vb = va + n;
    void *temp = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(vb + n) + 255) & ~(uintptr_t)255);
    const int blocks = stream_blocks(c, n);
    const unsigned *perm = nullptr;
    for (int pass = 0; pass < 3; ++pass) {   // np.lexsort((z, y, x)): z is the least significant key (:264)
        const unsigned *colkeys = keys3_dev + (size_t)(2 - pass) * n;
        hipLaunchKernelGGL(sog_order_keys_kernel, dim3(blocks), dim3(256), 0, c->stream, colkeys, perm, n, ka, va);
        unsigned *vo = pass == 2 ? perm_out_dev : vb;
        GSX_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, ka, kb, va, vo, (size_t)n, 0, 32, c->stream));   // stable
        perm = vo;
    }
    GSX_HIP(hipGetLastError());
    return 0;
}

int gsx_sog_gather_dev(gsx_ctx *c, const void *rows_dev, const gsx_sog_layout *layout, const uint32_t *perm_dev, int64_t n, int d_sh,
                       float *pos_dev, float *rot_dev, float *scale_dev, float *dc_dev, float *opacity_dev, float *sh_dev)
{
    if (!c || !rows_dev || !perm_dev || !pos_dev || !rot_dev || !scale_dev || !dc_dev || !opacity_dev) GSX_FAIL("gsx_sog_gather_dev: null argument");
    if (n <= 0 || n >= (1LL << 30)) GSX_FAIL("gsx_sog_gather_dev: 1 <= n < 2^30");
    if (d_sh != 0 && d_sh != 9 && d_sh != 24 && d_sh != 45) GSX_FAIL("gsx_sog_gather_dev: d_sh must be 0, 9, 24 or 45");
    if (d_sh > 0 && !sh_dev) GSX_FAIL("gsx_sog_gather_dev: null SH output");
    if (reinterpret_cast<uintptr_t>(rot_dev) & 15) GSX_FAIL("gsx_sog_gather_dev: the quaternion rows must be 16-byte aligned");
    SogLayoutDev L;
    GSX_CHECK(layout_to_dev(layout, d_sh, &L, "gsx_sog_gather_dev"));
    if (reinterpret_cast<uintptr_t>(rows_dev) & 3) GSX_FAIL("gsx_sog_gather_dev: rows must be 4-byte aligned");
    GSX_HIP(hipSetDevice(c->device));
    const unsigned ntiles = (unsigned)((n + SOGT_ROWS - 1) / SOGT_ROWS);
    const size_t lds = sizeof(unsigned) * ((size_t)SOGT_ROWS * L.row_dwords + 1);
    const unsigned *rows = reinterpret_cast<const unsigned *>(rows_dev);
    float4 *rot4 = reinterpret_cast<float4 *>(rot_dev);
#define GSX_SOG_GATHER(D)                                                                                                                     \
    do {                                                                                                                                      \
        if (L.bytes)                                                                                                                          \
            hipLaunchKernelGGL((sog_gather_kernel<D, true>), dim3(ntiles), dim3(256), lds, c->stream, rows, L, perm_dev, n, pos_dev, rot4,     \
                               scale_dev, dc_dev, opacity_dev, sh_dev);                                                                       \
        else                                                                                                                                  \
            hipLaunchKernelGGL((sog_gather_kernel<D, false>), dim3(ntiles), dim3(256), lds, c->stream, rows, L, perm_dev, n, pos_dev, rot4,    \
                               scale_dev, dc_dev, opacity_dev, sh_dev);                                                                       \
    } while (0)
    switch (d_sh) {
    case 0: GSX_SOG_GATHER(0); break;
    case 9: GSX_SOG_GATHER(9); break;
    case 24: GSX_SOG_GATHER(24); break;
    default: GSX_SOG_GATHER(45); break;
    }
#undef GSX_SOG_GATHER
    GSX_HIP(hipGetLastError());
    return 0;
}

static int texel_args(gsx_ctx *c, int64_t n, int64_t texels, const char *who)
{
    if (n < 0 || texels < n || texels >= (1LL << 30)) GSX_FAIL("%s: 0 <= n <= texels < 2^30", who);
    GSX_HIP(hipSetDevice(c->device));
    return 0;
}

int gsx_sog_means_texels_dev(gsx_ctx *c, const float *pos_dev, int64_t n, int64_t texels, const float *log_min3, const float *log_max3,
                             const float *arg_min3, const float *arg_max3, uint8_t *means_l_dev, uint8_t *means_u_dev, uint32_t *list_dev, int64_t cap, uint32_t *count_dev)
{
    if (!c || !pos_dev || !log_min3 || !log_max3 || !means_l_dev || !means_u_dev || !list_dev || !count_dev) GSX_FAIL("gsx_sog_means_texels_dev: null argument");
    if (cap < 0 || cap > 0xffffffffLL) GSX_FAIL("gsx_sog_means_texels_dev: bad list capacity");
    GSX_CHECK(texel_args(c, n, texels, "gsx_sog_means_texels_dev"));
    if (texels == 0) return 0;
    SogMeansArgs a;
    for (int k = 0; k < 3; ++k) {
        a.mn[k] = log_min3[k];
        a.mx[k] = log_max3[k];
        a.v_mn[k] = arg_min3 ? arg_min3[k] : __builtin_nanf("");   // a NaN equals nothing: no shortcut
        a.v_mx[k] = arg_max3 ? arg_max3[k] : __builtin_nanf("");
    }
    GSX_HIP(hipMemsetAsync(count_dev, 0, 4, c->stream));
    hipLaunchKernelGGL(sog_means_texels_kernel, dim3(stream_blocks(c, texels)), dim3(256), 0, c->stream, pos_dev, n, texels, a,
                       reinterpret_cast<uchar4 *>(means_l_dev), reinterpret_cast<uchar4 *>(means_u_dev), reinterpret_cast<uint2 *>(list_dev),
                       (unsigned)cap, count_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

int gsx_sog_quats_texels_dev(gsx_ctx *c, const float *rot_rows_dev, int64_t n, int64_t texels, uint8_t *out_dev)
{
    if (!c || !rot_rows_dev || !out_dev) GSX_FAIL("gsx_sog_quats_texels_dev: null argument");
    if (reinterpret_cast<uintptr_t>(rot_rows_dev) & 15) GSX_FAIL("gsx_sog_quats_texels_dev: rows must be 16-byte aligned");
    GSX_CHECK(texel_args(c, n, texels, "gsx_sog_quats_texels_dev"));
    if (texels == 0) return 0;
    hipLaunchKernelGGL(sog_quats_texels_kernel, dim3(stream_blocks(c, texels)), dim3(256), 0, c->stream,
                       reinterpret_cast<const float4 *>(rot_rows_dev), n, texels, reinterpret_cast<uchar4 *>(out_dev));
    GSX_HIP(hipGetLastError());
    return 0;
}

int gsx_sog_codes_texels_dev(gsx_ctx *c, const float *cols3_dev, int64_t n, int64_t texels, const float *codebook_dev, int kcb,
                             const float *opacity_dev, uint8_t *out_dev, uint32_t *list_dev, int64_t cap, uint32_t *count_dev)
{
    if (!c || !cols3_dev || !codebook_dev || !out_dev) GSX_FAIL("gsx_sog_codes_texels_dev: null argument");
    if (kcb < 1 || kcb > 256) GSX_FAIL("gsx_sog_codes_texels_dev: 1 <= codebook entries <= 256");
    if (opacity_dev && (!list_dev || !count_dev || cap < 0 || cap > 0xffffffffLL)) GSX_FAIL("gsx_sog_codes_texels_dev: the alpha channel needs its list");
    GSX_CHECK(texel_args(c, n, texels, "gsx_sog_codes_texels_dev"));
    if (texels == 0) return 0;
    if (opacity_dev) GSX_HIP(hipMemsetAsync(count_dev, 0, 4, c->stream));
    hipLaunchKernelGGL(sog_codes_texels_kernel, dim3(stream_blocks(c, texels)), dim3(256), 0, c->stream, cols3_dev, n, texels, codebook_dev, kcb,
                       opacity_dev, reinterpret_cast<uchar4 *>(out_dev), reinterpret_cast<uint2 *>(list_dev), (unsigned)(opacity_dev ? cap : 0), count_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

int gsx_sog_labels_texels_dev(gsx_ctx *c, const int32_t *labels_dev, int64_t n, int64_t texels, int64_t chunk_rows, int k, uint8_t *out_dev)
{
    if (!c || !labels_dev || !out_dev) GSX_FAIL("gsx_sog_labels_texels_dev: null argument");
    if (chunk_rows < 1 || k < 1) GSX_FAIL("gsx_sog_labels_texels_dev: bad chunking");
    GSX_CHECK(texel_args(c, n, texels, "gsx_sog_labels_texels_dev"));
    if (texels == 0) return 0;
    hipLaunchKernelGGL(sog_labels_texels_kernel, dim3(stream_blocks(c, texels)), dim3(256), 0, c->stream, labels_dev, n, texels, chunk_rows, k,
                       reinterpret_cast<uchar4 *>(out_dev));
    GSX_HIP(hipGetLastError());
    return 0;
}

int gsx_gather_rows_dev(gsx_ctx *c, const float *src_dev, int row_floats, const int64_t *idx_dev, int64_t m, float *dst_dev)
{
    if (!c || !src_dev || !idx_dev || !dst_dev) GSX_FAIL("gsx_gather_rows_dev: null argument");
    if (row_floats < 1 || m < 0) GSX_FAIL("gsx_gather_rows_dev: bad shape");
    GSX_HIP(hipSetDevice(c->device));
    if (m == 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(stream_blocks(c, m * row_floats)), dim3(256), 0, c->stream, src_dev, row_floats, idx_dev, m, dst_dev);
    GSX_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
