// density.hip -- voxel occupancy + per-point membership of the voxel-density filter.
//
// Replaces the O(N) parts of DataProcessor.apply_density_filter:
//   data_processor.py:38-39   keys = floor(f32 xyz / voxel_size)            -> voxel_key()
//   data_processor.py:43      np.unique(keys, axis=0, return_counts=True)   -> voxel_count_kernel
//   data_processor.py:48-52   dense = counts >= min_points                  -> voxel_collect_kernel
//   data_processor.py:111-114 mask = voxel(point) in kept clusters          -> voxel_mask_kernel
// The 6-connected BFS over the (<= ~1000) dense voxels stays on the host
// (3dgsconverter_amd/processing/clusters.py), as SURVEY.md 7.6 plans.
//
// Only counts per voxel and membership reach the mask, never np.unique's order, so the
// lexicographic sort (the reference's O(N log N) hot spot) is replaced by an open-addressing
// hash table in HBM keyed by a 63-bit packed voxel key (3 x 21 bits relative to the minimum
// key); scenes spanning more than 2^21 voxels on an axis (voxel 0.1 on a 10^6-unit scene needs 25
// bits per axis) take the WIDE layout: a 96-bit key in two words, claimed lock-free in two steps
// (see table_add).  Points arrive in arbitrary order; when few voxels hold most points the global
// atomics would pile onto a handful of addresses, so every workgroup first aggregates its
// 4096-point tile in a 1024-slot LDS table and flushes one global atomic per (tile, voxel).
// HBM-bound: 12 B read per point per pass (bbox, count, mask) + 1 B written.
#include <algorithm>
#include <utility>
#include <vector>

#include "gsx_common.h"

namespace gsx {

struct VoxelFrame {
    int kmin[3];
    int dim[3];   // kmax - kmin + 1
    int ok;       // 0: non-finite / absurd coordinates; 1: keys fit 3 x 21 bits; 2: WIDE (x,y in one 64-bit word, z in a second)
    int pad;
};

__device__ __forceinline__ int voxel_key(float v, float voxel)
{
    // np.floor(coords / voxel_size): IEEE f32 divide (hipcc keeps f32 divide correctly rounded), floor in f32
    return (int)floorf(v / voxel);
}

__global__ __launch_bounds__(256) void bbox_partial_kernel2(const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ z, int64_t stride, int64_t n,
                                                            float *__restrict__ part)
{
    __shared__ float red[6][4];
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v[3] = {x[i * stride], y[i * stride], z[i * stride]};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], v[a]);
            mx[a] = fmaxf(mx[a], v[a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            red[a][w] = mn[a];
            red[3 + a][w] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[threadIdx.x][0];
        for (int i = 1; i < 4; ++i) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][i]) : fmaxf(v, red[threadIdx.x][i]);
        part[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(64) void voxel_frame_kernel(const float *__restrict__ part, int nparts, float voxel,
                                                         VoxelFrame *__restrict__ vf)
{
    const int lane = threadIdx.x;
    float v[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        float acc = a < 3 ? __builtin_inff() : -__builtin_inff();
        for (int i = lane; i < nparts; i += 64) acc = a < 3 ? fminf(acc, part[i * 6 + a]) : fmaxf(acc, part[i * 6 + a]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            float o = __shfl_xor(acc, off);
            acc = a < 3 ? fminf(acc, o) : fmaxf(acc, o);
        }
        v[a] = acc;
    }
    if (lane != 0) return;
    int ok = 1;
    for (int a = 0; a < 3; ++a) {
        // x -> floor(x / voxel) is monotone, so the key range is the image of the coordinate range
        float lo = floorf(v[a] / voxel), hi = floorf(v[3 + a] / voxel);
        if (!(fabsf(lo) < 1.0e9f) || !(fabsf(hi) < 1.0e9f)) { ok = 0; lo = hi = 0.f; }
        int klo = (int)lo, khi = (int)hi;
        long long d = (long long)khi - klo + 1;
        if (d > (1 << 21) - 1 && ok) ok = 2;   // |keys| < 1e9 above => d < 2^31: fits the wide layout
        vf->kmin[a] = klo;
        vf->dim[a] = (int)d;
    }
    vf->ok = ok;
    vf->pad = 0;
}

// a = word compared first (0 = empty slot), b = second word (wide layout only; 0 = not yet written)
struct VKey {
    unsigned long long a;
    unsigned b;
};

template <bool WIDE>
__device__ __forceinline__ VKey pack_key(const VoxelFrame &f, int kx, int ky, int kz)
{
    VKey k;
    if (WIDE) {
        k.a = (((unsigned long long)(unsigned)(kx - f.kmin[0]) << 32) | (unsigned long long)(unsigned)(ky - f.kmin[1])) + 1ull;
        k.b = (unsigned)(kz - f.kmin[2]) + 1u;
    } else {
        // +1 so that 0 can mean "empty slot"
        k.a = (((unsigned long long)(unsigned)(kx - f.kmin[0]) << 42) | ((unsigned long long)(unsigned)(ky - f.kmin[1]) << 21) |
               (unsigned long long)(unsigned)(kz - f.kmin[2])) + 1ull;
        k.b = 0u;
    }
    return k;
}

__device__ __forceinline__ unsigned hash_key(unsigned long long k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (unsigned)k;
}
template <bool WIDE>
__device__ __forceinline__ unsigned hash_vkey(const VKey &k)
{
    return WIDE ? hash_key(k.a ^ ((unsigned long long)k.b * 0x9e3779b97f4a7c15ull)) : hash_key(k.a);
}

// Open addressing, linear probing, no locks and no spinning.  WIDE slots are claimed in two steps: CAS the first
// word from 0, then CAS the second from 0; a thread that finds its own first word in a slot whose second word is
// still 0 may complete the slot with ITS second word (the first claimer then sees a mismatch and probes on), so a
// half-written slot never blocks anybody.  Works for tables in HBM and in LDS alike.
template <bool WIDE>
__device__ __forceinline__ bool slot_claim(unsigned long long *ka, unsigned *kb, unsigned h, const VKey &key)
{
    unsigned long long old = ka[h];
    if (old == 0ull) old = atomicCAS(&ka[h], 0ull, key.a);
    if (old != 0ull && old != key.a) return false;
    if (!WIDE) return true;
    unsigned ob = kb[h];
    if (ob == 0u) ob = atomicCAS(&kb[h], 0u, key.b);
    return ob == 0u || ob == key.b;
}

template <bool WIDE>
__device__ __forceinline__ void table_add(unsigned long long *__restrict__ tkeys, unsigned *__restrict__ tkb,
                                          unsigned *__restrict__ tcnt, unsigned mask, const VKey &key, unsigned c)
{
    unsigned h = hash_vkey<WIDE>(key) & mask;
    for (;;) {
        if (slot_claim<WIDE>(tkeys, tkb, h, key)) {
            atomicAdd(&tcnt[h], c);
            return;
        }
        h = (h + 1) & mask;
    }
}

constexpr int VOX_TILE = 4096;   // points per workgroup tile
constexpr int VOX_LDS = 1024;    // LDS aggregation slots

template <bool WIDE>
__global__ __launch_bounds__(256) void voxel_count_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                          const float *__restrict__ z, int64_t stride, int64_t n,
                                                          float voxel, const VoxelFrame *__restrict__ vfp,
                                                          unsigned long long *__restrict__ tkeys, unsigned *__restrict__ tkb,
                                                          unsigned *__restrict__ tcnt, unsigned tmask,
                                                          unsigned *__restrict__ oob /* nullable: the frame came from a caller's box */)
{
    __shared__ unsigned long long lkeys[VOX_LDS];
    __shared__ unsigned lkb[WIDE ? VOX_LDS : 1];
    __shared__ unsigned lcnt[VOX_LDS];
    const VoxelFrame f = *vfp;
    if (!f.ok) return;
    const int64_t ntiles = (n + VOX_TILE - 1) / VOX_TILE;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        for (int i = threadIdx.x; i < VOX_LDS; i += 256) {
            lkeys[i] = 0ull;
            lcnt[i] = 0u;
            if (WIDE) lkb[i] = 0u;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < VOX_TILE; j += 256) {
            int64_t i = t * VOX_TILE + j;
            if (i >= n) break;
            const int kx = voxel_key(x[i * stride], voxel), ky = voxel_key(y[i * stride], voxel), kz = voxel_key(z[i * stride], voxel);
            if (oob) {
                // A frame derived from a box the CALLER supplied is only a promise: a row outside it would wrap in pack_key
                // and collide with a valid voxel, and more distinct keys than the table was sized for would make the linear
                // probe spin forever.  Such a row is not inserted; the host sees the flag and repeats the
                // call with the frame of the rows themselves.
                if ((unsigned)(kx - f.kmin[0]) >= (unsigned)f.dim[0] || (unsigned)(ky - f.kmin[1]) >= (unsigned)f.dim[1] ||
                    (unsigned)(kz - f.kmin[2]) >= (unsigned)f.dim[2]) {
                    *oob = 1u;
                    continue;
                }
            }
            const VKey key = pack_key<WIDE>(f, kx, ky, kz);
            unsigned h = hash_vkey<WIDE>(key) & (VOX_LDS - 1);
            bool done = false;
            for (int probe = 0; probe < 8 && !done; ++probe) {
                if (slot_claim<WIDE>(lkeys, lkb, h, key)) {
                    atomicAdd(&lcnt[h], 1u);
                    done = true;
                } else {
                    h = (h + 1) & (VOX_LDS - 1);
                }
            }
            if (!done) table_add<WIDE>(tkeys, tkb, tcnt, tmask, key, 1u);  // LDS table crowded: straight to HBM
        }
        __syncthreads();
        for (int i = threadIdx.x; i < VOX_LDS; i += 256) {
            unsigned c = lcnt[i];
            // (a wide slot whose second word was never written cannot have a count)
            if (c) table_add<WIDE>(tkeys, tkb, tcnt, tmask, VKey{lkeys[i], WIDE ? lkb[i] : 0u}, c);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void voxel_collect_kernel(const unsigned long long *__restrict__ tkeys,
                                                            const unsigned *__restrict__ tkb /* null: narrow */,
                                                            const unsigned *__restrict__ tcnt, unsigned tsize,
                                                            unsigned min_points, unsigned dense_cap,
                                                            unsigned long long *__restrict__ out_keys,
                                                            unsigned *__restrict__ out_kb, unsigned *__restrict__ out_cnt,
                                                            unsigned *__restrict__ counters /* [0]=unique [1]=dense */)
{
    unsigned uniq = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < tsize; i += gridDim.x * blockDim.x) {
        unsigned long long k = tkeys[i];
        unsigned c = tcnt[i];
        if (k == 0ull || c == 0u) continue;   // c == 0: a wide slot that was claimed but never completed
        ++uniq;
        if (c >= min_points) {
            unsigned slot = atomicAdd(&counters[1], 1u);
            if (slot < dense_cap) {
                out_keys[slot] = k;
                if (tkb) out_kb[slot] = tkb[i];
                out_cnt[slot] = c;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) uniq += __shfl_xor(uniq, off);
    if ((threadIdx.x & 63) == 0 && uniq) atomicAdd(&counters[0], uniq);
}

constexpr int KEPT_LDS = 4096;

template <bool WIDE>
__global__ __launch_bounds__(256) void voxel_mask_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                         const float *__restrict__ z, int64_t stride, int64_t n,
                                                         float voxel, const VoxelFrame *__restrict__ vfp,
                                                         const unsigned long long *__restrict__ kept,
                                                         const unsigned *__restrict__ kept_b, int n_kept,
                                                         uint8_t *__restrict__ mask)
{
    __shared__ unsigned long long lk[KEPT_LDS];
    __shared__ unsigned lkb[WIDE ? KEPT_LDS : 1];
    const VoxelFrame f = *vfp;
    const bool in_lds = n_kept <= KEPT_LDS;
    if (in_lds) {
        for (int i = threadIdx.x; i < n_kept; i += 256) {
            lk[i] = kept[i];
            if (WIDE) lkb[i] = kept_b[i];
        }
        __syncthreads();
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int kx = voxel_key(x[i * stride], voxel), ky = voxel_key(y[i * stride], voxel), kz = voxel_key(z[i * stride], voxel);
        const VKey key = pack_key<WIDE>(f, kx, ky, kz);
        int lo = 0, hi = n_kept;  // first index with kept[idx] >= key, keys ordered by (a, b)
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            const unsigned long long va = in_lds ? lk[mid] : kept[mid];
            const unsigned vb = WIDE ? (in_lds ? lkb[mid] : kept_b[mid]) : 0u;
            if (va < key.a || (va == key.a && vb < key.b)) lo = mid + 1; else hi = mid;
        }
        bool hit = false;
        if (lo < n_kept) {
            const unsigned long long va = in_lds ? lk[lo] : kept[lo];
            const unsigned vb = WIDE ? (in_lds ? lkb[lo] : kept_b[lo]) : 0u;
            hit = va == key.a && vb == key.b;
        }
        mask[i] = hit ? 1 : 0;
    }
}

static int blocks_for(const gsx_ctx *ctx, int64_t n, int per_block)
{
    int64_t want = (n + per_block - 1) / per_block;
    return (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cu * 8));
}

// frame (key range) of a device-resident cloud -> host copy
static int compute_frame(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                         float voxel, VoxelFrame *dev_vf, VoxelFrame *host_vf)
{
    const int blocks = blocks_for(c, n, 1024);
    GSX_CHECK(c->ws[0].bboxpart.reserve(sizeof(float) * 6 * (size_t)blocks));
    hipLaunchKernelGGL(bbox_partial_kernel2, dim3(blocks), dim3(256), 0, c->stream, x, y, z, stride, n,
                       c->ws[0].bboxpart.as<float>());
    hipLaunchKernelGGL(voxel_frame_kernel, dim3(1), dim3(64), 0, c->stream, c->ws[0].bboxpart.as<float>(), blocks, voxel, dev_vf);
    GSX_HIP(hipGetLastError());
    GSX_HIP(hipMemcpyAsync(host_vf, dev_vf, sizeof(VoxelFrame), hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    if (!host_vf->ok)
        GSX_FAIL("density: coordinates are not finite, or extent / voxel_size exceeds 1e9 voxels per axis");
    return 0;
}

// ---- the voxel table of one call: keys | counts | second key word (wide) in scratch2, then room for `cap` exported entries
struct VoxTable {
    bool wide;
    uint64_t tsize;
    unsigned long long *tkeys;
    unsigned *tcnt, *tkb;
    char *out_base;      // 16-byte aligned area behind the table (cap x (8 + 4 + 4) bytes + 16)
    size_t table_bytes;  // bytes to zero before inserting
};

static int alloc_table(gsx_ctx *c, const VoxelFrame &hvf, int64_t n_items, size_t out_bytes, VoxTable *t)
{
    // table size: power of two >= 2 x min(items, number of voxels in the frame)
    t->wide = hvf.ok == 2;
    const double space = (double)hvf.dim[0] * hvf.dim[1] * hvf.dim[2];
    const uint64_t need = (uint64_t)std::min<double>((double)n_items, space);
    uint64_t tsize = 1024;
    while (tsize < 2 * need) tsize <<= 1;
    if (tsize > (1ull << 31)) GSX_FAIL("density: too many points for the voxel table");
    t->tsize = tsize;
    const size_t off_cnt = sizeof(unsigned long long) * tsize;
    const size_t off_kb = off_cnt + sizeof(unsigned) * tsize;
    size_t off_out = off_kb + (t->wide ? sizeof(unsigned) * tsize : 0);
    off_out = (off_out + 15) & ~(size_t)15;
    GSX_CHECK(c->scratch2.reserve(off_out + out_bytes + 64));
    char *base = c->scratch2.as<char>();
    t->tkeys = reinterpret_cast<unsigned long long *>(base);
    t->tcnt = reinterpret_cast<unsigned *>(base + off_cnt);
    t->tkb = t->wide ? reinterpret_cast<unsigned *>(base + off_kb) : nullptr;
    t->out_base = base + off_out;
    t->table_bytes = off_out;
    GSX_HIP(hipMemsetAsync(base, 0, off_out, c->stream));
    return 0;
}

// voxels with count >= min_points, in np.unique(axis=0) row order, as absolute int64 triples (host arrays)
static int collect_dense(gsx_ctx *c, const VoxelFrame &hvf, const VoxTable &t, int64_t min_points, int64_t dense_cap,
                         int64_t *n_unique_out, int64_t *n_dense_out, int64_t *dense_keys_out, int64_t *dense_counts_out)
{
    const size_t cap = (size_t)std::max<int64_t>(dense_cap, 1);
    unsigned long long *okeys = reinterpret_cast<unsigned long long *>(t.out_base);
    unsigned *ocnt = reinterpret_cast<unsigned *>(t.out_base + sizeof(unsigned long long) * cap);
    unsigned *okb = ocnt + cap;
    unsigned *ctr = reinterpret_cast<unsigned *>((reinterpret_cast<uintptr_t>(okb + cap) + 15) & ~(uintptr_t)15);
    GSX_HIP(hipMemsetAsync(ctr, 0, 16, c->stream));
    const unsigned mp = (unsigned)std::min<int64_t>(std::max<int64_t>(min_points, 0), 0xffffffffll);
    hipLaunchKernelGGL(voxel_collect_kernel, dim3(blocks_for(c, (int64_t)t.tsize, 1024)), dim3(256), 0, c->stream, t.tkeys, t.tkb,
                       t.tcnt, (unsigned)t.tsize, mp, (unsigned)cap, okeys, okb, ocnt, ctr);
    GSX_HIP(hipGetLastError());
    unsigned hctr[2];
    GSX_HIP(hipMemcpyAsync(hctr, ctr, sizeof(hctr), hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    if ((int64_t)hctr[1] > dense_cap)
        GSX_FAIL("density: %u dense voxels exceed dense_cap=%lld", hctr[1], (long long)dense_cap);
    const size_t m = hctr[1];
    const bool wide = t.wide;
    std::vector<unsigned long long> hk(m);
    std::vector<unsigned> hc(m), hb(m, 0u);
    if (m) {
        GSX_HIP(hipMemcpy(hk.data(), okeys, sizeof(unsigned long long) * m, hipMemcpyDeviceToHost));
        GSX_HIP(hipMemcpy(hc.data(), ocnt, sizeof(unsigned) * m, hipMemcpyDeviceToHost));
        if (wide) GSX_HIP(hipMemcpy(hb.data(), okb, sizeof(unsigned) * m, hipMemcpyDeviceToHost));
    }
    // packed keys order like (x, y, z) tuples because kmin is subtracted per axis: sort = np.unique's row order
    std::vector<size_t> order(m);
    for (size_t i = 0; i < m; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return hk[a] != hk[b] ? hk[a] < hk[b] : hb[a] < hb[b]; });
    for (size_t i = 0; i < m; ++i) {
        unsigned long long k = hk[order[i]] - 1ull;
        if (wide) {
            dense_keys_out[3 * i + 0] = (int64_t)(k >> 32) + hvf.kmin[0];
            dense_keys_out[3 * i + 1] = (int64_t)(k & 0xffffffffull) + hvf.kmin[1];
            dense_keys_out[3 * i + 2] = (int64_t)(hb[order[i]] - 1u) + hvf.kmin[2];
        } else {
            dense_keys_out[3 * i + 0] = (int64_t)(k >> 42) + hvf.kmin[0];
            dense_keys_out[3 * i + 1] = (int64_t)((k >> 21) & 0x1fffff) + hvf.kmin[1];
            dense_keys_out[3 * i + 2] = (int64_t)(k & 0x1fffff) + hvf.kmin[2];
        }
        dense_counts_out[i] = hc[order[i]];
    }
    *n_unique_out = hctr[0];
    *n_dense_out = (int64_t)m;
    return 0;
}

// Round 5: a frame of at most VOX_DENSE_MAX voxels (BASELINE configs[2]: 5 x 5 x 5) is counted in a DIRECT-INDEXED LDS histogram per
// tile -- one ds_add per point instead of hash, read, compare-and-swap, add on ~125 addresses every lane fights over (91 of the
// density stage's 155 us at 10M splats) -- and flushed into the same HBM table as before (one table_add per occupied voxel and
// tile), so that the collect / cluster / mask kernels are untouched.
constexpr int VOX_DENSE_MAX = 4096;
constexpr int VOX_DENSE_LDS = 4096;   // histogram words per workgroup (16 KiB: ten workgroups per CU): the frame's bins, replicated while they fit
constexpr int VOX_DENSE_ROWS = 8;     // rows in flight per lane
// Round 6: (i) the kernel also leaves each point's dense voxel index in a 2-byte side array (`ids`, 0xffff = outside the frame),
// so that the membership pass reads 2 B per point instead of the 12-byte row again; (ii) the histogram is REPLICATED in LDS
// (copy = lane & (copies - 1)): 64 lanes adding into ~125 bins of one copy serialise, 16 copies spread them over 2000 words;
// (iii) (n,3) rows -- what the drop-in and the device chain pass -- are read as one 12-byte load per point instead of three
// 4-byte loads 12 bytes apart.
template <bool PACKED>
__global__ __launch_bounds__(256) void voxel_count_dense_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                const float *__restrict__ z, int64_t stride, int64_t n, float voxel,
                                                                const VoxelFrame *__restrict__ vfp, unsigned long long *__restrict__ tkeys,
                                                                unsigned *__restrict__ tcnt, unsigned tmask,
                                                                unsigned *__restrict__ oob /* nullable */, uint16_t *__restrict__ ids /* nullable */)
{
    __shared__ unsigned bins[VOX_DENSE_LDS];
    const VoxelFrame f = *vfp;
    if (!f.ok) return;
    const int d1 = f.dim[1], d2 = f.dim[2];
    const int nv = f.dim[0] * d1 * d2;   // <= VOX_DENSE_MAX (the host checked)
    int copies = 1;
    while (copies < 16 && 2 * copies * nv <= VOX_DENSE_LDS) copies *= 2;
    for (int i = threadIdx.x; i < copies * nv; i += 256) bins[i] = 0u;   // (copies * nv <= max(nv, VOX_DENSE_LDS / 2 ... VOX_DENSE_LDS))
    __syncthreads();
    unsigned *mine = bins + (threadIdx.x & (copies - 1)) * nv;
    bool out = false;
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += VOX_DENSE_ROWS * step) {   // 8 rows (24 dwords) in flight per lane
        float v[VOX_DENSE_ROWS][3];
#pragma unroll
        for (int u = 0; u < VOX_DENSE_ROWS; ++u) {
            const int64_t r = i0 + u * step < n ? i0 + u * step : i0;
            if (PACKED) {
                const float3 p = reinterpret_cast<const float3 *>(x)[r];
                v[u][0] = p.x;
                v[u][1] = p.y;
                v[u][2] = p.z;
            } else {
                v[u][0] = x[r * stride];
                v[u][1] = y[r * stride];
                v[u][2] = z[r * stride];
            }
        }
#pragma unroll
        for (int u = 0; u < VOX_DENSE_ROWS; ++u) {
            if (i0 + u * step >= n) continue;
            const unsigned a = (unsigned)(voxel_key(v[u][0], voxel) - f.kmin[0]), b = (unsigned)(voxel_key(v[u][1], voxel) - f.kmin[1]),
                           c = (unsigned)(voxel_key(v[u][2], voxel) - f.kmin[2]);
            if (a >= (unsigned)f.dim[0] || b >= (unsigned)d1 || c >= (unsigned)d2) {   // only possible with a caller's box (see `oob`)
                out = true;
                if (ids) ids[i0 + u * step] = 0xffffu;
                continue;
            }
            const unsigned id = (a * d1 + b) * d2 + c;
            atomicAdd(&mine[id], 1u);
            if (ids) ids[i0 + u * step] = (uint16_t)id;
        }
    }
    if (out && oob) *oob = 1u;
    __syncthreads();
    for (int i = threadIdx.x; i < nv; i += 256) {
        unsigned cnt = 0;
        for (int cp = 0; cp < copies; ++cp) cnt += bins[cp * nv + i];
        if (!cnt) continue;
        const int a = i / (d1 * d2), r = i - a * d1 * d2, b = r / d2, c = r - b * d2;
        table_add<false>(tkeys, nullptr, tcnt, tmask, pack_key<false>(f, a + f.kmin[0], b + f.kmin[1], c + f.kmin[2]), cnt);
    }
}

static bool dense_frame(const VoxelFrame *host_frame, const VoxTable &t)
{
    return host_frame && !t.wide && host_frame->ok == 1 &&
           (int64_t)host_frame->dim[0] * host_frame->dim[1] * host_frame->dim[2] <= VOX_DENSE_MAX;
}

static int count_points(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n, float voxel,
                        const VoxelFrame *dvf, const VoxTable &t, unsigned *oob = nullptr, const VoxelFrame *host_frame = nullptr,
                        uint16_t *ids = nullptr)
{
    if (dense_frame(host_frame, t)) {
        // measured at 10M points, 125 voxels (profiles/r06_variants.txt): 1 / 2 / 3 / 4 / 6 / 8 workgroups per CU -> 55 / 47 / 50 / 59 /
        // 77 / 91 us -- the same with the final per-bin atomics replaced by side-by-side histograms and one reducing workgroup
        static const int per_cu = getenv("GSX_VOX_DENSE_WGS") ? std::max(1, atoi(getenv("GSX_VOX_DENSE_WGS"))) : 2;
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(div_up(n, 4096), (int64_t)c->num_cu * per_cu));
        if (stride == 3 && y == x + 1 && z == x + 2)
            hipLaunchKernelGGL(voxel_count_dense_kernel<true>, dim3(blocks), dim3(256), 0, c->stream, x, y, z, stride, n, voxel, dvf, t.tkeys,
                               t.tcnt, (unsigned)(t.tsize - 1), oob, ids);
        else
            hipLaunchKernelGGL(voxel_count_dense_kernel<false>, dim3(blocks), dim3(256), 0, c->stream, x, y, z, stride, n, voxel, dvf, t.tkeys,
                               t.tcnt, (unsigned)(t.tsize - 1), oob, ids);
        GSX_HIP(hipGetLastError());
        return 0;
    }
    if (t.wide)
        hipLaunchKernelGGL((voxel_count_kernel<true>), dim3(blocks_for(c, n, VOX_TILE)), dim3(256), 0, c->stream, x, y, z, stride, n,
                           voxel, dvf, t.tkeys, t.tkb, t.tcnt, (unsigned)(t.tsize - 1), oob);
    else
        hipLaunchKernelGGL((voxel_count_kernel<false>), dim3(blocks_for(c, n, VOX_TILE)), dim3(256), 0, c->stream, x, y, z, stride, n,
                           voxel, dvf, t.tkeys, t.tkb, t.tcnt, (unsigned)(t.tsize - 1), oob);
    GSX_HIP(hipGetLastError());
    return 0;
}

int density_voxels_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                       double voxel_size, int64_t min_points, int64_t dense_cap, int64_t *n_unique_out,
                       int64_t *n_dense_out, int64_t *dense_keys_out, int64_t *dense_counts_out)
{
    const float voxel = (float)voxel_size;  // python float is a weak scalar next to the f32 array
    if (!(voxel > 0.0f)) GSX_FAIL("density: voxel_size must be > 0");
    if (dense_cap < 0) GSX_FAIL("density: bad dense_cap");
    GSX_CHECK(timing_begin(c, GSX_T_DENSITY));
    GSX_CHECK(c->scratch5.reserve(sizeof(VoxelFrame) + 64));
    VoxelFrame *dvf = c->scratch5.as<VoxelFrame>();
    VoxelFrame hvf;
    GSX_CHECK(compute_frame(c, x, y, z, stride, n, voxel, dvf, &hvf));
    VoxTable t;
    const size_t cap = (size_t)std::max<int64_t>(dense_cap, 1);
    GSX_CHECK(alloc_table(c, hvf, n, 16 * cap + 64, &t));
    GSX_CHECK(count_points(c, x, y, z, stride, n, voxel, dvf, t, nullptr, &hvf));
    const int rc = collect_dense(c, hvf, t, min_points, dense_cap, n_unique_out, n_dense_out, dense_keys_out, dense_counts_out);
    GSX_CHECK(timing_end(c, GSX_T_DENSITY));
    return rc;
}

// ---- multi-GPU density: per-rank voxel histograms -> merged counts -----------------------------
// every occupied voxel of the table as an absolute int64 key triple + int64 count (arbitrary order)
__global__ __launch_bounds__(256) void voxel_export_kernel(const unsigned long long *__restrict__ tkeys, const unsigned *__restrict__ tkb,
                                                           const unsigned *__restrict__ tcnt, unsigned tsize, VoxelFrame f, unsigned cap,
                                                           long long *__restrict__ keys3, long long *__restrict__ counts,
                                                           unsigned *__restrict__ counter)
{
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < tsize; i += gridDim.x * blockDim.x) {
        const unsigned long long kk = tkeys[i];
        const unsigned c = tcnt[i];
        if (kk == 0ull || c == 0u) continue;
        const unsigned slot = atomicAdd(counter, 1u);
        if (slot >= cap) continue;
        const unsigned long long k = kk - 1ull;
        long long kx, ky, kz;
        if (tkb) {
            kx = (long long)(k >> 32);
            ky = (long long)(k & 0xffffffffull);
            kz = (long long)(tkb[i] - 1u);
        } else {
            kx = (long long)(k >> 42);
            ky = (long long)((k >> 21) & 0x1fffff);
            kz = (long long)(k & 0x1fffff);
        }
        keys3[3 * (size_t)slot + 0] = kx + f.kmin[0];
        keys3[3 * (size_t)slot + 1] = ky + f.kmin[1];
        keys3[3 * (size_t)slot + 2] = kz + f.kmin[2];
        counts[slot] = (long long)c;
    }
}

int density_hist_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n, double voxel_size,
                     int64_t cap, int64_t *n_unique_out, int64_t *keys3_dev, int64_t *counts_dev)
{
    const float voxel = (float)voxel_size;
    if (!(voxel > 0.0f)) GSX_FAIL("density: voxel_size must be > 0");
    if (cap < 1) GSX_FAIL("density: bad cap");
    GSX_CHECK(timing_begin(c, GSX_T_DENSITY));
    GSX_CHECK(c->scratch5.reserve(sizeof(VoxelFrame) + 64));
    VoxelFrame *dvf = c->scratch5.as<VoxelFrame>();
    VoxelFrame hvf;
    GSX_CHECK(compute_frame(c, x, y, z, stride, n, voxel, dvf, &hvf));
    VoxTable t;
    GSX_CHECK(alloc_table(c, hvf, n, 64, &t));
    GSX_CHECK(count_points(c, x, y, z, stride, n, voxel, dvf, t));
    unsigned *ctr = reinterpret_cast<unsigned *>(t.out_base);
    GSX_HIP(hipMemsetAsync(ctr, 0, 16, c->stream));
    hipLaunchKernelGGL(voxel_export_kernel, dim3(blocks_for(c, (int64_t)t.tsize, 1024)), dim3(256), 0, c->stream, t.tkeys, t.tkb, t.tcnt,
                       (unsigned)t.tsize, hvf, (unsigned)std::min<int64_t>(cap, 0xffffffffll), reinterpret_cast<long long *>(keys3_dev),
                       reinterpret_cast<long long *>(counts_dev), ctr);
    GSX_HIP(hipGetLastError());
    unsigned h = 0;
    GSX_HIP(hipMemcpyAsync(&h, ctr, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    GSX_CHECK(timing_end(c, GSX_T_DENSITY));
    *n_unique_out = h;
    if ((int64_t)h > cap) GSX_FAIL("density: %u occupied voxels exceed cap=%lld", h, (long long)cap);
    return 0;
}

// (entries with count 0 are the padding of the all-gathered lists: their keys are ignored -- a zero key must not stretch
//  the frame of a scene far from the origin)
__global__ __launch_bounds__(256) void merge_minmax_kernel(const long long *__restrict__ keys3, const long long *__restrict__ counts,
                                                           int64_t m, long long *__restrict__ mm)
{
    long long mn[3] = {0x7fffffffffffffffll, 0x7fffffffffffffffll, 0x7fffffffffffffffll};
    long long mx[3] = {-0x7fffffffffffffffll, -0x7fffffffffffffffll, -0x7fffffffffffffffll};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        if (counts[i] == 0) continue;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const long long v = keys3[3 * i + a];
            mn[a] = v < mn[a] ? v : mn[a];
            mx[a] = v > mx[a] ? v : mx[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const long long o1 = __shfl_xor(mn[a], off), o2 = __shfl_xor(mx[a], off);
            mn[a] = o1 < mn[a] ? o1 : mn[a];
            mx[a] = o2 > mx[a] ? o2 : mx[a];
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&mm[a], mn[a]);
            atomicMax(&mm[3 + a], mx[a]);
        }
    }
}

template <bool WIDE>
__global__ __launch_bounds__(256) void merge_insert_kernel(const long long *__restrict__ keys3, const long long *__restrict__ counts,
                                                           int64_t m, VoxelFrame f, unsigned long long *__restrict__ tkeys,
                                                           unsigned *__restrict__ tkb, unsigned *__restrict__ tcnt, unsigned tmask)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        if (counts[i] == 0) continue;   // padding
        const VKey key = pack_key<WIDE>(f, (int)keys3[3 * i], (int)keys3[3 * i + 1], (int)keys3[3 * i + 2]);
        table_add<WIDE>(tkeys, tkb, tcnt, tmask, key, (unsigned)counts[i]);
    }
}

// G all-gathered (key, count) lists concatenated in device memory -> what gsx_density_voxels would have returned for the
// whole cloud: the counts of equal keys are added in a hash table, the dense voxels come back in np.unique row order
int density_merge_dev(gsx_ctx *c, const int64_t *keys3_dev, const int64_t *counts_dev, int64_t m, int64_t min_points, int64_t dense_cap,
                      int64_t *n_unique_out, int64_t *n_dense_out, int64_t *dense_keys_out, int64_t *dense_counts_out)
{
    if (dense_cap < 0 || m < 0) GSX_FAIL("density: bad sizes");
    *n_unique_out = *n_dense_out = 0;
    if (m == 0) return 0;
    GSX_CHECK(timing_begin(c, GSX_T_DENSITY));
    GSX_CHECK(c->scratch5.reserve(64));
    long long *mm = c->scratch5.as<long long>();
    const long long init[6] = {0x7fffffffffffffffll, 0x7fffffffffffffffll, 0x7fffffffffffffffll,
                               -0x7fffffffffffffffll, -0x7fffffffffffffffll, -0x7fffffffffffffffll};
    GSX_HIP(hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    const long long *k3 = reinterpret_cast<const long long *>(keys3_dev);
    const long long *cn = reinterpret_cast<const long long *>(counts_dev);
    hipLaunchKernelGGL(merge_minmax_kernel, dim3(blocks_for(c, m, 1024)), dim3(256), 0, c->stream, k3, cn, m, mm);
    long long h[6];
    GSX_HIP(hipMemcpyAsync(h, mm, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    VoxelFrame hvf;
    hvf.ok = 1;
    hvf.pad = 0;
    for (int a = 0; a < 3; ++a) {
        if (h[a] < -1000000000ll || h[3 + a] > 1000000000ll) GSX_FAIL("density: voxel keys out of range");
        const long long d = h[3 + a] - h[a] + 1;
        if (d > (1 << 21) - 1) hvf.ok = 2;
        hvf.kmin[a] = (int)h[a];
        hvf.dim[a] = (int)d;
    }
    VoxTable t;
    const size_t cap = (size_t)std::max<int64_t>(dense_cap, 1);
    GSX_CHECK(alloc_table(c, hvf, m, 16 * cap + 64, &t));
    if (t.wide)
        hipLaunchKernelGGL((merge_insert_kernel<true>), dim3(blocks_for(c, m, 1024)), dim3(256), 0, c->stream, k3, cn, m, hvf, t.tkeys, t.tkb,
                           t.tcnt, (unsigned)(t.tsize - 1));
    else
        hipLaunchKernelGGL((merge_insert_kernel<false>), dim3(blocks_for(c, m, 1024)), dim3(256), 0, c->stream, k3, cn, m, hvf, t.tkeys, t.tkb,
                           t.tcnt, (unsigned)(t.tsize - 1));
    GSX_HIP(hipGetLastError());
    const int rc = collect_dense(c, hvf, t, min_points, dense_cap, n_unique_out, n_dense_out, dense_keys_out, dense_counts_out);
    GSX_CHECK(timing_end(c, GSX_T_DENSITY));
    return rc;
}

int density_mask_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                     double voxel_size, const int64_t *kept_keys, int64_t n_kept, uint8_t *mask_dev)
{
    const float voxel = (float)voxel_size;
    if (!(voxel > 0.0f)) GSX_FAIL("density: voxel_size must be > 0");
    GSX_CHECK(timing_begin(c, GSX_T_DENSITY));
    GSX_CHECK(c->scratch5.reserve(sizeof(VoxelFrame) + 64));
    VoxelFrame *dvf = c->scratch5.as<VoxelFrame>();
    VoxelFrame hvf;
    GSX_CHECK(compute_frame(c, x, y, z, stride, n, voxel, dvf, &hvf));
    const bool wide = hvf.ok == 2;
    std::vector<std::pair<unsigned long long, unsigned>> packed;
    packed.reserve((size_t)n_kept);
    for (int64_t i = 0; i < n_kept; ++i) {
        int64_t r[3];
        bool inside = true;
        for (int a = 0; a < 3; ++a) {
            r[a] = kept_keys[3 * i + a] - hvf.kmin[a];
            inside &= r[a] >= 0 && r[a] < hvf.dim[a];
        }
        if (!inside) continue;  // a kept voxel outside the cloud's key range can match no point
        if (wide)
            packed.emplace_back((((unsigned long long)r[0] << 32) | (unsigned long long)r[1]) + 1ull, (unsigned)r[2] + 1u);
        else
            packed.emplace_back((((unsigned long long)r[0] << 42) | ((unsigned long long)r[1] << 21) | (unsigned long long)r[2]) + 1ull, 0u);
    }
    std::sort(packed.begin(), packed.end());
    packed.erase(std::unique(packed.begin(), packed.end()), packed.end());
    const int nk = (int)packed.size();
    std::vector<unsigned long long> pa((size_t)std::max(nk, 1));
    std::vector<unsigned> pb((size_t)std::max(nk, 1));
    for (int i = 0; i < nk; ++i) {
        pa[i] = packed[i].first;
        pb[i] = packed[i].second;
    }
    const size_t off_b = sizeof(unsigned long long) * (size_t)std::max(nk, 1);
    GSX_CHECK(c->scratch2.reserve(off_b + sizeof(unsigned) * (size_t)std::max(nk, 1)));
    unsigned long long *dka = c->scratch2.as<unsigned long long>();
    unsigned *dkb = reinterpret_cast<unsigned *>(c->scratch2.as<char>() + off_b);
    if (nk) {
        GSX_HIP(hipMemcpyAsync(dka, pa.data(), sizeof(unsigned long long) * nk, hipMemcpyHostToDevice, c->stream));
        GSX_HIP(hipMemcpyAsync(dkb, pb.data(), sizeof(unsigned) * nk, hipMemcpyHostToDevice, c->stream));
    }
    if (wide)
        hipLaunchKernelGGL((voxel_mask_kernel<true>), dim3(blocks_for(c, n, 1024)), dim3(256), 0, c->stream, x, y, z, stride, n, voxel,
                           dvf, dka, dkb, nk, mask_dev);
    else
        hipLaunchKernelGGL((voxel_mask_kernel<false>), dim3(blocks_for(c, n, 1024)), dim3(256), 0, c->stream, x, y, z, stride, n, voxel,
                           dvf, dka, dkb, nk, mask_dev);
    GSX_HIP(hipGetLastError());
    GSX_HIP(hipStreamSynchronize(c->stream));  // the host key vectors (pageable memory) must outlive the async copies
    GSX_CHECK(timing_end(c, GSX_T_DENSITY));
    return 0;
}

// ---- the whole filter on the device (gsx_density_filter_dev): count -> dense voxels -> 6-connected clusters -> keep rule ->
// membership mask, with no host round trip in between (data_processor.py:38-114 start to finish).
// The cluster step is one workgroup: at most n / min_points voxels are dense (<= 1001 for the reference's thresholds,
// threshold_percentage >= 0.1), so they fit LDS; more than CL_MAX, or a tie for the largest cluster without
// keep_multicluster (which the reference resolves by the iteration order of a python set), hands the decision to the host
// path (gsx_density_voxels_dev + processing/clusters.py): status GSX_DENSITY_HOST.
constexpr int CL_MAX = 1024;

struct ClusterOut {          // device-side result block (gsx_density_info's device half)
    unsigned n_unique, n_dense, n_kept_voxels, kept_clusters, largest, status;
    unsigned oob, pad;       // oob: a row fell outside the frame derived from the caller's box -- nothing of this block is valid
};

template <bool WIDE>
__device__ __forceinline__ bool key_less(unsigned long long a1, unsigned b1, unsigned long long a2, unsigned b2)
{
    return a1 < a2 || (WIDE && a1 == a2 && b1 < b2);
}

template <bool WIDE>
__global__ __launch_bounds__(256) void voxel_cluster_kernel(const unsigned long long *__restrict__ okeys, const unsigned *__restrict__ okb,
                                                            const unsigned *__restrict__ counters /* [0] unique, [1] dense */,
                                                            unsigned dense_cap, VoxelFrame f, int keep_multi,
                                                            unsigned long long *__restrict__ kept_a, unsigned *__restrict__ kept_b,
                                                            ClusterOut *__restrict__ out)
{
    __shared__ unsigned long long ka[CL_MAX];
    __shared__ unsigned kb[WIDE ? CL_MAX : 1];
    __shared__ unsigned short nb[CL_MAX][6];
    __shared__ unsigned lab[CL_MAX];
    __shared__ unsigned siz[CL_MAX];
    __shared__ unsigned s_scan[256];
    __shared__ unsigned s_largest, s_ties, s_nkept, s_ncl;
    const int tid = threadIdx.x;
    const unsigned D = counters[1];
    if (tid == 0) {
        out->n_unique = counters[0];
        out->n_dense = D;
        out->n_kept_voxels = 0;
        out->kept_clusters = 0;
        out->largest = 0;
        out->status = D == 0 ? 1u : ((D > (unsigned)CL_MAX || D > dense_cap) ? 2u : 0u);
        out->oob = counters[2];
        out->pad = 0;
    }
    if (D == 0 || D > (unsigned)CL_MAX || D > dense_cap) return;
    // ---- sorted keys (np.unique's row order; the mask kernel's binary search needs it too): bitonic sort, padded with max keys
    int P = 1;
    while (P < (int)D) P <<= 1;
    for (int i = tid; i < P; i += 256) {
        ka[i] = i < (int)D ? okeys[i] : ~0ull;
        if (WIDE) kb[i] = i < (int)D ? okb[i] : ~0u;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k2) == 0;
                    const unsigned long long a1 = ka[i], a2 = ka[l];
                    const unsigned b1 = WIDE ? kb[i] : 0u, b2 = WIDE ? kb[l] : 0u;
                    if (key_less<WIDE>(a2, b2, a1, b1) == up) {
                        ka[i] = a2;
                        ka[l] = a1;
                        if (WIDE) {
                            kb[i] = b2;
                            kb[l] = b1;
                        }
                    }
                }
            }
            __syncthreads();
        }
    // ---- the six face neighbours of every dense voxel (data_processor.py:67-68), by binary search
    for (int i = tid; i < (int)D; i += 256) {
        const unsigned long long k = ka[i] - 1ull;
        long long c[3];
        if (WIDE) {
            c[0] = (long long)(k >> 32);
            c[1] = (long long)(k & 0xffffffffull);
            c[2] = (long long)(kb[i] - 1u);
        } else {
            c[0] = (long long)(k >> 42);
            c[1] = (long long)((k >> 21) & 0x1fffff);
            c[2] = (long long)(k & 0x1fffff);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            long long t[3] = {c[0], c[1], c[2]};
            t[q >> 1] += (q & 1) ? 1 : -1;
            unsigned short found = 0xffff;
            if (t[q >> 1] >= 0 && t[q >> 1] < (long long)f.dim[q >> 1]) {
                unsigned long long na;
                unsigned nbb = 0u;
                if (WIDE) {
                    na = (((unsigned long long)t[0] << 32) | (unsigned long long)t[1]) + 1ull;
                    nbb = (unsigned)t[2] + 1u;
                } else {
                    na = (((unsigned long long)t[0] << 42) | ((unsigned long long)t[1] << 21) | (unsigned long long)t[2]) + 1ull;
                }
                int lo = 0, hi = (int)D;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (key_less<WIDE>(ka[mid], WIDE ? kb[mid] : 0u, na, nbb)) lo = mid + 1; else hi = mid;
                }
                if (lo < (int)D && ka[lo] == na && (!WIDE || kb[lo] == nbb)) found = (unsigned short)lo;
            }
            nb[i][q] = found;
        }
        lab[i] = (unsigned)i;
        siz[i] = 0u;
    }
    __syncthreads();
    // ---- connected components: minimum-label propagation + pointer jumping until nothing moves
    for (;;) {
        int changed = 0;
        for (int i = tid; i < (int)D; i += 256) {
            unsigned m = lab[i];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const unsigned short j = nb[i][q];
                if (j != 0xffff) m = min(m, lab[j]);
            }
            m = min(m, lab[m]);
            if (m < lab[i]) {
                atomicMin(&lab[i], m);
                changed = 1;
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
    for (int i = tid; i < (int)D; i += 256) {   // full compression, then the cluster sizes (in voxels: data_processor.py:95)
        unsigned r = lab[i];
        while (lab[r] != r) r = lab[r];
        lab[i] = r;
    }
    __syncthreads();
    for (int i = tid; i < (int)D; i += 256) atomicAdd(&siz[lab[i]], 1u);
    if (tid == 0) {
        s_largest = 0;
        s_ties = 0;
        s_ncl = 0;
    }
    __syncthreads();
    for (int i = tid; i < (int)D; i += 256)
        if (lab[i] == (unsigned)i) atomicMax(&s_largest, siz[i]);
    __syncthreads();
    const unsigned largest = s_largest;
    for (int i = tid; i < (int)D; i += 256)
        if (lab[i] == (unsigned)i && siz[i] == largest) atomicAdd(&s_ties, 1u);
    __syncthreads();
    if (!keep_multi && s_ties > 1) {   // which of the equally large clusters survives is the reference's set-iteration order
        if (tid == 0) out->status = 2u;
        return;
    }
    // keep rule (data_processor.py:96-106): all clusters of at least 5 % of the largest, or the largest alone
    const double floor_sz = keep_multi ? (double)largest * 0.05 : (double)largest;
    for (int i = tid; i < (int)D; i += 256)
        if (lab[i] == (unsigned)i && (double)siz[i] >= floor_sz) atomicAdd(&s_ncl, 1u);
    // ---- the kept voxels, still sorted: block scan of the keep flags
    const int per = (P + 255) / 256;
    unsigned mine = 0;
    for (int u = 0; u < per; ++u) {
        const int i = tid * per + u;
        if (i < (int)D && (double)siz[lab[i]] >= floor_sz) ++mine;
    }
    s_scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned v = tid >= off ? s_scan[tid - off] : 0u;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    unsigned at = s_scan[tid] - mine;
    for (int u = 0; u < per; ++u) {
        const int i = tid * per + u;
        if (i < (int)D && (double)siz[lab[i]] >= floor_sz) {
            kept_a[at] = ka[i];
            if (WIDE) kept_b[at] = kb[i];
            ++at;
        }
    }
    if (tid == 255) s_nkept = s_scan[255];
    __syncthreads();
    if (tid == 0) {
        out->n_kept_voxels = s_nkept;
        out->kept_clusters = s_ncl;
        out->largest = largest;
        out->status = s_nkept ? 0u : 1u;
    }
}

// membership against the kept list the cluster kernel left on the device (its length included)
template <bool WIDE>
__global__ __launch_bounds__(256) void voxel_mask_dev_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                             const float *__restrict__ z, int64_t stride, int64_t n, float voxel,
                                                             VoxelFrame f, const unsigned long long *__restrict__ kept,
                                                             const unsigned *__restrict__ kept_b, const ClusterOut *__restrict__ co,
                                                             uint8_t *__restrict__ mask)
{
    __shared__ unsigned long long lk[CL_MAX];
    __shared__ unsigned lkb[WIDE ? CL_MAX : 1];
    const int n_kept = co->status == 0u ? (int)co->n_kept_voxels : 0;
    for (int i = threadIdx.x; i < n_kept; i += 256) {
        lk[i] = kept[i];
        if (WIDE) lkb[i] = kept_b[i];
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const VKey key = pack_key<WIDE>(f, voxel_key(x[i * stride], voxel), voxel_key(y[i * stride], voxel), voxel_key(z[i * stride], voxel));
        int lo = 0, hi = n_kept;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (key_less<WIDE>(lk[mid], WIDE ? lkb[mid] : 0u, key.a, key.b)) lo = mid + 1; else hi = mid;
        }
        mask[i] = (lo < n_kept && lk[lo] == key.a && (!WIDE || lkb[lo] == key.b)) ? 1 : 0;
    }
}

// membership from the side array: mask[i] = the voxel with dense index ids[i] is in the kept list (narrow keys)
__global__ __launch_bounds__(256) void voxel_mask_ids_kernel(const uint16_t *__restrict__ ids, int64_t n, VoxelFrame f,
                                                             const unsigned long long *__restrict__ kept, const ClusterOut *__restrict__ co,
                                                             uint8_t *__restrict__ mask)
{
    __shared__ uint8_t flag[VOX_DENSE_MAX + 4];
    const int d1 = f.dim[1], d2 = f.dim[2];
    const int nv = f.dim[0] * d1 * d2;
    for (int i = threadIdx.x; i < nv; i += 256) flag[i] = 0;
    __syncthreads();
    const int n_kept = co->status == 0u ? (int)co->n_kept_voxels : 0;
    for (int i = threadIdx.x; i < n_kept; i += 256) {
        const unsigned long long k = kept[i] - 1ull;   // pack_key<false>: (a << 42 | b << 21 | c) + 1
        const unsigned a = (unsigned)(k >> 42), b = (unsigned)((k >> 21) & 0x1fffffu), c = (unsigned)(k & 0x1fffffu);
        flag[(a * d1 + b) * d2 + c] = 1;
    }
    __syncthreads();
    // eight points per lane and step: one 16-byte load of ids, one 8-byte store of mask bytes
    const int64_t n8 = n / 8;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n8; j += (int64_t)gridDim.x * 256) {
        const uint4 w = reinterpret_cast<const uint4 *>(ids)[j];
        const unsigned u[4] = {w.x, w.y, w.z, w.w};
        unsigned long long m = 0ull;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned lo = u[q] & 0xffffu, hi = u[q] >> 16;
            m |= (unsigned long long)(lo < (unsigned)nv ? flag[lo] : 0) << (16 * q);
            m |= (unsigned long long)(hi < (unsigned)nv ? flag[hi] : 0) << (16 * q + 8);
        }
        reinterpret_cast<unsigned long long *>(mask)[j] = m;
    }
    if (blockIdx.x == 0)
        for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += 256) {
            const unsigned id = ids[i];
            mask[i] = id < (unsigned)nv ? flag[id] : 0;
        }
}

// box6 (host, nullable): per-axis minima then maxima of a SUPERSET of the rows (e.g. the box of the table the rows were
// filtered from) -- the key frame then needs no pass over the rows and no synchronisation
int density_filter_dev(gsx_ctx *c, const float *x, const float *y, const float *z, int64_t stride, int64_t n, double voxel_size,
                       int64_t min_points, int keep_multi, const float *box6, uint8_t *mask_dev, gsx_density_info *info)
{
    const float voxel = (float)voxel_size;
    if (!(voxel > 0.0f)) GSX_FAIL("density: voxel_size must be > 0");
    GSX_CHECK(timing_begin(c, GSX_T_DENSITY));
    GSX_CHECK(c->scratch5.reserve(sizeof(VoxelFrame) + sizeof(ClusterOut) + 64));
    VoxelFrame *dvf = c->scratch5.as<VoxelFrame>();
    ClusterOut *dco = reinterpret_cast<ClusterOut *>(c->scratch5.as<char>() + 64);
    VoxelFrame hvf;
    bool from_box = box6 != nullptr;
    if (from_box) {
        hvf.ok = 1;
        hvf.pad = 0;
        for (int a = 0; a < 3; ++a) {   // voxel_frame_kernel's arithmetic (IEEE binary32 divide + floor) on the host
            float lo = floorf(box6[a] / voxel), hi = floorf(box6[3 + a] / voxel);
            // a superset box may reach where the rows themselves no longer do (far floaters an earlier filter removed):
            // out of the key range, or wide for nothing -> the frame of the rows themselves, below
            if (!(fabsf(lo) < 1.0e9f) || !(fabsf(hi) < 1.0e9f) || !(box6[a] <= box6[3 + a])) {
                from_box = false;
                break;
            }
            const long long d = (long long)hi - (long long)lo + 1;
            if (d > (1 << 21) - 1) from_box = false;
            hvf.kmin[a] = (int)lo;
            hvf.dim[a] = (int)d;
        }
    }
    if (from_box)
        GSX_HIP(hipMemcpyAsync(dvf, &hvf, sizeof(hvf), hipMemcpyHostToDevice, c->stream));
    else
        GSX_CHECK(compute_frame(c, x, y, z, stride, n, voxel, dvf, &hvf));
    VoxTable t;
    const int64_t dense_cap = std::min<int64_t>(n, n / std::max<int64_t>(min_points, 1) + 1);
    const size_t cap = (size_t)std::max<int64_t>(std::min<int64_t>(dense_cap, CL_MAX + 1), 1);
    GSX_CHECK(alloc_table(c, hvf, n, 16 * cap + 64 + 12 * (size_t)CL_MAX + 64, &t));
    unsigned long long *okeys = reinterpret_cast<unsigned long long *>(t.out_base);
    unsigned *ocnt = reinterpret_cast<unsigned *>(t.out_base + sizeof(unsigned long long) * cap);
    unsigned *okb = ocnt + cap;
    unsigned *ctr = reinterpret_cast<unsigned *>((reinterpret_cast<uintptr_t>(okb + cap) + 15) & ~(uintptr_t)15);
    unsigned long long *kept_a = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ctr) + 32);
    unsigned *kept_b = reinterpret_cast<unsigned *>(kept_a + CL_MAX);
    GSX_HIP(hipMemsetAsync(ctr, 0, 16, c->stream));   // [0] unique, [1] dense, [2] a row outside the caller's box
    // a frame of <= 4096 voxels (configs[2]: 5 x 5 x 5): ONE pass over the rows -- the count kernel parks every point's voxel index in
    // a 2-byte side array the membership pass reads instead of the rows (needs an 8-byte aligned mask)
    uint16_t *ids = nullptr;
    if (dense_frame(&hvf, t) && (reinterpret_cast<uintptr_t>(mask_dev) & 7) == 0) {
        GSX_CHECK(c->vox_ids.reserve(2 * (size_t)n + 64));
        ids = c->vox_ids.as<uint16_t>();
    }
    GSX_CHECK(count_points(c, x, y, z, stride, n, voxel, dvf, t, from_box ? ctr + 2 : nullptr, &hvf, ids));
    const unsigned mp = (unsigned)std::min<int64_t>(std::max<int64_t>(min_points, 0), 0xffffffffll);
    hipLaunchKernelGGL(voxel_collect_kernel, dim3(blocks_for(c, (int64_t)t.tsize, 1024)), dim3(256), 0, c->stream, t.tkeys, t.tkb,
                       t.tcnt, (unsigned)t.tsize, mp, (unsigned)cap, okeys, okb, ocnt, ctr);
    if (ids) {
        hipLaunchKernelGGL((voxel_cluster_kernel<false>), dim3(1), dim3(256), 0, c->stream, okeys, okb, ctr, (unsigned)cap, hvf, keep_multi,
                           kept_a, kept_b, dco);
        hipLaunchKernelGGL(voxel_mask_ids_kernel, dim3(blocks_for(c, n, 8192)), dim3(256), 0, c->stream, ids, n, hvf, kept_a, dco, mask_dev);
    } else
    if (t.wide) {
        hipLaunchKernelGGL((voxel_cluster_kernel<true>), dim3(1), dim3(256), 0, c->stream, okeys, okb, ctr, (unsigned)cap, hvf, keep_multi,
                           kept_a, kept_b, dco);
        hipLaunchKernelGGL((voxel_mask_dev_kernel<true>), dim3(blocks_for(c, n, 1024)), dim3(256), 0, c->stream, x, y, z, stride, n, voxel,
                           hvf, kept_a, kept_b, dco, mask_dev);
    } else {
        hipLaunchKernelGGL((voxel_cluster_kernel<false>), dim3(1), dim3(256), 0, c->stream, okeys, okb, ctr, (unsigned)cap, hvf, keep_multi,
                           kept_a, kept_b, dco);
        hipLaunchKernelGGL((voxel_mask_dev_kernel<false>), dim3(blocks_for(c, n, 1024)), dim3(256), 0, c->stream, x, y, z, stride, n, voxel,
                           hvf, kept_a, kept_b, dco, mask_dev);
    }
    GSX_HIP(hipGetLastError());
    GSX_CHECK(timing_end(c, GSX_T_DENSITY));
    ClusterOut h;
    GSX_HIP(hipMemcpyAsync(&h, dco, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));      // the call's one synchronisation
    if (from_box && h.oob)   // the box was not a superset of the rows: once more with the frame of the rows themselves
        return density_filter_dev(c, x, y, z, stride, n, voxel_size, min_points, keep_multi, nullptr, mask_dev, info);
    info->status = (int32_t)h.status;
    info->n_unique = h.n_unique;
    info->n_dense = h.n_dense;
    info->n_kept_voxels = h.n_kept_voxels;
    info->kept_clusters = h.kept_clusters;
    info->largest = h.largest;
    return 0;
}

}  // namespace gsx
