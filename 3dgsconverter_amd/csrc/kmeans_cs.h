// kmeans_cs.h -- centroid-stationary matrix-core assign (included by kmeans.hip after kmeans_assign_mfma_kernel).
//
// SQ counters of kmeans_assign_mfma_kernel (tools/run_km_pmc.sh) showed it bound by operand delivery: every wave streams
// all K/32 centroid operand tiles through its CU's vector L1 (3.6 MB per CU and launch) and its matrix-core and VALU work
// hardly overlap.  Here a 16-wave workgroup keeps ALL centroid operands in registers (wave w: tiles 2w, 2w+1 = 48 VGPRs,
// loaded once per workgroup), the points of a 128-point block are split into bf16 operand words ONCE, into LDS, and every
// wave runs the block's four point tiles past its two centroid tiles.  The per-wave (best, second, index) of each point
// meet in LDS.  Same MFMA sequence per (centroid tile, point tile) pair, same tournament, same certificate: the labels
// and the uncertain list are those of kmeans_assign_mfma_kernel (gpu_ops.py:57-73 is what both replace).
#pragma once

// Round 5: profiling builds only (results become wrong).  GSX_KM_ABL bit 0: no merge of the 16 per-wave views / no label store,
// bit 1: the next block is neither fetched nor split (every block computes on the first block's operands).
#ifndef GSX_KM_ABL
#define GSX_KM_ABL 0
#endif
#ifndef GSX_KM_ML   // lanes per point in the merge of the per-wave views: 1 (rounds 2-4), 2, 4 or 8
#define GSX_KM_ML 1
#endif
constexpr int KM_CS_WAVES = 16;                // 8 waves x 4 centroid tiles: equal (profiles/r02_variants.txt)
constexpr int KM_CS_CT = 32 / KM_CS_WAVES;    // centroid tiles per wave -> K <= KM_CS_WAVES * KM_CS_CT * 32 = 1024
constexpr int KM_CS_PTILES = 4;              // 32-point tiles per block
constexpr int KM_CS_BLOCK = 32 * KM_CS_PTILES;
constexpr int KM_CS_ML = GSX_KM_ML;

// Round 5: the shape is a template argument.  WAVES x CT centroid tiles = K / 32; the palette's K per chunk is 1024 for
// --compression_level 0-3, 256 for 4-6, 64 for 7-9 (sog.py:513-529), and the 16-wave workgroup ran the smaller two with 12 resp.
// 15 of its waves idle at every barrier: 1.35 / 1.24 ms per iteration of the 10M-splat palette against 2.72 at K = 1024.
// K <= 256: 4 waves x 2 tiles, K <= 64: 2 waves x 1 tile, both on 64-point blocks (half the LDS: four resp. five workgroups
// per CU keep the row stream going).
template <int D, int WAVES = KM_CS_WAVES, int CT = KM_CS_CT, int PTILES = KM_CS_PTILES>
__global__ __launch_bounds__(64 * WAVES) void kmeans_assign_mfma_cs_kernel(const float *__restrict__ data, int64_t n,
                                                                                const ku32x4 *__restrict__ opnd, int ktiles,
                                                                                const float *__restrict__ cmax2,
                                                                                int32_t *__restrict__ labels,
                                                                                unsigned *__restrict__ unc_list,
                                                                                unsigned *__restrict__ unc_count, KmBatch kb)
{
    constexpr int DP = km_dp(D), NS = DP / 16;
    constexpr int AW = NS * 2 * 64;
    constexpr int BLOCK = 32 * PTILES;
    {
        const int64_t r0 = km_problem_rows(kb, n);             // problem blockIdx.y: its rows, its centroid operands
        data += r0 * D;
        labels += r0;
        unc_list += r0;
        opnd += (size_t)blockIdx.y * ktiles * AW;
        cmax2 += (size_t)blockIdx.y * KM_META_WORDS;
        unc_count += (size_t)blockIdx.y * KM_META_WORDS;
    }
    // double buffered: block b+1 is fetched and split while block b runs through the matrix cores, and block b's
    // per-wave results are merged while block b+1 runs
    __shared__ ku32x4 s_x[2][PTILES][NS][2][64];   // point operand words: (tile, slice, hi/lo, lane)
    __shared__ float s_part[2][BLOCK][NS * 2];     // |x|^2 by operand word (summed in a fixed order by the merge)
    // (rows padded by two words: the merge reads view 2 sub + u of point p with 8 lanes per point -- bank 4 sub + 2 u + p)
    __shared__ float s_best[2][WAVES][BLOCK + 2], s_second[2][WAVES][BLOCK + 2];
    __shared__ int s_idx[2][WAVES][BLOCK + 2];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float nc2 = *cmax2, nc = __builtin_sqrtf(nc2);
    // this wave's centroid operands, resident for the whole launch
    ku32x4 a[CT][NS][2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int t = min(wv * CT + ct, ktiles - 1);
#pragma unroll
        for (int j = 0; j < NS; ++j)
#pragma unroll
            for (int v = 0; v < 2; ++v) a[ct][j][v] = opnd[(size_t)t * AW + (size_t)(j * 2 + v) * 64 + lane];
    }
    const int64_t nblocks = (n + BLOCK - 1) / BLOCK;
    // Round 5: a block's rows are ONE contiguous span of memory (row-major n x D), so every wave fetches ITS rows (BLOCK / WAVES
    // of them) as consecutive dwords -- two cache lines per wave instruction -- parks them in its own strip of LDS and cuts its
    // items (8 dimensions of one point = one operand word pair) out of that: only the wave itself reads the strip, no workgroup
    // barrier.  (Before, lane l loaded 8 dwords of row l: every wave instruction touched 64 different cache lines, 1.9 M cycles
    // of address processing per CU and iteration of the 10M-splat palette -- hidden behind the matrix cores at K = 1024, the
    // whole time at K <= 256.)  Strip reads are conflict-free: consecutive lanes read consecutive rows, D = 45 dwords apart.
    constexpr int RPW = BLOCK / WAVES, FPW = RPW * D, LPL = (FPW + 63) / 64;   // rows / dwords per wave, dwords per lane
    constexpr int NITEMS = RPW * NS * 2, IPT = (NITEMS + 63) / 64;
    static_assert(BLOCK % WAVES == 0, "rows per wave");
    __shared__ float s_raw[WAVES][FPW];
    float raw[LPL];
    auto fetch = [&](int64_t blk) __attribute__((always_inline)) {   // global loads only (consumed after the compute phase)
        const int64_t base = blk * BLOCK;
        const int rows = (int)(n - base < BLOCK ? n - base : BLOCK);
        const int mine = min(max(rows - wv * RPW, 0), RPW) * D;   // dwords of this wave's rows that exist
        const float *__restrict__ src = data + (base + (int64_t)wv * RPW) * D;
#pragma unroll
        for (int q = 0; q < LPL; ++q) {
            const int idx = lane + 64 * q;
            raw[q] = idx < mine ? src[idx] : 0.0f;
        }
    };
    auto split_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < LPL; ++q) {
            const int idx = lane + 64 * q;
            if (idx < FPW) s_raw[wv][idx] = raw[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
            const int item = lane + 64 * q;
            if (item >= NITEMS) continue;
            const int rl = item % RPW, rest = item / RPW, ij = rest % NS, half = rest / NS;
            const int r = wv * RPW + rl, ipt = r >> 5, il = (half << 5) | (r & 31);
            float v[8];
            float acc2 = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int d = 16 * ij + 8 * half + i;
                v[i] = d < D ? s_raw[wv][rl * D + d] : 0.0f;
                acc2 = __builtin_fmaf(v[i], v[i], acc2);
            }
            ku32x4 hi, lo;
            km_split8(v, hi, lo);
#pragma unroll
            for (int i = 0; i < 8; ++i) {   // 1.0 against the three |c|^2 pieces, in the high operand only
                const int d = 16 * ij + 8 * half + i;
                if (d >= D && d < D + 3) hi[i >> 1] |= 0x3f80u << ((i & 1) * 16);
            }
            s_x[buf][ipt][ij][0][il] = hi;
            s_x[buf][ipt][ij][1][il] = lo;
            s_part[buf][r][ij * 2 + half] = acc2;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (the strip is rewritten for the next block only after these reads)
        __builtin_amdgcn_wave_barrier();
    };
    if ((int64_t)blockIdx.x < nblocks) {
        fetch(blockIdx.x);
        split_store(0);
    }
    // every load so far (the centroid operands above all) has landed: without this the compiler keeps `s_waitcnt vmcnt(6)`
    // in front of the MFMA chain for the first trip's sake, which in every later trip waits for the NEXT block's prefetch
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    __syncthreads();
    int cur = 0;
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x, cur ^= 1) {
        const int64_t base = blk * BLOCK;
        const int rows = (int)(n - base < BLOCK ? n - base : BLOCK);
        const bool has_next = blk + gridDim.x < nblocks && !(GSX_KM_ABL & 2);
        if (has_next) fetch(blk + gridDim.x);
        // ---- every point tile of the block against this wave's centroid tiles
#pragma unroll 1
        for (int pt = 0; pt < PTILES; ++pt) {
            if (pt * 32 >= rows) break;   // (a ragged last block)
            ku32x4 xh[NS], xl[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                xh[j] = s_x[cur][pt][j][0][lane];
                xl[j] = s_x[cur][pt][j][1][lane];
            }
            float best = __builtin_inff(), second = __builtin_inff();
            int btile = 0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int t = wv * CT + ct;
                if (t >= ktiles) break;   // wave-uniform
                kf32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const kbf16x8 ah = __builtin_bit_cast(kbf16x8, a[ct][j][0]), al = __builtin_bit_cast(kbf16x8, a[ct][j][1]);
                    const kbf16x8 bh = __builtin_bit_cast(kbf16x8, xh[j]), bl = __builtin_bit_cast(kbf16x8, xl[j]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                }
                // the two smallest of the running pair and the 16 tagged values, one value at a time: with best <= second,
                // the new second is the MEDIAN of (best, second, v) and the new best min(best, v) -- 2 ops per value against
                // 2.6 for a tournament of pairs (which it replaced: 64.75 against 65.06 ms per scene on one lane); the same two values
                // come out
                const float best_in = best;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float v = __uint_as_float((__float_as_uint(acc[q]) & ~0xfu) | (unsigned)q);
                    second = km_med3(best, second, v);
                    best = km_min(best, v);
                }
                btile = best < best_in ? t : btile;
            }
            // the two half-waves hold the same points (different centroid rows): merge, publish this wave's view
            const int rb = (int)(__float_as_uint(best) & 0xfu);
            const int bidx = 32 * btile + (rb & 3) + 8 * (rb >> 2) + 4 * (lane >> 5);
            const float b2 = __shfl_xor(best, 32), s2 = __shfl_xor(second, 32);
            const int i2 = __shfl_xor(bidx, 32);
            if (lane < 32) {
                s_best[cur][wv][pt * 32 + lane] = fminf(best, b2);
                s_second[cur][wv][pt * 32 + lane] = fminf(fmaxf(best, b2), fminf(second, s2));
                s_idx[cur][wv][pt * 32 + lane] = b2 < best ? i2 : bidx;
            }
        }
        if (has_next) split_store(cur ^ 1);
        __syncthreads();   // this block's views are complete, the next block's operand words are in place
        // The merge of the 16 per-wave views: KM_CS_ML lanes per point, 16 / KM_CS_ML views per lane, log2(KM_CS_ML) shuffle
        // rounds, by the first 128 x KM_CS_ML threads.  (The min / second-min of a multiset does not depend on the merge order;
        // an exact tie of the two smallest makes the point uncertain whichever index is kept.)  Round 5 measured where this
        // sits: WITHOUT it (GSX_KM_ABL=1) the kernel is a quarter faster -- the merging waves start the next block late and the
        // others wait for them at its barrier -- but spreading it over all 16 waves (KM_CS_ML = 8) is slower still (41.4 against
        // 39.5 ms per palette): every wave then pays the LDS round trips.  profiles/r05_variants.txt.
        {
            constexpr int ML = KM_CS_ML, VPL = WAVES / ML;
            if ((int)threadIdx.x >= BLOCK * ML) continue;   // (whole waves: BLOCK * ML is a multiple of 64)
            const int p = (int)threadIdx.x / ML, sub = (int)threadIdx.x % ML;
            const int nw = min(WAVES, (ktiles + CT - 1) / CT);
            float mb = __builtin_inff(), ms = __builtin_inff();
            int mi = 0;
            if (p < rows) {
#pragma unroll
                for (int u = 0; u < VPL; ++u) {
                    const int w = VPL * sub + u;
                    if (w < nw) {
                        const float b = s_best[cur][w][p], sc = s_second[cur][w][p];
                        const int ix = s_idx[cur][w][p];
                        const float nsec = fminf(fmaxf(mb, b), fminf(ms, sc));
                        mi = b < mb ? ix : mi;
                        mb = fminf(mb, b);
                        ms = nsec;
                    }
                }
            }
#pragma unroll
            for (int off = 1; off < ML; off <<= 1) {
                const float ob = __shfl_xor(mb, off), os = __shfl_xor(ms, off);
                const int oi = __shfl_xor(mi, off);
                const float nsec = fminf(fmaxf(mb, ob), fminf(ms, os));
                // on an exact tie both lanes keep the index of the lower view group (such a point is uncertain anyway)
                const bool take = ob < mb || (ob == mb && (sub & off) != 0);
                mi = take ? oi : mi;
                mb = fminf(mb, ob);
                ms = nsec;
            }
            if (sub != 0 || p >= rows || (GSX_KM_ABL & 1)) continue;
            float nx2 = 0.0f;
#pragma unroll
            for (int q = 0; q < NS * 2; ++q) nx2 += s_part[cur][p][q];
            const float nx = __builtin_sqrtf(nx2);
            const float E = 6.1035156e-5f * (nx * nc + nc2) + 3.8146973e-6f * nx2 * (1.0f + 1e-6f);
            const bool sure = (ms - mb) > 2.0f * E;
            if (sure) labels[base + p] = mi;
            else unc_list[atomicAdd(unc_count, 1u)] = (unsigned)(base + p);
        }
    }
}

