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

template <int D>
__global__ __launch_bounds__(64 * KM_CS_WAVES) void kmeans_assign_mfma_cs_kernel(const float *__restrict__ data, int64_t n,
                                                                                const ku32x4 *__restrict__ opnd, int ktiles,
                                                                                const float *__restrict__ cmax2,
                                                                                int32_t *__restrict__ labels,
                                                                                unsigned *__restrict__ unc_list,
                                                                                unsigned *__restrict__ unc_count, KmBatch kb)
{
    constexpr int DP = km_dp(D), NS = DP / 16;
    constexpr int AW = NS * 2 * 64;
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
    __shared__ ku32x4 s_x[2][KM_CS_PTILES][NS][2][64];   // point operand words: (tile, slice, hi/lo, lane)
    __shared__ float s_part[2][KM_CS_BLOCK][NS * 2];     // |x|^2 by operand word (summed in a fixed order by the merge)
    // (rows padded by two words: the merge reads view 2 sub + u of point p with 8 lanes per point -- bank 4 sub + 2 u + p)
    __shared__ float s_best[2][KM_CS_WAVES][KM_CS_BLOCK + 2], s_second[2][KM_CS_WAVES][KM_CS_BLOCK + 2];
    __shared__ int s_idx[2][KM_CS_WAVES][KM_CS_BLOCK + 2];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float nc2 = *cmax2, nc = __builtin_sqrtf(nc2);
    // this wave's centroid operands, resident for the whole launch
    ku32x4 a[KM_CS_CT][NS][2];
#pragma unroll
    for (int ct = 0; ct < KM_CS_CT; ++ct) {
        const int t = min(wv * KM_CS_CT + ct, ktiles - 1);
#pragma unroll
        for (int j = 0; j < NS; ++j)
#pragma unroll
            for (int v = 0; v < 2; ++v) a[ct][j][v] = opnd[(size_t)t * AW + (size_t)(j * 2 + v) * 64 + lane];
    }
    const int64_t nblocks = (n + KM_CS_BLOCK - 1) / KM_CS_BLOCK;
    // one item = 8 dimensions of one point = one operand word pair; 768 items per block, IPT per thread
    constexpr int NITEMS = KM_CS_PTILES * NS * 64, NT = 64 * KM_CS_WAVES, IPT = (NITEMS + NT - 1) / NT;
    float v[IPT][8];
    auto fetch = [&](int64_t blk) __attribute__((always_inline)) {   // global loads only (consumed after the compute phase)
        const int64_t base = blk * KM_CS_BLOCK;
        const int rows = (int)(n - base < KM_CS_BLOCK ? n - base : KM_CS_BLOCK);
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
            const int item = (int)threadIdx.x + q * NT;
            if (item >= NITEMS) continue;
            const int il = item & 63, ij = (item >> 6) % NS, ipt = item / (64 * NS);
            const int r = ipt * 32 + (il & 31);
            const int rr = r < rows ? r : rows - 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int d = 16 * ij + 8 * (il >> 5) + i;
                v[q][i] = d < D ? data[(base + rr) * D + d] : 0.0f;
            }
        }
    };
    auto split_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
            const int item = (int)threadIdx.x + q * NT;
            if (item >= NITEMS) continue;
            const int il = item & 63, ij = (item >> 6) % NS, ipt = item / (64 * NS);
            float acc2 = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc2 = __builtin_fmaf(v[q][i], v[q][i], acc2);
            ku32x4 hi, lo;
            km_split8(v[q], hi, lo);
#pragma unroll
            for (int i = 0; i < 8; ++i) {   // 1.0 against the three |c|^2 pieces, in the high operand only
                const int d = 16 * ij + 8 * (il >> 5) + i;
                if (d >= D && d < D + 3) hi[i >> 1] |= 0x3f80u << ((i & 1) * 16);
            }
            s_x[buf][ipt][ij][0][il] = hi;
            s_x[buf][ipt][ij][1][il] = lo;
            s_part[buf][ipt * 32 + (il & 31)][ij * 2 + (il >> 5)] = acc2;
        }
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
        const int64_t base = blk * KM_CS_BLOCK;
        const int rows = (int)(n - base < KM_CS_BLOCK ? n - base : KM_CS_BLOCK);
        const bool has_next = blk + gridDim.x < nblocks && !(GSX_KM_ABL & 2);
        if (has_next) fetch(blk + gridDim.x);
        // ---- every point tile of the block against this wave's centroid tiles
#pragma unroll 1
        for (int pt = 0; pt < KM_CS_PTILES; ++pt) {
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
            for (int ct = 0; ct < KM_CS_CT; ++ct) {
                const int t = wv * KM_CS_CT + ct;
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
            constexpr int ML = KM_CS_ML, VPL = KM_CS_WAVES / ML;
            if ((int)threadIdx.x >= KM_CS_BLOCK * ML) continue;   // (whole waves: KM_CS_BLOCK * ML is a multiple of 64)
            const int p = (int)threadIdx.x / ML, sub = (int)threadIdx.x % ML;
            const int nw = min(KM_CS_WAVES, (ktiles + KM_CS_CT - 1) / KM_CS_CT);
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
