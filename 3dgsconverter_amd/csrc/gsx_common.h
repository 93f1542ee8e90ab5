// gsx_common.h -- context, error plumbing, workspace arena and timing for libgsx_hip.so.
// gfx950 only: wave64 is hard-coded throughout.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gsx_hip.h"

namespace gsx {

constexpr int WAVE = 64;

void set_error(const char *fmt, ...);

#define GSX_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            gsx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,  \
                           __LINE__);                                                         \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

#define GSX_CHECK(expr)            \
    do {                           \
        int r__ = (expr);          \
        if (r__ != 0) return r__;  \
    } while (0)

#define GSX_FAIL(...)                \
    do {                             \
        gsx::set_error(__VA_ARGS__); \
        return 1;                    \
    } while (0)

// A grow-only device buffer owned by the context (no hipMalloc inside a timed step once warm).
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T>
    T *as() const { return reinterpret_cast<T *>(p); }
};

// Buffers of one grid-binned KNN run (sor_grid.hip).  Refinement levels > 0 re-run the pipeline on the
// sub-cloud around the cells a uniform grid cannot resolve; each level owns a full set.
constexpr int KNN_MAX_LEVELS = 5;
struct KnnWs {
    DevBuf packed;      // float4[n_ref]  (brute: original order; grid: cell-sorted refs)
    DevBuf qsorted;     // float4[q_count] cell-sorted queries when q != all refs
    DevBuf bucketpts;   // float4[n_ref] points grouped by bucket (between the two sort levels)
    DevBuf bkcnt;       // u32 bucket sizes | starts | cursors
    DevBuf cellstart;   // u32[cap+1]
    DevBuf qcellstart;
    DevBuf gridparams;  // GridParams + work counters
    DevBuf bboxpart;    // float[7 * blocks]
    DevBuf faillist;    // u32[q_count]
    DevBuf extraitems;  // uint2[q_count/64 + 64]: (brick, first query) of every batch beyond a brick's first
    // adaptive refinement (level L -> L+1)
    DevBuf deferred;    // u32[nbricks]: bricks whose neighbourhood is too populated for this level's cells
    DevBuf cellflag;    // u8[ncells]: 1 = cell belongs to the sub-cloud, 2 = ... of a deferred brick (its points are queries)
    DevBuf subxyz;      // float[3 * n_sub] SoA sub-cloud
    DevBuf submap;      // u32[2 * n_sub]: original index | this level's sorted index (bit 31 of the first: is a query)
    DevBuf submean;     // float[n_sub] mean distances computed at the next level
    DevBuf subkth;      // double[n_sub] (k+1)-th squared distance computed at the next level
    DevBuf heavylist;   // u32[q_count] sorted indices of the knn_ring queries handed to knn_heavy
    DevBuf heavypart;   // double[batch * chunks * KCAP] per-chunk partial top lists
    DevBuf probe;       // u32[G^3 + 32]: density probe of a sub-cloud (counts of a coarse grid, point-weighted histogram of them)
    uint64_t refined_total = 0;  // host-side: points gathered into sub-clouds by the current call
    void release_all()
    {
        DevBuf *all[] = {&packed, &qsorted, &bucketpts, &bkcnt, &cellstart, &qcellstart, &gridparams, &bboxpart,
                         &faillist, &extraitems, &deferred, &cellflag, &subxyz, &submap, &submean, &subkth,
                         &heavylist, &heavypart, &probe};
        for (auto b : all) b->release();
    }
};

constexpr int GSX_TREE_UNSUITABLE = 2;   // launch_knn_tree(guard = true): the cloud exhausts the key resolution, nothing was computed

// Buffers of the Morton-tree KNN (sor_tree.hip)
struct TreeWs {
    DevBuf keys[2];     // u64[n]: Morton keys (unsorted | sorted)
    DevBuf vals[2];     // u32[n]: original indices (identity | key order)
    DevBuf refs;        // float4[n] points in key order
    DevBuf samples;     // u64[n/32 + 1]: the last key of every 32-key block
    DevBuf blockboxes;  // float4[2 * (n/64 + 1)]: tight box (minima, maxima) of every 64 consecutive points of the key order
    DevBuf flags;       // u8[n]: leaf bit level | 0x80 on a leaf's first point
    DevBuf tilecnt, tileoff;   // u32[tiles]: leaves starting in a tile, exclusive scan
    DevBuf leafstart;   // u32[leaves + 1]
    DevBuf leafbl;      // u8[leaves]
    DevBuf faillist;    // u32[n] sorted indices of the queries knn_leaf could not certify
    DevBuf failbound;   // double[n]: their k-th distance bound (>= 0) or minus the squared radius to start with
    DevBuf bboxpart;    // float[7 * blocks]
    DevBuf params;      // TreeParams
    DevBuf temp;        // rocprim temporary storage
    void release_all()
    {
        DevBuf *all[] = {&keys[0], &keys[1], &vals[0], &vals[1], &refs, &samples, &blockboxes, &flags, &tilecnt, &tileoff, &leafstart, &leafbl,
                         &faillist, &failbound, &bboxpart, &params, &temp};
        for (auto b : all) b->release();
    }
};

// Buffers of one multi-GPU slab step (dist_slab.hip: gsx_sor_slab_step_dev); grow-only like the rest
struct SlabWs {
    DevBuf plan_in;    // 8 words (global box) + 4096 u32 per rank (all-gathered histograms)
    DevBuf hist;       // u32[4096]: this rank's histogram (send side of the all-gather)
    DevBuf small;      // u32 cursor[32] | u32 uncertain[2] (all-reduced as one int64) | f32 stats[4] | i64 pack table
    DevBuf send;       // float[3 * n_send]: rows grouped by destination slab (own rows, then halo copies)
    DevBuf send_src;   // u32[n_local]: local index of every own row of the send buffer
    DevBuf slab;       // float[3 * (n_own + n_halo)]: what this rank received
    DevBuf md_slab;    // f32[n_own]
    DevBuf kth;        // f64[n_own]
    DevBuf ret;        // f32[n_local]: mean distances in send order
    DevBuf md;         // f32[n_local + 8192 + 4]: ... in index order (+ the right neighbour's head elements)
    DevBuf md_stats;   // aligned copy for piece sums when the first own piece starts off a 16-byte boundary
    DevBuf pieces, allpieces, packed;   // numpy's 8192-element piece sums: mine | all-gathered (padded) | packed
    DevBuf mask;       // u8[n_local] when the caller brings no buffer
    void *host = nullptr;   // pinned staging of plan_in (the step's one host synchronisation)
    size_t host_cap = 0;
    void release_all()
    {
        DevBuf *all[] = {&plan_in, &hist, &small, &send, &send_src, &slab, &md_slab, &kth, &ret, &md, &md_stats, &pieces, &allpieces, &packed, &mask};
        for (auto b : all) b->release();
        if (host) (void)hipHostFree(host);
        host = nullptr;
        host_cap = 0;
    }
};

struct TimingSlot {
    std::vector<hipEvent_t> ev;  // pairs: start, stop
    size_t used = 0;
    uint64_t launches = 0;       // resolved so far
    double total_ms = 0.0;
};

}  // namespace gsx

struct gsx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t owned_stream = nullptr;   // gsx_ctx_own_stream: a non-blocking stream created for (and destroyed with) the context
    bool timing = false;
    unsigned timing_mask = 0xffffffffu;  // slots that record events when timing is on (an event pair costs ~8 us of stream time)
    gsx::TimingSlot slots[GSX_T_SLOTS];
    int num_cu = 256;
    void *comm = nullptr;  // gsx_comm (RCCL communicator, dist_slab.hip) or null

    // tunables
    double grid_points_per_cell = 0.0;  // 0 = auto: 0.47 * (k + 1), see launch_knn_grid
    int64_t brute_below = 2048;
    int debug_skip = 0;  // profiling ablations of knn_brick (never set by the product path)
    int adaptive = 0;    // 1: bricks too populated for the grid are re-run on a finer grid (one host sync per call)
    int defer_words = 64;
    int last_knn_algo = 0;   // algorithm the last KNN call of this context ended up in (GSX_KNN_*)
    int tree = 1;        // adaptive mode: 1 = the Morton-tree path (sor_tree.hip, no host round trips), 0 = level-by-level grid refinement
    int kmeans_mfma = 1;  // K-Means assign for D in {9,24,45}, K >= 64: 1 = matrix-core filter + exact certificate, 0 = packed-f32 VALU scan
    int tree_leaf_cap = 0;         // points per leaf of the Morton-tree path: 0 = by k (sor_tree.hip: tree_leaf_cap_for), else 64 ... 256 (A/B)
    int tree_cand_limit = 0;       // candidates per 64 points of leaf capacity above which a leaf's queries go to the per-query kernels (0 = 4096; A/B)
    double tree_near_cell = 0.5;   // knn_tree_near: edge of the cover's cells as a fraction of the ball's radius (A/B)
    double tree_scale = 0.0;   // > 0: the Morton tree's fine cell edge is (extent / 2^21) x this (A/B of the leaf-shape rule); 0 = chosen by the density probe
    int km_exact_blocks = 4;   // workgroups per CU of kmeans_assign_exact_list (one uncertified point per workgroup and round: A/B)
    int km_seg_rows = 32;   // rows of the label-sorted permutation one wave of kmeans_segment_sum takes (64 / 32 / 16: A/B)
    int km_group_mb = 0;   // gsx_kmeans_lloyd_batch_dev: problems are run in groups whose rows fit this many MB (0: all at once) -- a group's
                           // rows then stay in the 256 MB Infinity Cache across its iterations (round 5, A/B)
    int km_small_wgs = 0;      // workgroups per CU of those shapes (0: 3 resp. 5; A/B)
    int kmeans_cs_small = 1;   // ... with 4-wave / 2-wave workgroups for K <= 256 / K <= 64 (0: the 16-wave shape for every K; A/B)
    int kmeans_cs = 1;   // centroid-stationary matrix-core assign for K <= 1024 (0: the streaming kernel; A/B)
    int ring_fast = 1;   // knn_ring_fast before knn_ring (0: A/B only)
    int phase2_net = 1;  // knn_brick phase 2: 1 = sorting-network block selection (TopNet), 0 = per-candidate bubble insert (A/B only)
    int filter_mfma = 1; // knn_brick phase 1: 1 = bf16-split MFMA filter for batches whose mask words fit LDS, 0 = scalar-load f32 VALU filter only

    // SOR workspace: one KnnWs per refinement level of the KNN grid (level 0 = the whole cloud)
    gsx::KnnWs ws[gsx::KNN_MAX_LEVELS];
    gsx::TreeWs tree_ws;     // Morton-tree KNN workspace
    gsx::SlabWs slab_ws;     // multi-GPU slab step
    gsx::DevBuf commscratch; // gsx_comm_barrier's word
    gsx::DevBuf devflags;    // u32[16]: device-side error word (bit 0: non-finite coordinates), read by gsx_ctx_check
    gsx::DevBuf vox_ids;     // u16[n]: dense voxel index of every point between the two passes of the density filter (density.hip)
    gsx::DevBuf nzmask;      // one 64-bit word: gsx_fields_nonzero_dev's result before it is copied out (cply.hip)
    gsx::DevBuf statspart;   // float chunk sums
    gsx::DevBuf scratch;     // host-API staging
    gsx::DevBuf scratch2;
    gsx::DevBuf scratch3;
    gsx::DevBuf scratch4;
    gsx::DevBuf scratch5;
};

namespace gsx {

// RAII-less timing helper: call begin/end around a group of launches on ctx->stream.
int timing_begin(gsx_ctx *ctx, int slot);
int timing_end(gsx_ctx *ctx, int slot);

inline int div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace gsx
