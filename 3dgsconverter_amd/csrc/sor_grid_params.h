// sor_grid_params.h -- device-resident description of the KNN cell grid + work counters.
#pragma once

namespace gsx {

struct GridParams {
    float ox, oy, oz, inv_h;
    float h;
    int nx, ny, nz;
    int ncells;          // padded: bk_count * bk_cells (cell index space, bucket-major)
    int bk_g;            // a BUCKET is bk_g x bk_g complete x-rows of cells (the unit of the 2-level sort)
    int bk_ny, bk_nz;    // buckets along y and z
    int bk_count;        // bk_ny * bk_nz
    int bk_cells;        // cells per bucket = bk_g * bk_g * nx
    int nbx, nby, nbz;
    int nbricks;
    int part_lo, part_hi;  // bricks [part_lo, part_hi) are this call's share (multi-GPU: one share per rank)
    int defer_words;       // > 0: a brick whose neighbourhood holds more than this many 32-candidate words (or
                           // batches x words > 2x this) is not searched at this level but deferred to a finer grid
    int bdx, bdy, bdz;  // brick size in cells: (2,2,2), (2,2,1), (2,1,1) or (1,1,1)
    int debug_skip;     // profiling only (results become wrong): 1 = skip phase 2, 2 = skip phase 1, 4 = skip epilogue
    float tau1;      // f32 filter bound for r1sq
    double r1sq;     // (h' * (1 - 1e-3))^2, h' = 1/inv_h
    double hprime;   // 1/inv_h
    // zeroed by grid_params_kernel every call
    unsigned fail_count;
    unsigned exhaustive_count;
    unsigned extra_count;   // (brick, batch >= 1) work items appended by the first knn_brick pass
    unsigned bad_input;     // 1: non-finite coordinates -- the KNN kernels do nothing, the host reports an error
    unsigned deferred_count;  // bricks appended to the deferred list by knn_brick
    unsigned sub_count;       // points of the sub-cloud gathered around the deferred bricks
    unsigned sub_queries;     // ... of which queries (points of the deferred bricks themselves)
    unsigned refined_count;   // queries whose result came from a finer level
    float qb_lo[3], qb_hi[3];  // bounding box of the points of the deferred bricks (the finer level's queries)
    int heavy_limit;          // > 0: a knn_ring query whose ring holds more candidates than this is not scanned by its
                              // single wave but handed to knn_heavy (the whole chip scans the whole cloud for it)
    unsigned heavy_count;
    unsigned ring2_count;     // queries knn_ring_fast left to knn_ring (its second list)
    unsigned brick_ctr[8 * 32];   // dynamic-tail counters, one per XCD, separate cache lines
    unsigned extra_ctr[8 * 32];
    unsigned ring_ctr[8 * 32];
    unsigned ringf_ctr[8 * 32];
    // arrival tickets of the kernels whose last workgroup finishes the job of a former one-workgroup launch
    // (bbox_partial -> grid parameters, bucket_hist -> bucket scan); never touched by grid_params, self-resetting
    unsigned ticket_bbox, ticket_hist;
    // multi-GPU slab step: the certificate of csrc/dist_slab.hip evaluated where the k-th distance is produced (instead
    // of an 8-byte array written here and read back by a kernel of its own).  cert_axis < 0: off
    int cert_axis;
    float cert_lo, cert_hi;      // planes between which this rank holds EVERY point of the cloud (+-inf at the cloud's ends)
    unsigned *cert_count;        // number of queries whose k-th neighbour might lie beyond them
};

#ifdef __HIPCC__
// every kernel of the grid path hands a finished query's (k+1)-th squared distance (the query itself included) here
__device__ __forceinline__ void kth_emit(GridParams *gp, double *kth_out, int idx, double kth, float qx, float qy, float qz)
{
    if (!kth_out) return;   // the single-GPU path: nothing is loaded, nothing is written
    const int axis = gp->cert_axis;
    if (axis >= 0) {
        const double v = (double)(axis == 0 ? qx : (axis == 1 ? qy : qz));
        // margin: the halo membership test was made in f32 on the same coordinates -- exact; 1e-6 relative for the
        // plane arithmetic itself (the rule of slab_certify_kernel, which the tree path still uses)
        const double d = fmin(v - (double)gp->cert_lo, (double)gp->cert_hi - v) * (1.0 - 1e-6);
        if (!(kth <= d * d)) atomicAdd(gp->cert_count, 1u);
    } else {
        kth_out[idx] = kth;
    }
}
#endif

// launch_knn_slab's optional extras (csrc/dist_slab.hip -> csrc/sor_grid.hip)
// a bounding box the caller already knows (the all-reduced global box, cut to the slab's bins along the partition
// axis): the box pass over the rows is skipped
struct KnownBox {
    const float *b7;     // device: max over the cloud of (-x,-y,-z,x,y,z), non-finite flag
    int axis;            // along this axis the rows lie in [lo, hi] instead
    float lo, hi;
};
struct SlabKnn {         // what launch_knn_slab's caller may add (both optional)
    int cert_axis = -1;
    float cert_lo = 0.0f, cert_hi = 0.0f;
    unsigned *cert_count = nullptr;
    KnownBox box{nullptr, 0, 0.0f, 0.0f};
};

// Work distribution shared by knn_brick / knn_ring (device): static stride + a small dynamic tail.
// XCD y owns the contiguous item range [n*y/8, n*(y+1)/8) (its L2 then sees a compact slab of
// the sorted array); the waves whose home is y (blockIdx % 8 -- a placement HINT only, any mapping
// is correct) stride through the first ~85 % of that range WITHOUT atomics and pull the rest from
// a per-XCD counter, which evens out the finish times (a static-only split ends with the
// slowest wave: +10 % at 10M splats).  History: a device-wide atomic work counter saturates at
// ~88 dequeues/us on MI355X; with 157k bricks at 10M splats the dequeues alone cost 1.0 of the
// kernel's 3.1 ms (ablation in profiles/r01_ablate_knn_brick.log), and eight counters 128 B
// apart did not help.  Every launched workgroup must be resident (grid sized from the occupancy
// query), otherwise the static share of a late workgroup would start late.
struct WorkQueue {
    unsigned *ctr;     // this XCD's tail counter
    int next, static_end, end, stride;
};

#ifdef __HIPCC__
__device__ __forceinline__ void wq_init(WorkQueue &q, unsigned *ctr8x32, int n, int waves_per_block)
{
    const int y = (int)(blockIdx.x & 7);
    const int lo = (int)(((long long)n * y) / 8), hi = (int)(((long long)n * (y + 1)) / 8);
    const int wl = (int)(blockIdx.x >> 3) * waves_per_block + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    q.stride = (((int)gridDim.x + 7 - y) / 8) * waves_per_block;  // waves whose home is y
    // share handed out statically: 85 % when a wave gets dozens of items (the atomics cost more than the
    // imbalance), less when it only gets a handful and one 2-batch brick too many is a 20 % longer wave
    const int per_wave = n / max(1, (int)gridDim.x * waves_per_block);
    const int pct = per_wave >= 16 ? 85 : (per_wave >= 6 ? 70 : 50);
    const int rounds = (int)(((long long)(hi - lo) * pct / 100) / q.stride);  // full static rounds
    q.next = lo + wl;
    q.static_end = lo + rounds * q.stride;
    q.end = hi;
    q.ctr = ctr8x32 + y * 32;
}

// next item for this wave, or -1.  Wave-uniform.
__device__ __forceinline__ int wq_next(WorkQueue &q)
{
    if (q.next < q.static_end) {
        const int b = q.next;
        q.next += q.stride;
        return b;
    }
    if (q.static_end >= q.end) return -1;
    int t = 0;
    if ((threadIdx.x & 63) == 0) t = (int)atomicAdd(q.ctr, 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    const int b = q.static_end + t;
    return b < q.end ? b : -1;
}
#endif

}  // namespace gsx
