// sor_grid_params.h -- device-resident description of the KNN cell grid + work counters.
#pragma once

namespace gsx {

struct GridParams {
    float ox, oy, oz, inv_h;
    float h;
    int nx, ny, nz;
    int ncells;
    int nbx, nby, nbz;
    int nbricks;
    int bdx, bdy, bdz;  // brick size in cells: (2,2,2), (2,2,1), (2,1,1) or (1,1,1)
    float tau1;      // f32 filter bound for r1sq
    double r1sq;     // (h' * (1 - 1e-3))^2, h' = 1/inv_h
    double hprime;   // 1/inv_h
    // zeroed by grid_params_kernel every call
    unsigned fail_count;
    unsigned exhaustive_count;
    unsigned extra_count;   // (brick, batch >= 1) work items appended by the first knn_brick pass
    unsigned bad_input;     // 1: non-finite coordinates -- the KNN kernels do nothing, the host reports an error
    // work queues: one counter per XCD, 128 B apart.  A single device-wide counter saturates at
    // ~88 dequeues/us on MI355X (MI355X_MICROARCH.md "dequeue"), which throttled knn_brick at
    // 10M splats (185k bricks); 8 counters on 8 cache lines/channels scale that 8x and keep a
    // contiguous brick range -- hence its L2 working set -- on one XCD.
    unsigned brick_ctr[8 * 32];
    unsigned extra_ctr[8 * 32];
    unsigned ring_ctr[8 * 32];
};

// Work distribution shared by knn_brick / knn_ring (device).  XCD y owns the contiguous item
// range [n*y/8, n*(y+1)/8).  A wave whose home is y (blockIdx % 8 -- a placement HINT only, any
// mapping is correct) takes item (range start + its index among y's waves) first WITHOUT an
// atomic, then pulls further items from y's counter, then helps the other XCDs' queues
// (a relaxed peek avoids the atomic when a queue is already drained).
struct WorkQueue {
    unsigned *ctr;  // 8 counters, 32 words apart
    int n;          // total items
    int wpb;        // waves per workgroup
    int cur;        // queue currently pulled from
    int wl;         // this wave's index among its home XCD's waves
    int visited;
    bool first;
};

#ifdef __HIPCC__
__device__ __forceinline__ int wq_waves_of(const WorkQueue &q, int y)
{
    return (((int)gridDim.x + 7 - y) / 8) * q.wpb;  // workgroups b with b % 8 == y, times waves each
}

__device__ __forceinline__ void wq_init(WorkQueue &q, unsigned *ctr, int n, int waves_per_block)
{
    q.ctr = ctr;
    q.n = n;
    q.wpb = waves_per_block;
    q.cur = (int)(blockIdx.x & 7);
    q.wl = (int)(blockIdx.x >> 3) * waves_per_block + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    q.visited = 0;
    q.first = true;
}

// next item for this wave, or -1 when every queue is drained.  Wave-uniform.
__device__ __forceinline__ int wq_next(WorkQueue &q)
{
    for (;;) {
        const int y = q.cur;
        const int lo = (int)(((long long)q.n * y) / 8), hi = (int)(((long long)q.n * (y + 1)) / 8);
        if (q.first) {
            q.first = false;
            if (lo + q.wl < hi) return lo + q.wl;
        }
        const int base = lo + wq_waves_of(q, y);
        unsigned seen = __hip_atomic_load(&q.ctr[y * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);
        if ((long long)base + (long long)seen < (long long)hi) {
            int t = 0;
            if ((threadIdx.x & 63) == 0) t = (int)atomicAdd(&q.ctr[y * 32], 1u);
            t = __builtin_amdgcn_readfirstlane(t);
            if (base + t < hi) return base + t;
        }
        if (++q.visited >= 8) return -1;
        q.cur = (q.cur + 1) & 7;
    }
}
#endif

}  // namespace gsx
