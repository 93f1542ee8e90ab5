/*
 * gsx_hip.h -- C ABI of libgsx_hip.so, the MI355X (gfx950) implementation of the
 * 3dgsconverter point-cloud filtering hot path.
 *
 * The reference (francescofugazzi/3dgsconverter v0.8) is pure Python and has no
 * FFI; its "operator API" for this path is a handful of Python callables.  Each
 * entry point below names the reference interface it replaces (file:line into
 * the reference tree) -- the ctypes stub a maintainer would add is shown in
 * INTEGRATION.md and implemented in 3dgsconverter_amd/_lib.py.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / numpy types;
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from gsx_last_error() (thread-local);
 *   - "host" entry points borrow caller-owned HOST buffers for one call and keep
 *     nothing; "_dev" entry points take DEVICE pointers that are already resident
 *     in HBM and enqueue work on the context's stream (asynchronous unless noted);
 *   - xyz is described by three base pointers and an element stride, so both the
 *     SoA columns and the reference's (N,3) row-major coords
 *     (data_processor.py:139, column_stack) can be passed without a copy
 *     (SoA: stride 1; (N,3): y = x + 1, z = x + 2, stride 3).
 */
#ifndef GSX_HIP_H
#define GSX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gsx_ctx gsx_ctx;

/* KNN algorithm selector for the SOR entry points */
enum {
    GSX_KNN_AUTO = 0,  /* grid-binned exact KNN, brute force for tiny inputs   */
    GSX_KNN_BRUTE = 1, /* LDS-tiled brute force (BASELINE.json configs[1])     */
    GSX_KNN_GRID = 2,  /* grid-binned exact KNN + exact fallback ring/brute    */
    GSX_KNN_TREE = 3   /* Morton-ordered leaves + exact tree descent: clouds whose density varies by orders of
                          magnitude (csrc/sor_tree.hip); same bits as the other two; k <= 64                      */
};

/* timing slots (HIP-event pairs recorded on the context stream when enabled) */
enum {
    GSX_T_SOR_KNN = 0,    /* dominant kernel: knn_brick / knn_brute             */
    GSX_T_SOR_BIN = 1,    /* bbox + cell histogram + scan + scatter             */
    GSX_T_SOR_FALLBACK = 2,
    GSX_T_SOR_STATS = 3,  /* numpy-exact mean/std/threshold + mask              */
    GSX_T_DENSITY = 4,
    GSX_T_KMEANS_ASSIGN = 5,
    GSX_T_KMEANS_UPDATE = 6,
    GSX_T_QUANTIZE = 7,
    /* multi-GPU slab step (gsx_sor_slab_step_dev), round 5: where a step's time goes besides the KNN slots above */
    GSX_T_SLAB_PREP = 8,   /* this rank's own passes: clear, box, histogram, partition, un-permute            */
    GSX_T_SLAB_ROWS = 9,   /* rows (own + halo) to the slab owners: the grouped send / receive over xGMI     */
    GSX_T_SLAB_MEANS = 10, /* mean distances back to the index owners (+ the < 8192 head elements)            */
    GSX_T_SLAB_COLL = 11,  /* the small collectives: box all-reduce, histogram and piece-sum all-gathers     */
    GSX_T_SLOTS = 12
};

/* diagnostics of one SOR KNN call (all counts are exact) */
typedef struct gsx_sor_info {
    int32_t algo;            /* algorithm actually used                          */
    int32_t grid_dim[3];     /* cells per axis (0 for brute force)               */
    float   cell_size;       /* cell edge h                                      */
    int64_t n_cells;
    int64_t n_bricks;
    int64_t n_fallback;      /* queries re-done by the expanding-ring kernel     */
    int64_t n_exhaustive;    /* of those, queries that needed the whole cloud    */
    int64_t n_deferred_bricks; /* bricks handed to a finer grid (adaptive mode)  */
    int64_t n_refined;       /* points of the sub-cloud that finer grid was built on */
} gsx_sor_info;

/* ---- library / device -------------------------------------------------- */
const char *gsx_version(void);
const char *gsx_last_error(void);
/* replaces the capability probe gpu_ops.py:8-23 (HAS_TAICHI) */
int gsx_device_count(void);
/* the PCI bus id of visible device `device` as text ("0000:c1:00.0"): two processes name the same physical GPU exactly
 * when the strings match, whatever HIP_VISIBLE_DEVICES each of them runs under.  launch.py picks the communicator's
 * transport with it (distinct GPUs -> RCCL, shared ones -> hostwire).  Launcher plumbing, no reference counterpart. */
int gsx_device_uid(int device, char *out, int cap);

/* ---- context ------------------------------------------------------------ */
int  gsx_ctx_create(int device, gsx_ctx **out);
void gsx_ctx_destroy(gsx_ctx *ctx);
/* use an existing hipStream_t (e.g. torch's current stream); NULL = legacy default stream */
int  gsx_ctx_set_stream(gsx_ctx *ctx, void *hip_stream);
/* give the context a non-blocking stream of its own (destroyed with it): several contexts then run concurrently on one
 * GPU -- the independent SOG palette chunks (formats/sog.py:536-552) are dealt out to a few such "lanes" */
int  gsx_ctx_own_stream(gsx_ctx *ctx);
/* GSX_KNN_* of the path the context's last KNN call ended up in (adaptive mode chooses between GRID and TREE); no synchronisation */
int  gsx_ctx_last_knn_algo(gsx_ctx *ctx);
int  gsx_ctx_synchronize(gsx_ctx *ctx);
/* synchronises, then reports (and clears) errors the asynchronous _dev calls detected ON THE DEVICE since the
 * last check: non-finite coordinates given to gsx_sor_knn_dev / gsx_sor_knn_share_dev (their mean_out was
 * filled with NaN, so statistics and masks derived from it are NaN / all-false, never garbage).  The host
 * entry points call it themselves -- the reference's cKDTree raises on such input (data_processor.py:160). */
int  gsx_ctx_check(gsx_ctx *ctx);
int  gsx_ctx_set_timing(gsx_ctx *ctx, int enable);
int  gsx_ctx_reset_timing(gsx_ctx *ctx);
/* synchronises, then returns the number of recorded launches and their summed duration */
int  gsx_ctx_get_timing(gsx_ctx *ctx, int slot, uint64_t *launches, double *total_ms);
/* knobs: "grid_points_per_cell" (0 = auto), "brute_below", "adaptive" (1: a cloud that no single grid
 * resolves -- its coarse histogram is uneven: a scene inside a box inflated by floaters, blobs of very
 * different density -- takes the Morton-tree path, csrc/sor_tree.hip; costs one host synchronisation
 * inside gsx_sor_knn_dev, so it is ON for the host entry point gsx_sor_filter and OFF by default for
 * contexts driven through the asynchronous _dev calls), "tree" (default 1; 0: adaptive mode refines the
 * grid level by level instead, DESIGN.md 5.5 -- kept for A/B and used by the replicated multi-GPU shares),
 * "tree_scale" (0, default: the tree path's density probe stretches the key grid by up to 2x so that the leaves of
 * a cloud with one dominant density are cubes of ~50 points; f > 0: that factor, no probe -- A/B; results are
 * bit-identical for every value), "filter_mfma" (1 = matrix-core phase-1 filter, default), "timing_mask" (bit s set = slot
 * GSX_T_s records events while timing is enabled; default all -- every event pair costs stream time),
 * "tree_leaf_cap" (0, default: points per leaf of the tree path by k -- 64 up to k = 28, 96 up to 34, 128 above; 64 ... 256: that
 * capacity; results are bit-identical for every value), "tree_cand_limit" (candidates per 64 points of leaf capacity above which
 * a leaf hands its queries to the per-query kernels; 0 = 4096), "kmeans_cs_small" (1, default: the K-Means assign kernel uses
 * 4-wave / 2-wave workgroups for K <= 256 / K <= 64), "km_small_wgs" (their workgroups per CU; 0 = 3), "debug_skip" (profiling only) */
int  gsx_ctx_set_param(gsx_ctx *ctx, const char *name, double value);

/* raw device memory for hosts that have no other allocator (bench without torch) */
int gsx_dev_malloc(gsx_ctx *ctx, size_t bytes, void **dptr);
int gsx_dev_free(gsx_ctx *ctx, void *dptr);
/* page-locked host memory (hipHostMalloc) for staging buffers that host routines fill and the device reads at link rate: the lazy
 * DataProcessor gathers `coords` (data_processor.py:38,139) straight into one, once per process */
int gsx_host_pinned_alloc(gsx_ctx *ctx, size_t bytes, void **hptr);
int gsx_host_pinned_free(gsx_ctx *ctx, void *hptr);
int gsx_dev_upload(gsx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int gsx_dev_download(gsx_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int gsx_dev_copy(gsx_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);  /* device to device, asynchronous */
int gsx_dev_upload_async(gsx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);  /* enqueue only */
int gsx_dev_memset(gsx_ctx *ctx, void *dst_dev, int value, size_t bytes);                    /* enqueue only */
/* bulk copies of PAGEABLE host memory (a numpy array), threaded (csrc/host_rows.hip).  Upload: the caller's pages are pinned IN
 * PLACE, 32 MiB at a time, by a few lanes that run ahead of the DMA (hipHostRegister on the lane's thread -> hipMemcpyAsync on
 * its stream -> hipHostUnregister two chunks later): link rate on the FIRST copy of a buffer, where hipMemcpy of pageable memory
 * pins, copies, pins ... (24-30 GB/s; its pinned-range cache is emptied by any munmap in the process).  A range the driver will
 * not pin, and every transfer below 128 / 256 MiB, goes through lanes that fill / drain one pinned 8 MiB staging buffer while the
 * DMA of their other one is in flight; large downloads pin the DESTINATION in place the same way (touch its pages first:
 * _lib.prefault does that on a helper thread while the device works).  Synchronous; waits for the
 * context's stream first.  What the writers move: the 2.48 GB splat table up, 240 MB of texels down. */
int gsx_dev_upload_staged(gsx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int gsx_dev_download_staged(gsx_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* ---- Host-side row operations around every filter -- */
/*
 * out[r*ncols + c] = *(float*)((char*)rows + r*row_bytes + offsets[c]) -- replaces
 * data_processor.py:38,139 (np.column_stack of the x,y,z fields of the structured array).
 * Threaded; rows are opaque bytes.
 */
int gsx_host_gather_f32(const void *rows, int64_t row_bytes, int64_t n, const int64_t *offsets,
                        int ncols, float *out);
/* the same gather with a COLUMN-major result, out[c*n + r]: one contiguous float32 column per field, as the writers upload them
 * (formats/compressed_ply.py:200-241 reads `data[name]` per column; formats/sog.py likewise) */
int gsx_host_gather_columns_f32(const void *rows, int64_t row_bytes, int64_t n, const int64_t *offsets,
                                int ncols, float *out);
/*
 * Stable compaction of the rows with mask[r] != 0 -- replaces data_processor.py:114,149
 * (self.data = vertices[mask]).  out must hold out_rows rows; *n_out = number of survivors
 * (error if it exceeds out_rows).  Threaded.
 */
int gsx_host_compact_rows(const void *rows, int64_t row_bytes, int64_t n, const uint8_t *mask,
                          void *out, int64_t out_rows, int64_t *n_out);
/* the same compaction from the survivor LIST the device chain returns (strictly ascending row indices): out[i] = rows[idx[i]] --
 * `self.data = vertices[mask]` (data_processor.py:114,149) without building the boolean mask first.  Threaded. */
int gsx_host_take_rows(const void *rows, int64_t row_bytes, int64_t n, const uint32_t *idx, int64_t n_idx, void *out);

/*
 * Device-resident filter chain (device half): stable compaction of (n,3) float32 rows by a device
 * mask -- the device-side `vertices[mask]` of data_processor.py:114,149 for the coordinates only.  orig_dev (nullable =
 * identity) / orig_out_dev carry the index into the table the chain started from, so that the host compacts its
 * 248-byte rows once, after the last filter.  *n_out = survivors (one small synchronisation).
 */
int gsx_compact_rows_dev(gsx_ctx *ctx, const float *rows_dev, const uint32_t *orig_dev, const uint8_t *mask_dev, int64_t n,
                         float *rows_out_dev, uint32_t *orig_out_dev, int64_t *n_out);

/*
 * data_processor.py:233-274,316-343 (add_rgb_from_sh / _compute_rgb_from_sh) for ONE colour channel:
 * u8((clip(0.5 + f_dc * C0, 0, 1) ** (1/2.2)) * 255) in numpy's float32 arithmetic.  The power is evaluated in float64 on
 * the device with a rounding certificate (see gsx_sog_positions): uncertain_out[i] = 1 marks the (~1e-4) elements the
 * CALLER evaluates with numpy's own expression.  Byte-identical colours.
 */
int gsx_rgb_from_sh(const float *f_dc, int64_t n, uint8_t *out, uint8_t *uncertain_out);
int gsx_rgb_from_sh_dev(gsx_ctx *ctx, const float *f_dc_dev, int64_t n, uint8_t *out_dev, uint8_t *uncertain_dev);
/* the same colours with the uncertain elements as a LIST of their indices (unordered; `cap` entries, *count keeps counting past
 * cap: the caller then falls back to the flag form) instead of one flag byte per value: for a 10M-splat table the 30 MB of flags
 * cost more to bring back and scan on the host (8 ms) than the colours themselves */
int gsx_rgb_from_sh_list(const float *f_dc, int64_t n, uint8_t *out, uint32_t *list_out, int64_t cap, int64_t *count_out);
int gsx_rgb_from_sh_list_dev(gsx_ctx *ctx, const float *f_dc_dev, int64_t n, uint8_t *out_dev, uint32_t *list_dev, int64_t cap,
                             uint32_t *count_dev);
/*
 * Host row helpers of the two table-shaping methods (threaded, no device code):
 *   gsx_host_zero_columns   -- data_processor.py:276-314 cap_sh_degree: `self.data[f_rest_i] = 0.0` for the columns above the
 *                              kept degree, ONE pass over the rows instead of up to 45 strided column fills;
 *   gsx_host_append_columns -- data_processor.py:262-274: the table widened by (red, green, blue) u1 fields, one pass instead
 *                              of one strided copy per field.  extra: n x extra_bytes row-major.
 */
int gsx_host_zero_columns(void *rows, int64_t row_bytes, int64_t n, const int64_t *offsets, int ncols);
int gsx_host_append_columns(const void *rows, int64_t row_bytes, int64_t n, const uint8_t *extra, int64_t extra_bytes,
                            void *out, int64_t out_row_bytes);
/* both steps at once -- `self.data = vertices[mask]` (data_processor.py:114,149) then the widened copy of add_rgb_from_sh
 * (:262-274), which is what the reference's converter does before every writer that needs colours (converter.py:243-252): output
 * row j = source row idx[j] + the extra_bytes of THAT source row (extra: n x extra_bytes, indexed like the source table; idx strictly
 * ascending).  One pass over the survivors instead of two copies of the table. */
int gsx_host_take_rows_append(const void *rows, int64_t row_bytes, int64_t n, const uint32_t *idx, int64_t n_idx,
                              const uint8_t *extra, int64_t extra_bytes, void *out, int64_t out_row_bytes);
/* ... and cap_sh_degree's column fill (data_processor.py:310-313 `self.data[f_rest_i] = 0.0`) in the same pass: zero_offsets = byte
 * offsets of nzero 4-byte fields to zero in every OUTPUT row; extra may be NULL with extra_bytes 0 (no columns appended).  The lazy
 * class's whole table shaping -- compaction, SH cap, colours -- is one read of the survivors and one write of the new table. */
int gsx_host_take_rows_shape(const void *rows, int64_t row_bytes, int64_t n, const uint32_t *idx, int64_t n_idx,
                             const uint8_t *extra, int64_t extra_bytes, const int64_t *zero_offsets, int nzero, void *out,
                             int64_t out_row_bytes);

/*
 * The O(N) row filters that run before density / SOR (converter.py:196-203), as device masks over the chain's rows --
 * SURVEY.md 8(f) rank 4.  gsx_mask_bbox_dev replaces crop_by_bbox's six comparisons (data_processor.py:217-224);
 * bounds6 = {min_x,min_y,min_z,max_x,max_y,max_z} on the host, compared in f64 (pass float32-rounded values for Python
 * floats, which numpy treats as weak scalars).  gsx_mask_ge_dev replaces apply_alpha_filter's
 * `self.data['opacity'] >= logit_thresh` (:210, an f64 comparison because logit_thresh is a np.float64): vals_dev is
 * the float32 column of the ORIGINAL table, orig_dev (nullable = identity) the chain's survivor list.
 */
int gsx_mask_bbox_dev(gsx_ctx *ctx, const float *rows_dev, int64_t n, const double *bounds6, uint8_t *mask_dev);
int gsx_mask_ge_dev(gsx_ctx *ctx, const float *vals_dev, const uint32_t *orig_dev, int64_t n, double threshold,
                    uint8_t *mask_dev);

/* ---- Statistical Outlier Removal --------------------------------------- */
/*
 * KNN mean distance -- replaces data_processor.py:156-173 (cKDTree build + chunked
 * query(k+1) + row mean) and the Taichi kernel gpu_ops.py:98-176 / its host prep
 * :193-256.  For the queries [q_begin, q_begin+q_count) of the n_ref reference
 * points writes mean_out[i - q_begin] = (float) mean of the k nearest neighbour
 * distances (self excluded), bit-identical to the reference's cKDTree path.
 */
int gsx_sor_knn_dev(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride,
                    int64_t n_ref, int64_t q_begin, int64_t q_count, int k, int algo,
                    float *mean_out_dev, gsx_sor_info *info /* nullable; forces a sync if given */);
/*
 * Multi-GPU building block: the same KNN over the whole cloud of n points, but
 * only for the queries of share `share` of `nshares`.  The reference already treats queries as
 * independent units -- data_processor.py:167-173 loops over 50 000-query chunks of one tree --
 * here a share is a contiguous range of the grid's bricks, i.e. a spatial slab (an index range
 * for the brute-force algorithm).  mean_out_dev has n entries in ORIGINAL order: the share's
 * queries are written, every other entry is set to +0.0f, so the shares of all ranks combine
 * with one sum all-reduce into exactly what gsx_sor_knn_dev computes for q = [0, n).
 */
int gsx_sor_knn_share_dev(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride,
                          int64_t n, int k, int algo, int share, int nshares, float *mean_out_dev,
                          gsx_sor_info *info /* nullable; forces a sync if given */);
/*
 * np.mean / np.std / threshold -- replaces data_processor.py:176-178 and
 * gpu_ops.py:259-261 with numpy's exact float32 arithmetic (8192-element buffered
 * pairwise sums, float64 division).  stats_dev[0..2] = mean, std, threshold.
 */
int gsx_sor_stats_dev(gsx_ctx *ctx, const float *mean_dists_dev, int64_t n, double threshold_factor,
                      float *stats_dev);
/* mask = mean_dists < threshold (strict) -- data_processor.py:180, gpu_ops.py:263 */
int gsx_sor_mask_dev(gsx_ctx *ctx, const float *mean_dists_dev, int64_t n, const float *threshold_dev,
                     uint8_t *mask_out_dev);

/* ---- multi-GPU SOR on one node: RCCL + the slab exchange ----
 * The reference has no distributed path: its queries are independent units over one reference set
 * (data_processor.py:167-173, 50 000-query chunks of one cKDTree) and its threshold is numpy's f32
 * mean/std over the whole mean-distance array (:176-178).  One process per GPU, splats sharded by index; the
 * host (3dgsconverter_amd/dist.py: slab_sor) drives these entry points; DESIGN.md section 7 has the protocol.
 *
 * gsx_comm_*: the collectives of the data path, on the context's stream.  Two transports behind the same calls
 * (csrc/comm.hip): "rccl" -- librccl dlopen'ed at first use, plain ncclAllReduce / ncclAllGather / grouped ncclSend +
 * ncclRecv over xGMI, one rank per GPU (the product) -- and "hostwire" -- POSIX shared-memory outboxes for ranks that SHARE
 * a GPU, which RCCL refuses (the full N-rank choreography on a one-GPU box: tests, `bench.py --gpus N` there).  The unique
 * id is created on rank 0 (under GSX_COMM_TRANSPORT=hostwire it names the shared-memory job) and handed to the other
 * ranks by the launcher (128 bytes; 3dgsconverter_amd/launch.py: a file next to the job). */
enum { GSX_COMM_F32_MAX = 0, GSX_COMM_F32_SUM = 1, GSX_COMM_I64_SUM = 2, GSX_COMM_F64_MAX = 3, GSX_COMM_I64_MIN = 4 };
int gsx_comm_unique_id(void *out128);
int gsx_comm_init(gsx_ctx *ctx, int rank, int world, const void *id128);
int gsx_comm_destroy(gsx_ctx *ctx);
int gsx_comm_abort(gsx_ctx *ctx);                 /* this rank gives up: peers waiting on it fail at once (hostwire) */
int gsx_comm_transport(gsx_ctx *ctx);             /* 0 = no communicator, 1 = rccl, 2 = hostwire */
int gsx_comm_rank(gsx_ctx *ctx, int *rank_out, int *world_out);   /* (0, 1) without a communicator */
int gsx_comm_barrier(gsx_ctx *ctx);               /* all stream work of every rank up to here has completed */
/* in place on device memory; kind: GSX_COMM_* above */
int gsx_comm_all_reduce(gsx_ctx *ctx, void *buf_dev, int64_t count, int kind);
int gsx_comm_all_gather(gsx_ctx *ctx, const void *send_dev, void *recv_dev, int64_t bytes_per_rank);
/* grouped ncclSend/ncclRecv: offsets and counts (host arrays, one entry per peer) in elements of elem_bytes
 * (1, 4, 8 or 12 = a row of three floats) */
int gsx_comm_all_to_all_v(gsx_ctx *ctx, const void *send_dev, const int64_t *send_off, const int64_t *send_cnt,
                          void *recv_dev, const int64_t *recv_off, const int64_t *recv_cnt, int elem_bytes);
/* nseg <= 2 such exchanges in ONE group (own rows and halo rows of the slab partition): entry [s * world + p] of the four
 * arrays = segment s to / from peer p */
int gsx_comm_all_to_all_segs(gsx_ctx *ctx, const void *send_dev, void *recv_dev, int nseg, const int64_t *send_off,
                             const int64_t *send_cnt, const int64_t *recv_off, const int64_t *recv_cnt, int elem_bytes);

/* ---- the slab step as ONE call (what bench.py --gpus N and dist_slab.LibSlab drive) ----
 * Why a rank may decline a step (status != 0; decided from all-gathered data, so every rank declines together and the
 * caller's replicated exchange -- dist.replicated_sor, exact for any cloud -- takes over): */
enum { GSX_SLAB_OK = 0, GSX_SLAB_EMPTY = 1, GSX_SLAB_NONFINITE = 2, GSX_SLAB_SMALL_SHARD = 3, GSX_SLAB_NO_STRUCTURE = 4 };
typedef struct gsx_slab_plan_t {
    int32_t status, world, rank, axis, halo_bins;
    float   lo, hi;                  /* binned range of the partition axis (the all-reduced box words)          */
    float   plane_lo, plane_hi;      /* this rank holds EVERY point of the cloud between these (+-inf at the ends) */
    int32_t cut[17];                 /* slab s owns the bins [cut[s], cut[s+1]) of 4096                          */
    int64_t n_local, n_total, n_own, n_halo, n_send, halo_total;
    int64_t sizes[16];               /* index-shard size of every rank                                           */
    int64_t own_cnt[16], halo_cnt[16], own_off[16], halo_off[16];       /* rows this rank sends slab s ...        */
    int64_t in_own[16], in_halo[16], r_own_off[16], r_halo_off[16];     /* ... and receives from source q         */
} gsx_slab_plan_t;
typedef struct gsx_slab_step_t {
    int32_t status;                  /* = plan.status                                                            */
    gsx_slab_plan_t plan;
    const uint8_t *mask_dev;         /* u8[n_local]: survivor mask of the local index range                      */
    const float   *mean_dists_dev;   /* f32[n_local], index order (context workspace: valid until the next step) */
    const float   *stats_dev;        /* mean, std, threshold: numpy's bits for the WHOLE cloud                    */
    const int64_t *uncertain_dev;    /* all-reduced number of queries the slabs could not certify: read it (one
                                        synchronisation) before trusting the buffers; non-zero = use the fallback */
} gsx_slab_step_t;
/* host-only: the plan from the step's gathered words (box words [0,7), histograms from word 8) */
int gsx_slab_plan(const uint32_t *words, int world, int rank, int64_t n_local, int k, double halo_cells, gsx_slab_plan_t *out);
/* rows_dev: this rank's (n_local,3) float32 index shard; replaces data_processor.py:156-180 across ranks */
int gsx_sor_slab_step_dev(gsx_ctx *ctx, const float *rows_dev, int64_t n_local, int k, double threshold_factor,
                          double halo_cells, uint8_t *mask_out_dev /* nullable */, gsx_slab_step_t *out);

/* out7_dev = max over the points of (-x,-y,-z,x,y,z) and a non-finite flag: ONE float32 max all-reduce gives the
 * global bounding box (the per-rank half of gpu_ops.py:203-206) */
int gsx_slab_bbox_dev(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                      float *out7_dev);
/* 4096-bin histogram (uint32) of the partition coordinate = the longest axis of the box in bbox7_dev (read ON THE DEVICE:
 * no host round trip after the bbox all-reduce).  All-gathered, the histograms give every rank the equal-count slab
 * cuts AND every (source, destination) row count, since slab membership is decided by bin index. */
int gsx_slab_hist_dev(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                      const float *bbox7_dev, uint32_t *hist4096_dev);
/* Slab s OWNS the bins [cut[s], cut[s+1]) of the axis range [lo, hi] and also receives, as REFERENCE-ONLY rows, the
 * bins within halo_bins of them.  Rows (3 floats) are written to send_dev starting at start_off[2*world] (HOST array:
 * first row of slot 2s = rows owned by slab s, slot 2s+1 = halo copies for slab s; cursor_dev[2*world] is device scratch,
 * zeroed by the call), send_src_dev[row] = local index of each own row.  planes_out (host, 2*world floats, nullable):
 * coordinates between which slab s holds EVERY point of the cloud. */
int gsx_slab_partition_dev(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                           int world, int axis, float lo, float hi, const int32_t *cut, int halo_bins,
                           const uint32_t *start_off, uint32_t *cursor_dev, float *send_dev, uint32_t *send_src_dev,
                           float *planes_out);
/* exact KNN mean distance of the first n_own rows against all n_own + n_halo rows (the halo rows are never
 * queries); kth_d2_dev[i] = squared distance of query i's (k+1)-th neighbour INCLUDING itself, i.e. its k-th other */
int gsx_sor_knn_slab_dev(gsx_ctx *ctx, const float *rows_dev, int64_t n_own, int64_t n_halo, int k,
                         float *mean_out_dev, double *kth_d2_dev);
/* *n_uncertain_dev = number of queries whose k-th neighbour is not provably among the rows this rank holds:
 * kth_d2 > min(coord - open_lo, open_hi - coord)^2 (open_* = the slab's planes_out entries, +-inf at the cloud's ends).
 * The counter is 8 bytes (uint32 count + a zeroed upper word, so that it can be all-reduced as an int64); set by the call */
int gsx_slab_certify_dev(gsx_ctx *ctx, const float *coord, int64_t stride, int64_t n_own, const double *kth_d2_dev,
                         float open_lo, float open_hi, uint32_t *n_uncertain_dev);
/* out_dev[send_src_dev[p]] = recv_dev[p]: mean distances come back in the order the rows were sent */
int gsx_slab_unpermute_dev(gsx_ctx *ctx, const float *recv_dev, const uint32_t *send_src_dev, int64_t n, float *out_dev);
/* numpy's float32 reduction (data_processor.py:176-177) is "pairwise inside 8192-element pieces, pieces added
 * sequentially": piece sums of a piece-aligned sub-range can be computed where the data lives ... */
int gsx_sor_piece_sums_dev(gsx_ctx *ctx, const float *a_dev, int64_t n, const float *mean_dev /* NULL: plain sum,
                           else sum of (a - *mean)^2 */, float *piece_out_dev);
/* ... and combined anywhere: mode 0: stats[0] = mean; mode 1: stats[1] = std, stats[2] = mean + factor * std */
int gsx_sor_stats_from_pieces_dev(gsx_ctx *ctx, const float *pieces_dev, int64_t npieces, int64_t n_total, int mode,
                                  double threshold_factor, float *stats_dev);

/* host buffers, one GPU: the whole of filter_sor_gpu (gpu_ops.py:193-263).
 * mean_out (n floats) and stats_out (3 floats) may be NULL. */
int gsx_sor_filter(const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                   int k, double threshold_factor, int algo, uint8_t *mask_out, float *mean_out,
                   float *stats_out, gsx_sor_info *info);

/* ---- voxel-density filter ---------------------------------------------- */
/*
 * Voxel occupancy -- replaces data_processor.py:38-52 (floor(xyz / voxel) keys,
 * np.unique(axis=0) counts, dense = count >= min_points).  Returns the number of
 * occupied voxels and the dense voxels (lexicographically sorted like np.unique)
 * in dense_keys_out (3 x int64 each) / dense_counts_out, at most dense_cap.
 */
int gsx_density_voxels(const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                       double voxel_size, int64_t min_points, int64_t dense_cap,
                       int64_t *n_unique_out, int64_t *n_dense_out, int64_t *dense_keys_out,
                       int64_t *dense_counts_out);
/*
 * Per-point membership -- replaces data_processor.py:111-114: mask[i] = voxel(i) in kept set.
 * kept_keys: n_kept x 3 int64.
 */
int gsx_density_mask(const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                     double voxel_size, const int64_t *kept_keys, int64_t n_kept, uint8_t *mask_out);

/* device-resident variants (dense_* outputs are HOST arrays; the kept set is a HOST array) */
int gsx_density_voxels_dev(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride,
                           int64_t n, double voxel_size, int64_t min_points, int64_t dense_cap,
                           int64_t *n_unique_out, int64_t *n_dense_out, int64_t *dense_keys_out,
                           int64_t *dense_counts_out);
/*
 * Multi-GPU density filter (the reference is single-process): rank r counts the voxels of ITS index
 * shard and exports every occupied voxel as an absolute int64 key triple + int64 count into DEVICE buffers
 * (gsx_density_hist_dev; cap entries each, arbitrary order; *n_unique_out = entries written), the lists are all-gathered
 * (gsx_comm_*), and gsx_density_merge_dev adds the counts of equal keys of the m gathered entries in a hash table and
 * returns what gsx_density_voxels would have returned for the whole cloud: data_processor.py:43-52 with
 * min_points = int(N_global * thr / 100) (:48).  The host BFS (:59-108) and gsx_density_mask_dev on the local rows follow.
 */
int gsx_density_hist_dev(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                         double voxel_size, int64_t cap, int64_t *n_unique_out, int64_t *keys3_dev, int64_t *counts_dev);
int gsx_density_merge_dev(gsx_ctx *ctx, const int64_t *keys3_dev, const int64_t *counts_dev, int64_t m, int64_t min_points,
                          int64_t dense_cap, int64_t *n_unique_out, int64_t *n_dense_out, int64_t *dense_keys_out,
                          int64_t *dense_counts_out);
/* The whole filter on device-resident rows in ONE call (data_processor.py:38-114): occupancy, dense voxels, their
 * 6-connected clusters (one workgroup: at most n / min_points <= ~1000 voxels are dense), the keep rule (:95-106) and the
 * membership mask -- one synchronisation, at the end.  box6 (host, nullable): minima then maxima of a superset of the rows
 * (saves the box pass and its synchronisation).  info->status: GSX_DENSITY_OK (mask_out_dev holds the mask),
 * GSX_DENSITY_EMPTY (no dense voxel: the filter removes everything, :54-57), GSX_DENSITY_HOST (more than 1024 dense voxels,
 * or two clusters tie for the largest without keep_multicluster -- which one survives is the reference's python-set
 * iteration order: decide with gsx_density_voxels_dev + processing/clusters.py + gsx_density_mask_dev). */
enum { GSX_DENSITY_OK = 0, GSX_DENSITY_EMPTY = 1, GSX_DENSITY_HOST = 2 };
typedef struct gsx_density_info {
    int32_t status;
    int64_t n_unique, n_dense, n_kept_voxels, kept_clusters, largest;
} gsx_density_info;
int gsx_density_filter_dev(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride, int64_t n,
                           double voxel_size, int64_t min_points, int keep_multicluster, const float *box6,
                           uint8_t *mask_out_dev, gsx_density_info *info);
int gsx_density_mask_dev(gsx_ctx *ctx, const float *x, const float *y, const float *z, int64_t stride,
                         int64_t n, double voxel_size, const int64_t *kept_keys, int64_t n_kept,
                         uint8_t *mask_out_dev);

/* ---- K-Means codebook (SOG writer) ------------------------------------- */
/*
 * Lloyd iterations with an injected initialisation -- replaces _kmeans_taichi
 * (gpu_ops.py:178-191) and its kernels k_means_assign (:57-73) / k_means_update
 * (:75-96): exactly max_iter x (assign, update), empty cluster -> 0, returned
 * labels are one step older than the returned centroids.  data: n x d row-major.
 */
int gsx_kmeans_lloyd(const float *data, int64_t n, int d, int k, int max_iter,
                     const float *init_centroids, float *centroids_out, int32_t *labels_out);
/* nearest entry of a sorted codebook -- replaces quantize_to_codebook, formats/sog.py:408-419 */
int gsx_quantize_sorted_codebook(const float *vals, int64_t n, const float *codebook, int kcb,
                                 uint8_t *idx_out);

/*
 * Scalar (D = 1) K-Means codebook -- replaces the two places where the reference fits a 1-D codebook with scikit-learn's
 * k-means++-seeded MiniBatchKMeans: formats/sog.py:561 (MiniBatchKMeans(n_clusters=256, n_init='auto') over the flattened
 * SH palette, hard-wired) and processing/gpu_ops.py:48-52 (_kmeans_sklearn, what gpu_ops.kmeans runs for
 * formats/sog.py:402,443 when Taichi is absent or use_gpu=False).  Both are unseeded in the reference: the contract is
 * quality (inertia <= sklearn's on the same data), not bits.  Deterministic: sort + float64 prefix sums + `iters` Lloyd
 * steps on run boundaries from two companded starts (csrc/kmeans1d.hip).  centroids_out: k floats, ASCENDING;
 * labels_out (may be NULL): nearest centroid per value, ties to the lower index (gpu_ops.py:66-70);
 * inertia3_out (may be NULL): {chosen, uniform-bin start, equal-count-bin start}.  1 <= k <= 1024, values finite.
 */
int gsx_kmeans1d(const float *vals, int64_t n, int k, int iters, float *centroids_out, int32_t *labels_out,
                 double *inertia3_out);
/* start_mask: bit 0 = uniform-bin start, bit 1 = equal-count-bin start (0 = both); inertia3_host is a HOST array */
int gsx_kmeans1d_dev(gsx_ctx *ctx, const float *vals_dev, int64_t n, int k, int iters, int start_mask,
                     float *centroids_dev, int32_t *labels_dev, double *inertia3_host);

/*
 * Greedy k-means++ seeding (D^2 sampling, n_local_trials candidates per centroid, the one that lowers the potential most
 * wins) -- the initialisation of the reference's CPU path, scikit-learn's MiniBatchKMeans (processing/gpu_ops.py:48-52;
 * sklearn uses n_local_trials = 2 + int(ln k)).  The host draws the 1 + (k-1) * n_local_trials uniform numbers in [0,1)
 * (numpy's global stream, like sklearn); the device does the O(n d) passes (csrc/kmeans_pp.hip).  Follow with gsx_kmeans_lloyd.
 */
int gsx_kmeans_pp(const float *data, int64_t n, int d, int k, const double *uniforms, int n_local_trials, float *centroids_out);
int gsx_kmeans_pp_dev(gsx_ctx *ctx, const float *data_dev, int64_t n, int d, int k, const double *uniforms_host,
                      int n_local_trials, float *centroids_dev);

/* device-resident variants: centroids_dev holds the init on entry and the result on exit */
int gsx_kmeans_lloyd_dev(gsx_ctx *ctx, const float *data_dev, int64_t n, int d, int k, int max_iter,
                         float *centroids_dev, int32_t *labels_dev);
/*
 * nprob INDEPENDENT Lloyd problems of the same d and k in one call -- the SH palette of the SOG writer is 64 such problems
 * (formats/sog.py:536-552: `for i in range(num_chunks): ... kmeans(chunk, k_per_chunk, max_iter=10)`).  Rows concatenated:
 * problem p = rows [row_off[p], row_off[p+1]) of data_dev / labels_dev (row_off: HOST array of nprob + 1 offsets, row_off[0]
 * = 0), its k x d centroids at centroids_dev + p*k*d (init on entry, result on exit), labels local to the problem (0..k-1,
 * as gpu_ops.kmeans returns them; the writer adds the chunk's offset, sog.py:546-552).  Per problem the same kernels,
 * launch order and arithmetic as gsx_kmeans_lloyd_dev; for the matrix-core shapes (d in {9, 24, 45}, k >= 64) every
 * iteration of ALL problems is one set of launches (the problem is a grid dimension) instead of nprob x 6.
 */
int gsx_kmeans_lloyd_batch_dev(gsx_ctx *ctx, const float *data_dev, const int64_t *row_off, int nprob, int d, int k,
                               int max_iter, float *centroids_dev, int32_t *labels_dev);
int gsx_quantize_sorted_codebook_dev(gsx_ctx *ctx, const float *vals_dev, int64_t n,
                                     const float *codebook_dev, int kcb, uint8_t *idx_out_dev);

/* ---- SOG writer numeric core next to the codebooks ---- */
/* perm = np.lexsort((k0, k1, k2)): k2 is the primary key -- formats/sog.py:264 lexsort((z, y, x)).  Three stable radix
 * passes; -0.0 == +0.0 and NaNs last, like numpy.  perm_out: n uint32 */
int gsx_lexsort3(const float *k0, const float *k1, const float *k2, int64_t n, uint32_t *perm_out);
int gsx_lexsort3_dev(gsx_ctx *ctx, const float *k0, const float *k1, const float *k2, int64_t stride, int64_t n,
                     uint32_t *perm_out_dev);
/*
 * formats/sog.py:279-309 for ONE axis: v -> sign(v) log(|v| + 1) -> (l - log_min) / (log_max - log_min) * 65535 -> clip ->
 * uint16, and formats/sog.py:457-459: 255 / (1 + exp(-opacity)) -> clip -> uint8.  The reference evaluates log / exp with
 * numpy's float32 SIMD routines (not correctly rounded), so the device computes them in float64, brackets numpy's possible
 * float32 result (+-5 ulp for log, +-4 for exp) and emits the texel only when both ends of the bracket quantise to the same integer through
 * numpy's exact float32 sequence; uncertain_out[i] = 1 marks the (~1 %) elements the CALLER must evaluate with numpy's
 * own expression.  log_min / log_max: numpy's np.min / np.max of the transformed axis (the host obtains them from the few
 * values next to the extremes of v: the transform is monotone).  Result: byte-identical textures.
 */
int gsx_sog_positions(const float *v, int64_t n, float log_min, float log_max, uint16_t *out, uint8_t *uncertain_out);
int gsx_sog_positions_dev(gsx_ctx *ctx, const float *v_dev, int64_t n, float log_min, float log_max, uint16_t *out_dev,
                          uint8_t *uncertain_dev);
int gsx_sog_alpha(const float *opacity, int64_t n, uint8_t *out, uint8_t *uncertain_out);
int gsx_sog_alpha_dev(gsx_ctx *ctx, const float *opacity_dev, int64_t n, uint8_t *out_dev, uint8_t *uncertain_dev);

/* formats/sog.py:315-386: normalise the (n,4) float32 quaternion rows, flip to the positive hemisphere of the largest
 * component, scale by sqrt(2), quantise the three others to bytes; out4[i] = (c0, c1, c2, 252 + argmax), byte-exact */
int gsx_sog_quats(const float *rot_rows, int64_t n, uint8_t *out4);
int gsx_sog_quats_dev(gsx_ctx *ctx, const float *rot_rows_dev, int64_t n, uint8_t *out4_dev);

/* ---- the SOG writer on a DEVICE-RESIDENT splat table (csrc/sog_table.hip) ----
 * formats/sog.py:249-600 between the structured table and the RGBA texel arrays of its WebP images, without the host copies
 * the reference makes (`data_s = data[indices]` :265, the quaternion / SH `column_stack`s :315,:499-503, `np.concatenate` of
 * the 3N scalars :391,:434): the raw rows are uploaded once (gsx_dev_upload), every stage reads and writes HBM, only texels
 * (4 bytes per splat and image) and short lists of texels for the host's numpy come back.  Call order:
 * scan -> extremes -> order -> gather -> *_texels (+ gsx_kmeans1d_dev / gsx_kmeans_lloyd_batch_dev for the codebooks). */
#define GSX_SOG_FIELDS 59
typedef struct gsx_sog_layout {
    int64_t row_bytes;                 /* itemsize of the structured dtype: at most 512; ANY size up to 500 (the table widened by the
                                          three u1 colour fields of data_processor.py:262-274 has 251-byte rows: the kernels then
                                          assemble every field from two words of their LDS tile, and read up to 3 bytes past the
                                          last row -- rows_dev must be readable up to the next 4-byte boundary) */
    int32_t n_rest;                    /* f_rest columns present (0, 9, 24 or 45: sog.py:468-474) */
    int32_t offset[GSX_SOG_FIELDS];    /* byte offset of the float32 fields x y z | rot_0..3 | scale_0..2 | f_dc_0..2 | opacity |
                                          f_rest_0..44 inside a row (any offset; entries beyond 14 + n_rest are ignored) */
} gsx_sog_layout;
typedef struct gsx_sog_scan {
    float    vmin[3], vmax[3];         /* np.min / np.max of x, y, z (a -0.0 is reported as +0.0) */
    uint32_t nonfinite;                /* bit a: axis a holds a NaN or an infinity */
    uint32_t reserved;
    uint64_t rest_nonzero;             /* bit i: np.any(data['f_rest_i'] != 0) -- the band detection of sog.py:476-486 */
} gsx_sog_scan;
/* one pass over the n rows in table order: keys3_dev[a * n + i] = the uint32 whose unsigned order is numpy's sort order of
 * axis a's float32 value (-0.0 == +0.0, NaN last); *out (HOST) after one small synchronisation */
int gsx_sog_scan_dev(gsx_ctx *ctx, const void *rows_dev, const gsx_sog_layout *layout, int64_t n, uint32_t *keys3_dev,
                     gsx_sog_scan *out);
/* the values v <= lo3[a] (list 2a) and v >= hi3[a] (list 2a + 1) of every axis, from the key columns: the candidates for
 * np.min / np.max of the log-transformed axis (sog.py:287-288; numpy's float32 log is monotone only up to a few ulp, so the
 * host evaluates its own expression on the values within a small window of the extremes).  vals_out: HOST float[6][cap],
 * counts6_out: HOST, the true list lengths (entries beyond cap are dropped).  Synchronises. */
int gsx_sog_extremes_dev(gsx_ctx *ctx, const uint32_t *keys3_dev, int64_t n, const float *lo3, const float *hi3, int cap,
                         float *vals_out, int64_t *counts6_out);
/* perm = np.lexsort((z, y, x)) (sog.py:264) from the key columns: three stable radix passes */
int gsx_sog_order_dev(gsx_ctx *ctx, const uint32_t *keys3_dev, int64_t n, uint32_t *perm_out_dev);
/* `data_s = data[indices]` (sog.py:265) for the columns the writer reads, one pass: pos_dev f32[3][n] (x, y, z columns),
 * rot_dev f32[n][4] (the column_stack of :315), scale_dev / dc_dev f32[3][n] (= the np.concatenate of :391 / :434),
 * opacity_dev f32[n], sh_dev f32[n][d_sh] (the column_stack of :499-503; d_sh = 0, 9, 24 or 45; nullable for 0) */
int gsx_sog_gather_dev(gsx_ctx *ctx, const void *rows_dev, const gsx_sog_layout *layout, const uint32_t *perm_dev, int64_t n,
                       int d_sh, float *pos_dev, float *rot_dev, float *scale_dev, float *dc_dev, float *opacity_dev,
                       float *sh_dev);
/* Texel writers: out = `texels` x 4 bytes, the flat RGBA array the reference hands to write_webp (texels = width * height
 * >= n; the padding texels get the reference's fill value).  Lists: `cap` entries of two uint32 (texel index * 4 + channel,
 * the bits of the float32 input value) for the values whose float32 log / exp bracket straddles a rounding boundary -- the
 * caller evaluates numpy's own expression for those (see gsx_sog_positions) and patches the texel; *count_dev (uint32,
 * zeroed by the call) keeps counting past cap.
 *   means:  sog.py:279-312 -> means_l (low bytes of the u16 x, y, z, 255) and means_u (high bytes, 255); padding 255.
 *           log_min3 / log_max3 = numpy's np.min / np.max of the transformed axes (gsx_sog_extremes_dev), arg_min3 / arg_max3
 *           = a coordinate value per axis whose transform numpy evaluated to exactly that minimum / maximum (texels 0 / 65535
 *           for every copy of it without consulting the bracket)
 *   quats:  sog.py:315-386 -> (c0, c1, c2, 252 + argmax); padding 255
 *   codes:  sog.py:408-431 / :446-459 -> (codebook index of the three columns, 255 or -- opacity_dev given -- the sigmoid of
 *           the opacity as a byte); padding 0.  cols3_dev = f32[3][n], codebook_dev = kcb <= 256 ascending float32
 *   labels: sog.py:546-552,:600-606 -> palette label = labels_dev[i] + (i / chunk_rows) * k as (low byte, high byte, 0, 255) */
int gsx_sog_means_texels_dev(gsx_ctx *ctx, const float *pos_dev, int64_t n, int64_t texels, const float *log_min3,
                             const float *log_max3, const float *arg_min3 /* nullable */, const float *arg_max3 /* nullable */,
                             uint8_t *means_l_dev, uint8_t *means_u_dev, uint32_t *list_dev,
                             int64_t cap, uint32_t *count_dev);
int gsx_sog_quats_texels_dev(gsx_ctx *ctx, const float *rot_rows_dev, int64_t n, int64_t texels, uint8_t *out_dev);
int gsx_sog_codes_texels_dev(gsx_ctx *ctx, const float *cols3_dev, int64_t n, int64_t texels, const float *codebook_dev, int kcb,
                             const float *opacity_dev /* nullable */, uint8_t *out_dev, uint32_t *list_dev, int64_t cap,
                             uint32_t *count_dev);
int gsx_sog_labels_texels_dev(gsx_ctx *ctx, const int32_t *labels_dev, int64_t n, int64_t texels, int64_t chunk_rows, int k,
                              uint8_t *out_dev);
/* dst row i = src row idx_dev[i] (rows of row_floats float32): `s_data[idx]` of the codebooks' 50 000-sample (sog.py:397-400,
 * row_floats = 1) and the palette's initial centroids `data[np.random.choice(N, k)]` (gpu_ops.py:182) */
int gsx_gather_rows_dev(gsx_ctx *ctx, const float *src_dev, int row_floats, const int64_t *idx_dev, int64_t m, float *dst_dev);

/* ---- compressed-PLY writer numeric core ---- */
/*
 * formats/compressed_ply.py:245-291 _sort_morton_order: 10-bit-per-axis Morton codes inside the bounding box, groups of
 * equal code with more than 256 members re-sorted inside their own box, recursively.  order_out_dev: n uint32, the
 * `indices` array the reference sorts in place.  The sequence of codes is the reference's; splats with EQUAL code keep
 * ascending input index here, while np.argsort (not stable) leaves them in a build-dependent order.  *levels_out
 * (nullable) = recursion depth reached.  Coordinates must be finite.
 */
int gsx_morton_order_dev(gsx_ctx *ctx, const float *x_dev, const float *y_dev, const float *z_dev, int64_t stride, int64_t n,
                         uint32_t *order_out_dev, int *levels_out);
/*
 * compressed_ply.py:205-234 (chunk loop) with :293-341 (_normalize_and_pack_11_10_11, _normalize_and_pack_8888,
 * _pack_quaternions): one workgroup per 256-splat chunk.  cols14_dev: HOST array of 14 device columns of the ORIGINAL
 * table (float32, contiguous): x y z, scale_0..2, f_dc_0..2, alpha = sigmoid(opacity) as numpy computed it (:200-203),
 * rot_0..3.  order_dev (nullable = identity): the Morton order.  chunk_out_dev: ceil(n/256) x 18 float32 in the field
 * order of the reference's chunk record (:167-174); vertex_out_dev: n x 4 uint32 (packed_position, packed_rotation,
 * packed_scale, packed_color), 16-byte aligned.  Bit-exact given the same order.
 */
int gsx_cply_pack_dev(gsx_ctx *ctx, const float *const *cols14_dev, const uint32_t *order_dev, int64_t n,
                      float *chunk_out_dev, uint32_t *vertex_out_dev);
/* compressed_ply.py:236-241: out[i, c] = uint8(clip((col_c[order[i]] / 8.0 + 0.5) * 256, 0, 255)); cols_dev = m columns
 * of col_stride floats each (m <= 45), out_dev = (n, m) bytes = the reference's `sh` element */
int gsx_cply_sh_dev(gsx_ctx *ctx, const float *cols_dev, int m, int64_t col_stride, const uint32_t *order_dev, int64_t n,
                    uint8_t *out_dev);
/* Round 6: the same two packers on RAW ROWS resident in HBM (the table uploaded once, no column gather on the host): column a of
 * splat s is cols14_dev[a][s * strides14[a]] (strides14: HOST array, 1 = a contiguous column such as the numpy-computed alpha,
 * row_bytes / 4 = a field inside the rows; NULL = all 1); coefficient c of splat s is cols_dev[c * col_stride + s * elem_stride]
 * (the consecutive f_rest fields of a row: col_stride 1, elem_stride = row_bytes / 4).  gsx_morton_order_dev takes its stride
 * argument the same way.  out_dev of the SH packer must be 4-byte aligned. */
int gsx_cply_pack_strided_dev(gsx_ctx *ctx, const float *const *cols14_dev, const int64_t *strides14, const uint32_t *order_dev,
                              int64_t n, float *chunk_out_dev, uint32_t *vertex_out_dev);
int gsx_cply_sh_strided_dev(gsx_ctx *ctx, const float *cols_dev, int m, int64_t col_stride, int64_t elem_stride,
                            const uint32_t *order_dev, int64_t n, uint8_t *out_dev);
/* compressed_ply.py:139-150 `np.any(data[f"f_rest_{i}"] != 0)` from 44 downwards (and sog.py:476-486, the same loop), for m <= 64
 * CONSECUTIVE float32 fields of rows resident in HBM: bit c of *mask_out (HOST word) is set iff field c -- first_field_dev[s * row_stride
 * + c], s < n -- holds a value != 0 (NaN counts, -0.0 does not, as in numpy).  One pass over the rows (10M x 45 fields: < 1 ms) where
 * the reference's loop costs ~0.1 s per strided column.  Synchronises the context's stream. */
int gsx_fields_nonzero_dev(gsx_ctx *ctx, const float *first_field_dev, int64_t row_stride, int64_t n, int m, uint64_t *mask_out);
/* rows of any size -> rows of out_pitch bytes (a multiple of 4 >= row_bytes; the padding bytes are zero), both resident in HBM: the
 * table the reference's converter widens by three u1 colour fields (data_processor.py:262-274, 251-byte rows) becomes one whose
 * float32 fields at 4-byte offsets can be read by the strided packers above.  rows_dev must be readable up to 8 bytes past the
 * last row (a device allocation's slack). */
int gsx_rows_repack_dev(gsx_ctx *ctx, const void *rows_dev, int64_t row_bytes, int64_t n, void *out_dev, int64_t out_pitch);
/* ... with column 9 = the OPACITY itself (compressed_ply.py:200-203 `1.0 / (1.0 + np.exp(-x))`, then :312 floor(a * 255 + 0.5)): the
 * alpha byte comes from a float64 exp + the rounding certificate of gsx_sog_alpha; list_dev receives `cap` entries of two uint32
 * (position in the NEW order, bits of the opacity) for the ~1e-4 splats whose byte the caller patches with numpy's own expression
 * (`vertex[pos, 3] = vertex[pos, 3] & ~0xff | byte`); *count_dev (zeroed by the call) keeps counting past cap */
int gsx_cply_pack_opacity_dev(gsx_ctx *ctx, const float *const *cols14_dev, const int64_t *strides14, const uint32_t *order_dev,
                              int64_t n, float *chunk_out_dev, uint32_t *vertex_out_dev, uint32_t *list_dev, int64_t cap,
                              uint32_t *count_dev);

#ifdef __cplusplus
}
#endif
#endif /* GSX_HIP_H */
