"""Multi-GPU SOR by spatial slabs: one process per GPU, RCCL over xGMI called from the C library.

The reference is single-process; its SOR treats queries as independent units over one reference
set (data_processor.py:160-173) and thresholds on numpy's f32 mean/std of the whole mean-distance
array (:176-180).  Rank r holds an INDEX shard of the cloud (as a loader hands it out) and gets the
survivor mask of that shard back -- bit-identical to the single-GPU / reference result.

One step (``slab_sor``); the device pieces are C-ABI entry points (include/gsx_hip.h,
csrc/dist_slab.hip), this module only moves sizes and offsets around:

  1. global bounding box: ``gsx_slab_bbox_dev`` + ONE float32 max all-reduce of 7 words (stays on the device);
  2. 4096-bin histogram of the longest axis, all-gathered (16 KB per rank): equal-count slab cuts AND -- because
     slab membership is decided by bin index -- every (source, destination) row count.  This is the step's only
     host synchronisation;
  3. ``gsx_slab_partition_dev``: every point is sent to the rank owning its slab and, as a
     reference-only copy, to every rank whose slab lies within ``halo_cells`` KNN cell edges (of the
     GLOBAL density, rounded up to whole bins) of it.  Grouped ncclSend/ncclRecv, 12 B per point -- no
     rank ever holds or bins the whole cloud (an all-gather of xyz costs 12 (G-1) N_local B per rank and
     G-fold redundant binning: measured dead end of round 1);
  4. ``gsx_sor_knn_slab_dev``: the single-GPU exact-KNN pipeline on (own + halo) rows, halo lanes dead;
  5. certificate ``gsx_slab_certify_dev``: a result is globally exact iff the query's k-th neighbour is
     nearer than the edge of what this rank received.  Sum all-reduce of the number of uncertified
     queries, read back lazily (``SlabResult.check``); if it is not zero (far floaters, slabs thinner
     than the neighbour distance) ``SlabUncertain`` is raised and the caller re-runs the step with the
     replicated exchange of dist.py, which is exact for any cloud -- results are identical either way;
  6. mean distances travel back to the index owners (4 B per point) and are un-permuted;
  7. statistics: numpy's float32 reduction is "pairwise inside 8192-element pieces, pieces added
     sequentially", so each rank sums the pieces it holds (after handing the < 8192 leading
     elements that belong to its left neighbour's last piece over) and the piece sums (KBs) are
     all-gathered: bit-identical mean / std / threshold without gathering the array;
  8. mask of the local index range.

The product runs the whole step as ONE C call (``gsx_sor_slab_step_dev``, csrc/dist_slab.hip: ``LibSlab`` below /
``slab_sor`` on a ``HipSlabBackend``), with the collectives of ``gsx_comm_*`` (csrc/comm.hip: RCCL, or the shared-memory
"hostwire" when ranks share a GPU).  ``slab_sor_steps`` spells the same choreography out call by call over an injectable
``Comm`` / backend: tests/ run it on CPU with gloo and a numpy backend built on the oracle (never the product), and on the
GPU against the fused call.  No torch anywhere in this module.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

KIND_F32_MAX, KIND_F32_SUM, KIND_I64_SUM, KIND_F64_MAX, KIND_I64_MIN = 0, 1, 2, 3, 4   # GSX_COMM_* (include/gsx_hip.h)
NP_PIECE = 8192   # numpy's reduction buffer (elements): see csrc/sor_stats.hip
BINS = 4096

PARALLELISM = ("index-sharded input; spatial slabs: all-to-all of xyz rows to the slab owners (+ reference-only halo), "
               "exact KNN per slab, all-to-all of the mean distances back, all-gather of numpy-exact 8192-piece "
               "sums (RCCL from libgsx_hip.so, no collective inside kernels)")


class SlabUncertain(RuntimeError):
    """some query's k-th neighbour may lie outside what its slab rank received (not an error of the data:
    the caller re-runs the step with the replicated exchange)"""


class SlabUnsupported(SlabUncertain):
    """the shard sizes cannot be served by the slab exchange (a shard below numpy's 8192-element reduction piece, an empty
    cloud).  Decided from the all-gathered histograms, i.e. raised by EVERY rank in the same step, and a SlabUncertain: the
    caller's fallback to the replicated exchange of dist.py -- exact for any shard sizes -- covers it."""


# ------------------------------------------------------------------------------------------ communicators
class RcclComm:
    """The C library's communicator (gsx_comm_*) on the backend context's stream: RCCL, or -- when the unique id was made
    under GSX_COMM_TRANSPORT=hostwire -- shared-memory outboxes for ranks that share a GPU.  Buffers: anything with ``.ptr``."""

    def __init__(self, ctx, rank: int, world: int, unique_id: bytes):
        from . import _lib
        self._lib, self.ctx, self.rank, self.world = _lib, ctx, int(rank), int(world)
        buf = C.create_string_buffer(unique_id, 128)
        _lib.check(ctx.lib.gsx_comm_init(ctx.handle, self.rank, self.world, buf), "gsx_comm_init")
        import atexit
        atexit.register(self.close)   # before the interpreter tears the HIP / RCCL libraries down

    def close(self):
        """ncclCommDestroy / unlink of the shared-memory outbox (idempotent)"""
        if self.ctx is not None and getattr(self.ctx, "handle", None):
            self.ctx.lib.gsx_comm_destroy(self.ctx.handle)
        self.ctx = None

    def abort(self):
        """tell the peers this rank gives up (hostwire: their barriers fail at once instead of timing out)"""
        if self.ctx is not None and getattr(self.ctx, "handle", None):
            self.ctx.lib.gsx_comm_abort(self.ctx.handle)

    @property
    def transport(self) -> str:
        return {0: "none", 1: "rccl", 2: "hostwire"}[int(self.ctx.lib.gsx_comm_transport(self.ctx.handle))]

    @staticmethod
    def unique_id(transport: str | None = None) -> bytes:
        """128 bytes made on rank 0 and handed to every rank (launch.py).  transport: None = GSX_COMM_TRANSPORT or rccl"""
        import os
        from . import _lib
        buf = C.create_string_buffer(128)
        old = os.environ.get("GSX_COMM_TRANSPORT")
        if transport is not None:
            os.environ["GSX_COMM_TRANSPORT"] = transport
        try:
            _lib.check(_lib.load().gsx_comm_unique_id(buf), "gsx_comm_unique_id")
        finally:
            if transport is not None:
                if old is None:
                    os.environ.pop("GSX_COMM_TRANSPORT", None)
                else:
                    os.environ["GSX_COMM_TRANSPORT"] = old
        return buf.raw

    def barrier(self):
        self._lib.check(self.ctx.lib.gsx_comm_barrier(self.ctx.handle), "gsx_comm_barrier")

    def all_reduce(self, buf, count: int, kind: int):
        self._lib.check(self.ctx.lib.gsx_comm_all_reduce(self.ctx.handle, buf.ptr, int(count), int(kind)), "gsx_comm_all_reduce")

    def all_gather(self, send, recv, nbytes: int):
        self._lib.check(self.ctx.lib.gsx_comm_all_gather(self.ctx.handle, send.ptr, recv.ptr, int(nbytes)), "gsx_comm_all_gather")

    def all_to_all_v(self, send, send_off, send_cnt, recv, recv_off, recv_cnt, elem_bytes: int):
        arr = lambda v: (C.c_int64 * self.world)(*[int(x) for x in v])
        self._lib.check(self.ctx.lib.gsx_comm_all_to_all_v(self.ctx.handle, send.ptr, arr(send_off), arr(send_cnt), recv.ptr,
                                                          arr(recv_off), arr(recv_cnt), int(elem_bytes)), "gsx_comm_all_to_all_v")

    def all_to_all_segs(self, send, recv, segs, elem_bytes: int):
        """segs: up to two (send_off, send_cnt, recv_off, recv_cnt) exchanges issued as ONE group"""
        flat = lambda i: (C.c_int64 * (self.world * len(segs)))(*[int(x) for sg in segs for x in sg[i]])
        self._lib.check(self.ctx.lib.gsx_comm_all_to_all_segs(self.ctx.handle, send.ptr, recv.ptr, len(segs), flat(0), flat(1), flat(2),
                                                             flat(3), int(elem_bytes)), "gsx_comm_all_to_all_segs")

    # host scalars through the device communicator (launchers: agreement flags, the MAX of the ranks' times)
    def reduce_scalar(self, value, kind: int):
        dt = np.float32 if kind in (KIND_F32_MAX, KIND_F32_SUM) else (np.float64 if kind == KIND_F64_MAX else np.int64)
        if getattr(self, "_scalar", None) is None:
            self._scalar = self.ctx.alloc(64)
        self._scalar.upload(np.array([value], dtype=dt))
        self.all_reduce(self._scalar, 1, kind)
        return self._scalar.download(dt, 1)[0].item()


# ------------------------------------------------------------------------------------------ device backend
class _View:
    """a device pointer inside a DeviceArray"""

    def __init__(self, ptr):
        self.ptr = int(ptr)


class HipSlabBackend:
    """The C ABI on one GPU (libgsx_hip.so); buffers are grow-only DeviceArrays owned here."""

    def __init__(self, device: int = 0, stream: int | None = None, ctx=None):
        from . import _lib
        self._lib = _lib
        self.ctx = ctx if ctx is not None else _lib.Context(device, stream)
        if ctx is None:
            # the slab KNN never refines, but with this set an uneven slab (blobs, a dense object) takes the Morton-tree path
            # instead of searching over-full grid cells quadratically (one more host synchronisation per step; a context the
            # caller brings keeps the caller's setting -- bench.py times the even case without it)
            self.ctx.set_param("adaptive", 1)
        self.lib = self.ctx.lib
        self._bufs = {}

    def buf(self, name: str, nbytes: int, keep: int = 0):
        """grow-only named buffer; `keep`: bytes of the old contents that survive a reallocation"""
        cur = self._bufs.get(name)
        if cur is None or cur.nbytes < nbytes:
            new = self.ctx.alloc(int(nbytes * 1.125) + 256)
            if cur is not None:
                if keep:
                    self._chk(self.lib.gsx_dev_copy(self.ctx.handle, new.ptr, cur.ptr, min(int(keep), cur.nbytes)), "gsx_dev_copy")
                    self.ctx.synchronize()
                cur.free()
            cur = new
            self._bufs[name] = cur
        return cur

    def set_adaptive(self, on: bool):
        """adaptive KNN for the calls below: exact and fast on clouds with far floaters -- what the
        replicated exchange is the fallback for -- at the price of one host synchronisation per call"""
        self.ctx.set_param("adaptive", 1 if on else 0)

    # ---- the replicated exchange (dist.py)
    def knn_share(self, xyz_all, n_total, k, share, nshares, md_all, algo=0):
        """md_all[f32 n_total]: this share's mean distances at their original indices, +0.0 elsewhere"""
        p = xyz_all.ptr
        self.ctx.sor_knn_share(p, p + 4, p + 8, 3, int(n_total), int(k), int(share), int(nshares), md_all.ptr, algo=algo)

    def knn_all(self, xyz_all, n, k, md, algo=0):
        p = xyz_all.ptr
        self.ctx.sor_knn(p, p + 4, p + 8, 3, int(n), 0, int(n), int(k), md.ptr, algo=algo)

    def stats(self, md_all, n_total, threshold_factor, stats):
        self.ctx.sor_stats(md_all.ptr, int(n_total), float(threshold_factor), stats.ptr)

    @staticmethod
    def at(buf, byte_off: int):
        return _View(buf.ptr + int(byte_off))

    def to_host(self, buf, dtype, count):
        out = np.empty(int(count), dtype=dtype)
        if out.nbytes:
            self._lib.check(self.lib.gsx_dev_download(self.ctx.handle, out.ctypes.data, buf.ptr, out.nbytes), "gsx_dev_download")
        return out

    def from_host(self, buf, arr):
        a = np.ascontiguousarray(arr)
        if a.nbytes > (1 << 16):
            self._lib.check(self.lib.gsx_dev_upload(self.ctx.handle, buf.ptr, a.ctypes.data, a.nbytes), "gsx_dev_upload")
        elif a.nbytes:   # small control data: enqueue only, no stream synchronisation (the runtime stages pageable
            # memory before returning, so the temporary may die)
            self._lib.check(self.lib.gsx_dev_upload_async(self.ctx.handle, buf.ptr, a.ctypes.data, a.nbytes), "gsx_dev_upload_async")

    def zero(self, buf, nbytes):
        self._lib.check(self.lib.gsx_dev_memset(self.ctx.handle, buf.ptr, 0, int(nbytes)), "gsx_dev_memset")

    def _chk(self, rc, what):
        self._lib.check(rc, what)

    # rows: device pointer to (n,3) float32
    def bbox(self, rows, n, out7):
        p = rows.ptr
        self._chk(self.lib.gsx_slab_bbox_dev(self.ctx.handle, p, p + 4, p + 8, 3, int(n), out7.ptr), "gsx_slab_bbox_dev")

    def hist(self, rows, n, bbox7, hist):
        p = rows.ptr
        self._chk(self.lib.gsx_slab_hist_dev(self.ctx.handle, p, p + 4, p + 8, 3, int(n), bbox7.ptr, hist.ptr), "gsx_slab_hist_dev")

    def partition(self, rows, n, world, axis, lo, hi, cut, halo_bins, start_off, cursor, send, send_src):
        """start_off: first row of every slot of the send buffer (2*world, host); cursor: device scratch"""
        p = rows.ptr
        cuts = (C.c_int32 * (world + 1))(*[int(c) for c in cut])
        offs = (C.c_uint32 * (2 * world))(*[int(o) for o in start_off])
        planes = (C.c_float * (2 * world))()
        self._chk(self.lib.gsx_slab_partition_dev(self.ctx.handle, p, p + 4, p + 8, 3, int(n), int(world), int(axis), float(lo),
                                                  float(hi), cuts, int(halo_bins), offs, cursor.ptr, send.ptr, send_src.ptr, planes),
                  "gsx_slab_partition_dev")
        return np.array(planes[:], dtype=np.float32).reshape(world, 2)

    def copy(self, dst, src, nbytes):
        self._chk(self.lib.gsx_dev_copy(self.ctx.handle, dst.ptr, src.ptr, int(nbytes)), "gsx_dev_copy")

    def slab_step(self, rows, n_local, k, threshold_factor, halo_cells=1.5, mask_out=None):
        """the whole step in ONE C call (gsx_sor_slab_step_dev), with the context's communicator (none: world 1)"""
        st = self._lib.SlabStep()
        self._chk(self.lib.gsx_sor_slab_step_dev(self.ctx.handle, rows.ptr, int(n_local), int(k), float(threshold_factor), float(halo_cells),
                                                 mask_out.ptr if mask_out is not None else None, C.byref(st)), "gsx_sor_slab_step_dev")
        p = st.plan.as_dict()
        _decline(p)
        return SlabResult({"mask": _View(st.mask_dev), "mean_dists": _View(st.mean_dists_dev), "stats": _View(st.stats_dev),
                           "uncertain": _View(st.uncertain_dev), "_be": self, "n_total": p["n_total"], "n_own": p["n_own"],
                           "n_halo": p["n_halo"], "plan": p, "info": {"axis": p["axis"], "halo_bins": p["halo_bins"], "cut": p["cut"]}})

    def knn_slab(self, rows, n_own, n_halo, k, mean_out, kth_out):
        self._chk(self.lib.gsx_sor_knn_slab_dev(self.ctx.handle, rows.ptr, int(n_own), int(n_halo), int(k), mean_out.ptr, kth_out.ptr),
                  "gsx_sor_knn_slab_dev")

    def certify(self, rows, axis, n_own, kth, open_lo, open_hi, n_uncertain):
        self._chk(self.lib.gsx_slab_certify_dev(self.ctx.handle, rows.ptr + 4 * axis, 3, int(n_own), kth.ptr, float(open_lo),
                                                float(open_hi), n_uncertain.ptr), "gsx_slab_certify_dev")

    def unpermute(self, recv, send_src, n, out):
        self._chk(self.lib.gsx_slab_unpermute_dev(self.ctx.handle, recv.ptr, send_src.ptr, int(n), out.ptr), "gsx_slab_unpermute_dev")

    def piece_sums(self, a, n, mean, out):
        self._chk(self.lib.gsx_sor_piece_sums_dev(self.ctx.handle, a.ptr, int(n), mean.ptr if mean is not None else None, out.ptr),
                  "gsx_sor_piece_sums_dev")

    def stats_from_pieces(self, pieces, npieces, n_total, mode, factor, stats):
        self._chk(self.lib.gsx_sor_stats_from_pieces_dev(self.ctx.handle, pieces.ptr, int(npieces), int(n_total), int(mode),
                                                         float(factor), stats.ptr), "gsx_sor_stats_from_pieces_dev")

    def mask(self, md, n, stats, out):
        self.ctx.sor_mask(md.ptr, int(n), stats.ptr + 8, out.ptr)

    # ---- multi-GPU density (dist_density.py)
    def density_hist(self, rows, n, voxel, cap, keys, counts) -> int:
        p = rows.ptr
        nu = C.c_int64()
        self._chk(self.lib.gsx_density_hist_dev(self.ctx.handle, p, p + 4, p + 8, 3, int(n), float(voxel), int(cap), C.byref(nu),
                                                keys.ptr, counts.ptr), "gsx_density_hist_dev")
        return int(nu.value)

    def pad_density_list(self, keys, counts, used, upto):
        """entries [used, upto): count 0 (gsx_density_merge_dev ignores the keys of such entries)"""
        self._chk(self.lib.gsx_dev_memset(self.ctx.handle, keys.ptr + 24 * int(used), 0, 24 * int(upto - used)), "gsx_dev_memset")
        self._chk(self.lib.gsx_dev_memset(self.ctx.handle, counts.ptr + 8 * int(used), 0, 8 * int(upto - used)), "gsx_dev_memset")

    def density_merge(self, keys, counts, m, min_points, dense_cap):
        dk = np.empty((max(int(dense_cap), 1), 3), dtype=np.int64)
        dc = np.empty(max(int(dense_cap), 1), dtype=np.int64)
        nu, nd = C.c_int64(), C.c_int64()
        self._chk(self.lib.gsx_density_merge_dev(self.ctx.handle, keys.ptr, counts.ptr, int(m), int(min_points), int(dense_cap),
                                                 C.byref(nu), C.byref(nd), dk.ctypes.data, dc.ctypes.data), "gsx_density_merge_dev")
        k = int(nd.value)
        return {"n_unique": int(nu.value), "dense_keys": dk[:k].copy(), "dense_counts": dc[:k].copy()}

    def density_mask(self, rows, n, voxel, kept_keys, mask):
        p = rows.ptr
        kk = np.ascontiguousarray(kept_keys, dtype=np.int64).reshape(-1, 3)
        self._chk(self.lib.gsx_density_mask_dev(self.ctx.handle, p, p + 4, p + 8, 3, int(n), float(voxel), kk.ctypes.data, len(kk),
                                                mask.ptr), "gsx_density_mask_dev")

    def compact_rows(self, rows, mask, n, rows_out, orig_out) -> int:
        """rows[mask != 0] (order kept) -> rows_out, their local indices -> orig_out (gsx_compact_rows_dev)"""
        n_out = C.c_int64()
        self._chk(self.lib.gsx_compact_rows_dev(self.ctx.handle, rows.ptr, None, mask.ptr, int(n), rows_out.ptr, orig_out.ptr,
                                                C.byref(n_out)), "gsx_compact_rows_dev")
        return int(n_out.value)

    def check(self):
        self.ctx.check()


# ------------------------------------------------------------------------------------------ the step
def pts_per_cell(k: int, n: int = 0) -> float:
    """the KNN grid's cell population for n reference points (csrc/sor_grid.hip: knn_grid_level)"""
    m = max(2.0, 0.47 * (k + 1))
    fill = 54.0 if n >= 4_000_000 else 58.0
    for cells in (8, 4, 2, 1):
        if fill < m * cells <= 66.0:
            m = fill / cells
    return m


def plan_slabs(hist: np.ndarray, world: int):
    """equal-count cuts of the global 4096-bin histogram: slab s owns bins [cut[s], cut[s+1])"""
    cum = np.concatenate([[0], np.cumsum(hist.astype(np.int64))])
    total = int(cum[-1])
    cut = [0]
    for s in range(1, world):
        b = int(np.searchsorted(cum, (total * s) // world, side="left"))
        cut.append(min(max(b, cut[-1]), BINS))
    cut.append(BINS)
    return cut, total


def slab_counts(allhist: np.ndarray, cut, halo_bins: int):
    """rows every source sends every slab, from the per-rank histograms alone: own[src, s], halo[src, s]"""
    world = allhist.shape[0]
    cum = np.concatenate([np.zeros((world, 1), np.int64), np.cumsum(allhist.astype(np.int64), axis=1)], axis=1)
    own = np.empty((world, world), np.int64)
    halo = np.empty((world, world), np.int64)
    for s in range(world):
        a, b = cut[s], cut[s + 1]
        ha, hb = max(a - halo_bins, 0), min(b + halo_bins, BINS)
        own[:, s] = cum[:, b] - cum[:, a]
        halo[:, s] = (cum[:, hb] - cum[:, ha]) - own[:, s]
    return own, halo


SLAB_OK, SLAB_EMPTY, SLAB_NONFINITE, SLAB_SMALL_SHARD, SLAB_NO_STRUCTURE = range(5)   # gsx_slab_plan_t.status


def plan_step(words: np.ndarray, G: int, r: int, n_local: int, k: int, halo_cells: float = 1.5) -> dict:
    """The plan of one step from what its host synchronisation brings back: words[0:7] = the all-reduced box words,
    words[8:] = every rank's histogram.  The Python restatement of ``gsx_slab_plan`` (csrc/dist_slab.hip) -- same fields,
    same arithmetic (tests/test_dist_cpu.py pins one against the other); every rank computes the identical plan."""
    hb = words[:7].view(np.float32)
    allhist = words[8:8 + BINS * G].reshape(G, BINS)
    sizes = allhist.sum(1).astype(np.int64)          # every rank's shard size: the histograms hold every point once
    p = {"status": SLAB_OK, "world": G, "rank": r, "n_local": int(n_local), "sizes": [int(v) for v in sizes], "n_total": int(sizes.sum())}
    if p["n_total"] == 0:
        p["status"] = SLAB_EMPTY
        return p
    if hb[6] > 0 or not np.all(np.isfinite(hb[:6])):
        p["status"] = SLAB_NONFINITE
        return p
    ext = hb[3:6] + hb[:3]                                                     # float32, like the device (slab_axis)
    axis = 0
    if ext[1] > ext[0]:
        axis = 1
    if ext[2] > max(ext[0], ext[1]):
        axis = 2
    lo, hi = np.float32(-hb[axis]), np.float32(hb[3 + axis])
    cut, n_total = plan_slabs(allhist.sum(0), G)
    p.update(axis=axis, lo=float(lo), hi=float(hi), cut=cut)
    if int(sizes[r]) != int(n_local):
        raise ValueError("slab_sor: %d rows given, %d binned" % (n_local, int(sizes[r])))
    if G > 1 and int(sizes.min()) < NP_PIECE:
        p["status"] = SLAB_SMALL_SHARD
        return p
    ext64 = ext.astype(np.float64)
    nd = int((ext64 > 0).sum())
    vol = 1.0
    for a in range(3):
        if ext64[a] > 0:
            vol *= float(ext64[a])
    per = vol * pts_per_cell(k, n_total // G) / n_total if nd else 0.0        # (slabs are equal-COUNT)
    h_est = per ** (1.0 / nd) if nd else 0.0
    bw = (float(hi) - float(lo)) / BINS if hi > lo else 0.0
    halo_bins = int(np.ceil(halo_cells * h_est / bw)) + 1 if bw > 0 else BINS
    own, halo = slab_counts(allhist, cut, halo_bins)
    p.update(halo_bins=halo_bins, halo_total=int(halo.sum()))
    # a cloud whose halos amount to most of it (a scene inside a box inflated by far floaters: the halo width comes from the
    # box-wide density) gains nothing from slabs -- every rank would receive nearly everything, search it, and then fail the
    # certificate for the floaters anyway.  Decided from the gathered histograms: every rank declines in the same step.
    if G > 1 and int(halo.sum()) > 0.75 * (G - 1) * n_total:
        p["status"] = SLAB_NO_STRUCTURE
        return p
    own_cnt, halo_cnt = own[r], halo[r]
    in_own, in_halo = own[:, r], halo[:, r]                                    # rows every source sends me
    n_own, n_halo = int(in_own.sum()), int(in_halo.sum())
    ex = lambda c, base=0: [int(v) for v in base + np.concatenate([[0], np.cumsum(c)[:-1]])]
    p.update(own_cnt=[int(v) for v in own_cnt], halo_cnt=[int(v) for v in halo_cnt], own_off=ex(own_cnt), halo_off=ex(halo_cnt, int(n_local)),
             in_own=[int(v) for v in in_own], in_halo=[int(v) for v in in_halo], r_own_off=ex(in_own), r_halo_off=ex(in_halo, n_own),
             n_own=n_own, n_halo=n_halo, n_send=int(n_local) + int(halo_cnt.sum()))
    lo64, hi64 = np.float64(lo), np.float64(hi)
    b0, b1 = cut[r] - halo_bins, cut[r + 1] + halo_bins
    p["plane_lo"] = float("-inf") if (r == 0 or b0 <= 0) else float(np.float32(lo64 + (b0 + 0.5) * bw))
    p["plane_hi"] = float("inf") if (r == G - 1 or b1 >= BINS) else float(np.float32(lo64 + (b1 - 0.5) * bw))
    return p


def _decline(p: dict):
    st = p["status"]
    if st == SLAB_EMPTY:
        raise SlabUnsupported("empty cloud")
    if st == SLAB_NONFINITE:
        raise ValueError("sor: coordinates are not finite (NaN/inf)")
    if st == SLAB_SMALL_SHARD:
        raise SlabUnsupported("index shards of %s points: the slab exchange needs >= %d per rank" % (p["sizes"], NP_PIECE))
    if st == SLAB_NO_STRUCTURE:
        raise SlabUnsupported("the halos hold %d of %d points per neighbour: no slab structure to exploit"
                              % (p["halo_total"] // max(p["world"] - 1, 1), p["n_total"]))


class SlabResult(dict):
    """buffers of one step.  The certificate is evaluated on the device and read back lazily: call ``check()`` (one
    synchronisation) before trusting the buffers; it raises SlabUncertain when the slabs could not certify every query."""

    def check(self):
        be = self["_be"]
        n_unc = int(be.to_host(self["uncertain"], np.int64, 1)[0])
        be.check()
        if n_unc:
            raise SlabUncertain("%d queries could not be certified inside their slab (halo of %d bins)" % (n_unc, self["info"]["halo_bins"]))
        return self


def _finish(out: SlabResult, be, n_local: int, want_host: bool):
    if want_host:
        out.check()
        out["mask_host"] = be.to_host(out["mask"], np.uint8, n_local).view(np.bool_)
        out["mean_dists_host"] = be.to_host(out["mean_dists"], np.float32, n_local)
        out["stats_host"] = be.to_host(out["stats"], np.float32, 3)
    return out


def slab_sor(be, comm, rows, n_local: int, k: int, threshold_factor: float, halo_cells: float = 1.5, want_host: bool = False,
             fused: bool | None = None):
    """rows: backend buffer holding this rank's (n_local,3) float32 index shard -- consecutive index ranges of the cloud in
    rank order, of ANY sizes >= 8192 (unequal shards, e.g. what a density filter leaves behind on every rank).
    -> SlabResult(mask, mean_dists, stats: backend buffers of the LOCAL index range; n_total, n_own, n_halo, info).
    ONE host synchronisation inside the step (the histograms, from which every size follows).
    On a HipSlabBackend whose context carries the communicator the whole step is ONE C call (``gsx_sor_slab_step_dev``);
    ``fused=False`` (and every other backend / communicator) takes the spelled-out ``slab_sor_steps``.
    halo_cells: halo width in KNN cell edges h of the global density.  On uniform data a query's k-th neighbour is at
    ~0.82 h and beyond 1.3 h with probability < 1e-13; the certificate catches whatever the halo does not cover."""
    can_fuse = hasattr(be, "slab_step") and (comm is None or (isinstance(comm, RcclComm) and comm.ctx is be.ctx))
    if fused is None:
        fused = can_fuse
    if fused:
        if not can_fuse:
            raise ValueError("slab_sor(fused=True) needs a HipSlabBackend and the RcclComm of its context")
        return _finish(be.slab_step(rows, n_local, k, threshold_factor, halo_cells), be, int(n_local), want_host)
    return slab_sor_steps(be, comm, rows, n_local, k, threshold_factor, halo_cells, want_host)


class _OneRank:
    """no communicator: world 1, the local block of an exchange is a device copy"""
    rank, world = 0, 1

    def __init__(self, be):
        self.be = be

    def all_to_all_v(self, send, send_off, send_cnt, recv, recv_off, recv_cnt, elem_bytes):
        if int(send_cnt[0]):
            self.be.copy(self.be.at(recv, int(recv_off[0]) * elem_bytes), self.be.at(send, int(send_off[0]) * elem_bytes), int(send_cnt[0]) * elem_bytes)


def slab_sor_steps(be, comm, rows, n_local: int, k: int, threshold_factor: float, halo_cells: float = 1.5, want_host: bool = False):
    """the step, call by call (what gsx_sor_slab_step_dev does inside the library)"""
    if comm is None:
        comm = _OneRank(be)
    G, r = comm.world, comm.rank
    n_local = int(n_local)
    # ---- 1. bounding box (device-resident), 2. histogram of the longest axis; all-gathered: cuts AND every row count.
    # Box and histograms share one buffer, so the step's host synchronisation is ONE download.
    plan_in = be.buf("plan_in", 32 + 4 * BINS * G)
    b7 = be.at(plan_in, 0)
    if n_local:
        be.bbox(rows, n_local, b7)
    else:   # an empty shard still takes part in the collectives (and then declines with every other rank, below)
        be.from_host(b7, np.array([-np.inf] * 6 + [0.0], dtype=np.float32))
    if G > 1:
        comm.all_reduce(b7, 7, KIND_F32_MAX)
        hist = be.buf("hist", 4 * BINS)
        if n_local:
            be.hist(rows, n_local, b7, hist)
        else:
            be.zero(hist, 4 * BINS)
        comm.all_gather(hist, be.at(plan_in, 32), 4 * BINS)
    elif n_local:
        be.hist(rows, n_local, b7, be.at(plan_in, 32))
    else:
        be.zero(be.at(plan_in, 32), 4 * BINS)
    words = be.to_host(plan_in, np.uint32, 8 + BINS * G)                       # <- the step's host synchronisation
    p = plan_step(words, G, r, n_local, k, halo_cells)
    _decline(p)
    axis, cut, halo_bins, n_total = p["axis"], p["cut"], p["halo_bins"], p["n_total"]
    sizes = np.array(p["sizes"], dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(sizes)])  # global index of every shard's first row
    # ---- 3. scatter into the send buffer, exchange the rows (sizes from the histograms: no counting pass)
    own_off, own_cnt, halo_off, halo_cnt = p["own_off"], p["own_cnt"], p["halo_off"], p["halo_cnt"]
    cursor = be.buf("cursor", 4 * 2 * G)
    cur = np.empty(2 * G, np.uint32)
    cur[0::2], cur[1::2] = own_off, halo_off
    send = be.buf("send", 12 * max(p["n_send"], 1))
    send_src = be.buf("send_src", 4 * max(n_local, 1))
    be.partition(rows, n_local, G, axis, np.float32(p["lo"]), np.float32(p["hi"]), cut, halo_bins, cur, cursor, send, send_src)
    in_own, in_halo, r_own_off, r_halo_off = p["in_own"], p["in_halo"], p["r_own_off"], p["r_halo_off"]
    n_own, n_halo = p["n_own"], p["n_halo"]
    slab = be.buf("slab", 12 * max(n_own + n_halo, 1))
    if hasattr(comm, "all_to_all_segs"):   # own rows and halo rows in ONE group of sends / receives
        comm.all_to_all_segs(send, slab, [(own_off, own_cnt, r_own_off, in_own), (halo_off, halo_cnt, r_halo_off, in_halo)], 12)
    else:
        comm.all_to_all_v(send, own_off, own_cnt, slab, r_own_off, in_own, 12)
        comm.all_to_all_v(send, halo_off, halo_cnt, slab, r_halo_off, in_halo, 12)
    # ---- 4./5. exact KNN on the slab; certificate (device-side count, summed over the ranks, read by check())
    md_slab = be.buf("md_slab", 4 * max(n_own, 1))
    kth = be.buf("kth", 8 * max(n_own, 1))
    if n_own:
        be.knn_slab(slab, n_own, n_halo, k, md_slab, kth)
    unc = be.buf("unc", 8)
    if n_own:
        be.certify(slab, axis, n_own, kth, p["plane_lo"], p["plane_hi"], unc)   # (zeroes the counter itself)
    else:
        be.zero(unc, 8)
    if G > 1:
        comm.all_reduce(unc, 1, KIND_I64_SUM)
    # ---- 6. mean distances back to the index owners, original order
    ret = be.buf("ret", 4 * max(n_local, 1))
    comm.all_to_all_v(md_slab, r_own_off, in_own, ret, own_off, own_cnt, 4)
    md = be.buf("md", 4 * (n_local + NP_PIECE + 4))
    be.unpermute(ret, send_src, n_local, md)
    # ---- 7. numpy-exact statistics from 8192-element piece sums
    stats = be.buf("stats", 16)
    heads = [(-int(starts[q])) % NP_PIECE for q in range(G)] + [0]              # leading elements owed to the left neighbour
    head, nxt_head = heads[r], heads[r + 1]
    # the piece sums read 16 bytes at a time: where the shard's first own piece does not start on a 16-byte boundary
    # (shard starts that are not multiples of 4) the elements are first moved to an aligned buffer (4 B/point, device copy)
    if head % 4 == 0:
        st_in, st_tail = be.at(md, 4 * head), be.at(md, 4 * n_local)
    else:
        st_buf = be.buf("md_stats", 4 * (n_local + NP_PIECE + 4))
        be.copy(st_buf, be.at(md, 4 * head), 4 * (n_local - head))
        st_in, st_tail = st_buf, be.at(st_buf, 4 * (n_local - head))
    if G > 1:
        s_off, s_cnt, r_off, r_cnt = [0] * G, [0] * G, [0] * G, [0] * G
        if r > 0:
            s_cnt[r - 1] = head
        if r + 1 < G:
            r_cnt[r + 1] = nxt_head
        comm.all_to_all_v(md, s_off, s_cnt, st_tail, r_off, r_cnt, 4)
    n_mine = n_local - head + nxt_head
    counts = [-(-(int(sizes[q]) - heads[q] + heads[q + 1]) // NP_PIECE) for q in range(G)]
    max_pieces = max(counts)
    pieces = be.buf("pieces", 4 * max_pieces)
    allp = be.buf("allpieces", 4 * max_pieces * G)
    packed = be.buf("packed", 4 * max_pieces * G)
    for mode in (0, 1):
        be.piece_sums(st_in, n_mine, stats if mode else None, pieces)
        if G > 1:
            comm.all_gather(pieces, allp, 4 * max_pieces)
            o = 0
            for q in range(G):   # drop the padding: a handful of small device-to-device copies
                be.copy(be.at(packed, 4 * o), be.at(allp, 4 * max_pieces * q), 4 * counts[q])
                o += counts[q]
            be.stats_from_pieces(packed, o, n_total, mode, threshold_factor, stats)
        else:
            be.stats_from_pieces(pieces, counts[0], n_total, mode, threshold_factor, stats)
    # ---- 8. mask of the local index range
    mask = be.buf("mask", n_local + 4)
    be.mask(md, n_local, stats, mask)
    out = SlabResult({"mask": mask, "mean_dists": md, "stats": stats, "uncertain": unc, "_be": be, "n_total": n_total,
                      "n_own": n_own, "n_halo": n_halo, "info": {"axis": axis, "halo_bins": halo_bins, "cut": cut}})
    return _finish(out, be, n_local, want_host)
