"""numpy stand-in for 3dgsconverter_amd.dist_slab.HipSlabBackend: the same method names on host arrays.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  It lets tests/test_dist_cpu.py run the multi-GPU slab
choreography (3dgsconverter_amd/dist_slab.py: slab_sor) with torch.distributed's gloo backend on CPU, world 2
and 3; the arithmetic is the oracle's (cKDTree exactly as data_processor.py:156-173, numpy's own float32
reductions for :176-178).  The binning formulas restate csrc/dist_slab.hip so that both sides cut identical slabs.
"""
from __future__ import annotations

import numpy as np

from . import sor as osor

BINS = 4096
NP_PIECE = 8192


class _Buf:
    def __init__(self, nbytes):
        self.a = np.zeros(int(nbytes), dtype=np.uint8)
        self.nbytes = int(nbytes)

    def view(self, dtype, count, byte_off=0):
        return self.a[byte_off:byte_off + int(count) * np.dtype(dtype).itemsize].view(dtype)


class _At:
    def __init__(self, buf, off):
        self.buf, self.off = buf, int(off)

    def view(self, dtype, count, byte_off=0):
        return self.buf.view(dtype, count, self.off + byte_off)


class NumpySlabBackend:
    def __init__(self):
        self._bufs = {}

    def buf(self, name, nbytes, keep=0):
        cur = self._bufs.get(name)
        if cur is None or cur.nbytes < nbytes:
            new = _Buf(int(nbytes) + 64)
            if cur is not None and keep:
                new.a[:min(int(keep), cur.nbytes)] = cur.a[:min(int(keep), cur.nbytes)]
            cur = new
            self._bufs[name] = cur
        return cur

    @staticmethod
    def at(buf, byte_off):
        return _At(buf, byte_off)

    @staticmethod
    def rows_buffer(xyz):
        b = _Buf(xyz.nbytes)
        b.view(np.float32, xyz.size)[:] = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1)
        return b

    def to_host(self, buf, dtype, count):
        return buf.view(dtype, count).copy()

    def from_host(self, buf, arr):
        a = np.ascontiguousarray(arr)
        buf.view(np.uint8, a.nbytes)[:] = a.view(np.uint8).reshape(-1)

    def zero(self, buf, nbytes):
        buf.view(np.uint8, nbytes)[:] = 0

    # ---- the device entry points, restated
    def bbox(self, rows, n, out7):
        x = rows.view(np.float32, 3 * n).reshape(n, 3)
        o = out7.view(np.float32, 7)
        o[:6] = -np.inf
        o[6] = 0.0
        if n:
            o[:3] = (-x).max(0)
            o[3:6] = x.max(0)
            o[6] = 0.0 if np.isfinite(x).all() else 1.0

    @staticmethod
    def _bin(c, lo, hi):
        inv_w = np.float32(BINS) / (np.float32(hi) - np.float32(lo)) if hi > lo else np.float32(0)
        return np.clip(((c - np.float32(lo)) * inv_w).astype(np.int32), 0, BINS - 1)

    @staticmethod
    def _axis(b7):
        e = b7[3:6] + b7[:3]    # float32, as csrc/dist_slab.hip: slab_axis
        axis = 0
        if e[1] > e[0]:
            axis = 1
        if e[2] > max(e[0], e[1]):
            axis = 2
        return axis, np.float32(-b7[axis]), np.float32(b7[3 + axis])

    def hist(self, rows, n, bbox7, hist):
        axis, lo, hi = self._axis(bbox7.view(np.float32, 7))
        c = rows.view(np.float32, 3 * n).reshape(n, 3)[:, axis]
        hist.view(np.uint32, BINS)[:] = np.bincount(self._bin(c, lo, hi), minlength=BINS)

    def partition(self, rows, n, world, axis, lo, hi, cut, halo_bins, start_off, cursor, send, send_src):
        x = rows.view(np.float32, 3 * n).reshape(n, 3)
        b = self._bin(x[:, axis], lo, hi)
        owner = np.searchsorted(np.asarray(cut[1:world], dtype=np.int64), b, side="right")
        lo64, hi64 = np.float64(np.float32(lo)), np.float64(np.float32(hi))
        bw = (hi64 - lo64) / BINS if hi > lo else 0.0
        pl = np.empty((world, 2), np.float32)
        cur = cursor.view(np.uint32, 2 * world)
        cur[:] = np.asarray(start_off, dtype=np.uint32)
        out = send.view(np.float32, 3 * (send.nbytes // 12)).reshape(-1, 3)
        src = send_src.view(np.uint32, n)
        for s in range(world):
            b0, b1 = cut[s] - halo_bins, cut[s + 1] + halo_bins
            pl[s, 0] = -np.inf if (s == 0 or b0 <= 0) else np.float32(lo64 + (b0 + 0.5) * bw)
            pl[s, 1] = np.inf if (s == world - 1 or b1 >= BINS) else np.float32(lo64 + (b1 - 0.5) * bw)
            idx = np.nonzero(owner == s)[0]
            o = int(cur[2 * s])
            out[o:o + len(idx)] = x[idx]
            src[o:o + len(idx)] = idx
            cur[2 * s] += len(idx)
            hidx = np.nonzero((owner != s) & (b >= b0) & (b < b1))[0]
            o = int(cur[2 * s + 1])
            out[o:o + len(hidx)] = x[hidx]
            cur[2 * s + 1] += len(hidx)
        return pl

    def copy(self, dst, src, nbytes):
        dst.view(np.uint8, nbytes)[:] = src.view(np.uint8, nbytes)

    def knn_slab(self, rows, n_own, n_halo, k, mean_out, kth_out):
        from scipy.spatial import cKDTree
        pts = rows.view(np.float32, 3 * (n_own + n_halo)).reshape(-1, 3)
        tree = cKDTree(pts)
        d, _ = tree.query(pts[:n_own], k=k + 1)   # data_processor.py:160-173
        d = np.asarray(d).reshape(n_own, -1)
        mean_out.view(np.float32, n_own)[:] = np.mean(d[:, 1:], axis=1).astype(np.float32)
        kth_out.view(np.float64, n_own)[:] = d[:, -1] ** 2 if d.shape[1] == k + 1 else np.inf

    def certify(self, rows, axis, n_own, kth, open_lo, open_hi, n_uncertain):
        c = rows.view(np.float32, 3 * n_own).reshape(-1, 3)[:, axis].astype(np.float64)
        d = np.minimum(c - np.float64(open_lo), np.float64(open_hi) - c) * (1.0 - 1e-6)
        # d**2 of a sqrt-ed cKDTree distance is not the exact squared distance: 1e-12 relative slack keeps this
        # stand-in from flagging what the device (which has the exact value) certifies
        bad = ~(kth.view(np.float64, n_own) * (1.0 - 1e-12) <= d * d)
        n_uncertain.view(np.uint32, 1)[0] = np.uint32(bad.sum())

    def unpermute(self, recv, send_src, n, out):
        out.view(np.float32, n)[send_src.view(np.uint32, n)] = recv.view(np.float32, n)

    def piece_sums(self, a, n, mean, out):
        v = a.view(np.float32, n)
        if mean is not None:
            m = mean.view(np.float32, 1)[0]
            v = (v - m) * (v - m)    # float32, as np.std's x - mean, then squared
        npieces = -(-n // NP_PIECE)
        o = out.view(np.float32, npieces)
        for p in range(npieces):
            o[p] = np.add.reduce(v[p * NP_PIECE:(p + 1) * NP_PIECE])   # <= 8192 elements: one pairwise sum
        return npieces

    def stats_from_pieces(self, pieces, npieces, n_total, mode, factor, stats):
        acc = np.float32(0.0)
        for v in pieces.view(np.float32, npieces):
            acc = np.float32(acc + v)
        q = np.float32(np.float64(acc) / np.float64(n_total))
        st = stats.view(np.float32, 3)
        if mode == 0:
            st[0] = q
        else:
            st[1] = np.sqrt(q)
            st[2] = st[0] + np.float32(factor) * st[1]

    def mask(self, md, n, stats, out):
        out.view(np.uint8, n)[:] = md.view(np.float32, n) < stats.view(np.float32, 3)[2]

    # ---- multi-GPU density (3dgsconverter_amd/dist_density.py): the reference's own expressions (data_processor.py:38-52)
    def density_hist(self, rows, n, voxel, cap, keys, counts):
        from . import density as oden
        x = rows.view(np.float32, 3 * n).reshape(n, 3)
        uniq, cnt = np.unique(oden.voxel_keys(x, voxel), axis=0, return_counts=True)
        keys.view(np.int64, 3 * len(uniq))[:] = uniq.reshape(-1)
        counts.view(np.int64, len(uniq))[:] = cnt
        return len(uniq)

    def pad_density_list(self, keys, counts, used, upto):
        keys.view(np.int64, 3 * upto)[3 * used:] = 0
        counts.view(np.int64, upto)[used:] = 0

    def density_merge(self, keys, counts, m, min_points, dense_cap):
        k = keys.view(np.int64, 3 * m).reshape(m, 3)
        c = counts.view(np.int64, m)
        live = c > 0
        uniq, inv = np.unique(k[live], axis=0, return_inverse=True)
        tot = np.bincount(inv.reshape(-1), weights=c[live].astype(np.float64), minlength=len(uniq)).astype(np.int64)
        dense = tot >= min_points
        return {"n_unique": len(uniq), "dense_keys": uniq[dense], "dense_counts": tot[dense]}

    def density_mask(self, rows, n, voxel, kept_keys, mask):
        from . import density as oden
        x = rows.view(np.float32, 3 * n).reshape(n, 3)
        kept = set(map(tuple, np.asarray(kept_keys).reshape(-1, 3).tolist()))
        keys = oden.voxel_keys(x, voxel)
        mask.view(np.uint8, n)[:] = np.fromiter((tuple(k) in kept for k in keys.tolist()), dtype=np.uint8, count=n)

    def compact_rows(self, rows, mask, n, rows_out, orig_out):
        x = rows.view(np.float32, 3 * n).reshape(n, 3)
        keep = mask.view(np.uint8, n).astype(bool)
        m = int(keep.sum())
        rows_out.view(np.float32, 3 * m)[:] = x[keep].reshape(-1)
        orig_out.view(np.uint32, m)[:] = np.nonzero(keep)[0]
        return m

    def check(self):
        pass

    # ---- the replicated exchange (3dgsconverter_amd/dist.py)
    def knn_share(self, xyz_all, n_total, k, share, nshares, md_all, algo=0):
        # any partition of the queries works for the choreography; like the GPU one this is spatial (slabs along z),
        # i.e. scattered in index space
        xyz = xyz_all.view(np.float32, 3 * n_total).reshape(n_total, 3)
        md = osor.mean_dists_ckdtree(xyz, k, workers=2)
        order = np.argsort(xyz[:, 2], kind="stable")
        mine = order[n_total * share // nshares: n_total * (share + 1) // nshares]
        out = md_all.view(np.float32, n_total)
        out[:] = 0
        out[mine] = md[mine]

    def knn_all(self, xyz_all, n, k, md, algo=0):
        xyz = xyz_all.view(np.float32, 3 * n).reshape(n, 3)
        md.view(np.float32, n)[:] = osor.mean_dists_ckdtree(xyz, k, workers=2)

    def stats(self, md_all, n_total, threshold_factor, stats):
        stats.view(np.float32, 3)[:] = [np.float32(v) for v in osor.threshold_numpy(md_all.view(np.float32, n_total), threshold_factor)]


class GlooHostComm:
    """torch.distributed (gloo) on host memory: the wire of the CPU tests of the multi-GPU choreography (the product's is
    gsx_comm_*: RCCL, or the shared-memory hostwire).  Buffers go through the backend's to_host / from_host."""

    def __init__(self, backend, group=None):
        import torch.distributed as dist
        self.dist, self.group, self.be = dist, group, backend
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def all_reduce(self, buf, count, kind):
        import torch
        dt = {0: np.float32, 1: np.float32, 2: np.int64, 3: np.float64, 4: np.int64}[kind]
        op = {0: self.dist.ReduceOp.MAX, 1: self.dist.ReduceOp.SUM, 2: self.dist.ReduceOp.SUM, 3: self.dist.ReduceOp.MAX,
              4: self.dist.ReduceOp.MIN}[kind]
        t = torch.from_numpy(self.be.to_host(buf, dt, count))
        self.dist.all_reduce(t, op=op, group=self.group)
        self.be.from_host(buf, t.numpy())

    def all_gather(self, send, recv, nbytes):
        import torch
        a = torch.from_numpy(self.be.to_host(send, np.uint8, nbytes))
        out = torch.empty(self.world * nbytes, dtype=torch.uint8)
        self.dist.all_gather_into_tensor(out, a, group=self.group)
        self.be.from_host(recv, out.numpy())

    def all_to_all_v(self, send, send_off, send_cnt, recv, recv_off, recv_cnt, elem_bytes):
        import torch
        total_s = max(int(o + c) for o, c in zip(send_off, send_cnt))
        total_r = max(int(o + c) for o, c in zip(recv_off, recv_cnt))
        s = self.be.to_host(send, np.uint8, total_s * elem_bytes)
        r = self.be.to_host(recv, np.uint8, total_r * elem_bytes) if total_r else np.zeros(0, np.uint8)
        reqs, bufs = [], []
        for p in range(self.world):
            so, sc, ro, rc = int(send_off[p]), int(send_cnt[p]), int(recv_off[p]), int(recv_cnt[p])
            if p == self.rank:
                r[ro * elem_bytes:(ro + rc) * elem_bytes] = s[so * elem_bytes:(so + sc) * elem_bytes]
                continue
            if sc:
                reqs.append(self.dist.isend(torch.from_numpy(s[so * elem_bytes:(so + sc) * elem_bytes].copy()), p, group=self.group))
            if rc:
                t = torch.empty(rc * elem_bytes, dtype=torch.uint8)
                bufs.append((ro, rc, t))
                reqs.append(self.dist.irecv(t, p, group=self.group))
        for q in reqs:
            q.wait()
        for ro, rc, t in bufs:
            r[ro * elem_bytes:(ro + rc) * elem_bytes] = t.numpy()
        if total_r:
            self.be.from_host(recv, r)


def reference_result(xyz_all, k, threshold_factor):
    """what the whole (un-sharded) cloud gives: oracle/sor.py (cKDTree + numpy, data_processor.py:156-180)"""
    return osor.sor(np.ascontiguousarray(xyz_all, dtype=np.float32), k, threshold_factor)
