"""The SOG writer's numeric core on a device-resident splat table (C ABI: the ``gsx_sog_*_dev`` block of include/gsx_hip.h,
kernels in csrc/sog_table.hip).

``encode(data, compression_level)`` takes the reference's structured table and returns every array ``SogFormat.write``
(formats/sog.py:249-639) hands to ``write_webp`` plus the numbers of its ``meta.json`` -- without the host passes the
reference makes over the table (``data[indices]`` :265, the ``column_stack`` s of :315 and :499-503, ``np.concatenate`` :391,
:434): the raw rows cross PCIe ONCE, texels (4 bytes per splat and image) come back.

  host                                           device
  ---------------------------------------------  ---------------------------------------------------------------------
  field offsets of the dtype, upload of the rows  sog_scan: sort keys of x, y, z; extremes; non-zero mask of f_rest_i
  band detection from the mask (:461-493)
  window around the extremes of x, y, z           sog_extremes: the few values inside the window
  numpy's OWN log-transform of those -> the
  np.min / np.max of :287-288, bit for bit
                                                  sog_order (lexsort, :264), sog_gather (data[indices], :265)
                                                  means / quats texels; scalar codebooks (50 000-sample gathered on the
  np.random: sample indices, initial centroids    device, sorted-run solver, :392-449) -> scales / sh0 texels;
  (indices only -- the rows never come back)      palette: ONE batched Lloyd call over the chunks (:527-552), centroid
                                                  codebook (:561) + quantiser, labels texels
  numpy's log / exp for the listed texels         <- texels + compact lists of (texel, value) next to a rounding boundary

Random draws: the reference calls ``np.random.choice(m, k, replace=False)`` -- a full permutation of ``m`` per call, 64 x 156 250
+ 2 x 30 M elements at 10M splats (~0.7 s of host time) -- from numpy's unseeded global stream.  Here ONE generator is seeded
from that global stream (so ``np.random.seed(s)`` before a write still makes the bundle reproducible) and draws the same
distribution -- uniform without replacement -- with Floyd's algorithm; the sample differs from the legacy call's, the
statistics do not.  The host-staged writer (``sog_writer._encode_host``) keeps the legacy calls.
"""
from __future__ import annotations

import ctypes as C
import threading
import time

import numpy as np

from .. import _lib
from .. import dist_palette
from ..utils import debug_print, status_print

MIN_ROWS = 1024        # below: the host-staged path (k >= N shortcuts of gpu_ops.py:30-31 live there)
EXTREME_CAP = 16384    # candidates per list for np.min / np.max of a log-transformed axis


class NotEligible(Exception):
    """the table (or the request) is one the device-resident path does not take; the caller uses the host-staged path"""


def _log_transform(v):
    return np.sign(v) * np.log(np.abs(v) + 1.0)          # formats/sog.py:280-281, numpy's own float32 arithmetic


def texture_size(n):
    width = int(np.ceil(np.sqrt(n) / 4) * 4)            # :260-261
    height = int(np.ceil(n / width / 4) * 4)
    return width, height


def sh_coeffs_present(names):
    """:466-474: bands by the NUMBER of f_rest fields present -> coefficients per splat (0, 9, 24, 45)"""
    if "f_rest_0" not in names:
        return 0
    count = sum(1 for i in range(45) if "f_rest_%d" % i in names)
    return 45 if count >= 45 else 24 if count >= 24 else 9 if count >= 9 else 0


def bands_from_mask(coeffs_present, nonzero_mask):
    """:476-491: downgrade when the trailing coefficients are all zero"""
    bands = {0: 0, 9: 1, 24: 2, 45: 3}[coeffs_present]
    if bands > 0:
        last = -1
        for i in range({3: 44, 2: 23, 1: 8}[bands], -1, -1):
            if (nonzero_mask >> i) & 1:
                last = i
                break
        bands = 3 if last >= 24 else 2 if last >= 9 else 1 if last >= 0 else 0
    return bands


def table_layout(data: np.ndarray):
    """-> (rows: a C-contiguous array whose bytes are the table, SogLayout) ; raises NotEligible"""
    if data.ndim != 1 or data.dtype.fields is None:
        raise NotEligible("not a structured table")
    fields = data.dtype.fields
    coeffs = sh_coeffs_present(data.dtype.names)
    names = _lib.SOG_FIELD_NAMES[:14 + coeffs]
    for nm in names[:14]:
        if nm not in fields:
            raise NotEligible("field %s is missing" % nm)      # the host-staged path raises what the reference raises
    if any(nm not in fields or fields[nm][0] != np.dtype("<f4") for nm in names):
        raise NotEligible("a field the writer reads is not float32")
    lay = _lib.SogLayout()
    lay.n_rest = coeffs
    aligned = data.dtype.itemsize % 4 == 0 and all(fields[nm][1] % 4 == 0 for nm in names)
    # (rows off the 4-byte grid -- the table widened by three u1 colour fields, which is what the reference's converter hands the SOG
    #  writer, converter.py:243-252: 251 bytes -- are read as they are: the kernels assemble a field from two words of their LDS tile)
    direct = data.flags.c_contiguous and data.dtype.itemsize <= (512 if aligned else 500)
    if direct:
        lay.row_bytes = data.dtype.itemsize
        for i, nm in enumerate(names):
            lay.offset[i] = int(fields[nm][1])
        return data, lay
    # rows of more than 500 / 512 bytes (many extra fields) or a table that is not contiguous: ONE threaded host pass packs the
    # float32 columns the writer reads into rows of their own
    packed = _lib.host_gather_xyz(data, names)
    lay.row_bytes = 4 * len(names)
    for i in range(len(names)):
        lay.offset[i] = 4 * i
    return packed, lay


def _draws():
    """one generator per write, seeded from numpy's global stream (module docstring)"""
    return np.random.default_rng([int(v) for v in np.random.randint(0, 2 ** 31 - 1, size=4)])


class _Stages:
    def __init__(self, ctx, on):
        self.ctx, self.on, self.ms, self.t = ctx, on, {}, time.perf_counter()

    def mark(self, name):
        if self.on:
            if self.on != "host":      # "host": the host thread's own clock, no extra synchronisation
                self.ctx.synchronize()
            now = time.perf_counter()
            self.ms[name] = self.ms.get(name, 0.0) + (now - self.t) * 1e3
            self.t = now


def encode(data: np.ndarray, compression_level: int = 0, device: int = 0, profile: bool = False, max_iter: int = 10) -> dict:
    """-> dict: n, width, height, textures (name -> (texels, 4) uint8: means_l, means_u, quats, scales, sh0[, shN_labels]),
    mins / maxs (np.float32 x 3), scale_codebook / color_codebook (float32[256] ascending), bands, and for bands > 0:
    palette (count), shn_codebook (float64[<=256]), shn_centroid_index (uint8[palette * coeffs]); stats: uncertain texels;
    stage_ms when profile.  Raises NotEligible before any device work when the table is not one this path takes."""
    lib = _lib.require_hip()
    n = len(data)
    if n < MIN_ROWS or n >= (1 << 30):
        raise NotEligible("table size")
    rows, lay = table_layout(data)
    coeffs_present = int(lay.n_rest)
    plan = dist_palette.palette_plan(n, int(compression_level))
    bounds = [(i * plan["chunk_size"], min((i + 1) * plan["chunk_size"], n)) for i in range(plan["num_chunks"])]
    bounds = [(s, e) for s, e in bounds if s < e]
    k = plan["k_per_chunk"]
    if coeffs_present and any(e - s <= k for s, e in bounds):
        raise NotEligible("a palette chunk with k >= rows")
    width, height = texture_size(n)
    texels = width * height
    t_enter = time.perf_counter()
    ar = _lib.arena(device)        # work buffers and contexts outlive the call (DeviceArena: no hipMalloc / hipFree in a repeated write)
    ctx = ar.ctx
    st = _Stages(ctx, profile)
    st.t = t_enter
    names = iter(range(1 << 30))

    def alloc(nbytes, name=None):
        return ar.buf("sog_%s" % (name if name is not None else next(names)), nbytes)

    out = {"n": n, "width": width, "height": height, "textures": {}, "stats": {}}
    try:
        n_img = 5 + (1 if coeffs_present else 0)
        host_tex_all = np.empty((n_img, texels, 4), np.uint8)
        # (touching the result pages during the upload -- _lib.prefault, what the compressed-PLY writer does -- slowed the upload's own
        #  page pinning by as much as it saved here, where the download hides behind the palette anyway: not done)
        # ---- the write's random draws (the 50 000-scalar samples of the two codebook fits, :392-397 / :436-440, and the palette's
        # initial centroids) need only n: a helper thread makes them while this one sits in the upload (6-7 ms at 10M splats)
        rng = _draws()
        drawn, draw_err = {}, []

        def draw_all():
            try:
                for which in ("scales", "sh0"):
                    drawn[which] = np.ascontiguousarray(rng.choice(3 * n, 50000, replace=False), dtype=np.int64) if 3 * n > 50000 else None
                if coeffs_present:
                    drawn["init"] = np.concatenate([s + rng.choice(e - s, k, replace=False) for s, e in bounds]).astype(np.int64)
            except BaseException as e:
                draw_err.append(e)
        drawer = threading.Thread(target=draw_all, name="gsx-sog-draws")
        drawer.start()
        # ---- the table crosses PCIe once
        d_rows = alloc(rows.nbytes)
        st.mark("alloc_rows")
        _lib.upload_table(lib, ctx, d_rows.ptr, rows)
        st.mark("upload")
        d_keys = alloc(12 * n)
        scan = _lib.SogScan()
        _lib.check(lib.gsx_sog_scan_dev(ctx.handle, d_rows.ptr, C.byref(lay), n, d_keys.ptr, C.byref(scan)), "gsx_sog_scan_dev")
        if scan.nonfinite:
            raise NotEligible("non-finite coordinates")
        bands = bands_from_mask(coeffs_present, int(scan.rest_nonzero))
        coeffs = [0, 9, 24, 45][bands]
        out["bands"] = bands
        debug_print(f"[DEBUG] SOG Write: Effective SH Bands detected: {bands}")
        st.mark("scan")

        # ---- np.min / np.max of the log-transformed axes (:287-288) from the values next to the extremes of x, y, z: the
        # transform is monotone, a relative window of 1e-3 holds every candidate (same rule as _lib.sog_positions)
        vmin, vmax = np.array(scan.vmin, np.float32), np.array(scan.vmax, np.float32)
        span = np.float32(1e-3) * np.maximum(np.abs(vmin), np.abs(vmax)) + np.float32(1e-30)
        lo_t, hi_t = (vmin + span).astype(np.float32), (vmax - span).astype(np.float32)
        cand = np.empty((6, EXTREME_CAP), np.float32)
        counts = np.zeros(6, np.int64)
        _lib.check(lib.gsx_sog_extremes_dev(ctx.handle, d_keys.ptr, n, lo_t.ctypes.data, hi_t.ctypes.data, EXTREME_CAP,
                                            cand.ctypes.data, counts.ctypes.data), "gsx_sog_extremes_dev")
        mins, maxs, arg_mn, arg_mx = [], [], [], []
        for a, name in enumerate("xyz"):
            if counts[2 * a] > EXTREME_CAP or counts[2 * a + 1] > EXTREME_CAP:
                lo_c = hi_c = np.ascontiguousarray(data[name])       # a degenerate axis: the reference's expression on the column
            else:
                lo_c, hi_c = cand[2 * a, :counts[2 * a]], cand[2 * a + 1, :counts[2 * a + 1]]
            t_lo, t_hi = _log_transform(lo_c), _log_transform(hi_c)
            i_lo, i_hi = int(np.argmin(t_lo)), int(np.argmax(t_hi))
            mins.append(t_lo[i_lo])                                  # == np.min / np.max of the whole transformed axis (:287-288)
            maxs.append(t_hi[i_hi])
            arg_mn.append(lo_c[i_lo])
            arg_mx.append(hi_c[i_hi])
        out["mins"], out["maxs"] = mins, maxs
        st.mark("extremes")

        # ---- lexsort + data[indices]
        d_perm = alloc(4 * n)
        _lib.check(lib.gsx_sog_order_dev(ctx.handle, d_keys.ptr, n, d_perm.ptr), "gsx_sog_order_dev")
        st.mark("order")
        d_pos, d_rot, d_scale, d_dc, d_op = alloc(12 * n), alloc(16 * n), alloc(12 * n), alloc(12 * n), alloc(4 * n)
        d_sh = alloc(4 * n * coeffs) if coeffs else None
        _lib.check(lib.gsx_sog_gather_dev(ctx.handle, d_rows.ptr, C.byref(lay), d_perm.ptr, n, coeffs, d_pos.ptr, d_rot.ptr, d_scale.ptr,
                                          d_dc.ptr, d_op.ptr, d_sh.ptr if d_sh else None), "gsx_sog_gather_dev")
        ctx.synchronize()
        st.mark("gather")

        # ---- positions, rotations
        cap = n // 8 + 4096
        d_list, d_cnt = alloc(8 * cap, "list"), alloc(16)
        # the texel images are slices of ONE device buffer and come back as slices of ONE host array: a single bulk copy
        tex_names = ["means_l", "means_u", "quats", "scales", "sh0"] + (["shN_labels"] if coeffs else [])
        d_tex = alloc(4 * texels * len(tex_names))

        class _Slice:
            def __init__(self, i):
                self.ptr = d_tex.ptr + 4 * texels * i
        tex = {nm: _Slice(i) for i, nm in enumerate(tex_names)}
        mn3, mx3 = np.array(mins, np.float32), np.array(maxs, np.float32)
        amn3, amx3 = np.array(arg_mn, np.float32), np.array(arg_mx, np.float32)
        for attempt in range(2):
            _lib.check(lib.gsx_sog_means_texels_dev(ctx.handle, d_pos.ptr, n, texels, mn3.ctypes.data, mx3.ctypes.data, amn3.ctypes.data,
                                                    amx3.ctypes.data, tex["means_l"].ptr,
                                                    tex["means_u"].ptr, d_list.ptr, cap, d_cnt.ptr), "gsx_sog_means_texels_dev")
            m_pos = int(d_cnt.download(np.uint32, 1)[0])
            if m_pos <= cap:
                break
            # a scene far from the origin on some axis: every log value is large against the axis' range, the +-5 ulp bracket of
            # numpy's log spans a sizeable part of a texel step and far more than the usual ~1 % of the texels are listed (x in
            # [10, 30]: ~18 %).  The kernel has counted them: once more with a list that holds them all -- numpy's log of the listed
            # values on the host is still orders of magnitude cheaper than the host-staged path
            cap = m_pos + 4096
            d_list = alloc(8 * cap, "list")
        else:
            raise NotEligible("too many position texels next to a rounding boundary")
        pos_list = d_list.download(np.uint32, 2 * m_pos).reshape(m_pos, 2) if m_pos else np.zeros((0, 2), np.uint32)
        _lib.check(lib.gsx_sog_quats_texels_dev(ctx.handle, d_rot.ptr, n, texels, tex["quats"].ptr), "gsx_sog_quats_texels_dev")
        st.mark("means_quats")

        # ---- scalar codebooks (:392-449): 50 000-sample -> sorted-run solver (the quality of the reference's sklearn path,
        # deterministic: DESIGN.md section 9) -> nearest-entry indices, written as texels
        drawer.join()
        if draw_err:
            raise draw_err[0]
        d_fit, d_idx, d_cb = alloc(4 * 50000), alloc(8 * 50000), alloc(4 * 256 * 2)
        books = {}
        for which, d_cols in (("scales", d_scale), ("sh0", d_dc)):
            status_print("Clustering Scales..." if which == "scales" else "Clustering Colors...")
            m = 3 * n
            fit_ptr, fit_n = d_cols.ptr, m
            if m > 50000:
                d_idx.upload(drawn[which])
                _lib.check(lib.gsx_gather_rows_dev(ctx.handle, d_cols.ptr, 1, d_idx.ptr, 50000, d_fit.ptr), "gsx_gather_rows_dev")
                fit_ptr, fit_n = d_fit.ptr, 50000
            cb_ptr = d_cb.ptr + (0 if which == "scales" else 4 * 256)
            _lib.check(lib.gsx_kmeans1d_dev(ctx.handle, fit_ptr, fit_n, 256, 50, 0, cb_ptr, None, None), "gsx_kmeans1d_dev")
            books[which] = cb_ptr
        cbs = d_cb.download(np.float32, 512)
        out["scale_codebook"], out["color_codebook"] = cbs[:256].copy(), cbs[256:].copy()
        _lib.check(lib.gsx_sog_codes_texels_dev(ctx.handle, d_scale.ptr, n, texels, books["scales"], 256, None, tex["scales"].ptr, None, 0, None),
                   "gsx_sog_codes_texels_dev")
        for attempt in range(2):
            _lib.check(lib.gsx_sog_codes_texels_dev(ctx.handle, d_dc.ptr, n, texels, books["sh0"], 256, d_op.ptr, tex["sh0"].ptr, d_list.ptr, cap,
                                                    d_cnt.ptr), "gsx_sog_codes_texels_dev")
            m_al = int(d_cnt.download(np.uint32, 1)[0])
            if m_al <= cap:
                break
            cap = m_al + 4096          # (opacities that are NaN or beyond +-80 by the million: listed all the same)
            d_list = alloc(8 * cap, "list")
        else:
            raise NotEligible("too many alpha texels next to a rounding boundary")
        al_list = d_list.download(np.uint32, 2 * m_al).reshape(m_al, 2) if m_al else np.zeros((0, 2), np.uint32)
        st.mark("codebooks_codes")

        # ---- SH palette (:496-552) as ONE batched Lloyd call on a worker thread (the C call releases the GIL): while its ~40 ms of
        # kernels run, this thread brings the five finished images back through a second context (its own stream, the staging
        # lanes' DMA engines) and evaluates numpy's log / exp for the listed texels
        host_tex = host_tex_all[:len(tex_names)]      # (one image fewer when the band detection found no SH)
        for i, nm in enumerate(tex_names):
            out["textures"][nm] = host_tex[i]
        worker, werr = None, []
        if coeffs:
            status_print(f"SOG Write Quality Level: {int(compression_level)} (0=Max, 9=Min)")
            status_print(f"SH Clustering: K={plan['target_k']}, Points={n}. Strategy: GPU (HIP gfx950)")
            nprob = len(bounds)
            off = np.array([b[0] for b in bounds] + [n], dtype=np.int64)
            init_rows = drawn["init"]
            d_init, d_cent, d_lab = alloc(8 * len(init_rows)), alloc(4 * nprob * k * coeffs), alloc(4 * n + 16)
            d_init.upload(init_rows)
            _lib.check(lib.gsx_gather_rows_dev(ctx.handle, d_sh.ptr, coeffs, d_init.ptr, len(init_rows), d_cent.ptr), "gsx_gather_rows_dev")
            _lib.check(lib.gsx_dev_memset(ctx.handle, d_lab.ptr, 0, 4 * n), "gsx_dev_memset")
            ctx.synchronize()          # the five images are complete in HBM

            def run_palette():
                try:
                    _lib.check(lib.gsx_kmeans_lloyd_batch_dev(ctx.handle, d_sh.ptr, off.ctypes.data, nprob, coeffs, k, int(max_iter), d_cent.ptr,
                                                              d_lab.ptr), "gsx_kmeans_lloyd_batch_dev")
                except BaseException as e:     # re-raised on the calling thread
                    werr.append(e)
            worker = threading.Thread(target=run_palette, name="gsx-sog-palette")
            worker.start()
        try:
            ctx2 = ar.side if worker else ctx
            _lib.check(lib.gsx_dev_download_staged(ctx2.handle, host_tex.ctypes.data, d_tex.ptr, 5 * 4 * texels), "gsx_dev_download_staged")
            with np.errstate(all="ignore"):
                if len(pos_list):
                    # one fancy assignment for the three channels: the tag (texel * 4 + channel) IS the flat index into the RGBA array
                    tag, ch = pos_list[:, 0].astype(np.int64), (pos_list[:, 0] & 3).astype(np.int64)
                    v = pos_list[:, 1].copy().view(np.float32)
                    mn_c, mx_c = np.array(mins, np.float32)[ch], np.array(maxs, np.float32)[ch]
                    t = (_log_transform(v) - mn_c) / (mx_c - mn_c)                                  # :293 per element, float32
                    u = np.clip(t * 65535, 0, 65535).astype(np.uint16)                              # :294-295
                    out["textures"]["means_l"].reshape(-1)[tag] = (u & 0xff).astype(np.uint8)
                    out["textures"]["means_u"].reshape(-1)[tag] = (u >> 8).astype(np.uint8)
                if len(al_list):
                    ti = (al_list[:, 0] >> 2).astype(np.int64)
                    o = al_list[:, 1].copy().view(np.float32)
                    out["textures"]["sh0"][ti, 3] = np.clip(1.0 / (1.0 + np.exp(-o)) * 255, 0, 255).astype(np.uint8)   # :457-459
        finally:
            if worker:
                worker.join()
        if werr:
            raise werr[0]
        st.mark("palette+download+host_patch" if coeffs else "download+host_patch")
        if coeffs:
            palette = nprob * k
            flat_n = palette * coeffs
            _lib.check(lib.gsx_sog_labels_texels_dev(ctx.handle, d_lab.ptr, n, texels, plan["chunk_size"], k, tex["shN_labels"].ptr),
                       "gsx_sog_labels_texels_dev")
            ctx.synchronize()              # the labels image is complete: it travels (second context, staging lanes) while the codebook is fitted
            lerr = []

            def fetch_labels():
                try:
                    _lib.check(lib.gsx_dev_download_staged(ar.side.handle, host_tex[5].ctypes.data, tex["shN_labels"].ptr, 4 * texels),
                               "gsx_dev_download_staged")
                except BaseException as e:
                    lerr.append(e)
            fetcher = threading.Thread(target=fetch_labels, name="gsx-sog-labels")
            fetcher.start()
            try:
                status_print("Clustering SH Centroids into Codebook...")
                d_cb2, d_cidx = alloc(4 * 256), alloc(flat_n)
                _lib.check(lib.gsx_kmeans1d_dev(ctx.handle, d_cent.ptr, flat_n, 256, 100, 0, d_cb2.ptr, None, None), "gsx_kmeans1d_dev")
                _lib.check(lib.gsx_quantize_sorted_codebook_dev(ctx.handle, d_cent.ptr, flat_n, d_cb2.ptr, 256, d_cidx.ptr), "gsx_quantize_sorted_codebook_dev")
                out["palette"] = palette
                out["shn_codebook"] = d_cb2.download(np.float32, 256).astype(np.float64)
                out["shn_centroid_index"] = d_cidx.download(np.uint8, flat_n)
            finally:
                fetcher.join()
            if lerr:
                raise lerr[0]
            st.mark("centroid_codebook+labels")
        if profile:
            out["_lists"] = (pos_list, al_list)
        out["stats"] = {"uncertain_positions": int(len(pos_list)), "uncertain_alpha": int(len(al_list)), "n": n}
        if profile:
            out["stage_ms"] = {k_: round(v_, 3) for k_, v_ in st.ms.items()}
        return out
    except _lib.GsxError:
        _lib.release_arenas()      # a failed HIP call: do not keep a context in an unknown state
        raise
