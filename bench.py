#!/usr/bin/env python
"""bench.py -- the point-cloud filtering hot path of 3dgsconverter on MI355X, every BASELINE.json config on one line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n SPLATS_PER_GPU] [--k 16]      (N > 1: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (ranks from a launcher)
    python bench.py --workload kmeans [--gpus N]   # BASELINE.json configs[4] alone (strong scaling over the chunks)

HEADLINE (the top-level fields; BASELINE.json's metric): one SOR step = one pass of the hot path over one batch of
synthetic splats whose xyz is already resident in HBM: grid binning -> exact KNN mean distance (knn_brick +
knn_ring_fast) -> numpy-exact mean/std/threshold -> survivor mask.  Workload at N=1: the configuration the target is
quoted on -- 10M uniform-random splats (L=5, seed 0: SURVEY.md 8(d) config 3's cloud), k=16, sigma=1.0; with N > 1 each
GPU adds one more 10M-splat index shard (weak scaling) and the step is the slab exchange of
3dgsconverter_amd/dist_slab.py (RCCL from the C library).

At N=1 the same run also times, and reports under "configs", every other BASELINE.json configuration and the clouds a
uniform grid is bad at -- each with its own roofline and CPU baseline:
    config1       1M splats SOR k=16 (configs[1])
    config2       10M splats: density filter (sensitivity 0.5) -> SOR k=16 on the survivors, rows resident in HBM across
                  both filters (configs[2]; the device chain `install()` gives the reference's orchestrator)
    config4       10M-splat SOG SH-palette K-Means (configs[4]: 64 chunks x (156 250 x 45), K=1024, 10 iterations)
    host_to_host  the 10M SOR call from a contiguous host xyz array to a host mask (gsx_sor_filter, PCIe included)
    clustered_1m / floaters_10m   Gaussian blobs of very different density / a scene + 0.5 % far floaters (adaptive mode:
                  the Morton-tree path)
No run of this file loads PyTorch: device memory comes from gsx_dev_malloc / gsx_dev_upload (include/gsx_hip.h), the clock is
time.perf_counter around gsx_ctx_synchronize.  With N > 1 every rank is a process of its own (3dgsconverter_amd/launch.py),
the communicator's unique id travels through a file, the barriers around the timed region and the MAX over the ranks' clocks
are gsx_comm_* calls like the data path's collectives.  Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EVENT_EVERY = 4              # steps of the timed region between two HIP-event pairs around the dominant kernel
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
N_SIMD = 256 * 4             # MI355X: 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9
# MI355X_MICROARCH.md "Per-instruction cycle constants": a wave64 VALU instruction issues over 2 cycles (f32 / int class,
# v_fma_f32 = 2 cyc); float64 arithmetic runs at half that rate (78.6 vs 157.3 TFLOP/s vector peak) = 4 cycles; the
# f64 transcendental seed (v_rsq_f64) at a quarter of the f64 rate = 16
CYC_F32, CYC_F64, CYC_TRANS_F64 = 2.0, 4.0, 16.0
BF16_MFMA_PEAK_TFLOPS = 2500.0
ONE_GPU_CONFIG3_MS = 16.67     # BASELINE configs[3] (50M splats, k=32) on ONE MI355X: ms per step, profiles/r05_bench_50m_k32.json


# --------------------------------------------------------------------------------------------------- synthetic inputs
# (bench.py owns its generators so that nothing outside the cpu_baseline legs touches oracle/; they restate
#  oracle/datasets.py, which the golden fixtures are built from)
def synth_uniform(n, extent, seed):
    """SURVEY.md 8(c) generator; shard r of the global cloud uses seed r."""
    return np.random.default_rng(seed).random((n, 3), dtype=np.float32) * np.float32(extent)


def synth_clustered(n, seed=0):
    """six Gaussian blobs (sigma 0.05 ... 1.5, peak densities 27 000 : 1) + 2 % far flyers"""
    rng = np.random.default_rng(seed)
    n_out = max(1, n // 50)
    n_in = n - n_out
    centers = rng.random((6, 3)) * 20.0 - 10.0
    sigmas = np.array([0.05, 0.2, 0.5, 1.0, 1.5, 0.1])
    which = rng.integers(0, 6, n_in)
    pts = centers[which] + rng.standard_normal((n_in, 3)) * sigmas[which, None]
    out = rng.random((n_out, 3)) * 80.0 - 40.0
    xyz = np.concatenate([pts, out]).astype(np.float32)
    return xyz[rng.permutation(n)]


def synth_scene_with_floaters(n, seed=0, far=500.0):
    """99.5 % of the splats in a 10^3 box, 0.5 % floaters spread over (2 far)^3: what SOR exists for"""
    rng = np.random.default_rng(seed)
    n_out = max(1, n // 200)
    pts = rng.random((n - n_out, 3)) * 10.0
    out = rng.random((n_out, 3)) * (2.0 * far) - far
    xyz = np.concatenate([pts, out]).astype(np.float32)
    return xyz[rng.permutation(n)]


def load_pmc(kernel, n, k):
    """Static PMC figures of the last committed rocprofv3 --pmc passes (counters cannot be read in-process)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            return json.load(f).get(kernel, {}).get("%d:%d" % (n, k))
    except Exception:
        return None


# --------------------------------------------------------------------------------------------------- rooflines
def sor_roofline(n, k, knn_ms, algo=0, n_total=None, single=True):
    """dominant kernel of a SOR step: knn_brick (or knn_brute with --algo 1)"""
    if algo != 1:
        # algorithmic bytes of ONE knn_brick launch: every brick streams its 4x4x4-cell neighbourhood
        # once (16 B/point, = 8x its own 2x2x2 cells on average), every query reads its own point (16 B) and writes one
        # f32: (8*16 + 16 + 4) B per splat
        bytes_per_splat = 8 * 16 + 16 + 4
        kernel = "knn_brick_kernel"
    else:
        bytes_per_splat = 16.0 * ((n_total or n) / 512.0) + 20      # 16 B per reference point per 512-query WG
        kernel = "knn_brute_kernel"
    alg_bytes = bytes_per_splat * n
    achieved = alg_bytes / (knn_ms * 1e-3) / 1e9 if knn_ms > 0 else 0.0
    pmc = load_pmc(kernel, n, k) if single else None
    # round 5: FETCH_SIZE calibrated on this code's own access patterns (tools/ubench/fetch_calib.hip, profiles/r05_fetch_calib.txt):
    # coalesced reads AND per-lane 16-byte gathers are counted at half their bytes (64 B per 128-byte request), writes exactly
    # (32-byte sectors) -> traffic = 2 x FETCH_SIZE + WRITE_SIZE, as MI355X_MICROARCH.md prescribes
    traffic = (2 * pmc["fetch_bytes"] + pmc["write_bytes"]) if pmc and "fetch_bytes" in pmc else None
    hbm = {"bound": "hbm", "kernel": kernel, "kernel_ms": round(knn_ms, 4), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "algorithmic_bytes": int(alg_bytes),
           "algorithmic_bytes_per_splat": bytes_per_splat, "traffic": traffic,
           "traffic_note": ("static: 2 x FETCH_SIZE + WRITE_SIZE per launch, rocprofv3 --pmc passes committed in profiles/pmc_latest.json "
                            "(fetch x2: the gfx950 correction of MI355X_MICROARCH.md, confirmed for this kernel's access patterns by "
                            "profiles/r05_fetch_calib.txt; writes are counted exactly, one 32-byte sector per isolated 4-byte store)") if traffic else None,
           "note": "BASELINE.json's metric: algorithmic HBM bytes of one launch / its HIP-event duration vs 8 TB/s.  The kernel "
                   "is bound by VALU issue and latency, not by HBM (see valu_issue)"}
    if pmc and pmc.get("valu_busy_frac"):
        # name the resource that actually binds next to BASELINE.json's HBM fraction
        hbm["binding_resource"] = {"name": "valu_issue", "busy_frac": pmc["valu_busy_frac"],
                                   "note": "the kernel is bound by VALU issue, not HBM: 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) "
                                           "of the committed rocprofv3 --pmc pass (static, %s); `frac` above stays BASELINE.json's metric" % pmc.get("source", "pmc_latest.json")}
    if pmc and pmc.get("valu_insts") and knn_ms > 0:
        # VALU issue cycles of one launch from the committed instruction counts (a property of build + cloud, labelled
        # static) weighted with the guide's per-instruction cycles, against SIMDs x clock x this run's kernel duration
        f64 = pmc.get("valu_f64_insts")
        trans = pmc.get("valu_trans_f64_insts", 0)
        if f64 is not None:
            cycles = CYC_F64 * f64 + CYC_TRANS_F64 * trans + CYC_F32 * (pmc["valu_insts"] - f64 - trans)
            mix = ("%d float64-rate (SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 as counted + v_min/max_f64 and v_cvt_f64_f32 derived from them, "
                   "profiles/pmc_latest.json) + %d f64 transcendental + %d other wave-instructions" % (f64, trans, pmc["valu_insts"] - f64 - trans))
        else:
            cycles = CYC_F32 * pmc["valu_insts"]
            mix = "no f32/f64 split committed for this build: every instruction priced at the f32 rate (a lower bound)"
        cap = N_SIMD * CLOCK_HZ * knn_ms * 1e-3
        hbm["valu_issue"] = {"frac": round(cycles / cap, 4), "issue_cycles": int(cycles), "capacity_cycles": int(cap),
                             "insts_per_launch": pmc["valu_insts"], "mix": mix, "busy_frac_pmc": pmc.get("valu_busy_frac"),
                             "peak_note": "MI355X_MICROARCH.md cycle constants: f32/int wave64 VALU 2 cyc, f64 4 cyc, v_rsq_f64 16 cyc; "
                                          "1024 SIMDs x 2.4 GHz nominal",
                             "kind": "static instruction counts: rocprofv3 --pmc pass of this build (%s); live duration" % pmc.get("source", "pmc_latest.json")}
    return hbm


def hbm_roofline(kernel, alg_bytes, ms, note):
    ach = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"bound": "hbm", "kernel": kernel, "kernel_ms": round(ms, 4), "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 5), "algorithmic_bytes": int(alg_bytes), "traffic": None, "note": note}


# --------------------------------------------------------------------------------------------------- single GPU, no torch
class SorBench:
    """one cloud resident in HBM + the buffers of a SOR step (gsx_dev_malloc; nothing is allocated inside a step)"""

    def __init__(self, L, ctx, xyz_host, k, sigma, algo=0):
        self.L, self.ctx, self.k, self.sigma, self.algo = L, ctx, int(k), float(sigma), int(algo)
        self.n = len(xyz_host)
        self.xyz = ctx.alloc(max(xyz_host.nbytes, 16)).upload(xyz_host)
        self.md = ctx.alloc(4 * self.n + 16)
        self.stats = ctx.alloc(16)
        self.mask = ctx.alloc(self.n + 16)

    def step(self):
        p, c = self.xyz.ptr, self.ctx
        c.sor_knn(p, p + 4, p + 8, 3, self.n, 0, self.n, self.k, self.md.ptr, algo=self.algo)
        c.sor_stats(self.md.ptr, self.n, self.sigma, self.stats.ptr)
        c.sor_mask(self.md.ptr, self.n, self.stats.ptr + 8, self.mask.ptr)

    def info(self):
        p = self.xyz.ptr
        return self.ctx.sor_knn(p, p + 4, p + 8, 3, self.n, 0, self.n, self.k, self.md.ptr, algo=self.algo, want_info=True)

    def results(self):
        self.ctx.check()
        mask = self.mask.download(np.uint8, self.n).view(np.bool_)
        return mask, self.stats.download(np.float32, 3)

    def free(self):
        for b in (self.xyz, self.md, self.stats, self.mask):
            b.free()


def time_steps(ctx, step, steps, warmup, event_slot=None, L=None):
    """W untimed steps, then K steps between two synchronisations; HIP events around `event_slot`'s kernels on every
    EVENT_EVERY-th timed step (an event pair leaves a ~10 us bubble on either side of the kernel it brackets)"""
    for _ in range(warmup):
        step()
    ctx.synchronize()
    if event_slot is not None:
        ctx.set_param("timing_mask", 1 << event_slot)
        ctx.set_timing(True)
        ctx.reset_timing()
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if event_slot is not None:
            ctx.set_timing(i % EVENT_EVERY == 0)
        step()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    out = {"dt": dt, "ms_per_step": dt / steps * 1e3}
    if event_slot is not None:
        ctx.set_timing(True)
        cnt, ms = ctx.timing(event_slot)
        out["event_ms"] = ms / max(cnt, 1)
        ctx.set_timing(False)
    return out


def kernel_groups(ctx, L, step, slots, steps):
    """a short separate pass with every slot recording (not in a timed region)"""
    ctx.set_param("timing_mask", 0xff)
    ctx.set_timing(True)
    ctx.reset_timing()
    for _ in range(steps):
        step()
    ctx.synchronize()
    out = {name: round(ctx.timing(slot)[1] / steps, 4) for name, slot in slots.items()}
    ctx.set_timing(False)
    return out


class ClockSampler:
    """the GPU's shader clock while a timed loop runs: the active level of pp_dpm_sclk (amdgpu sysfs; readable by an ordinary
    user) every 50 ms from a thread, `rocm-smi --showclocks` once as the fall-back.  Median / min / max in MHz, or None."""

    def __init__(self):
        import glob
        self.paths = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        self.samples, self._stop, self._th = [], False, None

    def _read(self):
        best = None
        for p in self.paths:
            try:
                with open(p) as f:
                    for ln in f:
                        if "*" in ln:
                            mhz = int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
                            best = mhz if best is None else max(best, mhz)   # several cards visible: the busy one runs fastest
            except (OSError, ValueError, IndexError):
                pass
        return best

    def __enter__(self):
        import threading

        def loop():
            while not self._stop:
                v = self._read()
                if v:
                    self.samples.append(v)
                time.sleep(0.05)
        if self.paths:
            self._th = threading.Thread(target=loop, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._th:
            self._th.join()

    def summary(self):
        if self.samples:
            v = sorted(self.samples)
            return {"sclk_mhz_median": v[len(v) // 2], "sclk_mhz_min": v[0], "sclk_mhz_max": v[-1], "samples": len(v), "source": "pp_dpm_sclk"}
        try:
            import re
            import subprocess
            txt = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
            m = [int(x) for x in re.findall(r"sclk clock level:.*?\((\d+)Mhz\)", txt)]
            if m:
                return {"sclk_mhz_median": max(m), "samples": 1, "source": "rocm-smi --showclocks, once after the loop"}
        except Exception:   # noqa: BLE001
            pass
        return None


def run_sustained(L, ctx, xyz_host, k, sigma, algo, min_seconds=3.2, window=100):
    """the headline step back to back for >= 3 s (>= 1500 steps at ~2.1 ms) -- long enough for clocks and
    thermals to settle and for a 5 s utilisation sampler to land inside it -- in windows of 100 steps (one synchronisation per
    window: ~10 us per 0.2 s)."""
    b = SorBench(L, ctx, xyz_host, k, sigma, algo)
    try:
        for _ in range(5):
            b.step()
        ctx.synchronize()
        wins = []
        with ClockSampler() as clk:
            t_all = time.perf_counter()
            while (time.perf_counter() - t_all) < min_seconds or len(wins) < 15:
                t0 = time.perf_counter()
                for _ in range(window):
                    b.step()
                ctx.synchronize()
                wins.append((time.perf_counter() - t0) / window * 1e3)
            total = time.perf_counter() - t_all
        mask, stats = b.results()
        n = b.n
        steps = len(wins) * window
        return {"workload": "the headline step (%d splats, k=%d) back to back, rows resident in HBM" % (n, k), "steps": steps,
                "seconds": round(total, 3), "ms_per_step": round(total / steps * 1e3, 4), "value": round(n * steps / total / 1e6, 2),
                "unit": "Msplats/s", "window_steps": window, "window_ms_min": round(min(wins), 4), "window_ms_max": round(max(wins), 4),
                "window_ms_first": round(wins[0], 4), "window_ms_last": round(wins[-1], 4), "clock": clk.summary(),
                "survivors": int(mask.sum()), "threshold": float(stats[2])}
    finally:
        b.free()


def cpu_sor(xyz_host, k, sigma, gpu_mask):
    """cpu_baseline leg: the reference's CPU path (cKDTree + numpy, restated in oracle/sor.py because the reference never
    returns its mask) on the SAME cloud, host cores"""
    from oracle import sor as osor
    workers = max(1, (os.cpu_count() or 2) - 1)
    t0 = time.perf_counter()
    ref = osor.sor(xyz_host, k, sigma, workers=workers)
    cpu_dt = time.perf_counter() - t0
    n = len(xyz_host)
    return {"value": round(n / cpu_dt / 1e6, 4), "unit": "Msplats/s", "cores": workers, "kind": "port",
            "sample": "the full %d-splat workload, once (%.2f s): scipy cKDTree query workers=%d + numpy stats" % (n, cpu_dt, workers),
            "mask_identical_to_gpu": bool(np.array_equal(ref["mask"], gpu_mask))}


def run_sor(L, ctx, xyz_host, k, sigma, steps, warmup, algo=0, groups=False, cpu=False, adaptive=False):
    ctx.set_param("adaptive", 1 if adaptive else 0)
    b = SorBench(L, ctx, xyz_host, k, sigma, algo)
    try:
        t = time_steps(ctx, b.step, steps, warmup, event_slot=L.T_SOR_KNN, L=L)
        mask, stats = b.results()
        n = b.n
        out = {"value": round(n * steps / t["dt"] / 1e6, 2), "unit": "Msplats/s", "ms_per_step": round(t["ms_per_step"], 4),
               "steps": steps, "warmup": warmup, "knn_kernel_ms": round(t["event_ms"], 4), "survivors": int(mask.sum()),
               "threshold": float(stats[2])}
        if groups:
            side = min(steps, 10)
            g = kernel_groups(ctx, L, b.step, {"bin": L.T_SOR_BIN, "fallback": L.T_SOR_FALLBACK, "stats": L.T_SOR_STATS}, side)
            out["kernel_ms_per_step"] = dict({"knn": round(t["event_ms"], 4)}, **g,
                                             note="knn: HIP events inside the timed region (every %d-th step); the others: a "
                                                  "separate pass of %d steps" % (EVENT_EVERY, side))
        out["grid"] = b.info()
        ctx.synchronize()
        if cpu:
            out["cpu_baseline"] = cpu_sor(xyz_host, k, sigma, mask)
        return out
    finally:
        b.free()
        ctx.set_param("adaptive", 0)


def run_host_to_host(L, ctx, xyz_host, k, sigma, resident_ms, reps=3):
    """SURVEY.md 8(d) "Metric": N / wall time of the filter call from contiguous host xyz to host mask (gsx_sor_filter:
    upload over PCIe, the device pipeline, mask download).  Its own roofline is PCIe: the (n,3) rows up and the mask down
    against the copy rate measured in this run on the same buffers; the exact KNN cannot start before the last row has
    arrived (the grid comes from the global bounding box), so copy time and pipeline time add up."""
    L.sor_filter(xyz_host, k, sigma, want_mean=False)   # warm: workspace allocation, first-touch of the pinned staging
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        res = L.sor_filter(xyz_host, k, sigma, want_mean=False)
        ts.append(time.perf_counter() - t0)
    n = len(xyz_host)
    best = min(ts)
    # the copies alone, same buffers (pageable numpy memory, as a caller of remove_flyers has it)
    dev, dmask = ctx.alloc(xyz_host.nbytes), ctx.alloc(n + 16)
    dev.upload(xyz_host)
    up, down = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        dev.upload(xyz_host)
        up.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        dmask.download(np.uint8, n)
        down.append(time.perf_counter() - t0)
    dev.free()
    dmask.free()
    up_ms, down_ms = min(up) * 1e3, min(down) * 1e3
    pcie_bytes = 12 * n + n
    return {"workload": "gsx_sor_filter: %d splats, host (n,3) f32 array -> host mask, k=%d (PCIe both ways included)" % (n, k),
            "ms_per_call": round(best * 1e3, 3), "ms_per_call_all": [round(t * 1e3, 3) for t in ts], "value": round(n / best / 1e6, 2),
            "unit": "Msplats/s", "pcie_bytes": pcie_bytes, "survivors": int(res["mask"].sum()),
            "roofline": {"bound": "pcie", "achieved": round(pcie_bytes / best / 1e9, 2), "unit": "GB/s",
                         "peak": round(pcie_bytes / ((up_ms + down_ms) * 1e-3) / 1e9, 2),
                         "frac": round((up_ms + down_ms) / (best * 1e3), 4),
                         "h2d_ms": round(up_ms, 3), "h2d_GBs": round(12 * n / (up_ms * 1e-3) / 1e9, 2), "d2h_ms": round(down_ms, 3),
                         "device_pipeline_ms": round(resident_ms, 4),
                         "floor_ms": round(up_ms + down_ms + resident_ms, 3),
                         "note": "peak = the same bytes at the copy rate measured in this run (gsx_dev_upload / gsx_dev_download of the "
                                 "same pageable buffers); frac = copy time / call time; floor = copies + the resident pipeline, which "
                                 "cannot overlap them: the exact KNN's grid needs the global bounding box, i.e. every row"}}


def run_chain(L, ctx_unused, gsx, xyz_host, sensitivity, k, sigma, steps, warmup, cpu=False):
    """BASELINE.json configs[2]: density filter -> SOR on rows that stay in HBM (what ChainedDataProcessor does between
    `DataProcessor(data)` and `.data`; converter.py:205-236 order).  One step = restore the pristine rows (device copy),
    the density filter as one device call (voxel occupancy, clusters of the <= 181 dense voxels, keep rule, membership mask)
    + compaction, SOR + compaction.  The box of the uploaded rows is computed once per chain (DeviceChain keeps it)."""
    dp = importlib.import_module("3dgsconverter_amd.processing.data_processor")
    clusters = importlib.import_module("3dgsconverter_amd.processing.clusters")
    voxel, thr = dp.density_params_from_sensitivity(sensitivity)
    ch = L.DeviceChain(xyz_host, keep_pristine=True)
    n = len(xyz_host)
    last = {}

    def step():
        ch.restart()
        min_points = int(ch.n * (thr / 100.0))
        res = ch.density_filter(float(voxel), min_points, False)     # ONE device call: gsx_density_filter_dev + compaction
        if res["status"] == L.DENSITY_HOST:   # (a tie for the largest cluster: the drop-in's host steps; not this cloud)
            occ = ch.density_voxels(float(voxel), min_points)
            comps = clusters.connected_clusters(map(tuple, occ["dense_keys"].tolist()))
            kept, _, _ = clusters.select_clusters(comps, False)
            res = {"left": ch.density_keep(float(voxel), np.array(sorted(kept), dtype=np.int64).reshape(-1, 3))}
        last["after_density"] = res["left"]
        last["sor"] = ch.sor_keep(k, sigma)

    try:
        c = ch.ctx
        t = time_steps(c, step, steps, warmup)
        g = kernel_groups(c, L, step, {"density": L.T_DENSITY, "knn": L.T_SOR_KNN, "bin": L.T_SOR_BIN, "fallback": L.T_SOR_FALLBACK,
                                       "stats": L.T_SOR_STATS}, min(steps, 5))
        survivors = ch.survivors()
        out = {"workload": "BASELINE.json configs[2]: %d uniform-random splats (L=5, seed 0), density sensitivity %.1f (voxel %.2f, "
                           "%.2f %%) then SOR k=%d sigma=%g on the survivors, rows resident in HBM across both filters" % (n, sensitivity, voxel, thr, k, sigma),
               "value": round(n * steps / t["dt"] / 1e6, 2), "unit": "Msplats/s", "ms_per_step": round(t["ms_per_step"], 4), "steps": steps,
               "after_density": int(last["after_density"]), "survivors": int(len(survivors)), "sor_threshold": float(last["sor"]["threshold"]),
               "kernel_ms_per_step": g,
               # density kernels: two passes over the rows (12 B each) + 1 B mask = 25 B/splat
               "roofline": hbm_roofline("voxel_count + voxel_cluster + voxel_mask (density.hip: gsx_density_filter_dev)", 25.0 * n, g["density"],
                                        "density stage: 25 algorithmic B/splat over the HIP-event time of its kernels; the SOR stage of "
                                        "this chain has the headline's roofline"),
               "note": "one host synchronisation per filter (the survivor count of its compaction) and two device compactions; the "
                       "248-byte host table is compacted once afterwards, outside this step"}
        if cpu:
            # cpu_baseline leg: the reference's apply_density_filter (pure numpy/Python, restated in oracle/density.py) on a
            # bounded sample
            from oracle import density as oden
            m = min(n, 2_000_000)
            t0 = time.perf_counter()
            oden.density_filter(xyz_host[:m], float(voxel), float(thr))
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": round(m / dt / 1e6, 4), "unit": "Msplats/s", "cores": 1, "kind": "port",
                                   "sample": "density stage only, the first %d splats of the cloud, once (%.2f s): np.unique(axis=0) + "
                                             "Python BFS as the reference does; the SOR stage's CPU baseline is the headline's" % (m, dt)}
        return out
    finally:
        ch.close()


def run_kmeans(L, ctx, gsx, n_scene, steps, warmup, cpu=False, params=(), lanes=0, level=2):
    """BASELINE.json configs[4] with SURVEY.md 8(d) config-5 data: f_rest ~ N(0, 0.1^2) f32 from numpy's seed-0 generator,
    initial centroids = random rows drawn like the reference's front door (np.random.seed(0); one np.random.choice per chunk,
    gpu_ops.py:182).  One step = all chunks of the scene, SH rows resident in HBM."""
    pal = importlib.import_module("3dgsconverter_amd.dist_palette")
    d, iters = 45, 10
    plan = pal.palette_plan(n_scene, level)
    nch, cs, k = plan["num_chunks"], plan["chunk_size"], plan["k_per_chunk"]
    rng = np.random.default_rng(0)
    np.random.seed(0)
    # the chunks are independent problems (sog.py:536-552): they are dealt out to a few contexts with streams of their own,
    # exactly as the product's palette path does (_lib.kmeans_lloyd_many), so that one chunk's small kernels fill the tails
    # of another's
    # round 5: the product's palette path (_lib.kmeans_lloyd_many -> kmeans_lloyd_batch) runs the chunks as ONE batched call,
    # the chunk being a grid dimension of every kernel; `--lanes N` (N > 0) times the former scheme instead: the chunks dealt
    # out to N contexts with streams of their own
    batched = not lanes
    nlanes = 1 if batched else max(1, min(lanes, nch))
    lane_ctx = [L.Context(0, own_stream=True) for _ in range(nlanes)]
    rows_of = [min(cs, n_scene - i * cs) for i in range(nch)]
    off = np.zeros(nch + 1, dtype=np.int64)
    np.cumsum(rows_of, out=off[1:])
    n_total = int(off[-1])
    data_all = ctx.alloc(4 * n_total * d)          # all chunks' rows, concatenated (what the writer's SH table is)
    init_all = ctx.alloc(4 * nch * k * d)
    cent_all = ctx.alloc(4 * nch * k * d)
    label_all = ctx.alloc(4 * n_total + 16)
    first_host = None

    class _View:   # a chunk's slice of the big buffers
        def __init__(self, base, ptr):
            self.base, self.ptr = base, ptr

        def download(self, dtype, count):
            return self.base.download_at(self.ptr - self.base.ptr, dtype, count)

    chunks, inits, cents, labels = [], [], [], []
    for i in range(nch):
        rows = rows_of[i]
        x = rng.standard_normal((rows, d), dtype=np.float32) * np.float32(0.1)
        init = np.ascontiguousarray(x[np.random.choice(rows, k, replace=False)])
        if i == 0:
            first_host = x
        L.check(ctx.lib.gsx_dev_upload_async(ctx.handle, data_all.ptr + 4 * d * int(off[i]), x.ctypes.data, x.nbytes), "gsx_dev_upload_async")
        L.check(ctx.lib.gsx_dev_upload_async(ctx.handle, init_all.ptr + 4 * k * d * i, init.ctypes.data, init.nbytes), "gsx_dev_upload_async")
        ctx.synchronize()                          # (x / init are released at the end of the iteration)
        chunks.append((_View(data_all, data_all.ptr + 4 * d * int(off[i])), rows))
        inits.append(_View(init_all, init_all.ptr + 4 * k * d * i))
        cents.append(_View(cent_all, cent_all.ptr + 4 * k * d * i))
        labels.append(_View(label_all, label_all.ptr + 4 * int(off[i])))
    for c in lane_ctx:
        for name, val in params:
            c.set_param(name, val)

    def step():
        if batched:
            c = lane_ctx[0]
            L.check(c.lib.gsx_dev_copy(c.handle, cent_all.ptr, init_all.ptr, 4 * nch * k * d), "gsx_dev_copy")
            L.check(c.lib.gsx_kmeans_lloyd_batch_dev(c.handle, data_all.ptr, off.ctypes.data, nch, d, k, iters, cent_all.ptr, label_all.ptr),
                    "gsx_kmeans_lloyd_batch_dev")
            return
        for j in range(nch):
            c = lane_ctx[j % nlanes]
            L.check(c.lib.gsx_dev_copy(c.handle, cents[j].ptr, inits[j].ptr, 4 * k * d), "gsx_dev_copy")
            L.check(c.lib.gsx_kmeans_lloyd_dev(c.handle, chunks[j][0].ptr, chunks[j][1], d, k, iters, cents[j].ptr, labels[j].ptr),
                    "gsx_kmeans_lloyd_dev")

    class _AllLanes:   # what time_steps synchronises: every lane
        def synchronize(self):
            for c in lane_ctx:
                c.synchronize()

    try:
        ctx.synchronize()
        t = time_steps(_AllLanes(), step, steps, warmup)
        # kernel groups: one extra pass on ONE lane with events (concurrent lanes would overlap each other's intervals)
        side = 1
        c0 = lane_ctx[0]
        c0.set_param("timing_mask", (1 << L.T_KMEANS_ASSIGN) | (1 << L.T_KMEANS_UPDATE))
        c0.set_timing(True)
        c0.reset_timing()
        if batched:
            step()
        else:
            for j in range(nch):
                L.check(c0.lib.gsx_dev_copy(c0.handle, cents[j].ptr, inits[j].ptr, 4 * k * d), "gsx_dev_copy")
                L.check(c0.lib.gsx_kmeans_lloyd_dev(c0.handle, chunks[j][0].ptr, chunks[j][1], d, k, iters, cents[j].ptr, labels[j].ptr),
                        "gsx_kmeans_lloyd_dev")
        c0.synchronize()
        n_as, ms_as = c0.timing(L.T_KMEANS_ASSIGN)
        n_up, ms_up = c0.timing(L.T_KMEANS_UPDATE)
        c0.set_timing(False)
        # one interval = operand prep + matrix-core assign + exact list of ONE iteration: of one chunk (lanes) or of all chunks
        # (batched) -- the roofline below is per chunk iteration either way
        assign_ms = ms_as / max(n_as, 1) / (nch if batched else 1)
        rows0 = chunks[0][1]
        ktiles, ns = (k + 31) // 32, 3              # 32-centroid tiles, three 16-wide slices of the 45 (+3) dimensions
        # three v_mfma_f32_32x32x16_bf16 per slice (xh.ch + xh.cl + xl.ch), 2*32*32*16 flops each, per (32 points x 32 centroids)
        flops = -(-rows0 // 32) * ktiles * ns * 3 * 2 * 32 * 32 * 16
        achieved = flops / (assign_ms * 1e-3) / 1e12 if assign_ms > 0 else 0.0
        alg_bytes = rows0 * (4 * d + 4)             # read the rows once, write one label
        out = {"workload": "BASELINE.json configs[4]: %d splats, degree-3 SH rows (45 f32 ~ N(0, 0.1^2), numpy seed 0), "
                           "compression_level %d -> %d chunks of %d rows, K=%d per chunk, %d Lloyd iterations, rows resident in HBM"
                           % (n_scene, level, nch, cs, k, iters),
               "value": round(n_scene * steps / t["dt"] / 1e6, 2), "unit": "Msplats/s", "ms_per_step": round(t["ms_per_step"], 3), "steps": steps,
               "lanes": nlanes, "batched": batched,
               "kernel_ms_per_step": {"assign (operands + mfma + exact list)": round(ms_as / side, 3),
                                      "update (label sort + segmented reduce)": round(ms_up / side, 3),
                                      "note": ("HIP-event sums of a separate pass: %d iterations, each ONE set of launches for all %d chunks "
                                               "(gsx_kmeans_lloyd_batch_dev: the chunk is a grid dimension)" % (iters, nch)) if batched else
                                              ("HIP-event sums of a separate pass on ONE lane (%d launches back to back); the timed region "
                                               "runs the chunks on %d concurrent lanes" % (nch * iters, nlanes))},
               "roofline": {"bound": "mfma", "kernel": "kmeans_assign_mfma_cs_kernel<45> (+ operand prep and exact list kernel in the same interval)",
                            # `achieved` / `frac` are the USEFUL work (2 N K D of the f32 problem); the matrix-core flops the
                            # filter actually issues are reported next to it (issued_*)
                            "achieved": round(2 * rows0 * k * d / (assign_ms * 1e-3) / 1e12, 1) if assign_ms > 0 else None,
                            "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(2 * rows0 * k * d / (assign_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4) if assign_ms > 0 else None,
                            "traffic": None, "kernel_ms": round(assign_ms, 4),
                            "useful_flops_per_launch": 2 * rows0 * k * d, "algorithmic_bytes": alg_bytes,
                            "issued_flops_per_launch": flops, "issued_tflops": round(achieved, 1),
                            "issued_frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4),
                            "useful_vs_fp32_vector_peak": round(2 * rows0 * k * d / (assign_ms * 1e-3) / 1e12 / 157.3, 3) if assign_ms > 0 else None,
                            "hbm_frac_of_8TBs": round(alg_bytes / (assign_ms * 1e-3) / 8e12, 5) if assign_ms > 0 else None,
                            "note": "frac: 2 N K D of the f32 problem over the assign interval against the dense bf16 matrix peak; issued_frac: the "
                                    "bf16 matrix-core flops actually ISSUED (a FILTER: 9 MFMAs per 32x32 tile incl. the two cross terms "
                                    "of the bf16 split, 45 -> 48 padded dimensions); useful_vs_fp32_vector_peak: against the 157.3 TFLOP/s f32 VALU peak, the roofline "
                                    "SURVEY.md 8(d) names for this arithmetic.  Labels are certified exact, the uncertified ~0.3 % rescanned in f32"}}
        if cpu:
            # cpu_baseline leg: the reference's CPU path for the same call (gpu_ops.py:48-52: MiniBatchKMeans, batch 16384,
            # n_init auto), ONE chunk
            from sklearn.cluster import MiniBatchKMeans
            from oracle import kmeans as okm
            t0 = time.perf_counter()
            km = MiniBatchKMeans(n_clusters=k, max_iter=iters, batch_size=min(4096 * 4, rows0), n_init="auto", compute_labels=True)
            km.fit(first_host)
            cpu_dt = time.perf_counter() - t0
            c0 = cents[0].download(np.float32, k * d).reshape(k, d)
            l0 = labels[0].download(np.int32, rows0)
            out["cpu_baseline"] = {"value": round(rows0 / cpu_dt / 1e6, 4), "unit": "Msplats/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": "one of the %d chunks (%d x %d, K=%d, max_iter=%d), once (%.2f s): sklearn MiniBatchKMeans "
                                             "called as the reference's _kmeans_sklearn does" % (nch, rows0, d, k, iters, cpu_dt),
                                   "inertia_cpu": round(okm.inertia(first_host, km.cluster_centers_, km.labels_), 2),
                                   "inertia_gpu_same_chunk": round(okm.inertia(first_host, c0, l0), 2)}
        return out
    finally:
        for c in lane_ctx:
            c.synchronize()
        for b in (data_all, init_all, cent_all, label_all):
            b.free()
        for c in lane_ctx:
            c.close()


def main_single(args):
    gsx = importlib.import_module("3dgsconverter_amd")
    L = gsx._lib
    ctx = L.Context(0)
    for kv in args.param:
        name, val = kv.split("=")
        ctx.set_param(name, float(val))
    want_cpu = not args.no_cpu_baseline
    small = max(3, min(args.steps, 5))
    t_start = time.perf_counter()

    if args.exchange == "slab":
        return main_slab_one_rank(args, gsx, L, ctx)

    xyz = synth_uniform(args.n, args.extent, 0)
    sustained = None
    if not args.no_secondary:   # FIRST: a utilisation sampler that looks at the first seconds of this process sees the GPU busy
        try:
            sustained = run_sustained(L, ctx, xyz, args.k, args.sigma, args.algo)
        except Exception as e:   # noqa: BLE001
            sustained = {"error": repr(e)}
    head = run_sor(L, ctx, xyz, args.k, args.sigma, args.steps, args.warmup, algo=args.algo, groups=True, cpu=want_cpu)
    roof = sor_roofline(args.n, args.k, head["knn_kernel_ms"], args.algo)
    out = {
        "metric": "Msplats/sec SOR k=%d" % args.k, "value": head["value"], "unit": "Msplats/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 select / f64 finalize", "data": "synthetic",
        "config": {"workload": "%d uniform-random splats per GPU (L=%g, seed=rank), SOR k=%d sigma=%g, exact KNN, xyz resident in HBM"
                               % (args.n, args.extent, args.k, args.sigma),
                   "splats_per_gpu": args.n, "k": args.k, "sigma": args.sigma,
                   "algo": "grid-binned exact KNN" if args.algo != 1 else "LDS-tiled brute force", "parallelism": "single GPU",
                   "host_runtime": "numpy + ctypes on libgsx_hip.so (gsx_dev_malloc / gsx_dev_upload); torch is not imported"},
        "roofline": roof, "kernel_ms_per_step": head["kernel_ms_per_step"], "survivors_rank0": head["survivors"],
        "threshold": head["threshold"], "grid": head["grid"]}
    if "cpu_baseline" in head:
        out["cpu_baseline"] = head["cpu_baseline"]

    if not args.no_secondary:
        configs = {}

        def attempt(name, fn):
            try:
                configs[name] = fn()
            except Exception as e:   # noqa: BLE001 -- the headline line must survive a failure in a secondary configuration
                configs[name] = {"error": repr(e)}

        def config1():
            x1 = synth_uniform(1_000_000, 10.0, 0)
            r = run_sor(L, ctx, x1, args.k, args.sigma, max(args.steps, 50), max(args.warmup, 5), cpu=want_cpu)
            r["workload"] = "BASELINE.json configs[1]: 1000000 uniform-random splats (L=10, seed 0), SOR k=%d sigma=%g" % (args.k, args.sigma)
            r["roofline"] = sor_roofline(1_000_000, args.k, r["knn_kernel_ms"])
            return r

        def clustered():
            xc = synth_clustered(1_000_000, 0)
            r = run_sor(L, ctx, xc, args.k, args.sigma, small, 2, cpu=want_cpu, adaptive=True)
            r["workload"] = ("1000000 splats in six Gaussian blobs (sigma 0.05...1.5) + 2 %% far flyers, SOR k=%d, adaptive mode "
                             "(routed to the Morton-tree path, csrc/sor_tree.hip)" % args.k)
            r["roofline"] = sor_roofline(1_000_000, args.k, r["knn_kernel_ms"], single=False)
            r["roofline"]["kernel"] = "knn_leaf_kernel<17>"
            r["roofline"]["note"] = "knn_leaf (one wave per Morton leaf), priced like knn_brick; " + r["roofline"]["note"]
            return r

        def floaters():
            xf = synth_scene_with_floaters(args.n, 0)
            r = run_sor(L, ctx, xf, args.k, args.sigma, small, 2, cpu=want_cpu, adaptive=True)
            r["workload"] = ("%d splats: a 10^3 scene + 0.5 %% floaters in a 1000^3 box, SOR k=%d, adaptive mode (routed to the "
                             "Morton-tree path, csrc/sor_tree.hip)" % (args.n, args.k))
            r["roofline"] = sor_roofline(args.n, args.k, r["knn_kernel_ms"], single=False)
            r["roofline"]["kernel"] = "knn_leaf_kernel<17>"
            r["roofline"]["note"] = "knn_leaf (one wave per Morton leaf), priced like knn_brick; " + r["roofline"]["note"]
            return r

        def config1_brute():
            # BASELINE.json configs[1] AS WORDED: "1M splats SOR k=16, 1xMI355X (LDS-tiled brute-force KNN)" -- csrc/sor_brute.hip,
            # --algo 1.  Its roofline is FP32 VALU issue: >= 7 slots per (query, reference) pair, 78.6 T slots/s
            x1 = synth_uniform(1_000_000, 10.0, 0)
            r = run_sor(L, ctx, x1, args.k, args.sigma, 3, 1, algo=1)
            pairs = 1e6 * 1e6
            rate = pairs / (r["knn_kernel_ms"] * 1e-3) if r["knn_kernel_ms"] > 0 else 0.0
            r["workload"] = "BASELINE.json configs[1] as worded: 1000000 uniform-random splats (L=10, seed 0), SOR k=%d, LDS-tiled BRUTE-FORCE exact KNN (--algo 1)" % args.k
            r["roofline"] = {"bound": "valu", "kernel": "knn_brute_kernel", "kernel_ms": r["knn_kernel_ms"], "achieved": round(rate / 1e12, 3),
                             "peak": 11.2, "unit": "Tpair/s", "frac": round(rate / 11.2e12, 4), "pairs_per_launch": int(pairs), "traffic": None,
                             "hbm": sor_roofline(1_000_000, args.k, r["knn_kernel_ms"], algo=1, single=False),
                             "note": "N^2 = 1e12 (query, reference) pairs per launch over the HIP-event kernel time against SURVEY.md 8(d)'s "
                                     "ceiling of 11.2 T pairs/s (7 FP32 VALU slots per pair: 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz / 7); its "
                                     "streaming bytes (16 B per reference point per 512-query workgroup) are under `hbm` -- brute force "
                                     "cannot be HBM-bound.  The grid kernel (configs.config1) answers the same 1M cloud ~500x faster"}
            return r

        def blobs_k25():
            # what `gsconverter --sor_k 25 --sor_sigma 10.5` (the CLI's defaults, converter.py:228-234) runs on a cloud that is
            # NOT uniform: six Gaussian blobs of very different density + far flyers -> the Morton-tree path
            xb = synth_clustered(args.n, 0)
            r = run_sor(L, ctx, xb, 25, 10.5, small, 2, cpu=want_cpu, adaptive=True)
            r["workload"] = ("%d splats in six Gaussian blobs (sigma 0.05...1.5) + 2 %% far flyers, SOR k=25 sigma=10.5 (the reference CLI's "
                             "defaults), adaptive mode (Morton-tree path, csrc/sor_tree.hip)" % args.n)
            r["roofline"] = sor_roofline(args.n, 25, r["knn_kernel_ms"], single=False)
            r["roofline"]["kernel"] = "knn_leaf_kernel<29> + the rim kernels (knn_tree_near, knn_tree_query) in the same interval"
            r["roofline"]["algorithmic_bytes_per_splat"] = 8 * 16 + 16 + 4
            r["roofline"]["note"] = "knn_leaf (one wave per Morton leaf), priced like knn_brick; " + r["roofline"]["note"]
            return r

        def dropin_e2e():
            # what a `gsconverter` user sees (converter.py:150-259): the 62 x f4 = 248-byte rows of a real 3DGS table on the host,
            # DataProcessor (the lazy class install() binds) -> density filter -> SOR -> `.data` (ONE host compaction)
            dpmod = importlib.import_module("3dgsconverter_amd.processing.data_processor")
            names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + ["f_rest_%d" % i for i in range(45)] + \
                    ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
            dt = np.dtype([(nm, "f4") for nm in names])
            assert dt.itemsize == 248
            table = np.zeros(args.n, dtype=dt)
            table["x"], table["y"], table["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
            times, kept = [], 0
            for rep in range(3):
                t0 = time.perf_counter()
                proc = dpmod.DataProcessor(table, lazy=True)
                proc.apply_density_filter(sensitivity=0.5)
                proc.remove_flyers(k=args.k, threshold_factor=args.sigma)
                result = proc.data                      # the chain's survivor list applied to the 248-byte rows, once
                times.append(time.perf_counter() - t0)
                kept = len(result)
                del result, proc
            best = min(times[1:])
            host_bytes = args.n * 12 + args.n * 248 + kept * 248      # gather xyz (strided read) + read every row + write the survivors
            return {"workload": "%d x 248-byte host rows (62 f4 fields, uniform xyz L=%g): DataProcessor(lazy) density s=0.5 -> SOR k=%d "
                                "sigma=%g -> .data, host to host" % (args.n, args.extent, args.k, args.sigma),
                    "value": round(args.n / best / 1e6, 2), "unit": "Msplats/s", "ms_per_step": round(best * 1e3, 2), "steps": 2, "survivors": kept,
                    "all_runs_ms": [round(t * 1e3, 2) for t in times],
                    "roofline": {"bound": "host-dram", "achieved": round(host_bytes / best / 1e9, 2), "unit": "GB/s", "peak": None, "frac": None,
                                 "traffic": None, "algorithmic_bytes": int(host_bytes),
                                 "note": "host side: threaded gather of the 12-byte xyz out of 248-byte rows, 120 MB over PCIe, the device "
                                         "chain (configs.config2: ~2.7 ms), survivor list back, ONE threaded compaction of the 2.5 GB table "
                                         "(read every row, write the survivors).  The reference does np.column_stack + vertices[mask] twice: "
                                         "~2.6 s at this size (profiles/r01_e2e_probe.log)"}}

        def config3_one_gpu():
            # BASELINE.json configs[3] (50M splats, k=32) on ONE GPU: the N = 1 point of its strong-scaling curve (the N > 1 runs
            # of this file report configs[3] split over their ranks, with speedup_vs_one_gpu against ONE_GPU_CONFIG3_MS)
            n3 = 50_000_000
            x3 = synth_uniform(n3, 10.0, 0)
            r = run_sor(L, ctx, x3, 32, args.sigma, 5, 2, groups=True)
            r["workload"] = "BASELINE.json configs[3] on one GPU: %d uniform-random splats (L=10, seed 0), SOR k=32 sigma=%g, xyz resident in HBM" % (n3, args.sigma)
            r["scaling"] = "strong"
            r["n_gpus"] = 1
            r["roofline"] = sor_roofline(n3, 32, r["knn_kernel_ms"], single=False)
            r["roofline"]["algorithmic_bytes_per_splat"] = 12 * 16 + 16 + 4
            r["roofline"]["note"] = ("k=32 runs 2x2x1-cell bricks at 13.5 points per cell: a brick streams 4x4x3 cells = 12x its own points; "
                                     "priced with knn_brick's 148 B/splat for comparability; ") + r["roofline"]["note"]
            if want_cpu:   # cpu_baseline leg: the reference's CPU path on a bounded subsample of the same cloud
                from oracle import sor as osor
                m = 4_000_000
                workers = max(1, (os.cpu_count() or 2) - 1)
                t0 = time.perf_counter()
                osor.mean_dists_ckdtree(x3[:m], 32, workers=workers)
                cdt = time.perf_counter() - t0
                r["cpu_baseline"] = {"value": round(m / cdt / 1e6, 4), "unit": "Msplats/s", "cores": workers, "kind": "port",
                                     "sample": "the first %d of the 50M splats, k=32, once (%.2f s): cKDTree build + query(k+1); a SUBSAMPLE -- the "
                                               "full 50M reference run took 286 s on 8 cores of the build container (tests/golden/large_cases.json)" % (m, cdt)}
            return r

        def writers_2m():
            # SURVEY.md 8(f) rows 2-4 with a clock on them: the SOG writer's numeric core (sog.py:264-386, 457-459), the
            # compressed-PLY writer's Morton order + packers (compressed_ply.py:126-340) and the O(N) row filters
            # (data_processor.py:184-224), on a synthetic 2M-splat degree-3 table (62 x f4), HOST arrays in, HOST arrays out --
            # how the writers call them; K-Means, the rest of the SOG writer, is configs.config4
            m = 2_000_000
            r = np.random.default_rng(0)
            names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + ["f_rest_%d" % i for i in range(45)] + \
                    ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
            tab = np.zeros(m, dtype=[(nm, "f4") for nm in names])
            for nm in ("x", "y", "z"):
                tab[nm] = r.standard_normal(m, dtype=np.float32) * np.float32(3.0)
            for i in range(3):
                tab["f_dc_%d" % i] = r.standard_normal(m, dtype=np.float32)
                tab["scale_%d" % i] = r.standard_normal(m, dtype=np.float32) - np.float32(4.0)
            for i in range(45):
                tab["f_rest_%d" % i] = r.standard_normal(m, dtype=np.float32) * np.float32(0.1)
            tab["opacity"] = r.standard_normal(m, dtype=np.float32) * np.float32(2.0)
            for i in range(4):
                tab["rot_%d" % i] = r.standard_normal(m, dtype=np.float32)
            cply = importlib.import_module("3dgsconverter_amd.formats.compressed_ply_writer")

            def best_of(fn, reps=3):
                ts, out = [], None
                for _ in range(reps):
                    out = None             # the previous result is freed OUTSIDE the clock (munmap of hundreds of MB takes milliseconds)
                    t0 = time.perf_counter()
                    out = fn()
                    ts.append(time.perf_counter() - t0)
                return min(ts) * 1e3, out

            rot = np.ascontiguousarray(np.column_stack([tab["rot_%d" % i] for i in range(4)]))
            xyz2 = np.ascontiguousarray(np.column_stack([tab["x"], tab["y"], tab["z"]]))
            opa = np.ascontiguousarray(tab["opacity"])
            def host_lexsort():   # as formats/sog_writer.py:_encode_host calls it (round 6: the keys gathered in one threaded pass)
                zyx = L.host_gather_columns(tab, ["z", "y", "x"])
                return L.lexsort3(zyx[0], zyx[1], zyx[2])
            t_order, order = best_of(host_lexsort)
            sogw = importlib.import_module("3dgsconverter_amd.formats.sog_writer")
            sogw.encode(tab, 2, device_resident=True)
            resident = sogw.encode(tab, 2, device_resident=True, profile=True)["stage_ms"]
            t_pos, _ = best_of(lambda: [L.sog_positions(c) for c in L.host_gather_columns(tab, ["x", "y", "z"])])   # as _encode_host calls it
            t_quat, _ = best_of(lambda: L.sog_quats(rot))
            t_alpha, _ = best_of(lambda: L.sog_alpha(opa))
            t_cply, enc = best_of(lambda: cply.encode(tab), reps=3)

            def row_filters():
                ch = L.DeviceChain(xyz2)
                try:
                    a = ch.bbox_keep([-6.0, -6.0, -6.0, 6.0, 6.0, 6.0])
                    b = ch.ge_keep(opa, float(np.log(0.1 / 0.9)))
                    return a, b, len(ch.survivors())
                finally:
                    ch.close()
            t_rows, kept = best_of(row_filters)
            out = {"workload": "%d-splat degree-3 table (62 x f4, synthetic), host arrays in and out: SOG numeric core, compressed-PLY encode, bbox + alpha "
                               "row filters on the device chain (upload, two masks, two compactions, survivor list back)" % m,
                   "unit": "ms", "ms": {"sog_lexsort3": round(t_order, 2), "sog_positions_xyz": round(t_pos, 2), "sog_quats": round(t_quat, 2),
                                        "sog_alpha": round(t_alpha, 2), "compressed_ply_encode": round(t_cply, 2), "row_filters_bbox_alpha": round(t_rows, 2)},
                   "value": round(m / ((t_order + t_pos + t_quat + t_alpha) * 1e-3) / 1e6, 2), "value_unit": "Msplats/s through the SOG numeric core (sum of its four stages)",
                   "survivors_after_row_filters": kept[2],
                   "sog_core_resident_stage_ms": resident,
                   "sog_core_resident_note": "the same table through the device-resident core (formats/sog_device.py; configs.sog_write_core_10m at full "
                                             "size): `order` is the lexsort, `means_quats` the positions (all three axes) + quaternions, no per-stage PCIe",
                   "roofline": {"bound": "pcie", "achieved": None, "peak": None, "unit": "GB/s", "frac": None, "traffic": None,
                                "note": "host-to-host calls: every stage uploads its columns and downloads its texels, so each is bound by PCIe "
                                        "(4-28 B per splat each way at ~56 GB/s) and by numpy's part (extrema, flagged texels), not by HBM"}}
            if want_cpu:   # cpu_baseline leg: the reference's own numpy expressions (restated in oracle/sog.py, oracle/cply.py) on the same table
                from oracle import sog as osog, cply as ocply
                t0 = time.perf_counter()
                o = osog.order(tab)
                c_order = (time.perf_counter() - t0) * 1e3
                t0 = time.perf_counter()
                osog.positions(tab)
                c_pos = (time.perf_counter() - t0) * 1e3
                t0 = time.perf_counter()
                osog.quats(tab)
                c_quat = (time.perf_counter() - t0) * 1e3
                t0 = time.perf_counter()
                osog.opacity_u8(tab)
                c_alpha = (time.perf_counter() - t0) * 1e3
                sub = tab[:200_000]
                t0 = time.perf_counter()
                oo, _ = ocply.morton_order(sub["x"], sub["y"], sub["z"])
                ocply.encode(sub, oo, ["f_rest_%d" % i for i in range(45)])
                c_cply = (time.perf_counter() - t0) * 1e3
                out["cpu_baseline"] = {"value": round(m / ((c_order + c_pos + c_quat + c_alpha) * 1e-3) / 1e6, 3), "unit": "Msplats/s", "cores": 1, "kind": "port",
                                       "ms": {"sog_lexsort3": round(c_order, 1), "sog_positions_xyz": round(c_pos, 1), "sog_quats": round(c_quat, 1),
                                              "sog_alpha": round(c_alpha, 1), "compressed_ply_encode_200k_splats": round(c_cply, 1)},
                                       "sample": "the same 2M-splat table, once, numpy single-threaded as the reference runs it; the compressed-PLY "
                                                 "encode (a Python loop over 256-splat chunks in the reference) on the first 200 000 splats only",
                                       "order_identical_to_gpu": bool(np.array_equal(o, order))}
            return out

        def sog_write_core():
            # the SOG writer's numeric core on a DEVICE-RESIDENT table (formats/sog_device.py, csrc/sog_table.hip):
            # 10M x 62-field host table in -> every texel array of SogFormat.write out (sog.py:249-600; WebP / zip excluded).  The
            # table crosses PCIe once, texels come back; K-Means = configs.config4's batched palette, inside the clock here.
            sogw = importlib.import_module("3dgsconverter_amd.formats.sog_writer")
            m = args.n
            r = np.random.default_rng(0)
            names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + ["f_rest_%d" % i for i in range(45)] + \
                    ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
            tab = np.zeros(m, dtype=[(nm, "f4") for nm in names])
            for nm in names:
                if nm[0] == "n":
                    continue
                v = r.standard_normal(m, dtype=np.float32)
                if nm in "xyz":
                    v *= np.float32(3.0)
                elif nm.startswith("f_rest"):
                    v *= np.float32(0.1)
                elif nm.startswith("scale"):
                    v -= np.float32(4.0)
                elif nm == "opacity":
                    v *= np.float32(2.0)
                tab[nm] = v
            level = 2
            np.random.seed(0)
            L.release_arenas()
            t0 = time.perf_counter()
            sogw.encode(tab, level, device_resident=True)     # the FIRST write of a process: code objects, ~5 GB of work buffers allocated
            first_ms = (time.perf_counter() - t0) * 1e3       # (_lib.DeviceArena keeps them for the next write), stream / event pool
            prof = sogw.encode(tab, level, device_resident=True, profile=True)   # stage clock (a synchronisation after every stage)
            runs, core = [], None
            for _ in range(5):
                core = None                # (the previous call's 240 MB of images are freed outside the clock: their munmap is ~10 ms)
                t0 = time.perf_counter()
                core = sogw.encode(tab, level, device_resident=True)
                runs.append((time.perf_counter() - t0) * 1e3)
            runs.sort()
            med = runs[len(runs) // 2]
            tex_bytes = int(sum(v.nbytes for v in core["textures"].values()))
            pcie = tab.nbytes + tex_bytes
            res = {"workload": "%d x %d-byte host rows (62 f4 fields, degree-3 SH) -> all texel arrays of the SOG bundle (means_l/u, quats, scales, "
                               "sh0, shN labels + centroid indices + 3 codebooks), compression_level %d (64 x K=1024 palette, 10 Lloyd iterations); "
                               "WebP / zip excluded" % (m, tab.dtype.itemsize, level),
                   "value": round(m / (med * 1e-3) / 1e6, 2), "unit": "Msplats/s", "ms_per_step": round(med, 2), "ms_min": round(runs[0], 2),
                   "all_runs_ms": [round(v, 2) for v in runs], "steps": len(runs), "first_call_ms": round(first_ms, 2),
                   "first_call_note": "value / ms_per_step are the steady state of a process that writes again (work buffers held by "
                                      "_lib.DeviceArena, %.1f GB); first_call_ms is the first write of the process" % (L.arena(0).held_bytes() / 1e9),
                   "stage_ms": prof["stage_ms"],
                   "uncertain_texels": core["stats"],
                   "roofline": {"bound": "pcie", "achieved": round(pcie / (med * 1e-3) / 1e9, 2), "unit": "GB/s", "peak": 56.0,
                                "frac": round(pcie / (med * 1e-3) / 1e9 / 56.0, 4), "traffic": None, "algorithmic_bytes": int(pcie),
                                "note": "bytes over PCIe (the table up once, texels down) / the call; peak = the pageable-copy rate measured on these "
                                        "boxes when the host pages sit on the GPU's NUMA node (configs.host_to_host); the upload alone is %.0f %% of the "
                                        "call, the palette K-Means (configs.config4) most of the rest" % (100.0 * prof["stage_ms"].get("upload", 0.0) / max(sum(prof["stage_ms"].values()), 1e-9))}}
            # the table as the reference's converter hands it to the SOG writer: three u1 colour fields appended by add_rgb_from_sh
            # (converter.py:243-252 -> 251-byte rows).  Read on the device as they are (fields assembled from two words of the LDS tile)
            try:
                wide = L.host_append_u8_columns(tab, ("red", "green", "blue"), np.zeros((m, 3), np.uint8))
                sogw.encode(wide, level, device_resident=True)
                wruns, wcore = [], None
                for _ in range(3):
                    wcore = None
                    t0 = time.perf_counter()
                    wcore = sogw.encode(wide, level, device_resident=True)
                    wruns.append((time.perf_counter() - t0) * 1e3)
                res["rows_251_bytes"] = {"ms_per_step": round(sorted(wruns)[1], 2), "all_runs_ms": [round(v, 2) for v in wruns],
                                         "note": "the same table widened by red / green / blue u1 fields (what converter.py:243-252 hands the writer): no host repack",
                                         "geometry_textures_identical_to_248_byte_rows": bool(all(
                                             np.array_equal(wcore["textures"][k_], core["textures"][k_]) for k_ in ("means_l", "means_u", "quats")))}
                del wide, wcore
            except Exception as e:   # noqa: BLE001
                res["rows_251_bytes"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if want_cpu:   # cpu_baseline leg: the reference's own statements minus its K-Means calls, on a bounded subsample
                from oracle import sog as osog
                sub_n = 2_000_000
                sub = tab[:sub_n].copy()
                np.random.seed(1)
                gsub = sogw.encode(sub, level, device_resident=True)
                t0 = time.perf_counter()
                ref = osog.write_core_without_kmeans(sub, gsub["scale_codebook"], gsub["color_codebook"])
                cdt = time.perf_counter() - t0
                same = all(np.array_equal(ref[k_], gsub["textures"][k_]) for k_ in ("means_l", "means_u", "quats", "scales", "sh0"))
                res["cpu_baseline"] = {"value": round(sub_n / cdt / 1e6, 3), "unit": "Msplats/s", "cores": 1, "kind": "port",
                                       "sample": "the first %d rows of the same table, once (%.1f s): every numpy statement of SogFormat.write between the "
                                                 "table and its texel arrays (sog.py:264-265,279-386,391,408-459,499-503) as the reference runs them, single-threaded, "
                                                 "with the two scalar codebooks handed over -- its K-Means fits (sklearn: ~2.6 s per palette chunk, "
                                                 "configs.config4) are NOT in this number" % (sub_n, cdt),
                                       "textures_identical_to_gpu": bool(same)}
            return res

        if sustained is not None:
            if "ms_per_step" in sustained:
                sustained["vs_timed_region"] = round(sustained["ms_per_step"] / head["ms_per_step"], 4)
            configs["headline_sustained"] = sustained
        attempt("config1", config1)
        attempt("config1_brute", config1_brute)
        attempt("config3_one_gpu", config3_one_gpu)
        attempt("host_to_host", lambda: run_host_to_host(L, ctx, xyz, args.k, args.sigma, head["ms_per_step"]))
        attempt("config2", lambda: run_chain(L, ctx, gsx, xyz, 0.5, args.k, args.sigma, small, 2, cpu=want_cpu))
        attempt("config4", lambda: run_kmeans(L, ctx, gsx, args.n, 6, 2, cpu=want_cpu, lanes=args.lanes))
        attempt("clustered_1m", clustered)
        attempt("floaters_10m", floaters)
        attempt("blobs_10m_k25", blobs_k25)
        attempt("dropin_e2e_10m", dropin_e2e)
        attempt("writers_2m", writers_2m)
        attempt("sog_write_core_10m", sog_write_core)
        out["configs"] = configs
        # SURVEY.md 8(d) names two numbers for the metric; both at the top level, unambiguously: `value` (= value_resident) is
        # the whole step with the rows already in HBM -- the harness's definition --, value_host_to_host the call a user of
        # remove_flyers makes (PCIe both ways included; never the headline)
        out["value_resident"] = out["value"]
        out["value_host_to_host"] = configs.get("host_to_host", {}).get("value")
    out["bench_wall_s"] = round(time.perf_counter() - t_start, 1)
    os.write(args.json_fd, (json.dumps(out) + "\n").encode())


def main_slab_one_rank(args, gsx, L, ctx):
    """--exchange slab with one GPU: what ONE rank of an N-GPU job executes (partition, self-exchange through a one-rank
    RCCL communicator, slab KNN, certificate, un-permute, piece-sum statistics) -- no torch either"""
    gslab = importlib.import_module("3dgsconverter_amd.dist_slab")
    be = gslab.HipSlabBackend(ctx=ctx)
    comm = gslab.RcclComm(ctx, 0, 1, gslab.RcclComm.unique_id())
    xyz = synth_uniform(args.n, args.extent, 0)
    dev = ctx.alloc(xyz.nbytes).upload(xyz)
    res = {}

    def step():
        res["r"] = gslab.slab_sor(be, comm, dev, args.n, args.k, args.sigma)    # ONE C call: gsx_sor_slab_step_dev

    step()
    res["r"].check()
    t = time_steps(ctx, step, args.steps, args.warmup, event_slot=L.T_SOR_KNN, L=L)
    res["r"].check()
    mask = be.to_host(res["r"]["mask"], np.uint8, args.n)
    stats = be.to_host(res["r"]["stats"], np.float32, 3)
    g = kernel_groups(ctx, L, step, {"bin": L.T_SOR_BIN, "fallback": L.T_SOR_FALLBACK, "stats": L.T_SOR_STATS}, min(args.steps, 10))
    roof = sor_roofline(args.n, args.k, t["event_ms"], single=False)
    out = {"metric": "Msplats/sec SOR k=%d" % args.k, "value": round(args.n * args.steps / t["dt"] / 1e6, 2), "unit": "Msplats/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(t["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32 select / f64 finalize", "data": "synthetic",
           "config": {"workload": "%d uniform-random splats (L=%g, seed 0), SOR k=%d sigma=%g: the slab pipeline of one rank" % (args.n, args.extent, args.k, args.sigma),
                      "splats_per_gpu": args.n, "k": args.k, "sigma": args.sigma, "parallelism": gslab.PARALLELISM + " -- one-rank communicator"},
           "roofline": roof, "kernel_ms_per_step": dict({"knn": round(t["event_ms"], 4)}, **g), "survivors_rank0": int(mask.astype(bool).sum()),
           "threshold": float(stats[2])}
    ctx.check()
    comm.close()
    os.write(args.json_fd, (json.dumps(out) + "\n").encode())


def _golden_files():
    out = {}
    for name in ("large_cases.json", "cases.json"):
        try:
            with open(os.path.join(ROOT, "tests", "golden", name)) as f:
                out[name] = json.load(f)
        except OSError:
            out[name] = {}
    return out


def golden_sor(n_total, extent, k, sigma):
    """the committed hash of a run of the REFERENCE on the uniform seed-0 cloud of this size (tests/golden/*.json, generated by
    oracle/make_golden*.py against /root/reference), or None"""
    files = _golden_files()
    cands = [(nm, c) for nm, c in files["large_cases.json"].items() if isinstance(c, dict)]
    cands += [(nm, c) for nm, c in files["cases.json"].get("sor", {}).items()]
    for nm, c in cands:
        ds = c.get("dataset", {})
        if (ds.get("kind") == "uniform" and ds.get("seed") == 0 and ds.get("n") == n_total and float(ds.get("extent", -1)) == float(extent)
                and c.get("k", c.get("k_used")) == k and float(c.get("sigma", c.get("sigma_used", -1))) == float(sigma) and "mask_sha" in c):
            return {"mask_sha": c["mask_sha"], "survivors": c["survivors"], "threshold_hex": c["threshold_hex"], "source": "tests/golden: " + nm}
    return None


def golden_chain(n_total, extent, sensitivity, k, sigma):
    for nm, c in _golden_files()["large_cases.json"].items():
        ds = c.get("dataset", {}) if isinstance(c, dict) else {}
        if ("final_mask_sha" in c and ds.get("kind") == "uniform" and ds.get("seed") == 0 and ds.get("n") == n_total
                and float(ds.get("extent", -1)) == float(extent) and float(c.get("sensitivity", -1)) == float(sensitivity)
                and c.get("k") == k and float(c.get("sigma", -1)) == float(sigma)):
            return dict(c, source="tests/golden: " + nm)
    return None


# --------------------------------------------------------------------------------------------------- N > 1: one process per GPU
def main_multi(args):
    """`python bench.py --gpus N` run plainly becomes the launcher of its own N ranks (3dgsconverter_amd/launch.py); under
    `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks already exist.  Either way: numpy +
    ctypes on libgsx_hip.so, every collective a gsx_comm_* call of the C library (RCCL over xGMI; the shared-memory hostwire
    when the ranks have to share GPUs), the unique id through a file."""
    launch = importlib.import_module("3dgsconverter_amd.launch")
    rank, local_rank, world = launch.rank_env()
    if world == 1:
        sys.exit(launch.spawn_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # stdout carries exactly ONE JSON line: RCCL (banner, warnings) and other libraries print to the C stdout, so
    # file descriptor 1 is pointed at stderr for the whole run and the JSON goes to a private copy of the real one
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # A rank that never gets past communicator set-up (ncclCommInitRank waiting for a peer that died, a GPU the launcher
    # did not show) must not hang the job: after GSX_COMM_TIMEOUT seconds (default 600; the ctypes calls release the GIL, so
    # this thread runs while the main one sits inside librccl) rank 0 prints a JSON line saying so and every rank exits.
    watchdog = launch.comm_watchdog(rank, world, json_fd, float(os.environ.get("GSX_COMM_TIMEOUT", "600")),
                                    {"metric": "Msplats/sec SOR k=%d" % args.k, "unit": "Msplats/s", "n_gpus": world, "steps": args.steps,
                                     "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "data": "synthetic"})
    gsx = importlib.import_module("3dgsconverter_amd")
    gdist = importlib.import_module("3dgsconverter_amd.dist")
    gslab = importlib.import_module("3dgsconverter_amd.dist_slab")
    L = gsx._lib
    device, transport = launch.pick_device_and_transport(local_rank, world, L.device_count())
    ctx = L.Context(device)
    for kv in args.param:
        name, val = kv.split("=")
        ctx.set_param(name, float(val))
    be = gslab.HipSlabBackend(ctx=ctx)
    if os.environ.get("GSX_TEST_STALL_RANK") == str(rank):   # tests/test_dist_gpu.py: a rank that never shows up
        time.sleep(1e6)
    transport = launch.agree_transport(rank, world, L.device_uid(device))   # distinct GPUs -> RCCL, shared -> hostwire
    uid = launch.exchange_unique_id(rank, lambda: gslab.RcclComm.unique_id(transport))
    comm = gslab.RcclComm(ctx, rank, world, uid)
    comm.barrier()                      # every rank holds its communicator: the id file has done its job
    launch.retire_unique_id(rank)
    watchdog.stage("communicator up (%s)" % comm.transport)
    exchange = {"path": "replicated (requested)" if args.exchange == "replicated" else "slab"}

    def run(n, extent, steps, warmup, k=None, shard=None):
        """Time `steps` SOR steps on a fresh n-splat shard (shard: this rank's rows of a GLOBAL cloud; None: a cloud of its own,
        seed = rank); returns a dict of raw measurements."""
        k = args.k if k is None else k
        rows = ctx.alloc(12 * n + 16).upload(synth_uniform(n, extent, rank) if shard is None else np.ascontiguousarray(shard, dtype=np.float32))
        exchange.pop("certified", None)

        def step():
            if exchange["path"] == "slab":
                try:
                    r = gslab.slab_sor(be, comm, rows, n, k, args.sigma)    # ONE C call: gsx_sor_slab_step_dev
                    if not exchange.get("certified"):   # first step on this cloud: read the certificate before relying on
                        r.check()                        # the slab path (later steps read it after the timed region)
                        exchange["certified"] = True
                    return r
                except gslab.SlabUncertain as e:   # raised on every rank together (decided from all-reduced / gathered data)
                    exchange["path"] = "replicated (slab exchange declined: %s)" % e
                    be.set_adaptive(True)          # the clouds that get here are the ones the adaptive paths exist for
            return gdist.replicated_sor(be, comm, rows, n, k, args.sigma, algo=args.algo)

        def host(res):
            return be.to_host(res["mask"], np.uint8, n).astype(bool), be.to_host(res["stats"], np.float32, 3)

        if exchange["path"] == "slab" and not exchange.get("cross_checked"):
            # the slab exchange and the replicated one (all-gather of the rows, every rank bins everything) must agree
            # bit for bit -- statistics and this rank's survivor mask -- before the slab path is what gets timed
            ref_mask, ref_stats = host(gdist.replicated_sor(be, comm, rows, n, k, args.sigma, algo=args.algo).check())
            ok = 1
            try:
                got = step()
                if exchange["path"] == "slab":
                    mask, stats = host(got)
                    ok = 1 if (np.array_equal(stats.view(np.uint32), ref_stats.view(np.uint32)) and np.array_equal(mask, ref_mask)) else 0
            except Exception as e:   # noqa: BLE001 -- anything the exchange raises on this rank alone
                sys.stderr.write("[bench] rank %d: slab exchange failed its cross-check: %r\n" % (rank, e))
                comm.abort()
                raise
            if int(comm.reduce_scalar(ok, gslab.KIND_I64_MIN)) == 0:
                exchange["path"] = "replicated (slab exchange disagreed with the replicated exchange on this cloud)"
            exchange["cross_checked"] = True
        res = None
        for _ in range(warmup):
            res = step()
        comm.barrier()
        ctx.set_param("timing_mask", 1 << L.T_SOR_KNN)
        ctx.set_timing(True)
        ctx.reset_timing()
        comm.barrier()                  # (drains this rank's stream, then every rank's)
        t0 = time.perf_counter()
        for i in range(steps):
            ctx.set_timing(i % EVENT_EVERY == 0)
            res = step()
        comm.barrier()
        dt = time.perf_counter() - t0
        ctx.set_timing(True)
        dt = float(comm.reduce_scalar(dt, gslab.KIND_F64_MAX))   # the slowest rank's clock
        res.check()                     # the certificate of the last timed step (same cloud every step)
        n_knn, ms_knn = ctx.timing(L.T_SOR_KNN)
        ctx.set_timing(False)
        mask, stats = host(res)
        # ---- where a step's time goes: a separate pass with every slot recording (HIP events on this rank's stream; not in
        # the timed region), the slowest rank's figure per phase
        phases = None
        if exchange["path"] == "slab":
            slots = {"knn_ms": L.T_SOR_KNN, "bin_ms": L.T_SOR_BIN, "fallback_ms": L.T_SOR_FALLBACK, "stats_ms": L.T_SOR_STATS,
                     "slab_kernels_ms": L.T_SLAB_PREP, "exchange_rows_ms": L.T_SLAB_ROWS, "exchange_means_ms": L.T_SLAB_MEANS,
                     "collectives_ms": L.T_SLAB_COLL}
            side = max(2, min(steps, 5))
            ctx.set_param("timing_mask", 0xfff)
            ctx.set_timing(True)
            ctx.reset_timing()
            comm.barrier()
            for _ in range(side):
                step()
            comm.barrier()
            phases = {}
            for name, slot in slots.items():
                phases[name] = round(float(comm.reduce_scalar(ctx.timing(slot)[1] / side, gslab.KIND_F64_MAX)), 4)
            ctx.set_timing(False)
            phases["note"] = ("HIP events around each phase on every rank's stream, a separate pass of %d steps, MAX over the ranks; "
                              "an exchange interval includes the wait for the slowest peer to arrive" % side)
        rows.free()
        return {"dt": dt, "knn_ms": ms_knn / max(n_knn, 1), "survivors": int(mask.sum()), "threshold": float(stats[2]), "phases": phases,
                "mask": mask, "stats": stats}

    def gather_mask(mask_local, sizes):
        """every rank's survivor mask (u8 per splat) on every rank, in index order: one all-gather of padded blocks"""
        cap = max(max(sizes), 1)
        sb, rb = be.buf("par_mask_s", cap), be.buf("par_mask_r", cap * world)
        padded = np.zeros(cap, np.uint8)
        padded[:len(mask_local)] = mask_local
        be.from_host(sb, padded)
        comm.all_gather(sb, rb, cap)
        allm = be.to_host(rb, np.uint8, cap * world).reshape(world, cap)
        return np.concatenate([allm[r, :sizes[r]] for r in range(world)]).astype(bool)

    def global_shard(n_total, extent):
        """this rank's index range of the GLOBAL seed-0 cloud SURVEY.md 8(c) defines (the cloud the reference-run hashes of
        tests/golden/ were computed on): every rank generates the whole cloud (~1 s at 50M) and keeps its rows"""
        sizes = [(r + 1) * n_total // world - r * n_total // world for r in range(world)]
        lo = rank * n_total // world
        full = synth_uniform(n_total, extent, 0)
        return np.ascontiguousarray(full[lo:lo + sizes[rank]]), sizes

    try:
        main_run = run(args.n, args.extent, args.steps, args.warmup)
        watchdog.stage("headline timed")
        config3 = None
        if not args.no_secondary:
            # BASELINE.json configs[3]: 50M splats, SOR k=32, sharded by index across the GPUs of the job (the 8-GPU case; at
            # other N the same 50M are split N ways).  Reported next to the headline, never instead of it.
            try:
                n_tot3, k3 = int(args.n3), int(args.k3)
                s3, w3 = max(3, min(args.steps, 10)), 2
                shard3, sizes3 = global_shard(n_tot3, 10.0)
                r3 = run(sizes3[rank], 10.0, s3, w3, k=k3, shard=shard3)
                del shard3
                full3 = gather_mask(r3["mask"].view(np.uint8), sizes3)
                gold3 = golden_sor(n_tot3, 10.0, k3, args.sigma)
                sha3 = hashlib.sha256(np.packbits(full3)).hexdigest()[:16]
                config3 = {"workload": "BASELINE.json configs[3]: %d uniform-random splats (L=10, the GLOBAL seed-0 cloud, sharded by index) over "
                                       "%d GPU(s), SOR k=%d sigma=%g" % (n_tot3, world, k3, args.sigma),
                           "value": round(n_tot3 * s3 / r3["dt"] / 1e6, 2), "unit": "Msplats/s",
                           "ms_per_step": round(r3["dt"] / s3 * 1e3, 4), "steps": s3, "exchange": exchange["path"],
                           "scaling": "strong", "n_gpus": world, "one_gpu_ms_per_step": ONE_GPU_CONFIG3_MS if (n_tot3, k3) == (50_000_000, 32) else None,
                           "speedup_vs_one_gpu": round(ONE_GPU_CONFIG3_MS / (r3["dt"] / s3 * 1e3), 3) if (n_tot3, k3) == (50_000_000, 32) else None,
                           "speedup_note": "strong scaling of BASELINE configs[3]: the same 50M splats on 1 GPU take %.2f ms per step "
                                           "(profiles/r05_bench_50m_k32.json, measured on 1xMI355X); this line divides that by this run's "
                                           "step time" % ONE_GPU_CONFIG3_MS,
                           "knn_kernel_ms": round(r3["knn_ms"], 4), "survivors_rank0": r3["survivors"], "phases": r3["phases"],
                           # parity INSIDE the multi-GPU run: the whole job's mask against the hash of a run of the
                           # reference itself on the same cloud
                           "survivors": int(full3.sum()), "mask_sha16": sha3, "threshold_hex": np.float32(r3["stats"][2]).tobytes().hex(),
                           "mask_matches_reference_run": None if gold3 is None else bool(
                               sha3 == gold3["mask_sha"] and int(full3.sum()) == gold3["survivors"]
                               and np.float32(r3["stats"][2]).tobytes().hex() == gold3["threshold_hex"]),
                           "reference_run": None if gold3 is None else gold3["source"]}
            except Exception as e:   # noqa: BLE001 -- the headline line must survive a failure here.  A GsxError is raised by every rank that
                # hits it, and so is anything this (deterministic, same-input) code raises before its collectives; a rank that fails ALONE
                # leaves its peers in a collective, which the watchdog ends with a JSON error line
                config3 = {"workload": "BASELINE.json configs[3]", "error": repr(e)}
        ctx.check()
        watchdog.stage("configs[3] timed")
        config2 = None
        if not args.no_secondary:
            # BASELINE.json configs[2] across the GPUs of the job: density (per-rank voxel histograms merged, dist_density.py) ->
            # device compaction of each rank's rows -> SOR k=16 on the survivors (slab exchange), on the GLOBAL seed-0 cloud, with
            # both masks checked against the hashes of the reference's own run of that chain
            try:
                import ctypes as C
                gdens = importlib.import_module("3dgsconverter_amd.dist_density")
                n_tot2, k2, sens2 = int(args.n2), 16, 0.5
                shard2, sizes2 = global_shard(n_tot2, 5.0)
                n_loc2 = sizes2[rank]
                rows2 = ctx.alloc(12 * n_loc2 + 16).upload(shard2)
                del shard2
                keep_rows, keep_orig = be.buf("c2_rows", 12 * n_loc2 + 16), be.buf("c2_orig", 4 * n_loc2 + 16)

                def sor_on(buf, n, k):
                    if exchange["path"] == "slab":
                        try:
                            return gslab.slab_sor(be, comm, buf, n, k, args.sigma).check(), "slab"
                        except gslab.SlabUncertain:     # raised on every rank together
                            pass
                    return gdist.replicated_sor(be, comm, buf, n, k, args.sigma, algo=args.algo).check(), "replicated"

                def chain_step():
                    d = gdens.sharded_density(be, comm, rows2, n_loc2, sensitivity=sens2)
                    if d["empty"]:
                        return d, 0, None, "none"
                    n_out = C.c_int64()
                    L.check(ctx.lib.gsx_compact_rows_dev(ctx.handle, rows2.ptr, None, d["mask"].ptr, n_loc2, keep_rows.ptr, keep_orig.ptr,
                                                         C.byref(n_out)), "gsx_compact_rows_dev")
                    r, path = sor_on(keep_rows, int(n_out.value), k2)
                    return d, int(n_out.value), r, path
                s2 = max(2, min(args.steps, 5))
                chain_step()
                comm.barrier()
                t0 = time.perf_counter()
                for _ in range(s2):
                    d2, nk2, r2, path2 = chain_step()
                comm.barrier()
                dt2 = float(comm.reduce_scalar(time.perf_counter() - t0, gslab.KIND_F64_MAX))
                dmask = be.to_host(d2["mask"], np.uint8, n_loc2).astype(bool) if not d2["empty"] else np.zeros(n_loc2, bool)
                final = np.zeros(n_loc2, bool)
                thr_hex = None
                if r2 is not None:
                    smask = be.to_host(r2["mask"], np.uint8, nk2).astype(bool)
                    final[be.to_host(keep_orig, np.uint32, nk2)[smask]] = True
                    thr_hex = np.float32(be.to_host(r2["stats"], np.float32, 3)[2]).tobytes().hex()
                full_d, full_f = gather_mask(dmask.view(np.uint8), sizes2), gather_mask(final.view(np.uint8), sizes2)
                rows2.free()
                gold2 = golden_chain(n_tot2, 5.0, sens2, k2, args.sigma)
                sha_d = hashlib.sha256(np.packbits(full_d)).hexdigest()[:16]
                sha_f = hashlib.sha256(np.packbits(full_f)).hexdigest()[:16]
                config2 = {"workload": "BASELINE.json configs[2] over %d GPU(s): %d uniform-random splats (L=5, the GLOBAL seed-0 cloud, sharded by index), "
                                       "density sensitivity 0.5 (merged per-rank voxel histograms) -> device compaction -> SOR k=%d sigma=%g on the survivors"
                                       % (world, n_tot2, k2, args.sigma),
                           "value": round(n_tot2 * s2 / dt2 / 1e6, 2), "unit": "Msplats/s", "ms_per_step": round(dt2 / s2 * 1e3, 4), "steps": s2,
                           "n_gpus": world, "scaling": "strong", "sor_exchange": path2,
                           "after_density": int(full_d.sum()), "survivors": int(full_f.sum()), "density_mask_sha16": sha_d, "final_mask_sha16": sha_f,
                           "sor_threshold_hex": thr_hex,
                           "mask_matches_reference_run": None if gold2 is None else bool(
                               sha_d == gold2["density_mask_sha"] and int(full_d.sum()) == gold2["density_kept"] and sha_f == gold2["final_mask_sha"]
                               and int(full_f.sum()) == gold2["final_survivors"] and thr_hex == gold2["sor_threshold_hex"]),
                           "reference_run": None if gold2 is None else gold2["source"]}
            except Exception as e:   # noqa: BLE001 -- as for configs[3]
                config2 = {"workload": "BASELINE.json configs[2]", "error": repr(e)}
        ctx.check()
        watchdog.stage("configs[2] timed")
        # cpu_baseline leg (rank 0, its own shard: what one GPU of the job replaces), while the other ranks wait at the barrier
        cpu = None
        if rank == 0 and not args.no_cpu_baseline:
            from oracle import sor as osor
            xyz0 = synth_uniform(args.n, args.extent, 0)
            workers = max(1, (os.cpu_count() or 2) - 1)
            t0 = time.perf_counter()
            osor.mean_dists_ckdtree(xyz0, args.k, workers=workers)
            cdt = time.perf_counter() - t0
            cpu = {"value": round(args.n / cdt / 1e6, 4), "unit": "Msplats/s", "cores": workers, "kind": "port",
                   "sample": "rank 0's shard (%d splats), once (%.2f s): scipy cKDTree build + query(k+1) workers=%d, as "
                             "data_processor.py:156-173 does -- the work ONE GPU of this job replaces; the job's CPU equivalent is the "
                             "same call on all %d splats" % (args.n, cdt, workers, args.n * world)}
        watchdog.stage("cpu baseline done")
    except BaseException:
        comm.abort()
        raise
    if rank == 0:
        dt = main_run["dt"]
        n_total = world * args.n
        shared = L.device_count() < world
        out = {
            "metric": "Msplats/sec SOR k=%d" % args.k, "value": round(n_total * args.steps / dt / 1e6, 2), "unit": "Msplats/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 select / f64 finalize", "data": "synthetic",
            "config": {"workload": "%d uniform-random splats per GPU (L=%g, seed=rank), SOR k=%d sigma=%g, exact KNN, xyz resident in HBM"
                                   % (args.n, args.extent, args.k, args.sigma),
                       "splats_per_gpu": args.n, "k": args.k, "sigma": args.sigma, "algo": "grid-binned exact KNN",
                       "parallelism": gslab.PARALLELISM if exchange["path"] == "slab" else gdist.PARALLELISM + " -- " + exchange["path"],
                       "transport": comm.transport + (" (%d ranks share %d GPU(s): a functional run of the N-rank code, not a scaling "
                                                      "measurement)" % (world, L.device_count()) if shared else " over xGMI, one rank per GPU"),
                       "launcher": "bench.py's own ranks (3dgsconverter_amd/launch.py)" if os.environ.get("GSX_SPAWNED") else "external (RANK / WORLD_SIZE)",
                       "host_runtime": "numpy + ctypes on libgsx_hip.so; torch is not imported"},
            "roofline": sor_roofline(args.n, args.k, main_run["knn_ms"], args.algo, n_total=n_total, single=False),
            "kernel_ms_per_step": {"knn": round(main_run["knn_ms"], 4)},
            "survivors_rank0": main_run["survivors"], "threshold": main_run["threshold"]}
        if main_run["phases"] is not None:
            out["phases_ms_per_step"] = main_run["phases"]
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if config3 is not None:
            out["config3"] = config3
        if config2 is not None:
            out["config2"] = config2
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    comm.barrier()
    watchdog.done()
    comm.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # (no flags: a 0.2 s timed region; the driver passes its own K and W)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="sor", choices=["sor", "kmeans"])
    ap.add_argument("--n", type=int, default=10_000_000, help="splats per GPU")
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--extent", type=float, default=5.0)
    ap.add_argument("--algo", type=int, default=0, help="0 auto (grid), 1 brute force, 2 grid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline only: skip the other configurations (N=1) / the 50M k=32 configs[3] line (N>1)")
    ap.add_argument("--param", action="append", default=[], help="name=value library knob (A/B runs)")
    ap.add_argument("--lanes", type=int, default=0, help="config4: concurrent contexts for the independent K-Means chunks (0 = the library's default)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "slab", "replicated"],
                    help="N>1 data path: slab (default) or the replicated all-gather; 'slab' with --gpus 1 runs the slab "
                         "pipeline through a one-rank RCCL communicator (what one rank of an N-GPU job executes)")
    ap.add_argument("--n3", type=int, default=50_000_000, help="N>1: total splats of the configs[3] line (k=32), split over the ranks")
    ap.add_argument("--k3", type=int, default=32, help="N>1: k of the configs[3] line (tests: a size / k pair a committed reference-run hash exists for)")
    ap.add_argument("--n2", type=int, default=10_000_000, help="N>1: total splats of the configs[2] line (density 0.5 -> SOR k=16), split over the ranks")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or args.gpus > 1:   # (either spawns its own ranks when no launcher did)
        if args.workload == "kmeans":
            return importlib.import_module("tools.bench_kmeans").main(args)
        return main_multi(args)
    # stdout carries exactly ONE JSON line: RCCL (banner, warnings) and other libraries print to the C stdout, so
    # file descriptor 1 is pointed at stderr for the whole run and the JSON goes to a private copy of the real one
    sys.stdout.flush()
    args.json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.workload == "kmeans":
        return importlib.import_module("tools.bench_kmeans").main(args)
    return main_single(args)


if __name__ == "__main__":
    main()
