#!/usr/bin/env python
"""bench.py -- SOR (Statistical Outlier Removal) throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n SPLATS_PER_GPU] [--k 16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of synthetic splats whose xyz is already
resident in HBM: (all-gather of xyz, N>1) -> grid binning -> exact KNN mean distance
(knn_brick + knn_ring) -> (all-gather of mean distances, N>1) -> numpy-exact mean/std/
threshold -> survivor mask.  Workload at N=1: BASELINE.json configs[1] (1M uniform splats,
L=10, seed 0, k=16, sigma=1.0); each extra GPU adds one more 1M-splat index shard (weak
scaling).  Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_SLOTS_PER_S = 78.6e12   # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz


def synth_shard(n, extent, seed):
    """SURVEY.md 8(c) generator; shard r of the global cloud uses seed r."""
    return np.random.default_rng(seed).random((n, 3), dtype=np.float32) * np.float32(extent)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=1_000_000, help="splats per GPU")
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--extent", type=float, default=10.0)
    ap.add_argument("--algo", type=int, default=0, help="0 auto (grid), 1 brute force, 2 grid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch  # first: its bundled HIP runtime (same SONAME) is the one the .so binds to
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    gsx = importlib.import_module("3dgsconverter_amd")
    from importlib import import_module
    gdist = import_module("3dgsconverter_amd.dist")
    L = gsx._lib
    compute = gdist.HipCompute(local_rank)
    ctx = compute.ctx

    xyz_host = synth_shard(args.n, args.extent, rank)
    xyz_local = torch.from_numpy(xyz_host).to(dev)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        return gdist.sharded_sor(xyz_local, args.k, args.sigma, compute, algo=args.algo)

    for _ in range(args.warmup):
        res = step()
    barrier()
    # HIP events around the dominant kernel only inside the timed region: each event pair costs stream
    # time (measured: all five slots = +0.04 ms on the 0.38 ms step, profiles/r01_timing_overhead.log)
    ctx.set_param("timing_mask", 1 << L.T_SOR_KNN)
    ctx.set_timing(True)
    ctx.reset_timing()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())
    n_knn, ms_knn = ctx.timing(L.T_SOR_KNN)

    # the other kernel groups: a short separate pass with every slot recording (not part of the timed region)
    ctx.set_param("timing_mask", 0xff)
    ctx.reset_timing()
    side_steps = min(args.steps, 10)
    for _ in range(side_steps):
        step()
    barrier()
    n_bin, ms_bin = ctx.timing(L.T_SOR_BIN)
    n_fb, ms_fb = ctx.timing(L.T_SOR_FALLBACK)
    n_st, ms_st = ctx.timing(L.T_SOR_STATS)
    ctx.set_timing(False)
    info = ctx.sor_knn(*(lambda t: (t.data_ptr(), t.data_ptr() + 4, t.data_ptr() + 8))(xyz_local), 3,
                       args.n, 0, args.n, args.k, res.mean_dists_local.data_ptr(), algo=args.algo,
                       want_info=True) if world == 1 else None

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    n_total = world * args.n
    ms_per_step = dt / args.steps * 1e3
    value = n_total * args.steps / dt / 1e6  # Msplats/s, whole job

    # ---- roofline of the dominant kernel (knn_brick, or knn_brute with --algo 1)
    knn_ms = ms_knn / max(n_knn, 1)
    if args.algo != 1:
        # algorithmic bytes of ONE knn_brick launch: every brick streams its
        # 4x4x4-cell neighbourhood once (16 B/point, = 8x its own 2x2x2 cells on average), every
        # query reads its own point (16 B) and writes one f32: (8*16 + 16 + 4) B per splat.
        bytes_per_splat = 8 * 16 + 16 + 4
        kernel = "knn_brick_kernel"
    else:
        # brute force, SURVEY.md 8(d): 16 B per reference point per 512-query workgroup + 20 B/query
        bytes_per_splat = 16.0 * (n_total / 512.0) + 20
        kernel = "knn_brute_kernel"
    alg_bytes = bytes_per_splat * args.n
    achieved = alg_bytes / (knn_ms * 1e-3) / 1e9 if knn_ms > 0 else 0.0
    traffic, traffic_note = None, None
    try:  # PMC counters cannot be read from inside the process: the last rocprofv3 --pmc passes are committed
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            pmc = json.load(f).get(kernel, {}).get("%d:%d" % (args.n, args.k))
        if pmc and world == 1:
            traffic = pmc["fetch_bytes"] + pmc["write_bytes"]
            traffic_note = ("rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE per launch (profiles/pmc_latest.json); fetch as "
                            "counted, x2-corrected upper bound %d B" % (pmc["fetch_x2_upper"] + pmc["write_bytes"]))
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_note": traffic_note, "algorithmic_bytes": int(alg_bytes),
                "valu_busy_frac_pmc": 0.82, "valu_source": "profiles/r01_pmc_10m_sq.txt (4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE/8), 10M splats)",
                "kernel_ms": round(knn_ms, 4), "algorithmic_bytes_per_splat": bytes_per_splat,
                "note": "kernel is FP32/FP64 VALU-issue bound, not HBM bound (DESIGN.md section 5); "
                        "PMC HBM traffic per launch is in profiles/"}

    out = {
        "metric": "Msplats/sec SOR k=%d" % args.k, "value": round(value, 2), "unit": "Msplats/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 select / f64 finalize",
        "data": "synthetic",
        "config": {"workload": "%d uniform-random splats per GPU (L=%g, seed=rank), SOR k=%d sigma=%g, "
                               "exact KNN, xyz resident in HBM" % (args.n, args.extent, args.k, args.sigma),
                   "splats_per_gpu": args.n, "k": args.k, "sigma": args.sigma,
                   "algo": "grid-binned exact KNN" if args.algo != 1 else "LDS-tiled brute force",
                   "parallelism": "index-sharded queries, all-gather xyz + all-gather mean_dists (RCCL)"
                   if world > 1 else "single GPU"},
        "roofline": roofline,
        "kernel_ms_per_step": {"knn": round(ms_knn / args.steps, 4), "bin": round(ms_bin / side_steps, 4),
                               "fallback": round(ms_fb / side_steps, 4), "stats": round(ms_st / side_steps, 4),
                               "note": "knn: HIP events inside the timed region; the others: a separate pass of %d steps" % side_steps},
        "survivors_rank0": int(res.mask_local.sum().item()),
        "threshold": float(res.stats[2].item()),
    }
    if info is not None:
        out["grid"] = info

    if world == 1 and not args.no_cpu_baseline:
        # reported baseline, not the target: the reference's CPU path (cKDTree + numpy, restated in
        # oracle/sor.py because the reference never returns its mask) on the SAME cloud, host cores.
        from oracle import sor as osor
        workers = max(1, (os.cpu_count() or 2) - 1)
        t0 = time.perf_counter()
        ref = osor.sor(xyz_host, args.k, args.sigma, workers=workers)
        cpu_dt = time.perf_counter() - t0
        same = bool(np.array_equal(ref["mask"], res.mask_local.cpu().numpy().astype(bool)))
        out["cpu_baseline"] = {"value": round(args.n / cpu_dt / 1e6, 4), "unit": "Msplats/s", "cores": workers,
                               "kind": "port", "sample": "the full %d-splat workload, once (%.2f s): "
                               "scipy cKDTree query workers=%d + numpy stats" % (args.n, cpu_dt, workers),
                               "mask_identical_to_gpu": same}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
