#!/usr/bin/env python
"""bench.py -- SOR (Statistical Outlier Removal) throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n SPLATS_PER_GPU] [--k 16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --workload kmeans        # BASELINE.json configs[4] (SOG palette K-Means), 1 GPU

One SOR step = one pass of the hot path over one batch of synthetic splats whose xyz is already
resident in HBM: grid binning -> exact KNN mean distance (knn_brick + knn_ring) -> numpy-exact
mean/std/threshold -> survivor mask; with N > 1 the splats are sharded by index and the step is the slab
exchange of 3dgsconverter_amd/dist_slab.py (RCCL all-to-all of xyz rows + halo, slab KNN, all-to-all of
the mean distances back, all-gather of numpy's 8192-piece sums).  Workload at N=1: the configuration
BASELINE.json's target is quoted on -- 10M uniform-random splats (L=5, seed 0: SURVEY.md 8(d)
config 3's cloud), k=16, sigma=1.0; each extra GPU adds one more 10M-splat index shard (weak
scaling).  The 1M-splat configs[1] cloud is timed in the same run and reported under "secondary".
Prints ONE JSON line (rank 0).
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

EVENT_EVERY = 4              # steps of the timed region between two HIP-event pairs around the dominant kernel
VALU_PEAK_GINST = 256 * 4 * 2.4 / 4   # G wave64 instructions per second (MI355X: 256 CUs, 4 SIMD16 each, 2.4 GHz)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_SLOTS_PER_S = 78.6e12   # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz


def synth_shard(n, extent, seed):
    """SURVEY.md 8(c) generator; shard r of the global cloud uses seed r."""
    return np.random.default_rng(seed).random((n, 3), dtype=np.float32) * np.float32(extent)


def load_pmc(kernel, n, k):
    """Static PMC figures of the last committed rocprofv3 --pmc passes (counters cannot be read in-process)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            return json.load(f).get(kernel, {}).get("%d:%d" % (n, k))
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sor", choices=["sor", "kmeans"])
    ap.add_argument("--n", type=int, default=10_000_000, help="splats per GPU")
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--extent", type=float, default=5.0)
    ap.add_argument("--algo", type=int, default=0, help="0 auto (grid), 1 brute force, 2 grid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 1M-splat configs[1] line (N=1) / the 50M k=32 configs[3] line (N>1)")
    ap.add_argument("--config3", action="store_true", help="N=1: also run the 50M-splat k=32 configs[3] workload on this one GPU")
    ap.add_argument("--param", action="append", default=[], help="name=value library knob (A/B runs)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "slab", "replicated"],
                    help="N>1 data path: slab (default) or the replicated all-gather; 'slab' with --gpus 1 runs the slab "
                         "pipeline through a one-rank RCCL communicator (what one rank of an N-GPU job executes)")
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line: RCCL (banner, warnings) and other libraries print to the C stdout, so
    # file descriptor 1 is pointed at stderr for the whole run and the JSON goes to a private copy of the real one
    sys.stdout.flush()
    args.json_fd = os.dup(1)
    os.dup2(2, 1)

    if args.workload == "kmeans":
        from importlib import import_module
        return import_module("tools.bench_kmeans").main(args)

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
    gdist = importlib.import_module("3dgsconverter_amd.dist")
    gslab = importlib.import_module("3dgsconverter_amd.dist_slab")
    L = gsx._lib
    compute = gdist.HipCompute(local_rank)
    ctx = compute.ctx
    for kv in args.param:
        name, val = kv.split("=")
        ctx.set_param(name, float(val))
    # N > 1: spatial slabs, every collective of the data path is RCCL called from libgsx_hip.so on the library's
    # stream (3dgsconverter_amd/dist_slab.py); torch.distributed only hands the 128-byte unique id out and times
    slab_be = slab_comm = None
    exchange = {"path": "single GPU"}
    if (world > 1 and args.exchange != "replicated") or args.exchange == "slab":
        uid = [gslab.RcclComm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        slab_be = gslab.HipSlabBackend(ctx=ctx)
        slab_comm = gslab.RcclComm(ctx, rank, world, uid[0])
        exchange["path"] = "slab"
    elif world > 1:
        exchange["path"] = "replicated (requested)"

    def barrier():
        ctx.synchronize()          # the library's stream (its RCCL calls too) is drained before torch's communicator runs
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(n, extent, steps, warmup, side=True, k=None):
        """Time `steps` SOR steps on a fresh n-splat shard; returns a dict of raw measurements."""
        k = args.k if k is None else k
        xyz_host = synth_shard(n, extent, rank)
        xyz_local = torch.from_numpy(xyz_host).to(dev)
        torch.cuda.synchronize()
        exchange.pop("certified", None)

        class _SlabRes:   # the fields the report below reads, from the slab path's device buffers
            def __init__(self, r):
                self.r = r

            @property
            def mask_local(self):
                return torch.from_numpy(slab_be.to_host(self.r["mask"], np.uint8, n))

            @property
            def stats(self):
                return torch.from_numpy(slab_be.to_host(self.r["stats"], np.float32, 3))

        def step():
            if exchange["path"] == "slab":
                try:
                    r = gslab.slab_sor(slab_be, slab_comm, gslab._View(xyz_local.data_ptr()), n, k, args.sigma)
                    if not exchange.get("certified"):   # first step on this cloud: read the certificate before relying on
                        r.check()                        # the slab path (later steps read it after the timed region)
                        exchange["certified"] = True
                    return _SlabRes(r)
                except gslab.SlabUncertain as e:   # raised on every rank together (the count is all-reduced)
                    exchange["path"] = "replicated (slab certificate failed: %s)" % e
                    compute.set_adaptive(True)     # the clouds that get here are the ones the adaptive grid exists for
            return gdist.sharded_sor(xyz_local, k, args.sigma, compute, algo=args.algo)

        if world > 1 and exchange["path"] == "slab" and not exchange.get("cross_checked"):
            # the slab exchange and the replicated one (all-gather of the rows, every rank bins everything) must agree
            # bit for bit -- statistics and this rank's survivor mask -- before the slab path is what gets timed
            ref = gdist.sharded_sor(xyz_local, k, args.sigma, compute, algo=args.algo)
            ok = 1
            try:
                got = step()
                if not isinstance(got, _SlabRes):
                    ok = 1   # the certificate already sent this cloud to the replicated path
                else:
                    same_stats = bool(np.array_equal(got.stats.numpy().view(np.uint32), ref.stats.cpu().numpy().view(np.uint32)[:3]))
                    same_mask = bool(np.array_equal(got.mask_local.numpy().astype(bool), ref.mask_local.cpu().numpy().astype(bool)))
                    ok = 1 if (same_stats and same_mask) else 0
            except gslab.SlabUncertain:
                ok = 1
            except Exception as e:   # noqa: BLE001 -- anything the exchange raises on this rank
                sys.stderr.write("[bench] rank %d: slab exchange failed its cross-check: %r\n" % (rank, e))
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                exchange["path"] = "replicated (slab exchange disagreed with the replicated exchange on this cloud)"
            exchange["cross_checked"] = True
            del ref
        for _ in range(warmup):
            res = step()
        barrier()
        # HIP events around the dominant kernel only inside the timed region: each event pair costs stream
        # time (all five slots = +0.04 ms on a 0.38 ms step, profiles/r01_timing_overhead.log)
        ctx.set_param("timing_mask", 1 << L.T_SOR_KNN)
        ctx.set_timing(True)
        ctx.reset_timing()
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            # an event pair leaves a ~10 us bubble on either side of the kernel it brackets (6 % of the 1M-splat step):
            # every fourth step of the timed region carries one, the kernel average below is over those steps
            ctx.set_timing(i % EVENT_EVERY == 0)
            res = step()
        barrier()
        dt = time.perf_counter() - t0
        ctx.set_timing(True)
        t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dt = float(t_max.item())
        if isinstance(res, _SlabRes):
            res.r.check()   # the certificate of the last timed step (same cloud every step)
        n_knn, ms_knn = ctx.timing(L.T_SOR_KNN)
        out = {"dt": dt, "knn_ms": ms_knn / max(n_knn, 1), "res": res, "xyz_host": xyz_host, "xyz_local": xyz_local}
        if side:
            # the other kernel groups: a short separate pass with every slot recording (not in the timed region)
            ctx.set_param("timing_mask", 0xff)
            ctx.reset_timing()
            side_steps = min(steps, 10)
            for _ in range(side_steps):
                step()
            barrier()
            out["side_steps"] = side_steps
            out["kernel_ms_per_step"] = {
                "knn": round(ms_knn / max(n_knn, 1), 4),
                "bin": round(ctx.timing(L.T_SOR_BIN)[1] / side_steps, 4),
                "fallback": round(ctx.timing(L.T_SOR_FALLBACK)[1] / side_steps, 4),
                "stats": round(ctx.timing(L.T_SOR_STATS)[1] / side_steps, 4),
                "note": "knn: HIP events inside the timed region (every %d-th step); the others: a separate pass of %d steps" % (EVENT_EVERY, side_steps)}
        ctx.set_timing(False)
        return out

    main_run = run(args.n, args.extent, args.steps, args.warmup)
    res = main_run["res"]
    # read the headline run's results NOW: later runs (secondary configurations) reuse the exchange's device buffers
    survivors_main = int(res.mask_local.sum().item())
    threshold_main = float(res.stats[2].item())
    mask_main_host = res.mask_local.cpu().numpy().astype(bool) if (world == 1 and not args.no_cpu_baseline) else None
    info = None
    if world == 1 and slab_comm is None:
        t = main_run["xyz_local"]
        info = ctx.sor_knn(t.data_ptr(), t.data_ptr() + 4, t.data_ptr() + 8, 3, args.n, 0, args.n, args.k,
                           res.mean_dists_local.data_ptr(), algo=args.algo, want_info=True)
    secondary = None
    if world == 1 and slab_comm is None and not args.no_secondary and args.n != 1_000_000:
        r2 = run(1_000_000, 10.0, max(args.steps, 50), max(args.warmup, 5), side=False)
        secondary = {"workload": "BASELINE.json configs[1]: 1000000 uniform-random splats (L=10, seed 0), SOR k=%d sigma=%g" % (args.k, args.sigma),
                     "value": round(1_000_000 * max(args.steps, 50) / r2["dt"] / 1e6, 2), "unit": "Msplats/s",
                     "ms_per_step": round(r2["dt"] / max(args.steps, 50) * 1e3, 4), "steps": max(args.steps, 50),
                     "knn_kernel_ms": round(r2["knn_ms"], 4),
                     "survivors": int(r2["res"].mask_local.sum().item())}
        del r2

    config3 = None
    if (world > 1 or args.config3) and not args.no_secondary:
        # BASELINE.json configs[3]: 50M splats, SOR k=32, sharded by index across the GPUs of the job (the 8-GPU case; at
        # other N the same 50M are split N ways).  Reported next to the headline, never instead of it.
        try:
            n3 = max(8192, (50_000_000 // world) // 4 * 4)
            s3, w3 = max(3, min(args.steps, 10)), 2
            r3 = run(n3, 10.0, s3, w3, side=False, k=32)
            config3 = {"workload": "BASELINE.json configs[3]: %d uniform-random splats (L=10, seed=rank) over %d GPU(s), SOR k=32 "
                                     "sigma=%g" % (n3 * world, world, args.sigma),
                         "value": round(n3 * world * s3 / r3["dt"] / 1e6, 2), "unit": "Msplats/s",
                         "ms_per_step": round(r3["dt"] / s3 * 1e3, 4), "steps": s3, "exchange": exchange["path"],
                         "knn_kernel_ms": round(r3["knn_ms"], 4)}
            del r3
        except Exception as e:   # noqa: BLE001 -- the headline line must survive a failure here
            config3 = {"workload": "BASELINE.json configs[3]", "error": repr(e)}

    if slab_comm is not None:
        ctx.check()
        slab_comm.close()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    dt = main_run["dt"]
    n_total = world * args.n
    ms_per_step = dt / args.steps * 1e3
    value = n_total * args.steps / dt / 1e6  # Msplats/s, whole job

    # ---- roofline of the dominant kernel (knn_brick, or knn_brute with --algo 1)
    knn_ms = main_run["knn_ms"]
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
    pmc = load_pmc(kernel, args.n, args.k) if world == 1 else None
    traffic = (pmc["fetch_bytes"] + pmc["write_bytes"]) if pmc and "fetch_bytes" in pmc else None
    valu = None
    if pmc and "valu_busy_frac" in pmc:
        valu = {"busy_frac": pmc["valu_busy_frac"], "insts_per_launch": pmc.get("valu_insts"),
                "kind": "static: rocprofv3 --pmc pass of this build committed under profiles/ (%s), not measured in this run" % pmc.get("source", "pmc_latest.json")}
    hbm = {"achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
           "algorithmic_bytes": int(alg_bytes), "algorithmic_bytes_per_splat": bytes_per_splat, "traffic": traffic,
           "traffic_note": ("static: rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE per launch from profiles/pmc_latest.json "
                            "(fetch as counted; the guide's x2 correction applies to wide streaming reads only)") if traffic else None,
           "note": "the figures BASELINE.json's metric asks for: algorithmic HBM bytes of one launch / its HIP-event duration vs 8 TB/s"}
    if valu and valu.get("insts_per_launch") and knn_ms > 0:
        # the binding resource: VALU issue.  One wave64 instruction occupies its SIMD16 for
        # >= 4 cycles, so the chip issues at most CUs x 4 SIMDs x clock / 4 wave-instructions per second.  The
        # instruction count of a launch is a property of (build, cloud) -- taken from the committed PMC pass and labelled
        # static -- the duration is this run's HIP-event average.
        peak_ginst = VALU_PEAK_GINST
        ach_ginst = valu["insts_per_launch"] / (knn_ms * 1e-3) / 1e9
        roofline = {"bound": "valu", "kernel": kernel, "achieved": round(ach_ginst, 2), "peak": peak_ginst,
                    "unit": "G wave-instructions/s", "frac": round(ach_ginst / peak_ginst, 4), "traffic": traffic,
                    "kernel_ms": round(knn_ms, 4),
                    "valu": dict(valu, name="VALU issue (f32 filter assembly + f64 exact distances, selection network, sqrt)",
                                 peak_note="256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction"),
                    "hbm": hbm}
    else:   # multi-GPU runs and configurations without a committed PMC pass: the HBM figures only
        roofline = dict(hbm, bound="hbm", kernel=kernel, kernel_ms=round(knn_ms, 4),
                        note="HBM figures only (no committed instruction count for this configuration); the kernel itself is "
                             "bound by VALU issue, see the N=1 line of the default configuration")

    out = {
        "metric": "Msplats/sec SOR k=%d" % args.k, "value": round(value, 2), "unit": "Msplats/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 select / f64 finalize",
        "data": "synthetic",
        "config": {"workload": "%d uniform-random splats per GPU (L=%g, seed=rank), SOR k=%d sigma=%g, "
                               "exact KNN, xyz resident in HBM" % (args.n, args.extent, args.k, args.sigma),
                   "splats_per_gpu": args.n, "k": args.k, "sigma": args.sigma,
                   "algo": "grid-binned exact KNN" if args.algo != 1 else "LDS-tiled brute force",
                   "parallelism": ((gslab.PARALLELISM if exchange["path"] == "slab" else gdist.PARALLELISM + " -- " + exchange["path"])
                                   if (world > 1 or slab_comm is not None) else "single GPU")},
        "roofline": roofline,
        "kernel_ms_per_step": main_run["kernel_ms_per_step"],
        "survivors_rank0": survivors_main,
        "threshold": threshold_main,
    }
    if info is not None:
        out["grid"] = info
    if secondary is not None:
        out["secondary"] = secondary
    if config3 is not None:
        out["config3"] = config3

    if world == 1 and not args.no_cpu_baseline:
        # reported baseline, not the target: the reference's CPU path (cKDTree + numpy, restated in
        # oracle/sor.py because the reference never returns its mask) on the SAME cloud, host cores.
        from oracle import sor as osor
        workers = max(1, (os.cpu_count() or 2) - 1)
        t0 = time.perf_counter()
        ref = osor.sor(main_run["xyz_host"], args.k, args.sigma, workers=workers)
        cpu_dt = time.perf_counter() - t0
        same = bool(np.array_equal(ref["mask"], mask_main_host))
        out["cpu_baseline"] = {"value": round(args.n / cpu_dt / 1e6, 4), "unit": "Msplats/s", "cores": workers,
                               "kind": "port", "sample": "the full %d-splat workload, once (%.2f s): "
                               "scipy cKDTree query workers=%d + numpy stats" % (args.n, cpu_dt, workers),
                               "mask_identical_to_gpu": same}
    os.write(args.json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
