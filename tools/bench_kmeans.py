"""bench.py --workload kmeans: BASELINE.json configs[4] -- the SOG SH-palette K-Means of a 10M-splat scene
(compression_level 2: 64 chunks x (156 250 x 45 f32), K = 1024 per chunk, 10 Lloyd iterations; formats/sog.py:513-552
calling gpu_ops.kmeans, reference kernels gpu_ops.py:57-96).

One step = all chunks of the scene, SH rows already resident in HBM (the reference re-uploads every chunk twice per
iteration); initial centroids are random rows (gpu_ops.py:182) restored before every step.  With N GPUs
the chunks are dealt out round robin (no collective): strong scaling, value = splats of the whole
scene / step time.  Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak (~2.5 PF)


def main(args):
    """no torch: ranks from 3dgsconverter_amd/launch.py (own ranks when run plainly with --gpus N), barrier and the MAX of the
    ranks' clocks through the C library's communicator"""
    launch = importlib.import_module("3dgsconverter_amd.launch")
    rank, local_rank, world = launch.rank_env()
    if world == 1 and args.gpus > 1:
        sys.exit(launch.spawn_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not hasattr(args, "json_fd"):
        sys.stdout.flush()
        args.json_fd = os.dup(1)
        os.dup2(2, 1)

    gsx = importlib.import_module("3dgsconverter_amd")
    pal = importlib.import_module("3dgsconverter_amd.dist_palette")
    gslab = importlib.import_module("3dgsconverter_amd.dist_slab")
    L = gsx._lib
    device, transport = launch.pick_device_and_transport(local_rank, world, L.device_count())
    ctx = L.Context(device)
    comm = None
    if world > 1:
        transport = launch.agree_transport(rank, world, L.device_uid(device))
        comm = gslab.RcclComm(ctx, rank, world, launch.exchange_unique_id(rank, lambda: gslab.RcclComm.unique_id(transport)))
        comm.barrier()
        launch.retire_unique_id(rank)

    n_scene = args.n if args.n != 10_000_000 or True else args.n
    d, iters, level = 45, 10, 2
    plan = pal.palette_plan(n_scene, level)
    nch, cs, k = plan["num_chunks"], plan["chunk_size"], plan["k_per_chunk"]
    mine = [i for i in range(nch) if i % world == rank]
    steps = args.steps if args.steps != 30 else 5
    warmup = min(args.warmup, 2)

    # synthetic SH rows (f_rest ~ N(0, 0.1^2) f32 from numpy's seed-0 generator, initial centroids
    # drawn like the reference's front door: np.random.seed(0), one np.random.choice per chunk, gpu_ops.py:182) -- the same
    # stream as bench.py's config4, walked by every rank so that a chunk's data does not depend on the number of GPUs
    rng = np.random.default_rng(0)
    np.random.seed(0)
    chunks, inits, first_host = [], [], None
    for i in range(nch):
        rows = min(cs, n_scene - i * cs)
        x = rng.standard_normal((rows, d), dtype=np.float32) * np.float32(0.1)
        pick = np.random.choice(rows, k, replace=False)
        if i in mine:
            if not chunks:
                first_host = x
            chunks.append((ctx.alloc(x.nbytes).upload(x), rows))
            inits.append(ctx.alloc(4 * k * d).upload(np.ascontiguousarray(x[pick])))
    cents = [ctx.alloc(4 * k * d) for _ in inits]
    labels = [ctx.alloc(4 * r + 16) for _, r in chunks]
    ctx.synchronize()

    def barrier():
        if comm is not None:
            comm.barrier()
        else:
            ctx.synchronize()

    def step():
        for j in range(len(mine)):
            L.check(ctx.lib.gsx_dev_copy(ctx.handle, cents[j].ptr, inits[j].ptr, 4 * k * d), "gsx_dev_copy")
            L.check(ctx.lib.gsx_kmeans_lloyd_dev(ctx.handle, chunks[j][0].ptr, chunks[j][1], d, k, iters, cents[j].ptr, labels[j].ptr),
                    "gsx_kmeans_lloyd_dev")

    for kv in getattr(args, "param", []) or []:
        name, val = kv.split("=")
        ctx.set_param(name, float(val))
    for _ in range(warmup):
        step()
    barrier()
    ctx.set_param("timing_mask", (1 << L.T_KMEANS_ASSIGN) | (1 << L.T_KMEANS_UPDATE))
    ctx.set_timing(True)
    ctx.reset_timing()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if comm is not None:
        dt = float(comm.reduce_scalar(dt, gslab.KIND_F64_MAX))
    n_as, ms_as = ctx.timing(L.T_KMEANS_ASSIGN)
    n_up, ms_up = ctx.timing(L.T_KMEANS_UPDATE)
    ctx.set_timing(False)
    if rank != 0:
        comm.barrier()
        comm.close()
        return

    ms_step = dt / steps * 1e3
    value = n_scene * steps / dt / 1e6
    assign_ms = ms_as / max(n_as, 1)            # one timing interval = operand prep + matrix-core assign + exact list, one chunk iteration
    rows0 = chunks[0][1]
    ktiles, ns = (k + 31) // 32, 3              # 32-centroid tiles, three 16-wide slices of the 45 (+3) dimensions
    # three v_mfma_f32_32x32x16_bf16 per slice (xh.ch + xh.cl + xl.ch), 2*32*32*16 flops each, per (32 points x 32 centroids)
    flops = -(-rows0 // 32) * ktiles * ns * 3 * 2 * 32 * 32 * 16
    achieved = flops / (assign_ms * 1e-3) / 1e12 if assign_ms > 0 else 0.0
    alg_bytes = rows0 * (4 * d + 4)             # read the rows once, write one label
    out = {
        "metric": "Msplats/sec SOG SH-palette K-Means (64 chunks x K=1024 x 10 iterations)", "value": round(value, 2),
        "unit": "Msplats/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_step, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16-split filter / f32 exact certificate (f64 cluster sums)",
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[4]: %d splats, degree-3 SH rows (45 f32), compression_level %d -> %d chunks "
                               "of %d rows, K=%d per chunk, %d Lloyd iterations, rows resident in HBM" % (n_scene, level, nch, cs, k, iters),
                   "splats": n_scene, "chunks": nch, "k_per_chunk": k, "iterations": iters,
                   "parallelism": "chunks dealt out round robin, no collective" if world > 1 else "single GPU"},
        "roofline": {"bound": "mfma", "kernel": "kmeans_assign_mfma_cs_kernel<45> (+ operand prep and exact list kernel in the same interval)",
                     "achieved": round(achieved, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": None, "kernel_ms": round(assign_ms, 4),
                     "flops_per_launch": flops, "algorithmic_bytes": alg_bytes,
                     "hbm_frac_of_8TBs": round(alg_bytes / (assign_ms * 1e-3) / 8e12, 5) if assign_ms > 0 else None,
                     "note": "bf16 matrix-core flops actually issued (a FILTER: 9 MFMAs per 32x32 tile incl. the two cross terms of the "
                             "bf16 split); labels are certified exact, the uncertified ~0.3 % are rescanned in f32"},
        "kernel_ms_per_step": {"assign (operands + mfma + exact list)": round(ms_as / steps, 3),
                               "update (label sort + segmented reduce)": round(ms_up / steps, 3)},
    }
    if world == 1 and not args.no_cpu_baseline:
        # the reference's CPU path for the same call (gpu_ops.py:48-52: MiniBatchKMeans, batch 16384, n_init auto), ONE chunk
        from sklearn.cluster import MiniBatchKMeans
        x = first_host
        t0 = time.perf_counter()
        km = MiniBatchKMeans(n_clusters=k, max_iter=iters, batch_size=min(4096 * 4, len(x)), n_init="auto", compute_labels=True)
        km.fit(x)
        cpu_dt = time.perf_counter() - t0
        from oracle import kmeans as okm
        i_cpu = okm.inertia(x, km.cluster_centers_, km.labels_)
        i_gpu = okm.inertia(x, cents[0].download(np.float32, k * d).reshape(k, d), labels[0].download(np.int32, rows0))
        out["cpu_baseline"] = {"value": round(rows0 / cpu_dt / 1e6, 4), "unit": "Msplats/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": "one of the %d chunks (%d x %d, K=%d, max_iter=%d), once (%.2f s): sklearn MiniBatchKMeans "
                                         "called as the reference's _kmeans_sklearn does" % (nch, rows0, d, k, iters, cpu_dt),
                               "inertia_cpu": round(i_cpu, 2), "inertia_gpu_same_chunk": round(i_gpu, 2)}
    os.write(args.json_fd, (json.dumps(out) + "\n").encode())   # bench.py pointed fd 1 at stderr; this is the real stdout
    if comm is not None:
        comm.barrier()
        comm.close()
