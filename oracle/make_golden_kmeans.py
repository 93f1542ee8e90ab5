"""K-Means / SOG golden vectors produced by RUNNING THE REFERENCE'S OWN CODE in the build container.

    python -m oracle.make_golden_kmeans          # needs /root/reference; ~1 min

TEST INFRASTRUCTURE ONLY.  Three kinds of fixtures -> tests/golden/kmeans_ref.npz + kmeans_ref.json:

1. ``lloyd/*``: the reference's ``_kmeans_taichi`` + ``k_means_assign`` + ``k_means_update``
   (gpu_ops.py:57-96,178-191) executed through oracle/taichi_shim.py (Taichi is not installable
   here; the shim runs the kernel bodies as Python with Taichi's default f32/i32 types).
   ``np.random.seed`` pins the reference's unseeded ``np.random.choice`` (gpu_ops.py:182); the drawn
   initial centroids are stored so that the GPU test can inject them.
2. ``sklearn/*``: the reference's CPU front door ``gpu_ops.kmeans(..., use_gpu=False)`` ->
   ``MiniBatchKMeans`` (gpu_ops.py:48-52) under ``np.random.seed`` (``random_state=None`` draws from
   numpy's global RNG): inertia fixtures for the quality bar of SURVEY.md 8(c).
3. ``sog/*``: the reference's ``SogFormat.write`` (formats/sog.py:249-639) on a synthetic degree-3
   scene, with a spy on ``gpu_ops.kmeans`` (the chunk plan of :513-552) and the written bundle
   decoded with pillow: the u8 codebook indices of ``quantize_to_codebook`` (:408-419) against the
   codebooks in meta.json, the position / quaternion / opacity textures of :279-386,457-459.
"""
from __future__ import annotations

import io
import json
import os
import sys
import tempfile
import zipfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import refload, kmeans as okm  # noqa: E402
from oracle.datasets import km_data, sog_scene  # noqa: E402

OUT_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")


LLOYD_CASES = [
    # name, data spec, K, max_iter, np.random.seed
    ("lloyd_1d_k32", {"kind": "normal", "n": 3000, "d": 1, "mu": -4.0, "sigma": 1.0, "seed": 1}, 32, 20, 11),
    ("lloyd_9d_k48", {"kind": "normal", "n": 2000, "d": 9, "sigma": 0.1, "seed": 2}, 48, 10, 12),
    ("lloyd_24d_k32", {"kind": "normal", "n": 1200, "d": 24, "sigma": 0.1, "seed": 3}, 32, 10, 13),
    ("lloyd_45d_k64", {"kind": "normal", "n": 1500, "d": 45, "sigma": 0.1, "seed": 4}, 64, 10, 14),
    ("lloyd_empty_clusters", {"kind": "repeated", "n": 1200, "d": 3, "rep": 3, "seed": 5}, 300, 6, 15),
    ("lloyd_ties_lattice", {"kind": "lattice2d", "m": 4, "rep": 20, "seed": 6}, 6, 8, 16),
    ("lloyd_3d_k1024", {"kind": "normal", "n": 4000, "d": 3, "sigma": 1.0, "seed": 7}, 1024, 4, 17),
    # the SOG scalar-codebook shape (sog.py:396-403) scaled down: 1-D, K=256, 20 iterations, Gaussian tails
    ("lloyd_1d_k256_tails", {"kind": "normal", "n": 8000, "d": 1, "mu": -4.0, "sigma": 1.0, "seed": 8}, 256, 20, 18),
]

SKLEARN_CASES = [
    ("sk_1d_50k_k256", {"kind": "normal", "n": 50000, "d": 1, "mu": -4.0, "sigma": 1.0, "seed": 21}, 256, 20, 31),
    ("sk_45d_20k_k256", {"kind": "normal", "n": 20000, "d": 45, "sigma": 0.1, "seed": 22}, 256, 10, 32),
    ("sk_9d_8k_k128", {"kind": "normal", "n": 8000, "d": 9, "sigma": 0.1, "seed": 23}, 128, 10, 33),
]


def decode_webp(zf, name):
    from PIL import Image
    img = Image.open(io.BytesIO(zf.read(name))).convert("RGBA")
    return np.asarray(img, dtype=np.uint8).reshape(-1, 4)


def run_reference_sog(n, seed, level, np_seed):
    refload.load()
    import gsconverter.formats.sog as sogmod
    calls = []
    real = sogmod.gpu_ops.kmeans

    def spy(data, k, max_iter=10, **kw):
        c, l = real(data, k, max_iter=max_iter, **kw)
        calls.append({"n": int(data.shape[0]), "d": int(data.shape[1]), "k": int(k), "max_iter": int(max_iter),
                      "centroids_shape": list(c.shape)})
        return c, l

    saved_status = sogmod.status_print
    sogmod.status_print = lambda *a, **k: None
    sogmod.gpu_ops.kmeans = spy
    data = sog_scene(n, seed)
    from gsconverter.structures import GaussianStruct
    assert data.dtype == np.dtype(GaussianStruct.define_dtype(sh_degree=3)[0]), "oracle.datasets.splat_dtype drifted from structures.py"
    try:
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "out.sog")
            np.random.seed(np_seed)
            sogmod.SogFormat().write(data, path, compression_level=level)
            with zipfile.ZipFile(path) as zf:
                meta = json.loads(zf.read("meta.json"))
                tex = {name[:-5]: decode_webp(zf, name) for name in zf.namelist() if name.endswith(".webp")}
    finally:
        sogmod.gpu_ops.kmeans = real
        sogmod.status_print = saved_status
    return data, meta, tex, calls


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    arrays, cases = {}, {"lloyd": {}, "sklearn": {}, "sog": {}, "front_door": {}}
    shim = refload.load_gpu_ops_with_taichi_shim()

    for name, spec, k, iters, seed in LLOYD_CASES:
        data = km_data(spec)
        n = len(data)
        np.random.seed(seed)
        init_idx = np.random.choice(n, k, replace=False)   # what gpu_ops.py:182 will draw
        np.random.seed(seed)
        cent, labels = shim._kmeans_taichi(data, k, iters)
        # the C restatement (binary32, index order) must reproduce the reference's own code bit for bit
        c2, l2, _ = okm.lloyd(data, data[init_idx], iters, accumulate="f32seq")
        assert np.array_equal(labels, l2) and np.array_equal(cent.view(np.uint32), c2.view(np.uint32)), name
        counts = np.bincount(labels, minlength=k)
        arrays[name + "__init"] = data[init_idx]
        arrays[name + "__cent"] = cent
        arrays[name + "__labels"] = labels.astype(np.int32)
        cases["lloyd"][name] = {"data": spec, "k": k, "max_iter": iters, "np_seed": seed, "n": n, "d": int(data.shape[1]),
                                "inertia": okm.inertia(data, cent, labels),
                                "empty_clusters_last_assign": int((counts == 0).sum()),
                                "zero_centroids": int((np.abs(cent).sum(1) == 0).sum()),
                                "source": "reference gpu_ops._kmeans_taichi via oracle/taichi_shim.py"}
        print(name, cases["lloyd"][name])

    # the reference's two paths differ in quality: its GPU path is random-sample init + plain Lloyd, its CPU
    # path MiniBatchKMeans with k-means++ init -- record both on the same 1-D data
    _, gpu_ops_cpu, _ = refload.load()
    spec = cases["lloyd"]["lloyd_1d_k256_tails"]
    np.random.seed(28)
    c_sk, l_sk = gpu_ops_cpu.kmeans(km_data(spec["data"]), spec["k"], max_iter=spec["max_iter"], use_gpu=False)
    spec["inertia_of_the_reference_sklearn_path_on_the_same_data"] = okm.inertia(km_data(spec["data"]), c_sk, l_sk)

    # front door: k >= N shortcut (gpu_ops.py:30-31) and the Taichi branch of kmeans() itself
    small = km_data({"kind": "normal", "n": 10, "d": 3, "sigma": 1.0, "seed": 9})
    c, l = shim.kmeans(small, 16)
    assert np.array_equal(c, small) and np.array_equal(l, np.arange(10)) and l.dtype == np.int32
    cases["front_door"]["k_ge_n"] = {"n": 10, "k": 16, "returns": "(data.copy(), arange(N, int32))", "centroid_dtype": str(c.dtype)}
    spec = LLOYD_CASES[1][1]
    np.random.seed(12)
    c, l = shim.kmeans(km_data(spec), 48, max_iter=10)
    assert np.array_equal(c, arrays["lloyd_9d_k48__cent"]) and np.array_equal(l, arrays["lloyd_9d_k48__labels"])
    cases["front_door"]["taichi_branch"] = "kmeans(data, 48, max_iter=10) == _kmeans_taichi fixture lloyd_9d_k48"

    _, gpu_ops, _ = refload.load()
    for name, spec, k, iters, seed in SKLEARN_CASES:
        data = km_data(spec)
        np.random.seed(seed)
        cent, labels = gpu_ops.kmeans(data, k, max_iter=iters, use_gpu=False)
        cases["sklearn"][name] = {"data": spec, "k": k, "max_iter": iters, "np_seed": seed,
                                  "inertia": okm.inertia(data, cent, labels),
                                  "source": "reference gpu_ops.kmeans(use_gpu=False) -> MiniBatchKMeans"}
        print(name, cases["sklearn"][name])

    # SogFormat.write end to end
    for name, n, seed, level, np_seed in (("sog_20k_l2", 20000, 41, 2, 51), ("sog_3k_l8", 3000, 42, 8, 52)):
        data, meta, tex, calls = run_reference_sog(n, seed, level, np_seed)
        order = np.lexsort((data["z"], data["y"], data["x"]))   # formats/sog.py:264
        cases["sog"][name] = {"n": n, "scene_seed": seed, "compression_level": level, "np_seed": np_seed,
                              "kmeans_calls": calls, "meta": meta,
                              "plan": okm.sog_sh_plan(n, level)}
        arrays[name + "__order_sha"] = np.frombuffer(bytes.fromhex(__import__("hashlib").sha256(order.tobytes()).hexdigest()), dtype=np.uint8)
        for t in ("scales", "sh0", "means_l", "means_u", "quats", "shN_labels"):
            arrays[name + "__" + t] = tex[t][:n].copy()
        arrays[name + "__shN_centroids"] = tex["shN_centroids"]
        print(name, "kmeans calls", len(calls), calls[:3], "palette", meta["shN"]["count"])

    import numpy, sklearn
    cases["_meta"] = {"numpy": numpy.__version__, "sklearn": sklearn.__version__,
                      "generator": "oracle/make_golden_kmeans.py run against /root/reference (v0.8)"}
    with open(os.path.join(OUT_DIR, "kmeans_ref.json"), "w") as f:
        json.dump(cases, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT_DIR, "kmeans_ref.npz"), **arrays)
    print("wrote kmeans_ref.json / kmeans_ref.npz")


if __name__ == "__main__":
    main()
