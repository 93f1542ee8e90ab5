"""GPU: compressed-PLY numeric core (csrc/cply.hip) through the C ABI against the reference's own output
(tests/golden/cply_ref.npz) and the oracle."""
import hashlib
import importlib
import os

import numpy as np
import pytest

from oracle import cply as ocply

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "cply_ref.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def writer():
    return importlib.import_module("3dgsconverter_amd.formats.compressed_ply_writer")


def _cases(gold):
    for tag in gold["cases"]:
        kind, n, seed = str(tag).rsplit("_", 2)
        yield str(tag), ocply.cply_scene(int(n), int(seed), kind)


def test_morton_order_is_the_references_with_stable_ties(gsx, gold):
    for tag, d in _cases(gold):
        order, levels = gsx._lib.morton_order(d["x"], d["y"], d["z"])
        np.testing.assert_array_equal(order, gold[tag + "/order_stable"], err_msg=tag)
        if tag.startswith("clustered"):
            assert levels >= 3
        if np.array_equal(gold[tag + "/order_ref"], gold[tag + "/order_stable"]):   # no ties: the un-patched reference's order
            np.testing.assert_array_equal(order, gold[tag + "/order_ref"])


def test_chunks_and_packed_words_are_the_references_given_its_order(gsx, gold, writer):
    """the three PLY elements bit for bit: with the order the un-patched reference produced, and end to end with the
    stable order"""
    for tag, d in _cases(gold):
        for suffix, okey in (("", "/order_ref"), ("_stable", None)):
            chunk, vertex, sh, order = writer.encode(d, order=None if okey is None else gold[tag + okey])
            if okey is None:
                np.testing.assert_array_equal(order, gold[tag + "/order_stable"])
            np.testing.assert_array_equal(chunk.view(np.uint32).reshape(-1, 18), gold[tag + "/chunk" + suffix].view(np.uint32), err_msg=tag)
            np.testing.assert_array_equal(vertex.view(np.uint32).reshape(-1, 4), gold[tag + "/vertex" + suffix], err_msg=tag)
            names = [str(s) for s in gold[tag + "/sh_names"]]
            assert list(sh.dtype.names if sh is not None else []) == names
            want = str(gold[tag + ("/sh_sha256" if not suffix else "/sh_stable_sha256")])
            assert (hashlib.sha256(sh.tobytes()).hexdigest() if sh is not None else "") == want


@pytest.mark.parametrize("n", [2, 255, 257, 511, 100_003])
def test_ragged_sizes_against_the_oracle(gsx, writer, n):
    d = ocply.cply_scene(n, 40 + n % 7)
    order, _ = gsx._lib.morton_order(d["x"], d["y"], d["z"])
    want, _ = ocply.morton_order(d["x"], d["y"], d["z"])
    np.testing.assert_array_equal(order, want)
    if n <= 1000:
        chunk, vertex, sh, _ = writer.encode(d, order=order)
        names = list(sh.dtype.names)
        oc, ov, osh = ocply.encode(d, order, names)
        np.testing.assert_array_equal(chunk.view(np.uint32).reshape(-1, 18), oc.view(np.uint32))
        np.testing.assert_array_equal(vertex.view(np.uint32).reshape(-1, 4), ov)
        np.testing.assert_array_equal(sh.view(np.uint8).reshape(n, -1), osh)


def test_degenerate_geometry(gsx):
    """all points coincident (the reference returns at once, :265), a line (two axes without extent), two coincident piles"""
    L = gsx._lib
    n = 1000
    one = np.full(n, 1.5, np.float32)
    order, levels = L.morton_order(one, one, one)
    np.testing.assert_array_equal(order, np.arange(n))
    assert levels == 1   # one level looked at, nothing sorted, nothing re-activated
    rng = np.random.default_rng(0)
    t = rng.random(n).astype(np.float32)
    for x, y, z in ((t, one, one), (one, t, one), (np.where(t < 0.5, one, one * 2), one, one)):
        got, _ = L.morton_order(x, y, z)
        want, _ = ocply.morton_order(x, y, z)
        np.testing.assert_array_equal(got, want)


def test_two_million_splats_properties(gsx, writer):
    """size-independent properties at a size the oracle's chunk loop does not finish quickly: a permutation; level-0 codes
    non-decreasing; chunk bounds contain their splats; decoding the packed position lands within half a step"""
    import time
    n = 2_000_000
    d = ocply.cply_scene(n, 9, "clustered")
    t0 = time.perf_counter()
    chunk, vertex, sh, order = writer.encode(d)
    dt = time.perf_counter() - t0
    assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32))
    codes = ocply.morton_codes(d["x"], d["y"], d["z"])[order]
    assert np.all(codes[1:] >= codes[:-1])
    want, _ = ocply.morton_order(d["x"], d["y"], d["z"])
    np.testing.assert_array_equal(order, want)
    sd = d[order]
    nc = len(chunk)
    pad = nc * 256 - n
    for ax, bits, shift in (("x", 11, 21), ("y", 10, 11), ("z", 11, 0)):
        v = np.concatenate([sd[ax], np.full(pad, sd[ax][-1])]).reshape(nc, 256)
        np.testing.assert_array_equal(chunk["min_" + ax], v.min(1))
        np.testing.assert_array_equal(chunk["max_" + ax], v.max(1))
        t = (1 << bits) - 1
        q = ((vertex["packed_position"] >> shift) & t).astype(np.float64)
        lo = np.repeat(chunk["min_" + ax], 256)[:n].astype(np.float64)
        hi = np.repeat(chunk["max_" + ax], 256)[:n].astype(np.float64)
        rng_ = hi - lo
        ok = rng_ >= 1e-5
        err = np.abs(lo + q / t * rng_ - sd[ax])[ok]
        assert np.all(err <= 0.5 * rng_[ok] / t * 1.001 + 1e-6)
    assert sh.dtype.names[0] == "f_rest_0" and len(sh) == n
    print("compressed-PLY encode of %d splats: %.1f ms (upload + Morton + pack + SH + download)" % (n, dt * 1e3))


def test_write_compressed_ply_file(gsx, writer, tmp_path, gold):
    tag, d = next(c for c in _cases(gold) if c[0].startswith("degree1"))
    path = str(tmp_path / "out.compressed.ply")
    writer.write_compressed_ply(d, path)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert b"element chunk 4" in head and b"element vertex 777" in head and b"element sh 777" in head
    assert head.count(b"property uchar f_rest_") == 9
    v = np.frombuffer(body[4 * 72:4 * 72 + 777 * 16], dtype=np.uint32).reshape(-1, 4)
    np.testing.assert_array_equal(v, gold[tag + "/vertex_stable"])


def test_pack_table_from_device_resident_rows_equals_the_host_gather_path(gsx):
    """round 6: the writer uploads the raw rows once and cuts its 59 columns out of them on the device
    (gsx_rows_to_columns_dev); tables it does not take, and calls with a caller's context, gather on the host as in round 5.
    Same kernels downstream: every output identical -- also for 251-byte rows (host path) and against the oracle."""
    lib = gsx._lib
    from oracle import cply as ocply
    for n, seed, kind in ((5000, 1, "clustered"), (70001, 2, "plain"), (12000, 3, "clustered")):
        scene = ocply.cply_scene(n, seed, kind)
        # opacities ON the rounding boundaries of the alpha byte (floor(a * 255 + 0.5) steps at a = (j + 0.5) / 255), far tails, zeros:
        # the device decides the byte from a float64 exp + certificate and lists what it cannot decide for numpy
        aa = (np.arange(0, 255, dtype=np.float64) + 0.5) / 255.0
        edge = np.log(aa / (1 - aa)).astype(np.float32)
        scene["opacity"][:len(edge)] = edge
        scene["opacity"][len(edge):len(edge) + 8] = [0.0, -0.0, 90.0, -90.0, 17.0, -17.0, 1e-8, -1e-8]
        sh_names = [nm for nm in scene.dtype.names if nm.startswith("f_rest_")]
        a = lib.cply_pack_table(scene, sh_names)                      # device-resident rows (n >= 1024, float32 fields)
        ctx = lib.Context(0)
        b = lib.cply_pack_table(scene, sh_names, ctx=ctx)             # a caller's context: the host gather path
        ctx.close()
        for u, v in zip(a[:4], b[:4]):
            if u is None:
                assert v is None
            else:
                np.testing.assert_array_equal(np.asarray(u).view(np.uint8), np.asarray(v).view(np.uint8))
        want_order, _ = ocply.morton_order(scene["x"], scene["y"], scene["z"])
        oc, ov, osh = ocply.encode(scene, want_order, sh_names)
        np.testing.assert_array_equal(a[3], want_order)
        np.testing.assert_array_equal(a[1], ov)
        np.testing.assert_array_equal(a[0].view(np.uint32), oc.view(np.uint32))
        if sh_names:
            np.testing.assert_array_equal(a[2], osh)
    lib.release_arenas()


def test_fields_nonzero_is_numpys_any_not_equal_zero(gsx):
    """gsx_fields_nonzero_dev against `np.any(data[f] != 0)` (compressed_ply.py:139-150): NaN counts, -0.0 does not; row counts
    around the eight-row groups of the kernel; the only non-zero in the first / last row; m up to 64 fields"""
    import ctypes as C
    lib = gsx._lib
    ctx = lib.Context(0)
    rng = np.random.default_rng(3)
    try:
        for n, m, stride in [(1, 1, 1), (7, 3, 5), (8, 45, 62), (9, 45, 62), (1000, 64, 64), (100_003, 45, 62), (300_000, 24, 30)]:
            for trial in range(4):
                rows = np.zeros((n, stride), np.float32)
                want = 0
                cols = rng.choice(m, size=int(rng.integers(0, min(m, 6) + 1)), replace=False)
                for c in cols:
                    r = [0, n - 1, int(rng.integers(0, n))][int(rng.integers(0, 3))]
                    rows[r, c] = [np.float32(np.nan), np.float32(1e-45), np.float32(-3.0)][int(rng.integers(0, 3))]
                    want |= 1 << int(c)
                rows[rng.integers(0, n, 5), rng.integers(0, m, 5)] += np.float32(-0.0)      # (-0.0 into zeros stays "zero"; into others no change)
                if stride > m:
                    rows[:, m:] = 7.0                                                       # fields beyond m are not looked at
                d = ctx.alloc(rows.nbytes).upload(rows)
                word = C.c_uint64(123)
                lib.check(lib.require_hip().gsx_fields_nonzero_dev(ctx.handle, d.ptr, stride, n, m, C.byref(word)), "gsx_fields_nonzero_dev")
                d.free()
                ref = sum(1 << c for c in range(m) if np.any(rows[:, c] != 0))
                assert word.value == ref == want, (n, m, stride, trial, hex(word.value), hex(ref))
    finally:
        ctx.close()


@pytest.mark.parametrize("last", [-1, 0, 8, 9, 23, 24, 44])
def test_degree_detection_on_resident_rows_is_the_host_scan(gsx, writer, last):
    """the writer's SH degree (compressed_ply.py:139-171) from the device's pass over the uploaded rows: tables whose coefficients
    are zero above index `last` (with -0.0 among the zeros), resident path against the reference's per-column scan"""
    d = ocply.cply_scene(5000, 17 + last, "plain")
    for i in range(last + 1, 45):
        d["f_rest_%d" % i] = np.float32(-0.0) if i % 3 == 0 else np.float32(0.0)
    if last >= 0:
        d["f_rest_%d" % last] = 0
        d["f_rest_%d" % last][4999] = np.float32(1e-30)       # one non-zero, in the last row
    want_names = writer.active_sh_names(d)
    assert len(want_names) == {-1: 0, 0: 9, 8: 9, 9: 24, 23: 24, 24: 45, 44: 45}[last]
    chunk, vertex, sh, order = writer.encode(d)
    assert (list(sh.dtype.names) if sh is not None else []) == want_names
    wc, wv, wsh = ocply.encode(d, order, want_names)
    np.testing.assert_array_equal(vertex.view(np.uint32).reshape(-1, 4), wv)
    if want_names:
        np.testing.assert_array_equal(sh.view(np.uint8).reshape(len(d), -1), wsh)


def test_rows_repack_and_the_writer_on_a_table_widened_by_colour_fields(gsx, writer):
    """round 6: gsx_rows_repack_dev (rows of any size -> rows padded to a multiple of 4, on the device) against numpy, and through it the
    compressed-PLY writer on the 251-byte rows add_rgb_from_sh leaves (data_processor.py:262-274; `--rgb` with this target):
    device-resident like the plain table, same elements"""
    lib = gsx._lib
    ctx = lib.Context(0)
    rng = np.random.default_rng(4)
    try:
        for rb, n in [(251, 1), (251, 1000), (249, 4097), (7, 333), (1, 17), (250, 100_003), (248, 50)]:
            raw = rng.integers(0, 256, (n, rb), dtype=np.uint8)
            pitch = (rb + 3) & ~3
            src = ctx.alloc(raw.nbytes + 16).upload(raw)
            dst = ctx.alloc(n * pitch)
            lib.check(lib.require_hip().gsx_rows_repack_dev(ctx.handle, src.ptr, rb, n, dst.ptr, pitch), "gsx_rows_repack_dev")
            got = dst.download(np.uint8, n * pitch).reshape(n, pitch)
            np.testing.assert_array_equal(got[:, :rb], raw)
            assert not got[:, rb:].any()
            src.free(), dst.free()
    finally:
        ctx.close()
    for last in (44, 8):
        d = ocply.cply_scene(30_011, 23, "clustered")
        for i in range(last + 1, 45):
            d["f_rest_%d" % i] = 0
        wide = lib.host_append_u8_columns(d, ("red", "green", "blue"), rng.integers(0, 256, (len(d), 3), dtype=np.uint8))
        assert wide.dtype.itemsize == 251
        stages = {}
        names = writer.active_sh_names(wide)
        lib.cply_pack_table(wide, names, None, None, stages)
        assert "repack" in stages and "upload" in stages          # the resident path, not the host gather
        a, b = writer.encode(wide), writer.encode(d)
        for x, y in zip(a[:3], b[:3]):
            assert (x is None and y is None) or x.tobytes() == y.tobytes()
        np.testing.assert_array_equal(a[3], b[3])
