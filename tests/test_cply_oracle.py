"""CPU: the compressed-PLY oracle (oracle/cply.py) against what the reference's own writer produced
(tests/golden/cply_ref.npz, oracle/make_golden_cply.py), and the PLY container written without plyfile."""
import hashlib
import os

import numpy as np
import pytest

from oracle import cply as ocply, refload

GOLD = os.path.join(os.path.dirname(__file__), "golden", "cply_ref.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def cases(gold):
    for tag in gold["cases"]:
        kind, n, seed = str(tag).rsplit("_", 2)
        yield str(tag), ocply.cply_scene(int(n), int(seed), kind)


def test_oracle_morton_order_is_the_references_with_stable_ties(gold):
    seen_ties = False
    for tag, d in cases(gold):
        order, depth = ocply.morton_order(d["x"], d["y"], d["z"])
        np.testing.assert_array_equal(order, gold[tag + "/order_stable"])
        ref = gold[tag + "/order_ref"]
        assert sorted(ref.tolist()) == list(range(len(d)))
        if not np.array_equal(ref, order):
            seen_ties = True
            # the un-patched reference differs only inside runs of coincident-code splats: same multiset per chunk-sized window
            # is too strong, so check the weaker invariant that the two orders visit the same points (as coordinates)
            # wherever the stable order has no tie
            same = ref == order
            pts = np.column_stack([d["x"], d["y"], d["z"]])
            assert same.mean() > 0.5 and np.array_equal(np.sort(pts[ref], axis=0), np.sort(pts[order], axis=0))
    assert seen_ties   # the clustered case must exercise the tie path


def test_oracle_encode_is_the_references(gold):
    for tag, d in cases(gold):
        sh_names = [str(s) for s in gold[tag + "/sh_names"]]
        for suffix, okey in (("", "/order_ref"), ("_stable", "/order_stable")):
            chunks, verts, sh = ocply.encode(d, gold[tag + okey], sh_names)
            np.testing.assert_array_equal(chunks.view(np.uint32), gold[tag + "/chunk" + suffix].view(np.uint32))
            np.testing.assert_array_equal(verts, gold[tag + "/vertex" + suffix])
            want = str(gold[tag + ("/sh_sha256" if not suffix else "/sh_stable_sha256")])
            assert (hashlib.sha256(sh.tobytes()).hexdigest() if sh is not None else "") == want


@pytest.mark.skipif(not refload.available(), reason="reference checkout not present")
def test_fixture_is_reproducible_from_the_reference(gold):
    tag, d = next(c for c in cases(gold) if c[0].startswith("clustered"))
    live = refload.reference_cply(d, stable_ties=True)
    np.testing.assert_array_equal(live["order"], gold[tag + "/order_stable"])
    np.testing.assert_array_equal(live["vertex"].view(np.uint32).reshape(-1, 4), gold[tag + "/vertex_stable"])


def test_ply_container_layout(tmp_path):
    import importlib
    w = importlib.import_module("3dgsconverter_amd.formats.compressed_ply_writer")
    chunk = np.zeros(2, dtype=w.CHUNK_DTYPE)
    chunk["max_b"] = [1.5, 2.5]
    vert = np.zeros(300, dtype=w.VERTEX_DTYPE)
    vert["packed_color"] = np.arange(300)
    sh = np.zeros(300, dtype=[("f_rest_0", "u1"), ("f_rest_1", "u1")])
    sh["f_rest_1"] = np.arange(300) % 251
    path = str(tmp_path / "t.ply")
    w._write_ply(path, [("chunk", chunk), ("vertex", vert), ("sh", sh)])
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().strip().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element chunk 2"]
    assert lines[3] == "property float min_x" and "element vertex 300" in lines and "property uint packed_position" in lines
    assert lines[-3:] == ["element sh 300", "property uchar f_rest_0", "property uchar f_rest_1"]
    assert len(body) == 2 * 72 + 300 * 16 + 300 * 2
    np.testing.assert_array_equal(np.frombuffer(body[:144], dtype=w.CHUNK_DTYPE), chunk)
    np.testing.assert_array_equal(np.frombuffer(body[144:144 + 4800], dtype=w.VERTEX_DTYPE), vert)
    np.testing.assert_array_equal(np.frombuffer(body[144 + 4800:], dtype=sh.dtype), sh)
