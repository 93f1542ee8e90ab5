"""CPU: the host half of the device-resident SOG core (formats/sog_device.py) -- which tables it takes and how it describes
them to the C ABI (gsx_sog_layout), the band detection of formats/sog.py:461-493 from the device's non-zero mask, the texture
geometry, and the random draws' reproducibility under np.random.seed."""
import importlib

import numpy as np
import pytest

from oracle import datasets

sd = importlib.import_module("3dgsconverter_amd.formats.sog_device")
lib = importlib.import_module("3dgsconverter_amd._lib")


def _reference_bands(data):
    """formats/sog.py:461-493 restated (the loop the device mask replaces)"""
    sh_bands = 0
    if "f_rest_0" in data.dtype.names:
        count_sh = sum(1 for i in range(45) if f"f_rest_{i}" in data.dtype.names)
        if count_sh >= 45:
            sh_bands = 3
        elif count_sh >= 24:
            sh_bands = 2
        elif count_sh >= 9:
            sh_bands = 1
        if sh_bands > 0:
            last_active_idx = -1
            for i in range({3: 44, 2: 23, 1: 8}[sh_bands], -1, -1):
                fn = f"f_rest_{i}"
                if fn in data.dtype.names and np.any(data[fn] != 0):
                    last_active_idx = i
                    break
            sh_bands = 3 if last_active_idx >= 24 else 2 if last_active_idx >= 9 else 1 if last_active_idx >= 0 else 0
    return sh_bands


def _mask_of(data):
    m = 0
    for i in range(45):
        if "f_rest_%d" % i in data.dtype.names and np.any(data["f_rest_%d" % i] != 0):
            m |= 1 << i
    return m


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
def test_band_detection_from_the_nonzero_mask(degree):
    rng = np.random.default_rng(degree)
    for trial in range(12):
        data = datasets.sog_scene(64, 100 + trial, sh_degree=degree)
        ncoef = 3 * ((degree + 1) ** 2 - 1)
        for i in range(ncoef):     # zero a random tail (and sometimes everything, sometimes a hole in the middle)
            if rng.random() < 0.5:
                data["f_rest_%d" % i] = 0.0
        if trial % 3 == 0 and ncoef:
            for i in range(int(rng.integers(0, ncoef)), ncoef):
                data["f_rest_%d" % i] = np.float32(-0.0) if trial % 2 else 0.0
        got = sd.bands_from_mask(sd.sh_coeffs_present(data.dtype.names), _mask_of(data))
        assert got == _reference_bands(data)


def test_layout_of_the_standard_table_is_its_dtype():
    data = datasets.sog_scene(10, 1)
    rows, lay = sd.table_layout(data)
    assert rows is data and lay.row_bytes == data.dtype.itemsize == 248 and lay.n_rest == 45
    for i, nm in enumerate(lib.SOG_FIELD_NAMES):
        assert lay.offset[i] == data.dtype.fields[nm][1]
        # the value the device reads at that offset is the field
        raw = data.view(np.uint8).reshape(10, 248)
        np.testing.assert_array_equal(raw[:, lay.offset[i]:lay.offset[i] + 4].copy().view(np.float32).reshape(-1), data[nm])


def test_layout_packs_odd_rows_and_declines_foreign_dtypes():
    src = datasets.sog_scene(100, 2)
    wide = np.zeros(100, dtype=np.dtype(src.dtype.descr + [("red", "u1"), ("green", "u1"), ("blue", "u1")]))
    for nm in src.dtype.names:
        wide[nm] = src[nm]
    # 251-byte rows (what the reference's converter hands the SOG writer after add_rgb_from_sh) are described as they are
    rows, lay = sd.table_layout(wide)
    assert rows is wide and lay.row_bytes == 251
    raw = wide.view(np.uint8).reshape(100, 251)
    for i, nm in enumerate(lib.SOG_FIELD_NAMES):
        assert lay.offset[i] == wide.dtype.fields[nm][1]
        np.testing.assert_array_equal(raw[:, lay.offset[i]:lay.offset[i] + 4].copy().view(np.float32).reshape(-1), src[nm])
    # a u1 field in FRONT of the floats: every offset off the 4-byte grid, still direct
    front = np.zeros(100, dtype=np.dtype([("tag", "u1")] + src.dtype.descr))
    for nm in src.dtype.names:
        front[nm] = src[nm]
    rows, lay = sd.table_layout(front)
    assert rows is front and lay.row_bytes == 249 and lay.offset[0] == 1
    # rows beyond what a tile of the kernels holds (500 bytes off the grid, 512 on it), or a strided view: one host pass packs the 59 columns
    for extra in (300, 272):      # 548-byte rows on the grid, 520-byte rows on it too -> both beyond 512
        big = np.zeros(100, dtype=np.dtype(src.dtype.descr + [("blob", "V%d" % extra)]))
        for nm in src.dtype.names:
            big[nm] = src[nm]
        rows, lay = sd.table_layout(big)
        assert rows.shape == (100, 59) and rows.dtype == np.float32 and lay.row_bytes == 236
        for i, nm in enumerate(lib.SOG_FIELD_NAMES):
            assert lay.offset[i] == 4 * i
            np.testing.assert_array_equal(rows[:, i], src[nm])
    rows, lay = sd.table_layout(wide[::2])
    assert rows.shape == (50, 59) and lay.row_bytes == 236
    with pytest.raises(sd.NotEligible):
        sd.table_layout(src[["x", "y", "z"]])                                    # fields the writer reads are missing
    f8 = src.astype([(nm, "f8" if nm == "scale_1" else "f4") for nm in src.dtype.names])
    with pytest.raises(sd.NotEligible):
        sd.table_layout(f8)
    with pytest.raises(sd.NotEligible):
        sd.table_layout(np.zeros((4, 3), np.float32))
    # a table with only some f_rest fields: the coefficient count follows the reference's thresholds (:468-474)
    part = np.zeros(5, dtype=[(nm, "f4") for nm in lib.SOG_FIELD_NAMES[:14 + 30]])
    assert sd.table_layout(part)[1].n_rest == 24


def test_texture_size_is_the_references():
    for n in (1, 15, 16, 17, 1024, 3000, 20000, 10_000_000, 12_345_678):
        w, h = sd.texture_size(n)
        assert (w, h) == (int(np.ceil(np.sqrt(n) / 4) * 4), int(np.ceil(n / w / 4) * 4)) and w * h >= n and w % 4 == 0 and h % 4 == 0


def test_draws_follow_numpys_global_seed():
    np.random.seed(7)
    a = sd._draws().choice(10 ** 6, 1000, replace=False)
    np.random.seed(7)
    b = sd._draws().choice(10 ** 6, 1000, replace=False)
    np.random.seed(8)
    c = sd._draws().choice(10 ** 6, 1000, replace=False)
    assert np.array_equal(a, b) and not np.array_equal(a, c) and len(np.unique(a)) == 1000


def test_encode_without_a_gpu_fails_loudly():
    if lib.has_hip():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.GsxError):
        sd.encode(datasets.sog_scene(2000, 1), 0)


def test_file_backed_tables_are_recognised(tmp_path):
    """views of a mapped file take the plain upload (pinning them in place would copy every page of a copy-on-write mapping or
    fail on a read-only one): np.memmap in every mode, views and reinterpretations of it, np.frombuffer over an mmap"""
    import mmap
    a = np.zeros(64, np.float32)
    assert not lib.file_backed(a) and not lib.file_backed(a[3:]) and not lib.file_backed(a.view(np.uint8)[8:])
    path = tmp_path / "rows.bin"
    a.tofile(path)
    for mode in ("r", "c", "r+"):
        m = np.memmap(path, dtype=np.float32, mode=mode)
        assert lib.file_backed(m) and lib.file_backed(m[5:]) and lib.file_backed(np.asarray(m).view(np.uint8))
        assert not lib.file_backed(np.array(m))          # a copy is anonymous memory again
        del m
    with open(path, "r+b") as fh:
        mm = mmap.mmap(fh.fileno(), 0)
        b = np.frombuffer(mm, dtype=np.float32)
        assert lib.file_backed(b) and lib.file_backed(b[1:])
        del b
        mm.close()
