"""On-disk images, host side only: the C writer/mapper in libmfuoco_gpu_debug.so and the numpy mirror in
c_lwe_snarks_amd.files produce and accept the same bytes, with the reference's sizes (CRS_SIZE src/snark.h:6, SSP_SIZE
src/ssp.h:6, d * CT_BYTES row files src/benchmark_eval.c:44-66).  No GPU call is made (the shim creates its GPU context
lazily, on the first compute call)."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import c_lwe_snarks_amd as mf  # noqa: E402
from c_lwe_snarks_amd import files as mff  # noqa: E402

P = mf.DEBUG
CTB = 92
ROW = ctypes.c_uint8 * CTB


class Crs(ctypes.Structure):  # struct crs, src/snark.h:27-33
    _fields_ = [("seed", ctypes.c_uint8 * 40), ("s", ctypes.c_void_p), ("as_", ctypes.c_void_p), ("v", ctypes.c_void_p), ("t", ctypes.c_void_p)]


@pytest.fixture(scope="module")
def shim():
    path = os.path.join(ROOT, "c-lwe-snarks_amd", "libmfuoco_gpu_debug.so")
    if not os.path.exists(path):
        pytest.skip("host shim not built (needs gmp.h at build time)")
    lib = ctypes.CDLL(path)
    lib.mfuoco_crs_save.argtypes = [ctypes.c_char_p, ctypes.POINTER(Crs)]
    lib.mfuoco_crs_map.argtypes = [ctypes.POINTER(Crs), ctypes.c_char_p, ctypes.c_int]
    lib.mfuoco_crs_unmap.argtypes = [ctypes.POINTER(Crs)]
    lib.mfuoco_ssp_save.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    lib.mfuoco_ssp_map.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.mfuoco_ssp_map.restype = ctypes.c_void_p
    lib.mfuoco_ssp_unmap.argtypes = [ctypes.c_void_p]
    lib.mfuoco_rows_save.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.mfuoco_rows_map.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]
    lib.mfuoco_rows_map.restype = ctypes.c_void_p
    lib.mfuoco_rows_unmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    return lib


def _random_crs(rng):
    seed = bytes(rng.integers(0, 256, 40, dtype=np.uint8))
    rows = rng.integers(0, 256, size=(2 * P.d + P.m, CTB), dtype=np.uint8)
    return seed, rows


def test_sizes_match_reference_macros():
    assert P.ctb == CTB
    assert mff.crs_size(P) == CTB * (2 * 256 + 64 + 1 + 2)
    assert mff.ssp_size(P) == 256 * 8 * (64 + 3)
    assert mff.crs_size(mf.DEFAULT) == CTB * (2 * 32768 + 21845 + 1 + 2)
    assert mff.ssp_size(mf.DEFAULT) == 32768 * 8 * (21845 + 3)


def test_crs_image_python_roundtrip_and_layout(tmp_path):
    rng = np.random.default_rng(1)
    seed, rows = _random_crs(rng)
    path = str(tmp_path / "crs.mfuoco")
    mff.crs_write(path, P, seed, rows)
    assert os.path.getsize(path) == mff.crs_size(P)
    raw = np.fromfile(path, dtype=np.uint8)
    assert np.array_equal(raw[: rows.size], rows.reshape(-1))
    assert not raw[rows.size: rows.size + CTB].any()  # v[M-1], unused
    assert bytes(raw[rows.size + CTB: rows.size + CTB + 40]) == seed
    assert not raw[rows.size + CTB + 40:].any()
    seed2, rows2 = mff.crs_map(path, P)
    assert seed2 == seed and np.array_equal(rows2, rows)
    with pytest.raises(ValueError):
        mff.crs_map(path, mf.DEFAULT)
    with pytest.raises(ValueError):
        mff.crs_write(path, P, seed[:39], rows)


def test_crs_image_c_and_python_agree(shim, tmp_path):
    rng = np.random.default_rng(2)
    seed, rows = _random_crs(rng)
    s = np.ascontiguousarray(rows[: P.d])
    as_ = np.ascontiguousarray(rows[P.d: 2 * P.d])
    t = np.ascontiguousarray(rows[2 * P.d])
    v = np.zeros((P.m, CTB), dtype=np.uint8)
    v[: P.m - 1] = rows[2 * P.d + 1:]
    crs = Crs()
    ctypes.memmove(crs.seed, seed, 40)
    crs.s, crs.as_, crs.t, crs.v = (a.ctypes.data for a in (s, as_, t, v))
    c_path, py_path = str(tmp_path / "c.mfuoco"), str(tmp_path / "py.mfuoco")
    assert shim.mfuoco_crs_save(c_path.encode(), ctypes.byref(crs)) == 0
    mff.crs_write(py_path, P, seed, rows)
    assert open(c_path, "rb").read() == open(py_path, "rb").read()
    # the C mapper hands back the four arrays of struct crs at the documented offsets
    m = Crs()
    assert shim.mfuoco_crs_map(ctypes.byref(m), py_path.encode(), 0) == 0
    assert bytes(m.seed) == seed
    assert m.as_ - m.s == P.d * CTB and m.t - m.s == 2 * P.d * CTB and m.v - m.t == CTB
    assert ctypes.string_at(m.s, rows.size) == rows.tobytes()
    shim.mfuoco_crs_unmap(ctypes.byref(m))
    assert m.s is None
    # a file of another size is refused
    with open(str(tmp_path / "short"), "wb") as f:
        f.write(b"\0" * 100)
    assert shim.mfuoco_crs_map(ctypes.byref(m), str(tmp_path / "short").encode(), 0) == -1
    assert shim.mfuoco_crs_map(ctypes.byref(m), str(tmp_path / "absent").encode(), 0) == -1


def test_ssp_and_row_images(shim, tmp_path):
    rng = np.random.default_rng(3)
    ssp = rng.integers(0, mf.P, size=(P.m + 3, P.d), dtype=np.uint64)
    c_path, py_path = str(tmp_path / "c_ssp"), str(tmp_path / "py_ssp")
    assert shim.mfuoco_ssp_save(c_path.encode(), ssp.ctypes.data) == 0
    mff.ssp_write(py_path, P, ssp)
    assert open(c_path, "rb").read() == open(py_path, "rb").read()
    assert os.path.getsize(c_path) == mff.ssp_size(P)
    assert np.array_equal(mff.ssp_map(c_path, P), ssp)
    ptr = shim.mfuoco_ssp_map(py_path.encode(), 0)
    assert ptr and ctypes.string_at(ptr, ssp.nbytes) == ssp.tobytes()
    shim.mfuoco_ssp_unmap(ptr)

    c8 = rng.integers(0, 256, size=(37, CTB), dtype=np.uint8)
    rp = str(tmp_path / "coeffs")
    assert shim.mfuoco_rows_save(rp.encode(), c8.ctypes.data, 37) == 0
    assert np.array_equal(mff.rows_map(rp, P), c8)
    n = ctypes.c_size_t(0)
    ptr = shim.mfuoco_rows_map(rp.encode(), ctypes.byref(n))
    assert ptr and n.value == 37 and ctypes.string_at(ptr, c8.size) == c8.tobytes()
    shim.mfuoco_rows_unmap(ptr, n.value)
    with open(rp, "ab") as f:
        f.write(b"\1")
    assert not shim.mfuoco_rows_map(rp.encode(), ctypes.byref(n))  # ragged tail refused
    with pytest.raises(ValueError):
        mff.rows_map(rp, P)


def test_proof_image_roundtrip(tmp_path):
    rng = np.random.default_rng(4)
    limbs = rng.integers(0, 1 << 63, size=(5, P.n + 1, P.L), dtype=np.uint64)
    limbs[..., P.L - 1] &= np.uint64(0xFFFFFFFF)  # 736-bit values: top limb holds 32 bits
    path = str(tmp_path / "proof.mfuoco")
    mff.proof_write(path, P, limbs)
    assert os.path.getsize(path) == 5 * (P.n + 1) * CTB
    assert np.array_equal(mff.proof_read(path, P), limbs)
    limbs[0, 0, P.L - 1] = np.uint64(1 << 40)
    with pytest.raises(ValueError):
        mff.proof_write(path, P, limbs)
