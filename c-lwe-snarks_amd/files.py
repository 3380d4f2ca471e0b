"""Flat on-disk images -- the Python mirror of c-lwe-snarks_amd/host/mfuoco_files.c.

CRS image, CRS_SIZE = CT_BYTES * (2D + M + 1 + 2) bytes (reference src/snark.h:6; "crs.mfuoco",
src/benchmark_snark.c:23,47-53), rows in keystream order:

    s[0..D) | as[0..D) | t | v[0..M)  (M-1 rows used) | 2 trailer rows: 40-byte public seed, zeros

The first (2D + M) rows are the device CRS that Context.setup() writes and Context.prove() reads, so a mapped image
goes to the GPU with one copy.  SSP image: SSP_SIZE = 8 * D * (M + 3) bytes of little-endian u64 (src/ssp.h:6-9).
Row files ("coeffs", src/benchmark_eval.c:44-66): any number of CT_BYTES rows.  Host only; nothing here computes.
"""
from __future__ import annotations

import numpy as np


def crs_size(p) -> int:
    return p.ctb * (2 * p.d + p.m + 3)


def ssp_size(p) -> int:
    return 8 * p.d * (p.m + 3)


def crs_write(path, p, seed: bytes, device_rows: np.ndarray) -> None:
    """device_rows: the (2D+M) * CT_BYTES bytes of the device CRS (numpy uint8, any shape)."""
    rows = np.ascontiguousarray(device_rows, dtype=np.uint8).reshape(-1)
    if len(seed) != 40 or rows.size != (2 * p.d + p.m) * p.ctb:
        raise ValueError("crs_write: need a 40-byte seed and (2D+M) rows of CT_BYTES")
    with open(path, "wb") as f:
        f.write(rows.tobytes())
        f.write(bytes(p.ctb))  # v[M-1]: allocated by the reference, never used
        f.write(seed + bytes(2 * p.ctb - 40))


def crs_map(path, p):
    """-> (seed, rows): rows is a read-only numpy memmap of shape (2D+M, CT_BYTES) ready for Context.to_device()."""
    m = np.memmap(path, dtype=np.uint8, mode="r")
    if m.size != crs_size(p):
        raise ValueError(f"{path}: {m.size} bytes, a CRS image for these parameters has {crs_size(p)}")
    nrows = 2 * p.d + p.m
    seed = bytes(m[(nrows + 1) * p.ctb:(nrows + 1) * p.ctb + 40])
    return seed, m[: nrows * p.ctb].reshape(nrows, p.ctb)


def ssp_write(path, p, ssp_u64: np.ndarray) -> None:
    a = np.ascontiguousarray(ssp_u64, dtype="<u8").reshape(-1)
    if a.size != p.d * (p.m + 3):
        raise ValueError("ssp_write: need D * (M + 3) u64 coefficients")
    a.tofile(path)


def ssp_map(path, p) -> np.ndarray:
    """-> read-only memmap of shape (M + 3, D) u64: slot 0 = t, slot i + 1 = v_i (ssp_v_offset, src/ssp.h:9)."""
    m = np.memmap(path, dtype="<u8", mode="r")
    if m.size * 8 != ssp_size(p):
        raise ValueError(f"{path}: {m.size * 8} bytes, an SSP image for these parameters has {ssp_size(p)}")
    return m.reshape(p.m + 3, p.d)


def rows_write(path, p, c8: np.ndarray) -> None:
    np.ascontiguousarray(c8, dtype=np.uint8).reshape(-1, p.ctb).tofile(path)


def rows_map(path, p) -> np.ndarray:
    m = np.memmap(path, dtype=np.uint8, mode="r")
    if m.size % p.ctb:
        raise ValueError(f"{path}: size is not a multiple of CT_BYTES = {p.ctb}")
    return m.reshape(-1, p.ctb)


def proof_write(path, p, proof_limbs_u64: np.ndarray) -> None:
    """proof_limbs_u64: 5 x (N+1) x L u64 limbs (the d_proof layout) -> 5 x (N+1) x CT_BYTES little-endian bytes."""
    a = np.ascontiguousarray(proof_limbs_u64, dtype="<u8").reshape(5 * (p.n + 1), p.L)
    b = a.view(np.uint8).reshape(5 * (p.n + 1), 8 * p.L)
    if b[:, p.ctb:].any():
        raise ValueError("proof_write: value wider than CT_BYTES")
    np.ascontiguousarray(b[:, : p.ctb]).tofile(path)


def proof_read(path, p) -> np.ndarray:
    b = np.fromfile(path, dtype=np.uint8)
    if b.size != 5 * (p.n + 1) * p.ctb:
        raise ValueError(f"{path}: not a proof image for these parameters")
    out = np.zeros((5 * (p.n + 1), 8 * p.L), dtype=np.uint8)
    out[:, : p.ctb] = b.reshape(-1, p.ctb)
    return out.view("<u8").reshape(5, p.n + 1, p.L)
