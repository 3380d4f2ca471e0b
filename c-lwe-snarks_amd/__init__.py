"""c_lwe_snarks_amd -- thin ctypes binding over libmfhip.so (the C ABI in include/mfhip.h).

The product is the HIP library and its C ABI; this module only exists so that pytest, bench.py and
__graft_entry__.py can drive that ABI with torch-owned device memory.  There is no CPU fallback: if the
shared library is missing, or no HIP device is usable, construction fails loudly.

(The directory is named ``c-lwe-snarks_amd``; ``c_lwe_snarks_amd.py`` at the repo root loads it under an
importable name.)
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MFHIP_LIB") or os.path.join(_HERE, "libmfhip.so")  # ($MFHIP_LIB: another build of the same ABI, for same-box A/B runs of the tools)
P = 0xFFFFFFFB  # GAMMA_P, reference src/lwe.h:25


class MfhError(RuntimeError):
    pass


@dataclass(frozen=True)
class Params:
    """GAMMA_* of reference src/lwe.h:14-31 as runtime values."""

    n: int = 1470
    logq: int = 736
    d: int = 1 << 15
    m: int = 21845

    @property
    def L(self):  # limbs per value
        return (self.logq + 63) // 64

    @property
    def K(self):  # limbs surviving modq
        return self.logq // 64

    @property
    def lanes(self):  # uint64 lanes of 56 bits per value in the multi-GPU sums (mfh_lanes_per_value)
        return (64 * self.K + 55) // 56

    @property
    def ctb(self):  # CT_BYTES
        return self.logq // 8

    @property
    def ctr_ct(self):  # CTR_CT, src/snark.h:8
        return self.ctb * self.n

    @property
    def ctr_s(self):
        return 0

    @property
    def ctr_as(self):
        return self.ctr_ct * self.d

    @property
    def ctr_bt(self):
        return 2 * self.ctr_ct * self.d

    @property
    def ctr_bv(self):
        return 2 * self.ctr_ct * self.d + self.ctr_ct

    @property
    def ct_limbs(self):
        return (self.n + 1) * self.L


DEBUG = Params(d=256, m=64)  # the reference's !NDEBUG parameters (src/lwe.h:18-21)
DEFAULT = Params()  # NDEBUG parameters (src/lwe.h:14-17)


class _CParams(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint32), ("logq", ctypes.c_uint32), ("d", ctypes.c_uint32), ("m", ctypes.c_uint32)]


_lib = None

EXPORTS = [
    "mfh_ctx_create", "mfh_ctx_destroy", "mfh_set_stream", "mfh_sync", "mfh_scrub_staging", "mfh_last_error", "mfh_set_seed",
    "mfh_keystream", "mfh_sample_rows", "mfh_ct_add", "mfh_ct_mul_ui", "mfh_ct_addmul_ui", "mfh_eval_rows",
    "mfh_encrypt_rows", "mfh_decrypt", "mfh_ct_smudge", "mfh_ssp_upload", "mfh_witness_poly", "mfh_version",
    "mfh_workspace_bytes", "mfh_last_kernel_ms", "mfh_set_timing", "mfh_set_overlap", "mfh_eval_rows_multi", "mfh_prove_batch", "mfh_crs_mm_image_bytes", "mfh_crs_expand_mm", "mfh_crs_set_resident_mm", "mfh_witness_poly_multi", "mfh_witness_poly_mm", "mfh_poly_h_multi", "mfh_poly_mul", "mfh_poly_add", "mfh_poly_prepare_t",
    "mfh_ssp_prepare", "mfh_poly_h", "mfh_setup_messages", "mfh_setup", "mfh_setup_image", "mfh_crs_image_set_b", "mfh_prove",
    "mfh_prove_partial", "mfh_prove_finish", "mfh_ct_to_lanes", "mfh_ct_from_lanes", "mfh_lanes_per_value", "mfh_digest128", "mfh_timing_drain", "mfh_timing_busy_ms", "mfh_timing_work_rows", "mfh_set_batch_image", "mfh_set_batch_slabs", "mfh_set_expand_path", "mfh_set_encrypt_path", "mfh_set_batch_launch", "mfh_set_batch_bw", "mfh_set_mm_chunk_rows", "mfh_set_mm_pack", "mfh_set_poly_exact", "mfh_poly_exact_fallbacks", "mfh_set_mm_stream", "mfh_add_dotp", "mfh_set_encrypt_chunks", "mfh_set_witness_per", "mfh_set_eval_path", "mfh_set_decrypt_path", "mfh_decrypt_rows",
    "mfh_crs_mm_share_bytes", "mfh_crs_expand_mm_share", "mfh_crs_set_resident_mm_share", "mfh_batch_chain", "mfh_batch_witness_cols", "mfh_batch_chain_from_w", "mfh_witness_poly_mm_cols", "mfh_prove_batch_partial", "mfh_prove_batch_finish",
    "mfh_resident_row_bytes", "mfh_crs_expand", "mfh_eval_rows_resident", "mfh_crs_set_resident",
    "mfh_witness_lanes", "mfh_witness_from_lanes", "mfh_prove_partial_w", "mfh_verify",
    "mfh_ssp_set_prg", "mfh_ssp_prg_make_t", "mfh_ssp_prg_fill",
    "mfh_resident_share_rows", "mfh_crs_expand_share", "mfh_crs_set_resident_share", "mfh_crs_set_resident_prefix",
    "mfh_prove_batch_supergroup", "mfh_prove_batch_stream_wait", "mfh_set_mm_width",
]


def load_library():
    """dlopen libmfhip.so; raises MfhError (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MfhError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, u64, sz, i32, u32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32
    sig = {
        "mfh_ctx_create": (i32, [ctypes.POINTER(vp), i32, ctypes.POINTER(_CParams)]),
        "mfh_ctx_destroy": (None, [vp]),
        "mfh_set_stream": (i32, [vp, vp]),
        "mfh_sync": (i32, [vp]),
        "mfh_scrub_staging": (i32, [vp]),
        "mfh_last_error": (ctypes.c_char_p, [vp]),
        "mfh_set_seed": (i32, [vp, ctypes.c_char_p]),
        "mfh_keystream": (i32, [vp, u64, vp, sz]),
        "mfh_sample_rows": (i32, [vp, u64, sz, vp]),
        "mfh_ct_add": (i32, [vp, vp, vp, vp, sz]),
        "mfh_ct_mul_ui": (i32, [vp, vp, vp, u32, sz]),
        "mfh_ct_addmul_ui": (i32, [vp, vp, vp, u32, sz]),
        "mfh_eval_rows": (i32, [vp, u64, sz, vp, vp, vp, vp, vp, i32]),
        "mfh_encrypt_rows": (i32, [vp, u64, sz, vp, vp, vp, vp]),
        "mfh_decrypt": (i32, [vp, vp, vp, sz, vp]),
        "mfh_ct_smudge": (i32, [vp, vp, sz, ctypes.c_char_p, sz, ctypes.c_char_p]),
        "mfh_ssp_upload": (i32, [vp, vp, vp, sz, sz]),
        "mfh_witness_poly": (i32, [vp, vp, ctypes.c_char_p, u32, vp]),
        "mfh_version": (ctypes.c_char_p, []),
        "mfh_workspace_bytes": (sz, [vp]),
        "mfh_last_kernel_ms": (ctypes.c_float, [vp, ctypes.c_char_p]),
        "mfh_set_timing": (i32, [vp, i32]),
        "mfh_set_overlap": (i32, [vp, i32]),
        "mfh_eval_rows_multi": (i32, [vp, u64, sz, vp, vp, u32, u32, vp, i32]),
        "mfh_poly_h_multi": (i32, [vp, vp, vp, u32]),
        "mfh_witness_poly_mm": (i32, [vp, vp, u32, ctypes.c_char_p, sz, vp, vp]),
        "mfh_witness_poly_multi": (i32, [vp, vp, u32, ctypes.c_char_p, sz, vp, vp]),
        "mfh_crs_mm_image_bytes": (sz, [vp]),
        "mfh_crs_expand_mm": (i32, [vp, vp, vp]),
        "mfh_crs_set_resident_mm": (i32, [vp, vp]),
        "mfh_prove_batch": (i32, [vp, vp, vp, u32, ctypes.c_char_p, sz, vp, ctypes.c_char_p, sz, ctypes.c_char_p, vp]),
        "mfh_prove_batch_supergroup": (u32, [vp]),
        "mfh_set_mm_width": (i32, [vp, u32, i32]),
        "mfh_prove_batch_stream_wait": (i32, [vp, u32, vp]),
        "mfh_poly_mul": (i32, [vp, vp, sz, vp, sz, vp]),
        "mfh_poly_add": (i32, [vp, vp, vp, sz, vp]),
        "mfh_poly_prepare_t": (i32, [vp, vp]),
        "mfh_ssp_prepare": (i32, [vp, vp]),
        "mfh_poly_h": (i32, [vp, vp, vp]),
        "mfh_setup_messages": (i32, [vp, vp, u32, u32, u32, vp]),
        "mfh_setup": (i32, [vp, vp, u32, u32, u32, vp, vp, vp]),
        "mfh_setup_image": (i32, [vp, vp, u32, u32, u32, vp, vp, vp, vp]),
        "mfh_prove": (i32, [vp, vp, vp, ctypes.c_char_p, u32, ctypes.c_char_p, sz, ctypes.c_char_p, vp]),
        "mfh_prove_partial": (i32, [vp, vp, vp, ctypes.c_char_p, u32, u32, u32, vp]),
        "mfh_prove_finish": (i32, [vp, vp, ctypes.c_char_p, sz, ctypes.c_char_p]),
        "mfh_lanes_per_value": (u32, [vp]),
        "mfh_digest128": (i32, [vp, vp, sz, vp]),
        "mfh_ct_to_lanes": (i32, [vp, vp, sz, vp]),
        "mfh_ct_from_lanes": (i32, [vp, vp, sz, vp]),
        "mfh_add_dotp": (i32, [vp, vp, vp, vp, sz]),
        "mfh_crs_set_resident_prefix": (i32, [vp, vp, sz]),
        "mfh_resident_share_rows": (sz, [vp, u32, u32]),
        "mfh_crs_expand_share": (i32, [vp, vp, u32, u32, vp]),
        "mfh_crs_set_resident_share": (i32, [vp, vp, u32, u32]),
        "mfh_ssp_set_prg": (i32, [vp, u64, vp]),
        "mfh_ssp_prg_make_t": (i32, [vp, u64, ctypes.c_char_p, vp]),
        "mfh_ssp_prg_fill": (i32, [vp, u64, sz, sz, vp]),
        "mfh_verify": (i32, [vp, vp, u32, u32, u32, vp, vp, sz, vp]),
        "mfh_witness_lanes": (i32, [vp, vp, ctypes.c_char_p, u32, u32, vp]),
        "mfh_witness_from_lanes": (i32, [vp, vp, vp, u32, vp]),
        "mfh_prove_partial_w": (i32, [vp, vp, vp, ctypes.c_char_p, u32, u32, u32, vp, vp]),
        "mfh_resident_row_bytes": (sz, [vp]),
        "mfh_crs_expand": (i32, [vp, u64, sz, vp, vp]),
        "mfh_crs_image_set_b": (i32, [vp, sz, sz, vp, vp]),
        "mfh_eval_rows_resident": (i32, [vp, vp, sz, sz, vp, vp, vp, vp, i32]),
        "mfh_crs_set_resident": (i32, [vp, vp]),
        "mfh_timing_drain": (i32, [vp, ctypes.c_char_p, ctypes.POINTER(u64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u64),
                                   ctypes.POINTER(ctypes.c_float)]),
        "mfh_timing_busy_ms": (ctypes.c_double, [vp]),
        "mfh_timing_work_rows": (u64, [vp]),
        "mfh_set_batch_image": (i32, [vp, i32]),
        "mfh_set_mm_chunk_rows": (i32, [vp, u32]),
        "mfh_set_mm_pack": (i32, [vp, i32]),
        "mfh_set_poly_exact": (i32, [vp, i32]),
        "mfh_poly_exact_fallbacks": (ctypes.c_long, [vp]),
        "mfh_set_mm_stream": (i32, [vp, i32, i32, i32, u32]),
        "mfh_set_batch_launch": (i32, [vp, u32, i32]),
        "mfh_set_batch_bw": (i32, [vp, i32]),
        "mfh_set_encrypt_path": (i32, [vp, i32]),
        "mfh_set_expand_path": (i32, [vp, i32]),
        "mfh_set_batch_slabs": (i32, [vp, u32]),
        "mfh_set_encrypt_chunks": (i32, [vp, u32]),
        "mfh_set_eval_path": (i32, [vp, i32]),
        "mfh_set_decrypt_path": (i32, [vp, i32]),
        "mfh_decrypt_rows": (i32, [vp, u64, sz, vp, vp, vp]),
        "mfh_set_witness_per": (i32, [vp, u32]),
        "mfh_crs_mm_share_bytes": (sz, [vp, u32, u32]),
        "mfh_crs_expand_mm_share": (i32, [vp, vp, u32, u32, vp]),
        "mfh_crs_set_resident_mm_share": (i32, [vp, vp, u32, u32]),
        "mfh_batch_chain": (i32, [vp, vp, u32, ctypes.c_char_p, sz, vp, vp, vp, vp]),
        "mfh_batch_witness_cols": (i32, [vp, vp, u32, ctypes.c_char_p, sz, vp, u32, u32, vp, sz]),
        "mfh_batch_chain_from_w": (i32, [vp, vp, u32, vp, vp, vp]),
        "mfh_witness_poly_mm_cols": (i32, [vp, vp, u32, ctypes.c_char_p, sz, vp, u32, u32, vp, sz]),
        "mfh_prove_batch_partial": (i32, [vp, vp, u32, u32, u32, ctypes.c_char_p, sz, vp, vp, vp, sz, vp]),
        "mfh_prove_batch_finish": (i32, [vp, vp, u32, vp, ctypes.c_char_p, sz, ctypes.c_char_p, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class Context:
    """One prover context on one GPU (one process per GPU).  Mirrors the reference's implicit global state:
    an `rng_t` seeded from the CRS seed (src/snark.c:59,119) plus the parameter macros."""

    def __init__(self, params: Params = DEFAULT, device: int = 0):
        import torch

        if not torch.cuda.is_available():
            raise MfhError("no HIP device visible: the MI355X path has no CPU fallback")
        self.torch = torch
        self.lib = load_library()
        self.params = params
        self.device = torch.device("cuda", device)
        self._h = ctypes.c_void_p()
        cp = _CParams(params.n, params.logq, params.d, params.m)
        rc = self.lib.mfh_ctx_create(ctypes.byref(self._h), device, ctypes.byref(cp))
        if rc != 0:
            raise MfhError(f"mfh_ctx_create failed ({rc})")
        # run on torch's current stream so torch allocations/copies and our kernels are ordered
        self._chk(self.lib.mfh_set_stream(self._h, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def close(self):
        if self._h:
            self.lib.mfh_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise MfhError(f"libmfhip error {rc}: {self.lib.mfh_last_error(self._h).decode()}")

    # -- helpers ---------------------------------------------------------------------------------------
    def empty(self, nbytes):
        return self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.device)

    def zeros(self, nbytes):
        return self.torch.zeros(int(nbytes), dtype=self.torch.uint8, device=self.device)

    def to_device(self, arr):
        a = np.ascontiguousarray(arr)
        if not a.flags.writeable:  # read-only file mappings (files.crs_map): the host tensor is only a copy source
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore", UserWarning)
                return self.torch.from_numpy(a.view(np.uint8).reshape(-1)).to(self.device)
        return self.torch.from_numpy(a.view(np.uint8).reshape(-1)).to(self.device)

    def to_host(self, t, dtype=np.uint8):
        return t.cpu().numpy().view(dtype)

    def sync(self):
        self._chk(self.lib.mfh_sync(self._h))

    def set_timing(self, on=True):
        self._chk(self.lib.mfh_set_timing(self._h, 1 if on else 0))

    def set_overlap(self, mode=1):
        """0 = one stream, 1 = two streams with the queueing order picked automatically, 2 = b_w first, 3 = chain first"""
        self._chk(self.lib.mfh_set_overlap(self._h, int(mode)))

    def set_batch_image(self, on=True):
        """prove_batch with more than 31 proofs: expand the CRS once per call into a transient image and stream it per group (default) or not"""
        self._chk(self.lib.mfh_set_batch_image(self._h, 1 if on else 0))

    def set_batch_slabs(self, nslabs=0):
        """prove_batch: 0 = row slabs only when the image does not fit HBM, n = always n slabs (each expanded once and streamed for every group)"""
        self._chk(self.lib.mfh_set_batch_slabs(self._h, int(nslabs)))

    def set_expand_path(self, path=0):
        """crs_expand_mm*: 0 = barrier-free writer (MFMA transposition), 1 = LDS-tile writer"""
        self._chk(self.lib.mfh_set_expand_path(self._h, int(path)))

    def set_encrypt_path(self, path=0):
        """encrypt_rows / setup: 0 = by batch size, 1 = VALU kernel, 2 = matrix-core kernel (<sk, a> as a Toeplitz int8 GEMM)"""
        self._chk(self.lib.mfh_set_encrypt_path(self._h, int(path)))

    def set_eval_path(self, path=0):
        """eval_rows / prove: 0 = tile kernel (k_eval, default), 1 = wave-autonomous kernel (k_eval_w, logq 736; measured 4 % slower)"""
        self._chk(self.lib.mfh_set_eval_path(self._h, int(path)))

    def set_encrypt_chunks(self, chunks=0):
        """k_encrypt_mm: column chunks per row (0 = picked from the batch size)"""
        self._chk(self.lib.mfh_set_encrypt_chunks(self._h, int(chunks)))

    def set_witness_per(self, statements=0):
        """statements per witness GEMM pass of the batch chain (0 = one pass per super-group of up to 255)"""
        self._chk(self.lib.mfh_set_witness_per(self._h, int(statements)))

    def set_batch_launch(self, groups_per_launch=4, merge_regions=True):
        """streaming regime of prove_batch: groups of 31 proofs per pass over a region's image; S and AS groups in one launch or two"""
        self._chk(self.lib.mfh_set_batch_launch(self._h, int(groups_per_launch), 1 if merge_regions else 0))

    def set_batch_bw(self, merged=True):
        """b_w of all super-groups of a call in one streaming launch per 8 of them (default) or one launch per super-group"""
        self._chk(self.lib.mfh_set_batch_bw(self._h, 1 if merged else 0))

    def set_mm_stream(self, map=0, persistent=False, sync_mode=0, spin_max=64):
        """layout of a streaming launch with several groups: slot map (0 | 1), persistent one-workgroup-per-CU grid, rendezvous of the sharers (0 | 1 | 2)"""
        self._chk(self.lib.mfh_set_mm_stream(self._h, int(map), int(persistent), int(sync_mode), int(spin_max)))

    def set_mm_width(self, per_xcd=32, early_chain=False):
        """workgroups per XCD of the persistent S / AS launch (32 = every CU); early_chain: chain / epilogues queued beside the streaming launches"""
        self._chk(self.lib.mfh_set_mm_width(self._h, int(per_xcd), 1 if early_chain else 0))

    def set_mm_pack(self, on=True):
        """streaming kernels hand their partial products to the epilogue recombined (default) or as int32 (rounds 1 - 5): same results"""
        self._chk(self.lib.mfh_set_mm_pack(self._h, 1 if on else 0))

    def set_poly_exact(self, mode=1):
        """batches of h = (v^2 - 1) / t try the exact-division path (two cyclic products of half the length, checked on the device) before Euclidean division: same results.
        0 / False: never; 1 / True: unless a recent batch failed the check (64 batches of Euclidean division alone follow); 2: always"""
        self._chk(self.lib.mfh_set_poly_exact(self._h, int(mode)))

    def poly_exact_fallbacks(self):
        """statements whose exact-division result failed the device check since the last call (they were recomputed by Euclidean division); -1: no exact path for this t"""
        return int(self.lib.mfh_poly_exact_fallbacks(self._h))

    def set_mm_chunk_rows(self, rows=0):
        """rows per row chunk of the matrix-core launches (<= 131071; 0 = default): smaller values force several chunks"""
        self._chk(self.lib.mfh_set_mm_chunk_rows(self._h, int(rows)))

    def timing_drain(self, which):
        """(launch count, total ms, total rows) of the launches of kind `which` since the last drain"""
        n, rows, ms, last = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_double(), ctypes.c_float()
        self._chk(self.lib.mfh_timing_drain(self._h, which.encode(), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(rows), ctypes.byref(last)))
        return n.value, ms.value, rows.value

    def timing_busy_ms(self):
        """union of the spans of the launches the last timing_drain matched (ms): < total ms when two streams overlapped"""
        return float(self.lib.mfh_timing_busy_ms(self._h))

    def timing_work_rows(self):
        """rows x evaluations served by the launches the last timing_drain matched"""
        return int(self.lib.mfh_timing_work_rows(self._h))

    def last_kernel_ms(self, which):
        return float(self.lib.mfh_last_kernel_ms(self._h, which.encode()))

    # -- reference-shaped operations ----------------------------------------------------------------------
    def set_seed(self, seed: bytes):
        """rng_init(rng, seed) -- src/entropy.c:58-61."""
        assert len(seed) == 40
        self._chk(self.lib.mfh_set_seed(self._h, bytes(seed)))

    def keystream(self, off, nbytes, out=None):
        """rng_seek(off); rng_gen(nbytes) -- src/entropy.c:46-56, src/aes.c:104-144."""
        out = self.empty(nbytes) if out is None else out
        self._chk(self.lib.mfh_keystream(self._h, off, _ptr(out), nbytes))
        return out

    def sample_rows(self, off, nrows):
        """mpz2_urandommv for nrows rows -- src/entropy.h:62-66.  Returns uint8 tensor of nrows*n*L*8 bytes."""
        p = self.params
        out = self.empty(nrows * p.n * p.L * 8)
        self._chk(self.lib.mfh_sample_rows(self._h, off, nrows, _ptr(out)))
        return out

    def ct_add(self, a, b, count=1, out=None):
        out = self.empty(a.numel()) if out is None else out
        self._chk(self.lib.mfh_ct_add(self._h, _ptr(out), _ptr(a), _ptr(b), count))
        return out

    def ct_mul_ui(self, a, x, count=1, out=None):
        out = self.empty(a.numel()) if out is None else out
        self._chk(self.lib.mfh_ct_mul_ui(self._h, _ptr(out), _ptr(a), x, count))
        return out

    def ct_addmul_ui(self, rop, a, x, count=1):
        self._chk(self.lib.mfh_ct_addmul_ui(self._h, _ptr(rop), _ptr(a), x, count))
        return rop

    def eval_rows(self, off, nrows, c8, coeff0, coeff1=None, rop0=None, rop1=None, accumulate=False):
        """eval_poly (src/lwe.c:176-186) for one or two coefficient vectors; returns (rop0, rop1)."""
        p = self.params
        if rop0 is None:
            rop0 = self.empty(p.ct_limbs * 8)
        if coeff1 is not None and rop1 is None:
            rop1 = self.empty(p.ct_limbs * 8)
        self._chk(self.lib.mfh_eval_rows(self._h, off, nrows, _ptr(c8), _ptr(coeff0), _ptr(coeff1), _ptr(rop0), _ptr(rop1),
                                         1 if accumulate else 0))
        return rop0, rop1

    def eval_rows_multi(self, off, nrows, c8, coeffs, nvec, out=None, accumulate=False, coeff_bytes=4):
        """coeffs: nvec x nrows uint32 on the device (vector-major) -> nvec ciphertexts (vector-major); matrix-core path.
        coeff_bytes = 1: every coefficient < 256 (up to 127 vectors); 4: any uint32 (up to 31 vectors)"""
        out = self.empty(nvec * self.params.ct_limbs * 8) if out is None else out
        self._chk(self.lib.mfh_eval_rows_multi(self._h, off, nrows, _ptr(c8), _ptr(coeffs), nvec, coeff_bytes, _ptr(out), 1 if accumulate else 0))
        return out

    def encrypt_rows(self, off, nrows, sk, msg, err, out=None):
        """regev_encrypt2 + ct_export batch (src/lwe.c:78-97,115-119)."""
        p = self.params
        out = self.empty(nrows * p.ctb) if out is None else out
        self._chk(self.lib.mfh_encrypt_rows(self._h, off, nrows, _ptr(sk), _ptr(msg), _ptr(err), _ptr(out)))
        return out

    def decrypt(self, sk, cts, count, out=None):
        """regev_decrypt (src/lwe.c:105-111) of `count` full ciphertexts -> count uint32"""
        out = self.empty(4 * count) if out is None else out
        self._chk(self.lib.mfh_decrypt(self._h, _ptr(sk), _ptr(cts), count, _ptr(out)))
        return out

    def set_decrypt_path(self, path=0):
        """decrypt: 0 = by batch size, 1 = VALU kernel, 2 = matrix-core kernel"""
        self._chk(self.lib.mfh_set_decrypt_path(self._h, int(path)))

    def decrypt_rows(self, off, nrows, sk, c8, out=None):
        """regev_decrypt of nrows seed-compressed ciphertexts (a regenerated from the stream at off + i * CTR_CT, b = c8[i]) -> nrows uint32"""
        out = self.empty(4 * nrows) if out is None else out
        self._chk(self.lib.mfh_decrypt_rows(self._h, off, nrows, _ptr(sk), _ptr(c8), _ptr(out)))
        return out

    def ct_smudge(self, cts, count, mag: bytes, maglen, sign: bytes):
        self._chk(self.lib.mfh_ct_smudge(self._h, _ptr(cts), count, bytes(mag), maglen, bytes(sign)))
        return cts

    def ssp_upload(self, ssp_host_u64: np.ndarray, d_ssp=None, first_slot=0, nslots=None):
        p = self.params
        nslots = p.m + 3 if nslots is None else nslots
        if d_ssp is None:
            d_ssp = self.empty((p.m + 3) * p.d * 4)
        a = np.ascontiguousarray(ssp_host_u64)
        self._chk(self.lib.mfh_ssp_upload(self._h, ctypes.c_void_p(a.ctypes.data), _ptr(d_ssp), first_slot, nslots))
        return d_ssp

    def witness_poly(self, d_ssp, witness_bits: bytes, delta):
        out = self.empty(self.params.d * 4)
        self._chk(self.lib.mfh_witness_poly(self._h, _ptr(d_ssp), bytes(witness_bits), delta, _ptr(out)))
        return out

    # -- polynomial step, setup, prover ---------------------------------------------------------------------
    def poly_mul(self, a, la, b, lb):
        out = self.empty((la + lb - 1) * 4)
        self._chk(self.lib.mfh_poly_mul(self._h, _ptr(a), la, _ptr(b), lb, _ptr(out)))
        return out

    def poly_add(self, a, b, count):
        """a + b over F_p, `count` coefficients (nmod_poly_add, src/snark.c:161)"""
        out = self.empty(count * 4)
        self._chk(self.lib.mfh_poly_add(self._h, _ptr(a), _ptr(b), count, _ptr(out)))
        return out

    def ssp_prepare(self, d_ssp):
        """per-SSP precomputation for the division by t(x) (src/snark.c:169)"""
        self._chk(self.lib.mfh_ssp_prepare(self._h, _ptr(d_ssp)))

    def poly_prepare_t(self, d_t):
        self._chk(self.lib.mfh_poly_prepare_t(self._h, _ptr(d_t)))

    def poly_h(self, d_v):
        out = self.empty(self.params.d * 4)
        self._chk(self.lib.mfh_poly_h(self._h, _ptr(d_v), _ptr(out)))
        return out

    def poly_h_many(self, d_v, nb):
        """h_k = floor((v_k^2 - 1) / t) for nb polynomials side by side (mfh_poly_h_multi): one set of launches for the batch"""
        out = self.empty(nb * self.params.d * 4)
        self._chk(self.lib.mfh_poly_h_multi(self._h, _ptr(d_v), _ptr(out), nb))
        return out

    def setup_messages(self, d_ssp, alpha, beta, s):
        p = self.params
        out = self.empty((2 * p.d + p.m) * 4)
        self._chk(self.lib.mfh_setup_messages(self._h, _ptr(d_ssp), alpha, beta, s, _ptr(out)))
        return out

    def setup(self, d_ssp, alpha, beta, s, d_sk, d_err, out=None):
        """setup() (src/snark.c:57-115): returns the device CRS, (2d+m)*CT_BYTES bytes in stream order."""
        p = self.params
        out = self.empty((2 * p.d + p.m) * p.ctb) if out is None else out
        self._chk(self.lib.mfh_setup(self._h, _ptr(d_ssp), alpha, beta, s, _ptr(d_sk), _ptr(d_err), _ptr(out)))
        return out

    def setup_image(self, d_ssp, alpha, beta, s, d_sk, d_err, out=None, rows=None):
        """setup() leaving the expanded rows behind: returns (device CRS, row image for set_resident)"""
        p = self.params
        out = self.empty((2 * p.d + p.m) * p.ctb) if out is None else out
        rows = self.empty((2 * p.d + p.m) * self.resident_row_bytes()) if rows is None else rows
        self._chk(self.lib.mfh_setup_image(self._h, _ptr(d_ssp), alpha, beta, s, _ptr(d_sk), _ptr(d_err), _ptr(out), _ptr(rows)))
        return out, rows

    def prove(self, d_crs, d_ssp, witness_bits: bytes, delta, smudge_mag: bytes, smudge_sign: bytes, maglen=80, out=None):
        """prover() (src/snark.c:117-190): returns 5 ciphertexts h | hat_h | hat_v | v_w | b_w."""
        p = self.params
        out = self.empty(5 * p.ct_limbs * 8) if out is None else out
        assert len(smudge_mag) == 5 * maglen and len(smudge_sign) == 5
        self._chk(self.lib.mfh_prove(self._h, _ptr(d_crs), _ptr(d_ssp), bytes(witness_bits), delta, bytes(smudge_mag), maglen,
                                     bytes(smudge_sign), _ptr(out)))
        return out

    def crs_expand_mm(self, d_crs, out=None):
        """the CRS expanded once for the matrix-core batch prover (11.3 GB at the default instance, MFMA A-fragment order)"""
        out = self.empty(int(self.lib.mfh_crs_mm_image_bytes(self._h))) if out is None else out
        self._chk(self.lib.mfh_crs_expand_mm(self._h, _ptr(d_crs), _ptr(out)))
        return out

    def set_resident_mm(self, image):
        self._resident_mm = image
        self._chk(self.lib.mfh_crs_set_resident_mm(self._h, _ptr(image)))

    def witness_poly_many(self, d_ssp, witness_bits_list, deltas, mm=True):
        """w polynomials of up to 256 (mm) / 12 statements in one read of the SSP -> len x d uint32 on the device"""
        p = self.params
        nb = len(witness_bits_list)
        stride = (p.m + 6) // 8
        bits = b"".join(bytes(w[:stride]).ljust(stride, b"\0") for w in witness_bits_list)
        dl = (ctypes.c_uint32 * nb)(*[int(x) for x in deltas])
        out = self.empty(nb * p.d * 4)
        fn = self.lib.mfh_witness_poly_mm if mm else self.lib.mfh_witness_poly_multi
        self._chk(fn(self._h, _ptr(d_ssp), nb, bits, stride, ctypes.cast(dl, ctypes.c_void_p), _ptr(out)))
        return out

    def prove_batch(self, d_crs, d_ssp, witness_bits_list, deltas, smudge_mags, smudge_signs, maglen=80, out=None):
        """prover() for len(witness_bits_list) statements under one CRS: regions expanded once per group of 31, MAC on the matrix cores"""
        p = self.params
        nb = len(witness_bits_list)
        stride = (p.m + 6) // 8
        bits = b"".join(bytes(w[:stride]).ljust(stride, b"\0") for w in witness_bits_list)
        dl = (ctypes.c_uint32 * nb)(*[int(x) for x in deltas])
        mags = b"".join(bytes(x) for x in smudge_mags)
        signs = b"".join(bytes(x) for x in smudge_signs)
        assert len(mags) == nb * 5 * maglen and len(signs) == nb * 5
        out = self.empty(nb * 5 * p.ct_limbs * 8) if out is None else out
        self._chk(self.lib.mfh_prove_batch(self._h, _ptr(d_crs), _ptr(d_ssp), nb, bits, stride, ctypes.cast(dl, ctypes.c_void_p), mags, maglen,
                                           signs, _ptr(out)))
        return out

    def prove_batch_supergroup(self):
        """statements per super-group of the last prove_batch call (255 streaming, 248 regenerating; 0 before the first call)"""
        return int(self.lib.mfh_prove_batch_supergroup(self._h))

    def prove_batch_stream_wait(self, upto, stream):
        """make the torch stream `stream` wait until statements [0, upto) of the last prove_batch call are final in its output"""
        self._chk(self.lib.mfh_prove_batch_stream_wait(self._h, int(upto), ctypes.c_void_p(stream.cuda_stream)))

    # -- row-sharded batch prover (one process per GPU; dist.prove_batch_sharded drives the sequence) ----------------
    def _pack_bits(self, witness_bits_list):
        stride = (self.params.m + 6) // 8
        return b"".join(bytes(w[:stride]).ljust(stride, b"\0") for w in witness_bits_list), stride

    def crs_expand_mm_share(self, d_crs, rank, world, out=None):
        """rank's row shares of the S | AS | BT+BV regions expanded in MFMA A-fragment order (45 GB per GPU for the 2^20-constraint CRS on 8)"""
        out = self.empty(int(self.lib.mfh_crs_mm_share_bytes(self._h, rank, world))) if out is None else out
        self._chk(self.lib.mfh_crs_expand_mm_share(self._h, _ptr(d_crs), rank, world, _ptr(out)))
        return out

    def set_resident_mm_share(self, image, rank, world):
        self._resident_mm = image
        self._chk(self.lib.mfh_crs_set_resident_mm_share(self._h, _ptr(image), rank, world))

    def batch_chain(self, d_ssp, witness_bits_list, deltas, out=None):
        """w | h | v of the statements (src/snark.c:141-169): int32 tensor [3][len][d] (the uint32 coefficients' bit patterns)"""
        p = self.params
        nb = len(witness_bits_list)
        out = self.torch.empty((3, nb, p.d), dtype=self.torch.int32, device=self.device) if out is None else out
        if nb == 0:
            return out
        bits, stride = self._pack_bits(witness_bits_list)
        dl = (ctypes.c_uint32 * nb)(*[int(x) for x in deltas])
        self._chk(self.lib.mfh_batch_chain(self._h, _ptr(d_ssp), nb, bits, stride, ctypes.cast(dl, ctypes.c_void_p), _ptr(out[0]), _ptr(out[1]), _ptr(out[2])))
        return out

    def batch_witness_cols(self, d_ssp, witness_bits_list, deltas, col0, ncols, out=None):
        """coefficients [col0, col0 + ncols) of w of every statement: int32 tensor [len][ncols]"""
        nb = len(witness_bits_list)
        out = self.torch.empty((nb, ncols), dtype=self.torch.int32, device=self.device) if out is None else out
        if nb == 0 or ncols == 0:
            return out
        bits, stride = self._pack_bits(witness_bits_list)
        dl = (ctypes.c_uint32 * nb)(*[int(x) for x in deltas])
        self._chk(self.lib.mfh_batch_witness_cols(self._h, _ptr(d_ssp), nb, bits, stride, ctypes.cast(dl, ctypes.c_void_p), int(col0), int(ncols), _ptr(out), int(ncols)))
        return out

    def batch_chain_from_w(self, d_ssp, whv):
        """whv: int32 [3][n][d] with the whole w polynomials in whv[0]; fills whv[1] = h and whv[2] = v"""
        n = whv.shape[1]
        if n:
            self._chk(self.lib.mfh_batch_chain_from_w(self._h, _ptr(d_ssp), n, _ptr(whv[0]), _ptr(whv[1]), _ptr(whv[2])))
        return whv

    def prove_batch_partial(self, d_crs, rank, world, witness_bits_list, d_w, d_h, d_v, coef_stride, out=None):
        """rank's row shares of the five ciphertexts of every statement: len x 5 partial ciphertexts (no delta ct_t term, un-smudged)"""
        p = self.params
        nb = len(witness_bits_list)
        need = nb * 5 * p.ct_limbs * 8
        out = self.empty(need) if out is None else out
        if out.numel() * out.element_size() < need:
            raise MfhError(f"prove_batch_partial: `out` holds {out.numel() * out.element_size()} bytes, {nb} statements need {need}")
        bits, stride = self._pack_bits(witness_bits_list)
        self._chk(self.lib.mfh_prove_batch_partial(self._h, _ptr(d_crs), rank, world, nb, bits, stride, _ptr(d_w), _ptr(d_h), _ptr(d_v), int(coef_stride),
                                                   _ptr(out)))
        return out

    def prove_batch_finish(self, d_crs, deltas, smudge_mags, smudge_signs, d_proofs, maglen=80):
        """b_w += delta ct_t, then the smudging, on len(deltas) summed proofs in place"""
        nb = len(deltas)
        if nb == 0:
            return d_proofs
        dl = (ctypes.c_uint32 * nb)(*[int(x) for x in deltas])
        mags = b"".join(bytes(x) for x in smudge_mags)
        signs = b"".join(bytes(x) for x in smudge_signs)
        assert len(mags) == nb * 5 * maglen and len(signs) == nb * 5
        self._chk(self.lib.mfh_prove_batch_finish(self._h, _ptr(d_crs), nb, ctypes.cast(dl, ctypes.c_void_p), mags, maglen, signs, _ptr(d_proofs)))
        return d_proofs

    # -- row-sharded prover (one process per GPU; see dist.py) ------------------------------------------------
    def prove_partial(self, d_crs, d_ssp, witness_bits: bytes, delta, rank, world, out=None):
        p = self.params
        out = self.empty(5 * p.ct_limbs * 8) if out is None else out
        self._chk(self.lib.mfh_prove_partial(self._h, _ptr(d_crs), _ptr(d_ssp), bytes(witness_bits), delta, rank, world, _ptr(out)))
        return out

    def prove_finish(self, d_proof, smudge_mag: bytes, smudge_sign: bytes, maglen=80):
        self._chk(self.lib.mfh_prove_finish(self._h, _ptr(d_proof), bytes(smudge_mag), maglen, bytes(smudge_sign)))
        return d_proof

    def digest128(self, d_buf, nbytes=None):
        """128-bit digest of a device buffer (a cache key, not a cryptographic hash): (lo, hi)"""
        h = (ctypes.c_uint64 * 2)()
        n = d_buf.numel() * d_buf.element_size() if nbytes is None else int(nbytes)
        self._chk(self.lib.mfh_digest128(self._h, _ptr(d_buf), n, ctypes.cast(h, ctypes.c_void_p)))
        return int(h[0]), int(h[1])

    def ct_to_lanes(self, d_cts, count, out=None):
        p = self.params
        out = self.torch.empty(count * (p.n + 1) * p.lanes, dtype=self.torch.int64, device=self.device) if out is None else out
        self._chk(self.lib.mfh_ct_to_lanes(self._h, _ptr(d_cts), count, _ptr(out)))
        return out

    def ct_from_lanes(self, d_lanes, count, out=None):
        p = self.params
        out = self.empty(count * p.ct_limbs * 8) if out is None else out
        self._chk(self.lib.mfh_ct_from_lanes(self._h, _ptr(d_lanes), count, _ptr(out)))
        return out

    def add_dotp(self, rop, a, b, length):
        """mpz_add_dotp (src/lwe.c:20-28) on device values"""
        self._chk(self.lib.mfh_add_dotp(self._h, _ptr(rop), _ptr(a), _ptr(b), length))
        return rop

    # -- resident (materialised) CRS ------------------------------------------------------------------------------
    def resident_row_bytes(self):
        return int(self.lib.mfh_resident_row_bytes(self._h))

    def crs_expand(self, off, nrows, c8, out=None):
        """expand rows (a-vectors from the stream + b from c8) once into the streaming layout"""
        out = self.empty(nrows * self.resident_row_bytes()) if out is None else out
        self._chk(self.lib.mfh_crs_expand(self._h, off, nrows, _ptr(c8), _ptr(out)))
        return out

    def crs_image_set_b(self, image, first_row, nrows, c8):
        """coordinate n (the b's) of rows [first_row, first_row + nrows) of a row image expanded with c8 = None"""
        self._chk(self.lib.mfh_crs_image_set_b(self._h, int(first_row), int(nrows), _ptr(c8), _ptr(image)))
        return image

    def eval_rows_resident(self, rows, first_row, nrows, coeff0, coeff1=None, rop0=None, rop1=None, accumulate=False):
        p = self.params
        if rop0 is None:
            rop0 = self.empty(p.ct_limbs * 8)
        if coeff1 is not None and rop1 is None:
            rop1 = self.empty(p.ct_limbs * 8)
        self._chk(self.lib.mfh_eval_rows_resident(self._h, _ptr(rows), first_row, nrows, _ptr(coeff0), _ptr(coeff1), _ptr(rop0), _ptr(rop1),
                                                  1 if accumulate else 0))
        return rop0, rop1

    def set_resident(self, rows):
        self._resident = rows  # keep the tensor alive
        self._chk(self.lib.mfh_crs_set_resident(self._h, _ptr(rows)))

    def witness_lanes(self, d_ssp, witness_bits: bytes, rank, world, out=None):
        out = self.torch.empty(self.params.d, dtype=self.torch.int64, device=self.device) if out is None else out
        self._chk(self.lib.mfh_witness_lanes(self._h, _ptr(d_ssp), bytes(witness_bits), rank, world, _ptr(out)))
        return out

    def prove_partial_w(self, d_crs, d_ssp, witness_bits: bytes, delta, rank, world, wlanes, out=None):
        p = self.params
        out = self.empty(5 * p.ct_limbs * 8) if out is None else out
        self._chk(self.lib.mfh_prove_partial_w(self._h, _ptr(d_crs), _ptr(d_ssp), bytes(witness_bits), delta, rank, world, _ptr(wlanes), _ptr(out)))
        return out

    def verify(self, d_ssp, alpha, beta, s, d_sk, d_proofs, count=1):
        """verifier() (src/snark.c:192-250) for `count` proofs on the device; returns a uint8 tensor of accept bits"""
        ok = self.empty(count)
        self._chk(self.lib.mfh_verify(self._h, _ptr(d_ssp), alpha, beta, s, _ptr(d_sk), _ptr(d_proofs), count, _ptr(ok)))
        return ok

    # -- generator-defined SSP (BASELINE configs 4/5): pass d_ssp=None to the SSP-consuming calls afterwards ---------
    def ssp_prg_make_t(self, seed64, witness_bits: bytes):
        t = self.empty(self.params.d * 4)
        self._chk(self.lib.mfh_ssp_prg_make_t(self._h, seed64, bytes(witness_bits), _ptr(t)))
        return t

    def ssp_set_prg(self, seed64, d_t):
        self._prg_t = d_t  # keep alive
        self._chk(self.lib.mfh_ssp_set_prg(self._h, seed64, _ptr(d_t)))

    def ssp_prg_fill(self, seed64, first_slot, nslots):
        out = self.empty(nslots * self.params.d * 4)
        self._chk(self.lib.mfh_ssp_prg_fill(self._h, seed64, first_slot, nslots, _ptr(out)))
        return out

    # -- sharded resident CRS (one share per rank) -------------------------------------------------------------------
    def crs_expand_share(self, d_crs, rank, world, out=None):
        rows = int(self.lib.mfh_resident_share_rows(self._h, rank, world))
        out = self.empty(rows * self.resident_row_bytes()) if out is None else out
        self._chk(self.lib.mfh_crs_expand_share(self._h, _ptr(d_crs), rank, world, _ptr(out)))
        return out

    def set_resident_share(self, image, rank, world):
        self._resident = image
        self._chk(self.lib.mfh_crs_set_resident_share(self._h, _ptr(image), rank, world))

    def set_resident_prefix(self, image, nrows_resident):
        """only stream rows [0, nrows_resident) are resident; the rest is regenerated"""
        self._resident = image
        self._chk(self.lib.mfh_crs_set_resident_prefix(self._h, _ptr(image), nrows_resident))
