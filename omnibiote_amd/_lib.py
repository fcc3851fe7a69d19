"""ctypes binding of libomnibiote_hip.so (C ABI: include/omnibiote_hip.h).

Loaded lazily and exactly once; a missing or unloadable library is a hard error (there is no fallback path).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OBTE_LIB_PATH") or os.path.join(_HERE, "libomnibiote_hip.so")   # override: A/B builds

c_bf16_p = C.c_void_p
c_f32_p = C.c_void_p
c_stream = C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("d", C.c_void_p), ("aux", C.c_void_p), ("d2", C.c_void_p),
                ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldd", C.c_int64),
                ("a_kmajor", C.c_int32), ("b_kmajor", C.c_int32), ("epilogue", C.c_int32), ("alpha", C.c_float),
                ("dropout_p", C.c_float), ("dropout_site", C.c_int32), ("dropout_seed", C.c_uint64),
                ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p), ("rope_T", C.c_int64), ("rope_head_dim", C.c_int32)]


class AttnFwdArgs(C.Structure):
    _fields_ = [("qkv", C.c_void_p), ("o", C.c_void_p), ("lse", C.c_void_p),
                ("key_ranges", C.c_void_p), ("mask", C.c_void_p),
                ("mask_sb", C.c_int64), ("mask_sh", C.c_int64), ("mask_sq", C.c_int64),
                ("B", C.c_int64), ("T", C.c_int64), ("n_head", C.c_int32), ("head_dim", C.c_int32), ("scale", C.c_float),
                ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64), ("ranges_exact", C.c_void_p), ("drop_bits", C.c_void_p)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("qkv", C.c_void_p), ("o", C.c_void_p), ("d_o", C.c_void_p), ("lse", C.c_void_p),
                ("delta", C.c_void_p), ("dqkv", C.c_void_p), ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
                ("key_ranges", C.c_void_p), ("mask", C.c_void_p),
                ("mask_sb", C.c_int64), ("mask_sh", C.c_int64), ("mask_sq", C.c_int64),
                ("B", C.c_int64), ("T", C.c_int64), ("n_head", C.c_int32), ("head_dim", C.c_int32), ("scale", C.c_float),
                ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64), ("query_bounds", C.c_void_p), ("ranges_exact", C.c_void_p),
                ("ws", C.c_void_p), ("ws_bytes", C.c_int64), ("drop_bits", C.c_void_p)]


class BlockDesc(C.Structure):
    _fields_ = [("B", C.c_int64), ("T", C.c_int64), ("n_embd", C.c_int32), ("n_head", C.c_int32),
                ("ln1_w", C.c_void_p), ("attn_w", C.c_void_p), ("proj_w", C.c_void_p),
                ("ln2_w", C.c_void_p), ("fc_w", C.c_void_p), ("mlp_w", C.c_void_p),
                ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
                ("key_ranges", C.c_void_p), ("mask", C.c_void_p),
                ("mask_sb", C.c_int64), ("mask_sh", C.c_int64), ("mask_sq", C.c_int64),
                ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64), ("query_bounds", C.c_void_p),
                ("ln1_partials", C.c_void_p), ("ln2_partials", C.c_void_p), ("ln_partial_mode", C.c_int32),
                ("ranges_exact", C.c_void_p), ("out_rows", C.c_void_p), ("n_out_rows", C.c_int64),
                ("dy_masked", C.c_void_p), ("dx_masked", C.c_void_p), ("dx_mask_seed", C.c_uint64)]


LN_PARTIAL_FIRST, LN_PARTIAL_MORE, LN_PARTIAL_LAST = 1, 2, 3
MT_MAX = 32


class MtArgs(C.Structure):
    _fields_ = [("p", C.c_void_p * MT_MAX), ("g", C.c_void_p * MT_MAX), ("m", C.c_void_p * MT_MAX), ("v", C.c_void_p * MT_MAX),
                ("n", C.c_int64 * MT_MAX), ("lr", C.c_float * MT_MAX), ("weight_decay", C.c_float * MT_MAX),
                ("step", C.c_int32 * MT_MAX), ("count", C.c_int32),
                ("lr64", C.c_double * MT_MAX), ("weight_decay64", C.c_double * MT_MAX)]


EPI_NONE, EPI_GELU, EPI_ADD, EPI_GELU_BWD, EPI_ADD_DROPOUT, EPI_ROPE_QK = 0, 1, 2, 3, 4, 5
SITE_EMBED, SITE_ATTN, SITE_RESID, SITE_MLP, SITE_USER = 0, 1, 2, 3, 7

# name -> (restype, argtypes); every symbol include/omnibiote_hip.h declares
SYMBOLS = {
    "obte_abi_version": (C.c_int, []),
    "obte_last_error": (C.c_char_p, []),
    "obte_struct_sizes": (C.c_int, [C.c_void_p, C.c_int]),
    "obte_device_status": (C.c_int, [C.c_int]),
    "obte_fault_inject": (C.c_int, [C.c_int]),
    "obte_profile_enable": (C.c_int, [C.c_int]),
    "obte_profile_collect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "obte_layernorm_fwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_float, c_stream]),
    "obte_layernorm_bwd_ws_rows": (C.c_int, []),
    "obte_layernorm_bwd": (C.c_int, [C.c_void_p] * 9 + [C.c_int64, C.c_int, c_stream]),
    "obte_layernorm_bwd_acc": (C.c_int, [C.c_void_p] * 9 + [C.c_int64, C.c_int, C.c_int, c_stream]),
    "obte_layernorm_bwd_partial": (C.c_int, [C.c_void_p] * 9 + [C.c_int64, C.c_int, C.c_int, c_stream]),
    "obte_layernorm_bwd_dropout": (C.c_int, [C.c_void_p] * 10 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_int32, c_stream]),
    "obte_gemm_bf16": (C.c_int, [C.POINTER(GemmArgs), c_stream]),
    "obte_gemm_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64]),
    "obte_gemm_bf16_ws": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p, C.c_int64, c_stream]),
    "obte_gemm_grouped_bf16": (C.c_int, [C.POINTER(GemmArgs), C.c_int, c_stream]),
    "obte_gemm_plan_set": (C.c_int, [C.c_int] * 3 + [C.c_int64] * 3 + [C.c_int] * 3),
    "obte_gemm_plan_clear": (C.c_int, []),
    "obte_gemm_workspace_bytes_max": (C.c_int64, [C.c_int64] * 3),
    "obte_rope_qk_inplace": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, c_stream]),
    "obte_attn_fwd": (C.c_int, [C.POINTER(AttnFwdArgs), c_stream]),
    "obte_mask_bounds": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, c_stream]),
    "obte_attn_bwd": (C.c_int, [C.POINTER(AttnBwdArgs), c_stream]),
    "obte_attn_drop_bits_bytes": (C.c_int64, [C.c_int64, C.c_int64, C.c_int32]),
    "obte_attn_bwd_ws_bytes": (C.c_int64, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "obte_attn_bwd_select": (C.c_int, [C.c_int]),
    "obte_embedding_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_int64, c_stream]),
    "obte_embedding_fwd_dropout": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_int64, C.c_float, C.c_uint64, c_stream]),
    "obte_embedding_bwd_dropout": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_uint64, c_stream]),
    "obte_dropout_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_uint64, C.c_int32, c_stream]),
    "obte_embedding_bwd_ws_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "obte_embedding_bwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int64, c_stream]),
    "obte_embedding_bwd_acc": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int64, C.c_int, c_stream]),
    "obte_masked_ce_fwd_bwd": (C.c_int, [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 3 + [C.c_int64, C.c_int64, c_stream]),
    "obte_masked_ce_fwd_bwd_reuse": (C.c_int, [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 2 + [C.c_int64, C.c_int64, c_stream]),
    "obte_rows_gather_bf16": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int64, C.c_int32, c_stream]),
    "obte_rows_scatter_bf16": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int64, C.c_int32, c_stream]),
    "obte_masked_ce_rows": (C.c_int, [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 3 + [C.c_int64, C.c_int64, C.c_int64, c_stream]),
    "obte_adamw_bf16": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] + [C.c_float] * 5 + [C.c_int32, C.c_void_p, c_stream]),
    "obte_sumsq_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, c_stream]),
    "obte_adamw_multi_bf16": (C.c_int, [C.POINTER(MtArgs), C.c_float, C.c_float, C.c_float, C.c_void_p, c_stream]),
    "obte_sumsq_multi_bf16": (C.c_int, [C.POINTER(MtArgs), C.c_void_p, c_stream]),
    "obte_sumsq_multi_bf16_each": (C.c_int, [C.POINTER(MtArgs), C.c_void_p, c_stream]),
    "obte_adamw_multi_bf16_ref": (C.c_int, [C.POINTER(MtArgs), C.c_double, C.c_double, C.c_double, C.c_void_p, c_stream]),
    "obte_block_act_bytes": (C.c_int64, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "obte_block_act_bytes_p": (C.c_int64, [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_float]),
    "obte_block_bwd_ws_bytes": (C.c_int64, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "obte_block_fwd": (C.c_int, [C.POINTER(BlockDesc), C.c_void_p, C.c_void_p, C.c_void_p, c_stream]),
    "obte_block_bwd": (C.c_int, [C.POINTER(BlockDesc)] + [C.c_void_p] * 11 + [c_stream]),
    "obte_block_bwd_acc": (C.c_int, [C.POINTER(BlockDesc)] + [C.c_void_p] * 11 + [C.c_int, c_stream]),
}

_lib = None
_lock = threading.Lock()


class HipLibraryError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises HipLibraryError if it is missing: build it with
    ``python -c 'import __graft_entry__ as g; g.build()'`` (or ``make -C omnibiote_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            # PyTorch-ROCm ships its own libamdhip64; load it first so that this library binds to the SAME HIP
            # runtime instance that owns torch's streams and allocations (two runtimes in one process do not
            # share devices, streams or pointers).
            import torch  # noqa: F401
            if not os.path.exists(LIB_PATH):
                raise HipLibraryError(f"{LIB_PATH} not found: the HIP library has not been built "
                                      "(make -C omnibiote_amd/csrc). There is no CPU fallback.")
            try:
                l = C.CDLL(LIB_PATH)
            except OSError as e:
                raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
            for name, (res, args) in SYMBOLS.items():
                try:
                    fn = getattr(l, name)
                except AttributeError as e:
                    raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
                fn.restype = res
                fn.argtypes = args
            sizes = (C.c_int64 * 8)()
            n = l.obte_struct_sizes(sizes, 8)
            mine = [C.sizeof(GemmArgs), C.sizeof(AttnFwdArgs), C.sizeof(AttnBwdArgs), C.sizeof(MtArgs), C.sizeof(BlockDesc)]
            if n != len(mine) or list(sizes[:n]) != mine:
                raise HipLibraryError(f"struct layout mismatch between _lib.py {mine} and {LIB_PATH} {list(sizes[:n])}")
            _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().obte_last_error()
        raise RuntimeError(f"{what or 'libomnibiote_hip'} failed (code {rc}): {msg.decode() if msg else ''}")


STATUS_ATTN_BWD_HANDOFF = 1


class DeviceStatusError(RuntimeError):
    """A kernel reported, through the library's device status word, that a launch's results are invalid."""


def check_device_status(what: str = "") -> None:
    """Raise if any kernel has reported a failure since the last check (include/omnibiote_hip.h, obte_device_status).  A plain host
    read of pinned memory: meaningful for work the caller has already synchronised with — call it right after a `.item()`, a
    stream / device synchronise or an event wait.  The bits are cleared, so one failure is raised once."""
    v = lib().obte_device_status(1)
    if v != 0:
        msg = lib().obte_last_error()
        raise DeviceStatusError(f"{what + ': ' if what else ''}{msg.decode() if msg else f'device status {v}'}")
