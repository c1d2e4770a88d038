"""ctypes binding of libechoseal_hip.so (C ABI: include/echoseal_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, an
exception is raised.  The hot path only ever runs on the HIP kernels.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  -- must be imported BEFORE the HIP library so that both share torch's HIP runtime
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libechoseal_hip.so")
if os.environ.get("ES_LIB_VARIANT"):           # development: an A/B build made by tools/build_variant.sh (same ABI check, same loud failure when missing)
    LIB_PATH = os.path.join(_HERE, f"libechoseal_hip_{os.environ['ES_LIB_VARIANT']}.so")

ES_ABI_VERSION = 2          # include/echoseal_hip.h; load() refuses a library built from another one
ES_FRAME_LEN = 1215
ES_PRE_L = 63
ES_NBANDS = 4
ES_MAX_TAPS = 576
ES_MAX_TAPS_FAST = 160
ES_MAX_PEAKS = 32
ES_MAX_LIST = 256
ES_PN_BYTES = 152
ES_INFO_BYTES = 55
ES_DTYPE_F32, ES_DTYPE_I16, ES_DTYPE_F64 = 0, 1, 2

# name -> (restype, argtypes); kept next to the header so a test can check both agree
SIGNATURES = {
    "es_create": (c_void_p, [c_int, c_int]),
    "es_destroy": (None, [c_void_p]),
    "es_last_error": (c_char_p, [c_void_p]),
    "es_abi_version": (c_int, []),
    "es_info_bytes": (c_int, [c_void_p]),
    "es_set_tables": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "es_bpf_batch": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "es_bpf2_batch": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "es_xcorr32_batch": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "es_pick_exact_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p]),
    "es_sync_fused_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p]),
    "es_front_batch": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "es_reserve": (c_int, [c_void_p, c_int64, c_int]),
    "es_xcorr_batch": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "es_pick_batch": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "es_sync_batch": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_void_p]),
    "es_llr_batch": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int,
                             c_void_p, c_void_p, c_void_p, c_void_p]),
    "es_header_batch": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p]),
    "es_scl_batch": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "es_polar_encode_batch": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "es_schedule_batch": (c_int, [c_void_p, c_char_p, c_char_p, c_void_p, ctypes.c_uint32, c_int64, c_void_p, c_void_p, c_void_p]),
    "es_tx_frames_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_char_p, c_char_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "es_resample_batch": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    "es_set_option": (c_int, [c_void_p, c_char_p, c_int]),
    "es_softplus_batch": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "es_aead_check_batch": (c_int, [c_void_p, c_char_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "es_aead_seal_batch": (c_int, [c_void_p, c_char_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "es_select_batch": (c_int, [c_void_p, c_char_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the HIP library (once).  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C echoseal_amd/csrc`.  There is no CPU fallback for the EchoSeal hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    try:
        lib.es_abi_version.restype = c_int
        got = int(lib.es_abi_version())
    except AttributeError:
        got = None
    if got != ES_ABI_VERSION:
        raise NativeError(f"{LIB_PATH} reports ABI version {got}, this package binds version {ES_ABI_VERSION} "
                          "(table strides and signatures differ between versions): rebuild it with `make -C echoseal_amd/csrc`")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx, rc: int, what: str) -> None:
    if rc != 0:
        msg = load().es_last_error(ctx)
        raise NativeError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
