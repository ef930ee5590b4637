"""ctypes binding of libotter_hip.so (include/otter_hip.h).

There is no CPU fallback: if the library is missing, or a call fails, the
caller gets an exception (contrast the swallow-and-continue `except` of the
reference's predict(), scripts/encode_corpus.py:138-143).
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OTTER_LIB", os.path.join(_PKG, "libotter_hip.so"))

OTTER_OK = 0
OTTER_EINVAL, OTTER_ENOMEM, OTTER_EHIP, OTTER_ESTATE, OTTER_ENOTIMPL = -1, -2, -3, -4, -5
OTTER_F32, OTTER_BF16, OTTER_F16, OTTER_I64 = 0, 1, 2, 3
OTTER_MAX_K = 2048
MERGE_MAX = 8192

POOL = {"mean": 0, "sum": 1, "max": 2, "cls": 3, "first": 4, "att_self_pool": 5}
POS_BERT, POS_ROBERTA = 0, 1

c_i64 = ctypes.c_int64
c_vp = ctypes.c_void_p
c_int = ctypes.c_int


class EncoderCfg(ctypes.Structure):
    _fields_ = [("vocab_size", ctypes.c_int32), ("hidden_size", ctypes.c_int32),
                ("num_hidden_layers", ctypes.c_int32), ("num_attention_heads", ctypes.c_int32),
                ("intermediate_size", ctypes.c_int32), ("max_position_embeddings", ctypes.c_int32),
                ("type_vocab_size", ctypes.c_int32), ("pad_token_id", ctypes.c_int32),
                ("position_mode", ctypes.c_int32), ("three_cat", ctypes.c_int32),
                ("no_proj", ctypes.c_int32), ("part_pooling", ctypes.c_int32),
                ("layer_norm_eps", ctypes.c_float)]


# name -> (restype, argtypes); every symbol include/otter_hip.h declares
SIGNATURES = {
    "otter_last_error": (ctypes.c_char_p, []),
    "otter_version": (c_int, []),
    "otter_device_count": (c_int, []),
    "otter_index_create": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_vp)]),
    "otter_index_add": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp]),
    "otter_index_search": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_int, c_vp]),
    "otter_index_search_noalloc": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "otter_index_ntotal": (c_i64, [c_vp]),
    "otter_index_d": (c_int, [c_vp]),
    "otter_index_reset": (c_int, [c_vp]),
    "otter_index_set_id_base": (c_int, [c_vp, c_i64]),
    "otter_index_coarse_state": (c_int, [c_vp]),
    "otter_index_set_planes": (c_int, [c_vp, c_int]),
    "otter_index_last_fallback": (c_i64, [c_vp]),
    "otter_index_last_refined": (c_i64, [c_vp]),
    "otter_index_destroy": (None, [c_vp]),
    "otter_index_set_timing": (c_int, [c_vp, c_int]),
    "otter_index_timing": (c_int, [c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)]),
    "otter_sharded_search": (c_int, [c_vp, c_int, c_vp, c_i64, c_int, c_vp, c_vp, c_int, c_vp]),
    "otter_merge_topk": (c_int, [c_int, c_vp, c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_vp]),
    "otter_inbatch_scores": (c_int, [c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "otter_encoder_create": (c_int, [ctypes.POINTER(EncoderCfg), c_int, ctypes.POINTER(c_vp)]),
    "otter_encoder_set_weight": (c_int, [c_vp, ctypes.c_char_p, c_vp, ctypes.POINTER(c_i64), c_int]),
    "otter_encoder_finalize": (c_int, [c_vp, c_int]),
    "otter_encoder_forward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "otter_encoder_forward_hidden": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "otter_gemm_bf16": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "otter_index_reconstruct_n": (c_int, [c_vp, c_i64, c_i64, c_vp, c_int, c_vp]),
    "otter_index_search_subset": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_vp]),
    "otter_ln_fold": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "otter_gemm_bf16_ln": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                                   c_int, c_int, c_vp]),
    "otter_ln_stats": (c_int, [c_int, c_vp, c_int, c_i64, c_int, ctypes.c_float, c_vp, c_vp]),
    "otter_encoder_out_dim": (c_int, [c_vp]),
    "otter_encoder_destroy": (None, [c_vp]),
}

_lib = None


class OtterError(RuntimeError):
    pass


def load():
    """Load the library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OtterError("libotter_hip.so not found at %s: build it with `python -m otter_amd.build` "
                         "(there is no CPU fallback)" % LIB_PATH)
    # Load order matters: PyTorch-ROCm ships its own HIP runtime.  If this library pulled in the
    # system runtime first, a later `import torch` would bring a second runtime into the process
    # and one of the two then sees no device.  Importing torch first makes both share one runtime.
    try:
        import torch  # noqa: F401
    except Exception:  # numpy-only callers: the system runtime is the only one
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    """Turn a negative return code into an exception carrying otter_last_error()."""
    if rc == OTTER_OK:
        return
    msg = load().otter_last_error()
    msg = msg.decode("utf-8", "replace") if msg else ""
    text = "%s failed (%d): %s" % (what or "libotter_hip call", rc, msg)
    if rc == OTTER_EINVAL:
        raise ValueError(text)
    if rc == OTTER_ENOMEM:
        raise MemoryError(text)
    if rc == OTTER_ENOTIMPL:
        raise NotImplementedError(text)
    raise OtterError(text)
