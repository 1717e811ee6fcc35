"""ctypes binding of librfa_hip.so — field-for-field mirror of include/rfa.h.

The library is the ONLY compute path of this package: if it cannot be loaded, every operator
raises (there is no CPU or PyTorch fallback on purpose — see DESIGN.md "fail loudly").
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RFA_LIB_PATH: A/B tooling only (tools/ab_variants.py builds tuning variants of the same library)
LIB_PATH = os.environ.get("RFA_LIB_PATH") or os.path.join(_HERE, "librfa_hip.so")

RFA_ABI_VERSION = 6
RFA_BF16, RFA_F16 = 0, 1
HALF_FULL, HALF_FRONT, HALF_BACK = 0, 1, 2
BWD_ALL, BWD_COMPUTE, BWD_REDUCE = 0, 1, 2
BWD_SKIP_DKDV, BWD_SKIP_DQ = 4, 8
BWD_KV_OVERWRITE = 16     # dk_acc / dv_acc are overwritten (dq_acc still follows acc_init)
DKDV_AUTO, DKDV_128, DKDV_256, DKDV_BAL = 0, 1, 2, 3
FWD_AUTO, FWD_8x32, FWD_4x32, FWD_P8x32 = 0, 1, 3, 4        # (2: a retired experiment, RFA_ERR_ARGS)


class Strides(C.Structure):
    _fields_ = [("batch", C.c_int64), ("row", C.c_int64), ("head", C.c_int64)]


class FwdArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p),
        ("q_st", Strides), ("k_st", Strides), ("v_st", Strides),
        ("out", C.c_void_p), ("out_st", Strides),
        ("lse", C.c_void_p), ("lse_batch", C.c_int64), ("lse_head", C.c_int64),
        ("out_acc", C.c_void_p), ("out_acc_st", Strides),
        ("lse_acc", C.c_void_p), ("lse_acc_batch", C.c_int64), ("lse_acc_head", C.c_int64),
        ("acc_init", C.c_int32),
        ("cu_seqlens_q", C.c_void_p), ("cu_seqlens_k", C.c_void_p),
        ("q_half", C.c_int32), ("k_half", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("Hk", C.c_int32), ("D", C.c_int32),
        ("Sq", C.c_int32), ("Sk", C.c_int32),
        ("softmax_scale", C.c_float),
        ("causal", C.c_int32),
        ("dtype", C.c_int32),
        ("window", C.c_int32), ("window_left", C.c_int32), ("window_right", C.c_int32),
        ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
        ("q_pos_offset", C.c_int64), ("k_pos_offset", C.c_int64), ("head_offset", C.c_int32),
        ("fwd_form", C.c_int32),
        ("workspace", C.c_void_p), ("kv_nsplit", C.c_int32), ("total_q", C.c_int64),
    ]


class BwdPreArgs(C.Structure):
    _fields_ = [
        ("dout", C.c_void_p), ("out", C.c_void_p),
        ("dout_st", Strides), ("out_st", Strides),
        ("delta", C.c_void_p), ("delta_batch", C.c_int64), ("delta_head", C.c_int64),
        ("cu_seqlens_q", C.c_void_p),
        ("q_half", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("Sq", C.c_int32),
        ("dtype", C.c_int32),
    ]


class BwdArgs(C.Structure):
    _fields_ = [
        ("dout", C.c_void_p), ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p),
        ("dout_st", Strides), ("q_st", Strides), ("k_st", Strides), ("v_st", Strides),
        ("lse", C.c_void_p), ("lse_batch", C.c_int64), ("lse_head", C.c_int64),
        ("delta", C.c_void_p), ("delta_batch", C.c_int64), ("delta_head", C.c_int64),
        ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p),
        ("dq_st", Strides), ("dk_st", Strides), ("dv_st", Strides),
        ("dq_acc", C.c_void_p), ("dk_acc", C.c_void_p), ("dv_acc", C.c_void_p),
        ("dq_acc_st", Strides), ("dk_acc_st", Strides), ("dv_acc_st", Strides),
        ("acc_init", C.c_int32),
        ("workspace", C.c_void_p),
        ("cu_seqlens_q", C.c_void_p), ("cu_seqlens_k", C.c_void_p),
        ("q_half", C.c_int32), ("k_half", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("Hk", C.c_int32), ("D", C.c_int32),
        ("Sq", C.c_int32), ("Sk", C.c_int32),
        ("total_k", C.c_int64),
        ("softmax_scale", C.c_float),
        ("causal", C.c_int32),
        ("deterministic", C.c_int32),
        ("dtype", C.c_int32),
        ("phases", C.c_int32),
        ("ds_scratch", C.c_void_p),
        ("window", C.c_int32), ("window_left", C.c_int32), ("window_right", C.c_int32),
        ("dkdv_form", C.c_int32), ("dkdv_nsplit", C.c_int32),
        ("prof_events", C.POINTER(C.c_void_p)),
        ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
        ("q_pos_offset", C.c_int64), ("k_pos_offset", C.c_int64), ("head_offset", C.c_int32),
        ("ds_scratch_bytes", C.c_int64),
        ("total_q", C.c_int64),
    ]


class MergeArgs(C.Structure):
    _fields_ = [
        ("out_acc", C.c_void_p), ("out_acc_st", Strides),
        ("lse_acc", C.c_void_p), ("lse_acc_batch", C.c_int64), ("lse_acc_head", C.c_int64),
        ("block_out", C.c_void_p), ("block_out_st", Strides),
        ("block_lse", C.c_void_p), ("block_lse_batch", C.c_int64), ("block_lse_head", C.c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("S", C.c_int32),
        ("acc_init", C.c_int32),
        ("dtype", C.c_int32),
        ("lse_acc_row", C.c_int64), ("block_lse_row", C.c_int64),
    ]


class SumSlotsArgs(C.Structure):
    _fields_ = [
        ("src", C.c_void_p), ("slot_stride", C.c_int64), ("nslots", C.c_int32),
        ("dst", C.c_void_p),
        ("src_st", Strides), ("dst_st", Strides),
        ("B", C.c_int32), ("S", C.c_int32), ("H", C.c_int32), ("D", C.c_int32),
        ("dtype", C.c_int32),
    ]


# every symbol include/rfa.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "rfa_abi_version": (C.c_int, []),
    "rfa_build_id": (C.c_char_p, []),
    "rfa_strerror": (C.c_char_p, [C.c_int]),
    "rfa_fwd": (C.c_int, [C.POINTER(FwdArgs), C.c_void_p]),
    "rfa_fwd_workspace_bytes": (C.c_int64, [C.POINTER(FwdArgs), C.POINTER(C.c_int32)]),
    "rfa_bwd_preprocess": (C.c_int, [C.POINTER(BwdPreArgs), C.c_void_p]),
    "rfa_bwd_workspace_bytes": (C.c_int64, [C.POINTER(BwdArgs)]),
    "rfa_bwd_ds_scratch_bytes": (C.c_int64, [C.POINTER(BwdArgs)]),
    "rfa_bwd_ds_scratch_min_bytes": (C.c_int64, [C.POINTER(BwdArgs)]),
    "rfa_bwd_ds_chunks": (C.c_int, [C.POINTER(BwdArgs), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int64)]),
    "rfa_bwd_plan": (C.c_int, [C.POINTER(BwdArgs), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rfa_bwd": (C.c_int, [C.POINTER(BwdArgs), C.c_void_p]),
    "rfa_merge": (C.c_int, [C.POINTER(MergeArgs), C.c_void_p]),
    "rfa_sum_slots": (C.c_int, [C.POINTER(SumSlotsArgs), C.c_void_p]),
    "rfa_cast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "rfa_lse_flatten": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int64, C.c_int64, C.c_void_p]),
    "rfa_lse_unflatten": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int64, C.c_int64, C.c_void_p]),
}

_lib = None


def load():
    """Load librfa_hip.so once; raise RuntimeError (never fall back) when it is unusable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"ring_flash_attn: HIP extension not built ({LIB_PATH} missing). "
            "Run `python ring-flash-attention_amd/build.py lib` (needs hipcc, ROCm >= 7). "
            "There is no CPU fallback."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing libamdhip64 etc.
        raise RuntimeError(f"ring_flash_attn: cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    ver = lib.rfa_abi_version()
    if ver != RFA_ABI_VERSION:
        raise RuntimeError(f"ring_flash_attn: librfa_hip.so ABI {ver} != binding {RFA_ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().rfa_strerror(rc).decode()
        raise RuntimeError(f"{what} failed: {msg} (rfa status {rc})")
