"""ctypes binding of the C ABI in include/seg_hip.h (libseg_hip.so, built in-tree by build_ext.py).

There is deliberately no fallback: if the shared library is missing or a symbol cannot be resolved the
import of the *engine* fails loudly (graph construction does not need this module)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEG_LIB=<path>: another build of the library (A/B runs and bisecting - tools/ab_bench.sh - select a build this way instead
# of overwriting the in-tree file)
LIB_PATH = os.environ.get("SEG_LIB") or os.path.join(_HERE, "libseg_hip.so")

MAX_SEG, MAX_TAPS, MAX_CLS = 8, 32, 4
SEG_VARIANT_A_HI = SEG_VARIANT_IN_HI = 0x1000          # include/seg_hip.h
SEG_VARIANT_DZ_HI = SEG_VARIANT_B_HI = 0x2000
ACT = {None: 0, "none": 0, "linear": 0, "relu": 1, "leaky_relu": 2, "sigmoid": 3, "softmax": 4}
LOSS = {"mse": 0, "bce": 1, "cce": 2, "mae": 3, "mape": 4, "msle": 5, "huber": 6, "logcosh": 7, "poisson": 8, "hinge": 9,
        "squared_hinge": 10, "binary_focal": 11, "kld": 12, "categorical_hinge": 13, "cosine": 14, "sparse_cce": 15}
# the names `TFOptimizers(name, lr).optimizer()` takes (2DCNN/utils/tf_optimizers.py:10-26); Adam has its own entry points
GRAD_F32, GRAD_BF16 = 0, 1
OPERAND_F16, OPERAND_SPLIT = 0, 1
OPTIMIZER = {"SGD": 0, "RMSprop": 1, "Adagrad": 2, "Adadelta": 3, "Adamax": 4, "Nadam": 5, "FTRL": 6}
# the names `TFLosses(name).loss()` takes (2DCNN/utils/tf_losses.py:9-39)
LOSS_ALIASES = {"MeanSquaredError": "mse", "BinaryCrossentropy": "bce", "CategoricalCrossentropy": "cce",
                "MeanAbsoluteError": "mae", "MeanAbsolutePercentageError": "mape", "MeanSquaredLogarithmicError": "msle",
                "Huber": "huber", "LogCosh": "logcosh", "Poisson": "poisson", "Hinge": "hinge", "SquaredHinge": "squared_hinge",
                "BinaryFocalCrossentropy": "binary_focal", "KLDivergence": "kld", "CategoricalHinge": "categorical_hinge",
                "CosineSimilarity": "cosine", "SparseCategoricalCrossentropy": "sparse_cce"}
for _long, _short in LOSS_ALIASES.items():
    LOSS[_long] = LOSS[_short]


class Geom(C.Structure):
    _fields_ = [("N", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Hl", C.c_int32), ("Wl", C.c_int32),
                ("isy", C.c_int32), ("isx", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
                ("osy", C.c_int32), ("osx", C.c_int32), ("ncls", C.c_int32), ("ntaps", C.c_int32),
                ("T_total", C.c_int32), ("oy", C.c_int32 * MAX_CLS), ("ox", C.c_int32 * MAX_CLS),
                ("tap_dy", C.c_int32 * MAX_TAPS), ("tap_dx", C.c_int32 * MAX_TAPS), ("tap_w", C.c_int32 * MAX_TAPS)]


class IgemmFwdArgs(C.Structure):
    _fields_ = [("g", Geom), ("nseg", C.c_int32), ("seg_c", C.c_int32 * MAX_SEG), ("seg_cs", C.c_int32 * MAX_SEG),
                ("seg_woff", C.c_int32 * MAX_SEG), ("seg_ptr", C.c_void_p * MAX_SEG), ("w", C.c_void_p),
                ("Kpad", C.c_int32), ("n_rows", C.c_int32), ("n_valid", C.c_int32), ("out", C.c_void_p),
                ("out_cs", C.c_int32), ("out_coff", C.c_int32), ("bias", C.c_void_p), ("act", C.c_int32),
                ("alpha", C.c_float), ("accumulate", C.c_int32), ("stats", C.c_void_p), ("tile_m", C.c_int32),
                ("tile_n", C.c_int32), ("nout", C.c_int32), ("out_span", C.c_int32 * MAX_SEG), ("out_cs_o", C.c_int32 * MAX_SEG),
                ("out_acc", C.c_int32 * MAX_SEG), ("out_ptr", C.c_void_p * MAX_SEG), ("out_y", C.c_void_p * MAX_SEG),
                ("out_dact", C.c_int32 * MAX_SEG), ("out_dalpha", C.c_float * MAX_SEG), ("variant", C.c_int32), ("ep_scale", C.c_void_p),
                ("splitk", C.c_int32), ("partial", C.c_void_p)]


class IgemmWgradArgs(C.Structure):
    _fields_ = [("g", Geom), ("nseg", C.c_int32), ("seg_c", C.c_int32 * MAX_SEG), ("seg_cs", C.c_int32 * MAX_SEG),
                ("seg_woff", C.c_int32 * MAX_SEG), ("seg_ptr", C.c_void_p * MAX_SEG), ("dz", C.c_void_p),
                ("dz_cs", C.c_int32), ("dz_coff", C.c_int32), ("dw", C.c_void_p), ("Kpad", C.c_int32),
                ("n_rows", C.c_int32), ("splitk", C.c_int32), ("scale", C.c_float), ("scale_ptr", C.c_void_p),
                ("tile_m", C.c_int32), ("tile_n", C.c_int32), ("variant", C.c_int32), ("bias_part", C.c_void_p),
                ("overwrite", C.c_int32), ("partial", C.c_void_p)]


class TransposeDesc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wt", C.c_void_p), ("rows", C.c_int32), ("T", C.c_int32), ("K", C.c_int32),
                ("k0", C.c_int32), ("kcount", C.c_int32), ("rows_pad", C.c_int32), ("dst_ld", C.c_int32),
                ("dst_coff", C.c_int32), ("tile0", C.c_int32), ("src16", C.c_int32), ("dst_T", C.c_int32), ("dst_t0", C.c_int32)]


class CatPiece(C.Structure):
    _fields_ = [("x", C.c_void_p), ("dx", C.c_void_p), ("cs", C.c_int32), ("C", C.c_int32), ("coff", C.c_int32), ("acc", C.c_int32)]


class CatDesc(C.Structure):
    """seg_cat_desc: a virtual concatenation of up to four tensors (MultiResBlock's cat[c3, c5, c7])."""
    _fields_ = [("n", C.c_int32), ("p", CatPiece * 4)]


class LazyBN(C.Structure):
    """seg_lazy_bn: a Conv_Block normalised on load by a fused add (its BN-backward sums ride on the fan-out launch)."""
    _fields_ = [("z", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p),
                ("relu", C.c_int32), ("partial", C.c_void_p)]


class BnBwdSet(C.Structure):
    _fields_ = [("x", C.c_void_p), ("dx", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p),
                ("c1", C.c_void_p), ("c2", C.c_void_p), ("act", C.c_int32), ("alpha", C.c_float)]


class HeadFwdDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("w", C.c_void_p * 2), ("b", C.c_void_p * 2), ("act", C.c_int32 * 2), ("y", C.c_void_p * 2)]


class HeadInject(C.Structure):
    _fields_ = [("n", C.c_int32), ("dy", C.c_void_p * 2), ("y", C.c_void_p * 2), ("w", C.c_void_p * 2), ("act", C.c_int32 * 2), ("gs_ptr", C.c_void_p)]


i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p

_PROTOS = {
    "seg_abi_version": ([], C.c_int),
    "seg_igemm_fwd": ([C.POINTER(IgemmFwdArgs), vp], C.c_int),
    "seg_igemm_wgrad": ([C.POINTER(IgemmWgradArgs), vp], C.c_int),
    "seg_igemm_fwd_stats_floats": ([C.POINTER(IgemmFwdArgs)], C.c_int64),
    "seg_igemm_fwd_workspace_bytes": ([C.POINTER(IgemmFwdArgs)], C.c_int64),
    "seg_igemm_wgrad_workspace_bytes": ([C.POINTER(IgemmWgradArgs)], C.c_int64),
    "seg_wgrad_taps_per_block": ([C.c_int, C.c_int, C.c_int, C.c_int], C.c_int),
    "seg_wgrad_halo_patch": ([C.POINTER(Geom)], C.c_int),
    "seg_wgrad_halo_class_groups": ([C.POINTER(Geom)], C.c_int),
    "seg_lds_dma_operand_limit": ([], C.c_int64),
    "seg_tail_row_limit": ([], C.c_int32),
    "seg_fwd_halo_patch": ([C.POINTER(Geom)], C.c_int),
    "seg_bn_stats": ([vp, i64, i32, i32, i32, vp, i32, i32, i32, vp], C.c_int),
    "seg_bn_finalize": ([vp, i32, i32, i64, vp, vp, f32, f32, vp, vp, i32, i32, vp, vp, vp, vp, vp], C.c_int),
    "seg_bn_fold_inference": ([vp, vp, vp, vp, f32, vp, i32, vp, vp, vp], C.c_int),
    "seg_bn_act_apply": ([vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, f32, vp, i32, i32, vp], C.c_int),
    "seg_bn_act_apply_head": ([vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, f32, vp, i32, i32, C.POINTER(HeadFwdDesc), vp], C.c_int),
    "seg_bn_act_bwd_reduce_inj": ([vp, i32, i32, vp, i32, i32, i64, i32, vp, vp, vp, vp, i32, f32, vp, i32, C.POINTER(HeadInject), vp], C.c_int),
    "seg_bn_act_bwd_apply_inj": ([vp, i32, i32, vp, i32, i32, vp, i32, i32, i64, i32, vp, vp, vp, vp, vp, vp, i32, f32, i32, C.POINTER(HeadInject), vp], C.c_int),
    "seg_bn_act_bwd_apply2": ([vp, C.POINTER(BnBwdSet), i64, i32, vp], C.c_int),
    "seg_bn_act_bwd_reduce": ([vp, i32, i32, vp, i32, i32, i64, i32, vp, vp, vp, vp, i32, f32, vp, i32, vp], C.c_int),
    "seg_bn_bwd_finalize": ([vp, i32, i32, i64, vp, vp, f32, vp, vp, vp, vp], C.c_int),
    "seg_bn_act_bwd_apply": ([vp, i32, i32, vp, i32, i32, vp, i32, i32, i64, i32, vp, vp, vp, vp, vp, vp, i32, f32, i32, vp], C.c_int),
    "seg_act_bwd": ([vp, vp, vp, i64, i32, i32, i32, f32, vp, f32, vp, vp], C.c_int),
    "seg_maxpool_fwd": ([vp, vp, i32, i32, i32, i32, i32, i32, vp], C.c_int),
    "seg_maxpool_bwd": ([vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], C.c_int),
    "seg_upsample_fwd": ([vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], C.c_int),
    "seg_upsample_bwd": ([vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp], C.c_int),
    "seg_add_act_fwd": ([C.POINTER(vp), i32, vp, i64, i32, i32, i32, f32, vp], C.c_int),
    "seg_add_act_stats_fwd": ([C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), i32, C.POINTER(CatDesc), vp, vp, vp, i64, i32, i32, i32, f32, vp, i32, i32, i32, vp], C.c_int),
    "seg_bn_bwd_apply_fanout": ([vp, vp, C.POINTER(vp), i32, i32, i64, i32, vp, vp, vp, vp, vp, i32, C.POINTER(CatDesc), vp, vp, vp, C.POINTER(LazyBN), i32, i32, vp], C.c_int),
    "seg_cat_bn_bwd_scatter": ([vp, C.POINTER(CatDesc), i64, i32, vp, vp, vp, vp, vp, vp], C.c_int),
    "seg_add_act_bwd": ([vp, vp, C.POINTER(vp), i32, i64, i32, f32, i32, vp], C.c_int),
    "seg_copy_channels": ([vp, i32, i32, vp, i32, i32, i32, i64, i32, vp], C.c_int),
    "seg_mul_bcast_fwd": ([vp, vp, vp, i64, i32, i32, vp], C.c_int),
    "seg_mul_bcast_bwd": ([vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp], C.c_int),
    "seg_convlstm_gate_fwd": ([vp, i32, vp, i32, i64, i32, vp], C.c_int),
    "seg_convlstm_gate_bwd": ([vp, i32, vp, i32, vp, i64, i32, vp], C.c_int),
    "seg_head_fwd": ([vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, i32, vp, vp], C.c_int),
    "seg_head_bwd": ([vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, i32, vp, i32, vp, vp, vp, vp, i64, vp], C.c_int),
    "seg_head_act_fwd": ([vp, i32, i64, i32, i32, vp, vp], C.c_int),
    "seg_head_act_bwd": ([vp, vp, i64, i32, i32, vp, vp, i32, i32, vp], C.c_int),
    "seg_loss_fwd_bwd": ([i32, vp, vp, i64, i32, vp, vp, vp], C.c_int),
    "seg_loss_fwd_bwd_w": ([i32, vp, vp, i64, i32, C.c_float, vp, vp, vp, vp], C.c_int),
    "seg_labels_binary": ([vp, i64, vp, vp], C.c_int),
    "seg_optimizer_step": ([i32, vp, vp, vp, vp, vp, i64, vp, C.c_float, vp], C.c_int),
    "seg_binary_counts": ([vp, vp, i64, C.c_float, vp, vp], C.c_int),
    "seg_argmax_match": ([vp, vp, i64, i32, i32, vp, vp], C.c_int),
    "seg_absmax": ([vp, i64, vp, vp], C.c_int),
    "seg_grad_scale": ([vp, f32, vp, vp], C.c_int),
    "seg_adam_step": ([vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp], C.c_int),
    "seg_adam_step_dev": ([vp, vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, vp], C.c_int),
    "seg_adam_step_ex": ([vp, vp, i32, vp, vp, vp, i32, i64, f32, vp, f32, f32, f32, f32, vp, vp], C.c_int),
    "seg_optimizer_step_ex": ([i32, vp, vp, i32, vp, vp, vp, i32, i64, vp, C.c_float, vp, vp], C.c_int),
    "seg_cast_f32_bf16": ([vp, vp, i64, vp], C.c_int),
    "seg_transpose_w": ([vp, vp, i32, i32, i32, i32, vp], C.c_int),
    "seg_transpose_w_multi": ([vp, i32, i32, vp], C.c_int),
    "seg_cast_f32_f16": ([vp, vp, i64, vp], C.c_int),
    "seg_cast_f32_split": ([vp, vp, i64, vp], C.c_int),
    "seg_pack_input": ([vp, vp, i64, i32, i32, vp, vp], C.c_int),
    "seg_unpack_f32": ([vp, vp, i64, i32, i32, i32, vp], C.c_int),
}

# entry points that touch activation tensors exist twice: fp16 (throughput mode) and *_f32 (parity mode: split fp16 pairs, 4 B per element)
DUAL = ("seg_igemm_fwd", "seg_igemm_wgrad", "seg_bn_stats", "seg_bn_act_apply", "seg_bn_act_apply_head", "seg_bn_act_bwd_reduce_inj", "seg_bn_act_bwd_apply_inj", "seg_bn_act_bwd_apply2", "seg_bn_act_bwd_reduce", "seg_bn_act_bwd_apply",
        "seg_act_bwd", "seg_maxpool_fwd", "seg_maxpool_bwd", "seg_upsample_fwd", "seg_upsample_bwd", "seg_add_act_fwd",
        "seg_add_act_bwd", "seg_add_act_stats_fwd", "seg_bn_bwd_apply_fanout", "seg_cat_bn_bwd_scatter", "seg_mul_bcast_fwd", "seg_mul_bcast_bwd", "seg_convlstm_gate_fwd", "seg_convlstm_gate_bwd",
        "seg_head_fwd", "seg_head_bwd", "seg_head_act_fwd", "seg_head_act_bwd", "seg_transpose_w", "seg_transpose_w_multi", "seg_pack_input", "seg_unpack_f32", "seg_copy_channels")
for _n in DUAL:
    _PROTOS[_n + "_f32"] = _PROTOS[_n]

EXPORTED_SYMBOLS = tuple(_PROTOS) + ("seg_last_error",)

_lib = None


class SegHipError(RuntimeError):
    pass


def lib():
    """Load libseg_hip.so once.  Raises (never falls back) when the HIP extension is missing."""
    global _lib
    if _lib is None:
        # torch first: its wheel bundles a HIP runtime (libamdhip64); loading ours before it would pull in a second
        # copy from /opt/rocm and kernels would be launched on a runtime that never saw torch's device / streams
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (argt, rest) in _PROTOS.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing -> loud failure
            fn.argtypes = argt
            fn.restype = rest
        L.seg_last_error.argtypes = []
        L.seg_last_error.restype = C.c_char_p
        L.seg_wgrad_taps_per_block.restype = C.c_int
        L.seg_wgrad_halo_patch.restype = C.c_int
        from .build_ext import abi_version
        if os.environ.get("SEG_LIB"):
            print(f"[seg_hip] SEG_LIB: loading {LIB_PATH}", flush=True)
        have, want = L.seg_abi_version(), abi_version()
        if have != want:
            raise ImportError(f"{LIB_PATH}: built against another include/seg_hip.h (ABI {have:#x}, this tree {want:#x}); rebuild "
                              "(python -c 'import __graft_entry__ as g; g.build()')")
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().seg_last_error()
        raise SegHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def call(name: str, *args):
    check(getattr(lib(), name)(*args), name)
