"""SegModel - executes a layer graph (graph.py) on one MI355X through the C ABI in include/seg_hip.h.

This object stands where the reference has a ``tf.keras.Model`` (returned by the encoder-named methods,
`TensorFlow/2DCNN/models/unet_variants.py:1109-1114`) and offers the calls the reference's scripts make on it
(SURVEY.md section 8b): ``predict`` (2DCNN/Test.py:153,161), ``get_weights``/``set_weights``/``count_params``
(Train.py:354-367), a ``train_step`` = Keras' default fit step (Train.py:322-325, 394-415: forward with
training-mode BN, sum of per-output losses with unit weights, gradients, Adam), and ``summary``.

torch is used for device memory, streams and (in dp.py) torch.distributed - every FLOP of the path runs in the
hand-written HIP kernels.  There is no CPU fallback: constructing a SegModel without a GPU / without
libseg_hip.so raises.

Lowering (Keras-level nodes -> fused launches)
  conv [-> bn] [-> act] [-> maxpool]   one implicit-GEMM launch (virtual concat, bias, BN partial statistics in
                                       the epilogue) + bn_finalize + bn_act_apply (activation and 2x2 pool fused)
  convT [-> bn] [-> act]               the same kernel with 4 (2 in 1D) sub-pixel classes; LeakyReLU in the epilogue
  concat                               never materialised when it feeds a conv / ConvLSTM
  convlstm                             implicit GEMM (4F gate pre-activations) + gate kernel
  1x1 conv that is a model output      head kernel (fp32 logits/probabilities, sigmoid / softmax fused)
  add [-> act], mul, upsample, maxpool, standalone bn/act      one HBM-bound kernel each
Backward mirrors it: bn/act backward in two passes, data gradients through the same implicit-GEMM kernel with
transposed weights (one launch per concat segment, accumulating where a tensor has several consumers), weight
gradients through the transposed-read MFMA kernel into ONE flat fp32 gradient buffer (single all-reduce, single
Adam launch).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import zlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _ffi
from ._ffi import TransposeDesc, ACT, LOSS, OPTIMIZER, Geom, IgemmFwdArgs, IgemmWgradArgs
from .graph import Graph, Node, ParamSpec
from .splitfmt import split_pack, split_unpack

BN_PARTS_BWD = int(os.environ.get("SEG_BN_PARTS_BWD", "512"))     # partial rows of the BN-backward reductions: 2 blocks per CU (same-box A/B against 1024: C4 47.4 -> 46.8 ms, C3 -0.1, C2 -0.06; 256 and 2048 lose)
GRAD_TARGET = 1024.0      # max |dLoss/dOut| is rescaled to [512, 1024) before the fp16 backward pass
WGRAD_VARIANT = 0         # 0: ds_read_b64_tr_b16 operand reads, 1: transposing LDS stores
WGRAD_OVERWRITE = os.environ.get("SEG_WGRAD_OVERWRITE", "1") != "0"   # single-writer weight-gradient tiles store into the cleared buffer
WGRAD_HALO = os.environ.get("SEG_WGRAD_HALO", "1") != "0"      # variant 2 (LDS-DMA halo tiles) for the 3x3 stride-1 layers
WGRAD_DETERMINISTIC = os.environ.get("SEG_WGRAD_DETERMINISTIC", "1") != "0"   # pixel-split weight gradients through per-layer partial tiles + an ordered reduce (no float atomics)


def _r8(v):
    return (v + 7) // 8 * 8


def _r32(v):
    return (v + 31) // 32 * 32


def _cdiv(a, b):
    return -(-a // b)


class _DryRun:
    """Host-logic test mode (tests/test_lowering.py): buffers live on the CPU and kernel calls are only recorded.
    It computes nothing - it is NOT a fallback; SegModel.build() refuses to run without a GPU unless a test
    switches this on explicitly."""
    on = False
    log: List[str] = []


def _dev():
    return "cpu" if _DryRun.on else "cuda"


class KernelProfile:
    """Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg)."""
    on = False
    records: List[tuple] = []     # (name, tag, flops, bytes, start_event, end_event)

    @classmethod
    def summary(cls):
        torch.cuda.synchronize()
        agg: Dict[str, List[float]] = {}
        for name, tag, flops, nbytes, e0, e1 in cls.records:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += flops
            a[3] += nbytes
        return {k: dict(launches=v[0], seconds=v[1], flops=v[2], bytes=v[3]) for k, v in agg.items()}


class _Mode:
    """Storage / MFMA operand type of the model currently executing: '' = fp16 (throughput mode), '_f32' = the parity mode (tensors
    stored as split fp16 pairs, three MFMAs per product)."""
    sfx = ""


def call(name, *args, flops=0.0, nbytes=0.0, tag=""):
    if name in _ffi.DUAL:
        name = name + _Mode.sfx
    if _DryRun.on:
        _DryRun.log.append(name)
        return
    if KernelProfile.on:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _ffi.call(name, *args)
        e1.record()
        KernelProfile.records.append((name, tag, flops, nbytes, e0, e1))
        return
    _ffi.call(name, *args)


def _stream():
    if _DryRun.on:
        return C.c_void_p(0)
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ------------------------------------------------------------------------------------------------ init
def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def init_param(spec, seed) -> np.ndarray:
    """Keras initialisers by name (he_uniform: unet_variants.py:7; glorot_uniform default; he_normal: :149)."""
    r = _rng(seed, spec.name)
    fi, fo = spec.fan
    shape, init = spec.shape, spec.init
    if init == "zeros":
        return np.zeros(shape, np.float32)
    if init == "ones":
        return np.ones(shape, np.float32)
    if init == "he_uniform":
        lim = math.sqrt(6.0 / fi)
        return r.uniform(-lim, lim, shape).astype(np.float32)
    if init == "glorot_uniform":
        lim = math.sqrt(6.0 / (fi + fo))
        return r.uniform(-lim, lim, shape).astype(np.float32)
    if init == "he_normal":
        std = math.sqrt(2.0 / fi) / 0.87962566103423978
        v = np.clip(r.standard_normal(shape), -2.0, 2.0)
        return (v * std).astype(np.float32)
    if init == "orthogonal":
        rows, cols = int(np.prod(shape[:-1])), int(shape[-1])
        a = r.standard_normal((max(rows, cols), min(rows, cols)))
        q, rr = np.linalg.qr(a)
        q = q * np.sign(np.diag(rr))
        if rows < cols:
            q = q.T
        return q.reshape(shape).astype(np.float32)
    if init == "unit_forget_bias":
        b = np.zeros(shape, np.float32)
        F4 = shape[0] // 4
        b[F4:2 * F4] = 1.0
        return b
    raise ValueError(init)


# ------------------------------------------------------------------------------------------------ geometry
def _same_pad(n_in, k, s):
    out = _cdiv(n_in, s)
    total = max((out - 1) * s + k - n_in, 0)
    return total // 2


def _fill_geom(g: Geom, N, Hin, Win, Hl, Wl, isy, isx, Hout, Wout, osy, osx, classes, T_total):
    """classes: list of (oy, ox, [(dy, dx, widx), ...]) with the same tap count each."""
    g.N, g.Hin, g.Win, g.Hl, g.Wl, g.isy, g.isx = N, Hin, Win, Hl, Wl, isy, isx
    g.Hout, g.Wout, g.osy, g.osx = Hout, Wout, osy, osx
    g.ncls, g.ntaps, g.T_total = len(classes), len(classes[0][2]), T_total
    assert all(len(c[2]) == g.ntaps for c in classes) and g.ncls * g.ntaps <= _ffi.MAX_TAPS and g.ncls <= _ffi.MAX_CLS
    for ci, (oy, ox, taps) in enumerate(classes):
        g.oy[ci], g.ox[ci] = oy, ox
        for t, (dy, dx, wi) in enumerate(taps):
            e = ci * g.ntaps + t
            g.tap_dy[e], g.tap_dx[e], g.tap_w[e] = dy, dx, wi
    return g


def geom_conv(N, H, W, kh, kw, sh, sw) -> Geom:
    """Conv2D(padding='same'): out(i,j) = sum_k in(i*s + k - pad_before)."""
    pt, pl = _same_pad(H, kh, sh), _same_pad(W, kw, sw)
    Ho, Wo = _cdiv(H, sh), _cdiv(W, sw)
    taps = [(ky - pt, kx - pl, ky * kw + kx) for ky in range(kh) for kx in range(kw)]
    return _fill_geom(Geom(), N, H, W, Ho, Wo, sh, sw, Ho, Wo, 1, 1, [(0, 0, taps)], kh * kw)


def geom_conv_dgrad(N, H, W, kh, kw, sh, sw) -> Geom:
    """Data gradient of the conv above, as a gather over dz.  stride 1: dx(y) = sum_k dz(y + pad - k) w[k];
    1x1/stride s: dx(i*s) = dz(i) w (other pixels untouched -> caller pre-zeroes)."""
    Ho, Wo = _cdiv(H, sh), _cdiv(W, sw)
    if sh == 1 and sw == 1:
        pt, pl = _same_pad(H, kh, 1), _same_pad(W, kw, 1)
        taps = [(pt - ky, pl - kx, ky * kw + kx) for ky in range(kh) for kx in range(kw)]
        return _fill_geom(Geom(), N, Ho, Wo, H, W, 1, 1, H, W, 1, 1, [(0, 0, taps)], kh * kw)
    if kh == 1 and kw == 1:
        return _fill_geom(Geom(), N, Ho, Wo, Ho, Wo, 1, 1, H, W, sh, sw, [(0, 0, [(0, 0, 0)])], 1)
    raise NotImplementedError("data gradient of a strided conv with a kernel larger than 1x1 is not on the hot path")


def geom_convT(N, H, W, kh, kw, sh, sw) -> Geom:
    """Conv2DTranspose(padding='same'): Y = s*i + k - pb  ->  sub-pixel class py = Y mod s gathers the taps
    k with (py + pb - k) % s == 0 at input offset (py + pb - k) / s."""
    pbh, pbw = max(kh - sh, 0) // 2, max(kw - sw, 0) // 2
    classes = []
    for py in range(sh):
        for px in range(sw):
            taps = []
            for ky in range(kh):
                if (py + pbh - ky) % sh:
                    continue
                for kx in range(kw):
                    if (px + pbw - kx) % sw:
                        continue
                    taps.append(((py + pbh - ky) // sh, (px + pbw - kx) // sw, ky * kw + kx))
            classes.append((py, px, taps))
    return _fill_geom(Geom(), N, H, W, H, W, 1, 1, H * sh, W * sw, sh, sw, classes, kh * kw)


def geom_convT_dgrad(N, H, W, kh, kw, sh, sw) -> Geom:
    """dx(i) = sum_k dY(s*i + k - pb) w[k]  (a strided gather over the up-sampled gradient)."""
    pbh, pbw = max(kh - sh, 0) // 2, max(kw - sw, 0) // 2
    taps = [(ky - pbh, kx - pbw, ky * kw + kx) for ky in range(kh) for kx in range(kw)]
    return _fill_geom(Geom(), N, H * sh, W * sw, H, W, sh, sw, H, W, 1, 1, [(0, 0, taps)], kh * kw)


def pack_gemm_weight(kern: np.ndarray, cin_l: Sequence[int], woff: Sequence[int], n_rows: int, Kpad: int,
                     transposed: bool) -> np.ndarray:
    """Keras kernel (kh,kw,Cin,Cout) [conv, ConvLSTM] or (kh,kw,Cout,Cin) [transposed conv] -> GEMM operand
    [n_rows][T][Kpad]: segment s of the (virtual) concat occupies K columns woff[s] .. woff[s]+cin_l[s]."""
    T = kern.shape[0] * kern.shape[1]
    k2 = np.asarray(kern, np.float32).reshape(T, kern.shape[2], kern.shape[3])
    cout = kern.shape[2] if transposed else kern.shape[3]
    E = np.zeros((n_rows, T, Kpad), np.float32)
    lo = 0
    for s, c in enumerate(cin_l):
        blk = k2[:, :, lo:lo + c].transpose(1, 0, 2) if transposed else k2[:, lo:lo + c, :].transpose(2, 0, 1)
        E[:cout, :, woff[s]:woff[s] + c] = blk
        lo += c
    return E


def unpack_gemm_weight(flat: np.ndarray, shape, cin_l, woff, n_rows, Kpad, transposed) -> np.ndarray:
    T = shape[0] * shape[1]
    cout = shape[2] if transposed else shape[3]
    E = np.asarray(flat, np.float32).reshape(n_rows, T, Kpad)
    out_k = np.zeros((T, shape[2], shape[3]), np.float32)
    lo = 0
    for s, c in enumerate(cin_l):
        blk = E[:cout, :, woff[s]:woff[s] + c]
        if transposed:
            out_k[:, :, lo:lo + c] = blk.transpose(1, 0, 2)
        else:
            out_k[:, lo:lo + c, :] = blk.transpose(1, 2, 0)
        lo += c
    return out_k.reshape(shape)


def pick_tile_fwd(M, n_rows, ncls, ktot=1 << 30):
    """(tile_m, tile_n).  256x64, 128x128 and 128x64 run on the LDS-DMA pipelined kernel (fp16); the 64-row tiles are
    the register-staged kernel for problems too small to fill 256 CUs otherwise.  ktot = taps x K channels: with a
    short reduction (the 2x2 transposed convs) the smaller tile's extra blocks per CU hide more of the prologue /
    epilogue latency than the larger tile saves in weight re-reads (measured, tools/op_bench.py)."""
    blocks = lambda tm, tn: ncls * _cdiv(M, tm) * _cdiv(n_rows, tn)
    if n_rows <= 64:
        if blocks(256, 64) >= 512 and ktot >= 1024:
            return 256, 64
        return (128, 64) if blocks(128, 64) >= 256 else (64, 64)
    if blocks(128, 128) >= 384:
        return 128, 128
    if blocks(128, 64) >= 128:          # even at half a block per CU the LDS-DMA kernel beats the 64-row register-staged one 2x (8x8 level)
        return 128, 64
    return (64, 128) if blocks(64, 128) >= 256 else (64, 64)


def wgrad_taps_per_block(tile_m, tile_n, ntaps, variant=0):
    """Mirror of seg_wgrad_taps_per_block() (csrc/igemm_wgrad.hip); tests/test_cabi.py keeps the two in step."""
    if variant != 0 or tile_m != 64:
        return 1
    if tile_n == 64:
        return ntaps if ntaps in (9, 4, 3) else 1
    if tile_n == 128:
        return 4 if ntaps == 4 else 3 if ntaps in (9, 3) else 1
    return 1


def fwd_halo_patch(g: Geom) -> int:
    """Mirror of seg_fwd_halo_patch() (csrc/igemm_halo.hip): patch width of the halo-tile forward / data-gradient kernel
    (variant 3; 32 = 8x32-pixel patches, 16 = 16x16), 0 when the geometry is neither a stride-1 'same' convolution with
    the nine tap offsets {-1,0,1}^2 (in any order) nor a 4x4 / stride-2 transposed convolution (four sub-pixel classes
    of 2x2 taps), or the image is not tiled by those patches."""
    if g.isy == 2:         # data gradient of the 4x4 / stride-2 transposed conv: four parity planes of the stride-2 input
        if (g.ncls, g.ntaps, g.isx, g.osy, g.osx) != (1, 16, 2, 1, 1) or (g.Hin, g.Win, g.Hout, g.Wout) != (2 * g.Hl, 2 * g.Wl, g.Hl, g.Wl):
            return 0
        if g.oy[0] != 0 or g.ox[0] != 0 or any(g.tap_dy[t] != (t >> 2) - 1 or g.tap_dx[t] != (t & 3) - 1 for t in range(16)):
            return 0
        return 32 if g.Wl % 32 == 0 and g.Hl % 8 == 0 else 16 if g.Wl % 16 == 0 and g.Hl % 16 == 0 else 0
    if (g.isy, g.isx) != (1, 1) or (g.Hl, g.Wl) != (g.Hin, g.Win):
        return 0
    if g.ncls == 1 and g.ntaps == 9:
        if (g.osy, g.osx, g.Hout, g.Wout, g.oy[0], g.ox[0]) != (1, 1, g.Hin, g.Win, 0, 0):
            return 0
        if {(g.tap_dy[t], g.tap_dx[t]) for t in range(9)} != {(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)}:
            return 0
    elif g.ncls == 4 and g.ntaps == 4:
        if (g.osy, g.osx, g.Hout, g.Wout) != (2, 2, 2 * g.Hin, 2 * g.Win):
            return 0
        if any(not (0 <= g.oy[c] <= 1 and 0 <= g.ox[c] <= 1) for c in range(4)):
            return 0
        if any(abs(g.tap_dy[e]) > 1 or abs(g.tap_dx[e]) > 1 for e in range(16)):
            return 0
    else:
        return 0
    if g.Win % 32 == 0 and g.Hin % 8 == 0:
        return 32
    if g.Win % 16 == 0 and g.Hin % 16 == 0:
        return 16
    return 0


FUSE_ACT_BWD = os.environ.get("SEG_FUSE_ACT_BWD", "1") != "0"
FUSE_BN_TAILS = os.environ.get("SEG_FUSE_BN_TAILS", "1") != "0"             # BN(ReLU(a + b)) tails of MultiResBlock / ResPath as fused passes
FUSE_INFERENCE_BN = os.environ.get("SEG_FUSE_INFERENCE_BN", "1") != "0"     # predict: conv -> BN(moving statistics) -> activation as ONE launch
# the fused tails, round 4 (each with its own switch for bisecting / same-box A/Bs):
TAIL_LAZY_INPUTS = os.environ.get("SEG_TAIL_LAZY_INPUTS", "1") != "0"       # a Conv_Block feeding only the add is normalised on load there (no bn_act_apply pass)
TAIL_SHARED_GRAD = os.environ.get("SEG_TAIL_SHARED_GRAD", "1") != "0"       # the add's gradient is written once; the Conv_Blocks behind it read the one buffer
TAIL_VIRTUAL_CAT = os.environ.get("SEG_TAIL_VIRTUAL_CAT", "1") != "0"       # MultiResBlock's cat[c3, c5, c7] exists only in the tail kernels' LDS tiles
TAIL_LAZY_SUMS = os.environ.get("SEG_TAIL_LAZY_SUMS", "1") != "0"           # the BN-backward sums of those Conv_Blocks ride on the tail's fan-out launch (no reduce pass)
GATHER_DGRAD = os.environ.get("SEG_GATHER_DGRAD", "1") != "0"
GATHER_DGRAD_1X1 = os.environ.get("SEG_GATHER_DGRAD_1X1", "1") != "0"       # the 1x1 shortcut conv joins the gathered data gradient of its 3x3 sibling
GATHER_1X1_MAX_C = int(os.environ.get("SEG_GATHER_1X1_MAX_C", "64"))         # ... for layers of at most this many channels (HBM-bound ones)
BIAS_FROM_WGRAD = os.environ.get("SEG_BIAS_FROM_WGRAD", "1") != "0"
ADAM_ASYNC = os.environ.get("SEG_ADAM_ASYNC", "1") != "0"
DP_ASYNC = os.environ.get("SEG_DP_ASYNC", "1") != "0"           # data parallel: exchanged slices are awaited + updated on the side stream under the next forward
DP_URGENT_ELEMS = int(os.environ.get("SEG_DP_URGENT_ELEMS", str(4 << 20)))     # size at which the first (most urgent) gradient slice closes
DP_BUSBW_GBS = float(os.environ.get("SEG_DP_BUSBW_GBS", "300"))                 # all-reduce bus bandwidth the hold-back estimate assumes
ADAM_SLICE_ELEMS = int(os.environ.get("SEG_ADAM_SLICE_ELEMS", str(4 << 20)))
ADAM_URGENT_FIRST = os.environ.get("SEG_ADAM_URGENT_FIRST", "1") != "0"
FWD_HALO = os.environ.get("SEG_FWD_HALO", "1") != "0"       # variant 3 (halo tiles) for the 3x3 stride-1 layers
FWD_HALO_MIN_BLOCKS = int(os.environ.get("SEG_FWD_HALO_MIN_BLOCKS", "384"))
FWD_HALO_MIN_BLOCKS_F32 = int(os.environ.get("SEG_FWD_HALO_MIN_BLOCKS_F32", os.environ.get("SEG_FWD_HALO_MIN_BLOCKS", "128")))


def pick_fwd(M, n_rows, g: Geom, ktot, f16: bool):
    """(tile_m, tile_n, variant) of a forward / data-gradient launch.  f16 = False: the parity mode (split-pair tensors, csrc
    igemm_fwd_split.hip / igemm_halo.hip SPLIT): the same kernel families with two accumulator sets per wave, hence the halo
    tile is 256 x 64 only and the per-tap LDS-DMA kernel has no 64-deep K step."""
    if FWD_HALO and fwd_halo_patch(g):
        tn = 64 if (n_rows <= 64 or not f16) else 128
        if not f16 and n_rows <= 32 and FWD_HALO_NARROW:
            tn = 32           # parity mode, at most 32 output channels: one 32-channel tile per wave (half the MFMAs of the 64-channel tile)
        # (parity mode: the halo kernel wins down to half a block per CU - same-box A/B of the threshold, 384 against 128 blocks:
        # C3 50.05 -> 49.8 ms, C2 17.20 -> 16.78 ms per step)
        if g.ncls * (M // 256) * _cdiv(n_rows, tn) >= (FWD_HALO_MIN_BLOCKS if f16 else FWD_HALO_MIN_BLOCKS_F32):
            return 256, tn, 3
    tm, tn = pick_tile_fwd(M, n_rows, g.ncls, ktot)
    if not f16 and n_rows > 64 and g.ncls * _cdiv(M, 128) * _cdiv(n_rows, 128) >= 256:
        tm, tn = 128, 128          # parity mode, per-tap kernel at one block per CU: X41 forward 499 -> 444 us (tools/op_bench.py --dtype f32)
    return tm, tn, 0


FWD_SPLITK = os.environ.get("SEG_FWD_SPLITK", "1") != "0"              # split reduction of under-filled forward / data-gradient grids
FWD_SPLITK_MAX_BLOCKS = int(os.environ.get("SEG_FWD_SPLITK_MAX_BLOCKS", "256"))     # ... grids of at most one block per CU
FWD_SPLITK_TARGET = int(os.environ.get("SEG_FWD_SPLITK_TARGET", "512"))             # ... are split towards two blocks per CU
FWD_SPLITK_MIN_K = int(os.environ.get("SEG_FWD_SPLITK_MIN_K", "3072"))              # ... while a slice keeps this much of the reduction


def plan_fwd(M, n_rows, g: Geom, ktot, f16: bool):
    """(tile_m, tile_n, variant, splitk) of a forward / data-gradient launch: pick_fwd, then the split of the reduction
    (seg_igemm_fwd_args.splitk) for grids that leave CUs idle - the 8x8 / 16x16 levels of a depth-5 encoder: M = 1 024 .. 4 096
    pixels against K = 4 608 .. 18 432.  Measured on MI355X, parity mode (tools/op_bench.py --splitk, 128x128 tile): enc6 data
    gradient (64 blocks) 291 -> 129 us, latent convs (128 blocks) 308 -> 220, X41 transposed conv 318 -> 212, enc5 data gradient
    (the halo kernel at 128 blocks: 194) -> 124, X31 transposed-conv data gradient (halo, 256 blocks: 278) -> 214; a 256-block
    grid with K = 4 608 (enc5 forward) does not gain, hence the floor on the slice."""
    tm, tn, variant = pick_fwd(M, n_rows, g, ktot, f16)
    if not FWD_SPLITK or g.ntaps not in (1, 2, 3, 4, 9, 16):
        return tm, tn, variant, 1
    ctn = 128 if n_rows > 64 else 64
    blocks = g.ncls * _cdiv(M, 128) * _cdiv(n_rows, ctn)
    if blocks > FWD_SPLITK_MAX_BLOCKS:
        return tm, tn, variant, 1
    sk = max(1, FWD_SPLITK_TARGET // blocks)
    kmin = FWD_SPLITK_MIN_K            # (throughput mode, same floor: C3 21.6 -> 21.45 ms per step; twice the floor: no change)
    while sk > 1 and ktot // sk < kmin:
        sk //= 2
    if sk <= 1:
        return tm, tn, variant, 1
    return 128, ctn, 0, min(sk, 64)


def fwd_workspace_bytes(a) -> int:
    """Mirror of seg_igemm_fwd_workspace_bytes(): accumulator images of a split reduction."""
    if a.splitk <= 1:
        return 0
    Mtot = a.g.N * a.g.Hl * a.g.Wl
    return a.splitk * a.g.ncls * _cdiv(Mtot, a.tile_m) * _cdiv(a.n_rows, a.tile_n) * a.tile_m * a.tile_n * 4


def wgrad_halo_patch(g: Geom) -> int:
    """Mirror of seg_wgrad_halo_patch() (csrc/igemm_wgrad.hip): patch width of the LDS-DMA halo-tile weight-gradient kernel
    (variant 2) for this geometry, 0 when it is neither a 3x3 / stride-1 / 'same' convolution nor a 4x4 / stride-2
    transposed convolution the kernel can tile."""
    if (g.isy, g.isx) != (1, 1) or (g.Hl, g.Wl) != (g.Hin, g.Win):
        return 0
    if g.ncls == 1 and g.ntaps == 9:
        if (g.osy, g.osx, g.Hout, g.Wout, g.oy[0], g.ox[0]) != (1, 1, g.Hin, g.Win, 0, 0):
            return 0
        if any(g.tap_dy[t] != t // 3 - 1 or g.tap_dx[t] != t % 3 - 1 for t in range(9)):
            return 0
    elif g.ncls == 4 and g.ntaps == 4:
        if (g.osy, g.osx, g.Hout, g.Wout) != (2, 2, 2 * g.Hin, 2 * g.Win):
            return 0
        for c in range(4):
            if not (0 <= g.oy[c] <= 1 and 0 <= g.ox[c] <= 1):
                return 0
            sh = [(g.tap_dy[c * 4 + t], g.tap_dx[c * 4 + t]) for t in range(4)]
            if any(abs(dy) > 1 or abs(dx) > 1 for dy, dx in sh) or len(set(sh)) != 4:
                return 0
    else:
        return 0
    if g.Win % 16 == 0 and g.Hin % 4 == 0:         # 4 x 16 patches first (smallest halo overhead that still tiles most images)
        return 16
    if g.Win % 32 == 0 and g.Hin % 2 == 0:
        return 32
    if g.Win % 8 == 0 and g.Hin % 8 == 0:
        return 8
    return 0


def wgrad_halo_class_groups(g: Geom) -> int:
    """Mirror of seg_wgrad_halo_class_groups(): 1 (conv), 2 (transposed conv whose taps follow the class-pair table), else ncls."""
    if g.ncls != 4 or g.ntaps != 4:
        return g.ncls
    for c in range(4):
        py, px = c >> 1, c & 1
        if (g.oy[c], g.ox[c]) != (py, px):
            return 4
        for t in range(4):
            iy, ix = t >> 1, t & 1
            dy = (0, -1)[iy] if py == 0 else (1, 0)[iy]
            dx = (0, -1)[ix] if px == 0 else (1, 0)[ix]
            if (g.tap_dy[c * 4 + t], g.tap_dx[c * 4 + t]) != (dy, dx):
                return 4
    return 2


FWD_HALO_NARROW = os.environ.get("SEG_FWD_HALO_NARROW", "1") != "0"    # parity mode: the 256 x 32 halo tile for layers of <= 32 output channels
WGRAD_FORK_EARLY = os.environ.get("SEG_WGRAD_FORK_EARLY", "1") != "0"   # side stream forks when dz is final, before the op's data gradients are enqueued
HEAD_DW_SIDE = os.environ.get("SEG_HEAD_DW_SIDE", "1") != "0"          # output heads: weight / bias gradient on the side stream
HEAD_FUSE_FWD = os.environ.get("SEG_HEAD_FUSE_FWD", "1") != "0"        # a one-channel head's forward rides on the bn_act_apply pass of the Conv_Block it reads
HEAD_INJECT = os.environ.get("SEG_HEAD_INJECT", "1") != "0"            # ... and its data gradient is added on the fly in that block's BN-backward passes
WGRAD_BLOCKS = int(os.environ.get("SEG_WGRAD_BLOCKS", "1536"))       # block target of the split-K of the per-tap weight-gradient kernels
HALO_BLOCKS_TC = int(os.environ.get("SEG_HALO_BLOCKS_TC", os.environ.get("SEG_HALO_BLOCKS", "256")))   # the same for the transposed-conv class-pair kernel
HALO_BLOCKS = int(os.environ.get("SEG_HALO_BLOCKS", "256"))          # blocks per round of the variant-2 kernel: ONE per CU - it shares the GPU with the main stream (measured: 512 = 2 per CU is faster alone, slower in the step)


def pick_split_halo(M, n_rows, seg_cs: Sequence[int], ncls=1, plane=64):
    """Patch-range split of the variant-2 kernel: fill whole rounds of HALO_BLOCKS co-resident blocks, keep at least 8
    patches (512 pixels) per block.  plane: input channels per block (64; parity mode 32)."""
    tiles = ncls * _cdiv(n_rows, 64) * sum(_cdiv(c, plane) for c in seg_cs)
    npatch = M // 64
    best, best_eff = 1, 0.0
    HB = HALO_BLOCKS_TC if ncls > 1 else HALO_BLOCKS
    for sk in range(1, max(1, min(npatch // 8, _cdiv(2 * HB, tiles))) + 1):
        blocks = tiles * sk
        eff = blocks / (_cdiv(blocks, HB) * HB)
        if eff > best_eff + 1e-9:
            best, best_eff = sk, eff
    return best


def pick_tile_wgrad(M, n_rows, seg_cs: Sequence[int], ncls, ntaps, variant=0, f16=True):
    if variant == 2:
        return 64, 64, pick_split_halo(M, n_rows, seg_cs, ncls, 64 if f16 else 32)
    if not f16:
        variant = 1          # (no multi-tap form in the parity mode: one tap per block)
    tm = 128 if n_rows > 64 else 64
    tn = 128 if max(seg_cs) > 64 else 64
    ntb = wgrad_taps_per_block(tm, tn, ntaps, variant)
    tiles = _cdiv(n_rows, tm) * sum(_cdiv(c, tn) for c in seg_cs) * ncls * (ntaps // ntb)
    splitk = max(1, min(_cdiv(WGRAD_BLOCKS, tiles), _cdiv(M, 512)))
    return tm, tn, splitk


# ------------------------------------------------------------------------------------------------ tensors / params
class DevTensor:
    """NHWC activation with channel stride cs (multiple of 8); pad channels are zero.  Storage: fp16 (throughput mode) or the
    parity mode's split pairs - 4 bytes per element that torch sees as float32 (splitfmt.py); values() / grad_values() decode."""
    registry: Optional[Dict[str, "DevTensor"]] = None
    dtype = torch.float16

    def __init__(self, name, N, H, W, Cl):
        self.name, self.N, self.H, self.W, self.C = name, N, H, W, Cl
        self.cs = _r8(Cl)
        self.buf = torch.zeros((N, H, W, self.cs), dtype=DevTensor.dtype, device=_dev())
        self.grad: Optional[torch.Tensor] = None
        self.materialised = True      # False: a fused consumer computes the values on the fly, `buf` is never written
        self.needs_grad = True
        self.gw = False          # "gradient written" flag used while the static backward schedule is built
        self.claimers: List[Optional[tuple]] = []      # who writes this gradient: (dgrad args, span) for conv data gradients, else None
        if DevTensor.registry is not None:
            DevTensor.registry[name] = self

    @property
    def npix(self):
        return self.N * self.H * self.W

    def ensure_grad(self):
        if self.grad is None:
            self.grad = torch.zeros_like(self.buf)
        return self.grad

    @staticmethod
    def decode(t: torch.Tensor) -> torch.Tensor:
        """fp32 values of a storage tensor (fp16, or split pairs typed float32)."""
        return split_unpack(t) if t.dtype == torch.float32 else t.float()

    def values(self) -> torch.Tensor:
        return self.decode(self.buf)[..., :self.C]

    def grad_values(self) -> Optional[torch.Tensor]:
        return None if self.grad is None else self.decode(self.grad)[..., :self.C]

    def claim(self, owner: Optional[tuple] = None) -> int:
        """Returns the accumulate flag for the next producer of this tensor's gradient."""
        acc = 1 if self.gw else 0
        self.gw = True
        self.claimers.append(owner)
        self.ensure_grad()
        return acc


class Slot:
    """A parameter as the engine stores it (GEMM layout, padded) inside the flat buffers."""

    def __init__(self, spec, numel, to_engine, from_engine, trainable):
        self.spec, self.numel, self.to_engine, self.from_engine, self.trainable = spec, numel, to_engine, from_engine, trainable
        self.off = -1


class ParamStore:
    ALIGN = 64

    def __init__(self):
        self.slots: Dict[str, Slot] = {}
        self.order: List[Tuple[int, int]] = []
        self.n_train = 0
        self.n_state = 0
        self.P = self.G = self.M = self.V = self.P16 = self.PS = self.S = None

    def add(self, spec, numel, to_engine, from_engine):
        s = Slot(spec, numel, to_engine, from_engine, spec.trainable)
        if spec.trainable:
            self.order.append((self.n_train, _cdiv(numel, self.ALIGN) * self.ALIGN))    # creation order == op order
            s.off = self.n_train
            self.n_train += _cdiv(numel, self.ALIGN) * self.ALIGN
        else:
            s.off = self.n_state
            self.n_state += _cdiv(numel, self.ALIGN) * self.ALIGN
        self.slots[spec.name] = s
        return s

    def allocate(self, dtype="f16"):
        n = max(self.n_train, self.ALIGN)
        self.P = torch.zeros(n, dtype=torch.float32, device=_dev())
        self.G = torch.zeros(n, dtype=torch.float32, device=_dev())
        self.M = torch.zeros(n, dtype=torch.float32, device=_dev())
        self.V = torch.zeros(n, dtype=torch.float32, device=_dev())
        # operand copy of the weights the GEMM kernels read: fp16 image (throughput mode) or split pairs (parity mode, 4 B each)
        self.P16 = torch.zeros(n, dtype=torch.float16, device=_dev()) if dtype == "f16" else None
        self.PS = torch.zeros(n, dtype=torch.float32, device=_dev()) if dtype != "f16" else None
        self.S = torch.zeros(max(self.n_state, self.ALIGN), dtype=torch.float32, device=_dev())
        self.G16 = None           # bf16 image of G for the data-parallel exchange (allocated by dp.DataParallel(grad_dtype="bf16"))
        self.nonfinite = torch.zeros(1, dtype=torch.int32, device=_dev())      # elements whose update was skipped (NaN / Inf gradient)

    def view(self, name, which="P") -> torch.Tensor:
        s = self.slots[name]
        buf = getattr(self, which) if s.trainable else self.S
        return buf[s.off:s.off + s.numel]

    def ptr(self, name, which="P") -> C.c_void_p:
        s = self.slots[name]
        buf = getattr(self, which) if s.trainable else self.S
        return C.c_void_p(buf.data_ptr() + s.off * buf.element_size())


# ------------------------------------------------------------------------------------------------ ops
class Op:
    def forward(self, training: bool):
        raise NotImplementedError

    def plan_backward(self):
        """Called once, in reverse order, to fix accumulate flags."""

    def backward(self):
        pass

    def post_step(self):
        """Refresh derived weight copies after an optimiser step / weight import."""


class InputOp(Op):
    def __init__(self, m, t: DevTensor, Cin):
        self.m, self.t, self.Cin = m, t, Cin
        t.needs_grad = False

    def forward(self, training):
        x = self.m._x
        call("seg_pack_input", _p(x), _p(self.t.buf), self.t.npix, self.Cin, self.t.cs, _p(self.m.input_overflow), _stream())


class ConvOp(Op):
    """conv / convT / convlstm input-conv (+ BN + activation + pool | + ConvLSTM gates)."""

    def __init__(self, m, node: Node, kind, segs: List[Tuple[DevTensor, int]], bn: Optional[Node], act, alpha,
                 out: DevTensor, pooled: Optional[DevTensor], pool_size, lstm_F=0):
        self.m, self.node, self.kind, self.segs, self.bn = m, node, kind, segs, bn
        self.act, self.alpha, self.out, self.pooled, self.pool_size, self.F = act, alpha, out, pooled, pool_size, lstm_F
        st = m.store
        kh, kw = node.attrs["kernel"]
        sh, sw = node.attrs.get("strides", (1, 1))
        t0 = segs[0][0]
        N, H, W = t0.N, t0.H, t0.W
        self.N, self.H, self.W = N, H, W
        self.T = kh * kw
        self.cout = node.shape[2] if kind != "convlstm" else 4 * lstm_F
        self.n_rows = _r8(self.cout)
        # ---- K layout over the (virtual) concat
        self.woff, k = [], 0
        for t, _c in segs:
            self.woff.append(k)
            k += _r32(t.cs)
        self.Kpad = k
        self.rows_pad = _r32(self.n_rows)
        if kind == "convT":
            self.g_fwd = geom_convT(N, H, W, kh, kw, sh, sw)
            self.g_dgrad = geom_convT_dgrad(N, H, W, kh, kw, sh, sw)
            Ho, Wo = H * sh, W * sw
        else:
            self.g_fwd = geom_conv(N, H, W, kh, kw, sh, sw)
            self.g_dgrad = geom_conv_dgrad(N, H, W, kh, kw, sh, sw)
            Ho, Wo = _cdiv(H, sh), _cdiv(W, sw)
        self.strided = (sh, sw) != (1, 1) and kind != "convT"
        self.Ho, self.Wo = Ho, Wo
        self.M = N * self.g_fwd.Hl * self.g_fwd.Wl
        # ---- z: the conv output before BN / gates (aliases `out` when nothing follows that needs it)
        self.has_bn = bn is not None
        if self.has_bn or kind == "convlstm":
            self.z = DevTensor(node.name + ":z", N, Ho, Wo, self.cout)
        else:
            self.z = out
        assert self.z.cs == self.n_rows, (node.name, self.z.cs, self.n_rows)
        # ---- parameters
        cin_l = [c for _t, c in segs]
        n_rows, T, Kpad, woff, cout = self.n_rows, self.T, self.Kpad, list(self.woff), self.cout
        transposed = kind == "convT"

        def to_engine(kern):
            return pack_gemm_weight(kern, cin_l, woff, n_rows, Kpad, transposed).reshape(-1)

        def from_engine(flat, shape=node.params[0].shape):
            return unpack_gemm_weight(flat, shape, cin_l, woff, n_rows, Kpad, transposed)

        kspec = node.params[0]
        self.wname = kspec.name
        st.add(kspec, n_rows * T * Kpad, to_engine, from_engine)
        self.bname = None
        for ps in node.params[1:]:
            if ps.name.endswith("/bias"):
                self.bname = ps.name
                st.add(ps, n_rows, lambda b: np.pad(b, (0, n_rows - b.shape[0])), lambda f, c=cout: f[:c].copy())
            else:   # ConvLSTM recurrent kernel: a weight of the Keras layer that the T = 1 forward never reads
                st.add(ps, int(np.prod(ps.shape)), lambda a: a.reshape(-1), lambda f, sh_=ps.shape: f.reshape(sh_).copy())
        if self.has_bn:
            for ps in bn.params:
                st.add(ps, n_rows, _pad_to(n_rows), lambda f, c=cout: f[:c].copy())
            self.bn_name = bn.name
            self.eps, self.momentum = float(bn.attrs["eps"]), float(bn.attrs["momentum"])
        # ---- tiles and scratch
        self.tm, self.tn, self.fwd_variant, self.fwd_splitk = plan_fwd(self.M, n_rows, self.g_fwd, self.g_fwd.ntaps * self.Kpad, m.dtype == "f16")
        self.nparts = self.g_fwd.ncls * _cdiv(self.M, self.tm)
        dev = _dev()
        if self.has_bn:
            self.stats = torch.zeros(self.nparts * 2 * n_rows, dtype=torch.float32, device=dev)
            self.mean, self.invstd, self.scale, self.shift, self.c1, self.c2 = (
                torch.zeros(n_rows, dtype=torch.float32, device=dev) for _ in range(6))
            self.bpart = torch.zeros(BN_PARTS_BWD * 2 * n_rows, dtype=torch.float32, device=dev)
        if not self.has_bn:
            self.bpart = torch.zeros(BN_PARTS_BWD * 2 * n_rows, dtype=torch.float32, device=dev)
            self.c1, self.c2 = torch.zeros(n_rows, dtype=torch.float32, device=dev), torch.zeros(n_rows, dtype=torch.float32, device=dev)
        self.need_dgrad = [t.needs_grad for t, _c in segs]
        # algorithmic FLOPs (2*MAC over logical channels): forward == weight gradient; data gradient per segment
        pix_taps = self.N * self.g_fwd.Hl * self.g_fwd.Wl * self.g_fwd.ncls * self.g_fwd.ntaps
        self.flops_fwd = 2.0 * pix_taps * sum(cin_l) * self.cout
        self.flops_seg = [2.0 * pix_taps * c * self.cout for c in cin_l]
        self.wt = torch.zeros(Kpad * self.T * self.rows_pad, dtype=DevTensor.dtype, device=dev) if any(self.need_dgrad) else None
        self.fwd_args = None
        self.gathered: Dict[int, "GatherDgrad"] = {}      # segment index -> gathered data gradient that covers it
        self.lazy = False            # training: `out` is never written - its only consumer (a fused add) normalises z on load
        self.grad_src: Optional[DevTensor] = None      # backward: dOut is read from this tensor's gradient (shared with a sibling)
        self.tail_sums = 0           # backward: > 0 = a fused tail's fan-out launch already left this BN's two sums in `bpart` (that many rows)
        self.heads: List["HeadOp"] = []            # one-channel output heads on `out` (at most two) whose forward rides on this block's bn_act_apply
        self._head_desc = None
        self.head_done = False                     # ... did so in the forward pass just run (not in the one-launch inference form)
        self.head_inj: Optional[_ffi.HeadInject] = None      # backward: that head's data gradient is injected into the BN-backward passes
        self.gather_launch: List["GatherDgrad"] = []      # gathered data gradients launched at the end of this op's backward
        self.dz_ptr = None

    # -- argument blocks (built after the parameter store has been allocated)
    def finalize(self):
        st = self.m.store
        a = IgemmFwdArgs()
        a.g = self.g_fwd
        a.nseg = len(self.segs)
        for s, (t, _c) in enumerate(self.segs):
            a.seg_c[s], a.seg_cs[s], a.seg_woff[s], a.seg_ptr[s] = t.cs, t.cs, self.woff[s], t.buf.data_ptr()
        a.w = st.ptr(self.wname, "P16" if self.m.dtype == "f16" else "PS").value
        a.Kpad, a.n_rows, a.n_valid = self.Kpad, self.n_rows, self.cout
        a.out, a.out_cs, a.out_coff = self.z.buf.data_ptr(), self.z.cs, 0
        a.bias = st.ptr(self.bname, "P").value if self.bname else None
        fused_act = (not self.has_bn) and self.kind != "convlstm"
        a.act, a.alpha = (ACT[self.act] if fused_act else 0), float(self.alpha)
        a.accumulate = 0
        a.stats = self.stats.data_ptr() if self.has_bn else None
        a.tile_m, a.tile_n, a.variant, a.splitk = self.tm, self.tn, self.fwd_variant, self.fwd_splitk
        self.fwd_args = a
        self.fwd_args_eval = IgemmFwdArgs.from_buffer_copy(a)
        self.fwd_args_eval.stats = None
        self.m.fwd_ws_users += [a, self.fwd_args_eval]
        # inference (model.predict, 2DCNN/Test.py:153,161): BatchNormalization with the moving statistics is a per-channel affine
        # known before the conv runs - folded into the GEMM epilogue (ep_scale / bias), activation applied there, the result written
        # straight to `out`: ONE launch per Conv_Block (unet_variants.py:7-14), no pre-BN tensor, no normalisation pass
        self.fwd_args_infer = None
        self._infer_valid = False
        if self.has_bn and FUSE_INFERENCE_BN:
            self.inf_scale = torch.zeros(self.n_rows, dtype=torch.float32, device=_dev())
            self.inf_shift = torch.zeros(self.n_rows, dtype=torch.float32, device=_dev())
            b = IgemmFwdArgs.from_buffer_copy(self.fwd_args_eval)
            b.out, b.out_cs, b.out_coff = self.out.buf.data_ptr(), self.out.cs, 0
            b.bias, b.ep_scale = self.inf_shift.data_ptr(), self.inf_scale.data_ptr()
            b.act, b.alpha = ACT[self.act], float(self.alpha)
            self.fwd_args_infer = b
            self.m.fwd_ws_users.append(b)
        # wgrad
        wg = IgemmWgradArgs()
        wg.g = self.g_fwd
        wg.nseg = a.nseg
        for s in range(a.nseg):
            wg.seg_c[s], wg.seg_cs[s], wg.seg_woff[s], wg.seg_ptr[s] = a.seg_c[s], a.seg_cs[s], a.seg_woff[s], a.seg_ptr[s]
        wg.dz_cs, wg.dz_coff = self.z.cs, 0
        wg.dw = st.ptr(self.wname, "G").value
        wg.Kpad, wg.n_rows = self.Kpad, self.n_rows
        variant = WGRAD_VARIANT
        f16 = self.m.dtype == "f16"
        if variant == 0 and WGRAD_HALO and wgrad_halo_patch(self.g_fwd):
            variant = 2         # 3x3 stride-1 layers and 4x4 / stride-2 transposed convs: LDS-DMA halo-tile kernel
        tm, tn, sk = pick_tile_wgrad(self.M, self.n_rows, [t.cs for t, _c in self.segs],
                                     wgrad_halo_class_groups(self.g_fwd) if (variant == 2 and f16) else self.g_fwd.ncls, self.g_fwd.ntaps,
                                     variant, f16)
        wg.tile_m, wg.tile_n, wg.splitk = tm, tn, sk
        wg.scale, wg.scale_ptr = 1.0, self.m.gs.data_ptr() + 4
        wg.variant = variant
        # the gradient buffer is cleared once per step (backward() / the side stream after the optimiser), and with
        # splitk == 1 every tile has one writer: it may store instead of read-modify-write
        wg.overwrite = 1 if (WGRAD_OVERWRITE and variant == 2 and sk == 1) else 0
        # deterministic pixel split: the split blocks store partial tiles into this layer's own workspace (zero-filled once;
        # positions no block writes stay zero) and a second launch inside seg_igemm_wgrad adds them up in a fixed order
        self.wg_partial = None
        if WGRAD_DETERMINISTIC and sk > 1:
            self.wg_partial = torch.zeros(sk * self.n_rows * self.T * self.Kpad, dtype=torch.float32, device=_dev())
            wg.partial = self.wg_partial.data_ptr()
        # transposed convs on the class-pair kernel: the bias gradient (column sums of dz) comes out of the weight-gradient
        # launch as per-block partial rows instead of a separate pass over dz
        self.bias_from_wgrad = (BIAS_FROM_WGRAD and variant == 2 and self.bname is not None and not self.has_bn and self.kind == "convT"
                                and (wgrad_halo_class_groups(self.g_fwd) == 2 if f16 else self.g_fwd.ncls == 4))
        if self.bias_from_wgrad:
            self.wg_bias_parts = (2 if f16 else 4) * sk          # one partial row per class pair (fp16) / class (parity mode) and patch range
            self.wg_bpart = torch.zeros(self.wg_bias_parts * 2 * self.n_rows, dtype=torch.float32, device=_dev())
            wg.bias_part = self.wg_bpart.data_ptr()
        self.wg_args = wg

    def transpose_jobs(self):
        """(w fp32 [rows][T][K], dst, rows, T, K, k0, kcount, rows_pad, dst_ld, dst_coff) of the derived operands."""
        if self.wt is None:
            return []
        return [(self.m.store.ptr(self.wname, self.m.src_buf).value, self.wt.data_ptr(), self.n_rows, self.T, self.Kpad, 0, self.Kpad,
                 self.rows_pad, self.rows_pad, 0)]

    def post_step(self):
        self._infer_valid = False          # the folded inference-mode BN coefficients follow the weights

    def forward(self, training):
        st, s = self.m.store, _stream()
        self.head_done = False
        if not training and self.fwd_args_infer is not None:
            bn = self.bn_name
            if not self._infer_valid:
                call("seg_bn_fold_inference", st.ptr(bn + "/gamma"), st.ptr(bn + "/beta"), st.ptr(bn + "/moving_mean"),
                     st.ptr(bn + "/moving_variance"), self.eps, st.ptr(self.bname, "P") if self.bname else None, self.n_rows,
                     _p(self.inf_scale), _p(self.inf_shift), s)
                self._infer_valid = True
            call("seg_igemm_fwd", C.byref(self.fwd_args_infer), s, flops=self.flops_fwd, tag="fwd:" + self.node.name)
            if self.pooled is not None:
                call("seg_maxpool_fwd", _p(self.out.buf), _p(self.pooled.buf), self.out.N, self.out.H, self.out.W, self.out.cs,
                     self.pool_size[0], self.pool_size[1], s)
            return
        a = self.fwd_args if (training and self.has_bn) else self.fwd_args_eval
        if training and self.has_bn:
            self._infer_valid = False      # (this pass updates the moving statistics)
        call("seg_igemm_fwd", C.byref(a), s, flops=self.flops_fwd, tag="fwd:" + self.node.name)
        if self.has_bn:
            bn = self.bn_name
            call("seg_bn_finalize", _p(self.stats), self.nparts, self.n_rows, self.z.npix, st.ptr(bn + "/gamma"),
                 st.ptr(bn + "/beta"), self.eps, self.momentum, st.ptr(bn + "/moving_mean"), st.ptr(bn + "/moving_variance"),
                 1 if self.m.graph.ndim == 2 else 0, 1 if training else 0, _p(self.mean), _p(self.invstd), _p(self.scale),
                 _p(self.shift), s)
            if self.lazy and training:
                return                   # (the fused add that consumes this block applies scale / shift / ReLU on load)
            ph, pw = self.pool_size if self.pooled is not None else (1, 1)
            if self.heads:                   # the heads' dot products over the channels come out of this pass
                if self._head_desc is None:
                    d = _ffi.HeadFwdDesc()
                    d.n = len(self.heads)
                    for k, h in enumerate(self.heads):
                        d.w[k], d.b[k], d.act[k], d.y[k] = st.ptr(h.wname).value, st.ptr(h.bname).value, h.act, h.y.data_ptr()
                    self._head_desc = d
                call("seg_bn_act_apply_head", _p(self.z.buf), self.z.cs, 0, _p(self.out.buf), self.out.cs, 0, self.z.N, self.z.H,
                     self.z.W, self.n_rows, self.cout, _p(self.scale), _p(self.shift), ACT[self.act], float(self.alpha),
                     _p(self.pooled.buf) if self.pooled is not None else None, ph, pw, C.byref(self._head_desc), s)
                self.head_done = True
                return
            call("seg_bn_act_apply", _p(self.z.buf), self.z.cs, 0, _p(self.out.buf), self.out.cs, 0, self.z.N, self.z.H,
                 self.z.W, self.n_rows, self.cout, _p(self.scale), _p(self.shift), ACT[self.act], float(self.alpha),
                 _p(self.pooled.buf) if self.pooled is not None else None, ph, pw, s)
        elif self.kind == "convlstm":
            call("seg_convlstm_gate_fwd", _p(self.z.buf), self.z.cs, _p(self.out.buf), self.out.cs, self.z.npix, self.F, s)
        elif self.pooled is not None:
            call("seg_maxpool_fwd", _p(self.out.buf), _p(self.pooled.buf), self.out.N, self.out.H, self.out.W, self.out.cs,
                 self.pool_size[0], self.pool_size[1], s)

    def _bias_grad(self, dz, s, gs_inv):
        """dbias += (1/S) * column sums of dz: deterministic per-block partials + one finalize launch (thousands of
        same-address float atomics serialise at one memory channel - measured 170 us per 134 MB tensor)."""
        if self.bname is None or self.bias_from_wgrad:
            return
        st = self.m.store
        call("seg_bn_stats", _p(dz), self.z.npix, self.z.cs, 0, self.n_rows, _p(self.bpart), BN_PARTS_BWD, self.n_rows, 0, s)
        call("seg_bn_bwd_finalize", _p(self.bpart), BN_PARTS_BWD, self.n_rows, self.z.npix, None, st.ptr(self.bname, "G"), 1.0, gs_inv,
             _p(self.c1), _p(self.c2), s)

    def plan_backward(self):
        self.pool_acc = None
        if self.pooled is not None and self.pooled.gw:
            self.pool_acc = self.out.claim()
        self.has_dout = self.out.gw           # somebody writes dOut (with an injected head alone the buffer is only this block's dz)
        self.live = self.out.gw or self.head_inj is not None
        self.fused_act_bwd = False
        if not self.live:
            return
        # activation backward folded into the epilogue of the only consumer's data gradient (the transposed convs:
        # LeakyReLU, no BN, consumed by one conv): saves a read-modify-write pass over the gradient
        if (FUSE_ACT_BWD and not self.has_bn and self.kind != "convlstm" and ACT[self.act] in (1, 2, 3) and self.pool_acc is None
                and len(self.out.claimers) == 1 and self.out.claimers[0] is not None):
            ca, span = self.out.claimers[0]
            ca.out_y[span], ca.out_dact[span], ca.out_dalpha[span] = self.out.buf.data_ptr(), ACT[self.act], float(self.alpha)
            self.fused_act_bwd = True
        if self.kind == "convlstm":
            self.z.ensure_grad()
        # one data-gradient launch for the whole (virtual) concat: GEMM N = the K axis of the forward weight,
        # split back onto the segment tensors by the epilogue.  Segments whose tensor is handled by a gathered data
        # gradient (GatherDgrad: one launch per tensor over all its consuming convs) or needs no gradient are left
        # out: each run of consecutive remaining segments is one launch on that row range of the transposed weight
        self.dg_args = []
        own = [t.needs_grad and s_i not in self.gathered for s_i, (t, _c) in enumerate(self.segs)]
        s_i = 0
        while s_i < len(self.segs):
            if not own[s_i]:
                s_i += 1
                continue
            e_i = s_i
            while e_i < len(self.segs) and own[e_i]:
                e_i += 1
            a = IgemmFwdArgs()
            a.g = self.g_dgrad
            a.nseg = 1
            a.seg_c[0], a.seg_cs[0], a.seg_woff[0] = self.n_rows, self.z.cs, 0
            a.seg_ptr[0] = None     # dz pointer filled in backward()
            width = sum(_r32(t.cs) for t, _c in self.segs[s_i:e_i])
            a.w = self.wt.data_ptr() + self.woff[s_i] * self.T * self.rows_pad * self.wt.element_size()
            a.Kpad, a.n_rows, a.n_valid = self.rows_pad, width, width
            a.out, a.out_cs, a.out_coff, a.accumulate = None, 0, 0, 0
            a.bias, a.act, a.alpha, a.stats = None, 0, 0.0, None
            a.nout = e_i - s_i
            zero_first, fl = [], 0.0
            for o_i, (t, _c) in enumerate(self.segs[s_i:e_i]):
                a.out_span[o_i], a.out_cs_o[o_i] = _r32(t.cs), t.cs
                acc = t.claim((a, o_i))
                a.out_ptr[o_i], a.out_acc[o_i] = t.grad.data_ptr(), acc
                fl += self.flops_seg[s_i + o_i]
                if self.strided and not acc:
                    zero_first.append(t)
            Md = self.N * self.g_dgrad.Hl * self.g_dgrad.Wl
            a.tile_m, a.tile_n, a.variant, a.splitk = plan_fwd(Md, width, self.g_dgrad, self.g_dgrad.ntaps * self.rows_pad, self.m.dtype == "f16")
            self.m.fwd_ws_users.append(a)
            self.dg_args.append((a, zero_first, fl))
            s_i = e_i
        for grp in self.gather_launch:       # gathered data gradients this op launches (it is their last-running consumer)
            grp.plan()

    def backward(self):
        st, s = self.m.store, _stream()
        if self.pool_acc is not None:
            call("seg_maxpool_bwd", _p(self.pooled.grad), _p(self.out.buf), _p(self.pooled.buf), _p(self.out.grad),
                 self.out.N, self.out.H, self.out.W, self.out.cs, self.pool_size[0], self.pool_size[1], self.pool_acc, s)
        if not self.live:
            return
        gs_inv = C.c_void_p(self.m.gs.data_ptr() + 4)
        if self.has_bn:
            bn = self.bn_name
            dA = self.out.ensure_grad()
            src = _p(self.grad_src.grad if self.grad_src is not None else dA)      # (a sibling's buffer: written once for both, read-only here)
            inj = self.head_inj
            if inj is not None and not self.has_dout:
                src = None                    # the injected head is the only consumer: no gradient tensor to read
            if not self.tail_sums:
                call("seg_bn_act_bwd_reduce" + ("_inj" if inj is not None else ""), src, self.out.cs, 0, _p(self.z.buf), self.z.cs, 0,
                     self.z.npix, self.n_rows, _p(self.scale), _p(self.shift), _p(self.mean), _p(self.invstd), ACT[self.act],
                     float(self.alpha), _p(self.bpart), BN_PARTS_BWD, *((C.byref(inj),) if inj is not None else ()), s)
            call("seg_bn_bwd_finalize", _p(self.bpart), self.tail_sums or BN_PARTS_BWD, self.n_rows, self.z.npix, st.ptr(bn + "/gamma", "G"),
                 st.ptr(bn + "/beta", "G"), 1.0, gs_inv, _p(self.c1), _p(self.c2), s)
            call("seg_bn_act_bwd_apply" + ("_inj" if inj is not None else ""), src, self.out.cs, 0, _p(self.z.buf), self.z.cs, 0, _p(dA),
                 self.out.cs, 0, self.z.npix, self.n_rows, _p(self.scale), _p(self.shift), _p(self.mean), _p(self.invstd), _p(self.c1),
                 _p(self.c2), ACT[self.act], float(self.alpha), 0, *((C.byref(inj),) if inj is not None else ()), s)
            dz = dA
        elif self.kind == "convlstm":
            dz = self.z.ensure_grad()
            call("seg_convlstm_gate_bwd", _p(self.out.grad), self.out.cs, _p(self.z.buf), self.z.cs, _p(dz), self.z.npix, self.F, s)
            self._bias_grad(dz, s, gs_inv)
        else:
            dz = self.out.grad
            if ACT[self.act] != 0 and not self.fused_act_bwd:
                call("seg_act_bwd", _p(dz), _p(self.out.buf), _p(dz), self.out.npix, self.out.cs, self.cout, ACT[self.act],
                     float(self.alpha), None, 1.0, None, s)
            self._bias_grad(dz, s, gs_inv)
        self.wg_args.dz = dz.data_ptr()
        def wgrad(stream):
            call("seg_igemm_wgrad", C.byref(self.wg_args), stream, flops=self.flops_fwd, tag="wgrad:" + self.node.name)
            if self.bias_from_wgrad:
                call("seg_bn_bwd_finalize", _p(self.wg_bpart), self.wg_bias_parts, self.n_rows, self.z.npix, None, st.ptr(self.bname, "G"),
                     1.0, gs_inv, _p(self.c1), _p(self.c2), stream)

        # the weight gradient is only needed by the optimiser: it runs on a second stream, next to this op's own data
        # gradients (dz is final here - forking before they are enqueued lets both start at once) and to the HBM-bound
        # BN / activation backward kernels of the layers that follow
        side = self.m.side_stream
        early = side is not None and WGRAD_FORK_EARLY
        if early:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                wgrad(_stream())
        for a, zero_first, fl in self.dg_args:
            for t in zero_first:
                t.grad.zero_()
            a.seg_ptr[0] = dz.data_ptr()
            call("seg_igemm_fwd", C.byref(a), s, flops=fl, tag="dgrad:" + self.node.name)
        self.dz_ptr = dz.data_ptr()
        for grp in self.gather_launch:
            grp.launch(s)
        if side is None:
            wgrad(s)
        elif not early:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                wgrad(_stream())


class GatherDgrad:
    """Data gradient of ONE tensor consumed by several 3x3 / stride-1 convs (the dense skips of UNet++ / UNet 3+: X00
    feeds X01 .. X05) as one launch: dT = sum_j dz_j (*) W_j[:, T's K columns]^T is a conv over the virtual concat of the
    dz_j with the row slices of the transposed weights side by side along K.  Against one split-output launch per
    consumer this writes T's gradient once instead of read-modify-writing it per consumer and runs N = T's channels
    at the full K depth (level 0 of C3: 530 -> 900 TFLOP/s)."""

    def __init__(self, m, t: DevTensor, members):
        self.m, self.t, self.members = m, t, members          # members: [(ConvOp, segment index)] in forward order
        self.coloff, k = [], 0
        for op, _si in members:
            self.coloff.append(k)
            k += op.rows_pad
        self.Kg = k
        # a 1x1 / stride-1 conv on the same tensor (the shortcut of a ResPath iteration / MultiResBlock, unet_variants.py:88,110)
        # rides along as the CENTRE TAP of the 3x3 operand: its data gradient is no read-modify-write launch of its own
        self.op3 = next(op for op, _si in members if op.T == 9)
        self.T = 9
        self.w = torch.zeros(t.cs * self.T * self.Kg, dtype=DevTensor.dtype, device=_dev())
        self.args = None
        for op, si in members:
            op.gathered[si] = self
        members[0][0].gather_launch.append(self)               # first in forward order = last in the backward sweep

    def transpose_jobs(self):
        return [(self.m.store.ptr(op.wname, self.m.src_buf).value, self.w.data_ptr(), op.n_rows, op.T, op.Kpad, op.woff[si], self.t.cs,
                 op.rows_pad, self.Kg, self.coloff[j]) + ((9, 4) if op.T == 1 else ()) for j, (op, si) in enumerate(self.members)]

    def plan(self):
        op0 = self.op3
        a = IgemmFwdArgs()
        a.g = op0.g_dgrad
        a.nseg = len(self.members)
        for j, (op, _si) in enumerate(self.members):
            a.seg_c[j], a.seg_cs[j], a.seg_woff[j], a.seg_ptr[j] = op.n_rows, op.z.cs, self.coloff[j], None
        a.w = self.w.data_ptr()
        a.Kpad, a.n_rows, a.n_valid = self.Kg, self.t.cs, self.t.cs
        acc = self.t.claim()
        a.out, a.out_cs, a.out_coff, a.accumulate = self.t.grad.data_ptr(), self.t.cs, 0, acc
        a.bias, a.act, a.alpha, a.stats, a.nout = None, 0, 0.0, None, 0
        Md = op0.N * op0.g_dgrad.Hl * op0.g_dgrad.Wl
        a.tile_m, a.tile_n, a.variant, a.splitk = plan_fwd(Md, self.t.cs, op0.g_dgrad, op0.g_dgrad.ntaps * self.Kg, self.m.dtype == "f16")
        self.m.fwd_ws_users.append(a)
        self.args = a
        self.flops = sum(op.flops_seg[si] for op, si in self.members)

    def launch(self, s):
        a = self.args
        for j, (op, _si) in enumerate(self.members):
            a.seg_ptr[j] = op.dz_ptr
        call("seg_igemm_fwd", C.byref(a), s, flops=self.flops, tag="dgrad:gather:" + self.t.name)


def plan_gathers(m, ops) -> List[GatherDgrad]:
    """Tensors with >= 2 consuming 3x3 stride-1 convs whose data-gradient geometry the halo-tile kernel takes."""
    if not GATHER_DGRAD:
        return []
    users: Dict[int, list] = {}
    tens: Dict[int, DevTensor] = {}
    for op in ops:
        if not isinstance(op, ConvOp) or op.kind != "conv" or op.wt is None:
            continue
        kern, unit = tuple(op.node.attrs["kernel"]), tuple(op.node.attrs.get("strides", (1, 1))) == (1, 1)
        if kern == (1, 1) and unit and GATHER_DGRAD_1X1:
            # the 1x1 shortcut next to a 3x3 conv on the same tensor: a member only where the layer is HBM-bound (few channels on a
            # large image - there the read-modify-write of its own launch is what it costs: level 0 of C4 154 us against ~60 as one
            # more K segment of the sibling's launch); deeper levels are MFMA-bound and eight zero taps per product would lose
            if not (len(op.segs) == 1 and op.segs[0][0].cs <= GATHER_1X1_MAX_C and op.n_rows <= GATHER_1X1_MAX_C):
                continue
        elif kern != (3, 3) or not unit or not fwd_halo_patch(op.g_dgrad):
            continue
        for si, (t, _c) in enumerate(op.segs):
            if t.needs_grad:
                users.setdefault(id(t), []).append((op, si))
                tens[id(t)] = t
    out = []
    for k, mem in users.items():
        if any(op.T == 9 for op, _si in mem) and 2 <= len(mem) <= _ffi.MAX_SEG:       # (a 1x1 conv alone is no gathered launch)
            out.append(GatherDgrad(m, tens[k], mem))
    return out


def _pad_to(n):
    return lambda a: np.pad(np.asarray(a, np.float32), (0, n - a.shape[0]))


class HeadOp(Op):
    """1x1 conv that is a model output: fp32 result, sigmoid / softmax fused (unet_variants.py:1106, :136)."""

    def __init__(self, m, node: Node, x: DevTensor):
        self.m, self.node, self.x = m, node, x
        sh, sw = node.attrs["strides"]
        self.sy, self.sx = sh, sw
        self.cin, self.cout = x.C, node.shape[2]
        if self.cout > 8:
            raise ValueError("HeadOp handles up to 8 output channels; wider heads are lowered as WideHeadOp")
        self.w_ld = x.cs
        self.act = ACT[node.attrs["activation"]]
        Ho, Wo = node.shape[0], node.shape[1]
        self.y = torch.zeros((x.N, Ho, Wo, self.cout), dtype=torch.float32, device=_dev())
        self.dy = torch.zeros_like(self.y)
        self.work = torch.zeros(1024 * self.cout * (self.w_ld + 1), dtype=torch.float32, device=_dev())     # weight-gradient partial rows
        cin, cout, w_ld = self.cin, self.cout, self.w_ld
        kspec, bspec = node.params[0], node.params[1]

        def to_engine(k):
            E = np.zeros((cout, w_ld), np.float32)
            E[:, :cin] = k.reshape(cin, cout).T
            return E.reshape(-1)
        m.store.add(kspec, cout * w_ld, to_engine, lambda f: f.reshape(cout, w_ld)[:, :cin].T.reshape(kspec.shape).copy())
        m.store.add(bspec, cout, lambda b: np.asarray(b, np.float32), lambda f: f[:cout].copy())
        self.wname, self.bname = kspec.name, bspec.name
        # one output channel on a Conv_Block's output (a deep-supervision level of UNet++ / the model output on X05): the forward
        # dot product rides on that block's bn_act_apply pass, the data gradient is injected into its BN-backward passes
        self.prod: Optional[ConvOp] = None
        self.inject = False
        CH = x.cs // 8
        prod = next((o for o in m.ops if isinstance(o, ConvOp) and o.out is x), None)
        if (prod is not None and prod.has_bn and prod.kind != "convlstm" and self.cout == 1 and (sh, sw) == (1, 1)
                and CH <= 64 and CH & (CH - 1) == 0 and self.act in (ACT[None], ACT["sigmoid"], ACT["relu"]) and len(prod.heads) < 2):
            if HEAD_FUSE_FWD:
                self.prod = prod
                prod.heads.append(self)
            self.inject = HEAD_INJECT
            if self.inject:
                self.prod_inj = prod

    def forward(self, training):
        st = self.m.store
        if self.prod is not None and self.prod.head_done:
            return                         # (computed inside the producing block's bn_act_apply pass)
        call("seg_head_fwd", _p(self.x.buf), self.x.cs, self.cin, self.x.N, self.x.H, self.x.W, self.sy, self.sx,
             st.ptr(self.wname), self.w_ld, st.ptr(self.bname), self.cout, self.act, _p(self.y), _stream())

    def plan_backward(self):
        if self.inject and self.x.needs_grad and (self.prod_inj.head_inj is None or self.prod_inj.head_inj.n < 2):
            # no dx launch: the producing block adds gs * dpre * w to the gradient it reads in its own BN-backward passes
            st = self.m.store
            d = self.prod_inj.head_inj
            if d is None:
                d = _ffi.HeadInject()
                d.n, d.gs_ptr = 0, self.m.gs.data_ptr()
                self.prod_inj.head_inj = d
            k = d.n
            d.dy[k], d.y[k], d.w[k], d.act[k] = self.dy.data_ptr(), self.y.data_ptr(), st.ptr(self.wname).value, self.act
            d.n = k + 1
            self.acc = None
            return
        self.inject = False
        self.acc = self.x.claim() if self.x.needs_grad else None

    def backward(self):
        st = self.m.store
        if self.acc == 0 and (self.sy, self.sx) != (1, 1):
            self.x.grad.zero_()
        def launch(dx, dw):
            call("seg_head_bwd", _p(self.dy), _p(self.y), _p(self.x.buf), self.x.cs, self.cin, self.x.N, self.x.H, self.x.W,
                 self.sy, self.sx, st.ptr(self.wname), self.w_ld, self.cout, self.act,
                 (_p(self.x.grad) if self.acc is not None else None) if dx else None, self.acc or 0, _p(self.m.gs),
                 st.ptr(self.wname, "G") if dw else None, st.ptr(self.bname, "G") if dw else None,
                 _p(self.work) if dw else None, self.work.numel() if dw else 0, _stream())

        side = self.m.side_stream if HEAD_DW_SIDE else None
        if side is None:
            launch(True, True)
        else:
            # like the convs' weight gradients: the head's dw / db only feed the optimiser, so they leave the main stream
            if self.acc is not None:
                launch(True, False)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                launch(False, True)


class WideHeadOp(HeadOp):
    """Model output with more than 8 channels: the 1x1 conv is a ConvOp on the implicit-GEMM path (its bias in the GEMM
    epilogue, its gradients from the GEMM kernels); this op is the fp32 activation on the logits and its backward."""

    def __init__(self, m, node: Node, logits: DevTensor):
        self.m, self.node, self.x = m, node, logits
        self.cin = self.cout = node.shape[2]
        self.act = ACT[node.attrs["activation"]]
        self.y = torch.zeros((logits.N, logits.H, logits.W, self.cout), dtype=torch.float32, device=_dev())
        self.dy = torch.zeros_like(self.y)

    def forward(self, training):
        x = self.x
        call("seg_head_act_fwd", _p(x.buf), x.cs, x.npix, self.cout, self.act, _p(self.y), _stream())

    def plan_backward(self):
        self.acc = self.x.claim()

    def backward(self):
        x = self.x
        call("seg_head_act_bwd", _p(self.dy), _p(self.y), x.npix, self.cout, self.act, _p(self.m.gs), _p(x.grad), x.cs, self.acc,
             _stream())


class ReshapeOp(Op):
    """Flatten / Reshape around the Dense pair of the autoencoder bottleneck (1DCNN unet_variants.py:127-133): the
    H*W*C values of a sample, channels fastest (Keras' Flatten order on NHWC), become the channel axis of a 1x1 image
    or the other way round.  One element-granular channel copy each way (`seg_copy_channels`); `to_flat` tells which
    side holds the (N, 1, 1, H*W*C) tensor."""

    def __init__(self, m, x: DevTensor, out: DevTensor, to_flat: bool):
        self.m, self.x, self.out, self.to_flat = m, x, out, to_flat
        img, flat = (x, out) if to_flat else (out, x)
        if flat.cs != img.H * img.W * img.C:
            raise NotImplementedError(f"{flat.name}: the flattened extent H*W*C = {img.H * img.W * img.C} must be a multiple of 8")
        self.img, self.flat = img, flat

    def _copy(self, src_is_img: bool, src, dst, accumulate: int):
        img = self.img
        if src_is_img:
            call("seg_copy_channels", _p(src), img.cs, 0, _p(dst), img.C, 0, img.C, img.npix, accumulate, _stream())
        else:
            call("seg_copy_channels", _p(src), img.C, 0, _p(dst), img.cs, 0, img.C, img.npix, accumulate, _stream())

    def forward(self, training):
        self._copy(self.to_flat, self.x.buf, self.out.buf, 0)

    def plan_backward(self):
        self.live = self.out.gw
        self.acc = self.x.claim() if (self.live and self.x.needs_grad) else None

    def backward(self):
        if not self.live or self.acc is None:
            return
        self._copy(not self.to_flat, self.out.grad, self.x.grad, self.acc)


def _tail_parts(npix: int, cs: int) -> int:
    """Partial rows (= blocks) of the tail kernels that walk tiles of RP * 4 pixels (csrc/elementwise.hip col_layout): at most
    BN_PARTS_BWD, never more than there are tiles."""
    chp = 1
    while chp < cs // 8:
        chp <<= 1
    tile = (256 // min(chp, 256)) * 4
    return max(1, min(BN_PARTS_BWD, _cdiv(npix, tile)))


class BNOp(Op):
    """Stand-alone BatchNormalization (+ activation) on an existing tensor (MultiResBlock / ResPath tails)."""

    def __init__(self, m, bn: Node, x: DevTensor, act, alpha, out: DevTensor):
        self.m, self.bn_name, self.x, self.act, self.alpha, self.out = m, bn.name, x, act, alpha, out
        self.C, self.cl = x.cs, x.C
        self.eps, self.momentum = float(bn.attrs["eps"]), float(bn.attrs["momentum"])
        for ps in bn.params:
            m.store.add(ps, self.C, _pad_to(self.C), lambda f, c=self.cl: f[:c].copy())
        self.nparts = max(1, min(512, x.npix // 64))
        dev = _dev()
        self.stats = torch.zeros(self.nparts * 2 * self.C, dtype=torch.float32, device=dev)
        self.bpart = torch.zeros(BN_PARTS_BWD * 2 * self.C, dtype=torch.float32, device=dev)
        self.mean, self.invstd, self.scale, self.shift, self.c1, self.c2 = (
            torch.zeros(self.C, dtype=torch.float32, device=dev) for _ in range(6))
        # fused tails (FUSE_BN_TAILS, see SegModel._fuse_bn_tail): `fused_add` - this BN follows ReLU(a + b): the add kernel delivers
        # the statistics and this op's backward writes the masked gradient to the add's inputs; `folded_into` - this BN feeds such an
        # add and has no other consumer: its normalisation is applied inside the add kernel, `out` is never written;
        # `pieces` - (folded only) the input is a concatenation that is never materialised: the tail kernels gather it (seg_cat_desc)
        self.fused_add: Optional["AddOp"] = None
        self.folded_into: Optional["AddOp"] = None
        self.pieces: Optional[List[Tuple[DevTensor, int]]] = None
        self.cat: Optional[_ffi.CatDesc] = None
        self.cat_parts = _tail_parts(x.npix, self.C)

    def make_virtual(self, pieces: List[Tuple[DevTensor, int]]):
        """The input becomes a virtual concatenation of `pieces` (never written: the tensor's buffer is released)."""
        self.pieces = pieces
        d = _ffi.CatDesc()
        d.n, off = len(pieces), 0
        for k, (t, c) in enumerate(pieces):
            d.p[k].x, d.p[k].dx, d.p[k].cs, d.p[k].C, d.p[k].coff, d.p[k].acc = t.buf.data_ptr(), None, t.cs, c, off, 0
            off += c
        self.cat = d
        self.x.materialised = False
        self.x.buf = torch.empty((0,), dtype=self.x.buf.dtype, device=self.x.buf.device)      # (nothing reads or writes the concatenated tensor any more)

    def forward(self, training):
        st, s, bn, x = self.m.store, _stream(), self.bn_name, self.x
        if training and self.fused_add is None:
            if self.pieces is not None:       # statistics of the virtual concatenation: one gathering pass over the pieces
                call("seg_add_act_stats_fwd", None, None, None, None, 0, C.byref(self.cat), None, None, None, x.npix, self.C, self.cl,
                     0, 0.0, _p(self.stats), self.nparts, s)
            else:
                call("seg_bn_stats", _p(x.buf), x.npix, x.cs, 0, self.C, _p(self.stats), self.nparts, self.C, 0, s)
        call("seg_bn_finalize", _p(self.stats), self.nparts, self.C, x.npix, st.ptr(bn + "/gamma"), st.ptr(bn + "/beta"),
             self.eps, self.momentum, st.ptr(bn + "/moving_mean"), st.ptr(bn + "/moving_variance"),
             1 if self.m.graph.ndim == 2 else 0, 1 if training else 0, _p(self.mean), _p(self.invstd), _p(self.scale),
             _p(self.shift), s)
        if self.folded_into is not None:
            return                            # normalised on the fly by the add that consumes it
        call("seg_bn_act_apply", _p(x.buf), x.cs, 0, _p(self.out.buf), self.out.cs, 0, x.N, x.H, x.W, self.C, self.cl,
             _p(self.scale), _p(self.shift), ACT[self.act], float(self.alpha), None, 1, 1, s)

    def plan_backward(self):
        self.live = self.out.gw
        if self.fused_add is not None:
            self.acc = None
            if self.live:
                self.x.gw = True              # the add's output does receive a gradient - handed straight on to the add's inputs
            return
        if self.pieces is not None:
            self.acc = None
            if self.live:
                for k, (t, _c) in enumerate(self.pieces):
                    if t.needs_grad:
                        self.cat.p[k].acc = t.claim()
                        self.cat.p[k].dx = t.grad.data_ptr()
            return
        self.acc = self.x.claim() if (self.live and self.x.needs_grad) else None

    def backward(self):
        if not self.live:
            return
        st, s, bn, x = self.m.store, _stream(), self.bn_name, self.x
        gs_inv = C.c_void_p(self.m.gs.data_ptr() + 4)
        if self.pieces is not None:
            # virtual concatenation (folded into a fused add): the fan-out launch of the BatchNormalization behind the add left this
            # BN's two sums in `bpart` and the add's gradient g in `tail_grad`; d cat = sc (g - c1 - xhat c2) goes to the pieces
            add = self.folded_into
            call("seg_bn_bwd_finalize", _p(self.bpart), self.cat_parts, self.C, x.npix, st.ptr(bn + "/gamma", "G"),
                 st.ptr(bn + "/beta", "G"), 1.0, gs_inv, _p(self.c1), _p(self.c2), s)
            call("seg_cat_bn_bwd_scatter", _p(add.tail_grad.grad), C.byref(self.cat), x.npix, self.C, _p(self.scale), _p(self.mean),
                 _p(self.invstd), _p(self.c1), _p(self.c2), s)
            return
        dA = self.out.grad
        call("seg_bn_act_bwd_reduce", _p(dA), self.out.cs, 0, _p(x.buf), x.cs, 0, x.npix, self.C, _p(self.scale), _p(self.shift),
             _p(self.mean), _p(self.invstd), ACT[self.act], float(self.alpha), _p(self.bpart), BN_PARTS_BWD, s)
        call("seg_bn_bwd_finalize", _p(self.bpart), BN_PARTS_BWD, self.C, x.npix, st.ptr(bn + "/gamma", "G"),
             st.ptr(bn + "/beta", "G"), 1.0, gs_inv, _p(self.c1), _p(self.c2), s)
        if self.fused_add is not None:
            a = self.fused_add                # dx = [x > 0] sc (g - c1 - xhat c2) to every input of the add: no add backward pass
            vb = a.virtual_bn                 # ... and, with a virtual concatenation among the inputs, the sums of ITS BatchNormalization
            call("seg_bn_bwd_apply_fanout", _p(dA), _p(x.buf), a.gptrs, len(a.xs), a.mask, x.npix, self.C, _p(self.scale), _p(self.mean),
                 _p(self.invstd), _p(self.c1), _p(self.c2), 1 if ACT[a.act] == ACT["relu"] else 0,
                 C.byref(vb.cat) if vb is not None else None, _p(vb.mean) if vb is not None else None,
                 _p(vb.invstd) if vb is not None else None, _p(vb.bpart) if vb is not None else None,
                 a.lazy_descs, a.n_lazy_descs, self.cat_parts, s)
            return
        if self.acc is None:          # the input needs no gradient (a BN placed on the network input): gamma / beta are done
            return
        call("seg_bn_act_bwd_apply", _p(dA), self.out.cs, 0, _p(x.buf), x.cs, 0, _p(x.grad), x.cs, 0, x.npix, self.C,
             _p(self.scale), _p(self.shift), _p(self.mean), _p(self.invstd), _p(self.c1), _p(self.c2), ACT[self.act],
             float(self.alpha), self.acc, s)


class ConcatOp(Op):
    """Materialised concatenate(axis=-1) for consumers that cannot walk the pieces (BatchNormalization over a
    concatenation: MultiResBlock, unet_variants.py:95-96).  Convolutions never need it (virtual concat); neither does a
    BatchNormalization that is folded into a fused tail (`virtual`: this op then does nothing - see BNOp.make_virtual)."""

    def __init__(self, m, segs: List[Tuple[DevTensor, int]], out: DevTensor):
        self.m, self.segs, self.out = m, segs, out
        self.virtual = False

    def forward(self, training):
        if self.virtual:
            return
        off = 0
        for t, c in self.segs:
            call("seg_copy_channels", _p(t.buf), t.cs, 0, _p(self.out.buf), self.out.cs, off, c, t.npix, 0, _stream())
            off += c

    def plan_backward(self):
        self.live = self.out.gw and not self.virtual
        self.accs = [t.claim() if (self.live and t.needs_grad) else None for t, _c in self.segs]

    def backward(self):
        if not self.live:
            return
        off = 0
        for (t, c), acc in zip(self.segs, self.accs):
            if acc is not None:
                if not acc and t.cs > c:
                    t.grad.zero_()                     # keep the channel padding of a freshly written gradient at zero
                call("seg_copy_channels", _p(self.out.grad), self.out.cs, off, _p(t.grad), t.cs, 0, c, t.npix, acc, _stream())
            off += c


class AddOp(Op):
    """add([...]) / Add() (+ activation); also used with one input as a plain Activation layer."""

    def __init__(self, m, xs: List[DevTensor], act, alpha, out: DevTensor):
        self.m, self.xs, self.act, self.alpha, self.out = m, xs, act, alpha, out
        if len(xs) > 4:
            raise NotImplementedError("add of more than 4 tensors")
        self.ptrs = (C.c_void_p * 4)(*[x.buf.data_ptr() for x in xs])
        self.fused_bn: Optional[BNOp] = None          # the BatchNormalization that follows (fused tail): gets its statistics from here
        self.folded: List[Optional[BNOp]] = [None] * len(xs)      # inputs that are BatchNormalizations applied on the fly
        self.lazy: List[Optional[ConvOp]] = [None] * len(xs)      # inputs that are Conv_Blocks normalised on load (training passes)
        self.virtual_bn: Optional[BNOp] = None        # the folded BatchNormalization whose input is a virtual concatenation
        self.tail_grad: Optional[DevTensor] = None    # backward: the input whose gradient buffer holds THE gradient of the add
        self.lazy_descs, self.n_lazy_descs = None, 0  # backward: seg_lazy_bn of the Conv_Blocks whose BN-backward sums ride on the fan-out

    def fuse(self, bn: "BNOp"):
        """Argument sets of the fused forward launch: [training] reads lazy Conv_Blocks from their raw conv output, [inference]
        from their (then materialised) output; a virtual concatenation is not a dense input - it travels as the cat descriptor."""
        self.fused_bn = bn
        self.virtual_bn = next((b for b in self.folded if b is not None and b.pieces is not None), None)
        self.fsets = {}
        for training in (True, False):
            ptr, sc, sh, relu = [], [], [], []
            for x, b, cv in zip(self.xs, self.folded, self.lazy):
                if b is not None and b.pieces is not None:
                    continue
                if b is not None:
                    ptr.append(b.x.buf.data_ptr()); sc.append(b.scale.data_ptr()); sh.append(b.shift.data_ptr()); relu.append(0)
                elif cv is not None and training:
                    ptr.append(cv.z.buf.data_ptr()); sc.append(cv.scale.data_ptr()); sh.append(cv.shift.data_ptr())
                    relu.append(1 if ACT[cv.act] == ACT["relu"] else 0)
                else:
                    ptr.append(x.buf.data_ptr()); sc.append(None); sh.append(None); relu.append(0)
            pad = lambda v, fill=None: list(v) + [fill] * (4 - len(v))
            self.fsets[training] = ((C.c_void_p * 4)(*pad(ptr)), (C.c_void_p * 4)(*pad(sc)), (C.c_void_p * 4)(*pad(sh)),
                                    (C.c_int32 * 4)(*pad(relu, 0)), len(ptr))

    def forward(self, training):
        if self.fused_bn is not None:
            b, vb = self.fused_bn, self.virtual_bn
            ptr, sc, sh, relu, n = self.fsets[bool(training)]
            call("seg_add_act_stats_fwd", ptr, sc, sh, relu, n, C.byref(vb.cat) if vb is not None else None,
                 _p(vb.scale) if vb is not None else None, _p(vb.shift) if vb is not None else None, _p(self.out.buf), self.out.npix,
                 self.out.cs, self.out.C, ACT[self.act], float(self.alpha), _p(b.stats), b.nparts, _stream())
            return
        call("seg_add_act_fwd", self.ptrs, len(self.xs), _p(self.out.buf), self.out.buf.numel(), ACT[self.act],
             float(self.alpha), _stream())

    def plan_backward(self):
        self.live = self.out.gw
        if not self.live:
            return
        self.mask, gp = 0, []
        fused = self.fused_bn is not None
        # fused tail: the add hands ONE gradient to all its inputs.  It is written once, into the buffer of the input whose producer
        # runs its backward LAST (the first in forward order); a Conv_Block behind another input reads that buffer (ConvOp.grad_src)
        # and writes its own dz, a virtual concatenation's BatchNormalization reads it in seg_cat_bn_bwd_scatter.
        prod = [self.m.producer_of.get(id(x)) for x in self.xs]
        share = (fused and TAIL_SHARED_GRAD and len(self.xs) >= 2 and all(x.needs_grad and not x.gw for x in self.xs)
                 and prod[0] is not None and prod[0].has_bn
                 and all((p is not None and p.has_bn and self.m.ops.index(p) > self.m.ops.index(prod[0])) or (b is not None and b.pieces is not None)
                         for p, b in list(zip(prod, self.folded))[1:]))
        for k, x in enumerate(self.xs):
            b = self.folded[k] if fused else None
            if b is not None and b.pieces is not None:
                x.gw = True                   # (no buffer: the gradient of BN(cat)'s OUTPUT is the add's gradient itself)
                gp.append(None)
                if self.tail_grad is None and not share:
                    raise NotImplementedError("a virtual concatenation needs a dense sibling input that carries the add's gradient")
                continue
            if share and k > 0:
                x.gw = True
                x.ensure_grad()               # the Conv_Block's own dz lands here; dOut is read from the shared buffer
                prod[k].grad_src = self.xs[0]
                gp.append(None)
                continue
            if x.needs_grad:
                if x.claim():
                    self.mask |= 1 << k
                gp.append(x.grad.data_ptr())
                if k == 0:
                    self.tail_grad = x
            else:
                gp.append(None)
        self.gptrs = (C.c_void_p * 4)(*gp)
        # Conv_Blocks normalised on load whose dOut is exactly the gradient written here (no other contribution): their
        # BatchNormalization-backward sums come out of the fan-out launch (at most two: seg_bn_bwd_apply_fanout)
        if fused and TAIL_LAZY_SUMS and self.mask == 0:
            cvs = [cv for cv in self.lazy if cv is not None][:2]
            if cvs:
                parts = self.fused_bn.cat_parts
                arr = (_ffi.LazyBN * 2)()
                for d, cv in zip(arr, cvs):
                    d.z, d.scale, d.shift, d.mean, d.invstd = (cv.z.buf.data_ptr(), cv.scale.data_ptr(), cv.shift.data_ptr(),
                                                               cv.mean.data_ptr(), cv.invstd.data_ptr())
                    d.relu, d.partial = (1 if ACT[cv.act] == ACT["relu"] else 0), cv.bpart.data_ptr()
                    cv.tail_sums = parts
                self.lazy_descs, self.n_lazy_descs = arr, len(cvs)

    def backward(self):
        if self.live and self.fused_bn is None:       # (fused tail: the BatchNormalization's backward wrote the inputs' gradients)
            call("seg_add_act_bwd", _p(self.out.grad), _p(self.out.buf), self.gptrs, len(self.xs), self.out.buf.numel(),
                 ACT[self.act], float(self.alpha), self.mask, _stream())


class MulOp(Op):
    def __init__(self, m, a: DevTensor, r: DevTensor, out: DevTensor):
        self.m, self.a, self.r, self.out = m, a, r, out
        if r.C != 1:
            raise NotImplementedError("mul: the second operand must be a 1-channel map (Attention_Block)")

    def forward(self, training):
        call("seg_mul_bcast_fwd", _p(self.a.buf), _p(self.r.buf), _p(self.out.buf), self.a.npix, self.a.cs, self.r.cs, _stream())

    def plan_backward(self):
        self.live = self.out.gw
        if self.live:
            self.acc_a = self.a.claim() if self.a.needs_grad else None
            self.acc_r = self.r.claim() if self.r.needs_grad else None

    def backward(self):
        if self.live:
            call("seg_mul_bcast_bwd", _p(self.out.grad), _p(self.a.buf), _p(self.r.buf),
                 _p(self.a.grad) if self.acc_a is not None else None, _p(self.r.grad) if self.acc_r is not None else None,
                 self.a.npix, self.a.cs, self.r.cs, self.acc_a or 0, self.acc_r or 0, _stream())


class UpsampleOp(Op):
    def __init__(self, m, x: DevTensor, size, bilinear, out: DevTensor):
        self.m, self.x, self.size, self.bilinear, self.out = m, x, size, 1 if bilinear else 0, out

    def forward(self, training):
        x = self.x
        call("seg_upsample_fwd", _p(x.buf), _p(self.out.buf), x.N, x.H, x.W, x.cs, self.size[0], self.size[1], self.bilinear, _stream())

    def plan_backward(self):
        self.live = self.out.gw and self.x.needs_grad
        if self.live:
            self.acc = self.x.claim()

    def backward(self):
        if self.live:
            x = self.x
            call("seg_upsample_bwd", _p(self.out.grad), _p(x.grad), x.N, x.H, x.W, x.cs, self.size[0], self.size[1],
                 self.bilinear, self.acc, _stream())


class PoolOp(Op):
    def __init__(self, m, x: DevTensor, size, out: DevTensor):
        self.m, self.x, self.size, self.out = m, x, size, out

    def forward(self, training):
        x = self.x
        call("seg_maxpool_fwd", _p(x.buf), _p(self.out.buf), x.N, x.H, x.W, x.cs, self.size[0], self.size[1], _stream())

    def plan_backward(self):
        self.live = self.out.gw and self.x.needs_grad
        if self.live:
            self.acc = self.x.claim()

    def backward(self):
        if self.live:
            x = self.x
            call("seg_maxpool_bwd", _p(self.out.grad), _p(x.buf), _p(self.out.buf), _p(x.grad), x.N, x.H, x.W, x.cs,
                 self.size[0], self.size[1], self.acc, _stream())


# ------------------------------------------------------------------------------------------------ model
class SegModel:
    def __init__(self, graph: Graph, batch_size: Optional[int] = None, seed: int = 0, dtype: str = "f32"):
        """dtype: 'f32' = the parity mode - tensors stored as split fp16 pairs (x ~= hi + lo / 2048, ~22 significant bits), every
        product three fp16 MFMAs into fp32 accumulators: outputs within 1e-3 of the reference's fp32 arithmetic, equal masks;
        'f16' = fp16 activations / MFMA operands with fp32 accumulation (throughput mode: 2.3x faster, misses the 1e-3 tolerance).
        RANGE: both modes keep fp16's dynamic range - an input, activation or gradient with |x| >= 65504 becomes Inf / NaN (the
        reference's fp32 does not).  Inputs are checked (check_input_range: predict / fit raise), non-finite gradients are skipped and
        counted by the optimiser (nonfinite_updates); activations are bounded by BatchNormalization in every model of the reference."""
        if dtype not in ("f16", "f32"):
            raise ValueError("dtype must be 'f16' or 'f32'")
        self.graph = graph
        self.dtype = dtype
        self.seed = seed
        self.name = graph.name
        self.built = False
        self.training = True
        self._x = None
        self.adam_t = 0
        self.lr, self.beta_1, self.beta_2, self.epsilon = 2e-4, 0.9, 0.999, 1e-7
        self.loss_name = "mse"
        self.loss_weights = None       # per-output loss weights (compile); None = unit weights
        self.optimizer = "Adam"        # one of utils/tf_optimizers.py's names
        self._u_product = 1.0          # Nadam: running product of the momentum schedule
        self.metric_names: List[str] = []   # compile(metrics=[...]): names of utils/tf_metrics.py, evaluated by harness.fit
        self._label_flag = None
        self.world_size = 1
        self.comm_timing, self._comm_events = False, []     # bench.py --gpus N: event-timed waits for the gradient slices
        self.grad_dtype = "f32"        # what the data-parallel exchange moves: "f32" (the flat buffer itself) or "bf16" (store.G16)
        self.grad_target = GRAD_TARGET  # max |dLoss/dOut| is scaled into [target/2, target) before the fp16 backward pass
        self.allreduce = None          # set by dp.DataParallel: whole-buffer exchange after backward
        self.bucket_hook = None        # set by dp.DataParallel(overlap=True): fn(lo, hi) called as soon as G[lo:hi] is final
        self.bucket_waits = None       # ... and fn() -> [(lo, hi, work)] handing the pending all-reduces to the optimiser
        self._g_clean = False          # the flat gradient buffer is already zero (cleared on the side stream after Adam)
        self._param_events: list = []  # (first op index, event) of optimiser slices still running on the side stream
        self._graph_mode = False       # inside capture / replay of a hipGraph of the training step
        self._alpha_dev = None
        self._adam_slices = None
        self.side_stream = None        # second HIP stream for the weight-gradient kernels (created in build())
        self.use_side_stream = os.environ.get("SEG_WGRAD_STREAM", "1") != "0"
        self.bucket_elems = 32 << 20
        if batch_size is not None:
            self.build(batch_size)

    # ---------------------------------------------------------------- Keras-like surface
    @property
    def output_names(self):
        return list(self.graph.output_names)

    def count_params(self) -> int:
        return self.graph.count_params()

    def summary(self, print_fn=print):
        print_fn(self.graph.summary())

    def graphed_train_step(self, x: torch.Tensor, targets: Sequence[torch.Tensor], loss: Optional[str] = None):
        """Captures one optimiser step (forward, loss, backward, Adam, derived operands - both streams) into a hipGraph
        and returns step(x, targets) -> loss that replays it: one graph launch instead of ~150-270 kernel launches,
        which is what small models are bound by (64x64 w16 d3: 1.5 ms eager).  Single GPU; shapes and dtypes are fixed
        by the example batch; x / targets of later calls are copied into the captured input buffers."""
        if self.world_size != 1 or self.bucket_hook is not None:
            raise NotImplementedError("graph capture of the data-parallel step is not implemented")
        targets = list(targets) if isinstance(targets, (list, tuple)) else [targets]
        sx = self._check_x(x).clone()
        st_ = [t.to(device=_dev(), dtype=torch.float32).contiguous().clone() for t in targets]
        self._alpha_dev = torch.zeros(1, dtype=torch.float32, device=_dev())
        self._join_side()
        torch.cuda.synchronize()

        def set_alpha():
            self.adam_t += 1
            self._alpha_dev.fill_(self.lr * math.sqrt(1.0 - self.beta_2 ** self.adam_t) / (1.0 - self.beta_1 ** self.adam_t))

        # the warm-up below is a real optimiser step on the example batch (PyTorch's capture protocol needs one eager run):
        # weights, optimiser state, BN moving statistics and the step count are put back afterwards, so that capturing
        # does not train the model
        st = self.store
        snap = (st.P.clone(), st.M.clone(), st.V.clone(), st.S.clone(), self.adam_t, self._u_product)
        self._graph_mode = True
        try:
            warm = torch.cuda.Stream()
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):                     # one eager step on a side stream (PyTorch's capture protocol)
                set_alpha()
                self.train_step(sx, st_, loss)
            torch.cuda.current_stream().wait_stream(warm)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):                      # records the launches, runs nothing
                lv = self.train_step(sx, st_, loss)
        finally:
            self._graph_mode = False
        st.P.copy_(snap[0]); st.M.copy_(snap[1]); st.V.copy_(snap[2]); st.S.copy_(snap[3])
        self.adam_t, self._u_product = snap[4], snap[5]
        del snap
        self._refresh_copies()
        st.G.zero_()
        self._g_clean = True          # what the captured step expects on entry (it ends by clearing the buffer itself)
        torch.cuda.synchronize()

        def step(xn, tn):
            tn = list(tn) if isinstance(tn, (list, tuple)) else [tn]
            # an eager step before this replay may have left Adam slices / the operand rebuild running on the side stream,
            # or a backward pass without an optimiser step may have left gradients in the buffer
            self._join_side()
            if not self._g_clean:
                st.G.zero_()
            sx.copy_(self._check_x(xn))
            for d, t in zip(st_, tn):
                d.copy_(t)
            set_alpha()
            graph.replay()
            self._g_clean = True          # the replayed step ended by clearing the gradient buffer
            for op in self.ops:           # the replay moved the weights and the moving statistics on the device: whatever the host
                op.post_step()            # caches about them (the folded inference-mode BN coefficients) is stale
            return lv
        step.graph = graph
        return step

    def compile(self, loss="mse", optimizer="Adam", lr=2e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7, metrics=None, loss_weights=None):
        """model.compile(...) (2DCNN/Train.py:322-325).  loss: any of the 16 names `TFLosses` takes
        (utils/tf_losses.py:9-39, e.g. 'MeanSquaredLogarithmicError', the Train_Configs.ini default) or its short form
        ('mse', 'bce', 'cce', 'mae', 'msle', ...: _ffi.LOSS); mean reduction, applied to every output and summed with
        loss_weights (a list in output order or {output name: weight}; default: unit weights as in Train.py:322 - the 1D
        notebook passes 1 - 0.1 k).  optimizer: 'Adam' | 'Adadelta' | 'Adagrad' | 'Adamax' | 'FTRL' | 'Nadam' | 'RMSprop' |
        'SGD' with the fixed hyper-parameters of utils/tf_optimizers.py:10-26 (changing it restarts the state).  metrics: names of utils/tf_metrics.py (metrics.py lists the available ones); `fit`
        logs them under Keras' names (`mean_squared_error`, `val_mean_squared_error`, with several outputs
        `<output>_...`).  Optimiser: Keras Adam."""
        if loss not in LOSS:
            raise ValueError("Please select a valid loss function. Check for spelling mistakes, capital/small letters, etc.")
        if optimizer != "Adam" and optimizer not in OPTIMIZER:
            if str(optimizer).lower() == "adam":
                optimizer = "Adam"
            else:
                raise ValueError("Please select a valid optimizer function. Check for spelling mistakes, capital/small letters, etc.")
        if optimizer != self.optimizer:
            self.optimizer = optimizer
            self.adam_t = 0
            if self.built:
                self._init_optimizer_state()
        names = self.output_names
        if loss_weights is None:
            lw = [1.0] * len(names)
        elif isinstance(loss_weights, dict):
            unknown = [k for k in loss_weights if k not in names]
            if unknown:
                raise ValueError(f"loss_weights for unknown output(s) {unknown}; the outputs are {names}")
            lw = [float(loss_weights.get(n, 1.0)) for n in names]
        else:
            lw = [float(w) for w in loss_weights]
            if len(lw) != len(names):
                raise ValueError(f"{len(lw)} loss_weights for {len(names)} outputs")
        self.loss_weights = lw
        if metrics:
            from .metrics import log_names
            self.metric_names = [metrics] if isinstance(metrics, str) else list(metrics)
            try:
                log_names(self.metric_names, names)
            except KeyError as e:
                raise ValueError(f"metric {e.args[0]!r} is not available (see metrics.py)") from None
        else:
            self.metric_names = []
        self.loss_name, self.lr, self.beta_1, self.beta_2, self.epsilon = loss, lr, beta_1, beta_2, epsilon

    # ---------------------------------------------------------------- build / lowering
    def _enter(self):
        _Mode.sfx = "" if self.dtype == "f16" else "_f32"

    def build(self, batch_size: int, dtype: Optional[str] = None):
        if dtype is not None:
            if dtype not in ("f16", "f32"):
                raise ValueError("dtype must be 'f16' or 'f32'")
            self.dtype = dtype
        self._enter()
        DevTensor.dtype = torch.float16 if self.dtype == "f16" else torch.float32
        if not _DryRun.on:
            if not torch.cuda.is_available():
                raise RuntimeError("SegModel needs an MI355X: torch.cuda.is_available() is False and there is no CPU fallback")
            _ffi.lib()
        g = self.graph
        self.N = int(batch_size)
        self.tensors: Dict[str, DevTensor] = {}
        DevTensor.registry = self.tensors
        self.store = ParamStore()
        self.ops: List[Op] = []
        self.fwd_ws_users: list = []           # argument blocks of forward / data-gradient launches (split reductions share one workspace)
        self.heads: List[HeadOp] = []
        self.gs = torch.ones(2, dtype=torch.float32, device=_dev())        # [grad scale S, 1/S]
        self.amax = torch.zeros(1, dtype=torch.float32, device=_dev())
        self.loss_buf = torch.zeros(1, dtype=torch.float32, device=_dev())
        self.input_overflow = torch.zeros(1, dtype=torch.int32, device=_dev())     # input elements no fp16 / split-pair element can hold (seg_pack_input)
        cons = g.consumers()
        live = g.live_nodes()
        outset = set(g.outputs)
        val: Dict[int, object] = {}        # node idx -> DevTensor | list[(DevTensor, C)] (virtual concat)
        done = set()
        N = self.N

        def only_consumer(i, kind):
            c = cons[i]
            if len(c) == 1 and g.nodes[c[0]].kind == kind and i not in outset:
                return g.nodes[c[0]]
            return None

        def segs_of(i) -> List[Tuple[DevTensor, int]]:
            v = val[i]
            return v if isinstance(v, list) else [(v, v.C)]

        def tensor_of(i) -> DevTensor:
            v = val[i]
            if isinstance(v, list):
                raise NotImplementedError(f"node {g.nodes[i].name}: a concat feeding this layer type must be materialised "
                                          "(not implemented yet)")
            return v

        self._op_grad_range: Dict[int, Tuple[int, int]] = {}     # index in self.ops -> [lo, hi) of its slice of G
        n_ops_seen, n_train_seen = 0, 0

        def note_ranges():
            nonlocal n_ops_seen, n_train_seen
            if len(self.ops) > n_ops_seen:
                if self.store.n_train > n_train_seen:
                    # owner = the FIRST op this graph node appended (a node may lower to several ops, e.g. a wide head =
                    # conv + activation): the forward pass must have these parameters before that op runs, and the backward
                    # sweep (descending op index) has finished their gradients only after it
                    self._op_grad_range[n_ops_seen] = (n_train_seen, self.store.n_train)
                n_ops_seen, n_train_seen = len(self.ops), self.store.n_train

        for i in live:
            note_ranges()
            if i in done:
                continue
            n = g.nodes[i]
            k = n.kind
            if k == "input":
                t = DevTensor(n.name, N, n.shape[0], n.shape[1], n.shape[2])
                self.ops.append(InputOp(self, t, n.shape[2]))
                val[i] = t
            elif k == "concat":
                segs = []
                for j in n.inputs:
                    segs.extend(segs_of(j))
                if len(segs) > _ffi.MAX_SEG:
                    raise NotImplementedError(f"{n.name}: more than {_ffi.MAX_SEG} concat segments")
                val[i] = segs
            elif k in ("conv", "convT", "convlstm"):
                if k == "conv" and i in outset and n.attrs["kernel"] == (1, 1):
                    if n.shape[2] > 8:       # a wide head: GEMM-shaped, so the conv goes where the other convs go
                        logits = DevTensor(n.name + ":logits", N, n.shape[0], n.shape[1], n.shape[2])
                        self.ops.append(ConvOp(self, n, k, segs_of(n.inputs[0]), None, None, 0.0, logits, None, (1, 1), 0))
                        op = WideHeadOp(self, n, logits)
                    else:
                        op = HeadOp(self, n, tensor_of(n.inputs[0]))
                    self.ops.append(op)
                    self.heads.append(op)
                    val[i] = op
                    continue
                if k == "conv" and n.attrs["activation"] not in (None, "linear"):
                    raise NotImplementedError(f"{n.name}: activation inside a non-output conv")
                last = n
                chain = [n]
                bn = only_consumer(i, "bn")
                if bn is not None:
                    last = bn
                    chain.append(bn)
                act, alpha = None, 0.0
                an = only_consumer(last.idx, "act") if k != "convlstm" else None
                if an is not None and an.attrs["fn"] != "softmax":
                    act, alpha, last = an.attrs["fn"], an.attrs["alpha"], an
                    chain.append(an)
                F = n.attrs["filters"] if k == "convlstm" else 0
                if k == "convlstm" and (F % 8):
                    raise NotImplementedError(f"{n.name}: ConvLSTM filters must be a multiple of 8 (got {F})")
                out = DevTensor(last.name, N, last.shape[0], last.shape[1], last.shape[2])
                pooled, psize, pool_node = None, (1, 1), None
                for c in cons[last.idx]:
                    cn = g.nodes[c]
                    if cn.kind == "maxpool" and bn is not None and last.idx not in outset and pooled is None:
                        pool_node, psize = cn, cn.attrs["size"]
                        pooled = DevTensor(cn.name, N, cn.shape[0], cn.shape[1], cn.shape[2])
                        break
                op = ConvOp(self, n, k, segs_of(n.inputs[0]), bn, act, alpha, out, pooled, psize, F)
                self.ops.append(op)
                for nd in chain:
                    val[nd.idx] = out
                    done.add(nd.idx)
                if pool_node is not None:
                    val[pool_node.idx] = pooled
                    done.add(pool_node.idx)
            elif k == "dense_ae":
                # Flatten -> Dense(features) -> Dense(H*W*C') -> Reshape: the two Dense layers are 1x1 convs over a 1x1 image
                # whose channel axis is the flattened sample (GEMM rows = the batch), on the implicit-GEMM path
                x = self._materialise(val, n.inputs[0])
                H_, W_, Co = n.shape
                nin, F_, nout = x.H * x.W * x.C, int(n.attrs["features"]), H_ * W_ * Co
                flat = DevTensor(n.name + ":flat", N, 1, 1, nin)
                feat = DevTensor(n.name + ":features", N, 1, 1, F_)
                expd = DevTensor(n.name + ":expand", N, 1, 1, nout)
                out = DevTensor(n.name, N, H_, W_, Co)
                self.ops.append(ReshapeOp(self, x, flat, True))
                for src, dst, specs, cin, cout_ in ((flat, feat, n.params[0:2], nin, F_), (feat, expd, n.params[2:4], F_, nout)):
                    k4 = ParamSpec(specs[0].name, (1, 1, cin, cout_), specs[0].init, True, specs[0].fan)
                    fake = Node(-1, "conv", specs[0].name.rsplit("/", 1)[0], [], (1, 1, cout_),
                                dict(kernel=(1, 1), strides=(1, 1), activation=None), [k4, specs[1]])
                    self.ops.append(ConvOp(self, fake, "conv", [(src, cin)], None, None, 0.0, dst, None, (1, 1), 0))
                    slot = self.store.slots[specs[0].name]                      # the parameter keeps its Dense shape (in, out)
                    te, fe = slot.to_engine, slot.from_engine
                    slot.spec = specs[0]
                    slot.to_engine = lambda kern, te=te, c=cin, o=cout_: te(np.asarray(kern).reshape(1, 1, c, o))
                    slot.from_engine = lambda f, fe=fe, c=cin, o=cout_: np.asarray(fe(f)).reshape(c, o)
                self.ops.append(ReshapeOp(self, expd, out, False))
                val[i] = out
            elif k == "bn":
                act, alpha, last = None, 0.0, n
                an = only_consumer(i, "act")
                if an is not None and an.attrs["fn"] != "softmax":
                    act, alpha, last = an.attrs["fn"], an.attrs["alpha"], an
                x = self._materialise(val, n.inputs[0])
                out = DevTensor(last.name, N, *last.shape)
                self.ops.append(BNOp(self, n, x, act, alpha, out))
                self._fuse_bn_tail(self.ops[-1], n, cons)
                val[i] = val[last.idx] = out
                done.add(last.idx)
            elif k == "add":
                act, alpha, last = None, 0.0, n
                an = only_consumer(i, "act")
                if an is not None and an.attrs["fn"] != "softmax":
                    act, alpha, last = an.attrs["fn"], an.attrs["alpha"], an
                xs = [tensor_of(j) for j in n.inputs]
                out = DevTensor(last.name, N, *last.shape)
                self.ops.append(AddOp(self, xs, act, alpha, out))
                val[i] = val[last.idx] = out
                done.add(last.idx)
            elif k == "act":
                x = tensor_of(n.inputs[0])
                out = DevTensor(n.name, N, *n.shape)
                self.ops.append(AddOp(self, [x], n.attrs["fn"], n.attrs["alpha"], out))
                val[i] = out
            elif k == "mul":
                a, r = tensor_of(n.inputs[0]), tensor_of(n.inputs[1])
                out = DevTensor(n.name, N, *n.shape)
                self.ops.append(MulOp(self, a, r, out))
                val[i] = out
            elif k == "upsample":
                x = tensor_of(n.inputs[0])
                out = DevTensor(n.name, N, *n.shape)
                self.ops.append(UpsampleOp(self, x, n.attrs["size"], n.attrs["interpolation"] == "bilinear", out))
                val[i] = out
            elif k == "maxpool":
                x = tensor_of(n.inputs[0])
                out = DevTensor(n.name, N, *n.shape)
                self.ops.append(PoolOp(self, x, n.attrs["size"], out))
                val[i] = out
            else:
                raise NotImplementedError(f"layer kind {k!r} ({n.name}) is not lowered to HIP yet")
        note_ranges()
        for oi in g.outputs:
            if not isinstance(val[oi], HeadOp):
                raise NotImplementedError("model outputs must be 1x1 conv heads")
        self.out_heads = [val[oi] for oi in g.outputs]
        # ---- parameters: ops allocate in creation order, so each op owns one contiguous slice of the flat buffers
        self.store.allocate(self.dtype)
        for op in self.ops:
            if isinstance(op, ConvOp):
                op.finalize()
        self.src_buf = "P16" if self.dtype == "f16" else "P"      # the derived operands are transposed from the fp16 copy Adam refreshes
        self.gathers = [] if _DryRun.on else plan_gathers(self, self.ops)
        self._build_transpose_table()
        self.set_weights({p.name: init_param(p, self.seed) for p in g.param_specs()})
        # ---- static backward schedule (accumulate flags)
        self.producer_of = {id(op.out): op for op in self.ops if isinstance(op, ConvOp)}
        for op in reversed(self.ops):
            op.plan_backward()
        # ONE workspace for every split reduction: forward and data-gradient launches all run on the main stream, in order
        need = max([fwd_workspace_bytes(a) for a in self.fwd_ws_users] + [0])
        self.fwd_ws = torch.empty(need // 4, dtype=torch.float32, device=_dev()) if need and not _DryRun.on else None
        for a in self.fwd_ws_users:
            if a.splitk > 1:
                a.partial = self.fwd_ws.data_ptr() if self.fwd_ws is not None else None
        DevTensor.registry = None
        DevTensor.dtype = torch.float16
        if self.use_side_stream and not _DryRun.on:
            self.side_stream = torch.cuda.Stream()
        self.built = True
        self._init_optimizer_state()
        return self

    def _fuse_bn_tail(self, bop: "BNOp", n: Node, cons):
        """BN(ReLU(a + b)) - the tail of a MultiResBlock and of every ResPath iteration (unet_variants.py:97-99, :118-120) - as two
        fused passes each way instead of five / seven: the add kernel applies the ReLU and delivers the statistics this BN needs
        (no seg_bn_stats pass), an input that is itself a BatchNormalization with no other consumer (MultiResBlock: BN(cat)) is
        normalised on the fly inside the add (its output is never written), and this BN's backward writes [s > 0] * dx to every
        input of the add (no seg_add_act_bwd pass).  Needs: this BN without an activation of its own, the add's output feeding
        only this BN, every input of the add feeding only the add."""
        g = self.graph
        if not FUSE_BN_TAILS or bop.act is not None:
            return
        add = next((o for o in reversed(self.ops[:-1]) if isinstance(o, AddOp) and o.out is bop.x), None)
        if add is None or len(add.xs) < 2 or add.fused_bn is not None or ACT[add.act] not in (ACT[None], ACT["relu"]):
            return
        src = g.nodes[n.inputs[0]]                            # the activation node (or the add itself)
        addn = g.nodes[src.inputs[0]] if src.kind == "act" else src
        if addn.kind != "add" or list(cons[src.idx]) != [n.idx] or (src is not addn and list(cons[addn.idx]) != [src.idx]):
            return
        if any(list(cons[j]) != [addn.idx] for j in addn.inputs) or not all(x.needs_grad and x.cs == bop.x.cs for x in add.xs):
            return
        for k, x in enumerate(add.xs):
            b1 = next((o for o in self.ops if isinstance(o, BNOp) and o.out is x), None)
            if b1 is not None and b1.act is None and b1.fused_add is None and b1.folded_into is None:
                add.folded[k] = b1
                b1.folded_into = add
                b1.out.materialised = False                   # (never written: tests / tools must not read it)
                # BN over a concatenation (MultiResBlock's BN(cat[c3, c5, c7]), unet_variants.py:95-96) that only this add reads: the
                # concatenation is never materialised either - the tail kernels gather the pieces (a dense sibling input must carry
                # the add's gradient for the backward scatter: input 0, checked in AddOp.plan_backward)
                co = next((o for o in self.ops if isinstance(o, ConcatOp) and o.out is b1.x), None)
                bn1 = g.nodes[addn.inputs[k]]
                if (TAIL_VIRTUAL_CAT and TAIL_SHARED_GRAD and co is not None and k > 0 and len(co.segs) <= 4 and b1.C <= 2048
                        and list(cons[bn1.inputs[0]]) == [bn1.idx]
                        and any(isinstance(o, ConvOp) and o.out is add.xs[0] and o.has_bn for o in self.ops)):
                    co.virtual = True
                    b1.make_virtual(co.segs)
                    b1.out.buf = torch.empty((0,), dtype=b1.out.buf.dtype, device=b1.out.buf.device)
                continue
            # a Conv_Block (conv -> BN -> ReLU) whose only consumer is this add: normalised on load inside the add (training passes)
            cv = next((o for o in self.ops if isinstance(o, ConvOp) and o.out is x), None)
            if (TAIL_LAZY_INPUTS and cv is not None and cv.has_bn and cv.kind == "conv" and cv.pooled is None
                    and ACT[cv.act] in (ACT[None], ACT["relu"])):
                add.lazy[k] = cv
                cv.lazy = True
                cv.out.materialised = False                   # (written by inference passes only)
        bop.fused_add = add
        add.fuse(bop)

    def _build_transpose_table(self):
        """Device table of every derived (transposed) weight operand: refreshed by ONE launch after each optimiser step."""
        jobs = []
        for op in self.ops:
            if isinstance(op, ConvOp):
                jobs += op.transpose_jobs()
        for grp in self.gathers:
            jobs += grp.transpose_jobs()
        self.tr_n, self.tr_tiles, self.tr_table = len(jobs), 0, None
        src16 = 1 if self.src_buf == "P16" else 0
        if not jobs or _DryRun.on:
            return
        arr = (TransposeDesc * len(jobs))()
        for d, job in zip(arr, jobs):
            w, wt, rows, T, K, k0, kcount, rows_pad, dst_ld, dst_coff = job[:10]
            d.w, d.wt, d.rows, d.T, d.K, d.k0, d.kcount, d.rows_pad, d.dst_ld, d.dst_coff = w, wt, rows, T, K, k0, kcount, rows_pad, dst_ld, dst_coff
            d.dst_T, d.dst_t0 = job[10:12] if len(job) > 10 else (0, 0)
            d.tile0, d.src16 = self.tr_tiles, src16
            self.tr_tiles += T * _cdiv(kcount, 32) * _cdiv(rows_pad, 32)
        self.tr_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(_dev())

    def refresh_derived(self):
        """Re-derive the data-gradient operands from the master weights (after an optimiser step / weight import)."""
        if self.tr_table is not None:
            call("seg_transpose_w_multi", _p(self.tr_table), self.tr_n, self.tr_tiles, _stream())

    def _materialise(self, val, i) -> DevTensor:
        v = val[i]
        if isinstance(v, list):
            n = self.graph.nodes[i]
            out = DevTensor(n.name, self.N, *n.shape)
            self.ops.append(ConcatOp(self, v, out))
            val[i] = out
            return out
        return v

    # ---------------------------------------------------------------- weights
    def set_weights(self, weights: Dict[str, np.ndarray]):
        """weights: structural name -> numpy array in Keras layout (see graph.py)."""
        st = self.store
        if self.built:
            self._join_side()
        host_p = st.P.cpu().numpy()
        host_s = st.S.cpu().numpy()
        for name, arr in weights.items():
            s = st.slots.get(name)
            if s is None:
                raise KeyError(f"unknown parameter {name!r}")
            arr = np.asarray(arr, np.float32)
            if tuple(arr.shape) != tuple(s.spec.shape):
                raise ValueError(f"{name}: shape {arr.shape} != {s.spec.shape}")
            flat = np.asarray(s.to_engine(arr), np.float32).reshape(-1)
            assert flat.size == s.numel, (name, flat.size, s.numel)
            (host_p if s.trainable else host_s)[s.off:s.off + s.numel] = flat
        st.P.copy_(torch.from_numpy(host_p))
        st.S.copy_(torch.from_numpy(host_s))
        self._refresh_copies()

    def _join_side(self):
        if self.side_stream is not None:
            torch.cuda.current_stream().wait_stream(self.side_stream)
        self._param_events = []

    def get_weights(self) -> Dict[str, np.ndarray]:
        st = self.store
        self._join_side()
        host_p, host_s = st.P.cpu().numpy(), st.S.cpu().numpy()
        out = {}
        for name, s in st.slots.items():
            src = host_p if s.trainable else host_s
            out[name] = np.asarray(s.from_engine(src[s.off:s.off + s.numel]), np.float32).reshape(s.spec.shape)
        return out

    def save_weights(self, path: str):
        """.npz keyed by structural parameter names, Keras layouts (the reference saves .keras files, Train.py:375)."""
        from .harness import save_weights_npz
        save_weights_npz(self.get_weights(), path)

    def load_weights(self, path: str):
        """model.load_weights (Train.py:363,367)."""
        from .harness import load_weights_npz
        self.set_weights(load_weights_npz(path))

    def load_keras_model(self, keras_model):
        """Take over the weights of a tf.keras model built by the reference for the same constructor call: every Keras layer
        is matched to its structural name by walking `inbound_nodes` (keras_map.py), never by list position or auto-name."""
        from .keras_map import keras_weights_by_structural_name
        self.set_weights(keras_weights_by_structural_name(keras_model, self.graph))

    def export_to_keras_model(self, keras_model):
        """The other direction: write this model's weights into the matching layers of a tf.keras model."""
        from .keras_map import export_to_keras
        export_to_keras(self.get_weights(), keras_model, self.graph)

    def get_grads(self) -> Dict[str, np.ndarray]:
        """Gradients of the last backward pass, Keras layout (already divided by the internal gradient scale).  Valid
        between backward() and optimizer_step(): the optimiser step clears the buffer for the next backward pass."""
        st = self.store
        host = st.G.cpu().numpy()
        return {name: np.asarray(s.from_engine(host[s.off:s.off + s.numel]), np.float32).reshape(s.spec.shape)
                for name, s in st.slots.items() if s.trainable}

    def nonfinite_updates(self, reset: bool = False) -> int:
        """Parameter elements whose update the optimiser skipped since the last reset because their gradient was NaN or
        +-Inf (an overflow of the fp16 backward pass; Keras would have propagated the NaN into the weights).  Reading it
        synchronises with the device - harness.fit reads it once per step next to the loss and halves `grad_target` (the
        fp16 loss scale) when it is non-zero, the way dynamic loss scaling backs off."""
        self._join_side()
        n = int(self.store.nonfinite.cpu())
        if reset and n:
            self.store.nonfinite.zero_()
        return n

    def nonfinite_updates_async(self) -> int:
        """The same counter without stalling the training loop: returns what the PREVIOUS call collected (one step late, 0 the
        first time) and queues, behind the optimiser slices of the step just issued, a copy of the counter into pinned host memory
        followed by its reset.  Nothing joins the side stream, so the optimiser keeps running under the next forward pass; the side
        stream first catches up with the current one (an optimiser that ran there - a graph replay, the single-stream schedule -
        must have finished counting before the counter is copied and cleared).  `nonfinite_drain()` collects what is still pending."""
        if _DryRun.on:
            return 0
        prev = 0
        pend = getattr(self, "_nf_pending", None)
        if pend is not None:
            pend[0].synchronize()                  # (queued a whole step ago: long finished)
            prev = int(pend[1][0])
        if getattr(self, "_nf_host", None) is None:
            self._nf_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        cur = torch.cuda.current_stream()
        stream = self.side_stream if self.side_stream is not None else cur
        if stream is not cur:
            stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            self._nf_host.copy_(self.store.nonfinite, non_blocking=True)
            self.store.nonfinite.zero_()
            ev = torch.cuda.Event()
            ev.record(stream)
        self._nf_pending = (ev, self._nf_host)
        return prev

    def nonfinite_drain(self) -> int:
        """Everything the asynchronous counter has not reported yet - the value queued by the last nonfinite_updates_async() plus
        what the optimiser counted since - with one blocking read (end of an epoch / of fit: the logs and the loss-scale back-off
        see every step, also the last one)."""
        if _DryRun.on:
            return 0
        n = 0
        pend = getattr(self, "_nf_pending", None)
        if pend is not None:
            pend[0].synchronize()
            n += int(pend[1][0])
            self._nf_pending = None
        return n + self.nonfinite_updates(reset=True)

    def check_input_range(self):
        """Raises when an input element handed to forward() since the last check was NaN, +-Inf or |x| >= 65504: the reference's fp32
        tensors hold such values, this engine's fp16 / split-pair elements do not - they would turn into Inf / NaN downstream.
        (Reads a device counter: synchronises.  predict() and harness.fit call it where they synchronise anyway.)"""
        if _DryRun.on or not self.built:
            return
        n = int(self.input_overflow.cpu())
        if n:
            self.input_overflow.zero_()
            raise ValueError(f"{n} input element(s) are NaN / Inf or outside |x| < 65504, the range of this engine's fp16 / split-pair storage "
                             "(DESIGN.md section 2); normalise the input (the reference's loaders divide images by 255)")

    def _refresh_copies(self):
        st = self.store
        self._enter()
        if self.dtype == "f16":
            call("seg_cast_f32_f16", _p(st.P), _p(st.P16), st.P.numel(), _stream())
        else:
            call("seg_cast_f32_split", _p(st.P), _p(st.PS), st.P.numel(), _stream())
        for op in self.ops:
            op.post_step()
        self.refresh_derived()

    # ---------------------------------------------------------------- execution
    def _check_x(self, x):
        g = self.graph
        H, W, Cin = g.nodes[g.inputs[0]].shape
        if g.ndim == 1 and x.dim() == 3:
            x = x.unsqueeze(1)
        if not self.built:
            self.build(x.shape[0])
        if tuple(x.shape) != (self.N, H, W, Cin):
            raise ValueError(f"input shape {tuple(x.shape)} != {(self.N, H, W, Cin)} (NHWC; the batch size is fixed at build)")
        return x.to(device=_dev(), dtype=torch.float32).contiguous()

    def forward(self, x: torch.Tensor, training: Optional[bool] = None) -> List[torch.Tensor]:
        """x: (N,H,W,C) [2D] or (N,L,C) [1D] float32.  Returns [out, level1, ...] fp32 NHWC (views of internal buffers)."""
        training = self.training if training is None else training
        self._x = self._check_x(x)
        self._enter()
        ev, k = self._param_events, 0          # optimiser slices still being updated on the side stream (forward order)
        main = None if _DryRun.on else torch.cuda.current_stream()
        for i, op in enumerate(self.ops):
            while k < len(ev) and ev[k][0] <= i:       # the optimiser slice that holds this op's parameters has been updated
                main.wait_event(ev[k][1])
                k += 1
            op.forward(training)
        for _first, e in ev[k:]:
            main.wait_event(e)
        self._param_events = []
        outs = [h.y for h in self.out_heads]
        return [o.squeeze(1) for o in outs] if self.graph.ndim == 1 else outs

    __call__ = forward

    def predict(self, x, batch_size=None, verbose=0):
        """model.predict (2DCNN/Test.py:153,161): inference-mode BN.  Returns numpy, a list when the model has several outputs.
        Any number of samples: the engine's batch is fixed at build, so the input is processed in chunks of that size, the
        last one padded with zeros (inference-mode BN keeps samples independent, the padding rows are dropped)."""
        x = torch.as_tensor(np.asarray(x, np.float32)) if not torch.is_tensor(x) else x
        if not self.built:
            self.build(x.shape[0])
        n, N = x.shape[0], self.N
        if n == N:
            outs = [o.cpu().numpy() for o in self.forward(x, training=False)]
        else:
            parts = []
            for lo in range(0, n, N):
                xb = x[lo:lo + N]
                k = xb.shape[0]
                if k < N:
                    xb = torch.cat([xb, torch.zeros((N - k,) + tuple(xb.shape[1:]), dtype=xb.dtype, device=xb.device)], 0)
                parts.append([o[:k].cpu().numpy() for o in self.forward(xb, training=False)])
            outs = [np.concatenate([p[i] for p in parts], 0) for i in range(len(parts[0]))]
        self.check_input_range()
        return outs if len(outs) > 1 else outs[0]

    def adam_slices(self) -> List[Tuple[int, int, int]]:
        """[(lo, hi, first_op)]: op-aligned slices of the flat parameter buffers in forward order that tile [0, numel);
        an op's parameters are final once the slices with first_op <= its index have been updated."""
        if self._adam_slices is None:
            owners = sorted(self._op_grad_range.items())
            out, lo, first = [], None, None
            for oi, (a, b) in owners:
                if lo is None:
                    lo, first = a, oi
                # the first slices are what the next forward pass waits for at the step boundary (the main stream idles
                # until the first layer's weights are updated): they are kept short - first layer alone, then 256 K elements
                limit = ADAM_SLICE_ELEMS if (len(out) >= 2 or not ADAM_URGENT_FIRST) else (1 if not out else 1 << 18)
                if b - lo >= limit:
                    out.append([lo, b, first])
                    lo = None
            if lo is not None:
                out.append([lo, owners[-1][1][1], first])
            if not out:
                out = [[0, self.store.P.numel(), 0]]
            out[0][0], out[-1][1] = 0, self.store.P.numel()
            self._adam_slices = [tuple(o) for o in out]
        return self._adam_slices

    def grad_buckets(self) -> List[Tuple[int, int, int]]:
        """[(lo, hi, op_index)]: contiguous slices of the flat gradient buffer, each complete once the backward sweep
        (which walks the ops in reverse) has finished op `op_index`.  Slices follow op boundaries, are at least
        bucket_elems long (except the last) and tile [0, n_train) exactly."""
        owners = sorted(self._op_grad_range.items())          # ascending op index == ascending offsets
        out, lo, first = [], None, None
        for oi, (a, b) in owners:
            if lo is None:
                lo, first = a, oi
            # the first slice (the encoder's first layers) is what the next forward pass needs first and what the backward
            # sweep finishes last: with more than one rank it is kept small so that its exchange is short (DP_ASYNC)
            limit = min(self.bucket_elems, DP_URGENT_ELEMS) if (not out and DP_ASYNC and self.world_size > 1) else self.bucket_elems
            if b - lo >= limit:
                out.append((lo, b, first))
                lo = None
        if lo is not None:
            out.append((lo, owners[-1][1][1], first))
        return out

    def bucket_holds(self) -> List[bool]:
        """For every slice of grad_buckets(): hold its all-reduce back until the backward sweep has ended?  A collective that
        cannot finish before the sweep does would still be on the (in-order) RCCL stream when the more urgent slices - the
        encoder's, needed first by the next forward pass, finished last by this sweep - become ready, and delay them.  Held
        slices are issued after the sweep in forward order.  Estimate: ring all-reduce time 2 (w-1)/w x bytes / bus bandwidth
        (SEG_DP_BUSBW_GBS, default 300: xGMI rings are per-link bound), collectives queued one after the other, against the
        remaining GEMM work of the sweep at 600 TFLOP/s (fwd-shaped data gradient + weight gradient = 4 x MACs)."""
        bk = self.grad_buckets()
        w = self.world_size
        if w <= 1 or not DP_ASYNC:
            return [False] * len(bk)
        cost = [4.0 * op.M * op.T * op.Kpad * op.n_rows / 600e12 if isinstance(op, ConvOp) else 0.0 for op in self.ops]
        remaining = [0.0]
        for c in cost:
            remaining.append(remaining[-1] + c)                  # remaining[oi] = cost of the ops the sweep still has to run after op oi
        holds, busy_until = [False] * len(bk), -1e30             # times relative to the end of the sweep (negative = before it)
        for j in range(len(bk) - 1, -1, -1):                     # the order in which the sweep completes the slices
            lo, hi, oi = bk[j]
            start = max(-remaining[oi], busy_until)              # collectives run one after the other
            end = start + 2.0 * (w - 1) / w * (hi - lo) * 4 / (DP_BUSBW_GBS * 1e9)
            if end > 0.0:
                holds[j] = True
            else:
                busy_until = end
        return holds

    def backward(self, grads: Optional[Sequence[torch.Tensor]] = None):
        """Backward pass.  grads[i] = dLoss/d output_i (fp32, same shape as the output); None: use the gradients the
        loss kernels left in the heads' dy buffers."""
        if self.side_stream is not None and not self._graph_mode:         # (a captured step joins the side stream at its end)
            torch.cuda.current_stream().wait_stream(self.side_stream)      # derived operands rebuilt after the last optimiser step
        s = _stream()
        self._enter()
        if grads is not None:
            for h, gt in zip(self.out_heads, grads):
                h.dy.copy_(gt.reshape(h.dy.shape))
        self.amax.zero_()
        for h in self.out_heads:
            call("seg_absmax", _p(h.dy), h.dy.numel(), _p(self.amax), s)
        call("seg_grad_scale", _p(self.amax), float(self.grad_target), _p(self.gs), s)
        if not self._g_clean:
            self.store.G.zero_()          # (after an optimiser step the side stream has already cleared it)
        self._g_clean = False
        ready = {}
        if self.bucket_hook is not None:
            for (lo, hi, oi), hold in zip(self.grad_buckets(), self.bucket_holds()):
                ready.setdefault(oi, []).append((lo, hi, hold))

        def exchange(lo, hi):
            if self.side_stream is not None:
                # the slice holds weight gradients (side stream) and BN / bias gradients (this stream): issue the
                # exchange from the side stream after it has caught up with this one - this stream never stalls
                self.side_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.side_stream):
                    self.bucket_hook(lo, hi)
            else:
                self.bucket_hook(lo, hi)

        held = []
        for oi in range(len(self.ops) - 1, -1, -1):
            self.ops[oi].backward()
            for lo, hi, hold in ready.get(oi, ()):
                if hold:
                    held.append((lo, hi))          # see bucket_holds(): issued after the sweep, most urgent first
                else:
                    exchange(lo, hi)
        for lo, hi in sorted(held):
            exchange(lo, hi)
        if self.side_stream is not None:
            torch.cuda.current_stream().wait_stream(self.side_stream)

    def compute_loss(self, targets: Sequence[torch.Tensor], loss: Optional[str] = None) -> torch.Tensor:
        """Sum over outputs of weight x mean-reduced loss (unit weights unless compile(loss_weights=...), 2DCNN/Train.py:322);
        fills the heads' dy buffers.  'sparse_cce' takes one class index per pixel as its target."""
        kind = LOSS[loss or self.loss_name]
        weights = self.loss_weights or [1.0] * len(self.out_heads)
        self.loss_buf.zero_()
        s = _stream()
        for h, t, w in zip(self.out_heads, targets, weights):
            t = t.to(device=_dev(), dtype=torch.float32).contiguous()
            npix = h.y.numel() // h.cout
            if t.numel() != (npix if kind == LOSS["sparse_cce"] else h.y.numel()):
                raise ValueError(f"target for {h.node.name}: {tuple(t.shape)} vs output {tuple(h.y.shape)}")
            flag = None
            if kind in (LOSS["hinge"], LOSS["squared_hinge"]):       # Keras reads {0,1} labels as {-1,+1} iff ALL of them are 0/1
                if self._label_flag is None:
                    self._label_flag = torch.ones(1, dtype=torch.int32, device=_dev())
                flag = self._label_flag
                call("seg_labels_binary", _p(t), t.numel(), _p(flag), s)
            call("seg_loss_fwd_bwd_w", kind, _p(h.y), _p(t), npix, h.cout, float(w), _p(flag) if flag is not None else None,
                 _p(self.loss_buf), _p(h.dy), s)
        return self.loss_buf

    def _init_optimizer_state(self):
        """The slot variables as Keras creates them: zeros, except the accumulators of Adagrad and FTRL (0.1)."""
        st = self.store
        self._join_side()
        st.M.zero_()
        st.V.zero_()
        if self.optimizer == "Adagrad":
            st.V.fill_(0.1)
        elif self.optimizer == "FTRL":
            st.M.fill_(0.1)
        self._u_product = 1.0

    def _optimizer_scalars(self):
        """The five per-step host scalars of seg_optimizer_step (include/seg_hip.h) for step self.adam_t."""
        t, lr, eps = max(self.adam_t, 1), self.lr, 1e-7
        o = self.optimizer
        if o == "RMSprop":
            c = [lr, 0.9, eps]
        elif o == "Adadelta":
            c = [lr, 0.95, eps]
        elif o == "Adamax":
            c = [lr / (1.0 - 0.9 ** t), 0.9, 0.999, eps]
        elif o == "Nadam":
            u_t = 0.9 * (1.0 - 0.5 * 0.96 ** t)
            u_t1 = 0.9 * (1.0 - 0.5 * 0.96 ** (t + 1))
            prod_t = self._u_product * u_t
            self._u_product = prod_t
            c = [lr, u_t1 / (1.0 - prod_t * u_t1), (1.0 - u_t) / (1.0 - prod_t), 1.0 / (1.0 - 0.999 ** t), eps]
        else:                           # SGD, Adagrad, FTRL
            c = [lr, 0.0, eps]
        return (C.c_float * 5)(*(c + [0.0] * (5 - len(c))))

    def optimizer_step(self):
        st = self.store
        self._enter()
        if not self._graph_mode:          # (a captured step gets its step count / size from graphed_train_step)
            self.adam_t += 1
        a_t = self.lr * math.sqrt(1.0 - self.beta_2 ** max(self.adam_t, 1)) / (1.0 - self.beta_1 ** max(self.adam_t, 1))
        other = None
        if self.optimizer != "Adam":
            if self._graph_mode:
                raise NotImplementedError("a captured training step supports the Adam optimiser only")
            other = (OPTIMIZER[self.optimizer], self._optimizer_scalars())

        def adam(lo, hi, exchanged=False):
            off = lambda t: C.c_void_p(t.data_ptr() + lo * t.element_size())
            p16, pfmt = (off(st.P16), _ffi.OPERAND_F16) if self.dtype == "f16" else (off(st.PS), _ffi.OPERAND_SPLIT)
            # a slice that went through the bf16 exchange is read from there (the sum over the ranks), everything else from G
            if exchanged and self.grad_dtype == "bf16":
                g, gdt = off(st.G16), _ffi.GRAD_BF16
            else:
                g, gdt = off(st.G), _ffi.GRAD_F32
            if other is not None:        # the rest of the registry: same slices, same streams, another update rule
                call("seg_optimizer_step_ex", other[0], off(st.P), g, gdt, off(st.M), off(st.V), p16, pfmt, hi - lo, other[1], 1.0 / self.world_size,
                     _p(st.nonfinite), _stream())
            else:                        # (a captured step reads its step size from device memory, refreshed before every replay)
                dev = self._graph_mode
                call("seg_adam_step_ex", off(st.P), g, gdt, off(st.M), off(st.V), p16, pfmt, hi - lo, 1.0 if dev else a_t,
                     _p(self._alpha_dev) if dev else None, self.beta_1, self.beta_2, self.epsilon, 1.0 / self.world_size, _p(st.nonfinite), _stream())

        n = st.P.numel()
        if self.bucket_waits is not None and DP_ASYNC and self.side_stream is not None and not self._graph_mode:
            # overlapped data parallel, cross-iteration: the waits for the exchanged slices and their optimiser updates go to
            # the side stream, in forward order, each followed by an event; the next forward pass starts at once and every op
            # waits only for the slice holding its own parameters.  The encoder's small first slice is exchanged first after
            # the sweep (bucket_holds), so the big bottleneck slices travel under the encoder's forward pass.
            first_of = {lo: oi for lo, _hi, oi in self.grad_buckets()}
            side = self.side_stream
            side.wait_stream(torch.cuda.current_stream())
            events, pos = [], 0
            with torch.cuda.stream(side):
                for lo, hi, work in sorted(self.bucket_waits(), key=lambda t: t[0]):
                    self._timed_wait(work, side)   # stream-side for RCCL: the side stream waits for this collective
                    if lo > pos:
                        adam(pos, lo)
                    adam(lo, hi, True)
                    pos = hi
                    e = torch.cuda.Event()
                    e.record(side)
                    events.append((first_of.get(lo, 0), e))
                if pos < n:                        # alignment padding behind the last parameter
                    adam(pos, n)
                    e = torch.cuda.Event()
                    e.record(side)
                    events.append((len(self.ops), e))
            self._param_events = events
        elif self.bucket_waits is not None:
            # overlapped data parallel: update each slice of the flat buffers as soon as ITS all-reduce has finished (the
            # waits are stream-side), so the optimiser of the decoder's slices runs under the exchange of the encoder's
            done = []
            for lo, hi, work in self.bucket_waits():
                self._timed_wait(work, None)
                adam(lo, hi, True)
                done.append((lo, hi))
            done.sort()
            pos = 0
            for lo, hi in done + [(n, n)]:            # whatever no bucket covered (alignment padding behind the last parameter)
                if lo > pos:
                    adam(pos, lo)
                pos = max(pos, hi)
        elif self.allreduce is None and self.side_stream is not None and ADAM_ASYNC and not self._graph_mode:
            # single GPU: the update itself moves under the next forward pass - slices in forward order on the side stream,
            # an event per slice; forward() makes each op wait only for the slice holding its own parameters (Adam
            # streams 30 B per parameter from HBM while the forward GEMMs leave the HBM mostly idle)
            side = self.side_stream
            side.wait_stream(torch.cuda.current_stream())
            events = []
            with torch.cuda.stream(side):
                for lo, hi, first_op in self.adam_slices():
                    adam(lo, hi)
                    e = torch.cuda.Event()
                    e.record(side)
                    events.append((first_op, e))
            self._param_events = events
        else:
            if self.allreduce is not None:
                self.allreduce(st.G)
            adam(0, n)
        for op in self.ops:
            op.post_step()
        if self.side_stream is None:
            self.refresh_derived()
        else:
            # the transposed operands are first needed by the next backward pass: rebuild them on the side stream, under
            # the next forward pass (backward() joins the side stream before its first kernel)
            self.side_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side_stream):
                self.store.G.zero_()      # likewise the gradient accumulators of the next backward pass
                self._g_clean = True
                self.refresh_derived()
            if self._graph_mode:          # a captured step has to end with every forked stream joined
                torch.cuda.current_stream().wait_stream(self.side_stream)

    def _timed_wait(self, work, stream):
        """Wait for one gradient slice's collective.  With `comm_timing` set (bench.py --gpus N) the wait is bracketed by two
        events on the waiting stream: their distance is the time that stream sat idle for the exchange, i.e. communication
        the backward sweep / the optimiser of earlier slices did not cover."""
        if not getattr(self, "comm_timing", False) or _DryRun.on:
            work.wait()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream = stream if stream is not None else torch.cuda.current_stream()
        e0.record(stream)
        work.wait()
        e1.record(stream)
        self._comm_events.append((e0, e1))

    def exposed_comm_ms(self, reset=True) -> float:
        """Sum of the timed waits since the last call (see _timed_wait); synchronises."""
        torch.cuda.synchronize()
        ms = sum(e0.elapsed_time(e1) for e0, e1 in self._comm_events)
        if reset:
            self._comm_events = []
        return ms

    def train_step(self, x, targets, loss: Optional[str] = None) -> torch.Tensor:
        """One optimiser step = Keras' default train_step (Train.py:394-415): forward (training BN) -> loss -> backward
        -> [gradient all-reduce] -> Adam.  Returns the (device) scalar loss; nothing synchronises with the host."""
        self.forward(x, training=True)
        if not isinstance(targets, (list, tuple)):
            targets = [targets]
        lv = self.compute_loss(targets, loss)
        self.backward(None)
        self.optimizer_step()
        return lv
