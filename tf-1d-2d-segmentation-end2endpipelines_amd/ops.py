"""The fused launches a layer graph lowers to (SegModel.build, model.py): device tensors, the flat parameter store, and one Op class
per kind of launch group - Conv_Block (conv [+ BN + activation + pool | + ConvLSTM gates]), output heads, the BatchNormalization /
add / concatenate tails of MultiResBlock and ResPath, and the HBM-bound single-kernel layers - each with its forward, its place in
the static backward schedule (plan_backward) and its backward.  Every FLOP runs in libseg_hip.so through `call` (ctypes, _ffi.py)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _ffi
from . import flags as F
from ._ffi import ACT, Geom, IgemmFwdArgs, IgemmWgradArgs
from .graph import Node
from .planning import (_r8, _r32, _cdiv, geom_conv, geom_conv_dgrad, geom_convT, geom_convT_dgrad, pack_gemm_weight, unpack_gemm_weight,
                       fwd_halo_patch, plan_fwd, fits_lds_dma, wgrad_halo_patch, wgrad_halo_class_groups, pick_tile_wgrad, _tail_parts)
from .splitfmt import split_unpack


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


def mfma_tag(m, variant: int, wgrad: bool = False) -> str:
    """"|m<k>" for a GEMM launch's profile tag: fp16 MFMAs per product (bench.py's roofline object states the MFMA work issued)."""
    if m.dtype == "f16":
        return "|m1"
    if wgrad:
        return "|m1" if variant & _ffi.SEG_VARIANT_DZ_HI else "|m2" if variant & _ffi.SEG_VARIANT_IN_HI else "|m3"
    return "|m2" if variant & (_ffi.SEG_VARIANT_A_HI | _ffi.SEG_VARIANT_B_HI) else "|m3"


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

    gate_pad: Optional[Tuple[int, int]] = None      # (F, F8): the four ConvLSTM gate blocks are stored F8 channels apart (ConvOp)

    def _logical(self, v: torch.Tensor) -> torch.Tensor:
        v = v[..., :self.C]
        if self.gate_pad is not None:
            Fl, F8 = self.gate_pad
            v = torch.cat([v[..., gi * F8:gi * F8 + Fl] for gi in range(4)], dim=-1)
        return v

    def values(self) -> torch.Tensor:
        return self._logical(self.decode(self.buf))

    def grad_values(self) -> Optional[torch.Tensor]:
        return None if self.grad is None else self._logical(self.decode(self.grad))

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
        # ConvLSTM: the conv produces the four gate pre-activations [i | f | c | o].  The gate kernel walks 8-channel groups, so a
        # gate block is padded to F8 = F rounded up to 8 (the reference takes any filter count, unet_variants.py:148): engine
        # channel g * F8 + c is Keras column g * F + c, the padding rows have zero weights and bias, z = 0 there, h = 0.5 tanh(0) = 0
        # and their gradients are exactly zero, so they stay zero under every optimiser.
        self.F8 = _r8(lstm_F)
        self.cout = node.shape[2] if kind != "convlstm" else 4 * self.F8
        self.cout_logical = node.shape[2] if kind != "convlstm" else 4 * lstm_F
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
            if kind == "convlstm" and self.F8 != lstm_F:
                self.z.gate_pad = (lstm_F, self.F8)       # values() / grad_values() hand out the Keras gate layout [i | f | c | o] x F
        else:
            self.z = out
        assert self.z.cs == self.n_rows, (node.name, self.z.cs, self.n_rows)
        # ---- parameters
        cin_l = [c for _t, c in segs]
        n_rows, T, Kpad, woff, cout = self.n_rows, self.T, self.Kpad, list(self.woff), self.cout
        transposed = kind == "convT"

        Fl, F8 = lstm_F, self.F8
        gate_pad = kind == "convlstm" and Fl != F8

        def spread(a):      # [..., 4 F] (Keras gate columns) -> [..., 4 F8]
            o = np.zeros(a.shape[:-1] + (4 * F8,), a.dtype)
            for gi in range(4):
                o[..., gi * F8:gi * F8 + Fl] = a[..., gi * Fl:(gi + 1) * Fl]
            return o

        def gather(a):      # the inverse
            return np.concatenate([a[..., gi * F8:gi * F8 + Fl] for gi in range(4)], axis=-1)

        def to_engine(kern):
            if gate_pad:
                kern = spread(np.asarray(kern))
            return pack_gemm_weight(kern, cin_l, woff, n_rows, Kpad, transposed).reshape(-1)

        def from_engine(flat, shape=node.params[0].shape):
            if gate_pad:
                return gather(unpack_gemm_weight(flat, tuple(shape[:-1]) + (4 * F8,), cin_l, woff, n_rows, Kpad, transposed)).copy()
            return unpack_gemm_weight(flat, shape, cin_l, woff, n_rows, Kpad, transposed)

        kspec = node.params[0]
        self.wname = kspec.name
        st.add(kspec, n_rows * T * Kpad, to_engine, from_engine)
        self.bname = None
        for ps in node.params[1:]:
            if ps.name.endswith("/bias"):
                self.bname = ps.name
                if gate_pad:
                    st.add(ps, n_rows, lambda b: np.pad(spread(np.asarray(b)), (0, n_rows - 4 * F8)), lambda f: gather(f[:4 * F8]).copy())
                else:
                    st.add(ps, n_rows, lambda b: np.pad(b, (0, n_rows - b.shape[0])), lambda f, c=cout: f[:c].copy())
            else:   # ConvLSTM recurrent kernel: a weight of the Keras layer that the T = 1 forward never reads
                st.add(ps, int(np.prod(ps.shape)), lambda a: a.reshape(-1), lambda f, sh_=ps.shape: f.reshape(sh_).copy())
        if self.has_bn:
            for ps in bn.params:
                st.add(ps, n_rows, _pad_to(n_rows), lambda f, c=cout: f[:c].copy())
            self.bn_name = bn.name
            self.eps, self.momentum = float(bn.attrs["eps"]), float(bn.attrs["momentum"])
        # ---- tiles and scratch
        self.tm, self.tn, self.fwd_variant, self.fwd_splitk = plan_fwd(self.M, n_rows, self.g_fwd, self.g_fwd.ntaps * self.Kpad, m.dtype == "f16",
                                                                       [t.cs for t, _c in self.segs], n_rows * self.g_fwd.T_total * self.Kpad)
        self.nparts = self.g_fwd.ncls * _cdiv(self.M, self.tm)
        dev = _dev()
        if self.has_bn:
            self.stats = torch.zeros(self.nparts * 2 * n_rows, dtype=torch.float32, device=dev)
            self.mean, self.invstd, self.scale, self.shift, self.c1, self.c2 = (
                torch.zeros(n_rows, dtype=torch.float32, device=dev) for _ in range(6))
            self.bpart = torch.zeros(F.BN_PARTS_BWD * 2 * n_rows, dtype=torch.float32, device=dev)
        if not self.has_bn:
            self.bpart = torch.zeros(F.BN_PARTS_BWD * 2 * n_rows, dtype=torch.float32, device=dev)
            self.c1, self.c2 = torch.zeros(n_rows, dtype=torch.float32, device=dev), torch.zeros(n_rows, dtype=torch.float32, device=dev)
        self.need_dgrad = [t.needs_grad for t, _c in segs]
        # algorithmic FLOPs (2*MAC over logical channels): forward == weight gradient; data gradient per segment
        pix_taps = self.N * self.g_fwd.Hl * self.g_fwd.Wl * self.g_fwd.ncls * self.g_fwd.ntaps
        self.flops_fwd = 2.0 * pix_taps * sum(cin_l) * self.cout_logical
        self.flops_seg = [2.0 * pix_taps * c * self.cout_logical for c in cin_l]
        # algorithmic HBM bytes of a launch (logical channels, every operand once): input + output + weights, in the storage format
        esz = self.esz = 2 if m.dtype == "f16" else 4
        self.in_pix, self.out_pix = N * H * W, N * Ho * Wo
        self.bytes_fwd = float(esz * (self.in_pix * sum(cin_l) + self.out_pix * self.cout + self.T * sum(cin_l) * self.cout))
        self.bytes_seg = [float(esz * (self.in_pix * c + self.out_pix * self.cout + self.T * c * self.cout)) for c in cin_l]
        self.wt = torch.zeros(Kpad * self.T * self.rows_pad, dtype=DevTensor.dtype, device=dev) if any(self.need_dgrad) else None
        self.fwd_args = None
        self.gathered: Dict[int, "GatherDgrad"] = {}      # segment index -> gathered data gradient that covers it
        self.lazy = False            # training: `out` is never written - its only consumer (a fused add) normalises z on load
        self.grad_src: Optional[DevTensor] = None      # backward: dOut is read from this tensor's gradient (shared with a sibling)
        self.tail_sums = 0           # backward: > 0 = a fused tail's fan-out launch already left this BN's two sums in `bpart` (that many rows)
        self._piece_args = None
        self.cat_stats = None        # (BNOp, column): this block's output is a piece of a virtual concatenation whose BN takes the piece's sums from here
        self.bn_bwd_done = False     # backward: the tail also ran this block's finalize + apply (one pass for both blocks of a ResPath tail)
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
        if self.has_bn and F.FUSE_INFERENCE_BN:
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
        variant = F.WGRAD_VARIANT
        f16 = self.m.dtype == "f16"
        if (variant == 0 and F.WGRAD_HALO and wgrad_halo_patch(self.g_fwd)
                and fits_lds_dma(self.g_fwd, [t.cs for t, _c in self.segs], 0, f16, self.N * self.g_fwd.Hout * self.g_fwd.Wout * self.z.cs)):
            variant = 2         # 3x3 stride-1 layers and 4x4 / stride-2 transposed convs: LDS-DMA halo-tile kernel
        tm, tn, sk = pick_tile_wgrad(self.M, self.n_rows, [t.cs for t, _c in self.segs],
                                     wgrad_halo_class_groups(self.g_fwd) if (variant == 2 and (f16 or (F.WGRAD_TC2_F32 and self.n_rows > 32)))
                                     else self.g_fwd.ncls, self.g_fwd.ntaps,
                                     variant, f16)
        wg.tile_m, wg.tile_n, wg.splitk = tm, tn, sk
        wg.scale, wg.scale_ptr = 1.0, self.m.gs.data_ptr() + 4
        wg.variant = variant
        if self.m.bwd_w < 3 and variant == 2 and self.n_rows > 32:
            wg.variant |= _ffi.SEG_VARIANT_IN_HI         # the layer input as its hi half in the weight gradient - two MFMAs per product
            if self.m.bwd_w == 1:
                wg.variant |= _ffi.SEG_VARIANT_DZ_HI     # ... and dz: one
        # the gradient buffer is cleared once per step (backward() / the side stream after the optimiser), and with
        # splitk == 1 every tile has one writer: it may store instead of read-modify-write
        wg.overwrite = 1 if (F.WGRAD_OVERWRITE and variant == 2 and sk == 1) else 0
        # deterministic pixel split: the split blocks store partial tiles into this layer's own workspace (zero-filled once;
        # positions no block writes stay zero) and a second launch inside seg_igemm_wgrad adds them up in a fixed order
        self.wg_partial = None
        if F.WGRAD_DETERMINISTIC and sk > 1:
            self.wg_partial = torch.zeros(sk * self.n_rows * self.T * self.Kpad, dtype=torch.float32, device=_dev())
            wg.partial = self.wg_partial.data_ptr()
        # transposed convs on the class-pair kernel: the bias gradient (column sums of dz) comes out of the weight-gradient
        # launch as per-block partial rows instead of a separate pass over dz
        self.bias_from_wgrad = (F.BIAS_FROM_WGRAD and variant == 2 and self.bname is not None and not self.has_bn and self.kind == "convT"
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
            call("seg_igemm_fwd", C.byref(self.fwd_args_infer), s, flops=self.flops_fwd, nbytes=self.bytes_fwd, tag="fwd:" + self.node.name + mfma_tag(self.m, 0))
            if self.pooled is not None:
                call("seg_maxpool_fwd", _p(self.out.buf), _p(self.pooled.buf), self.out.N, self.out.H, self.out.W, self.out.cs,
                     self.pool_size[0], self.pool_size[1], s)
            return
        a = self.fwd_args if (training and self.has_bn) else self.fwd_args_eval
        if training and self.has_bn:
            self._infer_valid = False      # (this pass updates the moving statistics)
        call("seg_igemm_fwd", C.byref(a), s, flops=self.flops_fwd, nbytes=self.bytes_fwd, tag="fwd:" + self.node.name + mfma_tag(self.m, 0))
        if self.has_bn:
            bn = self.bn_name
            call("seg_bn_finalize", _p(self.stats), self.nparts, self.n_rows, self.z.npix, st.ptr(bn + "/gamma"),
                 st.ptr(bn + "/beta"), self.eps, self.momentum, st.ptr(bn + "/moving_mean"), st.ptr(bn + "/moving_variance"),
                 1 if self.m.graph.ndim == 2 else 0, 1 if training else 0, _p(self.mean), _p(self.invstd), _p(self.scale),
                 _p(self.shift), s)
            if self.lazy and training:
                return                   # (the fused add that consumes this block applies scale / shift / ReLU on load)
            ph, pw = self.pool_size if self.pooled is not None else (1, 1)
            if self.cat_stats is not None and training:
                # a piece of a virtual concatenation: the apply pass in its tail-kernel form (one pre-affine input) also delivers the
                # piece's columns of the statistics of the BatchNormalization over the concatenation
                b, off = self.cat_stats
                if self._piece_args is None:
                    self._piece_args = ((C.c_void_p * 4)(self.z.buf.data_ptr(), None, None, None), (C.c_void_p * 4)(self.scale.data_ptr(), None, None, None),
                                        (C.c_void_p * 4)(self.shift.data_ptr(), None, None, None), (C.c_int32 * 4)(1 if ACT[self.act] == ACT["relu"] else 0, 0, 0, 0))
                ptr, sc, sh, relu = self._piece_args
                call("seg_add_act_stats_fwd", ptr, sc, sh, relu, 1, None, None, None, _p(self.out.buf), self.z.npix, self.n_rows, self.cout, 0, 0.0,
                     _p(b.stats), b.nparts, b.C, off, s)
                return
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
            call("seg_convlstm_gate_fwd", _p(self.z.buf), self.z.cs, _p(self.out.buf), self.out.cs, self.z.npix, self.F8, s)
        elif self.pooled is not None:
            call("seg_maxpool_fwd", _p(self.out.buf), _p(self.pooled.buf), self.out.N, self.out.H, self.out.W, self.out.cs,
                 self.pool_size[0], self.pool_size[1], s)

    def _bias_grad(self, dz, s, gs_inv):
        """dbias += (1/S) * column sums of dz: deterministic per-block partials + one finalize launch (thousands of
        same-address float atomics serialise at one memory channel - measured 170 us per 134 MB tensor)."""
        if self.bname is None or self.bias_from_wgrad:
            return
        st = self.m.store
        call("seg_bn_stats", _p(dz), self.z.npix, self.z.cs, 0, self.n_rows, _p(self.bpart), F.BN_PARTS_BWD, self.n_rows, 0, s)
        call("seg_bn_bwd_finalize", _p(self.bpart), F.BN_PARTS_BWD, self.n_rows, self.z.npix, None, st.ptr(self.bname, "G"), 1.0, gs_inv,
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
        if (F.FUSE_ACT_BWD and not self.has_bn and self.kind != "convlstm" and ACT[self.act] in (1, 2, 3) and self.pool_acc is None
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
        self.dg_args, self.dg_bytes = [], []
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
            zero_first, fl, nb = [], 0.0, 0.0
            for o_i, (t, _c) in enumerate(self.segs[s_i:e_i]):
                a.out_span[o_i], a.out_cs_o[o_i] = _r32(t.cs), t.cs
                acc = t.claim((a, o_i))
                a.out_ptr[o_i], a.out_acc[o_i] = t.grad.data_ptr(), acc
                fl += self.flops_seg[s_i + o_i]
                nb += self.esz * (self.in_pix * _c + self.T * _c * self.cout)          # dx of the segment written, its weight slice read
                if self.strided and not acc:
                    zero_first.append(t)
            Md = self.N * self.g_dgrad.Hl * self.g_dgrad.Wl
            a.tile_m, a.tile_n, a.variant, a.splitk = plan_fwd(Md, width, self.g_dgrad, self.g_dgrad.ntaps * self.rows_pad, self.m.dtype == "f16",
                                                               [a.seg_cs[j] for j in range(a.nseg)], width * self.g_dgrad.T_total * self.rows_pad)
            if self.m.bwd_d == 2 and (((a.variant & 255) == 3 and self.g_dgrad.ntaps != 4) or
                                      ((a.variant & 255) == 0 and self.m.bwd_d_flag == _ffi.SEG_VARIANT_B_HI and F.BWD_B2_PER_TAP)):
                a.variant |= self.m.bwd_d_flag              # (one operand as its hi half: two MFMAs per product in the data gradients)
            self.m.fwd_ws_users.append(a)
            self.dg_args.append((a, zero_first, fl))
            self.dg_bytes.append(float(nb + self.esz * self.out_pix * self.cout))          # ... and dz read once
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
            if not self.bn_bwd_done:          # (else: dz is already in dA - AddOp.pair ran the finalize and one apply pass for both blocks)
                if not self.tail_sums:
                    call("seg_bn_act_bwd_reduce" + ("_inj" if inj is not None else ""), src, self.out.cs, 0, _p(self.z.buf), self.z.cs, 0,
                         self.z.npix, self.n_rows, _p(self.scale), _p(self.shift), _p(self.mean), _p(self.invstd), ACT[self.act],
                         float(self.alpha), _p(self.bpart), F.BN_PARTS_BWD, *((C.byref(inj),) if inj is not None else ()), s)
                call("seg_bn_bwd_finalize", _p(self.bpart), self.tail_sums or F.BN_PARTS_BWD, self.n_rows, self.z.npix,
                     st.ptr(bn + "/gamma", "G"), st.ptr(bn + "/beta", "G"), 1.0, gs_inv, _p(self.c1), _p(self.c2), s)
                call("seg_bn_act_bwd_apply" + ("_inj" if inj is not None else ""), src, self.out.cs, 0, _p(self.z.buf), self.z.cs, 0, _p(dA),
                     self.out.cs, 0, self.z.npix, self.n_rows, _p(self.scale), _p(self.shift), _p(self.mean), _p(self.invstd), _p(self.c1),
                     _p(self.c2), ACT[self.act], float(self.alpha), 0, *((C.byref(inj),) if inj is not None else ()), s)
            dz = dA
        elif self.kind == "convlstm":
            dz = self.z.ensure_grad()
            call("seg_convlstm_gate_bwd", _p(self.out.grad), self.out.cs, _p(self.z.buf), self.z.cs, _p(dz), self.z.npix, self.F8, s)
            self._bias_grad(dz, s, gs_inv)
        else:
            dz = self.out.grad
            if ACT[self.act] != 0 and not self.fused_act_bwd:
                call("seg_act_bwd", _p(dz), _p(self.out.buf), _p(dz), self.out.npix, self.out.cs, self.cout, ACT[self.act],
                     float(self.alpha), None, 1.0, None, s)
            self._bias_grad(dz, s, gs_inv)
        self.wg_args.dz = dz.data_ptr()
        def wgrad(stream):
            call("seg_igemm_wgrad", C.byref(self.wg_args), stream, flops=self.flops_fwd, nbytes=self.bytes_fwd, tag="wgrad:" + self.node.name + mfma_tag(self.m, self.wg_args.variant, True))
            if self.bias_from_wgrad:
                call("seg_bn_bwd_finalize", _p(self.wg_bpart), self.wg_bias_parts, self.n_rows, self.z.npix, None, st.ptr(self.bname, "G"),
                     1.0, gs_inv, _p(self.c1), _p(self.c2), stream)

        # the weight gradient is only needed by the optimiser: it runs on a second stream, next to this op's own data
        # gradients (dz is final here - forking before they are enqueued lets both start at once) and to the HBM-bound
        # BN / activation backward kernels of the layers that follow
        side = self.m.side_stream
        early = side is not None and F.WGRAD_FORK_EARLY
        if early:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                wgrad(_stream())
        for (a, zero_first, fl), nb in zip(self.dg_args, self.dg_bytes):
            for t in zero_first:
                t.grad.zero_()
            a.seg_ptr[0] = dz.data_ptr()
            call("seg_igemm_fwd", C.byref(a), s, flops=fl, nbytes=nb, tag="dgrad:" + self.node.name + mfma_tag(self.m, a.variant))
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
        a.tile_m, a.tile_n, a.variant, a.splitk = plan_fwd(Md, self.t.cs, op0.g_dgrad, op0.g_dgrad.ntaps * self.Kg, self.m.dtype == "f16",
                                                           [a.seg_cs[j] for j in range(a.nseg)], self.t.cs * op0.g_dgrad.T_total * self.Kg)
        if self.m.bwd_d == 2 and (((a.variant & 255) == 3 and op0.g_dgrad.ntaps != 4) or
                                  ((a.variant & 255) == 0 and self.m.bwd_d_flag == _ffi.SEG_VARIANT_B_HI and F.BWD_B2_PER_TAP)):
            a.variant |= self.m.bwd_d_flag
        self.m.fwd_ws_users.append(a)
        self.args = a
        self.flops = sum(op.flops_seg[si] for op, si in self.members)
        op0_ = self.members[0][0]
        self.nbytes = float(op0_.esz * (op0_.in_pix * self.t.C + sum(op.out_pix * op.cout + op.T * self.t.C * op.cout for op, _si in self.members)))

    def launch(self, s):
        a = self.args
        for j, (op, _si) in enumerate(self.members):
            a.seg_ptr[j] = op.dz_ptr
        call("seg_igemm_fwd", C.byref(a), s, flops=self.flops, nbytes=self.nbytes, tag="dgrad:gather:" + self.t.name + mfma_tag(self.m, a.variant))


def plan_gathers(m, ops) -> List[GatherDgrad]:
    """Tensors with >= 2 consuming 3x3 stride-1 convs whose data-gradient geometry the halo-tile kernel takes."""
    if not F.GATHER_DGRAD:
        return []
    users: Dict[int, list] = {}
    tens: Dict[int, DevTensor] = {}
    for op in ops:
        if not isinstance(op, ConvOp) or op.kind != "conv" or op.wt is None:
            continue
        kern, unit = tuple(op.node.attrs["kernel"]), tuple(op.node.attrs.get("strides", (1, 1))) == (1, 1)
        if kern == (1, 1) and unit and F.GATHER_DGRAD_1X1:
            # the 1x1 shortcut next to a 3x3 conv on the same tensor: a member only where the layer is HBM-bound (few channels on a
            # large image - there the read-modify-write of its own launch is what it costs: level 0 of C4 154 us against ~60 as one
            # more K segment of the sibling's launch); deeper levels are MFMA-bound and eight zero taps per product would lose
            if not (len(op.segs) == 1 and op.segs[0][0].cs <= F.GATHER_1X1_MAX_C and op.n_rows <= F.GATHER_1X1_MAX_C):
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
            if F.HEAD_FUSE_FWD:
                self.prod = prod
                prod.heads.append(self)
            self.inject = F.HEAD_INJECT
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

        side = self.m.side_stream if F.HEAD_DW_SIDE else None
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
        self.img, self.flat = img, flat
        self.hwc = img.H * img.W * img.C
        assert flat.C == self.hwc and flat.cs >= self.hwc
        # flat.cs == H*W*C: the whole batch is ONE copy (the flat tensor read as N*H*W rows of C elements).  Otherwise (H*W*C not a
        # multiple of 8: the row of a sample is padded) one copy per sample, each from / to the start of that sample's row; the
        # padding of the flat row is never written and stays zero
        self.per_sample = flat.cs != self.hwc

    def _copy(self, src_is_img: bool, src, dst, accumulate: int):
        img, flat = self.img, self.flat
        esz = src.element_size()
        if not self.per_sample:
            if src_is_img:
                call("seg_copy_channels", _p(src), img.cs, 0, _p(dst), img.C, 0, img.C, img.npix, accumulate, _stream())
            else:
                call("seg_copy_channels", _p(src), img.C, 0, _p(dst), img.cs, 0, img.C, img.npix, accumulate, _stream())
            return
        hw = img.H * img.W
        for n in range(img.N):
            io, fo = n * hw * img.cs * esz, n * flat.cs * esz
            if src_is_img:
                call("seg_copy_channels", C.c_void_p(src.data_ptr() + io), img.cs, 0, C.c_void_p(dst.data_ptr() + fo), img.C, 0, img.C, hw,
                     accumulate, _stream())
            else:
                call("seg_copy_channels", C.c_void_p(src.data_ptr() + fo), img.C, 0, C.c_void_p(dst.data_ptr() + io), img.cs, 0, img.C, hw,
                     accumulate, _stream())

    def forward(self, training):
        self._copy(self.to_flat, self.x.buf, self.out.buf, 0)

    def plan_backward(self):
        self.live = self.out.gw
        self.acc = self.x.claim() if (self.live and self.x.needs_grad) else None

    def backward(self):
        if not self.live or self.acc is None:
            return
        self._copy(not self.to_flat, self.out.grad, self.x.grad, self.acc)


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
        self.bpart = torch.zeros(F.BN_PARTS_BWD * 2 * self.C, dtype=torch.float32, device=dev)
        self.mean, self.invstd, self.scale, self.shift, self.c1, self.c2 = (
            torch.zeros(self.C, dtype=torch.float32, device=dev) for _ in range(6))
        # fused tails (F.FUSE_BN_TAILS, see SegModel._fuse_bn_tail): `fused_add` - this BN follows ReLU(a + b): the add kernel delivers
        # the statistics and this op's backward writes the masked gradient to the add's inputs; `folded_into` - this BN feeds such an
        # add and has no other consumer: its normalisation is applied inside the add kernel, `out` is never written;
        # `pieces` - (folded only) the input is a concatenation that is never materialised: the tail kernels gather it (seg_cat_desc)
        self.fused_add: Optional["AddOp"] = None
        self.folded_into: Optional["AddOp"] = None
        self.pieces: Optional[List[Tuple[DevTensor, int]]] = None
        self.piece_stats = False      # (virtual) the pieces' producing Conv_Blocks write this BN's statistics from their apply passes
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
        # the statistics of the concatenation are the statistics of the pieces side by side: where every piece is the output of a
        # Conv_Block (plain bn_act_apply, no pooling, no head riding on it) that block's apply pass - run as the one-input form of the
        # tail kernel - writes the piece's sums into this BN's partial rows on its way out: no pass over the pieces afterwards
        prods = [next((o for o in self.m.ops if isinstance(o, ConvOp) and o.out is t), None) for t, _c in pieces]
        if F.TAIL_PIECE_STATS and all(p is not None and p.has_bn and p.pooled is None and not p.lazy and not p.heads and p.cat_stats is None
                                      for p in prods):
            off = 0
            for p, (_t, c) in zip(prods, pieces):
                p.cat_stats = (self, off)
                off += c
            self.piece_stats = True
        self.x.materialised = False
        self.x.buf = torch.empty((0,), dtype=self.x.buf.dtype, device=self.x.buf.device)      # (nothing reads or writes the concatenated tensor any more)

    def forward(self, training):
        st, s, bn, x = self.m.store, _stream(), self.bn_name, self.x
        if training and self.fused_add is None:
            if self.pieces is not None and self.piece_stats:
                pass                          # every piece delivered its columns of `stats` from its own apply pass (ConvOp.cat_stats)
            elif self.pieces is not None:     # statistics of the virtual concatenation: one gathering pass over the pieces
                call("seg_add_act_stats_fwd", None, None, None, None, 0, C.byref(self.cat), None, None, None, x.npix, self.C, self.cl,
                     0, 0.0, _p(self.stats), self.nparts, 0, 0, s)
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
             _p(self.mean), _p(self.invstd), ACT[self.act], float(self.alpha), _p(self.bpart), F.BN_PARTS_BWD, s)
        call("seg_bn_bwd_finalize", _p(self.bpart), F.BN_PARTS_BWD, self.C, x.npix, st.ptr(bn + "/gamma", "G"),
             st.ptr(bn + "/beta", "G"), 1.0, gs_inv, _p(self.c1), _p(self.c2), s)
        if self.fused_add is not None:
            a = self.fused_add                # dx = [x > 0] sc (g - c1 - xhat c2) to every input of the add: no add backward pass
            vb = a.virtual_bn                 # ... and, with a virtual concatenation among the inputs, the sums of ITS BatchNormalization
            call("seg_bn_bwd_apply_fanout", _p(dA), _p(x.buf), a.gptrs, len(a.xs), a.mask, x.npix, self.C, _p(self.scale), _p(self.mean),
                 _p(self.invstd), _p(self.c1), _p(self.c2), 1 if ACT[a.act] == ACT["relu"] else 0,
                 C.byref(vb.cat) if vb is not None else None, _p(vb.mean) if vb is not None else None,
                 _p(vb.invstd) if vb is not None else None, _p(vb.bpart) if vb is not None else None,
                 a.lazy_descs, a.n_lazy_descs, self.cat_parts, s)
            if a.pair is not None:
                # both Conv_Blocks behind the add read this one gradient: their finalize launches, then ONE pass that reads it
                # once and writes both dz (ConvOp.backward then goes straight to the weight / data gradients)
                for cv in a.pair:
                    call("seg_bn_bwd_finalize", _p(cv.bpart), cv.tail_sums, cv.n_rows, cv.z.npix, st.ptr(cv.bn_name + "/gamma", "G"),
                         st.ptr(cv.bn_name + "/beta", "G"), 1.0, gs_inv, _p(cv.c1), _p(cv.c2), s)
                call("seg_bn_act_bwd_apply2", _p(a.tail_grad.grad), a.pair_sets, x.npix, self.C, s)
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
        self.pair, self.pair_sets = None, None        # backward: the two Conv_Blocks whose dz come out of one pass over the shared gradient

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
                 self.out.cs, self.out.C, ACT[self.act], float(self.alpha), _p(b.stats), b.nparts, 0, 0, _stream())
            return
        call("seg_add_act_fwd", self.ptrs, len(self.xs), _p(self.out.buf), self.out.buf.numel(), self.out.cs, self.out.C, ACT[self.act],
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
        share = (fused and F.TAIL_SHARED_GRAD and len(self.xs) >= 2 and all(x.needs_grad and not x.gw for x in self.xs)
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
        if fused and F.TAIL_LAZY_SUMS and self.mask == 0:
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
                # the ResPath form - two blocks, one shared gradient buffer (the second block reads the first one's): one apply pass
                if (F.TAIL_PAIR_APPLY and share and len(cvs) == 2 and len(self.xs) == 2 and cvs[1].grad_src is self.xs[0]
                        and cvs[0].out is self.xs[0] and all(cv.head_inj is None for cv in cvs)):
                    sets = (_ffi.BnBwdSet * 2)()
                    for d, cv in zip(sets, cvs):
                        d.x, d.dx = cv.z.buf.data_ptr(), cv.out.ensure_grad().data_ptr()
                        d.scale, d.shift, d.mean, d.invstd = cv.scale.data_ptr(), cv.shift.data_ptr(), cv.mean.data_ptr(), cv.invstd.data_ptr()
                        d.c1, d.c2, d.act, d.alpha = cv.c1.data_ptr(), cv.c2.data_ptr(), ACT[cv.act], float(cv.alpha)
                        cv.bn_bwd_done = True
                    self.pair, self.pair_sets = cvs, sets

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
