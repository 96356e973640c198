"""Layer-graph IR for the UNet-family hot path.

The reference builds a tf.keras functional graph (2D: `TensorFlow/2DCNN/models/unet_variants.py:7-154,
277-343,459-487,750-792,936-1115`; 1D: `TensorFlow/1DCNN/Models/unet_variants.py:53-319,544-645`,
`BCDUNet.py:6-174`).  This module holds the same information as a flat list of nodes so that the HIP
engine can lower it to fused kernels.  Pure Python, no torch, no device work.

Conventions
-----------
* activations are NHWC; a 1D signal of length L is the 2D tensor H=1, W=L
* every node has a *structural* name (``enc3/conv``, ``X12/bn`` ...) chosen by the builder, never a
  Keras auto-name, so that weights can be exchanged by name
* parameters are stored in **Keras layout**:
    conv      kernel (kh, kw, Cin, Cout)   bias (Cout)
    convT     kernel (kh, kw, Cout, Cin)   bias (Cout)
    bn        gamma, beta, moving_mean, moving_var (C)
    convlstm  kernel (kh, kw, Cin, 4F)  recurrent_kernel (kh, kw, F, 4F)  bias (4F)
    dense     kernel (in, out)  bias (out)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple


@dataclass
class ParamSpec:
    name: str                 # "<node>/<slot>", e.g. "enc1/conv/kernel"
    shape: Tuple[int, ...]
    init: str                 # he_uniform | glorot_uniform | he_normal | orthogonal | zeros | ones | unit_forget_bias
    trainable: bool = True
    fan: Tuple[int, int] = (0, 0)   # (fan_in, fan_out) used by the initialisers


@dataclass
class Node:
    idx: int
    kind: str
    name: str
    inputs: List[int]
    shape: Tuple[int, int, int]          # (H, W, C) without the batch axis
    attrs: Dict[str, object] = field(default_factory=dict)
    params: List[ParamSpec] = field(default_factory=list)


class Sym:
    """Symbolic tensor handle returned by Graph layer calls."""
    __slots__ = ("g", "idx")

    def __init__(self, g: "Graph", idx: int):
        self.g, self.idx = g, idx

    @property
    def shape(self) -> Tuple[int, int, int]:
        return self.g.nodes[self.idx].shape

    @property
    def node(self) -> Node:
        return self.g.nodes[self.idx]

    def __repr__(self):
        n = self.node
        return f"Sym({n.name}:{n.kind}{n.shape})"


def _pair(v) -> Tuple[int, int]:
    if isinstance(v, (tuple, list)):
        return (int(v[0]), int(v[1]))
    return (int(v), int(v))


def _check_filters(filters, name):
    """Keras refuses a convolution without output channels when the layer is constructed (keras/layers/convolutional/base_conv.py:
    "Invalid value for argument `filters`. Expected a strictly positive value."): MultiResUNet with alpha * model_width < 6 asks for
    int(W * 0.167) = 0 filters (unet_variants.py:89) - the same ValueError here, not a division by zero in the planner."""
    if int(filters) <= 0:
        raise ValueError(f"Invalid value for argument `filters`. Expected a strictly positive value. Received filters={filters} ({name}).")


class Graph:
    """Append-only layer graph with Keras shape inference for the ops the hot path uses."""

    def __init__(self, ndim: int = 2, name: str = "model"):
        assert ndim in (1, 2)
        self.ndim = ndim
        self.name = name
        self.nodes: List[Node] = []
        self.inputs: List[int] = []
        self.outputs: List[int] = []
        self.output_names: List[str] = []
        self._names = set()

    # ------------------------------------------------------------------ plumbing
    def _add(self, kind, name, inputs: Sequence[Sym], shape, attrs=None, params=None) -> Sym:
        if name in self._names:
            raise ValueError(f"duplicate structural name {name!r}")
        self._names.add(name)
        n = Node(len(self.nodes), kind, name, [s.idx for s in inputs], tuple(int(v) for v in shape),
                 dict(attrs or {}), list(params or []))
        self.nodes.append(n)
        return Sym(self, n.idx)

    def set_outputs(self, syms: Sequence[Sym], names: Sequence[str]):
        self.outputs = [s.idx for s in syms]
        self.output_names = list(names)

    # ------------------------------------------------------------------ layers
    def input(self, h, w, c, name="input") -> Sym:
        s = self._add("input", name, [], (h, w, c))
        self.inputs.append(s.idx)
        return s

    def conv(self, x: Sym, filters, kernel, name, strides=1, use_bias=True, activation=None,
             init="glorot_uniform") -> Sym:
        """Conv1D/Conv2D, padding='same' (Keras: out = ceil(in/stride))."""
        kh, kw = _pair(kernel) if self.ndim == 2 else (1, int(kernel if not isinstance(kernel, (tuple, list)) else kernel[0]))
        sh, sw = _pair(strides) if self.ndim == 2 else (1, int(strides if not isinstance(strides, (tuple, list)) else strides[0]))
        H, W, C = x.shape
        _check_filters(filters, name)
        Ho, Wo = -(-H // sh), -(-W // sw)
        fan = (kh * kw * C, kh * kw * filters)
        ps = [ParamSpec(f"{name}/kernel", (kh, kw, C, filters), init, True, fan)]
        if use_bias:
            ps.append(ParamSpec(f"{name}/bias", (filters,), "zeros", True, fan))
        return self._add("conv", name, [x], (Ho, Wo, filters),
                         dict(kernel=(kh, kw), strides=(sh, sw), use_bias=use_bias, activation=activation), ps)

    def conv_transpose(self, x: Sym, filters, kernel, strides, name, init="glorot_uniform") -> Sym:
        """Conv1DTranspose/Conv2DTranspose, padding='same' (out = in * stride)."""
        kh, kw = _pair(kernel) if self.ndim == 2 else (1, int(kernel if not isinstance(kernel, (tuple, list)) else kernel[0]))
        sh, sw = _pair(strides) if self.ndim == 2 else (1, int(strides if not isinstance(strides, (tuple, list)) else strides[0]))
        H, W, C = x.shape
        _check_filters(filters, name)
        # keras fan computation for a (kh,kw,Cout,Cin) kernel: receptive*shape[-2] , receptive*shape[-1]
        fan = (kh * kw * filters, kh * kw * C)
        ps = [ParamSpec(f"{name}/kernel", (kh, kw, filters, C), init, True, fan),
              ParamSpec(f"{name}/bias", (filters,), "zeros", True, fan)]
        return self._add("convT", name, [x], (H * sh, W * sw, filters),
                         dict(kernel=(kh, kw), strides=(sh, sw)), ps)

    def bn(self, x: Sym, name, eps=1e-3, momentum=0.99) -> Sym:
        C = x.shape[2]
        ps = [ParamSpec(f"{name}/gamma", (C,), "ones"), ParamSpec(f"{name}/beta", (C,), "zeros"),
              ParamSpec(f"{name}/moving_mean", (C,), "zeros", False),
              ParamSpec(f"{name}/moving_variance", (C,), "ones", False)]
        return self._add("bn", name, [x], x.shape, dict(eps=eps, momentum=momentum), ps)

    def act(self, x: Sym, fn: str, name, alpha=0.0) -> Sym:
        return self._add("act", name, [x], x.shape, dict(fn=fn, alpha=alpha))

    def maxpool(self, x: Sym, size, name) -> Sym:
        ph, pw = _pair(size) if self.ndim == 2 else (1, int(size))
        H, W, C = x.shape
        return self._add("maxpool", name, [x], (H // ph, W // pw, C), dict(size=(ph, pw)))

    def upsample(self, x: Sym, size, name, interpolation="nearest") -> Sym:
        fh, fw = _pair(size) if self.ndim == 2 else (1, int(size))
        H, W, C = x.shape
        return self._add("upsample", name, [x], (H * fh, W * fw, C), dict(size=(fh, fw), interpolation=interpolation))

    def concat(self, xs: Sequence[Sym], name) -> Sym:
        H, W, _ = xs[0].shape
        for s in xs:
            if s.shape[:2] != (H, W):
                raise ValueError(f"concat {name}: spatial mismatch {[t.shape for t in xs]}")
        return self._add("concat", name, xs, (H, W, sum(s.shape[2] for s in xs)))

    def add(self, xs: Sequence[Sym], name) -> Sym:
        for s in xs:
            if s.shape != xs[0].shape:
                raise ValueError(f"add {name}: shape mismatch {[t.shape for t in xs]}")
        return self._add("add", name, xs, xs[0].shape)

    def mul(self, a: Sym, b: Sym, name) -> Sym:
        """a * b with b broadcast over channels when b has a single channel."""
        if a.shape[:2] != b.shape[:2] or b.shape[2] not in (1, a.shape[2]):
            raise ValueError(f"mul {name}: {a.shape} * {b.shape}")
        return self._add("mul", name, [a, b], a.shape)

    def convlstm(self, x: Sym, filters, kernel, name, init="he_normal") -> Sym:
        """ConvLSTM over a length-1 sequence with zero initial state."""
        kh, kw = _pair(kernel) if self.ndim == 2 else (1, int(kernel))
        H, W, C = x.shape
        F = int(filters)
        ps = [ParamSpec(f"{name}/kernel", (kh, kw, C, 4 * F), init, True, (kh * kw * C, kh * kw * 4 * F)),
              ParamSpec(f"{name}/recurrent_kernel", (kh, kw, F, 4 * F), "orthogonal", True, (kh * kw * F, kh * kw * 4 * F)),
              ParamSpec(f"{name}/bias", (4 * F,), "unit_forget_bias", True)]
        return self._add("convlstm", name, [x], (H, W, F), dict(kernel=(kh, kw), filters=F), ps)

    def dense_ae(self, x: Sym, feature_number, out_channels, name) -> Sym:
        """Flatten -> Dense(feature_number, name='features') -> Dense(C*H*W) -> Reshape (autoencoder mode)."""
        H, W, C = x.shape
        nin = H * W * C
        nout = out_channels * H * W
        ps = [ParamSpec(f"{name}/features/kernel", (nin, feature_number), "glorot_uniform", True, (nin, feature_number)),
              ParamSpec(f"{name}/features/bias", (feature_number,), "zeros"),
              ParamSpec(f"{name}/expand/kernel", (feature_number, nout), "glorot_uniform", True, (feature_number, nout)),
              ParamSpec(f"{name}/expand/bias", (nout,), "zeros")]
        return self._add("dense_ae", name, [x], (H, W, out_channels), dict(features=feature_number), ps)

    # ------------------------------------------------------------------ queries
    def live_nodes(self) -> List[int]:
        """Indices reachable backwards from the outputs (Keras prunes everything else), ascending."""
        seen = set()
        stack = list(self.outputs)
        while stack:
            i = stack.pop()
            if i in seen:
                continue
            seen.add(i)
            stack.extend(self.nodes[i].inputs)
        return sorted(seen)

    def param_specs(self, live_only=True) -> List[ParamSpec]:
        idxs = self.live_nodes() if live_only else range(len(self.nodes))
        out = []
        for i in idxs:
            out.extend(self.nodes[i].params)
        return out

    def count_params(self, live_only=True) -> int:
        tot = 0
        for p in self.param_specs(live_only):
            n = 1
            for d in p.shape:
                n *= d
            tot += n
        return tot

    def consumers(self) -> Dict[int, List[int]]:
        live = set(self.live_nodes())
        cons: Dict[int, List[int]] = {i: [] for i in live}
        for i in sorted(live):
            for j in self.nodes[i].inputs:
                cons[j].append(i)
        return cons

    def macs(self, convlstm_gates: int = 4) -> int:
        """Algorithmic multiply-accumulates per sample: conv, convT (no zero insertion),
        ConvLSTM input conv with all four gates as written (convlstm_gates=3: without the forget gate, which is dead at
        T = 1 - the conservative figure SURVEY.md 8d prescribes for roofline fractions)."""
        tot = 0
        for i in self.live_nodes():
            n = self.nodes[i]
            if n.kind == "conv":
                kh, kw = n.attrs["kernel"]
                cin = self.nodes[n.inputs[0]].shape[2]
                tot += n.shape[0] * n.shape[1] * kh * kw * cin * n.shape[2]
            elif n.kind == "convT":
                kh, kw = n.attrs["kernel"]
                hi, wi, cin = self.nodes[n.inputs[0]].shape
                tot += hi * wi * kh * kw * cin * n.shape[2]
            elif n.kind == "convlstm":
                kh, kw = n.attrs["kernel"]
                cin = self.nodes[n.inputs[0]].shape[2]
                tot += n.shape[0] * n.shape[1] * kh * kw * cin * convlstm_gates * n.attrs["filters"]
        return tot

    def longest_conv_chain(self) -> int:
        """Convolution-type layers (conv, transposed conv, ConvLSTM, the Dense bottleneck) on the longest path from the input to an
        output: how many data-gradient GEMMs the gradient of the first layer has passed through (SegModel.build(backward="auto"))."""
        depth: Dict[int, int] = {}
        for i in self.live_nodes():
            n = self.nodes[i]
            depth[i] = max((depth[j] for j in n.inputs), default=0) + (1 if n.kind in ("conv", "convT", "convlstm", "dense_ae") else 0)
        return max((depth[o] for o in self.outputs), default=0)

    def plain_conv_bn_stack(self) -> bool:
        """True when the graph is a plain Conv_Block stack: no multiplicative gate (`mul`: attention gates) and every BatchNormalization
        directly behind a convolution - not behind an add, a concatenation, an activation or another normalisation, as in the tails
        of MultiResBlock / ResPath (unet_variants.py:85-122).  SegModel.build(backward="auto") halves an operand of the data gradients
        only for such graphs: the rounding of a data gradient is inherited by everything below it, and normalisation-on-normalisation
        and gating amplify it (measured, flags.py)."""
        for i in self.live_nodes():
            n = self.nodes[i]
            if n.kind == "mul":
                return False
            if n.kind == "bn" and self.nodes[n.inputs[0]].kind not in ("conv", "convT"):
                return False
        return True

    def conv_io_elements(self) -> Tuple[int, int]:
        """(activation elements read + written by the conv / convT / ConvLSTM layers per sample, weight elements):
        the algorithmic HBM traffic of a fully fused forward pass (the HBM-bound configurations)."""
        act = wts = 0
        for i in self.live_nodes():
            n = self.nodes[i]
            if n.kind in ("conv", "convT", "convlstm"):
                hi, wi, ci = self.nodes[n.inputs[0]].shape
                act += hi * wi * ci + n.shape[0] * n.shape[1] * n.shape[2]
                k = n.params[0].shape
                wts += k[0] * k[1] * k[2] * k[3]
        return act, wts

    def summary(self) -> str:
        lines = [f"Graph {self.name!r}: {len(self.live_nodes())} live nodes, {self.count_params():,} params"]
        for i in self.live_nodes():
            n = self.nodes[i]
            ins = ",".join(self.nodes[j].name for j in n.inputs)
            lines.append(f"  {n.name:32s} {n.kind:9s} {str(n.shape):20s} <- {ins}")
        return "\n".join(lines)
