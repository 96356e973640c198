"""Test helper: run the product's layer-graph IR on torch-CPU with the ORACLE's primitives.

Lives under tests/ on purpose - it is a checker for the host-side graph builder (wiring, names, shapes),
not a CPU fallback of the product."""
from typing import Dict, List

import torch

from oracle import keras_ref as K


class _R16(torch.autograd.Function):
    """round to fp16 and back, straight-through gradient: emulates the storage precision of the HIP pipeline."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.float16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def run_graph(g, x: torch.Tensor, weights: Dict[str, torch.Tensor], training=True, return_all=False, emulate_fp16=False):
    """emulate_fp16: round every materialised tensor (and the MFMA weight operands) to fp16 exactly where the HIP
    engine does - model input, conv / convT / ConvLSTM weights, every layer output except the fp32 heads - so that
    activation masks (ReLU, max-pool arg-max, hard-sigmoid knees) agree with the GPU run and kernel logic can be
    checked tightly, separately from the fp16-vs-fp32 precision question."""
    if g.ndim == 1 and x.dim() == 3:
        x = x.unsqueeze(1)
    R = _R16.apply if emulate_fp16 else (lambda t: t)
    outset = set(g.outputs)
    vals = {}
    for i in g.live_nodes():
        n = g.nodes[i]
        ins = [vals[j] for j in n.inputs]
        a = n.attrs
        if n.kind == "input":
            v = R(x)
        elif n.kind == "conv":
            b = weights.get(f"{n.name}/bias") if a["use_bias"] else None
            head = i in outset and a["kernel"] == (1, 1)          # fp32 head kernel: fp32 weights, fp32 output
            kern = weights[f"{n.name}/kernel"] if head else R(weights[f"{n.name}/kernel"])
            v = K.activation(K.conv2d_same(ins[0], kern, b, a["strides"]), a["activation"])
            if not head:
                v = R(v)
        elif n.kind == "convT":
            v = R(K.conv2d_transpose_same(ins[0], R(weights[f"{n.name}/kernel"]), weights[f"{n.name}/bias"], a["strides"]))
        elif n.kind == "bn":
            v, _, _ = K.batchnorm(ins[0], weights[f"{n.name}/gamma"], weights[f"{n.name}/beta"],
                                  weights[f"{n.name}/moving_mean"], weights[f"{n.name}/moving_variance"], training,
                                  fused_4d=(g.ndim == 2), eps=a["eps"], momentum=a["momentum"])
            v = R(v)
        elif n.kind == "act":
            v = R(K.activation(ins[0], a["fn"], a["alpha"]))
        elif n.kind == "maxpool":
            v = K.maxpool(ins[0], a["size"])
        elif n.kind == "upsample":
            v = R(K.upsample(ins[0], a["size"], a["interpolation"]))
        elif n.kind == "concat":
            v = torch.cat(ins, -1)
        elif n.kind == "add":
            v = R(sum(ins[1:], ins[0]))
        elif n.kind == "mul":
            v = R(ins[0] * ins[1])
        elif n.kind == "convlstm":
            z = R(K.conv2d_same(ins[0], R(weights[f"{n.name}/kernel"]), weights[f"{n.name}/bias"]))
            zi, zf, zc, zo = torch.chunk(z, 4, dim=-1)
            v = R(K.hard_sigmoid(zo) * torch.tanh(K.hard_sigmoid(zi) * torch.tanh(zc)))
        elif n.kind == "dense_ae":
            N, H, W, C = ins[0].shape
            z = ins[0].reshape(N, -1) @ weights[f"{n.name}/features/kernel"] + weights[f"{n.name}/features/bias"]
            v = (z @ weights[f"{n.name}/expand/kernel"] + weights[f"{n.name}/expand/bias"]).reshape(N, H, W, -1)
        else:
            raise NotImplementedError(n.kind)
        assert tuple(v.shape[1:]) == tuple(n.shape), (n.name, v.shape, n.shape)
        vals[i] = v
    outs = [vals[i] for i in g.outputs]
    if g.ndim == 1:
        outs = [o.squeeze(1) for o in outs]
    if return_all:
        return outs, {g.nodes[i].name: v for i, v in vals.items()}
    return outs
