"""Test helper: run the product's layer-graph IR on torch-CPU with the ORACLE's primitives.

Lives under tests/ on purpose - it is a checker for the host-side graph builder (wiring, names, shapes),
not a CPU fallback of the product."""
from typing import Dict, List

import torch

from oracle import keras_ref as K


def run_graph(g, x: torch.Tensor, weights: Dict[str, torch.Tensor], training=True, return_all=False):
    if g.ndim == 1 and x.dim() == 3:
        x = x.unsqueeze(1)
    vals = {}
    for i in g.live_nodes():
        n = g.nodes[i]
        ins = [vals[j] for j in n.inputs]
        a = n.attrs
        if n.kind == "input":
            v = x
        elif n.kind == "conv":
            b = weights.get(f"{n.name}/bias") if a["use_bias"] else None
            v = K.activation(K.conv2d_same(ins[0], weights[f"{n.name}/kernel"], b, a["strides"]), a["activation"])
        elif n.kind == "convT":
            v = K.conv2d_transpose_same(ins[0], weights[f"{n.name}/kernel"], weights[f"{n.name}/bias"], a["strides"])
        elif n.kind == "bn":
            v, _, _ = K.batchnorm(ins[0], weights[f"{n.name}/gamma"], weights[f"{n.name}/beta"],
                                  weights[f"{n.name}/moving_mean"], weights[f"{n.name}/moving_variance"], training,
                                  fused_4d=(g.ndim == 2), eps=a["eps"], momentum=a["momentum"])
        elif n.kind == "act":
            v = K.activation(ins[0], a["fn"], a["alpha"])
        elif n.kind == "maxpool":
            v = K.maxpool(ins[0], a["size"])
        elif n.kind == "upsample":
            v = K.upsample(ins[0], a["size"], a["interpolation"])
        elif n.kind == "concat":
            v = torch.cat(ins, -1)
        elif n.kind == "add":
            v = sum(ins[1:], ins[0])
        elif n.kind == "mul":
            v = ins[0] * ins[1]
        elif n.kind == "convlstm":
            v = K.convlstm_t1(ins[0], weights[f"{n.name}/kernel"], weights[f"{n.name}/bias"])
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
