"""Keras functional model  <->  structural parameter names of this engine (SURVEY.md section 8 rows c / f3).

The reference returns `tf.keras.Model` objects whose layers carry process-global auto-names (`conv2d_17`) and whose
`model.layers` / `get_weights()` order is graph-depth order, not creation order - neither can key a weight file.  This
module maps a Keras model onto the layer graph the builders of this package emit for the SAME constructor call by walking
both graphs in lock-step from the outputs through `inbound_nodes`, so that every weight-bearing Keras layer gets the
structural name of its counterpart (`enc3/conv`, `X12/bn`, `dec0/ag0/theta/conv`, ...):

    names = map_keras_model(keras_model, seg_model.graph)       # {keras layer name: structural node name}
    weights = keras_weights_by_structural_name(keras_model, seg_model.graph)
    seg_model.set_weights(weights)                                # a TF-trained model now runs on the HIP engine
    export_to_keras(seg_model.get_weights(), keras_model, seg_model.graph)      # and back

Nothing here imports TensorFlow: the walk only uses the duck-typed surface `layer.name`, `layer.__class__.__name__`,
`layer.get_weights()` / `set_weights()`, `layer.inbound_nodes` (or `_inbound_nodes`) with `node.inbound_layers`, and
`model.outputs[i]._keras_history.layer` (or `model.output_layers`).  tests/test_keras_map.py drives it with a hand-built
fake layer graph; tools/tf_capture_spec.py uses it on a real TF 2.12-2.15 installation to write golden fixtures.

Differences between the two graphs the walk absorbs:
  * pass-through Keras layers (Reshape around the ConvLSTM, Dropout, ZeroPadding with no padding, Lambda identity) are skipped
  * `Concat_Block` folds pairwise (`concatenate([concatenate([a, b]), c])`, unet_variants.py:27-32); the builders emit one
    n-ary concat: both sides are flattened to their leaf list before they are compared
  * the autoencoder bottleneck is Flatten -> Dense -> Dense -> Reshape in Keras and ONE `dense_ae` node here
  * `a * b` on Keras tensors (Attention_Block, unet_variants.py:81) is a `TFOpLambda` layer of `math.multiply` in TF 2.x, a `Multiply`
    operation in Keras 3, and the `mul` node here
  * a conv head with a fused activation is one Conv layer on both sides; `Activation('ReLU')` may be an `Activation`, a
    `ReLU` or a `LeakyReLU` layer depending on the Keras version
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from .graph import Graph, Node

_KIND_OF_CLASS = {
    "InputLayer": "input",
    "Conv1D": "conv", "Conv2D": "conv",
    "Conv1DTranspose": "convT", "Conv2DTranspose": "convT",
    "BatchNormalization": "bn",
    "Activation": "act", "ReLU": "act", "LeakyReLU": "act", "Softmax": "act",
    "MaxPooling1D": "maxpool", "MaxPooling2D": "maxpool", "MaxPool1D": "maxpool", "MaxPool2D": "maxpool",
    "UpSampling1D": "upsample", "UpSampling2D": "upsample",
    "Concatenate": "concat", "Add": "add", "Multiply": "mul",
    "ConvLSTM1D": "convlstm", "ConvLSTM2D": "convlstm",
    "Dense": "dense", "Flatten": "flatten",
}
_PASS_THROUGH = {"Reshape", "Dropout", "SpatialDropout2D", "SpatialDropout1D", "Lambda", "Identity"}


class KerasMapError(ValueError):
    pass


def _inbound_layers(layer) -> list:
    nodes = getattr(layer, "inbound_nodes", None)
    if nodes is None:
        nodes = getattr(layer, "_inbound_nodes", [])
    if not nodes:
        return []
    if len(nodes) > 1:
        raise KerasMapError(f"layer {layer.name!r} is called {len(nodes)} times (shared layers are not part of the hot path)")
    inb = nodes[0].inbound_layers
    return list(inb) if isinstance(inb, (list, tuple)) else [inb]


def _kind(layer) -> str:
    cls = layer.__class__.__name__
    if cls in _PASS_THROUGH:
        return "skip"
    if cls == "TFOpLambda":
        # `skip_connection * resampler` in Attention_Block (unet_variants.py:81) is not a Multiply layer: TF 2.x wraps the operator in
        # a TFOpLambda layer named 'tf.math.multiply' (Keras 3: an operation of class Multiply, which the table below knows)
        sym = str(getattr(layer, "symbol", "") or getattr(layer, "name", ""))
        if "multiply" in sym:
            return "mul"
        if sym.endswith("add") or "add_v2" in sym:
            return "add"
        raise KerasMapError(f"layer {layer.name!r}: TFOpLambda of {sym!r} is not part of the UNet-family hot path")
    k = _KIND_OF_CLASS.get(cls)
    if k is None:
        raise KerasMapError(f"layer {layer.name!r}: class {cls} is not part of the UNet-family hot path")
    return k


def _resolve(layer):
    """Skip pass-through layers (exactly one input each)."""
    while _kind(layer) == "skip":
        inb = _inbound_layers(layer)
        if len(inb) != 1:
            raise KerasMapError(f"pass-through layer {layer.name!r} has {len(inb)} inputs")
        layer = inb[0]
    return layer


def _k_inputs(layer) -> list:
    """Inputs of a Keras layer as the builders see them: pass-through layers skipped, nested concatenates flattened."""
    out = []
    for l in _inbound_layers(layer):
        l = _resolve(l)
        if _kind(layer) == "concat" and _kind(l) == "concat":
            out.extend(_k_inputs(l))
        else:
            out.append(l)
    return out


def _g_inputs(g: Graph, n: Node) -> List[Node]:
    out = []
    for j in n.inputs:
        m = g.nodes[j]
        if n.kind == "concat" and m.kind == "concat":
            out.extend(_g_inputs(g, m))
        else:
            out.append(m)
    return out


def _output_layers(model) -> list:
    if hasattr(model, "output_layers"):
        return list(model.output_layers)
    outs = model.outputs if isinstance(model.outputs, (list, tuple)) else [model.outputs]
    return [o._keras_history.layer if hasattr(o._keras_history, "layer") else o._keras_history[0] for o in outs]


def map_keras_model(model, g: Graph) -> Dict[str, str]:
    """{Keras layer name: structural node name} for every layer of `model` that has a counterpart in `g` (weight-bearing or
    not).  Raises KerasMapError when the two graphs do not have the same structure."""
    pairs: Dict[int, Tuple[object, Node]] = {}       # id(keras layer) -> (layer, node)
    seen_nodes: Dict[int, object] = {}

    def walk(layer, node: Node, path: str):
        layer = _resolve(layer)
        if node.kind == "dense_ae":                   # Reshape <- Dense <- Dense <- Flatten <- x   vs   one node
            d2 = layer
            if _kind(d2) != "dense":
                raise KerasMapError(f"{path}: expected the expanding Dense of the autoencoder bottleneck, got {d2.__class__.__name__}")
            d1 = _resolve(_inbound_layers(d2)[0])
            fl = _resolve(_inbound_layers(d1)[0])
            if _kind(d1) != "dense" or _kind(fl) != "flatten":
                raise KerasMapError(f"{path}: expected Flatten -> Dense -> Dense, got {fl.__class__.__name__} -> {d1.__class__.__name__}")
            pairs[id(d1)] = (d1, Node(-1, "dense", node.name + "/features", [], (), {}, node.params[0:2]))
            pairs[id(d2)] = (d2, Node(-1, "dense", node.name + "/expand", [], (), {}, node.params[2:4]))
            walk(_inbound_layers(fl)[0], g.nodes[node.inputs[0]], path + "/ae")
            return
        k = _kind(layer)
        if k != node.kind:
            raise KerasMapError(f"{path}: Keras layer {layer.name!r} ({layer.__class__.__name__}) where the builder has {node.kind} {node.name!r}")
        if id(layer) in pairs:
            if pairs[id(layer)][1] is not node:
                raise KerasMapError(f"{path}: Keras layer {layer.name!r} reached as {node.name!r} and as {pairs[id(layer)][1].name!r}")
            return
        if node.idx in seen_nodes and seen_nodes[node.idx] is not layer:
            raise KerasMapError(f"{path}: builder node {node.name!r} reached from two Keras layers")
        pairs[id(layer)] = (layer, node)
        seen_nodes[node.idx] = layer
        ki, gi = _k_inputs(layer), _g_inputs(g, node)
        if len(ki) != len(gi):
            raise KerasMapError(f"{path}: {layer.name!r} has {len(ki)} inputs, {node.name!r} has {len(gi)}")
        for a, b in zip(ki, gi):
            walk(a, b, path + "/" + b.name)

    outs = _output_layers(model)
    if len(outs) != len(g.outputs):
        raise KerasMapError(f"the Keras model has {len(outs)} outputs, the builder graph {len(g.outputs)}")
    for layer, oi in zip(outs, g.outputs):
        walk(layer, g.nodes[oi], g.nodes[oi].name)
    return {layer.name: node.name for layer, node in pairs.values()}


# Keras `layer.get_weights()` order per layer class -> parameter slots of the graph node (graph.py)
_SLOTS = {"conv": ("kernel", "bias"), "convT": ("kernel", "bias"), "bn": ("gamma", "beta", "moving_mean", "moving_variance"),
          "convlstm": ("kernel", "recurrent_kernel", "bias"), "dense": ("kernel", "bias")}


def _spec_names(g: Graph, by_name: Dict[str, Node], node_name: str, kind: str) -> List[str]:
    if node_name.endswith("/features") or node_name.endswith("/expand"):          # halves of a dense_ae node
        return [f"{node_name}/kernel", f"{node_name}/bias"]
    n = by_name[node_name]
    return [p.name for p in n.params]


def keras_weights_by_structural_name(model, g: Graph) -> Dict[str, np.ndarray]:
    """Every weight of the Keras model under the structural name `SegModel.set_weights` takes (Keras layouts are the
    storage convention of graph.py, so no transposition happens here).  A 1D model's kernels (k, Cin, Cout) get the
    leading unit axis of the H = 1 convention."""
    m = map_keras_model(model, g)
    by_name = {n.name: n for n in g.nodes}
    layers = {l.name: l for l in model.layers}
    out: Dict[str, np.ndarray] = {}
    for lname, nname in m.items():
        layer = layers[lname]
        ws = layer.get_weights()
        if not ws:
            continue
        kind = _kind(layer)
        names = _spec_names(g, by_name, nname, kind)
        if len(names) != len(ws):
            raise KerasMapError(f"{lname} -> {nname}: {len(ws)} Keras weights for {len(names)} parameters {names}")
        specs = {p.name: p for n in g.nodes for p in n.params}
        for name, w in zip(names, ws):
            w = np.asarray(w, np.float32)
            want = tuple(specs[name].shape)
            if w.shape != want and g.ndim == 1 and (1,) + tuple(w.shape) == want:
                w = w[None]                                   # Conv1D kernel (k, Cin, Cout) -> (1, k, Cin, Cout)
            if tuple(w.shape) != want:
                raise KerasMapError(f"{lname} -> {name}: shape {w.shape} != {want}")
            out[name] = w
    return out


def export_to_keras(weights: Dict[str, np.ndarray], model, g: Graph):
    """The other direction: write engine weights (structural names) into the matching Keras layers."""
    m = map_keras_model(model, g)
    by_name = {n.name: n for n in g.nodes}
    layers = {l.name: l for l in model.layers}
    for lname, nname in m.items():
        layer = layers[lname]
        cur = layer.get_weights()
        if not cur:
            continue
        names = _spec_names(g, by_name, nname, _kind(layer))
        new = []
        for name, c in zip(names, cur):
            w = np.asarray(weights[name], np.float32)
            new.append(w.reshape(np.asarray(c).shape))
        layer.set_weights(new)
