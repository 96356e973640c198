"""`load_model(path)`: a Keras checkpoint of the reference -> a SegModel, from the FILE alone.

The reference's test script does not call the builder again: it rebuilds the network from the checkpoint,
`tf.keras.saving.load_model('..._Fold_1.keras')` (`2D/Test.py:114`), and predicts.  The equivalent here reads the functional
model's JSON configuration (the `.keras` archive's `config.json`, a legacy `.h5` file's `model_config`), turns every layer into the
node the builders of this package would have emitted for it, and takes the weights over - no constructor arguments, no TensorFlow:

    model = segamd.load_model("Results/UNet/Fold_1/UNet_256_Fold_1.keras")     # == tf.keras.saving.load_model(...)
    pred = model.predict(images)                                                # builds itself for the batch on first use

Layers understood - the ones the UNet-family builders of the reference emit (`2D/models/unet_variants.py`, `1D/Models/
unet_variants.py`, `BCDUNet.py`): InputLayer; Conv1D/2D and Conv1D/2DTranspose with padding='same', no dilation; BatchNormalization
over the channel axis; Activation / ReLU / LeakyReLU / Softmax; MaxPooling (stride = pool size); UpSampling (nearest / bilinear);
Concatenate over channels (nested concatenates are flattened - `Concat_Block` folds pairwise); Add; Multiply - also as the operator
`a * b`, TF 2.x's TFOpLambda of math.multiply (Attention_Block); ConvLSTM1D/2D on a
length-1 sequence (the Reshape in front of it is absorbed); Flatten -> Dense -> Dense -> Reshape as the autoencoder bottleneck;
Dropout-family layers of rate 0 and shape-preserving Reshapes as identity (a dropout that drops is refused: the engine has none).
Anything else - 'valid' padding, dilation, a Dense that is not part of that bottleneck, Lambda, shared layers, nested models - raises
KerasImportError naming the layer: a model the engine would compute differently is refused, not approximated.

Node names are the Keras layer names (`conv2d_17/kernel`), so `get_weights()` of an imported model is keyed the way the file is.
The configuration keys are restated from the Keras layer documentation (`get_config()`); like keras_file.py this has met no file
written by Keras itself yet (`tools/tf_capture_spec.py` saves them on a TensorFlow machine)."""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Tuple

import numpy as np

from .graph import Graph, Sym
from .keras_file import KerasFileError, KerasFileModel, _refs, _snake


class KerasImportError(ValueError):
    pass


_ACTS = {"relu": "relu", "sigmoid": "sigmoid", "softmax": "softmax", "linear": None, "leakyrelu": "leaky_relu", "leaky_relu": "leaky_relu"}


def _activation(v, layer) -> Tuple[Optional[str], float]:
    """(activation of this package or None for linear, slope of a leaky ReLU) of a serialised Keras activation.  The reference writes
    `Activation('ReLU')` / `Activation('LeakyReLU')` (unet_variants.py:7, 17): Keras 2 resolves those names to LAYER objects and
    serialises them as `{"class_name": "LeakyReLU", "config": {"alpha": 0.3, ...}}`; plain function names (`'relu'`, `'sigmoid'`)
    stay strings.  The layer `LeakyReLU` defaults to a slope of 0.3, Keras 3's FUNCTION `leaky_relu` to 0.2."""
    alpha = 0.0
    if isinstance(v, dict):
        cfg = v.get("config") or {}
        cls = v.get("class_name")
        if cls in ("ReLU", "LeakyReLU", "Softmax"):
            if cls == "ReLU":
                if cfg.get("max_value") is not None or float(cfg.get("threshold", 0.0) or 0.0) != 0.0:
                    raise KerasImportError(f"layer {layer!r}: ReLU with max_value / threshold")
                slope = float(cfg.get("negative_slope", 0.0) or 0.0)
                return ("relu", 0.0) if slope == 0.0 else ("leaky_relu", slope)
            if cls == "LeakyReLU":
                return "leaky_relu", float(cfg.get("negative_slope", cfg.get("alpha", 0.3)))
            return "softmax", 0.0
        v = cls if cls not in (None, "function") else cfg.get("name", cfg) if isinstance(cfg, dict) else cfg
    if v is None:
        return None, 0.0
    key = str(v).lower()
    if key not in _ACTS:
        raise KerasImportError(f"layer {layer!r}: activation {v!r} is not one the hot path has (relu, leaky_relu, sigmoid, softmax, linear)")
    if _ACTS[key] == "leaky_relu":
        alpha = 0.3 if key == "leakyrelu" else 0.2
    return _ACTS[key], alpha


def _act_name(v, layer):
    fn, _alpha = _activation(v, layer)
    if fn == "leaky_relu":
        raise KerasImportError(f"layer {layer!r}: a leaky ReLU as the fused activation of a convolution (the reference uses separate layers)")
    return fn


def _ints(v, n) -> Tuple[int, ...]:
    if isinstance(v, (list, tuple)):
        if len(v) != n:
            raise KerasImportError(f"expected {n} values, got {v}")
        return tuple(int(x) for x in v)
    return (int(v),) * n


def graph_from_keras_config(cfg: dict) -> Graph:
    """The builder graph of a serialised functional Keras model (module docstring).  Whatever is wrong with the configuration - a layer
    the engine would compute differently, shapes that do not fit, a missing key - comes out as KerasImportError."""
    try:
        return _graph_from_keras_config(cfg)
    except KerasImportError:
        raise
    except (KeyError, TypeError, ValueError, IndexError, AttributeError, ArithmeticError) as e:
        raise KerasImportError(f"the model configuration does not describe a UNet-family model of the reference ({type(e).__name__}: {str(e)[:200]})") from None


def _graph_from_keras_config(cfg: dict) -> Graph:
    if cfg.get("class_name") not in ("Functional", "Model"):
        raise KerasImportError(f"a {cfg.get('class_name')} model (the reference builds functional models)")
    body = cfg["config"]
    layers = {(lc.get("name") or lc["config"]["name"]): lc for lc in body["layers"]}
    ins_of: Dict[str, List[str]] = {}
    for name, lc in layers.items():
        nodes = lc.get("inbound_nodes", [])
        if len(nodes) > 1:
            raise KerasImportError(f"layer {name!r} is called {len(nodes)} times (shared layers are not part of the hot path)")
        ins_of[name] = _refs(nodes[0]) if nodes else []
    inputs = [n for n, lc in layers.items() if lc["class_name"] == "InputLayer"]
    if len(inputs) != 1:
        raise KerasImportError(f"{len(inputs)} input layers (the reference's models have one)")
    icfg = layers[inputs[0]]["config"]
    ishape = icfg.get("batch_input_shape") or icfg.get("batch_shape") or ([None] + list(icfg.get("shape") or []))
    spatial = [d for d in ishape[1:]]
    if any(d is None for d in spatial) or len(spatial) not in (2, 3):
        raise KerasImportError(f"input shape {ishape}: a fixed (H, W, C) or (L, C) is needed")
    ndim = len(spatial) - 1
    g = Graph(ndim=ndim, name=body.get("name", "model"))
    sym: Dict[str, Sym] = {}
    flat_shape: Dict[str, Tuple[int, ...]] = {}           # Keras-side shapes of layers that have no node of their own (Flatten, Dense)
    seq: set = set()                                      # layers whose Keras tensor carries the length-1 time axis of a ConvLSTM input

    def need(cond, layer, what):
        if not cond:
            raise KerasImportError(f"layer {layer!r} ({layers[layer]['class_name']}): {what}")

    def same_padding(c, name):
        # (a 1 x 1 kernel needs no padding: 'valid' - the default the reference leaves on its attention-gate convs and heads,
        # unet_variants.py:69-75 - gives ceil(extent / stride) outputs like 'same')
        pointwise = all(k == 1 for k in _ints(c.get("kernel_size", 0), ndim))
        need(str(c.get("padding", "valid")).lower() == "same" or pointwise, name, f"padding={c.get('padding')!r} with a kernel of {c.get('kernel_size')} (the hot path is padding='same')")
        need(all(d == 1 for d in _ints(c.get("dilation_rate", 1), ndim)), name, f"dilation_rate={c.get('dilation_rate')}")
        need(c.get("data_format") in (None, "channels_last"), name, f"data_format={c.get('data_format')!r}")
        need(int(c.get("groups", 1) or 1) == 1, name, f"groups={c.get('groups')}")

    def build(name: str) -> Sym:
        """The node for layer `name`; its inputs are there already (topological order below)."""
        lc = layers[name]
        cls, c = lc["class_name"], lc["config"]
        srcs = ins_of[name]
        xs = [sym[s] for s in srcs]
        k1 = (lambda v: _ints(v, ndim) if ndim == 2 else _ints(v, 1)[0])
        out: Optional[Sym]
        extra = [k for k, v in c.items() if k.endswith(("_regularizer", "_constraint")) and v is not None]
        need(not extra, name, f"{', '.join(extra)} set (the engine's loss and update have neither regularisers nor constraints)")
        if cls.startswith(("Conv", "BatchNormalization", "Dense")):
            # (train_mode='pretrained_encoder' freezes the encoder, unet_variants.py:1058: the engine trains every parameter it holds)
            need(c.get("trainable", True) is not False, name, "trainable=False (a frozen layer: the engine would train it)")
        if cls == "InputLayer":
            out = g.input(*(spatial if ndim == 2 else [1] + spatial), name=name)
        elif cls in ("Conv1D", "Conv2D"):
            need(cls == f"Conv{ndim}D", name, f"in a {ndim}-D model")
            same_padding(c, name)
            fused = _act_name(c.get("activation"), name)
            # (the engine fuses an activation into a conv only where the reference does - the output heads, unet_variants.py:300-302;
            # anywhere else SegModel.build would refuse it later, on the GPU and without the layer's name)
            need(fused is None or name in out_names, name, f"activation={c.get('activation')!r} fused into a convolution that is not an output "
                 "(the reference's blocks use a separate Activation layer)")
            out = g.conv(xs[0], int(c["filters"]), k1(c["kernel_size"]), name, strides=k1(c.get("strides", 1)),
                         use_bias=bool(c.get("use_bias", True)), activation=fused)
        elif cls in ("Conv1DTranspose", "Conv2DTranspose"):
            same_padding(c, name)
            need(c.get("output_padding") is None, name, f"output_padding={c.get('output_padding')}")
            need(bool(c.get("use_bias", True)), name, "use_bias=False")
            need(_act_name(c.get("activation"), name) is None, name, f"activation={c.get('activation')!r} (the reference's transposed convs are linear)")
            out = g.conv_transpose(xs[0], int(c["filters"]), k1(c["kernel_size"]), k1(c.get("strides", 1)), name)
        elif cls == "BatchNormalization":
            axis = c.get("axis", -1)
            axis = axis[0] if isinstance(axis, (list, tuple)) else axis
            need(axis in (-1, ndim + 1), name, f"axis={c.get('axis')} (channels last)")
            need(bool(c.get("center", True)) and bool(c.get("scale", True)), name, "center / scale switched off")
            out = g.bn(xs[0], name, eps=float(c.get("epsilon", 1e-3)), momentum=float(c.get("momentum", 0.99)))
        elif cls == "Activation":
            a, slope = _activation(c.get("activation"), name)
            out = xs[0] if a is None else g.act(xs[0], a, name, alpha=slope)
        elif cls == "ReLU":
            need(c.get("max_value") is None and float(c.get("threshold", 0.0) or 0.0) == 0.0, name, "max_value / threshold")
            slope = float(c.get("negative_slope", 0.0) or 0.0)
            out = g.act(xs[0], "relu", name) if slope == 0.0 else g.act(xs[0], "leaky_relu", name, alpha=slope)
        elif cls == "LeakyReLU":
            slope = c.get("negative_slope", c.get("alpha", 0.3))
            out = g.act(xs[0], "leaky_relu", name, alpha=float(slope))
        elif cls == "Softmax":
            need(c.get("axis", -1) in (-1, ndim + 1), name, f"axis={c.get('axis')}")
            out = g.act(xs[0], "softmax", name)
        elif cls in ("MaxPooling1D", "MaxPooling2D", "MaxPool1D", "MaxPool2D"):
            size = k1(c.get("pool_size", 2))
            need(c.get("strides") is None or k1(c["strides"]) == size, name, f"strides={c.get('strides')} != pool_size={c.get('pool_size')}")
            H, W, _C = xs[0].shape
            ph, pw = (size if ndim == 2 else (1, size))
            need(str(c.get("padding", "valid")).lower() == "valid" or (H % ph == 0 and W % pw == 0), name, "padding='same' on an extent the pool does not divide")
            out = g.maxpool(xs[0], size, name)
        elif cls in ("UpSampling1D", "UpSampling2D"):
            interp = str(c.get("interpolation", "nearest")).lower()
            need(interp in ("nearest", "bilinear"), name, f"interpolation={interp!r}")
            out = g.upsample(xs[0], k1(c.get("size", 2)), name, interpolation=interp)
        elif cls == "Concatenate":
            need(c.get("axis", -1) in (-1, ndim + 1, ndim + 2), name, f"axis={c.get('axis')} (channels)")
            # (the reference reshapes the skip connection and the decoder tensor to (1, H, W, C) EACH and concatenates the two
            # sequences in front of the ConvLSTM, unet_variants.py:145-147: the time axis rides through the concatenation)
            in_seq = [s_ in seq for s_ in srcs]
            need(all(in_seq) or not any(in_seq), name, "a concatenation of sequences with plain tensors")
            if all(in_seq):
                seq.add(name)
                need(c.get("axis", -1) in (-1, ndim + 2), name, f"axis={c.get('axis')} (channels of the sequence)")
            parts: List[Sym] = []
            for s, x in zip(srcs, xs):                   # concatenate([concatenate([a, b]), c]) == concatenate([a, b, c])
                parts += [Sym(g, j) for j in x.node.inputs] if x.node.kind == "concat" else [x]
            out = g.concat(parts, name)
        elif cls == "Add":
            out = g.add(xs, name)
        elif cls == "Multiply" or (cls == "TFOpLambda" and "multiply" in str(c.get("function", name))):
            need(len(xs) == 2, name, f"{len(xs)} factors")
            a, b = xs
            if a.shape[2] == 1 and b.shape[2] != 1:      # (the one-channel gate is the second operand of Graph.mul)
                a, b = b, a
            out = g.mul(a, b, name)
        elif cls in ("ConvLSTM1D", "ConvLSTM2D"):
            same_padding(c, name)
            need(not c.get("return_sequences", False) and not c.get("stateful", False), name, "return_sequences / stateful")
            need(all(s == 1 for s in _ints(c.get("strides", 1), ndim)), name, f"strides={c.get('strides')}")
            # (go_backwards=True, as the reference passes it, is the identity on a length-1 sequence; the gates are Keras 2's
            # defaults - tanh and HARD sigmoid, what the kernels compute: a file that says 'sigmoid' (Keras 3's default) is refused)
            need(str(c.get("activation", "tanh")) == "tanh" and str(c.get("recurrent_activation", "hard_sigmoid")) == "hard_sigmoid",
                 name, f"activation={c.get('activation')!r} / recurrent_activation={c.get('recurrent_activation')!r} (the engine's gates are tanh / hard_sigmoid)")
            need(srcs[0] in seq, name, "its input is not a length-1 sequence (Reshape((1, ...)) in front of it)")
            out = g.convlstm(xs[0], int(c["filters"]), k1(c["kernel_size"]), name)
        elif cls in ("Dropout", "SpatialDropout1D", "SpatialDropout2D", "Identity"):
            # (the UNet-family builders of the reference have no dropout; a checkpoint that has would TRAIN differently here - the engine
            # has none - while predicting the same: refused rather than imported with different training semantics)
            need(cls == "Identity" or float(c.get("rate", 0.0) or 0.0) == 0.0, name, f"rate={c.get('rate')} (the engine has no dropout: the model would train differently)")
            out = xs[0]
        elif cls == "Flatten":
            H, W, C = xs[0].shape
            flat_shape[name] = (H * W * C,)
            out = xs[0]
        elif cls == "Dense":
            need(srcs[0] in flat_shape and len(flat_shape[srcs[0]]) == 1, name, "a Dense layer outside the Flatten -> Dense -> Dense -> Reshape bottleneck")
            need(_act_name(c.get("activation"), name) is None and bool(c.get("use_bias", True)), name, "activation / use_bias (the bottleneck's Dense layers are linear with bias)")
            flat_shape[name] = (int(c["units"]),)
            out = xs[0]
        elif cls == "Reshape":
            target = tuple(int(d) for d in c["target_shape"])
            src_cls = layers[srcs[0]]["class_name"]
            if src_cls == "Dense":                       # ... -> Reshape closes the bottleneck
                d2, d1 = srcs[0], ins_of[srcs[0]][0]
                need(layers[d1]["class_name"] == "Dense" and layers[ins_of[d1][0]]["class_name"] == "Flatten", name,
                     "Reshape of a Dense output that is not Flatten -> Dense -> Dense -> Reshape")
                H, W, _C = xs[0].shape
                full = target if ndim == 2 else (1,) + target
                need(len(full) == 3 and full[:2] == (H, W) and flat_shape[d2][0] == full[0] * full[1] * full[2], name,
                     f"target_shape={target} against the flattened tensor {xs[0].shape}")
                out = g.dense_ae(xs[0], flat_shape[d1][0], full[2], name)
                out.node.attrs["keras_dense"] = (d1, d2)
            else:
                shp = xs[0].shape if ndim == 2 else xs[0].shape[1:]
                need(target in (tuple(shp), (1,) + tuple(shp)), name, f"target_shape={target} of a tensor {tuple(shp)} (only the length-1 sequence axis in front of a ConvLSTM)")
                if len(target) == len(shp) + 1:
                    seq.add(name)
                out = xs[0]
        else:
            raise KerasImportError(f"layer {name!r}: class {cls} is not part of the UNet-family hot path")
        if name not in seq and any(s_ in seq for s_ in srcs) and cls not in ("ConvLSTM1D", "ConvLSTM2D"):
            raise KerasImportError(f"layer {name!r} ({cls}): applied to the length-1 sequence of a ConvLSTM input (only Concatenate and the ConvLSTM itself are)")
        sym[name] = out
        return out

    try:
        out_names = _refs(body["output_layers"])
        if not out_names:
            raise KerasImportError("the model configuration names no output layer")
        order, state, stack = [], {}, [(n, 0) for n in reversed(out_names)]          # iterative depth-first post-order from the outputs
        while stack:
            n, i = stack.pop()
            if state.get(n) == 2:
                continue
            if i < len(ins_of[n]):
                stack.append((n, i + 1))
                m = ins_of[n][i]
                if state.get(m) == 1:
                    raise KerasImportError(f"the layer graph has a cycle through {m!r}")
                if state.get(m) is None:
                    state[m] = 1
                    stack.append((m, 0))
                state[n] = 1
            else:
                state[n] = 2
                order.append(n)
        for n in order:
            build(n)
        # Flatten and Dense have no node of their own: they exist only as links of Flatten -> Dense -> Dense -> Reshape, which the
        # closing Reshape turns into ONE bottleneck node.  A link that is a model output, or that anything but the next link reads,
        # would import as the tensor in front of the Flatten with its weights dropped - refused, not approximated.
        for n in flat_shape:
            readers = [m for m in order if n in ins_of[m]]
            nxt = ("Dense",) if layers[n]["class_name"] == "Flatten" or layers[ins_of[n][0]]["class_name"] == "Flatten" else ("Reshape",)
            need(n not in out_names, n, "is a model output (only the Flatten -> Dense -> Dense -> Reshape bottleneck is part of the hot path)")
            need(len(readers) == 1 and layers[readers[0]]["class_name"] in nxt, n,
                 f"is read by {readers or 'nothing'} - not the next link of a Flatten -> Dense -> Dense -> Reshape bottleneck")
    except KeyError as e:
        raise KerasImportError(f"the configuration names a layer or key that is not there: {e}") from None
    g.set_outputs([sym[n] for n in out_names], out_names)
    return g


_SLOTS = {"conv": 2, "convT": 2, "bn": 4, "convlstm": 3}


def weights_of_imported_graph(kfile: KerasFileModel, g: Graph) -> Dict[str, np.ndarray]:
    """{parameter name: array}: node names ARE the Keras layer names here, so every weight-bearing node takes its layer's
    `get_weights()` in Keras' order (kernel, bias / gamma, beta, moving_mean, moving_variance / kernel, recurrent_kernel, bias);
    the bottleneck node takes its two Dense layers'."""
    by_name = {l.name: l for l in kfile.layers}
    out = {}
    for i in g.live_nodes():
        n = g.nodes[i]
        if not n.params:
            continue
        if n.kind == "dense_ae":
            d1, d2 = n.attrs["keras_dense"]
            ws = by_name[d1].get_weights() + by_name[d2].get_weights()
        else:
            ws = by_name[n.name].get_weights()
        if len(ws) != len(n.params):
            raise KerasImportError(f"layer {n.name!r}: {len(ws)} arrays in the file for {[p.name for p in n.params]}")
        for p, w in zip(n.params, ws):
            w = np.asarray(w, np.float32)
            want = tuple(p.shape)
            if w.shape != want and g.ndim == 1 and (1,) + tuple(w.shape) == want:
                w = w[None]                               # Conv1D kernel (k, Cin, Cout) -> the H = 1 convention
            if tuple(w.shape) != want:
                raise KerasImportError(f"{p.name}: the file holds {w.shape}, the layer's configuration implies {want}")
            out[p.name] = w
    return out


def load_model(path: str, config: Optional[str] = None, batch_size: Optional[int] = None, dtype: str = "f32"):
    """tf.keras.saving.load_model (2D/Test.py:114): the SegModel of a Keras checkpoint file, weights taken over when it is built
    (`batch_size=` builds it right away; otherwise `predict` / `build` do)."""
    from .model import SegModel
    kfile = KerasFileModel(path, config)
    g = graph_from_keras_config(kfile.config)
    model = SegModel(g, dtype=dtype)
    model._weights_at_build = weights_of_imported_graph(kfile, g)
    if batch_size is not None:
        model.build(batch_size, dtype=dtype)
    return model


# ------------------------------------------------------------------------------------------------------------------ the way out
def _lname(s: str) -> str:
    return s.replace("/", ".")                            # (Keras 3 refuses '/' in layer names)


def config_from_graph(g: Graph) -> Tuple[dict, List[Tuple[str, str, List[str]]]]:
    """(functional-model configuration, [(Keras layer name, sub-object holding its variables or "", [parameter names in Keras' order])])
    for a layer graph of this package: the inverse of `graph_from_keras_config`, in Keras 2's spelling (`batch_input_shape`, nested-list
    `inbound_nodes`, `alpha`).  A ConvLSTM gets its Reshape((1, ...)) in front, the bottleneck its Flatten -> Dense -> Dense -> Reshape."""
    one_d = g.ndim == 1
    D = "1D" if one_d else "2D"
    sp = (lambda hw: [int(hw[1])] if one_d else [int(hw[0]), int(hw[1])])
    layers, holders, lname = [], [], {}

    def emit(cls, name, cfg, inbound, params=(), sub=""):
        layers.append({"module": "keras.layers", "class_name": cls, "config": dict(cfg, name=name, trainable=True, dtype="float32"),
                       "registered_name": None, "name": name, "inbound_nodes": [[[n, 0, 0, {}] for n in inbound]] if inbound else []})
        if params:
            holders.append((name, sub, [p.name for p in params]))
        return name

    for i in g.live_nodes():
        n = g.nodes[i]
        a, nm = n.attrs, _lname(n.name)
        ins = [lname[j] for j in n.inputs]
        H, W, C = n.shape
        if n.kind == "input":
            lname[i] = emit("InputLayer", nm, dict(batch_input_shape=[None] + sp((H, W)) + [C], sparse=False, ragged=False), [])
        elif n.kind in ("conv", "convT"):
            cfg = dict(filters=C, kernel_size=sp(a["kernel"]), strides=sp(a["strides"]), padding="same", data_format="channels_last",
                       dilation_rate=sp((1, 1)), groups=1, activation=a.get("activation") or "linear", use_bias=bool(a.get("use_bias", True)))
            if n.kind == "convT":
                cfg["output_padding"] = None
            lname[i] = emit(f"Conv{D}" + ("Transpose" if n.kind == "convT" else ""), nm, cfg, ins, n.params)
        elif n.kind == "bn":
            lname[i] = emit("BatchNormalization", nm, dict(axis=[2 if one_d else 3], momentum=float(a["momentum"]), epsilon=float(a["eps"]), center=True, scale=True),
                            ins, n.params)
        elif n.kind == "act":
            lname[i] = (emit("LeakyReLU", nm, dict(alpha=float(a["alpha"])), ins) if a["fn"] == "leaky_relu"
                        else emit("Activation", nm, dict(activation=a["fn"]), ins))
        elif n.kind == "maxpool":
            lname[i] = emit(f"MaxPooling{D}", nm, dict(pool_size=sp(a["size"]), strides=sp(a["size"]), padding="valid", data_format="channels_last"), ins)
        elif n.kind == "upsample":
            cfg = dict(size=sp(a["size"])[0]) if one_d else dict(size=sp(a["size"]), data_format="channels_last", interpolation=a["interpolation"])
            lname[i] = emit(f"UpSampling{D}", nm, cfg, ins)
        elif n.kind == "concat":
            lname[i] = emit("Concatenate", nm, dict(axis=-1), ins)
        elif n.kind == "add":
            lname[i] = emit("Add", nm, {}, ins)
        elif n.kind == "mul":
            lname[i] = emit("Multiply", nm, {}, ins)
        elif n.kind == "convlstm":
            Hi, Wi, Ci = g.nodes[n.inputs[0]].shape
            r = emit("Reshape", nm + ".sequence", dict(target_shape=[1] + sp((Hi, Wi)) + [Ci]), ins)
            lname[i] = emit(f"ConvLSTM{D}", nm, dict(filters=C, kernel_size=sp(a["kernel"]), strides=sp((1, 1)), padding="same", data_format="channels_last",
                                                      dilation_rate=sp((1, 1)), activation="tanh", recurrent_activation="hard_sigmoid", use_bias=True,
                                                      unit_forget_bias=True, return_sequences=False, go_backwards=True, stateful=False), [r], n.params, "cell")
        elif n.kind == "dense_ae":
            fl = emit("Flatten", nm + ".flatten", dict(data_format="channels_last"), ins)
            d1 = emit("Dense", nm + ".features", dict(units=int(a["features"]), activation="linear", use_bias=True), [fl], n.params[0:2])
            d2 = emit("Dense", nm + ".expand", dict(units=H * W * C, activation="linear", use_bias=True), [d1], n.params[2:4])
            lname[i] = emit("Reshape", nm, dict(target_shape=sp((H, W)) + [C]), [d2])
        else:
            raise KerasImportError(f"node {n.name!r}: kind {n.kind} has no Keras layer")
    cfg = {"module": "keras.src.engine.functional", "class_name": "Functional",
           "config": {"name": g.name, "trainable": True, "layers": layers,
                      "input_layers": [[lname[i], 0, 0] for i in g.inputs], "output_layers": [[lname[o], 0, 0] for o in g.outputs]},
           "registered_name": "Functional", "build_config": {"input_shape": [None] + sp(g.nodes[g.inputs[0]].shape[:2]) + [g.nodes[g.inputs[0]].shape[2]]}}
    return cfg, holders


def save_model(graph: Graph, weights: Dict[str, np.ndarray], path: str):
    """model.save('x.keras') / ModelCheckpoint (2D/Train.py:375): a `.keras`-layout archive - config.json in Keras 2's spelling,
    metadata.json, model.weights.h5 with `layers/<class name + counter>/vars/<i>` as Keras addresses them (a ConvLSTM's under `cell/`) - of a layer graph and its weights
    (structural names).  `segamd.load_model` / `SegModel.load_weights` read it back; laid out as Keras documents the format, not yet
    opened by a Keras installation (none here)."""
    import json
    import zipfile
    from .h5lite import H5Writer
    cfg, holders = config_from_graph(graph)
    w = H5Writer()
    w.create_group("vars")
    key_of, used = {}, {}
    for lc in cfg["config"]["layers"]:                   # Keras' own addressing: snake-cased class name + counter in model.layers order
        base = _snake(lc["class_name"])
        used[base] = used.get(base, -1) + 1
        key_of[lc["name"]] = base if used[base] == 0 else f"{base}_{used[base]}"
        w.create_group(f"layers/{key_of[lc['name']]}/vars")
    for lname, sub, pnames in holders:
        base = f"layers/{key_of[lname]}/{sub + '/' if sub else ''}vars"
        w.create_group(base)
        for i, pn in enumerate(pnames):
            if pn not in weights:
                raise KerasImportError(f"no value for parameter {pn!r}")
            a = np.asarray(weights[pn], np.float32)
            if graph.ndim == 1 and a.ndim == 4 and a.shape[0] == 1:
                a = a[0]                                  # Conv1D / ConvLSTM1D kernels are (k, Cin, Cout)
            w.create_dataset(f"{base}/{i}", a)
    tmp = f"{path}.tmp{os.getpid()}"                     # (ModelCheckpoint rewrites the best file in place: a kill mid-write must not cost the previous one)
    try:
        with zipfile.ZipFile(tmp, "w", zipfile.ZIP_STORED) as z:
            z.writestr("metadata.json", json.dumps({"keras_version": "2.15.0", "date_saved": "", "written_by": "segamd.keras_import.save_model"}))
            z.writestr("config.json", json.dumps(cfg))
            z.writestr("model.weights.h5", w.tobytes())
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def describe(path: str, config: Optional[str] = None) -> str:
    """What a checkpoint file is and whether this engine takes it: format, layer classes, parameters, input / outputs - or the layer
    that is in the way.  `python tools/keras_info.py x.keras` prints it (no GPU needed)."""
    from collections import Counter
    kfile = KerasFileModel(path, config)
    classes = Counter(l.__class__.__name__ for l in kfile.layers)
    held = sum(int(np.prod(w.shape)) for l in kfile.layers for w in l._w)
    lines = [f"{path}: {'.keras archive' if kfile.kind == 'keras' else 'HDF5 file'}, model {kfile.config['config'].get('name')!r}, {len(kfile.layers)} layers, "
             f"{held:,} weight values in {sum(len(l._w) for l in kfile.layers)} arrays",
             "  layer classes: " + ", ".join(f"{k} x {v}" for k, v in sorted(classes.items()))]
    try:
        g = graph_from_keras_config(kfile.config)
        w = weights_of_imported_graph(kfile, g)
        i0 = g.nodes[g.inputs[0]]
        lines.append(f"  imports as a {g.ndim}-D graph of {len(g.live_nodes())} nodes, {g.count_params():,} parameters ({sum(int(np.prod(v.shape)) for v in w.values()):,} found in the file); "
                     f"input {tuple(i0.shape if g.ndim == 2 else i0.shape[1:])}; outputs " + ", ".join(f"{n} {tuple(g.nodes[o].shape if g.ndim == 2 else g.nodes[o].shape[1:])}" for n, o in zip(g.output_names, g.outputs)))
        lines.append(f"  {2 * g.macs() / 1e6:,.1f} MFLOP per sample forward")
    except KerasImportError as e:
        lines.append(f"  NOT importable: {e}")
    return "\n".join(lines)
