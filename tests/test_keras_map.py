"""CPU: the Keras graph-walk mapper (keras_map.py, SURVEY.md section 8 rows c / f3) on a hand-built fake layer graph.

The fake model has what makes a real `tf.keras.Model` of the reference awkward: process-global auto-names, `model.layers`
in an order unrelated to creation, pairwise-folded concatenates (Concat_Block), Reshape layers around the ConvLSTM inputs
and the Flatten -> Dense -> Dense -> Reshape autoencoder bottleneck - only `inbound_nodes` tells the structure."""
import itertools
import random

import numpy as np
import pytest

import segamd
from segamd import keras_map as KM

CLASS_OF_KIND = {"input": "InputLayer", "conv": "Conv2D", "convT": "Conv2DTranspose", "bn": "BatchNormalization", "act": "Activation",
                 "maxpool": "MaxPooling2D", "upsample": "UpSampling2D", "concat": "Concatenate", "add": "Add", "mul": "Multiply",
                 "convlstm": "ConvLSTM2D"}


class _Node:
    def __init__(self, inbound):
        self.inbound_layers = inbound if len(inbound) != 1 else inbound[0]


class _History:
    def __init__(self, layer):
        self.layer = layer


class _Tensor:
    def __init__(self, layer):
        self._keras_history = _History(layer)


def _make_class(name):
    class L:
        def __init__(self, lname, inbound, weights=()):
            self.name = lname
            self._inbound_nodes = [_Node(list(inbound))] if inbound else []
            self._w = [np.array(w) for w in weights]

        def get_weights(self):
            return [w.copy() for w in self._w]

        def set_weights(self, ws):
            assert len(ws) == len(self._w) and all(a.shape == b.shape for a, b in zip(ws, self._w))
            self._w = [np.array(w) for w in ws]
    L.__name__ = name
    return L


_CLS = {}


def K(cls, *a, **k):
    if cls not in _CLS:
        _CLS[cls] = _make_class(cls)
    return _CLS[cls](*a, **k)


class FakeKerasModel:
    """Built from a layer graph of this package the way the reference would have built it in Keras."""

    def __init__(self, g, rng, name_offset=17):
        counters = {}
        self.planted = {}

        def auto(cls):
            base = {"Conv2D": "conv2d", "Conv2DTranspose": "conv2d_transpose", "BatchNormalization": "batch_normalization",
                    "Activation": "activation", "MaxPooling2D": "max_pooling2d", "UpSampling2D": "up_sampling2d",
                    "Concatenate": "concatenate", "Add": "add", "Multiply": "multiply", "ConvLSTM2D": "conv_lstm2d",
                    "InputLayer": "input", "Reshape": "reshape", "Flatten": "flatten", "Dense": "dense"}[cls]
            i = counters.get(cls, name_offset)
            counters[cls] = i + 1
            return f"{base}_{i}"

        layer_of = {}
        self.layers = []
        outset = set(g.outputs)

        def weights_of(n):
            ws = []
            for p in n.params:
                w = rng.standard_normal(p.shape).astype(np.float32)
                self.planted[p.name] = w
                ws.append(w)
            return ws

        for i in g.live_nodes():
            n = g.nodes[i]
            ins = [layer_of[j] for j in n.inputs]
            if n.kind == "concat":                 # Concat_Block: pairwise left fold
                cur = ins[0]
                for nxt in ins[1:]:
                    cur = K("Concatenate", auto("Concatenate"), [cur, nxt])
                    self.layers.append(cur)
                layer_of[i] = cur
                continue
            if n.kind == "convlstm":               # Reshape -> (1, H, W, C) in front of the recurrent layer
                r = K("Reshape", auto("Reshape"), ins)
                self.layers.append(r)
                ins = [r]
            if n.kind == "dense_ae":
                fl = K("Flatten", auto("Flatten"), ins)
                d1 = K("Dense", "features", [fl], weights_of(type("N", (), {"params": n.params[0:2]})()))
                d2 = K("Dense", auto("Dense"), [d1], weights_of(type("N", (), {"params": n.params[2:4]})()))
                rs = K("Reshape", auto("Reshape"), [d2])
                self.layers += [fl, d1, d2, rs]
                layer_of[i] = rs
                continue
            cls = CLASS_OF_KIND[n.kind]
            lname = n.name if i in outset else auto(cls)          # only the heads carry explicit names ('out', 'level1', ...)
            layer = K(cls, lname, ins, weights_of(n))
            self.layers.append(layer)
            layer_of[i] = layer
        self.outputs = [_Tensor(layer_of[o]) for o in g.outputs]
        random.Random(3).shuffle(self.layers)      # graph-depth order of a real model is NOT creation order either


CONFIGS = [
    ("UNet", dict(model_width=8, model_depth=2, num_channels=3)),
    ("UNetPP", dict(model_width=8, model_depth=3, num_channels=3, ds=1)),
    ("UNet", dict(model_width=8, model_depth=2, num_channels=1, lstm=1, ag=1, ds=1)),
    ("UNet", dict(model_width=8, model_depth=2, num_channels=3, ae=1, feature_number=16)),
    ("MultiResUNet", dict(model_width=16, model_depth=2, num_channels=3, ag=1, alpha=1.0)),
    ("UNet3P", dict(model_width=8, model_depth=2, num_channels=3)),
]


def _graph(dec, kw):
    kw = dict(kw)
    w, d = kw.pop("model_width"), kw.pop("model_depth")
    return segamd.unet_model_builder(dec, 32, 32, w, d, train_mode="from_scratch", **kw).graph()


@pytest.mark.parametrize("dec,kw", CONFIGS)
def test_every_weight_finds_its_structural_name(dec, kw):
    g = _graph(dec, kw)
    fake = FakeKerasModel(g, np.random.default_rng(0))
    got = KM.keras_weights_by_structural_name(fake, g)
    want = {p.name for p in g.param_specs()}
    assert set(got) == want
    for k, v in got.items():
        assert np.array_equal(v, fake.planted[k]), k
    names = KM.map_keras_model(fake, g)
    assert set(names.values()) >= {n.name for n in (g.nodes[i] for i in g.live_nodes()) if n.params and n.kind != "dense_ae"}
    assert all(not k.startswith("reshape") and not k.startswith("flatten") for k in names)          # pass-through layers have no counterpart


def test_export_round_trip_and_structure_mismatch():
    g = _graph("UNetPP", dict(model_width=8, model_depth=2, num_channels=3, ds=1))
    fake = FakeKerasModel(g, np.random.default_rng(1))
    new = {p.name: np.random.default_rng(2).standard_normal(p.shape).astype(np.float32) for p in g.param_specs()}
    KM.export_to_keras(new, fake, g)
    back = KM.keras_weights_by_structural_name(fake, g)
    assert all(np.array_equal(back[k], new[k]) for k in new)
    # a model built with other flags does not map: the walk reports where the structures part
    other = _graph("UNetPP", dict(model_width=8, model_depth=2, num_channels=3, ds=1, ag=1))
    with pytest.raises(KM.KerasMapError):
        KM.map_keras_model(fake, other)
    deeper = _graph("UNetPP", dict(model_width=8, model_depth=3, num_channels=3, ds=1))
    with pytest.raises(KM.KerasMapError):
        KM.map_keras_model(fake, deeper)
