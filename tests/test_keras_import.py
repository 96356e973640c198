"""CPU: `segamd.load_model(path)` - the reference's `tf.keras.saving.load_model` (2D/Test.py:114) - rebuilds the layer graph from the
checkpoint file alone (keras_import.py).  Held against the builder: the imported graph, run through the oracle's interpreter with the
file's weights, gives the outputs of the builder's graph with the same weights, bit for bit - for every fixture layout."""
import copy
import json
import os
import sys

import numpy as np
import pytest
import torch

import segamd
from segamd import keras_file as KF
from segamd import keras_import as KI
from ir_interp import run_graph

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, HERE)
from make_keras_files import CASES  # noqa: E402

FILES = os.path.join(HERE, "keras_files")
PLANTED = np.load(os.path.join(FILES, "planted.npz"))


def _builder_graph(dec, kw):
    kw = dict(kw)
    if dec == "UNet1D":
        return segamd.UNet1D(**kw).graph_UNet()
    if dec.startswith("1D:"):
        cls, method = dec[3:].split(".")
        return getattr((segamd.BCDUNet if cls == "BCDUNet" else segamd.UNet1D)(**kw), "graph_" + method)()
    w, d = kw.pop("model_width"), kw.pop("model_depth")
    return segamd.unet_model_builder(dec, 32, 32, w, d, train_mode="from_scratch", **kw).graph()


def _cfg(fname, layout):
    return open(os.path.join(FILES, fname.replace(".h5", ".config.json"))).read() if layout == "h5_weights" else None


@pytest.mark.parametrize("fname,dec,kw,layout", CASES, ids=[c[0] for c in CASES])
def test_imported_graph_computes_what_the_builders_graph_computes(fname, dec, kw, layout):
    kfile = KF.KerasFileModel(os.path.join(FILES, fname), _cfg(fname, layout))
    gi = KI.graph_from_keras_config(kfile.config)
    wi = KI.weights_of_imported_graph(kfile, gi)
    gb = _builder_graph(dec, kw)
    wb = {k.split("|", 1)[1].replace("|", "/"): v for k, v in PLANTED.items() if k.startswith(fname + "|")}
    assert gi.ndim == gb.ndim and gi.nodes[gi.inputs[0]].shape == gb.nodes[gb.inputs[0]].shape
    assert gi.count_params() == gb.count_params() and len(gi.outputs) == len(gb.outputs)
    assert sorted(gi.nodes[i].kind for i in gi.live_nodes()) == sorted(gb.nodes[i].kind for i in gb.live_nodes())
    assert [gi.nodes[o].shape for o in gi.outputs] == [gb.nodes[o].shape for o in gb.outputs]
    torch.manual_seed(0)
    H, W, C = gb.nodes[gb.inputs[0]].shape
    x = torch.randn(2, W, C) if gb.ndim == 1 else torch.randn(2, H, W, C)
    t = lambda w: {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in w.items()}
    for training in (False, True):
        a = run_graph(gi, x, t(wi), training=training)
        b = run_graph(gb, x, t(wb), training=training)
        for p, q in zip(a, b):
            assert torch.equal(p, q)


def _config_of(fname):
    return copy.deepcopy(KF.KerasFileModel(os.path.join(FILES, fname)).config)


def _layer(cfg, cls):
    return next(l for l in cfg["config"]["layers"] if l["class_name"] == cls)


@pytest.mark.parametrize("cls,key,value,says", [
    ("Conv2D", "padding", "valid", "padding='valid'"),
    ("Conv2D", "dilation_rate", [2, 2], "dilation_rate"),
    ("Conv2D", "activation", "tanh", "activation 'tanh'"),
    ("Conv2DTranspose", "output_padding", [1, 1], "output_padding"),
    ("BatchNormalization", "axis", [1], "axis"),
    ("MaxPooling2D", "strides", [1, 1], "strides"),
    ("ConvLSTM2D", "recurrent_activation", "sigmoid", "hard_sigmoid"),
    ("ConvLSTM2D", "return_sequences", True, "return_sequences"),
    ("Reshape", "target_shape", [2, 16, 16, 2], "target_shape"),
])
def test_what_the_engine_would_compute_differently_is_refused_by_name(cls, key, value, says):
    cfg = _config_of("unet_lstm_ag_ds.keras")
    l = _layer(cfg, cls)
    l["config"][key] = value
    with pytest.raises(KI.KerasImportError, match=says) as e:
        KI.graph_from_keras_config(cfg)
    assert l["name"] in str(e.value)


def test_layers_outside_the_hot_path_shared_layers_and_loose_dense_layers_are_refused():
    cfg = _config_of("unet_lstm_ag_ds.keras")
    _layer(cfg, "MaxPooling2D")["class_name"] = "AveragePooling2D"
    with pytest.raises(KI.KerasImportError, match="AveragePooling2D is not part of the UNet-family hot path"):
        KI.graph_from_keras_config(cfg)
    cfg = _config_of("unet_lstm_ag_ds.keras")
    l = _layer(cfg, "Conv2D")
    l["inbound_nodes"] = l["inbound_nodes"] * 2
    with pytest.raises(KI.KerasImportError, match="called 2 times"):
        KI.graph_from_keras_config(cfg)
    cfg = _config_of("unet_ae_tf212.keras")
    _layer(cfg, "Flatten")["class_name"] = "Dropout"          # a Dense on an unflattened tensor
    with pytest.raises(KI.KerasImportError, match="outside the Flatten -> Dense -> Dense -> Reshape bottleneck"):
        KI.graph_from_keras_config(cfg)
    # a Flatten / Dense that is a model output, or that something other than the next link of the bottleneck reads, has no node of
    # its own: importing it would return the tensor in front of the Flatten and drop the Dense weights (advisor, round 5)
    cfg = _config_of("unet_ae_tf212.keras")
    dense = [l["name"] for l in cfg["config"]["layers"] if l["class_name"] == "Dense"]
    cfg["config"]["output_layers"] = [[dense[0], 0, 0]]
    with pytest.raises(KI.KerasImportError, match="is a model output"):
        KI.graph_from_keras_config(cfg)
    cfg = _config_of("unet_ae_tf212.keras")
    flat = _layer(cfg, "Flatten")["name"]
    cfg["config"]["output_layers"] = cfg["config"]["output_layers"] + [[flat, 0, 0]]
    with pytest.raises(KI.KerasImportError, match="is a model output"):
        KI.graph_from_keras_config(cfg)
    cfg = _config_of("unet_ae_tf212.keras")
    extra = copy.deepcopy(_layer(cfg, "Activation"))
    extra["name"] = extra["config"]["name"] = "reads_the_first_dense"
    extra["inbound_nodes"] = [[[dense[0], 0, 0, {}]]]
    cfg["config"]["layers"].append(extra)
    cfg["config"]["output_layers"] = cfg["config"]["output_layers"] + [["reads_the_first_dense", 0, 0]]
    with pytest.raises(KI.KerasImportError, match="not the next link"):
        KI.graph_from_keras_config(cfg)
    # a non-linear activation fused into a conv that is not an output: refused at import time, by layer name
    cfg = _config_of("unet_lstm_ag_ds.keras")
    outs = {r[0] for r in cfg["config"]["output_layers"]}
    inner = next(l for l in cfg["config"]["layers"] if l["class_name"] == "Conv2D" and l["name"] not in outs)
    inner["config"]["activation"] = "relu"
    with pytest.raises(KI.KerasImportError, match="fused into a convolution that is not an output") as e:
        KI.graph_from_keras_config(cfg)
    assert inner["name"] in str(e.value)
    cfg = _config_of("unet_lstm_ag_ds.keras")
    cfg["class_name"] = "Sequential"
    with pytest.raises(KI.KerasImportError, match="Sequential"):
        KI.graph_from_keras_config(cfg)


def test_both_spellings_of_the_input_shape_and_of_leaky_relu():
    cfg = _config_of("unetpp_keras3.keras")
    ic = _layer(cfg, "InputLayer")["config"]
    ic["batch_shape"] = ic.pop("batch_input_shape")          # Keras 3
    leaky = [l for l in cfg["config"]["layers"] if l["class_name"] == "Activation" and isinstance(l["config"]["activation"], dict)
             and l["config"]["activation"]["class_name"] == "LeakyReLU"]
    assert leaky                                            # (the reference's Activation('LeakyReLU'): a serialised layer object)
    lr = leaky[0]["config"]["activation"]["config"]
    lr["negative_slope"] = lr.pop("alpha")                  # Keras 3's name for it
    if len(leaky) > 1:
        leaky[1]["config"]["activation"] = "LeakyReLU"      # the name as written in the reference: the layer's default slope, 0.3
    g = KI.graph_from_keras_config(cfg)
    assert g.nodes[g.inputs[0]].shape == (32, 32, 3)
    slopes = [n.attrs["alpha"] for n in g.nodes if n.kind == "act" and n.attrs["fn"] == "leaky_relu"]
    assert len(slopes) == len(leaky) and all(abs(a - 0.3) < 1e-7 for a in slopes)
    leaky[0]["config"]["activation"] = "leaky_relu"         # Keras 3's FUNCTION: slope 0.2
    g = KI.graph_from_keras_config(cfg)
    assert sorted(round(n.attrs["alpha"], 3) for n in g.nodes if n.kind == "act" and n.attrs["fn"] == "leaky_relu")[0] == 0.2
    relu = next(l for l in cfg["config"]["layers"] if l["class_name"] == "Activation" and isinstance(l["config"]["activation"], dict)
                and l["config"]["activation"]["class_name"] == "ReLU")
    relu["config"]["activation"]["config"]["max_value"] = 6.0
    with pytest.raises(KI.KerasImportError, match="max_value"):
        KI.graph_from_keras_config(cfg)
    relu["config"]["activation"]["config"]["max_value"] = None
    ic["batch_shape"] = [None, None, None, 3]
    with pytest.raises(KI.KerasImportError, match="fixed"):
        KI.graph_from_keras_config(cfg)


BUILDS = [
    ("UNet", dict(model_width=4, model_depth=2, num_channels=1, lstm=1, ag=1, ds=1)),
    ("UNetPP", dict(model_width=4, model_depth=3, num_channels=3, ds=1)),
    ("UNet", dict(model_width=4, model_depth=2, num_channels=3, ae=1, feature_number=12, is_transconv=False)),
    ("MultiResUNet", dict(model_width=8, model_depth=2, num_channels=3, ag=1, alpha=1.0)),
    ("UNet3P", dict(model_width=4, model_depth=2, num_channels=3, output_nums=3, final_activation="softmax")),
    ("UNetE", dict(model_width=4, model_depth=2, num_channels=2)),
    ("UNet1D", dict(length=64, model_depth=2, num_channel=2, model_width=4, kernel_size=3, problem_type="Regression", output_nums=1, ds=1, lstm=1, ag=1)),
    ("UNet1D", dict(length=64, model_depth=2, num_channel=1, model_width=4, kernel_size=5, problem_type="Classification", output_nums=3, ae=1, feature_number=6)),
    ("UNetP", dict(model_width=4, model_depth=3, num_channels=3, ds=1)),
    ("UNetPP", dict(model_width=6, model_depth=2, num_channels=5, lstm=1, is_transconv=False)),
    ("1D:UNet.UNetPP", dict(length=64, model_depth=3, num_channel=2, model_width=4, kernel_size=3, problem_type="Regression", output_nums=1, ds=1)),
    ("1D:UNet.MultiResUNet", dict(length=64, model_depth=2, num_channel=1, model_width=8, kernel_size=3, problem_type="Regression", output_nums=1, ds=0, alpha=1)),
    ("1D:BCDUNet.BCDUNet", dict(length=64, model_depth=2, num_channel=1, model_width=4, kernel_size=3, problem_type="Regression", output_nums=1, ds=0, lstm=1, dense_loop=2)),
]


@pytest.mark.parametrize("dec,kw", BUILDS, ids=[f"{d}-{i}" for i, (d, _k) in enumerate(BUILDS)])
def test_a_model_saved_as_keras_archive_comes_back_from_the_file_alone(dec, kw, tmp_path):
    """ModelCheckpoint('x.keras') on this side (SegModel.save_weights -> keras_import.save_model), then the Test.py flow:
    load_model(path) without the constructor call - same outputs - and load_weights(path) into a freshly built constructor call."""
    from segamd import harness
    g = _builder_graph(dec, kw)
    rng = np.random.default_rng(2)
    w = {}
    for p in g.param_specs():
        v = rng.standard_normal(p.shape).astype(np.float32)
        w[p.name] = (np.abs(v) + 0.5 if p.name.endswith("moving_variance") else v / np.sqrt(np.prod(p.shape[:-1])) if len(p.shape) >= 2 else v).astype(np.float32)
    path = str(tmp_path / "UNet_32_Fold_1.keras")
    harness.save_weights_file(w, path, g)
    # 1. the archive alone -> graph + weights -> same numbers as the builder's graph
    kfile = KF.KerasFileModel(path)
    gi = KI.graph_from_keras_config(kfile.config)
    wi = KI.weights_of_imported_graph(kfile, gi)
    assert gi.count_params() == g.count_params()
    torch.manual_seed(1)
    H, W, C = g.nodes[g.inputs[0]].shape
    x = torch.randn(2, W, C) if g.ndim == 1 else torch.randn(2, H, W, C)
    t = lambda d: {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in d.items()}
    for p_, q_ in zip(run_graph(gi, x, t(wi), training=True), run_graph(g, x, t(w), training=True)):
        assert torch.equal(p_, q_)
    # 2. the same file through the constructor-call route (resume branch of Train.py): structural names, bit for bit
    back = harness.load_weights_file(path, g)
    assert set(back) == set(w) and all(np.array_equal(back[k], w[k]) for k in w)
    # 3. the configuration is plain JSON in Keras 2's spelling: no '/' in layer names, every layer has the keys load_model needs
    cfg = kfile.config
    assert all("/" not in l["name"] and l["config"]["name"] == l["name"] for l in cfg["config"]["layers"])
    assert json.loads(json.dumps(cfg)) == cfg


def test_mangled_configurations_raise_kerasimporterror():
    """Random edits of a real configuration - dropped keys, wrong types, rewired inputs, changed extents: a graph comes out, or
    KerasImportError; nothing else."""
    import random
    rng = random.Random(11)
    base = _config_of("unet_lstm_ag_ds.keras")
    built = refused = 0
    for _ in range(300):
        cfg = copy.deepcopy(base)
        layers = cfg["config"]["layers"]
        for _m in range(rng.choice([1, 1, 2, 3])):
            l = rng.choice(layers)
            what = rng.random()
            if what < 0.3 and l["config"]:
                l["config"].pop(rng.choice(list(l["config"])))
            elif what < 0.55 and l["config"]:
                l["config"][rng.choice(list(l["config"]))] = rng.choice([None, -1, 0, 7, "x", [3], [1, 2, 3], {}, 2.5, True])
            elif what < 0.7:
                l["inbound_nodes"] = rng.choice(layers)["inbound_nodes"]
            elif what < 0.8:
                l["class_name"] = rng.choice(["Conv2D", "Dense", "Add", "Concatenate", "Reshape", "Cropping2D", "BatchNormalization"])
            elif what < 0.9:
                layers.remove(l)
            else:
                cfg["config"]["output_layers"] = rng.choice([[], [["nope", 0, 0]], None, [[l["name"], 0, 0]]])
        try:
            KI.graph_from_keras_config(cfg)
            built += 1
        except KI.KerasImportError:
            refused += 1
    assert built > 20 and refused > 100


def test_load_model_needs_no_gpu_to_read_the_file_and_the_engine_still_has_no_cpu_fallback():
    m = segamd.load_model(os.path.join(FILES, "unetpp_keras3.keras"))
    assert not m.built and len(m._weights_at_build) == 60 and m.graph.nodes[m.graph.inputs[0]].shape == (32, 32, 3)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.predict(np.zeros((2, 32, 32, 3), np.float32))


def test_only_concatenate_and_the_convlstm_touch_the_length_one_sequence():
    cfg = _config_of("unet_lstm_ag_ds.keras")
    layers = {l["name"]: l for l in cfg["config"]["layers"]}
    lstm = _layer(cfg, "ConvLSTM2D")
    cat = layers[KF._refs(lstm["inbound_nodes"])[0]]
    assert cat["class_name"] == "Concatenate" and all(layers[r]["class_name"] == "Reshape" for r in KF._refs(cat["inbound_nodes"]))
    g = KI.graph_from_keras_config(cfg)                  # Reshape((1, H, W, C)) twice -> Concatenate (5-D) -> ConvLSTM2D, as unet_variants.py:145-149
    assert sum(1 for i in g.live_nodes() if g.nodes[i].kind == "convlstm") == 2
    cat["class_name"], cat["config"] = "Add", {"name": cat["name"]}
    with pytest.raises(KI.KerasImportError, match="length-1 sequence"):
        KI.graph_from_keras_config(cfg)


def test_describe_says_what_a_file_is_and_what_is_in_the_way(tmp_path):
    text = KI.describe(os.path.join(FILES, "unet_lstm_ag_ds.keras"))
    assert ".keras archive" in text and "TFOpLambda x 2" in text and "imports as a 2-D graph" in text and "level2 (8, 8, 1)" in text
    cfg = _config_of("unet_lstm_ag_ds.keras")
    pool = _layer(cfg, "MaxPooling2D")
    pool["config"]["strides"] = [1, 1]
    text = KI.describe(os.path.join(FILES, "unet_lstm_ag_ds.keras"), config=json.dumps(cfg))
    assert "NOT importable" in text and pool["name"] in text and "strides" in text


def test_random_builder_calls_survive_the_archive_round_trip():
    """A fixed sample of tools/fuzz_archives.py (CPU only): 40 random constructor calls of the builder API -> save_weights('x.keras') ->
    the archive alone back to a graph (load_model's path) and onto the constructor call (load_weights's path); the imported graph
    computes what the builder's graph computes, bit for bit."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_archives as FA
    ok, refused, fails = FA.run(5, 40, verbose=False)
    assert not fails, fails[:3]
    assert ok >= 30


def test_a_dropout_that_drops_is_refused():
    cfg = _config_of("unet_lstm_ag_ds.keras")
    pool = _layer(cfg, "MaxPooling2D")
    name = pool["name"]
    pool["class_name"], pool["config"] = "Dropout", {"name": name, "rate": 0.0}
    # (a rate-0 Dropout in place of a pool changes the shapes downstream: what matters here is that IT is accepted and a later layer objects)
    with pytest.raises(KI.KerasImportError) as e:
        KI.graph_from_keras_config(cfg)
    assert name not in str(e.value).split(":")[0]
    pool["config"]["rate"] = 0.3
    with pytest.raises(KI.KerasImportError, match="no dropout"):
        KI.graph_from_keras_config(cfg)


def test_a_frozen_layer_is_refused():
    cfg = _config_of("unet_lstm_ag_ds.keras")
    l = _layer(cfg, "BatchNormalization")
    l["config"]["trainable"] = False
    with pytest.raises(KI.KerasImportError, match="frozen layer") as e:
        KI.graph_from_keras_config(cfg)
    assert l["name"] in str(e.value)


def test_regularisers_and_constraints_are_refused():
    cfg = _config_of("unet_lstm_ag_ds.keras")
    l = _layer(cfg, "Conv2D")
    l["config"]["kernel_regularizer"] = {"class_name": "L2", "config": {"l2": 0.01}}
    with pytest.raises(KI.KerasImportError, match="kernel_regularizer"):
        KI.graph_from_keras_config(cfg)
    l["config"]["kernel_regularizer"] = None                  # (what get_config() writes when there is none)
    l["config"]["bias_constraint"] = None
    KI.graph_from_keras_config(cfg)
