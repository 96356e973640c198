#!/usr/bin/env python3
"""Writes tests/golden/keras_files/: Keras checkpoint files of small models of the reference's builder calls, in the layouts
keras_file.py documents, through the REAL HDF5 library - plus planted.npz, the weights each file holds under their structural names.

Two stages, because no interpreter of this image has everything: the system python (this package, no h5py) lays the models out -
the Keras view of a builder graph (`keras_view`: get_config()-style layer configurations, auto-names with process-global counters,
model.layers in an unrelated order, pairwise concatenates, Reshape in front of ConvLSTM, Flatten -> Dense -> Dense -> Reshape) - and hands a JSON + npz
to /opt/conda's python3.9 (h5py 3.3 / HDF5 1.10.6, no torch), which writes the files:

    python tests/golden/make_keras_files.py

NOT output of Keras (no TensorFlow in this image): the layouts are restated from the formats' documentation, so these fixtures pin
h5lite / keras_file against the real HDF5 library and the documented layout, not against a Keras release.  Build-authored."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "keras_files")
CONDA = "/opt/conda/bin/python3.9"

CASES = [      # (file, decoder, constructor keywords, layout)
    ("unet_lstm_ag_ds.keras", "UNet", dict(model_width=4, model_depth=2, num_channels=1, lstm=1, ag=1, ds=1), "keras_counted"),
    ("unetpp_keras3.keras", "UNetPP", dict(model_width=4, model_depth=2, num_channels=3, ds=1), "keras_counted_k3"),
    ("unet_ae_tf212.keras", "UNet", dict(model_width=4, model_depth=2, num_channels=3, ae=1, feature_number=12), "keras_tf212"),
    ("multires_legacy.h5", "MultiResUNet", dict(model_width=8, model_depth=1, num_channels=3, ag=1, alpha=1.0), "h5_model"),
    ("unet3p_weights_only.h5", "UNet3P", dict(model_width=4, model_depth=2, num_channels=3), "h5_weights"),
    ("unet1d.keras", "UNet1D", dict(length=64, model_depth=2, num_channel=2, model_width=4, kernel_size=3, problem_type="Regression", output_nums=1, ds=1), "keras_counted"),
    # auto-names of a FRESH process (conv2d, conv2d_1, ...): the same strings as the counted weight keys, which follow model.layers order
    ("unet_fresh_names.keras", "UNet", dict(model_width=4, model_depth=2, num_channels=3, ag=1, ds=1), "keras_counted_fresh"),
    ("unet1d_lstm_ag.keras", "UNet1D", dict(length=64, model_depth=2, num_channel=1, model_width=4, kernel_size=3, problem_type="Regression", output_nums=1, ds=0, ag=1, lstm=1), "keras_counted"),
]

ONE_D = {"Conv2D": "Conv1D", "Conv2DTranspose": "Conv1DTranspose", "MaxPooling2D": "MaxPooling1D", "UpSampling2D": "UpSampling1D", "ConvLSTM2D": "ConvLSTM1D"}
AUTO = {"TFOpLambda": "tf.math.multiply", "Conv2D": "conv2d", "Conv2DTranspose": "conv2d_transpose", "BatchNormalization": "batch_normalization", "Activation": "activation",
        "LeakyReLU": "leaky_re_lu", "MaxPooling2D": "max_pooling2d", "UpSampling2D": "up_sampling2d", "Concatenate": "concatenate", "Add": "add",
        "Multiply": "multiply", "ConvLSTM2D": "conv_lstm2d", "InputLayer": "input", "Reshape": "reshape", "Flatten": "flatten", "Dense": "dense"}


def keras_view(g, rng, one_d, name_offset=17):
    """The layers a tf.keras model of the reference has for builder graph `g`, as (name, class, get_config()-style dict, inbound
    layer names, [(structural parameter name, array)]): process-global auto-names, pairwise-folded concatenates (Concat_Block,
    unet_variants.py:27-32), Reshape((1, ...)) in front of every ConvLSTM (:147), Flatten -> Dense('features') -> Dense -> Reshape for
    the autoencoder bottleneck (:41-47), separate Activation / LeakyReLU layers, heads with their explicit names."""
    counters, layers, layer_of = {}, [], {}
    outset = set(g.outputs)
    cons, sequences = g.consumers(), set()
    sp = (lambda hw: [int(hw[1])] if one_d else [int(hw[0]), int(hw[1])])

    def auto(cls):
        i = counters.get(cls, name_offset)
        counters[cls] = i + 1
        base = AUTO[cls]
        base = base.replace('2d', '1d') if one_d else base
        return base if i == 0 else f"{base}_{i}"          # (a fresh process: conv2d, conv2d_1, ... - the strings of the counted weight keys, in CREATION order)

    def emit(cls, name, cfg, inbound, weights=()):
        layers.append(dict(name=name, cls=ONE_D.get(cls, cls) if one_d else cls, config=dict(cfg, name=name), inbound=list(inbound), weights=list(weights)))
        return name

    def planted(n, params):
        out = []
        for p in params:
            w = rng.standard_normal(p.shape).astype(np.float32)
            if p.name.endswith("moving_variance"):
                w = np.abs(w) + 0.5
            elif len(p.shape) >= 2:
                w = (w / np.sqrt(np.prod(p.shape[:-1]))).astype(np.float32)          # (values a trained model could have: the file runs)
            out.append((p.name, w))
        return out

    for i in g.live_nodes():
        n = g.nodes[i]
        ins = [layer_of[j] for j in n.inputs]
        a = n.attrs
        H, W, C = n.shape
        if n.kind == "input":
            layer_of[i] = emit("InputLayer", "input_1", dict(batch_input_shape=[None] + sp((H, W)) + [C], dtype="float32", sparse=False), [])
        elif n.kind in ("conv", "convT"):
            cls = "Conv2D" if n.kind == "conv" else "Conv2DTranspose"
            pointwise = all(k == 1 for k in sp(a["kernel"])) and n.kind == "conv"          # the reference leaves padding at 'valid' on its 1x1 convs
            cfg = dict(filters=C, kernel_size=sp(a["kernel"]), strides=sp(a["strides"]), padding="valid" if pointwise else "same", data_format="channels_last",
                       dilation_rate=sp((1, 1)), groups=1, activation=a.get("activation") or "linear", use_bias=a.get("use_bias", True))
            if n.kind == "convT":
                cfg["output_padding"] = None
            layer_of[i] = emit(cls, n.name if i in outset else auto(cls), cfg, ins, planted(n, n.params))
        elif n.kind == "bn":
            layer_of[i] = emit("BatchNormalization", auto("BatchNormalization"),
                               dict(axis=[2 if one_d else 3], momentum=a["momentum"], epsilon=a["eps"], center=True, scale=True), ins, planted(n, n.params))
        elif n.kind == "act":
            # the reference writes Activation('ReLU') / Activation('LeakyReLU') in its Conv_Block / trans_conv2D (unet_variants.py:7, 17):
            # Keras 2 resolves those to layer objects and serialises the activation as a dictionary; Activation('relu') / ('sigmoid')
            # elsewhere stay strings.  Every other ReLU here takes the dictionary form, so that both spellings are in the fixtures.
            nm = n.name if i in outset else auto("Activation")
            if a["fn"] == "leaky_relu":
                act = {"class_name": "LeakyReLU", "config": {"name": "leaky_re_lu", "trainable": True, "dtype": "float32", "alpha": a["alpha"]}}
            elif a["fn"] == "relu" and counters.get("Activation", 0) % 2 == 0:
                act = {"class_name": "ReLU", "config": {"name": "re_lu", "trainable": True, "dtype": "float32", "max_value": None, "negative_slope": 0.0, "threshold": 0.0}}
            else:
                act = a["fn"]
            layer_of[i] = emit("Activation", nm, dict(activation=act), ins)
        elif n.kind == "maxpool":
            layer_of[i] = emit("MaxPooling2D", auto("MaxPooling2D"), dict(pool_size=sp(a["size"]), strides=sp(a["size"]), padding="valid"), ins)
        elif n.kind == "upsample":
            cfg = dict(size=sp(a["size"])[0] if one_d else sp(a["size"]))
            if not one_d:
                cfg["interpolation"] = a["interpolation"]
            layer_of[i] = emit("UpSampling2D", auto("UpSampling2D"), cfg, ins)
        elif n.kind == "concat":
            cur = ins[0]
            feeds_lstm = [g.nodes[j].kind for j in cons.get(i, [])] == ["convlstm"]
            if feeds_lstm:          # unet_variants.py:145-147: each input reshaped to a length-1 sequence, THEN concatenated
                ins = [emit("Reshape", auto("Reshape"), dict(target_shape=[1] + sp(g.nodes[j].shape[:2]) + [g.nodes[j].shape[2]]), [l_])
                       for j, l_ in zip(n.inputs, ins)]
                cur = ins[0]
            for nxt in ins[1:]:
                cur = emit("Concatenate", auto("Concatenate"), dict(axis=-1), [cur, nxt])
            layer_of[i] = cur
            if feeds_lstm:
                sequences.add(i)
        elif n.kind == "add":
            layer_of[i] = emit("Add", auto("Add"), {}, ins)
        elif n.kind == "mul":
            # `skip_connection * resampler` (unet_variants.py:81): TF 2.x wraps the operator in a TFOpLambda layer; the layouts that
            # carry Keras 3's spelling get its Multiply operation instead (stage 2 looks at the class)
            k_ = counters.setdefault("TFOpLambda", 0)
            layer_of[i] = emit("TFOpLambda", "tf.math.multiply" if k_ == 0 else f"tf.math.multiply_{k_}", dict(function="math.multiply"), ins)
            counters["TFOpLambda"] += 1
        elif n.kind == "convlstm":
            Hi, Wi, Ci = g.nodes[n.inputs[0]].shape
            r = ins[0] if n.inputs[0] in sequences else emit("Reshape", auto("Reshape"), dict(target_shape=[1] + sp((Hi, Wi)) + [Ci]), ins)
            layer_of[i] = emit("ConvLSTM2D", auto("ConvLSTM2D"),
                               dict(filters=C, kernel_size=sp(a["kernel"]), strides=sp((1, 1)), padding="same", dilation_rate=sp((1, 1)), activation="tanh",
                                    recurrent_activation="hard_sigmoid", use_bias=True, return_sequences=False, go_backwards=True, stateful=False),
                               [r], planted(n, n.params))
        elif n.kind == "dense_ae":
            ws = planted(n, n.params)
            fl = emit("Flatten", auto("Flatten"), dict(data_format="channels_last"), ins)
            d1 = emit("Dense", "features", dict(units=a["features"], activation="linear", use_bias=True), [fl], ws[0:2])
            d2 = emit("Dense", auto("Dense"), dict(units=H * W * C, activation="linear", use_bias=True), [d1], ws[2:4])
            layer_of[i] = emit("Reshape", auto("Reshape"), dict(target_shape=sp((H, W)) + [C]), [d2])
        else:
            raise ValueError(n.kind)
    outputs = [layer_of[o] for o in g.outputs]
    import random
    random.Random(3).shuffle(layers)                    # model.layers is in graph-depth order, not creation order
    return layers, outputs


def stage1(tmp):
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import segamd
    from segamd import keras_file
    spec, arrays, planted = [], {}, {}
    for ci, (fname, dec, kw, layout) in enumerate(CASES):
        kw = dict(kw)
        one_d = dec == "UNet1D"
        if one_d:
            g = segamd.UNet1D(**kw).graph_UNet()
        else:
            w, d = kw.pop("model_width"), kw.pop("model_depth")
            g = segamd.unet_model_builder(dec, 32, 32, w, d, train_mode="from_scratch", **kw).graph()
        layers, outputs = keras_view(g, np.random.default_rng(100 + ci), one_d, name_offset=0 if layout.endswith("_fresh") else 17)
        for l in layers:
            keys = []
            for wi, (pname, w) in enumerate(l["weights"]):
                planted[f"{fname}|{pname.replace('/', '|')}"] = w
                if one_d and w.ndim == 4 and w.shape[0] == 1:
                    w = w[0]                            # Conv1D kernels are (k, Cin, Cout)
                key = f"{ci}:{l['name']}:{wi}"
                arrays[key] = w
                keys.append(key)
            l["weights"] = keys
        spec.append(dict(file=fname, layout=layout, layers=layers, outputs=outputs,
                         snake={l["cls"]: keras_file._snake(l["cls"]) for l in layers}))
    with open(os.path.join(tmp, "spec.json"), "w") as fh:
        json.dump(spec, fh)
    np.savez(os.path.join(tmp, "arrays.npz"), **arrays)
    os.makedirs(OUT, exist_ok=True)
    np.savez(os.path.join(OUT, "planted.npz"), **planted)


WEIGHT_NAMES = {"ConvLSTM": ("kernel", "recurrent_kernel", "bias"), "Conv": ("kernel", "bias"),
                "BatchNormalization": ("gamma", "beta", "moving_mean", "moving_variance"), "Dense": ("kernel", "bias")}


def stage2(tmp):
    import io
    import zipfile

    import h5py
    spec = json.load(open(os.path.join(tmp, "spec.json")))
    arrays = np.load(os.path.join(tmp, "arrays.npz"))
    for case in spec:
        layers, layout = case["layers"], case["layout"]

        def node(l):
            if not l["inbound"]:
                return []
            if l["cls"] == "TFOpLambda" and layout != "keras_counted_k3":          # TF 2.x: the second operand travels in the call's kwargs
                a_, b_ = l["inbound"]
                return [[a_, 0, 0, {"y": [b_, 0, 0], "name": None}]]
            if layout == "keras_counted_k3":              # Keras 3 spelling
                kts = [{"class_name": "__keras_tensor__", "config": {"shape": [None], "dtype": "float32", "keras_history": [n, 0, 0]}} for n in l["inbound"]]
                return [{"args": [kts if len(kts) > 1 else kts[0]], "kwargs": {}}]
            return [[[n, 0, 0, {}] for n in l["inbound"]]]

        for l in layers:
            if l["cls"] == "TFOpLambda" and layout == "keras_counted_k3":          # Keras 3: the operator is an operation of class Multiply
                l["cls"], l["config"] = "Multiply", {"name": l["name"]}
        cfg_layers = [{"module": "keras.layers", "class_name": l["cls"], "config": dict(l["config"], trainable=True, dtype="float32"),
                       "registered_name": None, "name": l["name"], "inbound_nodes": node(l)} for l in layers]
        outs = [[n, 0, 0] for n in case["outputs"]]
        config = {"module": "keras.src.engine.functional", "class_name": "Functional",
                  "config": {"name": "model", "trainable": True, "layers": cfg_layers,
                             "input_layers": [[l["name"], 0, 0] for l in layers if l["cls"] == "InputLayer"],
                             "output_layers": outs[0] if (layout == "keras_counted_k3" and len(outs) == 1) else outs},
                  "registered_name": "Functional", "build_config": {"input_shape": None}}
        path = os.path.join(OUT, case["file"])
        if layout.startswith("keras"):
            bio = io.BytesIO()
            with h5py.File(bio, "w") as f:
                top = "_layer_checkpoint_dependencies" if layout == "keras_tf212" else "layers"
                f.create_group("vars")
                used = {}
                for l in layers:
                    base = case["snake"][l["cls"]]
                    if base in used:
                        used[base] += 1
                        counted = f"{base}_{used[base]}"
                    else:
                        used[base] = 0
                        counted = base
                    key = counted
                    g = f.create_group(f"{top}/{key}")
                    holder = g
                    g.create_group("vars")
                    if l["cls"].startswith("ConvLSTM"):          # the recurrent layer keeps its variables in its cell
                        holder = g.create_group("cell")
                        holder.create_group("vars")
                    for i, k in enumerate(l["weights"]):
                        holder["vars"].create_dataset(str(i), data=arrays[k])
                f.create_dataset("optimizer/vars/0", data=np.int64(1234))
            with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
                z.writestr("metadata.json", json.dumps({"keras_version": "2.13.1", "date_saved": "2026-10-03@00:00:00"}))
                z.writestr("config.json", json.dumps(config))
                z.writestr("model.weights.h5", bio.getvalue())
        else:
            with h5py.File(path, "w") as f:
                root = f
                if layout == "h5_model":
                    f.attrs["model_config"] = json.dumps(config)
                    f.attrs["keras_version"] = "2.11.0"
                    f.attrs["backend"] = "tensorflow"
                    root = f.create_group("model_weights")
                else:
                    with open(os.path.join(OUT, case["file"].replace(".h5", ".config.json")), "w") as fh:
                        json.dump(config, fh)
                root.attrs["layer_names"] = np.array([l["name"].encode() for l in layers])
                root.attrs["backend"] = b"tensorflow"
                root.attrs["keras_version"] = b"2.11.0"
                for l in layers:
                    g = root.create_group(l["name"])
                    fam = next((v for k, v in WEIGHT_NAMES.items() if l["cls"].startswith(k)), ())
                    names = [f"{l['name']}/{fam[i]}:0" for i in range(len(l["weights"]))]
                    g.attrs["weight_names"] = np.array([n.encode() for n in names]) if names else np.zeros((0,), "S1")
                    for n, k in zip(names, l["weights"]):
                        g.create_dataset(n, data=arrays[k])
        print(case["file"], os.path.getsize(path), "bytes,", len(layers), "layers")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--write":
        stage2(sys.argv[2])
    else:
        with tempfile.TemporaryDirectory() as tmp:
            stage1(tmp)
            env = {k: v for k, v in os.environ.items() if k not in ("PYTHONHOME", "PYTHONPATH")}
            subprocess.run([CONDA, os.path.abspath(__file__), "--write", tmp], check=True, env=env)
