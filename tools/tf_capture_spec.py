#!/usr/bin/env python3
"""Golden-fixture capture for a machine that HAS TensorFlow 2.12-2.15 and a checkout of the reference repository
(Sakib1263/TF-1D-2D-Segmentation-End2EndPipelines).  Not runnable in the build container (no TensorFlow there): this is
the committed SPEC of the capture SURVEY.md section 8c asks for, and the only route by which the oracle's parity
status changes from "unpinned" to "pinned".  It copies nothing from the reference: it imports whatever
`models.unet_variants` is importable and calls the public constructors.

    PYTHONPATH=<reference>/TensorFlow/2DCNN:<reference>/TensorFlow/1DCNN:<this repo> python tools/tf_capture_spec.py --out tests/golden

For every case below it
  1. builds the Keras model through the reference's own `unet_model_builder(...)` call (d2 / w8 / 64x64: fixtures < 1 MB),
  2. re-initialises the BatchNormalization statistics to non-trivial values (seeded) so that inference-mode outputs depend
     on them,
  3. maps every Keras layer to the structural name of this package's graph for the same constructor call by walking
     `inbound_nodes` (keras_map.map_keras_model - never by list position or auto-name),
  4. runs `model.predict(x)` (inference-mode BN) and one `train_on_batch`-free training-mode forward
     (`model(x, training=True)`) on x = np.random.default_rng(0).random((2, H, W, C), float32),
  5. keeps the BatchNormalization moving statistics as that training-mode call left them ('moving_after/<name>': Keras'
     momentum 0.99 update with the UNBIASED batch variance on 4-D inputs - one of the semantics SURVEY 8c marks),
  6. takes ONE optimiser step - `compile(Adam(1e-3), 'mse')`, `train_on_batch(x, t)` with t = 0.5 everywhere - and keeps the loss it
     reports and every weight afterwards ('loss', 'w_after/<name>': Keras' Adam with epsilon outside the root, the MSE reduction,
     the gradient of every layer type on the path),
  6b. takes EXTRA_STEPS more steps on the same batch and keeps `predict(x)` afterwards ('predict_after_steps/<i>'),
  7. writes tests/golden/tf_<case>.npz: 'meta' (json: constructor arguments, TF / Keras versions), 'x', 'w/<structural name>'
     for every weight, 'predict/<i>' and 'train_fwd/<i>' for every output, and the arrays of 5. and 6.
The cases cover every semantic SURVEY.md 8c marks as a risk in ONE run: Conv2DTranspose 'same' cropping and LeakyReLU 0.3
(every transposed-conv case), the hard-sigmoid ConvLSTM gates (`lstm=1`, 2D and 1D), the attention gate's bilinear + 1-channel
transposed conv (`ag=1`), BatchNormalization over a concatenation (MultiResUNet), the 1D builders (Conv1D / k2-s2
Conv1DTranspose with BN + ReLU, UpSampling1D) and BCDUNet.
tests/test_tf_fixture.py picks the files up: the oracle (CPU) and the HIP engine in f32 mode (GPU) must reproduce
'predict/*' and 'train_fwd/*' within 1e-4 / 1e-3 with bit-exact masks; the tests skip while no fixture is present.
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CASES = {
    # name: (decoder, kwargs of unet_model_builder beyond (decoder_name, length, width, model_width, model_depth))
    "unet": ("UNet", dict(num_channels=3)),
    "unet_softmax4": ("UNet", dict(num_channels=3, output_nums=4, final_activation="softmax")),
    "unetpp_ds": ("UNetPP", dict(num_channels=3, ds=1)),
    "unet_ag_ds": ("UNet", dict(num_channels=3, ag=1, ds=1)),
    "unet_lstm": ("UNet", dict(num_channels=1, lstm=1)),
    "unet_upsampling": ("UNet", dict(num_channels=3, is_transconv=False)),
    "unet3p": ("UNet3P", dict(num_channels=3)),
    "multiresunet_ag": ("MultiResUNet", dict(num_channels=3, ag=1, alpha=1.0)),
    # widths that are NOT multiples of 8 (third element: model_width instead of WIDTH).  Round 5's fuzzer found every defect of the
    # round behind them: padding channels (sigmoid(0) = 0.5 leaking into them: UNet 3+, the attention gate's 1-channel map), ConvLSTM
    # gate blocks, Dense-bottleneck extents - and none of them shows in the first step: hence 'predict_after_steps/*' below
    "unet3p_w12": ("UNet3P", dict(num_channels=3), 12),
    "multiresunet_ag_w12": ("MultiResUNet", dict(num_channels=3, ag=1, alpha=1.67), 12),
    "unet_lstm_ae_w12": ("UNet", dict(num_channels=5, lstm=1, ae=1, feature_number=10), 12),
}
EXTRA_STEPS = 2       # optimiser steps after the first one before 'predict_after_steps/*' is taken
# 1D: (module of the reference, class, method, positional (length, model_depth, num_channel, model_width, kernel_size), kwargs)
CASES_1D = {
    "unet1d": ("Models.unet_variants", "UNet", "UNet", (256, 2, 1, 8, 3), dict(problem_type="Regression", output_nums=1, ds=1)),
    "unet1d_lstm_ag": ("Models.unet_variants", "UNet", "UNet", (256, 2, 1, 8, 3), dict(problem_type="Regression", output_nums=1, ds=0, ag=1, lstm=1)),
    "unetpp1d_cls": ("Models.unet_variants", "UNet", "UNetPP", (256, 2, 1, 8, 3), dict(problem_type="Classification", output_nums=2, ds=0)),
    "multiresunet1d": ("Models.unet_variants", "UNet", "MultiResUNet", (256, 2, 1, 8, 3), dict(problem_type="Regression", output_nums=1, ds=0, alpha=1)),
    "bcdunet1d": ("Models.BCDUNet", "BCDUNet", "BCDUNet", (256, 2, 1, 8, 3), dict(problem_type="Regression", output_nums=1, ds=0, lstm=1, dense_loop=2)),
}
H = W = 64
WIDTH, DEPTH = 8, 2


def capture(tf, kmodel, g, x, meta, keras_map, path):
    """Steps 2 - 7 of the module docstring for one Keras model / one graph of this package."""
    rng = np.random.default_rng(1)
    for layer in kmodel.layers:                            # non-trivial moving statistics, gamma / beta
        if layer.__class__.__name__ == "BatchNormalization":
            gm, bt, mu, var = layer.get_weights()
            layer.set_weights([rng.uniform(0.5, 1.5, gm.shape), rng.normal(0, 0.1, bt.shape),
                               rng.normal(0, 0.1, mu.shape), rng.uniform(0.5, 1.5, var.shape)])
    weights = keras_map.keras_weights_by_structural_name(kmodel, g)
    missing = {p.name for p in g.param_specs()} - set(weights)
    assert not missing, f"{meta['case']}: parameters without a Keras counterpart: {sorted(missing)}"
    # the same weights as checkpoint FILES, the way the reference's ModelCheckpoint / model.save leave them (Train.py:375-379): what
    # keras_file.py reads without TensorFlow is held against 'w/*' by tests/test_tf_fixture.py - the first REAL Keras files it sees
    for ext in (".keras", ".h5"):
        try:
            kmodel.save(path[:-4] + ext)
        except Exception as e:          # noqa: BLE001  (a Keras release without one of the formats: say so, keep going)
            print(f"{meta['case']}: model.save('{ext}') failed: {type(e).__name__}: {e}")
    pred = kmodel.predict(x, verbose=0)
    pred = pred if isinstance(pred, list) else [pred]
    trn = kmodel(x, training=True)
    trn = [np.asarray(t) for t in (trn if isinstance(trn, (list, tuple)) else [trn])]
    after_fwd = keras_map.keras_weights_by_structural_name(kmodel, g)
    moving = {k: v for k, v in after_fwd.items() if k.endswith("moving_mean") or k.endswith("moving_variance")}
    # one optimiser step from the ORIGINAL moving statistics (so that the consumer starts from 'w/*' alone)
    keras_map.export_to_keras(weights, kmodel, g)
    kmodel.compile(optimizer=tf.keras.optimizers.Adam(learning_rate=1e-3), loss="mse")
    targets = [np.full(t.shape, 0.5, np.float32) for t in trn]
    loss = kmodel.train_on_batch(x, targets if len(targets) > 1 else targets[0])
    loss = float(loss[0] if isinstance(loss, (list, tuple)) else loss)
    w_after = keras_map.keras_weights_by_structural_name(kmodel, g)
    # ... and the trajectory: EXTRA_STEPS more steps on the same batch, then inference (a defect that leaves the first step exact -
    # weights that should never move, a stale operand copy - shows here)
    for _ in range(EXTRA_STEPS):
        kmodel.train_on_batch(x, targets if len(targets) > 1 else targets[0])
    pred_after = kmodel.predict(x, verbose=0)
    pred_after = pred_after if isinstance(pred_after, list) else [pred_after]
    meta = dict(meta, tensorflow=tf.__version__, keras=getattr(tf.keras, "__version__", "?"), output_names=list(kmodel.output_names),
                train_step=dict(optimizer="Adam", lr=1e-3, loss="mse", target=0.5, extra_steps=EXTRA_STEPS))
    arrays = {"meta": np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), "x": x, "loss": np.float32(loss)}
    arrays.update({"w/" + k.replace("/", "|"): v for k, v in weights.items()})
    arrays.update({"moving_after/" + k.replace("/", "|"): v for k, v in moving.items()})
    arrays.update({"w_after/" + k.replace("/", "|"): v for k, v in w_after.items()})
    arrays.update({f"predict/{i}": np.asarray(p, np.float32) for i, p in enumerate(pred)})
    arrays.update({f"train_fwd/{i}": np.asarray(p, np.float32) for i, p in enumerate(trn)})
    arrays.update({f"predict_after_steps/{i}": np.asarray(p, np.float32) for i, p in enumerate(pred_after)})
    np.savez_compressed(path, **arrays)
    print(f"{meta['case']}: {len(weights)} weights, {len(pred)} outputs, loss {loss:.6f} -> {path} ({os.path.getsize(path) / 1e3:.0f} kB)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--cases", default=",".join(list(CASES) + list(CASES_1D)))
    args = ap.parse_args()
    import tensorflow as tf                                    # noqa: F401  (the whole point of this script)
    segamd = importlib.import_module("segamd")
    from segamd import keras_map

    os.makedirs(args.out, exist_ok=True)
    for case in args.cases.split(","):
        tf.keras.backend.clear_session()
        tf.keras.utils.set_random_seed(0)
        path = os.path.join(args.out, f"tf_{case}.npz")
        if case in CASES:
            ref = importlib.import_module("models.unet_variants")     # the reference's 2D module, from PYTHONPATH
            dec, kw = CASES[case][:2]
            width = CASES[case][2] if len(CASES[case]) > 2 else WIDTH
            kmodel = ref.unet_model_builder(dec, H, W, width, DEPTH, train_mode="from_scratch", **kw).ResNet50()
            g = segamd.unet_model_builder(dec, H, W, width, DEPTH, train_mode="from_scratch", **kw).graph()
            x = np.random.default_rng(0).random((2, H, W, kw.get("num_channels", 3)), dtype=np.float32)
            meta = dict(case=case, ndim=2, decoder=dec, kwargs=kw, H=H, W=W, width=width, depth=DEPTH)
        else:
            mod, cls, method, pos, kw = CASES_1D[case]
            ref = importlib.import_module(mod)                         # the reference's 1D module (TensorFlow/1DCNN on PYTHONPATH)
            kmodel = getattr(getattr(ref, cls)(*pos, **kw), method)()
            builder = (segamd.BCDUNet if cls == "BCDUNet" else segamd.UNet1D)(*pos, **kw)
            g = getattr(builder, "graph_" + method)()
            x = np.random.default_rng(0).random((2, pos[0], pos[2]), dtype=np.float32)
            meta = dict(case=case, ndim=1, cls=cls, method=method, args=list(pos), kwargs=kw)
        capture(tf, kmodel, g, x, meta, keras_map, path)


if __name__ == "__main__":
    main()
