#!/usr/bin/env python3
"""Golden-fixture capture for a machine that HAS TensorFlow 2.12-2.15 and a checkout of the reference repository
(Sakib1263/TF-1D-2D-Segmentation-End2EndPipelines).  Not runnable in the build container (no TensorFlow there): this is
the committed SPEC of the capture SURVEY.md section 8c asks for, and the only route by which the oracle's parity
status changes from "unpinned" to "pinned".  It copies nothing from the reference: it imports whatever
`models.unet_variants` is importable and calls the public constructors.

    PYTHONPATH=<reference>/TensorFlow/2DCNN:<this repo> python tools/tf_capture_spec.py --out tests/golden

For every case below it
  1. builds the Keras model through the reference's own `unet_model_builder(...)` call (d2 / w8 / 64x64: fixtures < 1 MB),
  2. re-initialises the BatchNormalization statistics to non-trivial values (seeded) so that inference-mode outputs depend
     on them,
  3. maps every Keras layer to the structural name of this package's graph for the same constructor call by walking
     `inbound_nodes` (keras_map.map_keras_model - never by list position or auto-name),
  4. runs `model.predict(x)` (inference-mode BN) and one `train_on_batch`-free training-mode forward
     (`model(x, training=True)`) on x = np.random.default_rng(0).random((2, H, W, C), float32),
  5. writes tests/golden/tf_<case>.npz: 'meta' (json: constructor arguments, TF / Keras versions), 'x', 'w/<structural name>'
     for every weight, 'predict/<i>' and 'train_fwd/<i>' for every output.
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
}
H = W = 64
WIDTH, DEPTH = 8, 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--cases", default=",".join(CASES))
    args = ap.parse_args()
    import tensorflow as tf                                    # noqa: F401  (the whole point of this script)
    ref = importlib.import_module("models.unet_variants")     # the reference's module, from PYTHONPATH
    segamd = importlib.import_module("segamd")
    from segamd import keras_map

    os.makedirs(args.out, exist_ok=True)
    for case in args.cases.split(","):
        dec, kw = CASES[case]
        tf.keras.backend.clear_session()
        tf.keras.utils.set_random_seed(0)
        kmodel = ref.unet_model_builder(dec, H, W, WIDTH, DEPTH, train_mode="from_scratch", **kw).ResNet50()
        g = segamd.unet_model_builder(dec, H, W, WIDTH, DEPTH, train_mode="from_scratch", **kw).graph()
        rng = np.random.default_rng(1)
        for layer in kmodel.layers:                            # non-trivial moving statistics, gamma / beta
            if layer.__class__.__name__ == "BatchNormalization":
                gm, bt, mu, var = layer.get_weights()
                layer.set_weights([rng.uniform(0.5, 1.5, gm.shape), rng.normal(0, 0.1, bt.shape),
                                   rng.normal(0, 0.1, mu.shape), rng.uniform(0.5, 1.5, var.shape)])
        weights = keras_map.keras_weights_by_structural_name(kmodel, g)
        missing = {p.name for p in g.param_specs()} - set(weights)
        assert not missing, f"{case}: parameters without a Keras counterpart: {sorted(missing)}"
        C = kw.get("num_channels", 3)
        x = np.random.default_rng(0).random((2, H, W, C), dtype=np.float32)
        pred = kmodel.predict(x, verbose=0)
        pred = pred if isinstance(pred, list) else [pred]
        trn = kmodel(x, training=True)
        trn = [np.asarray(t) for t in (trn if isinstance(trn, (list, tuple)) else [trn])]
        meta = dict(case=case, decoder=dec, kwargs=kw, H=H, W=W, width=WIDTH, depth=DEPTH, tensorflow=tf.__version__,
                    keras=getattr(tf.keras, "__version__", "?"), output_names=list(kmodel.output_names))
        arrays = {"meta": np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), "x": x}
        arrays.update({"w/" + k.replace("/", "|"): v for k, v in weights.items()})
        arrays.update({f"predict/{i}": np.asarray(p, np.float32) for i, p in enumerate(pred)})
        arrays.update({f"train_fwd/{i}": np.asarray(p, np.float32) for i, p in enumerate(trn)})
        path = os.path.join(args.out, f"tf_{case}.npz")
        np.savez_compressed(path, **arrays)
        print(f"{case}: {len(weights)} weights, {len(pred)} outputs -> {path} ({os.path.getsize(path) / 1e3:.0f} kB)")


if __name__ == "__main__":
    main()
