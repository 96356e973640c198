#!/usr/bin/env python3
"""CPU-only fuzz of the checkpoint-file path over the builder API: random constructor calls (tools/fuzz_models.py's sampler - every decoder,
1-D and 2-D, ds / ae / ag / lstm, any width and kernel size) -> `save_weights('x.keras')` (keras_import.save_model: configuration +
model.weights.h5 through h5lite) -> the archive alone back to a graph and weights (`load_model`'s path) and onto the constructor call
(`load_weights`'s path).  The imported graph must compute what the builder's graph computes, bit for bit, in the oracle's interpreter.

    python tools/fuzz_archives.py <seed> <n>          # ~6 configurations a second
    python tools/fuzz_archives.py <seed> <n> --keras2 # the archive as TF 2.13 - 2.15 would spell it for the reference's code (keras_view)
    python tools/fuzz_archives.py <seed> <n> --h5     # the same model as a legacy whole-model .h5 (model_config attribute, weights by name)

Build-authored test tooling (uses the oracle's interpreter: not part of the product path)."""
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import segamd
from segamd import keras_file as KF, keras_import as KI, harness
from ir_interp import run_graph
import fuzz_models as FZ


def write_keras2_archive(g, one_d, path, rng_np, legacy=False):
    """The archive TF 2.13 - 2.15 would leave for builder graph `g`, in the reference's own spelling (tests/golden/make_keras_files.py
    `keras_view`: Activation('ReLU') as a serialised layer object, a * b as a TFOpLambda with its second operand in the call's kwargs,
    1x1 convs at padding='valid', Reshape + 5-D concatenate in front of a ConvLSTM, pairwise concatenates, auto-names, model.layers out
    of creation order) with class-counter weight keys - written with h5lite.  Returns {structural parameter name: array}."""
    import json
    import zipfile
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_keras_files as MK
    from segamd.h5lite import H5Writer
    from segamd.keras_file import _snake
    layers, outputs = MK.keras_view(g, rng_np, one_d, name_offset=0 if rng_np.random() < 0.5 else 17)      # (half of them with a fresh process's auto-names)
    planted = {}
    w = H5Writer()
    w.create_group("vars")
    used = {}
    cfg_layers = []
    for l in layers:
        base = _snake(l["cls"])
        used[base] = used.get(base, -1) + 1
        key = base if used[base] == 0 else f"{base}_{used[base]}"
        holder = f"layers/{key}/cell/vars" if l["cls"].startswith("ConvLSTM") else f"layers/{key}/vars"
        w.create_group(f"layers/{key}/vars")
        w.create_group(holder)
        for i, (pname, arr) in enumerate(l["weights"]):
            planted[pname] = arr
            a = arr[0] if (one_d and arr.ndim == 4 and arr.shape[0] == 1) else arr
            w.create_dataset(f"{holder}/{i}", a)
        if l["cls"] == "TFOpLambda" and l["inbound"]:
            a_, b_ = l["inbound"]
            node = [[a_, 0, 0, {"y": [b_, 0, 0], "name": None}]]
        else:
            node = [[[n_, 0, 0, {}] for n_ in l["inbound"]]] if l["inbound"] else []
        cfg_layers.append({"module": "keras.layers", "class_name": l["cls"], "config": dict(l["config"], trainable=True, dtype="float32"),
                           "registered_name": None, "name": l["name"], "inbound_nodes": node})
    config = {"module": "keras.src.engine.functional", "class_name": "Functional",
              "config": {"name": "model", "trainable": True, "layers": cfg_layers,
                         "input_layers": [[l["name"], 0, 0] for l in layers if l["cls"] == "InputLayer"], "output_layers": [[n_, 0, 0] for n_ in outputs]}}
    if legacy:
        # model.save('x.h5'): the configuration as the root attribute `model_config`, the weights under model_weights/<layer>/ by name
        names = {"ConvLSTM": ("kernel", "recurrent_kernel", "bias"), "Conv": ("kernel", "bias"),
                 "BatchNormalization": ("gamma", "beta", "moving_mean", "moving_variance"), "Dense": ("kernel", "bias")}
        w = H5Writer()
        w.set_attr("/", "model_config", json.dumps(config))
        w.set_attr("/", "keras_version", "2.11.0")
        w.set_attr("/", "backend", "tensorflow")
        w.create_group("model_weights")
        w.set_attr("model_weights", "layer_names", [l["name"].encode() for l in layers])
        for l in layers:
            w.create_group(f"model_weights/{l['name']}")
            fam = next((v for k, v in names.items() if l["cls"].startswith(k)), ())
            wn = [f"{l['name']}/{fam[i]}:0" for i in range(len(l["weights"]))]
            if wn:
                w.set_attr(f"model_weights/{l['name']}", "weight_names", [n_.encode() for n_ in wn])
            for n_, (pname, arr) in zip(wn, l["weights"]):
                w.create_dataset(f"model_weights/{l['name']}/{n_}", arr[0] if (one_d and arr.ndim == 4 and arr.shape[0] == 1) else arr)
        w.save(path)
        return planted
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
        z.writestr("metadata.json", json.dumps({"keras_version": "2.13.1"}))
        z.writestr("config.json", json.dumps(config))
        z.writestr("model.weights.h5", w.tobytes())
    return planted


def run(seed, n, verbose=True, keras2=False, legacy=False):
  rng = random.Random(seed)
  ok = refused = 0
  fails = []
  tmp = tempfile.mkdtemp()
  for i in range(n):
      two_d = rng.random() < 0.65
      call, make, xf, loss = (FZ.sample_2d if two_d else FZ.sample_1d)(rng)
      try:
          try:
              m = make()
          except Exception as e:
              text = f"{type(e).__name__}: {e}"
              if any(text.startswith(t) and frag in text for t, frag in FZ.EXPECTED):
                  refused += 1; continue
              raise
          g = m.graph
          wr = np.random.default_rng(i)
          w = {}
          for p in g.param_specs():
              v = wr.standard_normal(p.shape).astype(np.float32)
              w[p.name] = (np.abs(v) + 0.5 if p.name.endswith("moving_variance") else v / np.sqrt(np.prod(p.shape[:-1])) if len(p.shape) >= 2 else v).astype(np.float32)
          path = os.path.join(tmp, "m.h5" if legacy else "m.keras")
          if keras2 or legacy:
              w = write_keras2_archive(g, g.ndim == 1, path, wr, legacy)
          else:
              harness.save_weights_file(w, path, g)
          kfile = KF.KerasFileModel(path)
          gi = KI.graph_from_keras_config(kfile.config)
          wi = KI.weights_of_imported_graph(kfile, gi)
          back = harness.load_weights_file(path, g)
          assert set(back) == set(w) and all(np.array_equal(back[k], w[k]) for k in w), "constructor-call route"
          torch.manual_seed(i)
          x = xf()
          x = x[:2]
          t = lambda d: {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in d.items()}
          a = run_graph(gi, x, t(wi), training=True)
          b = run_graph(g, x, t(w), training=True)
          assert len(a) == len(b) and all(torch.equal(p, q) or (torch.isnan(p) == torch.isnan(q)).all() and torch.equal(torch.nan_to_num(p), torch.nan_to_num(q)) for p, q in zip(a, b)), "imported graph computes something else"
          ok += 1
      except Exception as e:
          fails.append((i, call, f"{type(e).__name__}: {str(e)[:200]}"))
          if verbose and len(fails) <= 5:
              print(f"[{i}] FAIL {call}\n     {type(e).__name__}: {str(e)[:300]}", flush=True)
  return ok, refused, fails


if __name__ == "__main__":
    torch.set_num_threads(8)
    seed, n = int(sys.argv[1]), int(sys.argv[2])
    t0 = time.time()
    keras2, legacy = "--keras2" in sys.argv, "--h5" in sys.argv
    ok, refused, fails = run(seed, n, keras2=keras2, legacy=legacy)
    print(f"==== archive round trip over random builder calls{' (legacy whole-model .h5, TF 2.x spelling)' if legacy else ' (TF 2.x spelling)' if keras2 else ''}: {ok} ok, {refused} refused as the reference refuses them, {len(fails)} failed, in {time.time() - t0:.0f} s (seed {seed})")
