"""Keras checkpoint FILES <-> the structural parameter names of this engine, without TensorFlow.

The reference checkpoints with `ModelCheckpoint(... '.keras')` and resumes with `model.load_weights` (`2D/Train.py:375-379`,
`1D/Train.py`); a user who switches brings those files along.  `keras_map.py` needs a live `tf.keras.Model`; this module rebuilds
the duck-typed surface that walk uses - layer names, classes, `inbound_nodes`, `get_weights()` - from the FILE (the model's JSON
configuration plus its HDF5 weight groups, read by `h5lite.py`), so that

    seg_model = segamd.unet_model_builder(...).ResNet50()        # the constructor call the checkpoint was trained with
    seg_model.build(batch, dtype="f32")
    seg_model.load_keras_file("UNet_256_Fold_1.keras")           # every Conv / BN / ConvLSTM / Dense finds its structural name
    ... train / predict on the MI355X ...
    seg_model.save_keras_file("trained.keras", like="UNet_256_Fold_1.keras")      # same archive, new numbers: Keras loads it

File layouts understood (restated from the Keras documentation / the files' own structure; NO Keras installation was available to
produce a fixture - the test files are written by the real HDF5 library in the layout described here, `tests/golden/make_keras_files.py`,
and `tools/tf_capture_spec.py` saves `tf_<case>.keras` / `.h5` beside every fixture on a TensorFlow machine - the first real Keras
files this reader will see, held against the captured weights by `tests/test_tf_fixture.py`):

  * `.keras` archive (Keras "v3" saving format, TF >= 2.12): a zip of `config.json` (the functional model: `config.layers[*]` with
    `class_name`, `name`, `inbound_nodes`; `config.output_layers`), `metadata.json` and `model.weights.h5` whose groups are
    `layers/<key>/vars/<i>` - sub-objects of a layer one level down (`layers/conv_lstm2d/cell/vars/0..2`) - where `<key>` is the
    snake-cased CLASS name with a running counter in model.layers order (`conv2d`, `conv2d_1`, ...: Keras does not use the layers'
    own names there, on purpose); TF 2.12 called the top group `_layer_checkpoint_dependencies`.  A file keyed by layer names is
    read too, when it has no counted keys.
  * legacy whole-model HDF5 (`model.save('x.h5')`): root attribute `model_config` (the same JSON), group `model_weights` with the
    attribute `layer_names`, per layer a group with the attribute `weight_names` listing its datasets in `get_weights()` order.
  * legacy weights-only HDF5 (`model.save_weights('x.h5')`) and `.weights.h5`: the weight groups alone; the layer graph has to come
    with it (`config=` the JSON text of `model.to_json()` / a `config.json`), otherwise KerasFileError says what is missing.

`inbound_nodes` is read in both spellings: Keras 2's nested lists `[[["conv2d", 0, 0, {}], ...]]` and Keras 3's
`[{"args": [{"class_name": "__keras_tensor__", "config": {"keras_history": ["conv2d", 0, 0]}}], "kwargs": {}}]`."""
from __future__ import annotations

import io
import json
import re
import zipfile
import zlib
from typing import Dict, List, Optional

import numpy as np

from .h5lite import Dataset, Group, H5Error, H5File, H5Writer


class KerasFileError(ValueError):
    pass


def _snake(name: str) -> str:
    """Keras' auto-naming of a class (`Conv2DTranspose` -> `conv2d_transpose`, `LeakyReLU` -> `leaky_re_lu`)."""
    name = re.sub(r"\W+", "", name)
    name = re.sub("(.)([A-Z][a-z]+)", r"\1_\2", name)
    return re.sub("([a-z])([A-Z])", r"\1_\2", name).lower()


def _refs(x) -> List[str]:
    """Names of the layers a serialised call node takes its tensors from, in argument order (both Keras spellings)."""
    out: List[str] = []
    if isinstance(x, dict):
        if x.get("class_name") == "__keras_tensor__":
            out.append(x["config"]["keras_history"][0])
        else:
            for v in x.values():
                out += _refs(v)
    elif isinstance(x, (list, tuple)):
        if len(x) >= 3 and isinstance(x[0], str) and isinstance(x[1], int) and isinstance(x[2], int):
            out.append(x[0])
            if len(x) >= 4 and isinstance(x[3], dict):   # Keras 2 keeps further tensor arguments in the call's kwargs: `a * b` is the
                out += _refs(x[3])                       # TFOpLambda node ["a", 0, 0, {"y": ["b", 0, 0], "name": null}]
        else:
            for v in x:
                out += _refs(v)
    return out


class _CallNode:
    def __init__(self, inbound):
        self.inbound_layers = inbound if len(inbound) != 1 else inbound[0]


class _FileLayer:
    """One layer of the file: the surface keras_map.py walks."""

    def __init__(self, name: str):
        self.name = name
        self._inbound_nodes: List[_CallNode] = []
        self._w: List[np.ndarray] = []
        self._paths: List[str] = []                     # where each weight lives in the HDF5 file (for save_keras_file)

    def get_weights(self):
        return [w.copy() for w in self._w]

    def set_weights(self, ws):
        if len(ws) != len(self._w) or any(tuple(np.shape(a)) != b.shape for a, b in zip(ws, self._w)):
            raise KerasFileError(f"layer {self.name!r}: {[tuple(np.shape(a)) for a in ws]} does not fit {[b.shape for b in self._w]}")
        self._w = [np.asarray(a, dtype=b.dtype).reshape(b.shape) for a, b in zip(ws, self._w)]


_CLASSES: Dict[str, type] = {}


def _layer_of_class(cls: str, name: str) -> _FileLayer:
    if cls not in _CLASSES:
        _CLASSES[cls] = type(cls, (_FileLayer,), {})
    return _CLASSES[cls](name)


def _num_sorted(keys):
    return sorted(keys, key=lambda k: (0, int(k)) if k.isdigit() else (1, k))


def _vars_below(g: Group, prefix: str) -> List[str]:
    """Dataset paths of one saved object: its own `vars/0..n`, then those of its sub-objects (a ConvLSTM's `cell`) in name order."""
    out = []
    if "vars" in g and isinstance(g["vars"], Group):
        out += [f"{prefix}/vars/{k}" for k in _num_sorted(g["vars"].keys()) if isinstance(g["vars"][k], Dataset)]
    for k in g.keys():
        if k != "vars" and isinstance(g[k], Group):
            out += _vars_below(g[k], f"{prefix}/{k}")
    return out


def _text(v) -> str:
    if isinstance(v, bytes):
        return v.decode("utf-8")
    if isinstance(v, np.ndarray) and v.shape == ():
        v = v[()]
    return v.decode("utf-8") if isinstance(v, bytes) else str(v)


class KerasFileModel:
    """What a Keras checkpoint file says about its model: `layers` (name, class, inbound layers, weights), `output_layers`."""

    def __init__(self, path: str, config: Optional[str] = None):
        try:
            self._open(path, config)
        except KerasFileError:
            raise
        except H5Error as e:
            raise KerasFileError(f"{path}: {e}") from None
        except (zipfile.BadZipFile, zlib.error, json.JSONDecodeError, UnicodeDecodeError, KeyError, TypeError, AttributeError, IndexError, OSError, EOFError, NotImplementedError) as e:
            if isinstance(e, FileNotFoundError):
                raise
            raise KerasFileError(f"{path}: damaged or not a Keras checkpoint ({type(e).__name__}: {str(e)[:160]})") from None

    def _open(self, path: str, config: Optional[str]):
        self.path = path
        self.kind = None
        self._zip_members: Dict[str, bytes] = {}
        if zipfile.is_zipfile(path):
            with zipfile.ZipFile(path) as z:
                names = z.namelist()
                wname = next((n for n in names if n.endswith("model.weights.h5")), None)
                if wname is None:
                    raise KerasFileError(f"{path}: a zip archive without model.weights.h5 ({names}): not a .keras file")
                self._zip_members = {n: z.read(n) for n in names}
            self._weights_member = wname
            self.h5 = H5File(self._zip_members[wname])
            cfg_text = config or next((self._zip_members[n].decode("utf-8") for n in names if n.endswith("config.json")), None)
            self.kind = "keras"
        else:
            try:
                self.h5 = H5File(path)
            except H5Error as e:
                raise KerasFileError(f"{path}: neither a .keras archive nor an HDF5 file ({e})") from None
            cfg_text = config
            if cfg_text is None and "model_config" in self.h5.attrs:
                cfg_text = _text(self.h5.attrs["model_config"])
            self.kind = "h5"
        if cfg_text is None:
            raise KerasFileError(f"{path}: a weights-only file - the layer graph is not in it.  Save the whole model "
                                 "(`model.save('x.keras')`) or pass config=<the text of model.to_json()>")
        cfg = json.loads(cfg_text) if isinstance(cfg_text, str) else cfg_text
        self.config = cfg
        if cfg.get("class_name") not in ("Functional", "Model"):
            raise KerasFileError(f"{path}: a {cfg.get('class_name')} model (the reference builds functional models)")
        body = cfg["config"]
        by_name: Dict[str, _FileLayer] = {}
        self.layers: List[_FileLayer] = []
        for lc in body["layers"]:
            name = lc.get("name") or lc["config"]["name"]
            layer = _layer_of_class(lc["class_name"], name)
            layer.symbol = str((lc.get("config") or {}).get("function", ""))          # (TFOpLambda: 'math.multiply')
            by_name[name] = layer
            self.layers.append(layer)
        for lc, layer in zip(body["layers"], self.layers):
            for node in lc.get("inbound_nodes", []):
                try:
                    layer._inbound_nodes.append(_CallNode([by_name[r] for r in _refs(node)]))
                except KeyError as e:
                    raise KerasFileError(f"{path}: layer {layer.name!r} takes a tensor from {e}, which is not a layer of the model") from None
        try:
            self.output_layers = [by_name[r] for r in _refs(body["output_layers"])]
        except KeyError as e:
            raise KerasFileError(f"{path}: output layer {e} is not a layer of the model") from None
        self._attach_weights(body)

    # -- where the numbers are
    def _attach_weights(self, body):
        h5 = self.h5
        top = next((t for t in ("layers", "_layer_checkpoint_dependencies") if t in h5 and isinstance(h5[t], Group)), None)
        if top is not None:                                              # the .keras / .weights.h5 layout
            g = h5[top]
            keys = set(g.keys())
            # Keras addresses the layers of the `layers` container by the snake-cased CLASS name with a running counter in
            # model.layers order - deliberately not by `layer.name` ("names are usually autogenerated and thus not reproducible",
            # saving_lib).  In a fresh process the auto-names ARE conv2d, conv2d_1, ... - the same strings in another order - so
            # the counted reading must win whenever it fits; the by-name reading is only for files that have no counted keys.
            counted_of, used = {}, {}
            for layer in self.layers:
                base = _snake(layer.__class__.__name__)
                used[base] = used.get(base, -1) + 1
                counted_of[id(layer)] = base if used[base] == 0 else f"{base}_{used[base]}"
            weightless = ("InputLayer",)
            counted_ok = all(counted_of[id(l)] in keys for l in self.layers if l.__class__.__name__ not in weightless)
            by_layer_name = not counted_ok and all(l.name in keys for l in self.layers if l.__class__.__name__ not in weightless)
            for layer in self.layers:
                counted = counted_of[id(layer)]
                key = layer.name if by_layer_name else counted
                if key not in keys:
                    if layer.__class__.__name__ == "InputLayer":
                        continue
                    raise KerasFileError(f"{self.path}: no group {top}/{key} for layer {layer.name!r} ({layer.__class__.__name__}); "
                                         f"the file has {sorted(keys)[:6]} ...")
                layer._paths = _vars_below(g[key], f"{top}/{key}")
                layer._w = [np.asarray(h5[p].read()) for p in layer._paths]
            return
        root = h5["model_weights"] if "model_weights" in h5 else h5.root
        self._legacy_root = "model_weights/" if "model_weights" in h5 else ""
        if "layer_names" not in root.attrs:
            raise KerasFileError(f"{self.path}: no `layers` group and no `layer_names` attribute: not a Keras weights file "
                                 f"(top level: {h5.keys()[:8]})")
        names = [_text(n) for n in np.asarray(root.attrs["layer_names"]).reshape(-1)]
        by_name = {l.name: l for l in self.layers}
        for n in names:
            grp = root[n]
            wn = grp.attrs.get("weight_names")
            wnames = [] if wn is None else [_text(x) for x in np.asarray(wn).reshape(-1)]
            # (Keras splits long name lists over weight_names0, weight_names1, ...)
            i = 0
            while wn is None and f"weight_names{i}" in grp.attrs:
                wnames += [_text(x) for x in np.asarray(grp.attrs[f"weight_names{i}"]).reshape(-1)]
                i += 1
            if n not in by_name:
                if wnames:
                    raise KerasFileError(f"{self.path}: weights of layer {n!r}, which the model configuration does not have")
                continue
            layer = by_name[n]
            layer._paths = [f"{self._legacy_root}{n}/{w}" for w in wnames]
            layer._w = [np.asarray(h5[p].read()) for p in layer._paths]

    # -- writing the file back with other numbers
    def save(self, path: str):
        """This checkpoint with the layers' CURRENT weights (`set_weights`): every other byte of content - configuration, metadata,
        optimiser variables, attributes - as it was read."""
        new = {p: w for l in self.layers for p, w in zip(l._paths, l._w)}
        w = H5Writer()

        def copy(g: Group, prefix: str):
            for k, v in g.attrs.items():
                w.set_attr(prefix or "/", k, _attr_for_writer(v, f"{prefix}@{k}"))
            for k in g.keys():
                x = g[k]
                p = f"{prefix}/{k}" if prefix else k
                if isinstance(x, Group):
                    w.create_group(p)
                    copy(x, p)
                else:
                    w.create_dataset(p, new[p] if p in new else x.read())
                    for ak, av in x.attrs.items():
                        w.set_attr(p, ak, _attr_for_writer(av, f"{p}@{ak}"))

        copy(self.h5.root, "")
        data = w.tobytes()
        if self.kind == "keras":
            with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
                for n, b in self._zip_members.items():
                    z.writestr(n, data if n == self._weights_member else b)
        else:
            with open(path, "wb") as fh:
                fh.write(data)


def _attr_for_writer(v, what: str):
    if isinstance(v, np.ndarray) and v.dtype == object:
        return [str(x) for x in v.reshape(-1)]
    return v


def load_keras_file(path: str, graph, config: Optional[str] = None) -> Dict[str, np.ndarray]:
    """{structural parameter name: array in Keras layout} of a Keras checkpoint file, for `SegModel.set_weights`."""
    from .keras_map import keras_weights_by_structural_name
    return keras_weights_by_structural_name(KerasFileModel(path, config), graph)


def save_keras_file(weights: Dict[str, np.ndarray], path: str, like: str, graph, config: Optional[str] = None):
    """Write `weights` (structural names) as a Keras checkpoint shaped like the file `like` (the checkpoint the model came from, or
    any checkpoint of the same constructor call): Keras' `load_model` / `load_weights` takes it from there."""
    from .keras_map import export_to_keras
    m = KerasFileModel(like, config)
    export_to_keras(weights, m, graph)
    m.save(path)
