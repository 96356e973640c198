"""Callers on either side of the hot path, host logic only.

f1  training harness of `2DCNN/Train.py:372-415`: Keras `fit` loop semantics with the three callbacks the reference
    installs - EarlyStopping(monitor, patience, mode), ModelCheckpoint(save_best_only=True), ReduceLROnPlateau(
    factor, patience, min_delta=1e-4, cooldown=0, min_lr=0) - over any model exposing train_step / evaluate-able
    forward (SegModel).
f2  inference / mask path of `2DCNN/Test.py:153-182`: predict, deep-supervision output 0, threshold rule
    `np.where(pred < threshold, 0, 1)` for one class, channel arg-max for several (the reference's multi-class
    branch references an undefined variable and cannot run), grey levels `mask * round(norm / classes)`.
f3  weight files keyed by structural parameter names (.npz, .safetensors); Keras checkpoint files: keras_file.py.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np


# ------------------------------------------------------------------------------------------------ f1: callbacks
def _better(mode: str, min_delta: float = 0.0) -> Callable[[float, float], bool]:
    if mode not in ("min", "max"):
        raise ValueError("mode must be 'min' or 'max'")
    if mode == "min":
        return lambda cur, best: cur < best - min_delta
    return lambda cur, best: cur > best + min_delta


class EarlyStopping:
    """keras.callbacks.EarlyStopping(monitor, patience, mode) with min_delta 0 (Train.py:372-374)."""

    def __init__(self, monitor="val_loss", patience=0, mode="min"):
        self.monitor, self.patience, self.better = monitor, patience, _better(mode)
        self.best = math.inf if mode == "min" else -math.inf
        self.wait = 0
        self.stopped_epoch = None

    def on_epoch_end(self, epoch: int, logs: Dict[str, float], state: "FitState"):
        cur = logs.get(self.monitor)
        if cur is None:
            return
        if self.better(cur, self.best):
            self.best, self.wait = cur, 0
            return
        self.wait += 1
        if self.wait >= self.patience and epoch > 0:
            self.stopped_epoch = epoch
            state.stop_training = True


class ModelCheckpoint:
    """keras.callbacks.ModelCheckpoint(path, monitor, save_best_only=True, mode) (Train.py:375-379)."""

    def __init__(self, path: str, monitor="val_loss", mode="min", save_best_only=True):
        self.path, self.monitor, self.better, self.save_best_only = path, monitor, _better(mode), save_best_only
        self.best = math.inf if mode == "min" else -math.inf
        self.saved_epochs: List[int] = []

    def on_epoch_end(self, epoch, logs, state):
        cur = logs.get(self.monitor)
        if self.save_best_only:
            if cur is None or not self.better(cur, self.best):
                return
            self.best = cur
        if _dist_rank() == 0:          # data parallel: the replicas are identical; one writer, the others wait for the file
            state.model.save_weights(self.path)
        _dist_barrier()
        self.saved_epochs.append(epoch)


class ReduceLROnPlateau:
    """keras.callbacks.ReduceLROnPlateau(monitor, factor, patience, mode, min_delta=1e-4, cooldown=0, min_lr=0)
    (Train.py:380-387)."""

    def __init__(self, monitor="val_loss", factor=0.1, patience=10, mode="min", min_delta=1e-4, cooldown=0, min_lr=0.0):
        if factor >= 1.0:
            raise ValueError("ReduceLROnPlateau does not support a factor >= 1.0.")
        self.monitor, self.factor, self.patience, self.cooldown, self.min_lr = monitor, factor, patience, cooldown, min_lr
        self.better = _better(mode, min_delta)
        self.best = math.inf if mode == "min" else -math.inf
        self.wait = 0
        self.cooldown_counter = 0

    def on_epoch_end(self, epoch, logs, state):
        logs["lr"] = state.model.lr
        cur = logs.get(self.monitor)
        if cur is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self.better(cur, self.best):
            self.best, self.wait = cur, 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                old = state.model.lr
                if old > self.min_lr:
                    state.model.lr = max(old * self.factor, self.min_lr)
                    self.cooldown_counter = self.cooldown
                    self.wait = 0


class FitState:
    def __init__(self, model):
        self.model = model
        self.stop_training = False


# ---- data parallel (dp.DataParallel installed on the model): every rank feeds its own shard, so the per-rank losses and
# metrics differ.  The callbacks must see the SAME numbers on every rank, or one rank stops / lowers its learning rate
# while the others sit in the next gradient all-reduce.
def _dist():
    try:
        import torch.distributed as dist
    except Exception:          # pragma: no cover
        return None
    return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None


def _dist_rank() -> int:
    d = _dist()
    return d.get_rank() if d is not None else 0


def _dist_barrier():
    d = _dist()
    if d is not None:
        d.barrier()


def _dist_device():
    import torch
    d = _dist()
    return torch.device("cuda", torch.cuda.current_device()) if (d is not None and d.get_backend() == "nccl") else torch.device("cpu")


def sync_logs(logs: Dict[str, float]) -> Dict[str, float]:
    """Mean over the ranks of every entry (same keys in the same order on every rank); identity without a process group."""
    d = _dist()
    if d is None or not logs:
        return logs
    import torch
    keys = sorted(logs)
    t = torch.tensor([float(logs[k]) for k in keys], dtype=torch.float64, device=_dist_device())
    d.all_reduce(t, op=d.ReduceOp.SUM)
    t /= d.get_world_size()
    return {k: float(v) for k, v in zip(keys, t.cpu().tolist())}


def sync_decisions(state: "FitState"):
    """After the callbacks: rank 0's stop flag and learning rate are everybody's (they were computed from identical logs,
    this makes the agreement unconditional - e.g. against a callback with rank-local state)."""
    d = _dist()
    if d is None:
        return
    import torch
    t = torch.tensor([1.0 if state.stop_training else 0.0, float(state.model.lr)], dtype=torch.float64, device=_dist_device())
    d.broadcast(t, src=0)
    state.stop_training = bool(t[0].item() != 0.0)
    state.model.lr = float(t[1].item())


def _as_batch(model, x, targets):
    """Loader output -> what the engine takes: numpy arrays become torch tensors, a Keras-style {output name: array}
    dictionary (data.prepare_train_dict) becomes a list in the order of the model's outputs, a single target a list."""
    import torch
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, np.float32))
    if isinstance(targets, dict):
        from .data import targets_for
        targets = targets_for(model, targets)
    elif not isinstance(targets, (list, tuple)):
        targets = [targets]
    targets = [torch.from_numpy(np.ascontiguousarray(t, np.float32)) if isinstance(t, np.ndarray) else t for t in targets]
    return x, targets


def _batches(src):
    """A loader is a callable returning an iterable, or an iterable itself (e.g. data.CustomDataGenerator)."""
    return src() if callable(src) else src


def _metric_set(model):
    """The device-side accumulator of the metrics the model was compiled with (None: no metrics / a duck-typed model)."""
    if getattr(model, "metric_names", None) and hasattr(model, "out_heads"):
        from .metrics import MetricSet
        return MetricSet(model, model.metric_names)
    return None


def evaluate(model, batches: Iterable, loss: Optional[str] = None, return_dict: bool = False):
    """Mean over batches of the summed per-output loss, inference-mode BN (Keras `evaluate`); with return_dict the
    compiled metrics as well ({'loss': ..., '<metric>': ...})."""
    tot, n = 0.0, 0
    ms = _metric_set(model) if return_dict else None
    for x, targets in _batches(batches):
        x, targets = _as_batch(model, x, targets)
        model.forward(x, training=False)
        tot += float(model.compute_loss(targets, loss).cpu())
        if ms is not None:
            ms.update(targets)
        n += 1
    if not return_dict:
        return tot / max(n, 1)
    logs = {"loss": tot / max(n, 1)}
    if ms is not None:
        logs.update(ms.result())
    return logs


def fit(model, train_batches, epochs: int, val_batches=None,
        callbacks: Sequence = (), verbose: bool = False, loss_scale_growth_interval: Optional[int] = None) -> Dict[str, List[float]]:
    """model.fit(...) of Train.py:394-415: per epoch one pass over the training loader (optimiser step per batch), then
    validation, then the callbacks in order.  A loader is a callable returning an iterable of (x, targets), or such an
    iterable itself (data.CustomDataGenerator: its on_epoch_end() reshuffles after every epoch, as Keras calls it).
    Returns `history.history`.  `loss_scale_growth_interval=` sets the model's attribute of that name (clean optimiser steps after
    which a backed-off loss scale doubles again; default: what the model has - `flags.LOSS_SCALE_GROWTH_INTERVAL` unless the caller
    changed it).  In graph mode (`graphed_train_step`) the replayed step has no host-side feedback: the scale is adjusted at the
    epoch-end drain only."""
    hist: Dict[str, List[float]] = {}
    state = FitState(model)
    ms = _metric_set(model)
    if loss_scale_growth_interval is not None:
        model.loss_scale_growth_interval = int(loss_scale_growth_interval)
    for epoch in range(int(epochs)):
        tot, n, skipped = 0.0, 0, 0
        if ms is not None:
            ms.reset()
        for x, targets in _batches(train_batches):
            x, targets = _as_batch(model, x, targets)
            tot += float(model.train_step(x, targets).cpu())
            if ms is not None:                  # Keras updates the metrics from the training-mode predictions of the step
                ms.update(targets)
            n += 1
            # (fp16-range backward overflow: the optimiser skipped and counted the affected elements and train_step backed the
            # loss scale off - SegModel.loss_scale_feedback, one step late from pinned memory, no stall of the side stream on which
            # the optimiser runs under the next forward pass; Keras would have put NaNs into the weights instead)
        if hasattr(model, "nonfinite_drain"):
            # what the one-step-late counter has not reported yet (the epoch's last step, or all of it when the loader had one
            # batch): this epoch's log and the loss-scale back-off see every step
            bad = model.nonfinite_drain()
            if bad:
                model.loss_scale_feedback(bad)
            skipped += model.take_skipped()
        if hasattr(model, "check_input_range"):
            model.check_input_range()            # raises on input elements the fp16 / split-pair storage cannot hold
        if hasattr(train_batches, "on_epoch_end"):
            train_batches.on_epoch_end()
        logs = {"loss": tot / max(n, 1)}
        if ms is not None:
            logs.update(ms.result())
        if val_batches is not None:
            logs.update({"val_" + k: v for k, v in evaluate(model, val_batches, return_dict=True).items()})
        if skipped or _dist() is not None:       # (every rank must bring the same keys to the reduction)
            logs["nonfinite_grads"] = float(skipped)
        logs = sync_logs(logs)                   # data parallel: every rank's callbacks see the mean over the shards
        for cb in callbacks:
            cb.on_epoch_end(epoch, logs, state)
        sync_decisions(state)
        for k, v in logs.items():
            hist.setdefault(k, []).append(v)
        if verbose:
            print(f"epoch {epoch + 1}/{epochs}: " + ", ".join(f"{k} {v:.6g}" for k, v in logs.items()))
        if state.stop_training:
            break
    return hist


# ------------------------------------------------------------------------------------------------ f2: masks
def masks_from_predictions(pred, threshold: float = 0.5, multiclass: str = "argmax") -> np.ndarray:
    """pred: model.predict output (a list when the model has deep-supervision outputs: Test.py:162-163 takes [0]);
    one class: np.where(pred < threshold, 0, 1) (Test.py:176); several classes: channel arg-max, or with
    multiclass='count' what Test.py:169-174 spells out (the number of channels at or above the threshold; that branch
    reads a variable it never initialises, so the reference itself cannot run it)."""
    if isinstance(pred, (list, tuple)):
        pred = pred[0]
    pred = np.asarray(pred)
    if pred.shape[-1] == 1:
        return np.where(pred < threshold, 0, 1).astype(np.int64)
    if multiclass == "count":
        return np.where(pred < threshold, 0, 1).sum(axis=-1, keepdims=True).astype(np.int64)
    if multiclass != "argmax":
        raise ValueError("multiclass must be 'argmax' or 'count'")
    return np.argmax(pred, axis=-1)[..., None].astype(np.int64)


def mask_grey_levels(mask: np.ndarray, normalizing_factor_msk: float = 255.0, class_number: int = 1) -> np.ndarray:
    """Test.py:181-183: `pred.astype(int8) * round(normalizing_factor_msk / class_number)` as an 8-bit 'L' image."""
    m = np.asarray(mask)[..., 0].astype(np.int8)
    return (m * np.round(normalizing_factor_msk / class_number)).astype(np.uint8)


def predict_directory(model, image_dir: str, mask_dir: str, save_dir: Optional[str] = None, img_size=None,
                      num_channels: int = 3, class_number: int = 1, threshold: float = 0.5,
                      normalizing_factor_img: float = 255.0, normalizing_factor_msk: float = 255.0,
                      patch_shape: Optional[Sequence[int]] = None, overlap_ratio: float = 0.5,
                      multiclass: str = "argmax") -> Dict[str, np.ndarray]:
    """The prediction loop of Test.py:118-200 for one fold: every image of image_dir (file order of os.listdir; the
    mask of the same position in mask_dir's listing) is opened with PIL, resized with nearest-neighbour resampling when
    its size differs, divided by its normalising factor, predicted ALONE (inference-mode BN; with patch_shape: cut
    into overlapping patches, predicted, laid back with unpatchify), reduced to its first deep-supervision output,
    thresholded, written to save_dir under the image's name as an 8-bit 'L' image of `mask * round(norm / classes)`,
    and accumulated into the fold's confusion matrices and per-image Dice scores.

    `model` may be the path of a checkpoint (`..._Fold_1.keras`): Test.py:114 loads the model from the file, so does this
    (keras_import.load_model).

    Returns {'names', 'cm' (classes x classes), 'cm_per_class' (classes x 2 x 2), 'dice' (per image)}."""
    if isinstance(model, (str, os.PathLike)):
        from .keras_import import load_model
        model = load_model(os.fspath(model))
    from PIL import Image
    from . import data as D
    H, W = img_size if img_size is not None else model.graph.nodes[model.graph.inputs[0]].shape[:2]
    imgs, msks = os.listdir(image_dir), os.listdir(mask_dir)
    if len(imgs) != len(msks):
        raise ValueError("Number of Images and Masks should be the same!")
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
    labels = np.arange(max(2, class_number + (1 if multiclass == "count" else 0)))
    cm = np.zeros((len(labels), len(labels)), np.int64)
    cmc = np.zeros((len(labels), 2, 2), np.int64)
    dices = []

    def first(p):
        return p[0] if isinstance(p, (list, tuple)) else p

    for name, mname in zip(imgs, msks):
        with Image.open(os.path.join(image_dir, name)) as im:
            if im.size != (W, H):
                im = im.resize((W, H), Image.Resampling.NEAREST)
            img = np.asarray(im, np.float64) / normalizing_factor_img
        if img.ndim == 2:
            img = img[:, :, None]
        if img.shape[2] != num_channels:
            raise ValueError(f"{name}: {img.shape[2]} channels, the model takes {num_channels}")
        with Image.open(os.path.join(mask_dir, mname)) as mk:
            if mk.size != (W, H):
                mk = mk.resize((W, H), Image.Resampling.NEAREST)
            msk = np.asarray(mk, np.float64) / normalizing_factor_msk
        if msk.ndim == 3:
            msk = msk[:, :, 0]
        msk = msk.astype(np.uint8)                                   # Test.py:124,147: a uint8 staging array
        if patch_shape is not None:
            pw, ph = int(patch_shape[0]), int(patch_shape[1])
            p, n = D.create_patches(img, (pw, ph), overlap_ratio)
            pred_p = first(model.predict(np.reshape(np.squeeze(p), (n, pw, ph, num_channels)).astype(np.float32)))
            if pred_p.shape[-1] != 1:
                raise ValueError("the patch path of the reference (Test.py:156-159) handles one-channel predictions only")
            mp, _ = D.create_patches(msk, (pw, ph), overlap_ratio)
            pred = D.unpatchify(np.reshape(np.squeeze(pred_p), mp.shape), msk.shape)[None, :, :, None]
        else:
            pred = first(model.predict(img[None].astype(np.float32)))
        m = masks_from_predictions(pred, threshold, multiclass)[0]
        if save_dir is not None:
            Image.fromarray(mask_grey_levels(m[None], normalizing_factor_msk, class_number)[0], mode="L").save(os.path.join(save_dir, name))
        yt, yp = msk.ravel().astype(np.int64), m.ravel()
        cm += D.confusion_matrix(yt, yp, labels)
        cmc += D.multilabel_confusion_matrix(yt, yp, labels)
        dices.append(D.dice(msk, m[..., 0]))
    return {"names": imgs, "cm": cm, "cm_per_class": cmc, "dice": np.asarray(dices)}


# ------------------------------------------------------------------------------------------------ f1: the history file
def save_history_h5(history: Dict[str, Sequence[float]], path: str):
    """Train.py:417-430: `model_history.history` as an HDF5 file with one dataset per key (`loss`, `val_loss`, `lr`, metric names),
    written without h5py (h5lite.H5Writer); h5py / the reference's plotting code read it as they read their own."""
    from .h5lite import H5Writer
    w = H5Writer()
    for k, v in history.items():
        w.create_dataset(k, np.asarray(v, np.float64))
    w.save(path)


def save_arrays_h5(path: str, arrays: Dict[str, np.ndarray]):
    """The 1D notebook's outcome files (`1D_Segmentation.ipynb`: `File.create_dataset('SM' / 'App' / 'App_Pred', data=...)`): named
    arrays, one dataset each (nested names make groups), float / integer dtypes as they are."""
    from .h5lite import H5Writer
    w = H5Writer()
    for k, v in arrays.items():
        w.create_dataset(k, np.asarray(v))
    w.save(path)


def load_arrays_h5(path: str) -> Dict[str, np.ndarray]:
    """... and `fl_Data = h5py.File(path, 'r'); fl_Data['SM']`: every dataset of the file by its path."""
    from .h5lite import H5File
    with H5File(path) as f:
        return {k: np.asarray(d.read()) for k, d in f.visit_datasets()}


def load_history_h5(path: str) -> Dict[str, List[float]]:
    from .h5lite import H5File
    with H5File(path) as f:
        return {k: [float(x) for x in np.asarray(f[k].read()).reshape(-1)] for k in f.keys()}


# ------------------------------------------------------------------------------------------------ f3: weight files
def save_weights_npz(weights: Dict[str, np.ndarray], path: str):
    np.savez(path if path.endswith(".npz") else path + ".npz", **{k.replace("/", "|"): v for k, v in weights.items()})


def save_weights_file(weights: Dict[str, np.ndarray], path: str, graph=None):
    """By extension: `.safetensors` (structural names as they are); `.keras` - what the reference's ModelCheckpoint is given
    (Train.py:375) - a Keras-layout archive of the builder graph and the weights (keras_import.save_model; needs `graph`); anything
    else the .npz form."""
    if path.endswith(".keras") and graph is not None:
        from .keras_import import save_model
        save_model(graph, weights, path)
    elif path.endswith(".safetensors"):
        from safetensors.numpy import save_file
        save_file({k: np.ascontiguousarray(v, np.float32) for k, v in weights.items()}, path, metadata={"format": "segamd structural names, Keras layouts"})
    else:
        save_weights_npz(weights, path)


def load_weights_file(path: str, graph=None, config: Optional[str] = None) -> Dict[str, np.ndarray]:
    """{structural name: array} of a weight file, by what the file IS: `.safetensors` / `.npz` of this package, or - given the
    builder graph - a Keras checkpoint of the reference (`.keras` archive, `.h5` / `.hdf5`: keras_file.py).  The reference's
    training script hands ONE path to ModelCheckpoint and to `load_weights` (Train.py:363-379, `..._Fold_1.keras`): what
    `save_weights(path)` of this package wrote for such a path is `path + '.npz'`, and is found under the path it was asked for."""
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file
        return dict(load_file(path))
    if path.endswith((".keras", ".h5", ".hdf5")):
        if not os.path.exists(path) and os.path.exists(path + ".npz"):
            return load_weights_npz(path + ".npz")
        if graph is None:
            raise ValueError(f"{path}: a Keras checkpoint needs the builder graph to be mapped onto (SegModel.load_weights)")
        from .keras_file import load_keras_file
        return load_keras_file(path, graph, config)
    return load_weights_npz(path)


def load_weights_npz(path: str) -> Dict[str, np.ndarray]:
    with np.load(path if path.endswith(".npz") else path + ".npz") as z:
        return {k.replace("|", "/"): z[k] for k in z.files}
