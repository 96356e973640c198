"""Callers on either side of the hot path, host logic only.

f1  training harness of `2DCNN/Train.py:372-415`: Keras `fit` loop semantics with the three callbacks the reference
    installs - EarlyStopping(monitor, patience, mode), ModelCheckpoint(save_best_only=True), ReduceLROnPlateau(
    factor, patience, min_delta=1e-4, cooldown=0, min_lr=0) - over any model exposing train_step / evaluate-able
    forward (SegModel).
f2  inference / mask path of `2DCNN/Test.py:153-182`: predict, deep-supervision output 0, threshold rule
    `np.where(pred < threshold, 0, 1)` for one class, channel arg-max for several (the reference's multi-class
    branch references an undefined variable and cannot run), grey levels `mask * round(norm / classes)`.
f3  weight files keyed by structural parameter names (.npz).
"""
from __future__ import annotations

import math
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
        state.model.save_weights(self.path)
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


def evaluate(model, batches: Iterable, loss: Optional[str] = None) -> float:
    """Mean over batches of the summed per-output loss, inference-mode BN (Keras `evaluate`)."""
    tot, n = 0.0, 0
    for x, targets in batches:
        model.forward(x, training=False)
        tot += float(model.compute_loss(targets if isinstance(targets, (list, tuple)) else [targets], loss).cpu())
        n += 1
    return tot / max(n, 1)


def fit(model, train_batches: Callable[[], Iterable], epochs: int, val_batches: Optional[Callable[[], Iterable]] = None,
        callbacks: Sequence = (), verbose: bool = False) -> Dict[str, List[float]]:
    """model.fit(...) of Train.py:394-415: per epoch one pass over train_batches() (optimiser step per batch), then
    validation, then the callbacks in order.  Returns `history.history`."""
    hist: Dict[str, List[float]] = {}
    state = FitState(model)
    for epoch in range(int(epochs)):
        tot, n = 0.0, 0
        for x, targets in train_batches():
            tot += float(model.train_step(x, targets).cpu())
            n += 1
        logs = {"loss": tot / max(n, 1)}
        if val_batches is not None:
            logs["val_loss"] = evaluate(model, val_batches())
        for cb in callbacks:
            cb.on_epoch_end(epoch, logs, state)
        for k, v in logs.items():
            hist.setdefault(k, []).append(v)
        if verbose:
            print(f"epoch {epoch + 1}/{epochs}: " + ", ".join(f"{k} {v:.6g}" for k, v in logs.items()))
        if state.stop_training:
            break
    return hist


# ------------------------------------------------------------------------------------------------ f2: masks
def masks_from_predictions(pred, threshold: float = 0.5) -> np.ndarray:
    """pred: model.predict output (a list when the model has deep-supervision outputs: Test.py:162-163 takes [0]);
    one class: np.where(pred < threshold, 0, 1) (Test.py:176); several classes: channel arg-max."""
    if isinstance(pred, (list, tuple)):
        pred = pred[0]
    pred = np.asarray(pred)
    if pred.shape[-1] == 1:
        return np.where(pred < threshold, 0, 1).astype(np.int64)
    return np.argmax(pred, axis=-1)[..., None].astype(np.int64)


def mask_grey_levels(mask: np.ndarray, normalizing_factor_msk: float = 255.0, class_number: int = 1) -> np.ndarray:
    """Test.py:181-183: `pred.astype(int8) * round(normalizing_factor_msk / class_number)` as an 8-bit 'L' image."""
    m = np.asarray(mask)[..., 0].astype(np.int8)
    return (m * np.round(normalizing_factor_msk / class_number)).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ f3: weight files
def save_weights_npz(weights: Dict[str, np.ndarray], path: str):
    np.savez(path if path.endswith(".npz") else path + ".npz", **{k.replace("/", "|"): v for k, v in weights.items()})


def load_weights_npz(path: str) -> Dict[str, np.ndarray]:
    with np.load(path if path.endswith(".npz") else path + ".npz") as z:
        return {k.replace("|", "/"): z[k] for k in z.files}
