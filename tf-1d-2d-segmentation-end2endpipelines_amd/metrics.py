"""Metrics of `2DCNN/utils/tf_metrics.py:18-96` evaluated on the device next to the loss (SURVEY.md section 8f, row f1:
the reference's callbacks monitor a metric - `Train_Configs.ini`: monitor_param = val_mean_squared_error).

Keras semantics restated (keras/metrics: a third-party dependency, absent here):
* mean-type metrics (MeanMetricWrapper over the loss function of the same name) keep total / count over the epoch;
  with equal batch sizes that is the mean of the per-batch means, which is what is accumulated here (one value-only
  launch of the loss kernel per output and batch, `seg_loss_fwd_bwd_w` with dpred = NULL).  RootMeanSquaredError is the
  root of the epoch's mean squared error; CosineSimilarity the positive cosine.
* BinaryAccuracy / Precision / Recall / TruePositives ... threshold with `pred > 0.5`, BinaryIoU with `pred >= 0.5`
  (its default target_class_ids (0, 1): mean of the two class IoUs, a class without any pixel left out);
  CategoricalAccuracy compares channel arg-maxes.  Counts are exact (uint64 on the device, `seg_binary_counts`,
  `seg_argmax_match`).
* log names: Keras' default metric names; a model with several outputs prefixes each with the output's name
  (`out_mean_squared_error`, `level1_mean_squared_error`, ...), `fit` adds `val_`.

Not available: AUC and the *At* family (threshold sweeps), IoU / MeanIoU / OneHot* (integer-label confusion matrices),
Accuracy (exact float equality), TopK*, Mean / Sum (no y_true/y_pred meaning).  Asking for one raises ValueError.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch

from ._ffi import LOSS

# reference name -> (Keras log name, loss kind, post-processing of the epoch mean)
MEAN_METRICS = {
    "MeanSquaredError": ("mean_squared_error", "mse", None),
    "MeanAbsoluteError": ("mean_absolute_error", "mae", None),
    "MeanAbsolutePercentageError": ("mean_absolute_percentage_error", "mape", None),
    "MeanSquaredLogarithmicError": ("mean_squared_logarithmic_error", "msle", None),
    "RootMeanSquaredError": ("root_mean_squared_error", "mse", "sqrt"),
    "BinaryCrossentropy": ("binary_crossentropy", "bce", None),
    "CategoricalCrossentropy": ("categorical_crossentropy", "cce", None),
    "SparseCategoricalCrossentropy": ("sparse_categorical_crossentropy", "sparse_cce", None),
    "Hinge": ("hinge", "hinge", None),
    "SquaredHinge": ("squared_hinge", "squared_hinge", None),
    "CategoricalHinge": ("categorical_hinge", "categorical_hinge", None),
    "CosineSimilarity": ("cosine_similarity", "cosine", "neg"),
    "KLDivergence": ("kullback_leibler_divergence", "kld", None),
    "LogCoshError": ("logcosh", "logcosh", None),
    "Poisson": ("poisson", "poisson", None),
}
# reference name -> Keras log name; evaluated from the 8 binary counts / the arg-max match count
COUNT_METRICS = {
    "BinaryAccuracy": "binary_accuracy", "BinaryIoU": "binary_io_u", "Precision": "precision", "Recall": "recall",
    "tf.keras.metrics.TruePositives": "true_positives", "tf.keras.metrics.TrueNegatives": "true_negatives",
    "tf.keras.metrics.FalsePositives": "false_positives", "tf.keras.metrics.FalseNegatives": "false_negatives",
    "CategoricalAccuracy": "categorical_accuracy", "SparseCategoricalAccuracy": "sparse_categorical_accuracy",
}
for _k in ("TruePositives", "TrueNegatives", "FalsePositives", "FalseNegatives"):      # also without the registry's odd prefix
    COUNT_METRICS[_k] = COUNT_METRICS["tf.keras.metrics." + _k]


def _div_no_nan(a: float, b: float) -> float:
    return a / b if b else 0.0


def log_names(metric_names: Sequence[str], output_names: Sequence[str]) -> List[str]:
    """The keys `fit` will log for these metrics (without the val_ prefix)."""
    base = [MEAN_METRICS[m][0] if m in MEAN_METRICS else COUNT_METRICS[m] for m in metric_names]
    if len(output_names) == 1:
        return base
    return [f"{o}_{b}" for o in output_names for b in base]


class MetricSet:
    """Accumulates the compiled metrics over the batches of one epoch; everything stays on the device until result()."""

    def __init__(self, model, metric_names: Sequence[str]):
        bad = [m for m in metric_names if m not in MEAN_METRICS and m not in COUNT_METRICS]
        if bad:
            raise ValueError(f"metric(s) {bad} are not available; choose from {sorted(MEAN_METRICS) + sorted(COUNT_METRICS)}")
        self.model, self.names = model, list(metric_names)
        self.mean_kinds = sorted({MEAN_METRICS[m][1] for m in self.names if m in MEAN_METRICS})
        self.need_counts = any(m in COUNT_METRICS and "Categorical" not in m for m in self.names)
        self.need_argmax = [m for m in self.names if m in ("CategoricalAccuracy", "SparseCategoricalAccuracy")]
        if len(self.need_argmax) > 1:
            raise ValueError("CategoricalAccuracy and SparseCategoricalAccuracy take different targets; compile one of them")
        nh = len(model.out_heads)
        dev = model.out_heads[0].y.device
        self.vals = torch.zeros((nh, max(1, len(self.mean_kinds))), dtype=torch.float32, device=dev)
        self.counts = torch.zeros((nh, 10), dtype=torch.int64, device=dev)       # 8 binary counts, arg-max matches, pixels
        self.flag = torch.ones(1, dtype=torch.int32, device=dev)
        self.batches = 0

    def reset(self):
        self.vals.zero_()
        self.counts.zero_()
        self.batches = 0

    def update(self, targets: Sequence[torch.Tensor]):
        """After a forward pass: the heads' outputs against these targets."""
        from .model import call, _p, _stream
        s = _stream()
        for i, (h, t) in enumerate(zip(self.model.out_heads, targets)):
            t = t.to(device=h.y.device, dtype=torch.float32).contiguous()
            npix = h.y.numel() // h.cout
            for j, kind in enumerate(self.mean_kinds):
                if t.numel() != (npix if kind == "sparse_cce" else h.y.numel()):
                    raise ValueError(f"metric target for {h.node.name}: {tuple(t.shape)} vs output {tuple(h.y.shape)}")
                flag = None
                if kind in ("hinge", "squared_hinge"):
                    call("seg_labels_binary", _p(t), t.numel(), _p(self.flag), s)
                    flag = _p(self.flag)
                call("seg_loss_fwd_bwd_w", LOSS[kind], _p(h.y), _p(t), npix, h.cout, 1.0, flag, _p(self.vals[i, j:j + 1]), None, s)
            if self.need_counts:
                call("seg_binary_counts", _p(h.y), _p(t), h.y.numel(), 0.5, _p(self.counts[i]), s)
            for m in self.need_argmax:
                sparse = 1 if m.startswith("Sparse") else 0
                call("seg_argmax_match", _p(h.y), _p(t), npix, h.cout, sparse, _p(self.counts[i, 8:9]), s)
                self.counts[i, 9] += npix
        self.batches += 1

    def result(self) -> Dict[str, float]:
        vals = (self.vals / max(self.batches, 1)).cpu().tolist()
        counts = self.counts.cpu().tolist()
        outs = self.model.output_names
        logs: Dict[str, float] = {}
        for i, o in enumerate(outs):
            pre = "" if len(outs) == 1 else o + "_"
            tp, fp, fn, tn, tp2, fp2, fn2, tn2, match, pix = counts[i]
            for m in self.names:
                if m in MEAN_METRICS:
                    name, kind, post = MEAN_METRICS[m]
                    v = vals[i][self.mean_kinds.index(kind)]
                    v = math.sqrt(max(v, 0.0)) if post == "sqrt" else (-v if post == "neg" else v)
                else:
                    name = COUNT_METRICS[m]
                    if m == "BinaryAccuracy":
                        v = _div_no_nan(tp + tn, tp + tn + fp + fn)
                    elif m == "BinaryIoU":
                        ious = []
                        if tn2 + fn2 + fp2:
                            ious.append(tn2 / (tn2 + fn2 + fp2))          # class 0
                        if tp2 + fn2 + fp2:
                            ious.append(tp2 / (tp2 + fn2 + fp2))          # class 1
                        v = sum(ious) / len(ious) if ious else 0.0
                    elif m == "Precision":
                        v = _div_no_nan(tp, tp + fp)
                    elif m == "Recall":
                        v = _div_no_nan(tp, tp + fn)
                    elif m.endswith("TruePositives"):
                        v = float(tp)
                    elif m.endswith("TrueNegatives"):
                        v = float(tn)
                    elif m.endswith("FalsePositives"):
                        v = float(fp)
                    elif m.endswith("FalseNegatives"):
                        v = float(fn)
                    else:                                               # (Sparse)CategoricalAccuracy
                        v = _div_no_nan(match, pix)
                logs[pre + name] = float(v)
        return logs
