"""ORACLE (test infrastructure) - Keras default train_step restated: forward (training-mode BN) -> sum of
per-output losses with unit weights (`2D/Train.py:322-325`, no loss_weights) -> gradients -> Keras Adam
(`2D/utils/tf_optimizers.py:11`: beta (0.9, 0.999), epsilon 1e-7).  PARITY UNPINNED (keras_ref.py header)."""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Sequence

import torch

K_EPS = 1e-7   # keras.backend.epsilon()


def mse(pred, target):
    """tf.keras.losses.MeanSquaredError: mean over the last axis, then over every other axis."""
    return ((pred - target) ** 2).mean()


def binary_crossentropy(pred, target):
    """tf.keras.losses.BinaryCrossentropy(from_logits=False): probabilities clipped to [eps, 1-eps]."""
    p = pred.clamp(K_EPS, 1.0 - K_EPS)
    return -(target * torch.log(p) + (1.0 - target) * torch.log(1.0 - p)).mean()


def categorical_crossentropy(pred, target):
    """tf.keras.losses.CategoricalCrossentropy(from_logits=False): renormalise, clip, -sum(t log p), mean."""
    p = pred / pred.sum(-1, keepdim=True)
    p = p.clamp(K_EPS, 1.0 - K_EPS)
    return -(target * torch.log(p)).sum(-1).mean()


LOSSES = {"mse": mse, "bce": binary_crossentropy, "cce": categorical_crossentropy}


class KerasAdam:
    """Keras >= 2.11 Adam: alpha_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  w -= alpha_t * m / (sqrt(v) + eps)
    (epsilon OUTSIDE the bias-corrected root, unlike torch.optim.Adam)."""

    def __init__(self, params: Dict[str, torch.Tensor], lr=2e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.params = params
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, epsilon
        self.t = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor]):
        self.t += 1
        a = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for k, p in self.params.items():
            g = grads.get(k)
            if g is None:
                continue
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            p.sub_(a * self.m[k] / (self.v[k].sqrt() + self.eps))


def loss_and_grads(layers, forward: Callable[[], List[torch.Tensor]], targets: Sequence[torch.Tensor], loss: str):
    """One forward+backward.  Returns (total_loss, per-output losses, outputs, grads by parameter name)."""
    for p in layers.params.values():
        p.grad = None
    outs = forward()
    fn = LOSSES[loss]
    per = [fn(o, t) for o, t in zip(outs, targets)]
    total = sum(per)
    total.backward()
    grads = {k: v.grad.detach().clone() for k, v in layers.params.items() if v.grad is not None}
    return float(total.detach()), [float(p.detach()) for p in per], [o.detach() for o in outs], grads
