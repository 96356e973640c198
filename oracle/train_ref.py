"""ORACLE (test infrastructure) - Keras default train_step restated: forward (training-mode BN) -> sum of
per-output losses with unit weights (`2D/Train.py:322-325`, no loss_weights; the 1D notebook passes some) ->
gradients -> Keras Adam (`2D/utils/tf_optimizers.py:11`: beta (0.9, 0.999), epsilon 1e-7).  The losses are the 16
names of `2D/utils/tf_losses.py:9-39`; the reference only names them - the formulas are Keras 2.x's (keras/losses.py,
keras/backend.py: a third-party dependency that is not in this image), restated here with torch ops and
differentiated by autograd.  PARITY UNPINNED (keras_ref.py header)."""
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


def mae(pred, target):
    """MeanAbsoluteError: mean |pred - target|."""
    return (pred - target).abs().mean()


def mape(pred, target):
    """MeanAbsolutePercentageError: 100 * mean |(target - pred) / max(|target|, eps)|."""
    return 100.0 * ((target - pred) / target.abs().clamp_min(K_EPS)).abs().mean()


def msle(pred, target):
    """MeanSquaredLogarithmicError: mean (log(max(pred, eps) + 1) - log(max(target, eps) + 1))^2."""
    return ((torch.log(pred.clamp_min(K_EPS) + 1.0) - torch.log(target.clamp_min(K_EPS) + 1.0)) ** 2).mean()


def huber(pred, target, delta=1.0):
    """Huber(delta=1): 0.5 e^2 where |e| <= delta, else delta |e| - 0.5 delta^2."""
    e = pred - target
    a = e.abs()
    return torch.where(a <= delta, 0.5 * e * e, delta * a - 0.5 * delta * delta).mean()


def logcosh(pred, target):
    """LogCosh: mean(x + softplus(-2x) - log 2), x = pred - target."""
    x = pred - target
    return (x + torch.nn.functional.softplus(-2.0 * x) - math.log(2.0)).mean()


def poisson(pred, target):
    """Poisson: mean(pred - target * log(pred + eps))."""
    return (pred - target * torch.log(pred + K_EPS)).mean()


def _pm1(target):
    """keras.losses._maybe_convert_labels: {0,1} labels become {-1,+1} when EVERY label is 0 or 1."""
    binary = bool(((target == 0) | (target == 1)).all())
    return 2.0 * target - 1.0 if binary else target


def hinge(pred, target):
    """Hinge: mean max(1 - t' pred, 0)."""
    return (1.0 - _pm1(target) * pred).clamp_min(0.0).mean()


def squared_hinge(pred, target):
    """SquaredHinge: mean max(1 - t' pred, 0)^2."""
    return ((1.0 - _pm1(target) * pred).clamp_min(0.0) ** 2).mean()


def binary_focal_crossentropy(pred, target, gamma=2.0):
    """BinaryFocalCrossentropy(gamma=2, from_logits=False): (1 - p_t)^gamma * bce with p_t on the unclipped
    prediction and the clipped prediction inside the cross-entropy (keras.backend.binary_focal_crossentropy)."""
    p_t = target * pred + (1.0 - target) * (1.0 - pred)
    pc = pred.clamp(K_EPS, 1.0 - K_EPS)
    bce = -(target * torch.log(pc) + (1.0 - target) * torch.log(1.0 - pc))
    return ((1.0 - p_t) ** gamma * bce).mean()


def kl_divergence(pred, target):
    """KLDivergence: both clipped to [eps, 1]; sum over the last axis of t log(t / p); mean."""
    t, p = target.clamp(K_EPS, 1.0), pred.clamp(K_EPS, 1.0)
    return (t * torch.log(t / p)).sum(-1).mean()


def categorical_hinge(pred, target):
    """CategoricalHinge: max(0, max((1 - t) p) - sum(t p) + 1) over the last axis; mean."""
    pos = (target * pred).sum(-1)
    neg = ((1.0 - target) * pred).max(-1).values
    return (neg - pos + 1.0).clamp_min(0.0).mean()


def cosine_similarity(pred, target):
    """CosineSimilarity(axis=-1): -sum(l2n(t) l2n(p)), l2n(x) = x * rsqrt(max(sum x^2, 1e-12)) (tf.nn.l2_normalize)."""
    def l2n(x):
        return x * torch.rsqrt((x * x).sum(-1, keepdim=True).clamp_min(1e-12))
    return -(l2n(target) * l2n(pred)).sum(-1).mean()


def sparse_categorical_crossentropy(pred, target):
    """SparseCategoricalCrossentropy(from_logits=False): probabilities clipped to [eps, 1 - eps], their logs fed to a
    softmax cross-entropy against the class index (target: one index per pixel, trailing axis of size 1 or none)."""
    q = pred.clamp(K_EPS, 1.0 - K_EPS)
    idx = target.reshape(pred.shape[:-1]).long()
    logq = torch.log(q)
    return -(logq.gather(-1, idx[..., None])[..., 0] - torch.logsumexp(logq, -1)).mean()


LOSSES = {"mse": mse, "bce": binary_crossentropy, "cce": categorical_crossentropy, "mae": mae, "mape": mape, "msle": msle,
          "huber": huber, "logcosh": logcosh, "poisson": poisson, "hinge": hinge, "squared_hinge": squared_hinge,
          "binary_focal": binary_focal_crossentropy, "kld": kl_divergence, "categorical_hinge": categorical_hinge,
          "cosine": cosine_similarity, "sparse_cce": sparse_categorical_crossentropy}


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


class KerasOptimizer:
    """The other seven optimisers `2D/utils/tf_optimizers.py:12-26` can select, with the hyper-parameters fixed there and
    the update rules of Keras >= 2.11 (keras/optimizers/{sgd,rmsprop,adagrad,adadelta,adamax,nadam,ftrl}.py - third-party,
    absent here; restated):
      SGD       w -= lr g
      RMSprop   v = rho v + (1 - rho) g^2 ; w -= lr g / sqrt(v + eps)                           rho 0.9
      Adagrad   a += g^2 (a0 = 0.1) ; w -= lr g / sqrt(a + eps)
      Adadelta  a = rho a + (1 - rho) g^2 ; d = -sqrt(D + eps) g / sqrt(a + eps) ;
                D = rho D + (1 - rho) d^2 ; w += lr d                                            rho 0.95
      Adamax    m += (g - m)(1 - b1) ; u = max(b2 u, |g|) ; w -= lr m / ((1 - b1^t)(u + eps))
      Nadam     u_k = b1 (1 - 0.5 * 0.96^k) ; m += (g - m)(1 - b1) ; v += (g^2 - v)(1 - b2) ;
                m^ = u_{t+1} m / (1 - prod u_{1..t+1}) + (1 - u_t) g / (1 - prod u_{1..t}) ;
                w -= lr m^ / (sqrt(v / (1 - b2^t)) + eps)
      FTRL      (lr_power -0.5, l1 = l2 = shrinkage = beta = 0, a0 = 0.1)  a' = a + g^2 ;
                z += g - (sqrt(a') - sqrt(a)) / lr * w ; w = -z / (sqrt(a') / lr) ; a = a'"""

    def __init__(self, name: str, params: Dict[str, torch.Tensor], lr=2e-4):
        if name not in ("SGD", "RMSprop", "Adagrad", "Adadelta", "Adamax", "Nadam", "FTRL"):
            raise ValueError(name)
        self.name, self.params, self.lr, self.t = name, params, lr, 0
        self.eps, self.b1, self.b2 = 1e-7, 0.9, 0.999
        init1 = 0.1 if name == "FTRL" else 0.0
        init2 = 0.1 if name == "Adagrad" else 0.0
        self.s1 = {k: torch.full_like(v, init1) for k, v in params.items()}
        self.s2 = {k: torch.full_like(v, init2) for k, v in params.items()}
        self.u_product = 1.0

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor]):
        self.t += 1
        t, lr, eps, b1, b2 = self.t, self.lr, self.eps, self.b1, self.b2
        if self.name == "Nadam":
            u_t = b1 * (1.0 - 0.5 * 0.96 ** t)
            u_t1 = b1 * (1.0 - 0.5 * 0.96 ** (t + 1))
            self.u_product *= u_t
            prod_t, prod_t1 = self.u_product, self.u_product * u_t1
        for k, w in self.params.items():
            g = grads.get(k)
            if g is None:
                continue
            s1, s2 = self.s1[k], self.s2[k]
            if self.name == "SGD":
                w.sub_(lr * g)
            elif self.name == "RMSprop":
                s2.mul_(0.9).add_(g * g, alpha=0.1)
                w.sub_(lr * g / torch.sqrt(s2 + eps))
            elif self.name == "Adagrad":
                s2.add_(g * g)
                w.sub_(lr * g / torch.sqrt(s2 + eps))
            elif self.name == "Adadelta":
                s1.mul_(0.95).add_(g * g, alpha=0.05)
                d = -torch.sqrt(s2 + eps) * g / torch.sqrt(s1 + eps)
                s2.mul_(0.95).add_(d * d, alpha=0.05)
                w.add_(lr * d)
            elif self.name == "Adamax":
                s1.add_((g - s1) * (1 - b1))
                s2.copy_(torch.maximum(b2 * s2, g.abs()))
                w.sub_(lr * s1 / ((1 - b1 ** t) * (s2 + eps)))
            elif self.name == "Nadam":
                s1.add_((g - s1) * (1 - b1))
                s2.add_((g * g - s2) * (1 - b2))
                m_hat = u_t1 * s1 / (1 - prod_t1) + (1 - u_t) * g / (1 - prod_t)
                w.sub_(lr * m_hat / (torch.sqrt(s2 / (1 - b2 ** t)) + eps))
            else:   # FTRL
                a1 = s1 + g * g
                s2.add_(g - (torch.sqrt(a1) - torch.sqrt(s1)) / lr * w)
                w.copy_(-s2 / (torch.sqrt(a1) / lr))
                s1.copy_(a1)


def loss_and_grads(layers, forward: Callable[[], List[torch.Tensor]], targets: Sequence[torch.Tensor], loss: str,
                   loss_weights: Sequence[float] = None):
    """One forward+backward.  Returns (total_loss, per-output losses, outputs, grads by parameter name)."""
    for p in layers.params.values():
        p.grad = None
    outs = forward()
    fn = LOSSES[loss]
    per = [fn(o, t) for o, t in zip(outs, targets)]
    total = sum(per) if loss_weights is None else sum(w * l for w, l in zip(loss_weights, per))
    total.backward()
    grads = {k: v.grad.detach().clone() for k, v in layers.params.items() if v.grad is not None}
    return float(total.detach()), [float(p.detach()) for p in per], [o.detach() for o in outs], grads
