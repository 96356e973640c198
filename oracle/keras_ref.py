"""ORACLE (test infrastructure, not product code) - Keras-2.x layer semantics restated on torch-CPU fp32.

PARITY UNPINNED: the reference (`/root/reference`, pure tf.keras) cannot be imported in the build container
(no TensorFlow wheel, no network) and ships no tests or golden vectors, so nothing produced by the reference
itself pins this file.  It is pinned instead by (1) hand-computed known-answer tests with integer-valued
tensors (tests/test_oracle_primitives.py), (2) an independent numpy loop implementation of each primitive
(oracle/numpy_ref.py) and (3) the published Keras-2.x layer definitions summarised in SURVEY.md section 8(c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Every primitive cites the reference call site whose behaviour it restates.  Tensors are NHWC at the API
boundary; a 1D signal (N, L, C) is handled as (N, 1, L, C).
"""
from __future__ import annotations

import contextlib
import math
import warnings
import zlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3          # tf.keras.layers.BatchNormalization() default epsilon
BN_MOMENTUM = 0.99     # default momentum
LEAKY_ALPHA = 0.3      # Activation('LeakyReLU') resolves to the layer class LeakyReLU(alpha=0.3) in Keras 2.x


# ------------------------------------------------------------------------------------------ initialisers
def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def init_param(name: str, shape, init: str, fan: Tuple[int, int], seed: int) -> np.ndarray:
    """Keras initialisers (he_uniform for 2D Conv_Block `2D/models/unet_variants.py:7`, glorot_uniform default
    everywhere else, he_normal for ConvLSTM `:149`)."""
    r = _rng(seed, name)
    fan_in, fan_out = fan
    if init == "zeros":
        return np.zeros(shape, np.float32)
    if init == "ones":
        return np.ones(shape, np.float32)
    if init == "he_uniform":
        lim = math.sqrt(6.0 / fan_in)
        return r.uniform(-lim, lim, shape).astype(np.float32)
    if init == "glorot_uniform":
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return r.uniform(-lim, lim, shape).astype(np.float32)
    if init == "he_normal":
        std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
        v = r.standard_normal(shape)
        bad = np.abs(v) > 2.0
        while bad.any():                         # truncated normal: resample outside 2 sigma
            v[bad] = r.standard_normal(int(bad.sum()))
            bad = np.abs(v) > 2.0
        return (v * std).astype(np.float32)
    if init == "orthogonal":
        rows = int(np.prod(shape[:-1]))
        cols = int(shape[-1])
        a = r.standard_normal((max(rows, cols), min(rows, cols)))
        q, rr = np.linalg.qr(a)
        q = q * np.sign(np.diag(rr))
        if rows < cols:
            q = q.T
        return q.reshape(shape).astype(np.float32)
    if init == "unit_forget_bias":               # ConvLSTM bias: zeros with the f-gate slice = 1 (gate order i,f,c,o)
        b = np.zeros(shape, np.float32)
        F4 = shape[0] // 4
        b[F4:2 * F4] = 1.0
        return b
    raise ValueError(init)


# ------------------------------------------------------------------------------------------ primitives
def _nchw(x: torch.Tensor) -> torch.Tensor:      # NHWC tensor -> NCHW view (channels_last memory, no copy)
    return x.permute(0, 3, 1, 2)


def _nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 2, 3, 1)


def same_pad(n_in: int, k: int, s: int) -> Tuple[int, int]:
    """TensorFlow 'SAME' padding: out = ceil(in/s); total = max((out-1)*s + k - in, 0); before = total // 2."""
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    return total // 2, total - total // 2


_NARROW = 16


@contextlib.contextmanager
def _no_onednn():
    with warnings.catch_warnings():
        warnings.filterwarnings("ignore", message="TF32 acceleration on top of oneDNN")      # (what flags() says on a CPU-only torch)
        with torch.backends.mkldnn.flags(enabled=False):
            yield


class _NativeConv(torch.autograd.Function):
    """F.conv2d / F.conv_transpose2d with oneDNN switched off for the forward AND the backward call (torch picks the convolution
    backend when each of them runs, so a context manager around the forward alone is not enough), on contiguous NCHW tensors."""

    @staticmethod
    def forward(ctx, xin, w, bias, stride, transposed):
        ctx.stride, ctx.transposed = stride, transposed
        ctx.save_for_backward(xin, w, bias)
        with _no_onednn():
            return (F.conv_transpose2d if transposed else F.conv2d)(xin, w, bias, stride=stride)

    @staticmethod
    def backward(ctx, gy):
        xin, w, bias = ctx.saved_tensors
        args = [t.detach().requires_grad_(True) if t is not None else None for t in (xin, w, bias)]
        need = [a for a, n in zip(args, ctx.needs_input_grad[:3]) if a is not None and n]
        with torch.enable_grad(), _no_onednn():
            y = (F.conv_transpose2d if ctx.transposed else F.conv2d)(args[0], args[1], args[2], stride=ctx.stride)
            got = iter(torch.autograd.grad(y, need, gy.contiguous()) if need else ())
        out = [next(got) if (a is not None and n) else None for a, n in zip(args, ctx.needs_input_grad[:3])]
        return out[0], out[1], out[2], None, None


def _conv(xin, w, bias, stride, transposed=False):
    """F.conv2d / F.conv_transpose2d on the NHWC tensor's NCHW view (channels-last strides: oneDNN's fast path, ~1.6 x the NCHW
    kernels on wide layers).  NARROW layers (at most 16 input or output channels) take torch's native kernels on contiguous NCHW
    copies instead: with torch 2.10 / oneDNN on this image the BACKWARD of a convolution with a handful of channels corrupts the heap
    when it runs on several threads - "double free or corruption", "corrupted size vs. prev_size", segmentation faults or bus
    errors: in about one of six random attention-gate models on the channels-last path (theta / phi / psi convs at model widths
    4 - 8, unet_variants.py:69-78; tools/fuzz_models.py), and on the NCHW path still for a 1x1 / stride-2 convolution of ONE input
    channel (tools/fuzz_ops.py) - reproducible with the oracle alone, never with one thread or with oneDNN off.  Same arithmetic
    either way; the native kernels are 3 - 6 x slower on wide layers, which therefore stay on oneDNN."""
    cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    if min(cin, cout) <= _NARROW:
        return _NativeConv.apply(xin.contiguous(), w.contiguous(), bias, tuple(stride), transposed)
    return (F.conv_transpose2d if transposed else F.conv2d)(xin, w, bias, stride=stride)


def conv2d_same(x: torch.Tensor, kernel: torch.Tensor, bias: Optional[torch.Tensor], strides=(1, 1)) -> torch.Tensor:
    """tf.keras.layers.Conv2D(padding='same') - cross-correlation, kernel (kh,kw,Cin,Cout).
    Call sites: `2D/models/unet_variants.py:9,69,71,75,136,1106`; 1D `1D/unet_variants.py:55`."""
    kh, kw = kernel.shape[0], kernel.shape[1]
    pt, pb = same_pad(x.shape[1], kh, strides[0])
    pl, pr = same_pad(x.shape[2], kw, strides[1])
    xin = F.pad(_nchw(x), (pl, pr, pt, pb))
    y = _conv(xin, kernel.permute(3, 2, 0, 1).contiguous(), bias, strides)
    return _nhwc(y)


def conv2d_transpose_same(x: torch.Tensor, kernel: torch.Tensor, bias: Optional[torch.Tensor], strides=(2, 2)) -> torch.Tensor:
    """tf.keras.layers.Conv2DTranspose(padding='same'): out = in*stride, y[o] += x[i] * w[o - i*s + pad_before],
    kernel (kh,kw,Cout,Cin), no spatial flip.  Call sites: `2D/models/unet_variants.py:19`; 1D
    `1D/unet_variants.py:104` (k=2, s=2)."""
    kh, kw = kernel.shape[0], kernel.shape[1]
    sh, sw = strides
    H, W = x.shape[1], x.shape[2]
    full = _conv(_nchw(x), kernel.permute(3, 2, 0, 1).contiguous(), None, strides, transposed=True)   # (in-1)*s + k
    pbh = max(kh - sh, 0) // 2
    pbw = max(kw - sw, 0) // 2
    need_h, need_w = pbh + H * sh, pbw + W * sw
    if need_h > full.shape[2] or need_w > full.shape[3]:
        full = F.pad(full, (0, max(need_w - full.shape[3], 0), 0, max(need_h - full.shape[2], 0)))
    y = full[:, :, pbh:pbh + H * sh, pbw:pbw + W * sw]
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return _nhwc(y)


def batchnorm(x, gamma, beta, moving_mean, moving_var, training: bool, fused_4d: bool = True,
              eps: float = BN_EPS, momentum: float = BN_MOMENTUM):
    """tf.keras.layers.BatchNormalization() (axis -1).  Training: normalise with the biased batch variance;
    the moving variance is updated with the UNBIASED estimate on the fused 4-D path (2D models) and the
    biased one on the non-fused 3-D path (1D models).  Returns (y, new_moving_mean, new_moving_var).
    Call sites: `2D/models/unet_variants.py:11,70,72,76,96,99`."""
    if training:
        mean = x.mean(dim=(0, 1, 2))
        var = x.var(dim=(0, 1, 2), unbiased=False)
        n = x.shape[0] * x.shape[1] * x.shape[2]
        upd = var * (n / max(n - 1, 1)) if fused_4d else var
        with torch.no_grad():
            nm = moving_mean * momentum + mean.detach() * (1 - momentum)
            nv = moving_var * momentum + upd.detach() * (1 - momentum)
    else:
        mean, var, nm, nv = moving_mean, moving_var, moving_mean, moving_var
    y = (x - mean) * (gamma * torch.rsqrt(var + eps)) + beta
    return y, nm, nv


def hard_sigmoid(x):
    """Keras 2.x recurrent_activation default of ConvLSTM2D: clip(0.2 x + 0.5, 0, 1)."""
    return torch.clamp(0.2 * x + 0.5, 0.0, 1.0)


def activation(x, fn: str, alpha: float = 0.0):
    """Activation('ReLU'|'relu'|'LeakyReLU'|'sigmoid') and the softmax head.  `unet_variants.py:13,23,74,77`."""
    if fn == "relu":
        return torch.relu(x)
    if fn == "leaky_relu":
        return torch.where(x > 0, x, x * alpha)
    if fn == "sigmoid":
        return torch.sigmoid(x)
    if fn == "softmax":
        return torch.softmax(x, dim=-1)
    if fn in (None, "linear"):
        return x
    raise ValueError(fn)


def maxpool(x, size: Tuple[int, int]):
    """MaxPooling2D(pool_size) - strides = pool_size, 'valid'.  `unet_variants.py:790`."""
    return _nhwc(F.max_pool2d(_nchw(x), size, size))


def upsample(x, size: Tuple[int, int], interpolation: str):
    """UpSampling2D(size, 'bilinear') = tf.image.resize half-pixel centres, edge clamp, no antialias
    (`unet_variants.py:37`); UpSampling1D = nearest repeat (`1D/unet_variants.py:122`)."""
    if interpolation == "bilinear":
        return _nhwc(F.interpolate(_nchw(x), scale_factor=(float(size[0]), float(size[1])), mode="bilinear",
                                   align_corners=False))
    return x.repeat_interleave(size[0], dim=1).repeat_interleave(size[1], dim=2)


def convlstm_t1(x, kernel, bias):
    """ConvLSTM2D/1D over a length-1 sequence with zero initial state (`unet_variants.py:145-149`):
    z = conv(x, W) + b; (i, f, c, o) = split(z); h = hs(o) * tanh(hs(i) * tanh(c)).  The recurrent kernel
    and the f gate never touch the output."""
    z = conv2d_same(x, kernel, bias)
    zi, zf, zc, zo = torch.chunk(z, 4, dim=-1)
    return hard_sigmoid(zo) * torch.tanh(hard_sigmoid(zi) * torch.tanh(zc))


# ------------------------------------------------------------------------------------------ eager layer kit
class Layers:
    """Eager 'Keras-like' layer factory: creates parameters by structural name on first use and applies the
    primitive.  One instance == one model's weights."""

    def __init__(self, ndim: int = 2, seed: int = 0):
        self.ndim = ndim
        self.seed = seed
        self.params: Dict[str, torch.Tensor] = {}
        self.trainable: Dict[str, bool] = {}
        self.training = True
        self.update_moving = True

    # -- parameters
    def p(self, name, shape, init, fan=(0, 0), trainable=True) -> torch.Tensor:
        t = self.params.get(name)
        if t is None:
            t = torch.from_numpy(init_param(name, tuple(shape), init, fan, self.seed))
            t.requires_grad_(trainable)
            self.params[name] = t
            self.trainable[name] = trainable
        elif tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {tuple(t.shape)} != {tuple(shape)}")
        return t

    def _k2(self, k):
        if self.ndim == 1:
            return (1, int(k if not isinstance(k, (tuple, list)) else k[0]))
        return (int(k[0]), int(k[1])) if isinstance(k, (tuple, list)) else (int(k), int(k))

    # -- layers
    def conv(self, x, filters, k, name, strides=1, act=None, init="glorot_uniform"):
        kh, kw = self._k2(k)
        s = self._k2(strides)
        cin = x.shape[-1]
        fan = (kh * kw * cin, kh * kw * filters)
        w = self.p(f"{name}/kernel", (kh, kw, cin, filters), init, fan)
        b = self.p(f"{name}/bias", (filters,), "zeros")
        return activation(conv2d_same(x, w, b, s), act)

    def convT(self, x, filters, k, strides, name):
        kh, kw = self._k2(k)
        s = self._k2(strides)
        cin = x.shape[-1]
        fan = (kh * kw * filters, kh * kw * cin)
        w = self.p(f"{name}/kernel", (kh, kw, filters, cin), "glorot_uniform", fan)
        b = self.p(f"{name}/bias", (filters,), "zeros")
        return conv2d_transpose_same(x, w, b, s)

    def bn(self, x, name):
        C = x.shape[-1]
        g = self.p(f"{name}/gamma", (C,), "ones")
        b = self.p(f"{name}/beta", (C,), "zeros")
        mm = self.p(f"{name}/moving_mean", (C,), "zeros", trainable=False)
        mv = self.p(f"{name}/moving_variance", (C,), "ones", trainable=False)
        y, nm, nv = batchnorm(x, g, b, mm, mv, self.training, fused_4d=(self.ndim == 2))
        if self.training and self.update_moving:
            with torch.no_grad():
                mm.copy_(nm)
                mv.copy_(nv)
        return y

    def convlstm(self, x, filters, name):
        kh, kw = self._k2(3)
        cin = x.shape[-1]
        F4 = 4 * filters
        w = self.p(f"{name}/kernel", (kh, kw, cin, F4), "he_normal", (kh * kw * cin, kh * kw * F4))
        self.p(f"{name}/recurrent_kernel", (kh, kw, filters, F4), "orthogonal")
        b = self.p(f"{name}/bias", (F4,), "unit_forget_bias")
        return convlstm_t1(x, w, b)

    def dense_ae(self, x, features, out_channels, name):
        N, H, W, C = x.shape
        nin, nout = H * W * C, out_channels * H * W
        k1 = self.p(f"{name}/features/kernel", (nin, features), "glorot_uniform", (nin, features))
        b1 = self.p(f"{name}/features/bias", (features,), "zeros")
        k2 = self.p(f"{name}/expand/kernel", (features, nout), "glorot_uniform", (features, nout))
        b2 = self.p(f"{name}/expand/bias", (nout,), "zeros")
        z = x.reshape(N, nin) @ k1 + b1
        return (z @ k2 + b2).reshape(N, H, W, out_channels)

    def pool(self, x, size):
        return maxpool(x, self._k2(size))

    def up(self, x, size, interpolation):
        return upsample(x, self._k2(size), interpolation)

    # -- weights in/out (Keras layouts, numpy)
    def get_weights(self) -> Dict[str, np.ndarray]:
        return {k: v.detach().numpy().copy() for k, v in self.params.items()}

    def set_weights(self, w: Dict[str, np.ndarray]):
        for k, v in w.items():
            if k in self.params:
                with torch.no_grad():
                    self.params[k].copy_(torch.from_numpy(np.asarray(v, np.float32)))
            else:
                t = torch.from_numpy(np.asarray(v, np.float32).copy())
                tr = not (k.endswith("moving_mean") or k.endswith("moving_variance"))
                t.requires_grad_(tr)
                self.params[k] = t
                self.trainable[k] = tr

    def trainable_names(self) -> List[str]:
        return [k for k in self.params if self.trainable[k]]

    def count_params(self) -> int:
        return sum(int(v.numel()) for v in self.params.values())
