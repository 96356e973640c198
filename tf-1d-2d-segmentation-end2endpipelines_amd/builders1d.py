"""1D model-builder API of the reference (signals are H=1 images in the IR).

Mirrors `TensorFlow/1DCNN/Models/unet_variants.py:222-253` (class ``UNet`` with methods ``UNet``,
``UNetPP``, ``MultiResUNet``) and `TensorFlow/1DCNN/Models/BCDUNet.py:79-174` (class ``BCDUNet``).
Differences from 2D that matter for parity: two Conv_Blocks per level,
``Conv1DTranspose(k=2, s=2) -> BN -> ReLU``, nearest ``UpSampling1D``, glorot_uniform everywhere,
softmax/linear head chosen by ``problem_type``, default ``ds=1``.

Structural names: enc{i}/a, enc{i}/b (the two Conv_Blocks), mid/a, mid/b, dec{j}/..., X{j}{i}/...
"""
from __future__ import annotations

from typing import Dict, List, Tuple

from .graph import Graph, Sym


def cb1(g: Graph, x: Sym, filters: int, k: int, p: str) -> Sym:
    """Conv1D('same') -> BN -> ReLU  (1D unet_variants.py:53-60)."""
    y = g.conv(x, filters, k, f"{p}/conv", init="glorot_uniform")
    y = g.bn(y, f"{p}/bn")
    return g.act(y, "relu", f"{p}/relu")


def cb1x2(g, x, filters, k, p):
    return cb1(g, cb1(g, x, filters, k, f"{p}/a"), filters, k, f"{p}/b")


def tc1(g: Graph, x: Sym, filters: int, p: str) -> Sym:
    """Conv1DTranspose(2, strides 2) -> BN -> ReLU  (1D unet_variants.py:102-108)."""
    y = g.conv_transpose(x, filters, 2, 2, f"{p}/tconv")
    y = g.bn(y, f"{p}/tconv_bn")
    return g.act(y, "relu", f"{p}/tconv_relu")


def ag1(g: Graph, skip: Sym, gate: Sym, nf: int, mult: int, p: str) -> Sym:
    """1D Attention_Block (1D unet_variants.py:154-170): nearest upsample + trans_conv1D(.., 1, 1)."""
    th = g.bn(g.conv(skip, nf * mult, 1, f"{p}/theta/conv", strides=2), f"{p}/theta/bn")
    ph = g.bn(g.conv(gate, nf * mult, 1, f"{p}/phi/conv"), f"{p}/phi/bn")
    s = g.act(g.add([th, ph], f"{p}/add"), "relu", f"{p}/relu")
    ps = g.bn(g.conv(s, 1, 1, f"{p}/psi/conv"), f"{p}/psi/bn")
    ps = g.act(ps, "sigmoid", f"{p}/sigmoid")
    r1 = g.upsample(ps, 2, f"{p}/up_nearest", interpolation="nearest")
    r2 = tc1(g, ps, 1, f"{p}/up")
    return g.mul(skip, g.add([r1, r2], f"{p}/resample"), f"{p}/mul")


def mrb1(g: Graph, x: Sym, width: int, k: int, mult: int, alpha: float, p: str) -> Sym:
    """1D MultiResBlock (1D unet_variants.py:173-193): filters = int(alpha*w*f) * multiplier."""
    w = alpha * width
    a, b, c = int(w * 0.167) * mult, int(w * 0.333) * mult, int(w * 0.5) * mult
    sc = cb1(g, x, (int(w * 0.167) + int(w * 0.333) + int(w * 0.5)) * mult, 1, f"{p}/sc")
    c3 = cb1(g, x, a, k, f"{p}/c3")
    c5 = cb1(g, c3, b, k, f"{p}/c5")
    c7 = cb1(g, c5, c, k, f"{p}/c7")
    cat = g.bn(g.concat([c3, c5, c7], f"{p}/cat"), f"{p}/bn_cat")
    s = g.act(g.add([sc, cat], f"{p}/add"), "relu", f"{p}/relu")
    return g.bn(s, f"{p}/bn_out")


def respath1(g: Graph, x: Sym, length: int, filters: int, k: int, p: str) -> Sym:
    """1D ResPath (1D unet_variants.py:196-219)."""
    out = x
    for it in range(max(int(length), 1)):
        sc = cb1(g, out, filters, 1, f"{p}/{it}/sc")
        cv = cb1(g, out, filters, k, f"{p}/{it}/cv")
        out = g.bn(g.act(g.add([sc, cv], f"{p}/{it}/add"), "relu", f"{p}/{it}/relu"), f"{p}/{it}/bn")
    return out


def _head(g, y, problem_type, output_nums):
    if problem_type == "Classification":
        return g.conv(y, int(output_nums), 1, "out", activation="softmax")
    if problem_type == "Regression":
        return g.conv(y, int(output_nums), 1, "out", activation=None)
    raise ValueError("problem_type must be 'Classification' or 'Regression'")   # reference leaves outputs=[] and fails later


def _finish(g, out, levels: Dict[str, Sym], ds, d):
    if ds == 1:
        names = ["out"] + [f"level{k}" for k in range(1, d + 1)]
        g.set_outputs([out] + [levels[n] for n in names[1:]], names)
    else:
        g.set_outputs([out], ["out"])
    return g


def _lstm_filters(w, e):
    return int(w * (2.0 ** e))


class UNet:
    """Same signature as `TensorFlow/1DCNN/Models/unet_variants.py:223-224` (note the default ds=1)."""

    def __init__(self, length, model_depth, num_channel, model_width, kernel_size, problem_type="Regression",
                 output_nums=1, ds=1, ae=0, ag=0, lstm=0, alpha=1, t=2, feature_number=1024, is_transconv=True, q=3):
        self.length = length
        self.model_depth = model_depth
        self.num_channel = num_channel
        self.model_width = model_width
        self.kernel_size = kernel_size
        self.problem_type = problem_type
        self.output_nums = output_nums
        self.D_S = ds
        self.A_E = ae
        self.A_G = ag
        self.LSTM = lstm
        self.alpha = alpha
        self.feature_number = feature_number
        self.is_transconv = is_transconv
        self.t = t
        self.q = q

    # -- helpers -------------------------------------------------------------------------------
    def _check(self):
        if self.length == 0 or self.model_depth == 0 or self.model_width == 0 or self.num_channel == 0 \
                or self.kernel_size == 0:
            raise ValueError("Please Check the Values of the Input Parameters!")

    def _encoder(self, g):
        w, d, k = int(self.model_width), int(self.model_depth), int(self.kernel_size)
        x = g.input(1, int(self.length), int(self.num_channel))
        skips, pool = [], x
        for i in range(1, d + 1):
            c = cb1x2(g, pool, w * 2 ** (i - 1), k, f"enc{i}")
            pool = g.maxpool(c, 2, f"enc{i}/pool")
            skips.append(c)
        if self.A_E == 1:
            pool = g.dense_ae(pool, int(self.feature_number), w, "ae")
        mid = cb1x2(g, pool, w * 2 ** d, k, "mid")
        return skips, mid

    def _up(self, g, src, filters, p):
        if self.is_transconv:
            return tc1(g, src, filters, p)
        return g.upsample(src, 2, f"{p}/up", interpolation="nearest")

    def _wrap(self, g):
        from .model import SegModel
        return SegModel(g)

    # -- graphs --------------------------------------------------------------------------------
    def graph_UNet(self) -> Graph:
        """1D unet_variants.py:255-319."""
        self._check()
        g = Graph(ndim=1, name="UNet1D")
        w, d, k = int(self.model_width), int(self.model_depth), int(self.kernel_size)
        skips, de = self._encoder(g)
        levels: Dict[str, Sym] = {}
        for j in range(d):
            L = d - j - 1
            C = w * 2 ** L
            p = f"dec{j}"
            s = skips[L]
            if self.A_G == 1:
                s = ag1(g, s, de, w, 2 ** L, f"{p}/ag0")
            if self.D_S == 1:
                levels[f"level{d - j}"] = g.conv(de, 1, 1, f"level{d - j}")
            de = self._up(g, de, C, p)
            if self.LSTM == 1:
                de = g.convlstm(g.concat([s, de], f"{p}/lstm_cat"), _lstm_filters(w, L - 1), 3, f"{p}/lstm")
            else:
                de = g.concat([de, s], f"{p}/cat")
            de = cb1x2(g, de, C, k, p)
        return _finish(g, _head(g, de, self.problem_type, self.output_nums), levels, self.D_S, d)

    def graph_UNetPP(self) -> Graph:
        """1D unet_variants.py:544-645."""
        self._check()
        g = Graph(ndim=1, name="UNetPP1D")
        w, d, k = int(self.model_width), int(self.model_depth), int(self.kernel_size)
        skips, mid = self._encoder(g)
        skips = skips + [mid]
        levels: Dict[str, Sym] = {}
        if self.D_S == 1:
            levels[f"level{d}"] = g.conv(skips[0], 1, 1, f"level{d}")
        X: Dict[Tuple[int, int], Sym] = {}
        for i in range(1, d + 1):
            for j in range(0, d - i + 1):
                p = f"X{j}{i}"
                C = w * 2 ** j
                src = skips[j + 1] if i == 1 else X[(j + 1, i - 1)]
                dense = [X[(j, q)] for q in range(1, i)]
                s = skips[j]
                if self.A_G == 1:
                    gated = [ag1(g, t, src, w, 2 ** j, f"{p}/ag{n}") for n, t in enumerate(dense)]
                    s = ag1(g, s, src, w, 2 ** j, f"{p}/ag{len(dense)}")
                    dense = gated
                de = self._up(g, src, C, p)
                if self.LSTM == 1:
                    de = g.convlstm(g.concat([s, de] + dense, f"{p}/lstm_cat"), _lstm_filters(w, j - 1), 3, f"{p}/lstm")
                else:
                    de = g.concat([de] + dense + [s], f"{p}/cat")
                X[(j, i)] = cb1x2(g, de, C, k, p)
                if self.D_S == 1 and j == 0 and i < d:
                    levels[f"level{d - i}"] = g.conv(X[(0, i)], 1, 1, f"level{d - i}")
        return _finish(g, _head(g, X[(0, d)], self.problem_type, self.output_nums), levels, self.D_S, d)

    def graph_MultiResUNet(self) -> Graph:
        """1D unet_variants.py:836-899."""
        self._check()
        g = Graph(ndim=1, name="MultiResUNet1D")
        w, d, k = int(self.model_width), int(self.model_depth), int(self.kernel_size)
        x = g.input(1, int(self.length), int(self.num_channel))
        skips, pool = [], x
        for i in range(1, d + 1):
            m = mrb1(g, pool, w, k, 2 ** (i - 1), self.alpha, f"enc{i}/mrb")
            pool = g.maxpool(m, 2, f"enc{i}/pool")
            skips.append(respath1(g, m, d - i + 1, w * 2 ** (i - 1), k, f"enc{i}/respath"))
        if self.A_E == 1:
            pool = g.dense_ae(pool, int(self.feature_number), w, "ae")
        de = mrb1(g, pool, w, k, 2 ** d, self.alpha, "mid/mrb")
        levels: Dict[str, Sym] = {}
        for j in range(d):
            L = d - j - 1
            p = f"dec{j}"
            s = skips[L]
            if self.A_G == 1:
                s = ag1(g, s, de, w, 2 ** L, f"{p}/ag0")
            if self.D_S == 1:
                levels[f"level{d - j}"] = g.conv(de, 1, 1, f"level{d - j}")
            de = self._up(g, de, w * 2 ** L, p)
            if self.LSTM == 1:
                de = g.convlstm(g.concat([s, de], f"{p}/lstm_cat"), _lstm_filters(w, L - 1), 3, f"{p}/lstm")
            else:
                de = g.concat([de, s], f"{p}/cat")
            de = mrb1(g, de, w, k, 2 ** L, self.alpha, f"{p}/mrb")
        return _finish(g, _head(g, de, self.problem_type, self.output_nums), levels, self.D_S, d)

    # -- reference method names ----------------------------------------------------------------
    def UNet(self):
        return self._wrap(self.graph_UNet())

    def UNetPP(self):
        return self._wrap(self.graph_UNetPP())

    def MultiResUNet(self):
        return self._wrap(self.graph_MultiResUNet())


class BCDUNet:
    """Same signature as `TensorFlow/1DCNN/Models/BCDUNet.py:80-81`."""

    def __init__(self, length, model_depth, num_channel, model_width, kernel_size, problem_type="Regression",
                 output_nums=1, ds=1, ae=0, ag=0, lstm=0, dense_loop=1, feature_number=1024, is_transconv=True):
        self.length = length
        self.model_depth = model_depth
        self.num_channel = num_channel
        self.model_width = model_width
        self.kernel_size = kernel_size
        self.problem_type = problem_type
        self.output_nums = output_nums
        self.D_S = ds
        self.A_E = ae
        self.A_G = ag
        self.LSTM = lstm
        self.dense_loop = dense_loop
        self.feature_number = feature_number
        self.is_transconv = is_transconv

    def graph_BCDUNet(self) -> Graph:
        """BCDUNet.py:111-174.  With lstm=0 the reference never merges the skip (no `elif LSTM == 0`
        branch, :152-158); that behaviour is preserved, not repaired."""
        if self.length == 0 or self.model_depth == 0 or self.model_width == 0 or self.num_channel == 0 \
                or self.kernel_size == 0:
            raise ValueError("Please Check the Values of the Input Parameters!")
        g = Graph(ndim=1, name="BCDUNet1D")
        w, d, k = int(self.model_width), int(self.model_depth), int(self.kernel_size)
        x = g.input(1, int(self.length), int(self.num_channel))
        skips, pool = [], x
        for i in range(1, d + 1):
            c = cb1x2(g, pool, w * 2 ** (i - 1), k, f"enc{i}")
            pool = g.maxpool(c, 2, f"enc{i}/pool")
            skips.append(c)
        conv = pool
        for n in range(int(self.dense_loop) - 1):                     # dense_block, BCDUNet.py:70-76
            cb = cb1x2(g, conv, w * 2 ** d, k, f"dense{n}")
            conv = g.concat([conv, cb], f"dense{n}/cat")
        if self.A_E == 1:
            conv = g.dense_ae(conv, int(self.feature_number), w, "ae")
        de = cb1x2(g, conv, w * 2 ** d, k, "mid")
        levels: Dict[str, Sym] = {}
        for j in range(d):
            L = d - j - 1
            C = w * 2 ** L
            p = f"dec{j}"
            s = skips[L]
            if self.A_G == 1:
                s = ag1(g, s, de, w, 2 ** L, f"{p}/ag0")
            if self.D_S == 1:
                levels[f"level{d - j}"] = g.conv(de, 1, 1, f"level{d - j}")
            if self.is_transconv:
                de = tc1(g, de, C, p)
            else:
                de = g.upsample(de, 2, f"{p}/up", interpolation="nearest")
            if self.LSTM == 1:
                de = g.convlstm(g.concat([s, de], f"{p}/lstm_cat"), _lstm_filters(w, L - 1), 3, f"{p}/lstm")
            de = cb1x2(g, de, C, k, p)
        return _finish(g, _head(g, de, self.problem_type, self.output_nums), levels, self.D_S, d)

    def BCDUNet(self):
        from .model import SegModel
        return SegModel(self.graph_BCDUNet())
