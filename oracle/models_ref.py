"""ORACLE (test infrastructure) - eager restatement of the reference's model wiring.  PARITY UNPINNED (see
keras_ref.py header).  Written independently of the product's graph builder (`builders2d.py`/`builders1d.py`)
so that wiring mistakes on either side show up as a numerical mismatch.

2D: `TensorFlow/2DCNN/models/unet_variants.py` - blocks :7-122, decoders UNet :125-154, UNetE :157-214,
UNetP :217-274, UNetPP :277-343, UNet3P :346-376, MultiResUNet :459-487, scratch encoder :750-792,
latent :966-974, builder from_scratch path :1089-1115.
1D: `TensorFlow/1DCNN/Models/unet_variants.py` :53-60, :102-124, :154-219, UNet :255-319, UNetPP :544-645,
MultiResUNet :836-899; `BCDUNet.py` :70-76, :111-174.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from .keras_ref import Layers, activation, LEAKY_ALPHA


# =============================================================================================== 2D blocks
def Conv_Block(L: Layers, x, width, k, p):                      # unet_variants.py:7-14
    x = L.conv(x, width, k, f"{p}/conv", init="he_uniform")
    x = L.bn(x, f"{p}/bn")
    return activation(x, "relu")


def trans_conv2D(L, x, width, p):                               # :17-24 (bn=False, LeakyReLU layer alpha 0.3)
    x = L.convT(x, width, (4, 4), (2, 2), f"{p}/tconv")
    return activation(x, "leaky_relu", LEAKY_ALPHA)


def Attention_Block(L, skip, gate, nf, mult, p):                # :67-82
    a = L.bn(L.conv(skip, nf * mult, 1, f"{p}/theta/conv", strides=2), f"{p}/theta/bn")
    b = L.bn(L.conv(gate, nf * mult, 1, f"{p}/phi/conv"), f"{p}/phi/bn")
    c = activation(a + b, "relu")
    c = L.bn(L.conv(c, 1, 1, f"{p}/psi/conv"), f"{p}/psi/bn")
    c = activation(c, "sigmoid")
    r = L.up(c, 2, "bilinear") + trans_conv2D(L, c, 1, f"{p}/up")
    return skip * r


def MultiResBlock(L, x, width, alpha, p):                       # :85-100
    w = alpha * width
    a, b, c = int(w * 0.167), int(w * 0.333), int(w * 0.5)
    sc = Conv_Block(L, x, a + b + c, 1, f"{p}/sc")
    c3 = Conv_Block(L, x, a, 3, f"{p}/c3")
    c5 = Conv_Block(L, c3, b, 3, f"{p}/c5")
    c7 = Conv_Block(L, c5, c, 3, f"{p}/c7")
    out = L.bn(torch.cat([c3, c5, c7], -1), f"{p}/bn_cat")
    out = activation(sc + out, "relu")
    return L.bn(out, f"{p}/bn_out")


def ResPath(L, x, depth, width, p):                             # :103-122
    def step(t, k):
        sc = Conv_Block(L, t, width, 1, f"{p}/{k}/sc")
        o = Conv_Block(L, t, width, 3, f"{p}/{k}/cv")
        return L.bn(activation(sc + o, "relu"), f"{p}/{k}/bn")
    out = step(x, 0)
    for k in range(1, depth):
        out = step(out, k)
    return out


def _named_cb(L, x, width, k):
    x = L.conv(x, width, 3, f"latent/conv{k}", init="he_uniform")
    return activation(L.bn(x, f"latent/bn{k}"), "relu")


def dense_block(L, x, width, loops):                            # :51-56
    x = _named_cb(L, x, width, 0)
    for k in range(loops):
        x = x + _named_cb(L, x, width, k + 1)
    return x


def _merge_lstm(L, tensors, filters, p):                        # :145-149 / :327-333 (T = 1, channel concat)
    return L.convlstm(torch.cat(tensors, -1), filters, f"{p}/lstm")


# =============================================================================================== 2D decoders
def dec_UNet(L, convs, w, d, DS, AG, LSTM, transconv, multires=False, alpha=1.0):     # :125-154, :459-487
    levels = {}
    deconv = convs[-1]
    for j in range(d):
        lvl = d - j - 1
        p = f"dec{j}"
        skip = convs[lvl]
        if AG == 1:
            skip = Attention_Block(L, convs[lvl], deconv, w, 2 ** lvl, f"{p}/ag0")
        if DS == 1:
            levels[f"level{d - j}"] = L.conv(deconv, 1, 1, f"level{d - j}")
        deconv = trans_conv2D(L, deconv, w * 2 ** lvl, p) if transconv else L.up(deconv, 2, "bilinear")
        if LSTM == 1:
            if multires:
                raise NameError("name 'length' is not defined")
            deconv = _merge_lstm(L, [skip, deconv], int(w * (2.0 ** (lvl - 1))), p)
        else:
            deconv = torch.cat([deconv, skip], -1)
        deconv = MultiResBlock(L, deconv, w * 2 ** lvl, alpha, f"{p}/mrb") if multires else \
            Conv_Block(L, deconv, w * 2 ** lvl, 3, p)
    return deconv, levels


def _nested(L, convs, w, d, DS, AG, LSTM, transconv, kind):     # UNetE :157-214 / UNetP :217-274 / UNetPP :277-343
    levels = {}
    if DS == 1:
        levels[f"level{d}"] = L.conv(convs[0], 1, 1, f"level{d}")
    node: Dict[str, torch.Tensor] = {}
    for i in range(1, d + 1):
        for j in range(0, d - i + 1):
            p = f"X{j}{i}"
            below = convs[j + 1] if i == 1 else node[f"{j + 1}{i - 1}"]
            extra: List[torch.Tensor] = []
            nag = 0
            if i == 1:
                skip = convs[j]
            elif kind == "UNetE":
                skip = convs[j]
            elif kind == "UNetP":
                skip = node[f"{j}{i - 1}"]
            else:
                tot = node[f"{j}1"]
                if AG == 1:
                    tot = Attention_Block(L, tot, below, w, 2 ** j, f"{p}/ag{nag}")
                    nag += 1
                extra.append(tot)
                for k in range(2, i):
                    t = node[f"{j}{k}"]
                    if AG == 1:
                        t = Attention_Block(L, t, below, w, 2 ** j, f"{p}/ag{nag}")
                        nag += 1
                    extra.append(t)
                skip = convs[j]
            if AG == 1:
                skip = Attention_Block(L, skip, below, w, 2 ** j, f"{p}/ag{nag}")
            up = trans_conv2D(L, below, w * 2 ** j, p) if transconv else L.up(below, 2, "bilinear")
            if LSTM == 1:
                up = _merge_lstm(L, [skip, up] + extra, int(w * (2.0 ** (j - 1))), p)
            else:
                up = torch.cat([up] + extra + [skip], -1)
            node[f"{j}{i}"] = Conv_Block(L, up, w * 2 ** j, 3, p)
            if DS == 1 and j == 0 and i < d:
                levels[f"level{d - i}"] = L.conv(node[f"0{i}"], 1, 1, f"level{d - i}")
    return node[f"0{d}"], levels


def dec_UNet3P(L, convs, w, d, DS):                             # :346-376
    levels = {}
    deconv = convs[-1]
    done = {}
    for j in range(d):
        p = f"dec{j}"
        lvl = d - j - 1
        allsk = Conv_Block(L, convs[lvl], w, 3, f"{p}/same")
        for k in range(0, lvl):
            f = 2 ** (lvl - k)
            s = Conv_Block(L, L.pool(convs[k], f), w, 3, f"{p}/down{k}")
            allsk = torch.cat([allsk, s], -1)
        t = Conv_Block(L, deconv, w, 3, f"{p}/below")
        t = activation(L.up(t, 2, "bilinear"), "sigmoid")
        tot = torch.cat([allsk, t], -1)
        for m in range(0, j):
            t = Conv_Block(L, done[m], w, 3, f"{p}/deep{m}")
            t = activation(L.up(t, 2 ** (j - m), "bilinear"), "sigmoid")
            tot = torch.cat([tot, t], -1)
        deconv = Conv_Block(L, tot, w * (d + 1), 3, p)
        done[j] = deconv
        if DS == 1:
            levels[f"level{d - j}"] = L.conv(deconv, 1, 1, f"level{d - j}", strides=2)
    return deconv, levels


def unet2d_forward(L: Layers, x, decoder_name, model_width, model_depth, output_nums=1, ds=0, ae=0, ag=0, lstm=0,
                   dense_loop=1, feature_number=1024, is_transconv=True, alpha=1.0, final_activation="sigmoid"):
    """from_scratch path, unet_variants.py:1089-1115.  x: (N,H,W,C) float32.  Returns [out, level1..level_d]
    (or [out])."""
    w, d = model_width, model_depth
    multires = decoder_name == "MultiResUNet"
    convs = []
    pool = x
    conv = x
    for i in range(1, d + 2):                                   # encoder_block_scratch :750-792
        if multires:
            conv = MultiResBlock(L, pool, w * 2 ** (i - 1), alpha, f"enc{i}/mrb")
            pool = L.pool(conv, 2)
            if i <= d:                                          # level d+1's ResPath is pruned by Keras (:757, :1100)
                convs.append(ResPath(L, conv, d - i + 1, w * 2 ** (i - 1), f"enc{i}/respath"))
        else:
            conv = Conv_Block(L, pool, w * 2 ** (i - 1), 3, f"enc{i}")
            pool = L.pool(conv, 2)
            convs.append(conv)
    if multires:                                                # latent_layer :966-974
        lat = MultiResBlock(L, conv, w * 2 ** d, alpha, "latent/mrb")
    else:
        lat = dense_block(L, conv, w * 2 ** d, dense_loop)
    if ae == 1:
        lat = L.dense_ae(lat, feature_number, w * 2 ** d, "ae")
    convs = convs[:d] + [lat]
    if decoder_name == "UNet":
        y, levels = dec_UNet(L, convs, w, d, ds, ag, lstm, is_transconv)
    elif decoder_name == "MultiResUNet":
        y, levels = dec_UNet(L, convs, w, d, ds, ag, lstm, is_transconv, True, alpha)
    elif decoder_name in ("UNetE", "UNetP", "UNetPP"):
        y, levels = _nested(L, convs, w, d, ds, ag, lstm, is_transconv, decoder_name)
    elif decoder_name == "UNet3P":
        y, levels = dec_UNet3P(L, convs, w, d, ds)
    else:
        raise NotImplementedError(decoder_name)
    out = L.conv(y, output_nums, 1, "out", act=final_activation)
    if ds == 1:
        return [out] + [levels[f"level{k}"] for k in range(1, d + 1)]
    return [out]


# =============================================================================================== 1D
def CB1(L, x, filters, k, p):                                   # 1D unet_variants.py:53-60
    x = L.conv(x, filters, k, f"{p}/conv")
    return activation(L.bn(x, f"{p}/bn"), "relu")


def CB1x2(L, x, filters, k, p):
    return CB1(L, CB1(L, x, filters, k, f"{p}/a"), filters, k, f"{p}/b")


def TC1(L, x, filters, p):                                      # :102-108
    x = L.convT(x, filters, 2, 2, f"{p}/tconv")
    return activation(L.bn(x, f"{p}/tconv_bn"), "relu")


def AG1(L, skip, gate, nf, mult, p):                            # :154-170
    a = L.bn(L.conv(skip, nf * mult, 1, f"{p}/theta/conv", strides=2), f"{p}/theta/bn")
    b = L.bn(L.conv(gate, nf * mult, 1, f"{p}/phi/conv"), f"{p}/phi/bn")
    c = activation(a + b, "relu")
    c = activation(L.bn(L.conv(c, 1, 1, f"{p}/psi/conv"), f"{p}/psi/bn"), "sigmoid")
    return skip * (L.up(c, 2, "nearest") + TC1(L, c, 1, f"{p}/up"))


def MRB1(L, x, width, k, mult, alpha, p):
    w = alpha * width
    fa, fb, fc = int(w * 0.167), int(w * 0.333), int(w * 0.5)
    sc = CB1(L, x, (fa + fb + fc) * mult, 1, f"{p}/sc")
    c3 = CB1(L, x, fa * mult, k, f"{p}/c3")
    c5 = CB1(L, c3, fb * mult, k, f"{p}/c5")
    c7 = CB1(L, c5, fc * mult, k, f"{p}/c7")
    out = L.bn(torch.cat([c3, c5, c7], -1), f"{p}/bn_cat")
    return L.bn(activation(sc + out, "relu"), f"{p}/bn_out")


def ResPath1(L, x, depth, filters, k, p):
    def step(t, n):
        return L.bn(activation(CB1(L, t, filters, 1, f"{p}/{n}/sc") + CB1(L, t, filters, k, f"{p}/{n}/cv"), "relu"),
                    f"{p}/{n}/bn")
    out = step(x, 0)
    for n in range(1, depth):
        out = step(out, n)
    return out


def _head1(L, y, problem_type, output_nums):
    return L.conv(y, output_nums, 1, "out", act="softmax" if problem_type == "Classification" else None)


def _ret(out, levels, ds, d):
    return [out] + [levels[f"level{k}"] for k in range(1, d + 1)] if ds == 1 else [out]


def unet1d_forward(L: Layers, x, method, model_depth, model_width, kernel_size, problem_type="Regression",
                   output_nums=1, ds=1, ae=0, ag=0, lstm=0, alpha=1, feature_number=1024, is_transconv=True):
    """x: (N, L, C).  method in {'UNet','UNetPP','MultiResUNet'}."""
    x = x.unsqueeze(1)
    w, d, k = model_width, model_depth, kernel_size
    up = (lambda t, f, p: TC1(L, t, f, p)) if is_transconv else (lambda t, f, p: L.up(t, 2, "nearest"))
    levels = {}
    if method in ("UNet", "UNetPP"):
        convs, pool = [], x
        for i in range(1, d + 1):
            c = CB1x2(L, pool, w * 2 ** (i - 1), k, f"enc{i}")
            pool = L.pool(c, 2)
            convs.append(c)
        if ae == 1:
            pool = L.dense_ae(pool, feature_number, w, "ae")
        mid = CB1x2(L, pool, w * 2 ** d, k, "mid")
    if method == "UNet":
        de = mid
        for j in range(d):
            lvl = d - j - 1
            p = f"dec{j}"
            s = convs[lvl]
            if ag == 1:
                s = AG1(L, s, de, w, 2 ** lvl, f"{p}/ag0")
            if ds == 1:
                levels[f"level{d - j}"] = L.conv(de, 1, 1, f"level{d - j}")
            de = up(de, w * 2 ** lvl, p)
            de = L.convlstm(torch.cat([s, de], -1), int(w * 2.0 ** (lvl - 1)), f"{p}/lstm") if lstm == 1 \
                else torch.cat([de, s], -1)
            de = CB1x2(L, de, w * 2 ** lvl, k, p)
        outs = _ret(_head1(L, de, problem_type, output_nums), levels, ds, d)
    elif method == "UNetPP":
        convs = convs + [mid]
        if ds == 1:
            levels[f"level{d}"] = L.conv(convs[0], 1, 1, f"level{d}")
        node = {}
        for i in range(1, d + 1):
            for j in range(0, d - i + 1):
                p = f"X{j}{i}"
                below = convs[j + 1] if i == 1 else node[(j + 1, i - 1)]
                extra = [node[(j, q)] for q in range(1, i)]
                s = convs[j]
                if ag == 1:
                    extra = [AG1(L, t, below, w, 2 ** j, f"{p}/ag{n}") for n, t in enumerate(extra)]
                    s = AG1(L, s, below, w, 2 ** j, f"{p}/ag{len(extra)}")
                u = up(below, w * 2 ** j, p)
                u = L.convlstm(torch.cat([s, u] + extra, -1), int(w * 2.0 ** (j - 1)), f"{p}/lstm") if lstm == 1 \
                    else torch.cat([u] + extra + [s], -1)
                node[(j, i)] = CB1x2(L, u, w * 2 ** j, k, p)
                if ds == 1 and j == 0 and i < d:
                    levels[f"level{d - i}"] = L.conv(node[(0, i)], 1, 1, f"level{d - i}")
        outs = _ret(_head1(L, node[(0, d)], problem_type, output_nums), levels, ds, d)
    elif method == "MultiResUNet":
        skips, pool = [], x
        for i in range(1, d + 1):
            m = MRB1(L, pool, w, k, 2 ** (i - 1), alpha, f"enc{i}/mrb")
            pool = L.pool(m, 2)
            skips.append(ResPath1(L, m, d - i + 1, w * 2 ** (i - 1), k, f"enc{i}/respath"))
        if ae == 1:
            pool = L.dense_ae(pool, feature_number, w, "ae")
        de = MRB1(L, pool, w, k, 2 ** d, alpha, "mid/mrb")
        for j in range(d):
            lvl = d - j - 1
            p = f"dec{j}"
            s = skips[lvl]
            if ag == 1:
                s = AG1(L, s, de, w, 2 ** lvl, f"{p}/ag0")
            if ds == 1:
                levels[f"level{d - j}"] = L.conv(de, 1, 1, f"level{d - j}")
            de = up(de, w * 2 ** lvl, p)
            de = L.convlstm(torch.cat([s, de], -1), int(w * 2.0 ** (lvl - 1)), f"{p}/lstm") if lstm == 1 \
                else torch.cat([de, s], -1)
            de = MRB1(L, de, w, k, 2 ** lvl, alpha, f"{p}/mrb")
        outs = _ret(_head1(L, de, problem_type, output_nums), levels, ds, d)
    else:
        raise NotImplementedError(method)
    return [o.squeeze(1) for o in outs]


def bcdunet1d_forward(L: Layers, x, model_depth, model_width, kernel_size, problem_type="Regression", output_nums=1,
                      ds=1, ae=0, ag=0, lstm=0, dense_loop=1, feature_number=1024, is_transconv=True):
    """BCDUNet.py:111-174 (lstm=0 silently drops the skip, :152-158)."""
    x = x.unsqueeze(1)
    w, d, k = model_width, model_depth, kernel_size
    convs, pool, levels = [], x, {}
    for i in range(1, d + 1):
        c = CB1x2(L, pool, w * 2 ** (i - 1), k, f"enc{i}")
        pool = L.pool(c, 2)
        convs.append(c)
    conv = pool
    for n in range(dense_loop - 1):
        conv = torch.cat([conv, CB1x2(L, conv, w * 2 ** d, k, f"dense{n}")], -1)
    if ae == 1:
        conv = L.dense_ae(conv, feature_number, w, "ae")
    de = CB1x2(L, conv, w * 2 ** d, k, "mid")
    for j in range(d):
        lvl = d - j - 1
        p = f"dec{j}"
        s = convs[lvl]
        if ag == 1:
            s = AG1(L, s, de, w, 2 ** lvl, f"{p}/ag0")
        if ds == 1:
            levels[f"level{d - j}"] = L.conv(de, 1, 1, f"level{d - j}")
        de = TC1(L, de, w * 2 ** lvl, p) if is_transconv else L.up(de, 2, "nearest")
        if lstm == 1:
            de = L.convlstm(torch.cat([s, de], -1), int(w * 2.0 ** (lvl - 1)), f"{p}/lstm")
        de = CB1x2(L, de, w * 2 ** lvl, k, p)
    outs = _ret(_head1(L, de, problem_type, output_nums), levels, ds, d)
    return [o.squeeze(1) for o in outs]
