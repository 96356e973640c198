"""2D model-builder API of the reference, re-stated as a graph IR producer.

Mirrors the public surface of `TensorFlow/2DCNN/models/unet_variants.py:977-1115` (class
``unet_model_builder`` + its 33 encoder-named zero-argument methods) and
`TensorFlow/2DCNN/models/model_selector.py:8-94` (class ``model_selector``): same positional
order, keyword names, defaults and ``ValueError``s.  Only ``train_mode='from_scratch'`` can be
built: the pretrained ImageNet/CheXNet encoders need downloaded weights and are outside the hot
path.

Structural parameter names (shared with the oracle, DESIGN.md "naming"):
  enc{i}/conv, enc{i}/bn                      i = 1..d+1        encoder Conv_Block
  latent/conv{k}, latent/bn{k}                k = 0..dense_loop latent dense_block
  dec{j}/...  (UNet, MultiResUNet)            j = 0..d-1
  X{j}{i}/... (UNetE / UNetP / UNetPP)        node (j, i)
  .../tconv, .../lstm, .../conv, .../bn, .../ag{k}/{theta,phi,psi,up}/...
  MultiResBlock: <p>/sc/{conv,bn} <p>/c3/.. <p>/c5/.. <p>/c7/.. <p>/bn_cat <p>/bn_out
  ResPath: <p>/{k}/sc/{conv,bn} <p>/{k}/cv/{conv,bn} <p>/{k}/bn
  heads: level{k}, out
"""
from __future__ import annotations

from typing import Dict, List, Tuple

from .graph import Graph, Sym

ENCODER_METHODS = (
    "ResNet50", "ResNet50V2", "ResNet101", "ResNet101V2", "ResNet152", "ResNet152V2", "VGG16", "VGG19",
    "DenseNet121", "DenseNet169", "DenseNet201", "MobileNet", "MobileNetV2", "MobileNetV3Small",
    "MobileNetV3Large", "InceptionV3", "InceptionResNetV2", "EfficientNetB0", "EfficientNetB1",
    "EfficientNetB2", "EfficientNetB3", "EfficientNetB4", "EfficientNetB5", "EfficientNetB6",
    "EfficientNetB7", "EfficientNetV2B0", "EfficientNetV2B1", "EfficientNetV2B2", "EfficientNetV2B3",
    "EfficientNetV2S", "EfficientNetV2M", "EfficientNetV2L", "CheXNet",
)

SUPPORTED_DECODERS = ("UNet", "UNetE", "UNetP", "UNetPP", "UNet3P", "MultiResUNet")


# ---------------------------------------------------------------------------------------------- blocks
def conv_block(g: Graph, x: Sym, filters: int, k, p: str, bn=True, act="relu", init="he_uniform") -> Sym:
    """Conv2D('same', he_uniform) -> BN -> ReLU   (unet_variants.py:7-14)."""
    y = g.conv(x, filters, k, f"{p}/conv", init=init)
    if bn:
        y = g.bn(y, f"{p}/bn")
    if act is not None:
        y = g.act(y, act, f"{p}/{act}")
    return y


def trans_conv(g: Graph, x: Sym, filters: int, p: str) -> Sym:
    """Conv2DTranspose(4x4, s2, 'same') -> LeakyReLU(0.3), no BN   (unet_variants.py:17-24)."""
    y = g.conv_transpose(x, filters, (4, 4), (2, 2), f"{p}/tconv")
    return g.act(y, "leaky_relu", f"{p}/tconv_act", alpha=0.3)


def up_block(g: Graph, x: Sym, p: str, size=2) -> Sym:
    """UpSampling2D(bilinear)   (unet_variants.py:35-38)."""
    return g.upsample(x, size, f"{p}/up", interpolation="bilinear")


def attention_gate(g: Graph, skip: Sym, gate: Sym, nf: int, mult: int, p: str) -> Sym:
    """Attention_Block (unet_variants.py:67-82)."""
    th = g.bn(g.conv(skip, nf * mult, 1, f"{p}/theta/conv", strides=2), f"{p}/theta/bn")
    ph = g.bn(g.conv(gate, nf * mult, 1, f"{p}/phi/conv"), f"{p}/phi/bn")
    s = g.act(g.add([th, ph], f"{p}/add"), "relu", f"{p}/relu")
    ps = g.bn(g.conv(s, 1, 1, f"{p}/psi/conv"), f"{p}/psi/bn")
    ps = g.act(ps, "sigmoid", f"{p}/sigmoid")
    r1 = g.upsample(ps, 2, f"{p}/up_bilinear", interpolation="bilinear")
    r2 = trans_conv(g, ps, 1, f"{p}/up")
    r = g.add([r1, r2], f"{p}/resample")
    return g.mul(skip, r, f"{p}/mul")


def multires_block(g: Graph, x: Sym, width: int, alpha: float, p: str) -> Sym:
    """MultiResBlock (unet_variants.py:85-100)."""
    w = alpha * width
    a, b, c = int(w * 0.167), int(w * 0.333), int(w * 0.5)
    sc = conv_block(g, x, a + b + c, 1, f"{p}/sc")
    c3 = conv_block(g, x, a, 3, f"{p}/c3")
    c5 = conv_block(g, c3, b, 3, f"{p}/c5")
    c7 = conv_block(g, c5, c, 3, f"{p}/c7")
    cat = g.bn(g.concat([c3, c5, c7], f"{p}/cat"), f"{p}/bn_cat")
    s = g.act(g.add([sc, cat], f"{p}/add"), "relu", f"{p}/relu")
    return g.bn(s, f"{p}/bn_out")


def res_path(g: Graph, x: Sym, length: int, width: int, p: str) -> Sym:
    """ResPath (unet_variants.py:103-122): at least one iteration even for length 0."""
    out = x
    for k in range(max(int(length), 1)):
        sc = conv_block(g, out, width, 1, f"{p}/{k}/sc")
        cv = conv_block(g, out, width, 3, f"{p}/{k}/cv")
        out = g.bn(g.act(g.add([sc, cv], f"{p}/{k}/add"), "relu", f"{p}/{k}/relu"), f"{p}/{k}/bn")
    return out


def _conv_block_named(g, x, filters, k, conv_name, bn_name, act_name):
    y = g.conv(x, filters, k, conv_name, init="he_uniform")
    y = g.bn(y, bn_name)
    return g.act(y, "relu", act_name)


def latent_dense(g: Graph, x: Sym, filters: int, loops: int) -> Sym:
    x = _conv_block_named(g, x, filters, 3, "latent/conv0", "latent/bn0", "latent/relu0")
    for k in range(1, loops + 1):
        cb = _conv_block_named(g, x, filters, 3, f"latent/conv{k}", f"latent/bn{k}", f"latent/relu{k}")
        x = g.add([x, cb], f"latent/add{k}")
    return x


def convlstm_merge(g: Graph, parts: List[Sym], filters: int, p: str) -> Sym:
    """Reshape->(1,H,W,C) x n, channel concat, ConvLSTM2D(T=1)  (unet_variants.py:145-149, 327-333)."""
    m = g.concat(parts, f"{p}/lstm_cat")
    return g.convlstm(m, filters, 3, f"{p}/lstm")


# ---------------------------------------------------------------------------------------------- decoders
def _upsample_step(g, src, filters, is_transconv, p):
    return trans_conv(g, src, filters, p) if is_transconv else up_block(g, src, p)


def decoder_unet(g, skips: List[Sym], w, d, ds, ag, lstm, is_transconv, multires=False, alpha=1.0):
    """UNet (unet_variants.py:125-154) and MultiResUNet (:459-487) decoders."""
    levels: Dict[str, Sym] = {}
    de = skips[-1]
    for j in range(d):
        L = d - j - 1
        C = w * (2 ** L)
        p = f"dec{j}"
        s = skips[L]
        if ag:
            s = attention_gate(g, s, de, w, 2 ** L, f"{p}/ag0")
        if ds:
            levels[f"level{d - j}"] = g.conv(de, 1, 1, f"level{d - j}")
        de = _upsample_step(g, de, C, is_transconv, p)
        if lstm:
            if multires:
                # the reference reads undefined globals here (unet_variants.py:477); same failure, earlier
                raise NameError("name 'length' is not defined (reference MultiResUNet has no working lstm branch)")
            de = convlstm_merge(g, [s, de], int(w * (2.0 ** (L - 1))), p)
        else:
            de = g.concat([de, s], f"{p}/cat")
        if multires:
            de = multires_block(g, de, C, alpha, f"{p}/mrb")
        else:
            de = conv_block(g, de, C, 3, p)
    return de, levels


def decoder_nested(g, skips: List[Sym], w, d, ds, ag, lstm, is_transconv, variant: str):
    """UNetE (:157-214), UNetP (:217-274), UNetPP (:277-343): node (j, i), i = 1..d, j = 0..d-i."""
    levels: Dict[str, Sym] = {}
    if ds:
        levels[f"level{d}"] = g.conv(skips[0], 1, 1, f"level{d}")
    X: Dict[Tuple[int, int], Sym] = {}
    for i in range(1, d + 1):
        for j in range(0, d - i + 1):
            p = f"X{j}{i}"
            C = w * (2 ** j)
            src = skips[j + 1] if i == 1 else X[(j + 1, i - 1)]
            dense: List[Sym] = []
            if i == 1 or variant == "UNetE":
                s = skips[j]
            elif variant == "UNetP":
                s = X[(j, i - 1)]
            else:  # UNetPP
                dense = [X[(j, k)] for k in range(1, i)]
                s = skips[j]
            nag = 0
            if ag:
                gated = []
                for t in dense:                      # creation order: dense[0], dense[1..], then skip
                    gated.append(attention_gate(g, t, src, w, 2 ** j, f"{p}/ag{nag}"))
                    nag += 1
                dense = gated
                s = attention_gate(g, s, src, w, 2 ** j, f"{p}/ag{nag}")
            de = _upsample_step(g, src, C, is_transconv, p)
            if lstm:
                de = convlstm_merge(g, [s, de] + dense, int(w * (2.0 ** (j - 1))), p)
            else:
                de = g.concat([de] + dense + [s], f"{p}/cat")
            X[(j, i)] = conv_block(g, de, C, 3, p)
            if ds and j == 0 and i < d:
                levels[f"level{d - i}"] = g.conv(X[(0, i)], 1, 1, f"level{d - i}")
    return X[(0, d)], levels


def decoder_unet3p(g, skips: List[Sym], w, d, ds):
    """UNet3P (unet_variants.py:346-376)."""
    levels: Dict[str, Sym] = {}
    de = skips[-1]
    D: Dict[int, Sym] = {}
    for j in range(d):
        p = f"dec{j}"
        L = d - j - 1
        parts = [conv_block(g, skips[L], w, 3, f"{p}/same")]
        for k in range(0, L):
            f = 2 ** (L - k)
            t = g.maxpool(skips[k], f, f"{p}/down{k}/pool")
            parts.append(conv_block(g, t, w, 3, f"{p}/down{k}"))
        t = conv_block(g, de, w, 3, f"{p}/below")
        t = g.upsample(t, 2, f"{p}/below/up", interpolation="bilinear")
        parts.append(g.act(t, "sigmoid", f"{p}/below/sigmoid"))
        for m in range(0, j):
            t = conv_block(g, D[m], w, 3, f"{p}/deep{m}")
            t = g.upsample(t, 2 ** (j - m), f"{p}/deep{m}/up", interpolation="bilinear")
            parts.append(g.act(t, "sigmoid", f"{p}/deep{m}/sigmoid"))
        cat = g.concat(parts, f"{p}/cat")
        de = conv_block(g, cat, w * (d + 1), 3, p)
        D[j] = de
        if ds:
            levels[f"level{d - j}"] = g.conv(de, 1, 1, f"level{d - j}", strides=2)
    return de, levels


# ---------------------------------------------------------------------------------------------- builder
def build_graph_2d(decoder_name, length, width, model_width, model_depth, num_channels, output_nums, ds, ae, ag,
                   lstm, dense_loop, feature_number, is_transconv, alpha, final_activation, model_name) -> Graph:
    """from_scratch path of every encoder-named method (unet_variants.py:1089-1115)."""
    if decoder_name not in SUPPORTED_DECODERS:
        raise NotImplementedError(
            f"decoder {decoder_name!r} is outside the hot path (SURVEY.md section 2); supported: {SUPPORTED_DECODERS}")
    g = Graph(ndim=2, name=model_name)
    w, d = int(model_width), int(model_depth)
    x = g.input(int(length), int(width), int(num_channels))
    multires = decoder_name == "MultiResUNet"
    # ---- encoder (unet_variants.py:750-792): d+1 levels, returns the un-pooled last conv
    skips: List[Sym] = []
    pool = x
    conv = x
    for i in range(1, d + 2):
        C = w * (2 ** (i - 1))
        if multires:
            conv = multires_block(g, pool, C, alpha, f"enc{i}/mrb")
            pool = g.maxpool(conv, 2, f"enc{i}/pool")
            skips.append(res_path(g, conv, d - i + 1, C, f"enc{i}/respath"))
        else:
            conv = conv_block(g, pool, C, 3, f"enc{i}")
            pool = g.maxpool(conv, 2, f"enc{i}/pool")
            skips.append(conv)
    # ---- latent (unet_variants.py:966-974)
    F = w * (2 ** d)
    if multires:
        lat = multires_block(g, conv, F, alpha, "latent/mrb")
    else:
        lat = latent_dense(g, conv, F, int(dense_loop))
    if ae == 1:
        lat = g.dense_ae(lat, int(feature_number), F, "ae")
    skips = skips[:d] + [lat]                       # unet_variants.py:1100-1101
    # ---- decoder (unet_variants.py:936-963)
    if decoder_name == "UNet":
        y, levels = decoder_unet(g, skips, w, d, ds == 1, ag == 1, lstm == 1, is_transconv)
    elif decoder_name == "MultiResUNet":
        y, levels = decoder_unet(g, skips, w, d, ds == 1, ag == 1, lstm == 1, is_transconv, True, alpha)
    elif decoder_name in ("UNetE", "UNetP", "UNetPP"):
        y, levels = decoder_nested(g, skips, w, d, ds == 1, ag == 1, lstm == 1, is_transconv, decoder_name)
    else:
        y, levels = decoder_unet3p(g, skips, w, d, ds == 1)
    out = g.conv(y, int(output_nums), 1, "out", activation=final_activation)
    if ds == 1:                                     # [out, level1, ..., level_d]   (:1111-1114)
        names = ["out"] + [f"level{k}" for k in range(1, d + 1)]
        g.set_outputs([out] + [levels[n] for n in names[1:]], names)
    else:
        g.set_outputs([out], ["out"])
    return g


class unet_model_builder:
    """Same signature as `TensorFlow/2DCNN/models/unet_variants.py:977-998`."""

    def __init__(self, decoder_name, length, width, model_width, model_depth, num_channels=3, output_nums=1, ds=0,
                 ae=0, ag=0, lstm=0, dense_loop=1, feature_number=1024, is_transconv=True, alpha=1.0, q=3,
                 final_activation="sigmoid", train_mode="pretrained_encoder", is_base_model_trainable=False):
        self.decoder_name = decoder_name
        self.length = length
        self.width = width
        self.model_depth = model_depth
        self.model_width = model_width
        self.num_channels = num_channels
        self.output_nums = output_nums
        self.D_S = ds
        self.A_E = ae
        self.A_G = ag
        self.LSTM = lstm
        self.dense_loop = dense_loop
        self.feature_number = feature_number
        self.is_transconv = is_transconv
        self.final_activation = final_activation
        self.train_mode = train_mode
        self.is_base_model_trainable = is_base_model_trainable
        self.alpha = alpha
        self.q = q
        # argument validation, unet_variants.py:1036-1043
        if self.train_mode == "pretrained_encoder":
            if (self.model_depth > 5) or (self.model_depth < 1):
                raise ValueError("The depth of a TF-ImageNet Pretrained model can only be discretely varied from 1 to 5")
        elif self.train_mode == "from_scratch":
            if self.model_depth < 1:
                raise ValueError("The depth of the model cannot be less than 1")
        else:
            raise ValueError('The Train Mode can only be: "pretrained_encoder" or "from_scratch"')

    def graph(self, encoder_name="ResNet50") -> Graph:
        if self.length == 0:
            raise ValueError("Please Check the Values of the Input Parameters!")
        if self.train_mode != "from_scratch":
            raise NotImplementedError(
                "train_mode='pretrained_encoder' needs downloaded ImageNet/CheXNet weights; only "
                "'from_scratch' is on the MI355X hot path")
        return build_graph_2d(self.decoder_name, self.length, self.width, self.model_width, self.model_depth,
                              self.num_channels, self.output_nums, self.D_S, self.A_E, self.A_G, self.LSTM,
                              self.dense_loop, self.feature_number, self.is_transconv, self.alpha,
                              self.final_activation, f"{encoder_name}_{self.decoder_name}")

    def _build(self, encoder_name):
        from .model import SegModel          # late import: graph building must not need torch
        return SegModel(self.graph(encoder_name))


def _make_encoder_method(enc):
    def method(self):
        return self._build(enc)
    method.__name__ = enc
    method.__doc__ = (f"{enc}_<decoder>: with train_mode='from_scratch' every encoder-named method runs the "
                      "same scratch encoder (unet_variants.py:1089-1115); only the model name differs.")
    return method


for _enc in ENCODER_METHODS:
    setattr(unet_model_builder, _enc, _make_encoder_method(_enc))


# encoder-name aliases accepted by the reference ladder (model_selector.py:75-1330)
_ALIASES = {e.lower(): e for e in ENCODER_METHODS}
_ALIASES.update({"mobilenetv3s": "MobileNetV3Small", "mobilenetv3_small": "MobileNetV3Small",
                 "mobilenetv3l": "MobileNetV3Large", "mobilenetv3_large": "MobileNetV3Large",
                 "inception_v3": "InceptionV3", "inceptionresnet_v2": "InceptionResNetV2"})


class model_selector:
    """Same signature as `TensorFlow/2DCNN/models/model_selector.py:8-31`; the if/elif ladder is a table."""

    def __init__(self, model_genre, encoder_name, decoder_name, imlength, imwidth, model_width, model_depth,
                 num_channels=3, output_nums=1, ds=0, ae=0, ag=0, lstm=0, dense_loop=1, feature_number=1024,
                 is_transconv=True, alpha=1.0, q=3, final_activation="sigmoid", train_mode="pretrained_encoder",
                 is_base_model_trainable=False):
        self.model_genre = model_genre
        self.encoder_name = encoder_name
        self.decoder_name = decoder_name
        self.imlength = imlength
        self.imwidth = imwidth
        self.model_depth = model_depth
        self.model_width = model_width
        self.num_channels = num_channels
        self.output_nums = output_nums
        self.D_S = ds
        self.A_E = ae
        self.A_G = ag
        self.LSTM = lstm
        self.dense_loop = dense_loop
        self.feature_number = feature_number
        self.is_transconv = is_transconv
        self.final_activation = final_activation
        self.train_mode = train_mode
        self.is_base_model_trainable = is_base_model_trainable
        self.alpha = alpha
        self.q = q

    def _builder(self):
        if self.model_genre not in ("UNet", "unet", "U-Net"):
            raise NotImplementedError(f"model_genre {self.model_genre!r}: only the UNet genre is on the hot path "
                                      "(FPN builders are out of scope, SURVEY.md section 2 row 6)")
        enc = _ALIASES.get(str(self.encoder_name).lower())
        if enc is None:
            raise ValueError(f"unknown encoder_name {self.encoder_name!r}")
        b = unet_model_builder(self.decoder_name, self.imlength, self.imwidth, self.model_width, self.model_depth,
                               num_channels=self.num_channels, output_nums=self.output_nums, ds=self.D_S,
                               ae=self.A_E, ag=self.A_G, lstm=self.LSTM, dense_loop=self.dense_loop,
                               feature_number=self.feature_number, is_transconv=self.is_transconv, alpha=self.alpha,
                               q=self.q, final_activation=self.final_activation, train_mode=self.train_mode,
                               is_base_model_trainable=self.is_base_model_trainable)
        return b, enc

    def graph(self) -> Graph:
        b, enc = self._builder()
        return b.graph(enc)

    def segmentation_model(self):
        b, enc = self._builder()
        return getattr(b, enc)()
