"""Data formats on either side of the hot path (the callers of rows f1/f2): host logic only.

What the reference's loaders hand to `model.fit` / take from `model.predict`, restated in numpy + PIL:

* `CustomDataGenerator` - `2DCNN/utils/DataGenerator.py:7-114`: a folder of images and a folder of masks with the
  same file names -> float32 image batches (RGB, resized with Lanczos, divided by norm_factor_img) and int8 mask
  batches (grey, nearest-neighbour resize, divided by norm_factor_msk and truncated), optionally cut into
  overlapping square patches, optionally expanded into the deep-supervision target dictionary.
* `prepare_train_dict` - `2DCNN/utils/helper_functions.py:359-380` (the variant the generator imports): 'out' is
  the mask batch, 'level{i}' its 2^i x 2^i max-pool for ds_type 'UNet' and the mask itself for 'UNetPP'.
* `create_patches` - `helper_functions.py:18-28`; `patchify` / `unpatchify` restate the third-party package the
  reference imports (patchify 0.2.3, absent from this image): a sliding-window view with a fixed step, and its
  inverse which lays the patches back in raster order, later patches overwriting earlier ones, skipping patches
  whose area the previous one already covered.
* `dice` - `helper_functions.py:383-411`; `confusion_matrix` / `multilabel_confusion_matrix` are the two sklearn
  calls of `2DCNN/Test.py:191-193` in plain numpy.

Nothing here touches the GPU; `targets_for` orders a target dictionary the way the engine's outputs are ordered.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


# ------------------------------------------------------------------------------------------------ patches
def patchify(image: np.ndarray, patch_size: Sequence[int], step: int = 1) -> np.ndarray:
    """Sliding windows of `patch_size` every `step` elements along every axis (patchify 0.2.3 `patchify`, itself
    skimage's view_as_windows): result shape = (n_0, n_1, [n_2,] p_0, p_1, [p_2]) with
    n_k = (image.shape[k] - p_k) // step + 1.  The result is a read-only view."""
    image = np.asarray(image)
    patch_size = tuple(int(p) for p in patch_size)
    if image.ndim != len(patch_size):
        raise ValueError("`patch_size` is incompatible with the image shape")
    step = int(step)
    if step < 1:
        raise ValueError("`step` must be >= 1")
    if any(p < 1 or p > s for p, s in zip(patch_size, image.shape)):
        raise ValueError("`patch_size` must be between 1 and the image extent on every axis")
    counts = tuple((s - p) // step + 1 for s, p in zip(image.shape, patch_size))
    strides = tuple(st * step for st in image.strides) + image.strides
    return np.lib.stride_tricks.as_strided(image, shape=counts + patch_size, strides=strides, writeable=False)


def unpatchify(patches: np.ndarray, imsize: Sequence[int]) -> np.ndarray:
    """Inverse of `patchify` for 2-D images (patchify 0.2.3 `unpatchify`, the form `Test.py:159` uses): the step is
    recovered from the counts, must be an integer, and the patches are written back in raster order - a later patch
    overwrites an earlier one, and after writing a patch the walk jumps to the first patch that starts at or beyond
    its far edge (clamped to the last one), so with overlapping patches only that subset is used."""
    patches = np.asarray(patches)
    imsize = tuple(int(v) for v in imsize)
    if patches.ndim != 2 * len(imsize) or len(imsize) != 2:
        raise ValueError("unpatchify: expected (n_h, n_w, p_h, p_w) patches and a 2-D image size")
    i_h, i_w = imsize
    n_h, n_w, p_h, p_w = patches.shape
    s_h = 0 if n_h <= 1 else (i_h - p_h) / (n_h - 1)
    s_w = 0 if n_w <= 1 else (i_w - p_w) / (n_w - 1)
    if int(s_h) != s_h or int(s_w) != s_w:
        raise ValueError("unpatchify: the patches do not tile the image with a uniform integer step")
    s_h, s_w = int(s_h), int(s_w)
    if (n_h > 1 and s_h == 0) or (n_w > 1 and s_w == 0):
        raise ValueError("unpatchify: more than one patch along an axis the patch already spans")
    image = np.zeros(imsize, dtype=patches.dtype)
    i = j = 0
    while True:
        i_o, j_o = i * s_h, j * s_w
        image[i_o:i_o + p_h, j_o:j_o + p_w] = patches[i, j]
        if j < n_w - 1:
            j = min((j_o + p_w) // s_w, n_w - 1)
        elif i < n_h - 1:
            i = min((i_o + p_h) // s_h, n_h - 1)
            j = 0
        else:
            break
    return image


def create_patches(input_image, patch_shape: Sequence[int], overlap_ratio: float) -> Tuple[np.ndarray, int]:
    """helper_functions.py:18-28: square patches with step int(patch * (1 - overlap_ratio)); a 3-D image keeps all its
    channels in every patch.  Returns (patches, rows * columns)."""
    img = np.asarray(input_image)
    pw, ph = int(patch_shape[0]), int(patch_shape[1])
    if pw != ph:
        raise AssertionError("The patches are required to be squared shape")
    shape = (pw, ph, img.shape[2]) if img.ndim == 3 else (pw, ph)
    patches = patchify(img, shape, step=int(ph * (1 - overlap_ratio)))
    return patches, patches.shape[0] * patches.shape[1]


# ------------------------------------------------------------------------------------------------ deep supervision
def max_pool2d(batch: np.ndarray, w: int) -> np.ndarray:
    """Keras MaxPooling2D(pool_size=(w, w)), 'valid': (N, H, W, C) -> (N, H // w, W // w, C)."""
    n, h, wd, c = batch.shape
    ho, wo = h // w, wd // w
    return batch[:, :ho * w, :wo * w, :].reshape(n, ho, w, wo, w, c).max(axis=(2, 4))


def prepare_train_dict(mask_batch, model_depth: int, model_type: str) -> Dict[str, np.ndarray]:
    """helper_functions.py:359-380: {'out': masks, 'level1': ..., 'level<depth>': ...}; (N, H, W) input gets a channel
    axis.  model_type 'UNet': level i is the 2^i max-pool of the masks; 'UNetPP': every level is the masks."""
    y = np.array(mask_batch)
    if y.ndim == 3:
        y = y[..., None]
    out = {"out": y}
    for i in range(1, int(model_depth) + 1):
        if model_type == "UNet":
            out[f"level{i}"] = max_pool2d(y, 2 ** i)
        elif model_type == "UNetPP":
            out[f"level{i}"] = y
        else:                       # the reference leaves the key unset and fails at the dictionary read
            raise KeyError(f"level{i}")
    return out


def targets_for(model, y) -> List[np.ndarray]:
    """Order a Keras-style target (array, list or {output name: array}) like the model's outputs."""
    names = list(model.output_names)
    if isinstance(y, dict):
        missing = [n for n in names if n not in y]
        if missing:
            raise KeyError(f"no target for output(s) {missing}")
        return [np.asarray(y[n], np.float32) for n in names]
    if isinstance(y, (list, tuple)):
        if len(y) != len(names):
            raise ValueError(f"{len(y)} targets for {len(names)} outputs")
        return [np.asarray(t, np.float32) for t in y]
    if len(names) != 1:
        raise ValueError(f"one target for {len(names)} outputs")
    return [np.asarray(y, np.float32)]


# ------------------------------------------------------------------------------------------------ metrics
def dice(im1, im2, empty_score: float = 1.0) -> float:
    """helper_functions.py:383-411: 2 |A and B| / (|A| + |B|) on boolean casts; empty_score when both are empty."""
    a, b = np.asarray(im1).astype(bool), np.asarray(im2).astype(bool)
    if a.shape != b.shape:
        raise ValueError("Shape mismatch: im1 and im2 must have the same shape.")
    tot = a.sum() + b.sum()
    if tot == 0:
        return empty_score
    return 2.0 * np.logical_and(a, b).sum() / tot


def _labels(y_true, y_pred):
    return np.unique(np.concatenate([np.asarray(y_true).ravel(), np.asarray(y_pred).ravel()]))


def confusion_matrix(y_true, y_pred, labels: Optional[Sequence[int]] = None) -> np.ndarray:
    """sklearn.metrics.confusion_matrix (Test.py:191): C[i, j] = samples of class labels[i] predicted as labels[j];
    labels default to the sorted union of the values present."""
    yt, yp = np.asarray(y_true).ravel(), np.asarray(y_pred).ravel()
    labels = _labels(yt, yp) if labels is None else np.asarray(labels)
    k = len(labels)
    it, ip = np.searchsorted(labels, yt), np.searchsorted(labels, yp)
    ok = (it < k) & (ip < k)
    ok &= (labels[np.minimum(it, k - 1)] == yt) & (labels[np.minimum(ip, k - 1)] == yp)
    return np.bincount(it[ok] * k + ip[ok], minlength=k * k).reshape(k, k).astype(np.int64)


def multilabel_confusion_matrix(y_true, y_pred, labels: Optional[Sequence[int]] = None) -> np.ndarray:
    """sklearn.metrics.multilabel_confusion_matrix on 1-D multiclass input (Test.py:193): one [[tn, fp], [fn, tp]]
    per class, one-vs-rest."""
    yt, yp = np.asarray(y_true).ravel(), np.asarray(y_pred).ravel()
    labels = _labels(yt, yp) if labels is None else np.asarray(labels)
    out = np.zeros((len(labels), 2, 2), np.int64)
    for i, lab in enumerate(labels):
        t, p = yt == lab, yp == lab
        out[i] = [[np.sum(~t & ~p), np.sum(~t & p)], [np.sum(t & ~p), np.sum(t & p)]]
    return out


# ------------------------------------------------------------------------------------------------ files
def load_img(path: str, color_mode: str, target_size: Tuple[int, int], interpolation: str) -> np.ndarray:
    """tf.keras.utils.load_img(path, color_mode, target_size=(height, width), interpolation, keep_aspect_ratio=False)
    followed by the array cast the arithmetic on it implies: PIL open, convert to 'RGB' / 'L', resize to
    (width, height) when the size differs.  Returns float32 (H, W, 3) or (H, W)."""
    from PIL import Image
    resample = {"nearest": Image.Resampling.NEAREST, "bilinear": Image.Resampling.BILINEAR,
                "bicubic": Image.Resampling.BICUBIC, "lanczos": Image.Resampling.LANCZOS}[interpolation]
    with Image.open(path) as im:
        if color_mode == "grayscale":
            im = im.convert("L") if im.mode not in ("L", "I;16", "I") else im
        elif color_mode == "rgb":
            im = im.convert("RGB") if im.mode != "RGB" else im
        else:
            raise ValueError('color_mode must be "grayscale" or "rgb"')
        wh = (int(target_size[1]), int(target_size[0]))
        if im.size != wh:
            im = im.resize(wh, resample)
        return np.asarray(im, dtype=np.float32)


class CustomDataGenerator:
    """DataGenerator.py:7-114, the keras.utils.Sequence the reference trains from.

    len(g) batches per epoch; g[k] -> (x, y): x float32 (n, H, W, num_img_channel), y int8 (n, H, W, 1) or, with
    deep_supervision, the `prepare_train_dict` dictionary.  The file order is os.listdir's (shuffled per epoch when
    is_train).  With patchify the batch holds every patch of its images (the reference's 10000-patch staging buffer
    is the same cap).  drop_remainder (not in the reference): leave out the last, smaller batch - the engine's batch
    size is fixed when the model is built, Keras' is not."""

    def __init__(self, img_dir, msk_dir, img_size=(224, 224), batch_size=8, num_img_channel=3, num_msk_channel=1,
                 norm_factor_img=255, norm_factor_msk=255, num_class=1, is_train=True, patchify=False,
                 patch_shape=(64, 64), overlap_ratio=0.5, deep_supervision=False, model_depth=5, ds_type="UNet",
                 drop_remainder=False, rng: Optional[np.random.Generator] = None):
        self.img_dir, self.msk_dir = img_dir, msk_dir
        self.img_list, self.msk_list = os.listdir(img_dir), os.listdir(msk_dir)
        assert len(self.img_list) == len(self.msk_list), "Number of Images and Masks should be the same!"
        self.img_size, self.batch_size = tuple(img_size), int(batch_size)
        self.num_img_channel, self.num_msk_channel = num_img_channel, num_msk_channel
        self.norm_factor_img, self.norm_factor_msk = norm_factor_img, norm_factor_msk
        self.num_class, self.is_train = num_class, is_train
        self.patchify, self.patch_shape, self.overlap_ratio = patchify, tuple(patch_shape), overlap_ratio
        self.deep_supervision, self.model_depth, self.ds_type = bool(deep_supervision), model_depth, ds_type
        self.drop_remainder = drop_remainder
        self.rng = rng if rng is not None else np.random.default_rng()
        self.on_epoch_end()

    def __len__(self):
        n = len(self.img_list) / self.batch_size
        return int(math.floor(n) if self.drop_remainder else math.ceil(n))

    def __getitem__(self, index):
        if not 0 <= index < len(self):
            raise IndexError(index)
        ids = self.indexes[index * self.batch_size:(index + 1) * self.batch_size]
        return self._data_generation([self.img_list[k] for k in ids])

    def __iter__(self):
        for k in range(len(self)):
            yield self[k]

    def on_epoch_end(self):
        self.indexes = np.arange(len(self.img_list))
        if self.is_train:
            self.rng.shuffle(self.indexes)

    def _data_generation(self, names):
        xs, ys = [], []
        for name in names:
            img = load_img(f"{self.img_dir}/{name}", "rgb", self.img_size, "lanczos") / self.norm_factor_img
            msk = load_img(f"{self.msk_dir}/{name}", "grayscale", self.img_size, "nearest") / self.norm_factor_msk
            if self.patchify:
                p, n = create_patches(img, self.patch_shape, self.overlap_ratio)
                xs.append(np.reshape(np.squeeze(p), (n, self.patch_shape[0], self.patch_shape[1], self.num_img_channel)))
                p, n = create_patches(msk, self.patch_shape, self.overlap_ratio)
                ys.append(np.reshape(np.squeeze(p), (n, self.patch_shape[0], self.patch_shape[1], self.num_msk_channel)))
            else:
                xs.append(img.reshape(1, self.img_size[0], self.img_size[1], self.num_img_channel))
                ys.append(msk.reshape(1, self.img_size[0], self.img_size[1], self.num_msk_channel))
        x = np.concatenate(xs, 0).astype(np.float32)
        y = np.concatenate(ys, 0).astype(np.int8)                     # the reference's int8 staging array truncates
        if self.patchify and len(x) > 10000:
            raise ValueError("more than 10000 patches in one batch (the reference's staging buffer)")
        if self.deep_supervision:
            y = prepare_train_dict(y, self.model_depth, self.ds_type)
        return x, y
