"""MI355X-native UNet-family segmentation engine (hot path of Sakib1263/TF-1D-2D-Segmentation-End2EndPipelines).

Host side: the reference's model-builder API (`unet_model_builder`, `model_selector`, 1D `UNet`, `BCDUNet`)
producing a layer graph; device side: hand-written HIP kernels for gfx950 behind the C-ABI declared in
`include/seg_hip.h`.  Graph construction needs neither torch nor a GPU; executing a model needs both, and
fails loudly when the HIP library is missing (there is no CPU fallback in this package).
"""
from .graph import Graph, Node, ParamSpec, Sym
from .builders2d import unet_model_builder, model_selector, ENCODER_METHODS, SUPPORTED_DECODERS
from . import builders1d
from .builders1d import BCDUNet
from .builders1d import UNet as UNet1D
from . import harness
from . import data
from . import keras_map
from . import h5lite, keras_file, keras_import
from .keras_import import load_model
from .data import CustomDataGenerator, prepare_train_dict, create_patches, patchify, unpatchify
from .harness import (EarlyStopping, ModelCheckpoint, ReduceLROnPlateau, fit, masks_from_predictions, mask_grey_levels,
                      predict_directory)

__all__ = ["Graph", "Node", "ParamSpec", "Sym", "unet_model_builder", "model_selector", "ENCODER_METHODS",
           "SUPPORTED_DECODERS", "builders1d", "BCDUNet", "UNet1D", "harness", "EarlyStopping", "ModelCheckpoint",
           "ReduceLROnPlateau", "fit", "masks_from_predictions", "mask_grey_levels", "predict_directory", "data",
           "CustomDataGenerator", "prepare_train_dict", "create_patches", "patchify", "unpatchify", "keras_map", "keras_file", "keras_import", "h5lite", "load_model"]
__version__ = "0.1.0"
