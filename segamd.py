"""Importable alias for the hyphenated package directory `tf-1d-2d-segmentation-end2endpipelines_amd/`.

    import segamd                    # == importlib.import_module("tf-1d-2d-segmentation-end2endpipelines_amd")
    from segamd import unet_model_builder
"""
import importlib
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
_pkg = importlib.import_module("tf-1d-2d-segmentation-end2endpipelines_amd")
sys.modules[__name__] = _pkg
