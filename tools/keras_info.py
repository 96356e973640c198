#!/usr/bin/env python3
"""What a Keras checkpoint of the reference is, and whether `segamd.load_model` takes it (no GPU, no TensorFlow, no h5py):

    python tools/keras_info.py Results/UNet/Fold_1/UNet_256_Fold_1.keras [more files ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import segamd

for p in sys.argv[1:]:
    try:
        print(segamd.keras_import.describe(p))
    except (segamd.keras_file.KerasFileError, OSError) as e:
        print(e if str(e).startswith(p) else f"{p}: {e}")
