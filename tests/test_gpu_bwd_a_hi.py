"""MI355X, EXPERIMENT (branch r6-2mfma-backward, written without GPU minutes at the end of round 5 - run it first thing): the halo-tile data
gradient with the incoming gradient read as its hi half (SEG_VARIANT_A_HI: two MFMAs per product).  What it must show before anything
else is built on it: same result as the three-MFMA launch to ~2^-11 of dz (elementwise, unbiased), nothing else changed."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import gpu_util as U

ffi, M = U.ffi, U.M


@pytest.mark.parametrize("N,H,W,cin,cout", [(2, 32, 32, 64, 64), (1, 64, 64, 32, 17), (2, 16, 32, 40, 96)])
def test_hi_only_gradient_operand_matches_the_three_mfma_launch(N, H, W, cin, cout):
    torch.manual_seed(0)
    x = torch.randn(N, H, W, cin)
    kern = (np.random.default_rng(0).standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    tile = (256, 64 if cout > 32 else 32)
    full, _s, pad0 = U.igemm_conv([x], kern, None, "conv", (1, 1), None, 0.0, False, tile, f32=True, variant=3)
    half, _s, pad1 = U.igemm_conv([x], kern, None, "conv", (1, 1), None, 0.0, False, tile, f32=True, variant=3 | ffi.SEG_VARIANT_A_HI)
    ref = U.ref_conv([x], kern, None, "conv", (1, 1))
    e_full = float((full - ref).abs().max() / ref.abs().max())
    e_half = float((half - ref).abs().max() / ref.abs().max())
    print(f"three MFMAs {e_full:.1e}, two MFMAs (A operand hi only) {e_half:.1e}")
    assert pad0 == 0.0 and pad1 == 0.0
    assert e_full < 2e-5
    assert 2e-5 < e_half < 2e-3          # ~2^-11 per product, averaged over the contraction: present, and no more than that
    # unbiased: the mean signed difference is far below the mean absolute difference
    d = (half - full)
    assert abs(float(d.mean())) < 0.1 * float(d.abs().mean())


@pytest.mark.parametrize("N,H,W,cin,cout,kind", [(2, 32, 32, 64, 64, "conv"), (2, 16, 32, 40, 96, "conv"), (2, 16, 16, 64, 64, "convT")])
def test_hi_only_input_operand_in_the_weight_gradient(N, H, W, cin, cout, kind):
    torch.manual_seed(1)
    x = torch.randn(N, H, W, cin)
    k = (3, 3) if kind == "conv" else (4, 4)
    s = (1, 1) if kind == "conv" else (2, 2)
    kshape = (k[0], k[1], cin, cout) if kind == "conv" else (k[0], k[1], cout, cin)
    Ho, Wo = (H, W) if kind == "conv" else (2 * H, 2 * W)
    dz = torch.randn(N, Ho, Wo, cout)
    full = U.igemm_wgrad([x], dz, kshape, kind, s, 2, None, None, f32=True)
    half = U.igemm_wgrad([x], dz, kshape, kind, s, 2 | ffi.SEG_VARIANT_A_HI, None, None, f32=True)
    e = float(np.abs(half - full).max() / np.abs(full).max())
    print(f"weight gradient, input operand hi only against three MFMAs: {e:.1e}")
    assert 1e-6 < e < 2e-3
    d = half - full
    assert abs(float(d.mean())) < 0.1 * float(np.abs(d).mean())
