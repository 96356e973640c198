"""Host-side view of the parity mode's storage format (csrc/seg_common.h, split_t).

A tensor of the parity mode (dtype='f32') occupies 4 bytes per element like fp32, but an element is a PAIR of fp16 numbers,
x ~= hi + lo / 2048 (hi = fp16(x), lo = fp16((x - hi) * 2048): ~22 significant bits inside fp16's range), and the eight
channels of an 8-aligned group are stored as [8 hi | 8 lo] = 32 bytes, so that every 16-byte chunk is an MFMA-ready fragment.
torch sees such a buffer as float32 of the same shape (opaque bits); these two functions convert whole tensors whose last
axis is a multiple of 8.  They are used by tests and tools to hand operands to / read results from single kernels - the
engine itself converts at its boundary on the device (seg_pack_input_f32 / the fp32 output heads / seg_unpack_f32_f32)."""
import torch

SPLIT_SCALE = 2048.0


def split_pack(x: torch.Tensor) -> torch.Tensor:
    """fp32 [..., C] (C % 8 == 0) -> the split image, returned as a float32-typed tensor of the same shape."""
    assert x.dtype == torch.float32 and x.shape[-1] % 8 == 0
    x = x.contiguous()
    hi = x.to(torch.float16)
    lo = ((x - hi.to(torch.float32)) * SPLIT_SCALE).to(torch.float16)
    g = (-1, x.shape[-1] // 8, 1, 8)
    img = torch.cat([hi.reshape(g), lo.reshape(g)], dim=2).contiguous()          # [rows][groups][hi | lo][8]
    return img.view(torch.float32).reshape(x.shape)


def split_unpack(t: torch.Tensor) -> torch.Tensor:
    """the inverse: split image (float32-typed, last axis % 8 == 0) -> fp32 values."""
    assert t.dtype == torch.float32 and t.shape[-1] % 8 == 0
    h = t.contiguous().view(torch.float16).reshape(-1, t.shape[-1] // 8, 2, 8).to(torch.float32)
    return (h[:, :, 0] + h[:, :, 1] / SPLIT_SCALE).reshape(t.shape)
