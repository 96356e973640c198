"""Geometry and launch planning of the implicit-GEMM kernels: the seg_geom descriptions of Conv / Conv-transpose / their data
gradients (Keras 'same' padding), the GEMM weight layout, and the tile / variant / split choices - each rule tied to a measured
sweep (tools/op_bench.py).  Pure host arithmetic: no device memory, no kernel calls (tests/test_lowering.py, tests/test_cabi.py keep
the mirrors of the library's own eligibility functions in step)."""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from . import _ffi
from . import flags as F
from ._ffi import Geom


def _r8(v):
    return (v + 7) // 8 * 8


def _r32(v):
    return (v + 31) // 32 * 32


def _cdiv(a, b):
    return -(-a // b)


# ------------------------------------------------------------------------------------------------ geometry
def _same_pad(n_in, k, s):
    out = _cdiv(n_in, s)
    total = max((out - 1) * s + k - n_in, 0)
    return total // 2


def _fill_geom(g: Geom, N, Hin, Win, Hl, Wl, isy, isx, Hout, Wout, osy, osx, classes, T_total):
    """classes: list of (oy, ox, [(dy, dx, widx), ...]) with the same tap count each."""
    g.N, g.Hin, g.Win, g.Hl, g.Wl, g.isy, g.isx = N, Hin, Win, Hl, Wl, isy, isx
    g.Hout, g.Wout, g.osy, g.osx = Hout, Wout, osy, osx
    g.ncls, g.ntaps, g.T_total = len(classes), len(classes[0][2]), T_total
    assert all(len(c[2]) == g.ntaps for c in classes) and g.ncls <= _ffi.MAX_CLS
    if g.ncls * g.ntaps > _ffi.MAX_TAPS:
        raise NotImplementedError(f"a convolution of {g.ncls * g.ntaps} taps: the implicit-GEMM geometry (seg_geom) holds at most "
                                  f"SEG_MAX_TAPS = {_ffi.MAX_TAPS} (1-D kernel_size <= {_ffi.MAX_TAPS}, 2-D kernels up to 5x5)")
    for ci, (oy, ox, taps) in enumerate(classes):
        g.oy[ci], g.ox[ci] = oy, ox
        for t, (dy, dx, wi) in enumerate(taps):
            e = ci * g.ntaps + t
            g.tap_dy[e], g.tap_dx[e], g.tap_w[e] = dy, dx, wi
    return g


def geom_conv(N, H, W, kh, kw, sh, sw) -> Geom:
    """Conv2D(padding='same'): out(i,j) = sum_k in(i*s + k - pad_before)."""
    pt, pl = _same_pad(H, kh, sh), _same_pad(W, kw, sw)
    Ho, Wo = _cdiv(H, sh), _cdiv(W, sw)
    taps = [(ky - pt, kx - pl, ky * kw + kx) for ky in range(kh) for kx in range(kw)]
    return _fill_geom(Geom(), N, H, W, Ho, Wo, sh, sw, Ho, Wo, 1, 1, [(0, 0, taps)], kh * kw)


def geom_conv_dgrad(N, H, W, kh, kw, sh, sw) -> Geom:
    """Data gradient of the conv above, as a gather over dz.  stride 1: dx(y) = sum_k dz(y + pad - k) w[k];
    1x1/stride s: dx(i*s) = dz(i) w (other pixels untouched -> caller pre-zeroes)."""
    Ho, Wo = _cdiv(H, sh), _cdiv(W, sw)
    if sh == 1 and sw == 1:
        pt, pl = _same_pad(H, kh, 1), _same_pad(W, kw, 1)
        taps = [(pt - ky, pl - kx, ky * kw + kx) for ky in range(kh) for kx in range(kw)]
        return _fill_geom(Geom(), N, Ho, Wo, H, W, 1, 1, H, W, 1, 1, [(0, 0, taps)], kh * kw)
    if kh == 1 and kw == 1:
        return _fill_geom(Geom(), N, Ho, Wo, Ho, Wo, 1, 1, H, W, sh, sw, [(0, 0, [(0, 0, 0)])], 1)
    raise NotImplementedError("data gradient of a strided conv with a kernel larger than 1x1 is not on the hot path")


def geom_convT(N, H, W, kh, kw, sh, sw) -> Geom:
    """Conv2DTranspose(padding='same'): Y = s*i + k - pb  ->  sub-pixel class py = Y mod s gathers the taps
    k with (py + pb - k) % s == 0 at input offset (py + pb - k) / s."""
    pbh, pbw = max(kh - sh, 0) // 2, max(kw - sw, 0) // 2
    classes = []
    for py in range(sh):
        for px in range(sw):
            taps = []
            for ky in range(kh):
                if (py + pbh - ky) % sh:
                    continue
                for kx in range(kw):
                    if (px + pbw - kx) % sw:
                        continue
                    taps.append(((py + pbh - ky) // sh, (px + pbw - kx) // sw, ky * kw + kx))
            classes.append((py, px, taps))
    return _fill_geom(Geom(), N, H, W, H, W, 1, 1, H * sh, W * sw, sh, sw, classes, kh * kw)


def geom_convT_dgrad(N, H, W, kh, kw, sh, sw) -> Geom:
    """dx(i) = sum_k dY(s*i + k - pb) w[k]  (a strided gather over the up-sampled gradient)."""
    pbh, pbw = max(kh - sh, 0) // 2, max(kw - sw, 0) // 2
    taps = [(ky - pbh, kx - pbw, ky * kw + kx) for ky in range(kh) for kx in range(kw)]
    return _fill_geom(Geom(), N, H * sh, W * sw, H, W, sh, sw, H, W, 1, 1, [(0, 0, taps)], kh * kw)


def pack_gemm_weight(kern: np.ndarray, cin_l: Sequence[int], woff: Sequence[int], n_rows: int, Kpad: int,
                     transposed: bool) -> np.ndarray:
    """Keras kernel (kh,kw,Cin,Cout) [conv, ConvLSTM] or (kh,kw,Cout,Cin) [transposed conv] -> GEMM operand
    [n_rows][T][Kpad]: segment s of the (virtual) concat occupies K columns woff[s] .. woff[s]+cin_l[s]."""
    T = kern.shape[0] * kern.shape[1]
    k2 = np.asarray(kern, np.float32).reshape(T, kern.shape[2], kern.shape[3])
    cout = kern.shape[2] if transposed else kern.shape[3]
    E = np.zeros((n_rows, T, Kpad), np.float32)
    lo = 0
    for s, c in enumerate(cin_l):
        blk = k2[:, :, lo:lo + c].transpose(1, 0, 2) if transposed else k2[:, lo:lo + c, :].transpose(2, 0, 1)
        E[:cout, :, woff[s]:woff[s] + c] = blk
        lo += c
    return E


def unpack_gemm_weight(flat: np.ndarray, shape, cin_l, woff, n_rows, Kpad, transposed) -> np.ndarray:
    T = shape[0] * shape[1]
    cout = shape[2] if transposed else shape[3]
    E = np.asarray(flat, np.float32).reshape(n_rows, T, Kpad)
    out_k = np.zeros((T, shape[2], shape[3]), np.float32)
    lo = 0
    for s, c in enumerate(cin_l):
        blk = E[:cout, :, woff[s]:woff[s] + c]
        if transposed:
            out_k[:, :, lo:lo + c] = blk.transpose(1, 0, 2)
        else:
            out_k[:, lo:lo + c, :] = blk.transpose(1, 2, 0)
        lo += c
    return out_k.reshape(shape)


def pick_tile_fwd(M, n_rows, ncls, ktot=1 << 30):
    """(tile_m, tile_n).  256x64, 128x128 and 128x64 run on the LDS-DMA pipelined kernel (fp16); the 64-row tiles are
    the register-staged kernel for problems too small to fill 256 CUs otherwise.  ktot = taps x K channels: with a
    short reduction (the 2x2 transposed convs) the smaller tile's extra blocks per CU hide more of the prologue /
    epilogue latency than the larger tile saves in weight re-reads (measured, tools/op_bench.py)."""
    blocks = lambda tm, tn: ncls * _cdiv(M, tm) * _cdiv(n_rows, tn)
    if n_rows <= 64:
        if blocks(256, 64) >= 512 and ktot >= 1024:
            return 256, 64
        return (128, 64) if blocks(128, 64) >= 256 else (64, 64)
    if blocks(128, 128) >= 384:
        return 128, 128
    if blocks(128, 64) >= 128:          # even at half a block per CU the LDS-DMA kernel beats the 64-row register-staged one 2x (8x8 level)
        return 128, 64
    return (64, 128) if blocks(64, 128) >= 256 else (64, 64)


def wgrad_taps_per_block(tile_m, tile_n, ntaps, variant=0):
    """Mirror of seg_wgrad_taps_per_block() (csrc/igemm_wgrad.hip); tests/test_cabi.py keeps the two in step."""
    if variant != 0 or tile_m != 64:
        return 1
    if tile_n == 64:
        return ntaps if ntaps in (9, 4, 3) else 1
    if tile_n == 128:
        return 4 if ntaps == 4 else 3 if ntaps in (9, 3) else 1
    return 1


def fwd_halo_patch(g: Geom) -> int:
    """Mirror of seg_fwd_halo_patch() (csrc/igemm_halo.hip): patch width of the halo-tile forward / data-gradient kernel
    (variant 3; 32 = 8x32-pixel patches, 16 = 16x16), 0 when the geometry is neither a stride-1 'same' convolution with
    the nine tap offsets {-1,0,1}^2 (in any order) nor a 4x4 / stride-2 transposed convolution (four sub-pixel classes
    of 2x2 taps), or the image is not tiled by those patches."""
    if g.isy == 2:         # data gradient of the 4x4 / stride-2 transposed conv: four parity planes of the stride-2 input
        if (g.ncls, g.ntaps, g.isx, g.osy, g.osx) != (1, 16, 2, 1, 1) or (g.Hin, g.Win, g.Hout, g.Wout) != (2 * g.Hl, 2 * g.Wl, g.Hl, g.Wl):
            return 0
        if g.oy[0] != 0 or g.ox[0] != 0 or any(g.tap_dy[t] != (t >> 2) - 1 or g.tap_dx[t] != (t & 3) - 1 for t in range(16)):
            return 0
        return 32 if g.Wl % 32 == 0 and g.Hl % 8 == 0 else 16 if g.Wl % 16 == 0 and g.Hl % 16 == 0 else 0
    if (g.isy, g.isx) != (1, 1) or (g.Hl, g.Wl) != (g.Hin, g.Win):
        return 0
    if g.ncls == 1 and g.ntaps == 9:
        if (g.osy, g.osx, g.Hout, g.Wout, g.oy[0], g.ox[0]) != (1, 1, g.Hin, g.Win, 0, 0):
            return 0
        if {(g.tap_dy[t], g.tap_dx[t]) for t in range(9)} != {(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)}:
            return 0
    elif g.ncls == 4 and g.ntaps == 4:
        if (g.osy, g.osx, g.Hout, g.Wout) != (2, 2, 2 * g.Hin, 2 * g.Win):
            return 0
        if any(not (0 <= g.oy[c] <= 1 and 0 <= g.ox[c] <= 1) for c in range(4)):
            return 0
        if any(abs(g.tap_dy[e]) > 1 or abs(g.tap_dx[e]) > 1 for e in range(16)):
            return 0
    else:
        return 0
    if g.Win % 32 == 0 and g.Hin % 8 == 0:
        return 32
    if g.Win % 16 == 0 and g.Hin % 16 == 0:
        return 16
    return 0


LDS_DMA_LIMIT = 0x78000000       # mirror of seg_lds_dma_operand_limit() (tests/test_cabi.py keeps the two in step)
LDS_DMA_MAX_SHIFT = 0x80000000 - LDS_DMA_LIMIT      # largest tap byte shift of the per-tap LDS-DMA kernels (igemm_common.h)
TAIL_ROW_LIMIT = 2048            # mirror of seg_tail_row_limit(): widest BN(ReLU(a + b)) tail the fused tail kernels take (same test)


def fits_lds_dma(g: Geom, seg_cs: Sequence[int], w_elems: int, f16: bool, dz_elems: int = 0) -> bool:
    """Every operand the LDS-DMA kernels would address through a buffer resource - the input segments (N x Hin x Win pixels of seg_cs
    channels each), the weight operand, for a weight gradient also dz - is below the 32-bit offset limit of those kernels."""
    esz = 2 if f16 else 4
    pix = g.N * g.Hin * g.Win
    # ... and the largest tap shift of the per-tap kernels keeps a masked lane (offset 2 GiB) beyond every operand: below 128 MiB
    shift = max([abs(g.tap_dy[e] * g.Win + g.tap_dx[e]) for e in range(g.ncls * g.ntaps)] + [0])
    return (all(pix * cs * esz < LDS_DMA_LIMIT and shift * cs * esz < LDS_DMA_MAX_SHIFT for cs in seg_cs) and w_elems * esz < LDS_DMA_LIMIT
            and dz_elems * esz < LDS_DMA_LIMIT)


def pick_fwd(M, n_rows, g: Geom, ktot, f16: bool):
    """(tile_m, tile_n, variant) of a forward / data-gradient launch.  f16 = False: the parity mode (split-pair tensors, csrc
    igemm_fwd_split.hip / igemm_halo.hip SPLIT): the same kernel families with two accumulator sets per wave, hence the halo
    tile is 256 x 64 only and the per-tap LDS-DMA kernel has no 64-deep K step."""
    if F.FWD_HALO and fwd_halo_patch(g):
        tn = 64 if (n_rows <= 64 or not f16) else 128
        if not f16 and n_rows <= 32 and F.FWD_HALO_NARROW:
            tn = 32           # parity mode, at most 32 output channels: one 32-channel tile per wave (half the MFMAs of the 64-channel tile)
        # (parity mode: the halo kernel wins down to half a block per CU - same-box A/B of the threshold, 384 against 128 blocks:
        # C3 50.05 -> 49.8 ms, C2 17.20 -> 16.78 ms per step)
        if g.ncls * (M // 256) * _cdiv(n_rows, tn) >= (F.FWD_HALO_MIN_BLOCKS if f16 else F.FWD_HALO_MIN_BLOCKS_F32):
            return 256, tn, 3
    tm, tn = pick_tile_fwd(M, n_rows, g.ncls, ktot)
    if not f16 and n_rows > 64 and g.ncls * _cdiv(M, 128) * _cdiv(n_rows, 128) >= 256:
        tm, tn = 128, 128          # parity mode, per-tap kernel at one block per CU: X41 forward 499 -> 444 us (tools/op_bench.py --dtype f32)
    return tm, tn, 0


def plan_fwd(M, n_rows, g: Geom, ktot, f16: bool, seg_cs: Optional[Sequence[int]] = None, w_elems: int = 0):
    """(tile_m, tile_n, variant, splitk) of a forward / data-gradient launch: pick_fwd, then the split of the reduction
    (seg_igemm_fwd_args.splitk) for grids that leave CUs idle - the 8x8 / 16x16 levels of a depth-5 encoder: M = 1 024 .. 4 096
    pixels against K = 4 608 .. 18 432.  Measured on MI355X, parity mode (tools/op_bench.py --splitk, 128x128 tile): enc6 data
    gradient (64 blocks) 291 -> 129 us, latent convs (128 blocks) 308 -> 220, X41 transposed conv 318 -> 212, enc5 data gradient
    (the halo kernel at 128 blocks: 194) -> 124, X31 transposed-conv data gradient (halo, 256 blocks: 278) -> 214; a 256-block
    grid with K = 4 608 (enc5 forward) does not gain, hence the floor on the slice."""
    if seg_cs is not None and not fits_lds_dma(g, seg_cs, w_elems, f16):
        # an operand of 1.875 GiB or more (C3 beyond batch 119): the register-staged kernel with 64-bit addresses - 128-row tiles, no split
        tm, tn = pick_tile_fwd(M, n_rows, g.ncls, ktot)
        return min(tm, 128), tn, 0, 1
    tm, tn, variant = pick_fwd(M, n_rows, g, ktot, f16)
    if not F.FWD_SPLITK or g.ntaps not in (1, 2, 3, 4, 9, 16):
        return tm, tn, variant, 1
    ctn = 128 if n_rows > 64 else 64
    blocks = g.ncls * _cdiv(M, 128) * _cdiv(n_rows, ctn)
    if blocks > F.FWD_SPLITK_MAX_BLOCKS:
        return tm, tn, variant, 1
    sk = max(1, F.FWD_SPLITK_TARGET // blocks)
    kmin = F.FWD_SPLITK_MIN_K            # (throughput mode, same floor: C3 21.6 -> 21.45 ms per step; twice the floor: no change)
    while sk > 1 and ktot // sk < kmin:
        sk //= 2
    if sk <= 1:
        return tm, tn, variant, 1
    return 128, ctn, 0, min(sk, 64)


def fwd_workspace_bytes(a) -> int:
    """Mirror of seg_igemm_fwd_workspace_bytes(): accumulator images of a split reduction."""
    if a.splitk <= 1:
        return 0
    Mtot = a.g.N * a.g.Hl * a.g.Wl
    return a.splitk * a.g.ncls * _cdiv(Mtot, a.tile_m) * _cdiv(a.n_rows, a.tile_n) * a.tile_m * a.tile_n * 4


def wgrad_halo_patch(g: Geom) -> int:
    """Mirror of seg_wgrad_halo_patch() (csrc/igemm_wgrad.hip): patch width of the LDS-DMA halo-tile weight-gradient kernel
    (variant 2) for this geometry, 0 when it is neither a 3x3 / stride-1 / 'same' convolution nor a 4x4 / stride-2
    transposed convolution the kernel can tile."""
    if (g.isy, g.isx) != (1, 1) or (g.Hl, g.Wl) != (g.Hin, g.Win):
        return 0
    if g.ncls == 1 and g.ntaps == 9:
        if (g.osy, g.osx, g.Hout, g.Wout, g.oy[0], g.ox[0]) != (1, 1, g.Hin, g.Win, 0, 0):
            return 0
        if any(g.tap_dy[t] != t // 3 - 1 or g.tap_dx[t] != t % 3 - 1 for t in range(9)):
            return 0
    elif g.ncls == 4 and g.ntaps == 4:
        if (g.osy, g.osx, g.Hout, g.Wout) != (2, 2, 2 * g.Hin, 2 * g.Win):
            return 0
        for c in range(4):
            if not (0 <= g.oy[c] <= 1 and 0 <= g.ox[c] <= 1):
                return 0
            sh = [(g.tap_dy[c * 4 + t], g.tap_dx[c * 4 + t]) for t in range(4)]
            if any(abs(dy) > 1 or abs(dx) > 1 for dy, dx in sh) or len(set(sh)) != 4:
                return 0
    else:
        return 0
    if g.Win % 16 == 0 and g.Hin % 4 == 0:         # 4 x 16 patches first (smallest halo overhead that still tiles most images)
        return 16
    if g.Win % 32 == 0 and g.Hin % 2 == 0:
        return 32
    if g.Win % 8 == 0 and g.Hin % 8 == 0:
        return 8
    return 0


def wgrad_halo_class_groups(g: Geom) -> int:
    """Mirror of seg_wgrad_halo_class_groups(): 1 (conv), 2 (transposed conv whose taps follow the class-pair table), else ncls."""
    if g.ncls != 4 or g.ntaps != 4:
        return g.ncls
    for c in range(4):
        py, px = c >> 1, c & 1
        if (g.oy[c], g.ox[c]) != (py, px):
            return 4
        for t in range(4):
            iy, ix = t >> 1, t & 1
            dy = (0, -1)[iy] if py == 0 else (1, 0)[iy]
            dx = (0, -1)[ix] if px == 0 else (1, 0)[ix]
            if (g.tap_dy[c * 4 + t], g.tap_dx[c * 4 + t]) != (dy, dx):
                return 4
    return 2


def pick_split_halo(M, n_rows, seg_cs: Sequence[int], ncls=1, plane=64):
    """Patch-range split of the variant-2 kernel: fill whole rounds of F.HALO_BLOCKS co-resident blocks, keep at least 8
    patches (512 pixels) per block.  plane: input channels per block (64; parity mode 32)."""
    tiles = ncls * _cdiv(n_rows, 64) * sum(_cdiv(c, plane) for c in seg_cs)
    npatch = M // 64
    best, best_eff = 1, 0.0
    HB = F.HALO_BLOCKS_TC if ncls > 1 else F.HALO_BLOCKS
    for sk in range(1, max(1, min(npatch // 8, _cdiv(2 * HB, tiles))) + 1):
        blocks = tiles * sk
        eff = blocks / (_cdiv(blocks, HB) * HB)
        if eff > best_eff + 1e-9:
            best, best_eff = sk, eff
    return best


def pick_tile_wgrad(M, n_rows, seg_cs: Sequence[int], ncls, ntaps, variant=0, f16=True):
    if variant == 2:
        return 64, 64, pick_split_halo(M, n_rows, seg_cs, ncls, 64 if f16 else 32)
    if not f16:
        variant = 1          # (no multi-tap form in the parity mode: one tap per block)
    tm = 128 if n_rows > 64 else 64
    tn = 128 if max(seg_cs) > 64 else 64
    ntb = wgrad_taps_per_block(tm, tn, ntaps, variant)
    tiles = _cdiv(n_rows, tm) * sum(_cdiv(c, tn) for c in seg_cs) * ncls * (ntaps // ntb)
    splitk = max(1, min(_cdiv(F.WGRAD_BLOCKS, tiles), _cdiv(M, 512)))
    return tm, tn, splitk


def _tail_parts(npix: int, cs: int) -> int:
    """Partial rows (= blocks) of the tail kernels that walk tiles of RP * 4 pixels (csrc/elementwise.hip col_layout): at most
    F.BN_PARTS_BWD, never more than there are tiles."""
    chp = 1
    while chp < cs // 8:
        chp <<= 1
    tile = (256 // min(chp, 256)) * 4
    return max(1, min(F.BN_PARTS_BWD, _cdiv(npix, tile)))
