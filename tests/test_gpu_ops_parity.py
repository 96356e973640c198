"""GPU: the streaming kernels in the PARITY mode (entry points with the suffix _f32: tensors stored as split fp16 pairs, csrc
seg_common.h split_t) called one by one through the C ABI and compared with the oracle's primitives / torch autograd on the same
fp32 inputs.  The pair keeps ~22 significant bits, so every comparison here is held to ~1e-6 relative - three orders of magnitude
tighter than the fp16-mode operator tests in test_gpu_ops.py, which exercise the same kernel templates with T = fp16.
(The GEMM entry points in this mode: test_gpu_fp32.py.)"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import gpu_util as U
from gpu_util import P, ffi, stream
from oracle import keras_ref as K

ACT = ffi.ACT
TOL = 2e-6
F32 = torch.float32


def dev(x):
    """fp32 NHWC CPU tensor -> split-pair device tensor (channel stride padded to 8)."""
    return U.to_dev_nhwc(x, F32)


def val(t, Cc):
    return U.dev_values(t)[..., :Cc]


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def zeros_like_dev(N, H, W, Cc):
    return torch.zeros((N, H, W, U.r8(Cc)), dtype=F32, device="cuda")


def fdev(a):
    return torch.as_tensor(np.asarray(a, np.float32)).cuda().contiguous()


@pytest.mark.parametrize("Cc,act,pool", [(64, "relu", None), (24, None, None), (40, "relu", (2, 2)), (8, "sigmoid", None)])
def test_bn_statistics_apply_and_backward(Cc, act, pool):
    N, H, W = 2, 12, 16
    cs, npix = U.r8(Cc), N * H * W
    z = torch.randn(N, H, W, Cc) * 2.0 + 0.5
    gamma, beta = torch.rand(Cc) + 0.5, torch.randn(Cc)
    zd = dev(z)
    nparts = 7
    stats = torch.zeros(nparts * 2 * cs, device="cuda")
    ffi.call("seg_bn_stats_f32", P(zd), npix, cs, 0, cs, P(stats), nparts, cs, 0, stream())
    pad = lambda v, fill=0.0: fdev(np.concatenate([v.detach().numpy(), np.full(cs - Cc, fill, np.float32)]))
    g_d, b_d, mm, mv = pad(gamma), pad(beta), pad(torch.zeros(Cc)), pad(torch.ones(Cc), 1.0)
    mean, invstd, scale, shift = (torch.zeros(cs, device="cuda") for _ in range(4))
    ffi.call("seg_bn_finalize", P(stats), nparts, cs, npix, P(g_d), P(b_d), 1e-3, 0.99, P(mm), P(mv), 1, 1, P(mean), P(invstd), P(scale), P(shift), stream())
    y = torch.zeros_like(zd)
    pooled = torch.zeros((N, H // 2, W // 2, cs), dtype=F32, device="cuda") if pool else None
    ffi.call("seg_bn_act_apply_f32", P(zd), cs, 0, P(y), cs, 0, N, H, W, cs, Cc, P(scale), P(shift), ACT[act], 0.0, P(pooled) if pool else None,
             *(pool or (1, 1)), stream())
    zr = z.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref, _nm, _nv = K.batchnorm(zr, gr, br, torch.zeros(Cc), torch.ones(Cc), True, fused_4d=True)
    ref = K.activation(ref, act)
    torch.cuda.synchronize()
    assert rel(val(y, Cc), ref.detach()) < TOL and rel(mean[:Cc].cpu(), z.mean((0, 1, 2))) < TOL
    if pool:
        assert torch.equal(val(pooled, Cc), K.maxpool(val(y, Cc), pool))          # the pooled tensor is the max of what was STORED
    if cs > Cc:
        assert float(U.dev_values(y)[..., Cc:].abs().max()) == 0.0
    # ---- backward
    dA = torch.randn(N, H, W, Cc)
    (ref * dA).sum().backward()
    dAd = dev(dA)
    part = torch.zeros(16 * 2 * cs, device="cuda")
    c1, c2, dg, db = (torch.zeros(cs, device="cuda") for _ in range(4))
    ffi.call("seg_bn_act_bwd_reduce_f32", P(dAd), cs, 0, P(zd), cs, 0, npix, cs, P(scale), P(shift), P(mean), P(invstd), ACT[act], 0.0, P(part), 16, stream())
    ffi.call("seg_bn_bwd_finalize", P(part), 16, cs, npix, P(dg), P(db), 1.0, None, P(c1), P(c2), stream())
    dz = torch.zeros_like(zd)
    ffi.call("seg_bn_act_bwd_apply_f32", P(dAd), cs, 0, P(zd), cs, 0, P(dz), cs, 0, npix, cs, P(scale), P(shift), P(mean), P(invstd), P(c1), P(c2),
             ACT[act], 0.0, 0, stream())
    torch.cuda.synchronize()
    assert rel(val(dz, Cc), zr.grad) < 2e-5          # (cancellation in g - mean(g) - xhat mean(g xhat) amplifies the 2^-23 of the stored operands)
    assert rel(dg[:Cc].cpu(), gr.grad) < 1e-5 and rel(db[:Cc].cpu(), br.grad) < 1e-5


@pytest.mark.parametrize("f32", [True, False])
@pytest.mark.parametrize("Cc,pool,nheads", [(64, None, 1), (64, (2, 2), 2), (16, None, 2), (24, None, 0)])
def test_output_heads_ride_on_the_block_they_read(Cc, pool, nheads, f32):
    """seg_bn_act_apply_head: the one-channel heads' forward (dot product over the channels, bias, activation) out of the
    bn_act_apply pass (plain and pooling form); seg_bn_act_bwd_reduce_inj / _apply_inj: their data gradient gs * dpre * w added on
    the fly to the gradient the BatchNormalization-backward passes read - against the separate kernels (seg_head_fwd,
    seg_head_bwd writing dx first).  A channel count whose group count is not a power of two is refused."""
    sfx = "_f32" if f32 else ""
    dt = F32 if f32 else torch.float16
    N, H, W = 2, 8, 16
    cs, npix = U.r8(Cc), N * H * W
    g = torch.Generator().manual_seed(3)
    z = torch.randn(N, H, W, Cc, generator=g)
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.2
    zd = U.to_dev_nhwc(z if f32 else U.h16(z), dt)
    pad = lambda v: fdev(np.concatenate([v.numpy(), np.zeros(cs - Cc, np.float32)]))
    scd, shd = pad(sc), pad(sh)
    if nheads == 0:                                       # 3 groups of 8 channels: not a power of two
        d = ffi.HeadFwdDesc()
        d.n = 1
        w0, y0 = fdev(np.zeros(cs, np.float32)), torch.zeros(npix, device="cuda")
        d.w[0], d.y[0] = w0.data_ptr(), y0.data_ptr()
        out = torch.zeros_like(zd)
        with pytest.raises(ffi.SegHipError, match="power of two"):
            ffi.call("seg_bn_act_apply_head" + sfx, P(zd), cs, 0, P(out), cs, 0, N, H, W, cs, Cc, P(scd), P(shd), ACT["relu"], 0.0, None, 1, 1, C.byref(d), stream())
        return
    acts = [ACT["sigmoid"], ACT[None]][:nheads]
    ws = [fdev(np.concatenate([torch.randn(Cc, generator=g).numpy() * 0.3, np.zeros(cs - Cc, np.float32)])) for _ in range(nheads)]
    bs = [fdev(np.array([0.1 * (k + 1)], np.float32)) for k in range(nheads)]
    ys = [torch.full((npix,), 7.0, device="cuda") for _ in range(nheads)]
    d = ffi.HeadFwdDesc()
    d.n = nheads
    for k in range(nheads):
        d.w[k], d.b[k], d.act[k], d.y[k] = ws[k].data_ptr(), bs[k].data_ptr(), acts[k], ys[k].data_ptr()
    out = torch.zeros_like(zd)
    pooled = torch.zeros((N, H // 2, W // 2, cs), dtype=dt if not f32 else F32, device="cuda") if pool else None
    ffi.call("seg_bn_act_apply_head" + sfx, P(zd), cs, 0, P(out), cs, 0, N, H, W, cs, Cc, P(scd), P(shd), ACT["relu"], 0.0,
             P(pooled) if pool else None, *(pool or (1, 1)), C.byref(d), stream())
    # the separate kernels on the same stored tensor
    out2 = torch.zeros_like(zd)
    pooled2 = torch.zeros_like(pooled) if pool else None
    ffi.call("seg_bn_act_apply" + sfx, P(zd), cs, 0, P(out2), cs, 0, N, H, W, cs, Cc, P(scd), P(shd), ACT["relu"], 0.0,
             P(pooled2) if pool else None, *(pool or (1, 1)), stream())
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and (pooled is None or torch.equal(pooled, pooled2))
    for k in range(nheads):
        y2 = torch.zeros(npix, device="cuda")
        ffi.call("seg_head_fwd" + sfx, P(out2), cs, Cc, N, H, W, 1, 1, P(ws[k]), cs, P(bs[k]), 1, acts[k], P(y2), stream())
        torch.cuda.synchronize()
        assert float((ys[k] - y2).abs().max()) < 2e-6 * max(1.0, float(y2.abs().max())), k
    # ---- backward: injected against written-first
    dA = torch.randn(N, H, W, Cc, generator=g)
    dAd = U.to_dev_nhwc(dA if f32 else U.h16(dA), dt)
    dys = [torch.randn(npix, generator=g).cuda() for _ in range(nheads)]
    gs = fdev(np.array([4.0, 0.25], np.float32))
    mean, invstd = pad(z.mean((0, 1, 2))), pad(torch.rsqrt(z.var((0, 1, 2), unbiased=False) + 1e-3))
    inj = ffi.HeadInject()
    inj.n, inj.gs_ptr = nheads, gs.data_ptr()
    for k in range(nheads):
        inj.dy[k], inj.y[k], inj.w[k], inj.act[k] = dys[k].data_ptr(), ys[k].data_ptr(), ws[k].data_ptr(), acts[k]
    for have_dout in (True, False):
        # reference: seg_head_bwd writes / accumulates gs * dpre * w into the gradient tensor, then the plain passes
        gref = dAd.clone() if have_dout else torch.zeros_like(dAd)
        for k in range(nheads):
            ffi.call("seg_head_bwd" + sfx, P(dys[k]), P(ys[k]), P(out2), cs, Cc, N, H, W, 1, 1, P(ws[k]), cs, 1, acts[k], P(gref), 1, P(gs),
                     None, None, None, 0, stream())
        part_a, part_b = torch.zeros(16 * 2 * cs, device="cuda"), torch.zeros(16 * 2 * cs, device="cuda")
        common = (npix, cs, P(scd), P(shd), P(mean), P(invstd), ACT["relu"], 0.0)
        ffi.call("seg_bn_act_bwd_reduce" + sfx, P(gref), cs, 0, P(zd), cs, 0, *common, P(part_a), 16, stream())
        ffi.call("seg_bn_act_bwd_reduce_inj" + sfx, P(dAd) if have_dout else None, cs, 0, P(zd), cs, 0, *common, P(part_b), 16, C.byref(inj), stream())
        c1, c2 = (torch.zeros(cs, device="cuda") for _ in range(2))
        ffi.call("seg_bn_bwd_finalize", P(part_a), 16, cs, npix, None, None, 1.0, None, P(c1), P(c2), stream())
        dz_a, dz_b = torch.zeros_like(zd), torch.zeros_like(zd)
        tail = (npix, cs, P(scd), P(shd), P(mean), P(invstd), P(c1), P(c2), ACT["relu"], 0.0, 0)
        ffi.call("seg_bn_act_bwd_apply" + sfx, P(gref), cs, 0, P(zd), cs, 0, P(dz_a), cs, 0, *tail, stream())
        ffi.call("seg_bn_act_bwd_apply_inj" + sfx, P(dAd) if have_dout else None, cs, 0, P(zd), cs, 0, P(dz_b), cs, 0, *tail, C.byref(inj), stream())
        torch.cuda.synchronize()
        # (the written-first form rounds the sum to the storage format before the passes read it; the injected one adds in fp32)
        tol = 1e-5 if f32 else 4e-3
        pa, pb = part_a.reshape(16, 2, cs).sum(0), part_b.reshape(16, 2, cs).sum(0)
        assert float((pa - pb).abs().max()) < tol * float(pa.abs().max())
        va, vb = U.dev_values(dz_a), U.dev_values(dz_b)
        assert float((va - vb).abs().max()) < tol * float(va.abs().max())


@pytest.mark.parametrize("f32", [True, False])
def test_pair_apply_equals_two_single_passes(f32):
    """seg_bn_act_bwd_apply2: the dz of two Conv_Blocks that received the same dOut from ONE pass (the second set writing in place
    over dOut) - equal to two seg_bn_act_bwd_apply launches up to the contraction of the fused multiply-adds."""
    sfx = "_f32" if f32 else ""
    dt = F32 if f32 else torch.float16
    N, H, W, Cc = 2, 6, 10, 44
    cs, npix = U.r8(Cc), N * H * W
    g = torch.Generator().manual_seed(5)
    dyd = U.to_dev_nhwc(torch.randn(N, H, W, Cc, generator=g), dt)
    sets = (ffi.BnBwdSet * 2)()
    keep, singles = [], []
    for j in range(2):
        zd = U.to_dev_nhwc(torch.randn(N, H, W, Cc, generator=g), dt)
        co = [fdev(np.concatenate([torch.randn(Cc, generator=g).numpy() * 0.5 + (1.0 if k in (0, 3) else 0.0), np.zeros(cs - Cc, np.float32)])) for k in range(6)]
        act = ACT["relu"] if j == 0 else ACT[None]
        ref = torch.zeros_like(zd)
        ffi.call("seg_bn_act_bwd_apply" + sfx, P(dyd), cs, 0, P(zd), cs, 0, P(ref), cs, 0, npix, cs, *[P(t) for t in co], act, 0.0, 0, stream())
        dx = torch.zeros_like(zd) if j == 0 else dyd.clone()       # (set 1: in place over its own copy of dOut)
        keep += [zd, co, dx]
        singles.append(ref)
        d = sets[j]
        d.x, d.dx, d.scale, d.shift, d.mean, d.invstd, d.c1, d.c2, d.act, d.alpha = (zd.data_ptr(), dx.data_ptr(), *[t.data_ptr() for t in co], act, 0.0)
    # the pass reads sets[1].dx as dOut: hand it that very buffer
    ffi.call("seg_bn_act_bwd_apply2" + sfx, P(keep[5]), sets, npix, cs, stream())
    torch.cuda.synchronize()
    for got, want in ((keep[2], singles[0]), (keep[5], singles[1])):
        a, b = U.dev_values(got), U.dev_values(want)
        assert float((a - b).abs().max()) < (2e-6 if f32 else 2e-3) * float(b.abs().max())
        assert float(a[..., Cc:].abs().max()) == 0.0


def test_fused_tail_add_relu_statistics_and_fanout_backward():
    """seg_add_act_stats_fwd_f32 / seg_bn_bwd_apply_fanout_f32: BN2(ReLU(a + BN1(b))) with BN1 normalised on the fly."""
    N, H, W, Cc = 2, 8, 16, 24
    cs, npix = U.r8(Cc), N * H * W
    a, b = torch.randn(N, H, W, Cc), torch.randn(N, H, W, Cc) * 3.0 + 1.0
    s1, h1 = torch.rand(Cc) + 0.5, torch.randn(Cc)
    ad, bd = dev(a), dev(b)
    pad = lambda v: fdev(np.concatenate([v.numpy(), np.zeros(cs - Cc, np.float32)]))
    s1d, h1d = pad(s1), pad(h1)
    xs = (C.c_void_p * 4)(ad.data_ptr(), bd.data_ptr(), None, None)
    scs = (C.c_void_p * 4)(None, s1d.data_ptr(), None, None)
    shs = (C.c_void_p * 4)(None, h1d.data_ptr(), None, None)
    y = torch.zeros_like(ad)
    nparts = 5
    stats = torch.zeros(nparts * 2 * cs, device="cuda")
    ffi.call("seg_add_act_stats_fwd_f32", xs, scs, shs, None, 2, None, None, None, P(y), npix, cs, Cc, ACT["relu"], 0.0, P(stats), nparts, 0, 0, stream())
    torch.cuda.synchronize()
    s = torch.relu(a + b * s1 + h1)
    assert rel(val(y, Cc), s) < TOL
    st = stats.reshape(nparts, 2, cs).sum(0).cpu()
    assert rel(st[0, :Cc], s.sum((0, 1, 2))) < 1e-5 and rel(st[1, :Cc], (s * s).sum((0, 1, 2))) < 1e-5
    assert float(U.dev_values(y)[..., Cc:].abs().max()) == 0.0 if cs > Cc else True
    # backward of BN2 over s with the ReLU mask, fanned out to both inputs (the second accumulating)
    gamma = torch.rand(Cc) + 0.5
    sr = s.clone().requires_grad_(True)
    out, _a, _b = K.batchnorm(sr, gamma, torch.zeros(Cc), torch.zeros(Cc), torch.ones(Cc), True, fused_4d=True)
    dy = torch.randn(N, H, W, Cc)
    (out * dy).sum().backward()
    want = sr.grad * (s > 0)
    mean, var = s.mean((0, 1, 2)), s.var((0, 1, 2), unbiased=False)
    invstd = torch.rsqrt(var + 1e-3)
    xhat = (s - mean) * invstd
    c1, c2 = dy.mean((0, 1, 2)), (dy * xhat).mean((0, 1, 2))
    d1, prev = torch.zeros_like(ad), torch.randn(N, H, W, Cc)
    d2 = dev(prev)
    gp = (C.c_void_p * 4)(d1.data_ptr(), d2.data_ptr(), None, None)
    dyd, keep = dev(dy), [pad(v) for v in (gamma * invstd, mean, invstd, c1, c2)]      # (held: P() only takes the address)
    ffi.call("seg_bn_bwd_apply_fanout_f32", P(dyd), P(y), gp, 2, 0b10, npix, cs, *[P(t) for t in keep], 1, None, None, None, None, None, 0, 0, stream())
    torch.cuda.synchronize()
    assert rel(val(d1, Cc), want) < 2e-5 and rel(val(d2, Cc), want + prev) < 2e-5


@pytest.mark.parametrize("f32", [True, False])
@pytest.mark.parametrize("chans,W", [((5, 10, 16), 16), ((10, 21, 32), 24), ((3, 9), 8), ((85, 170, 256), 4)])
def test_multires_tail_over_a_virtual_concatenation(chans, W, f32):
    """BN2(ReLU(ReLU(za * sa + ha) + BN1(cat[p1, p2, p3]))) - MultiResBlock's tail (unet_variants.py:95-99) with the shortcut
    Conv_Block normalised on load and the concatenation never materialised (seg_cat_desc): the statistics of the concatenation,
    the fused forward, the fan-out backward with BN1's sums riding along, and the scatter of d cat onto the pieces, against autograd.
    Channel counts as in the MultiResUNet at W = 32 (level 0: 5 + 10 + 16; level 1: 10 + 21 + 32; level 4: 85 + 170 + 256)."""
    sfx = "_f32" if f32 else ""
    dt = F32 if f32 else torch.float16
    tol = 2e-5 if f32 else 6e-3
    N, H = 2, 6
    Cc = sum(chans)
    cs, npix = U.r8(Cc), N * H * W
    g = torch.Generator().manual_seed(7)
    pieces = [torch.randn(N, H, W, c, generator=g) * 1.5 + 0.3 for c in chans]
    za = torch.randn(N, H, W, Cc, generator=g)
    sa, ha = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    gamma1, beta1 = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.2
    gamma2 = torch.rand(Cc, generator=g) + 0.5
    dy = torch.randn(N, H, W, Cc, generator=g)
    rd = (lambda t: t) if f32 else U.h16                    # the values the kernels see
    pieces, za, dy = [rd(t) for t in pieces], rd(za), rd(dy)
    pdev = [U.to_dev_nhwc(t, dt) for t in pieces]
    pgrad = [torch.full_like(t, 3.0) for t in pdev]         # (stale contents: the scatter must overwrite whole groups, padding included)
    zad = U.to_dev_nhwc(za, dt)
    pad = lambda v, fill=0.0: fdev(np.concatenate([v.detach().numpy(), np.full(cs - Cc, fill, np.float32)]))
    cat = ffi.CatDesc()
    cat.n, off = len(chans), 0
    for k, (t, c) in enumerate(zip(pdev, chans)):
        cat.p[k].x, cat.p[k].dx, cat.p[k].cs, cat.p[k].C, cat.p[k].coff, cat.p[k].acc = t.data_ptr(), pgrad[k].data_ptr(), t.shape[3], c, off, 0
        off += c
    # ---- statistics of the concatenation (the operand of BN1's finalize)
    nparts = 6
    st1 = torch.full((nparts * 2 * cs,), 9.0, device="cuda")
    ffi.call("seg_add_act_stats_fwd" + sfx, None, None, None, None, 0, C.byref(cat), None, None, None, npix, cs, Cc, 0, 0.0, P(st1), nparts, 0, 0, stream())
    catv = torch.cat(pieces, -1)
    s1 = st1.reshape(nparts, 2, cs).sum(0).cpu()
    assert rel(s1[0, :Cc], catv.sum((0, 1, 2))) < 1e-5 and rel(s1[1, :Cc], (catv * catv).sum((0, 1, 2))) < 1e-5
    assert float(s1[:, Cc:].abs().max()) == 0.0 if cs > Cc else True
    # ... and the same rows written piece by piece from the pieces' own apply passes (n = 1, stats_ld / stats_coff): the form the engine
    # uses - here with an identity affine, so that the pass re-writes the piece and its sums are the piece's
    st3 = torch.zeros(nparts * 2 * cs, device="cuda")
    off = 0
    for t, c in zip(pdev, chans):
        pcs_ = t.shape[3]
        one, zero = fdev(np.ones(pcs_, np.float32)), fdev(np.zeros(pcs_, np.float32))
        outp = torch.full_like(t, 6.0)
        x1 = (C.c_void_p * 4)(t.data_ptr(), None, None, None)
        s1_, h1_ = (C.c_void_p * 4)(one.data_ptr(), None, None, None), (C.c_void_p * 4)(zero.data_ptr(), None, None, None)
        ffi.call("seg_add_act_stats_fwd" + sfx, x1, s1_, h1_, (C.c_int32 * 4)(0, 0, 0, 0), 1, None, None, None, P(outp), npix, pcs_, c, 0, 0.0, P(st3), nparts, cs, off, stream())
        torch.cuda.synchronize()
        assert torch.equal(U.dev_values(outp)[..., :c], U.dev_values(t)[..., :c]) and (pcs_ == c or float(U.dev_values(outp)[..., c:].abs().max()) == 0.0)
        off += c
    s3 = st3.reshape(nparts, 2, cs).sum(0).cpu()
    assert rel(s3[:, :Cc], s1[:, :Cc]) < 1e-5 and (cs == Cc or float(s3[:, Cc:].abs().max()) == 0.0)
    # ---- reference graph
    prs = [t.clone().requires_grad_(True) for t in pieces]
    zar = za.clone().requires_grad_(True)
    catr = torch.cat(prs, -1)
    mean1, var1 = catr.mean((0, 1, 2)), catr.var((0, 1, 2), unbiased=False)
    invstd1 = torch.rsqrt(var1 + 1e-3)
    bn1 = (catr - mean1) * invstd1 * gamma1 + beta1
    a = torch.relu(zar * sa + ha)
    a.retain_grad()
    s = torch.relu(a + bn1)
    mean2, var2 = s.mean((0, 1, 2)), s.var((0, 1, 2), unbiased=False)
    invstd2 = torch.rsqrt(var2 + 1e-3)
    out = (s - mean2) * invstd2 * gamma2
    (out * dy).sum().backward()
    # ---- fused forward
    sc1, sh1 = (gamma1 * invstd1).detach(), (beta1 - mean1 * gamma1 * invstd1).detach()
    keep = [pad(sa), pad(ha), pad(sc1), pad(sh1)]
    xs = (C.c_void_p * 4)(zad.data_ptr(), None, None, None)
    scs = (C.c_void_p * 4)(keep[0].data_ptr(), None, None, None)
    shs = (C.c_void_p * 4)(keep[1].data_ptr(), None, None, None)
    relus = (C.c_int32 * 4)(1, 0, 0, 0)
    y = torch.full_like(zad, 5.0)
    st2 = torch.zeros(nparts * 2 * cs, device="cuda")
    ffi.call("seg_add_act_stats_fwd" + sfx, xs, scs, shs, relus, 1, C.byref(cat), P(keep[2]), P(keep[3]), P(y), npix, cs, Cc, ACT["relu"], 0.0, P(st2), nparts, 0, 0, stream())
    torch.cuda.synchronize()
    yv = U.dev_values(y)
    assert rel(yv[..., :Cc], s.detach()) < (2e-6 if f32 else 2e-3)
    assert float(yv[..., Cc:].abs().max()) == 0.0 if cs > Cc else True
    s2 = st2.reshape(nparts, 2, cs).sum(0).cpu()
    assert rel(s2[0, :Cc], yv[..., :Cc].sum((0, 1, 2))) < 1e-5 and rel(s2[1, :Cc], (yv[..., :Cc] ** 2).sum((0, 1, 2))) < 1e-5
    # ---- fan-out backward of BN2 (+ ReLU mask) with BN1's two sums, from the exact s (so that only the backward kernels are measured)
    sd, dyd = U.to_dev_nhwc(rd(s.detach()), dt), U.to_dev_nhwc(dy, dt)
    sv = U.dev_values(sd)[..., :Cc]
    m2, v2 = sv.mean((0, 1, 2)), sv.var((0, 1, 2), unbiased=False)
    is2 = torch.rsqrt(v2 + 1e-3)
    xh2 = (sv - m2) * is2
    c1, c2 = dy.mean((0, 1, 2)), (dy * xh2).mean((0, 1, 2))
    want_g = (gamma2 * is2 * (dy - c1 - xh2 * c2)) * (sv > 0)
    k2 = [pad(v) for v in (gamma2 * is2, m2, is2, c1, c2)]
    k1 = [pad(mean1.detach()), pad(invstd1.detach())]
    gbuf = torch.full_like(zad, 4.0)
    gp = (C.c_void_p * 4)(gbuf.data_ptr(), None, None, None)
    nb = 5
    part = torch.full((nb * 2 * cs,), 9.0, device="cuda")
    # (the dense input is a Conv_Block normalised on load: the sums of ITS BatchNormalization backward ride along - seg_lazy_bn)
    mu_a, is_a = za.mean((0, 1, 2)), torch.rsqrt(za.var((0, 1, 2), unbiased=False) + 1e-3)
    kl = [pad(mu_a), pad(is_a)]
    lpart = torch.full((nb * 2 * cs,), 9.0, device="cuda")
    lz = (ffi.LazyBN * 2)()
    lz[0].z, lz[0].scale, lz[0].shift, lz[0].mean, lz[0].invstd, lz[0].relu, lz[0].partial = (
        zad.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(), kl[0].data_ptr(), kl[1].data_ptr(), 1, lpart.data_ptr())
    ffi.call("seg_bn_bwd_apply_fanout" + sfx, P(dyd), P(sd), gp, 2, 0, npix, cs, *[P(t) for t in k2], 1, C.byref(cat), P(k1[0]), P(k1[1]), P(part),
             lz, 1, nb, stream())
    torch.cuda.synchronize()
    gv = U.dev_values(gbuf)[..., :Cc]
    assert rel(gv, want_g) < tol
    gm = gv * ((za * sa + ha) > 0)
    lp = lpart.reshape(nb, 2, cs).sum(0).cpu()
    assert rel(lp[0, :Cc], gm.sum((0, 1, 2))) < 1e-4 and rel(lp[1, :Cc], (gm * (za - mu_a) * is_a).sum((0, 1, 2))) < 1e-4
    # the same launch without the concatenation (ResPath: two Conv_Blocks normalised on load) gives the same dense result and sums
    g2 = torch.full_like(zad, 4.0)
    gp2 = (C.c_void_p * 4)(g2.data_ptr(), None, None, None)
    lz[1].z, lz[1].scale, lz[1].shift, lz[1].mean, lz[1].invstd, lz[1].relu, lz[1].partial = (
        zad.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(), kl[0].data_ptr(), kl[1].data_ptr(), 0, st1.data_ptr())      # (no ReLU: plain sums)
    lpart.fill_(9.0)
    ffi.call("seg_bn_bwd_apply_fanout" + sfx, P(dyd), P(sd), gp2, 2, 0, npix, cs, *[P(t) for t in k2], 1, None, None, None, None, lz, 2, nb, stream())
    torch.cuda.synchronize()
    assert torch.equal(U.dev_values(g2), U.dev_values(gbuf))
    lp2 = lpart.reshape(nb, 2, cs).sum(0).cpu()
    assert rel(lp2[:, :Cc], lp[:, :Cc]) < 1e-5
    lp3 = st1[:nb * 2 * cs].reshape(nb, 2, cs).sum(0).cpu()
    assert rel(lp3[0, :Cc], gv.sum((0, 1, 2))) < 1e-4 and rel(lp3[1, :Cc], (gv * (za - mu_a) * is_a).sum((0, 1, 2))) < 1e-4
    xh1 = ((catv - mean1) * invstd1).detach()
    ps = part.reshape(nb, 2, cs).sum(0).cpu()
    assert rel(ps[0, :Cc], gv.sum((0, 1, 2))) < 1e-4 and rel(ps[1, :Cc], (gv * xh1).sum((0, 1, 2))) < 1e-4
    # ---- BN1's backward onto the pieces
    b1c1, b1c2 = gv.mean((0, 1, 2)), (gv * xh1).mean((0, 1, 2))
    k3 = [pad(v) for v in ((gamma1 * invstd1).detach(), mean1.detach(), invstd1.detach(), b1c1, b1c2)]
    ffi.call("seg_cat_bn_bwd_scatter" + sfx, P(gbuf), C.byref(cat), npix, cs, *[P(t) for t in k3], stream())
    torch.cuda.synchronize()
    want_cat = (gamma1 * invstd1).detach() * (gv - b1c1 - xh1 * b1c2)
    off = 0
    for t, c, gt in zip(pgrad, chans, prs):
        got = U.dev_values(t)
        assert rel(got[..., :c], want_cat[..., off:off + c]) < tol
        if t.shape[3] > c:
            assert float(got[..., c:].abs().max()) == 0.0          # the padding channels of a freshly written gradient are zero
        off += c
    # the whole chain against autograd (the masks of s agree: s was handed over exactly)
    if f32:
        assert rel(gv, a.grad) < 5e-5
        off = 0
        for t, c, gt in zip(pgrad, chans, prs):
            assert rel(U.dev_values(t)[..., :c], gt.grad) < 1e-4
            off += c
    # accumulate form of the scatter
    for k in range(cat.n):
        cat.p[k].acc = 1
    before = [U.dev_values(t).clone() for t in pgrad]
    ffi.call("seg_cat_bn_bwd_scatter" + sfx, P(gbuf), C.byref(cat), npix, cs, *[P(t) for t in k3], stream())
    torch.cuda.synchronize()
    for t, bf in zip(pgrad, before):
        assert rel(U.dev_values(t), 2 * bf) < (1e-6 if f32 else 2e-3)


@pytest.mark.parametrize("pool", [(2, 2), (1, 2), (4, 4)])
def test_maxpool(pool):
    N, H, W, Cc = 2, 8, 16, 24
    x = torch.randn(N, H, W, Cc, requires_grad=True)
    y = K.maxpool(x, pool)
    dy = torch.randn_like(y)
    (y * dy).sum().backward()
    xd, yd = dev(x.detach()), torch.zeros((N, H // pool[0], W // pool[1], Cc), dtype=F32, device="cuda")
    ffi.call("seg_maxpool_fwd_f32", P(xd), P(yd), N, H, W, Cc, *pool, stream())
    dxd = torch.zeros_like(xd)
    dyd = dev(dy)
    ffi.call("seg_maxpool_bwd_f32", P(dyd), P(xd), P(yd), P(dxd), N, H, W, Cc, *pool, 0, stream())
    torch.cuda.synchronize()
    assert rel(val(yd, Cc), y.detach()) < TOL and rel(val(dxd, Cc), x.grad) < TOL


@pytest.mark.parametrize("f,bil", [((2, 2), 1), ((2, 2), 0), ((4, 4), 1), ((1, 2), 0)])
def test_upsample(f, bil):
    N, H, W, Cc = 2, 6, 8, 16
    x = torch.randn(N, H, W, Cc, requires_grad=True)
    y = K.upsample(x, f, "bilinear" if bil else "nearest")
    dy = torch.randn_like(y)
    (y * dy).sum().backward()
    xd, yd = dev(x.detach()), torch.zeros((N, H * f[0], W * f[1], Cc), dtype=F32, device="cuda")
    ffi.call("seg_upsample_fwd_f32", P(xd), P(yd), N, H, W, Cc, *f, bil, stream())
    dxd = torch.zeros_like(xd)
    dyd = dev(dy)
    ffi.call("seg_upsample_bwd_f32", P(dyd), P(dxd), N, H, W, Cc, *f, bil, 0, stream())
    torch.cuda.synchronize()
    assert rel(val(yd, Cc), y.detach()) < TOL and rel(val(dxd, Cc), x.grad) < 4e-6


@pytest.mark.parametrize("n,act", [(2, "relu"), (3, None), (1, "leaky_relu")])
def test_add_act(n, act):
    shape = (2, 8, 8, 16)
    xs = [torch.randn(shape, requires_grad=True) for _ in range(n)]
    y = K.activation(sum(xs), act) if act != "leaky_relu" else F.leaky_relu(sum(xs), 0.3)
    dy = torch.randn(shape)
    (y * dy).sum().backward()
    ds = [dev(x.detach()) for x in xs]
    ptrs = (C.c_void_p * 4)(*([d.data_ptr() for d in ds] + [None] * (4 - n)))
    yd = torch.zeros_like(ds[0])
    ffi.call("seg_add_act_fwd_f32", ptrs, n, P(yd), yd.numel(), yd.shape[-1], yd.shape[-1], ACT[act], 0.3, stream())
    gs = [torch.zeros_like(d) for d in ds]
    gp = (C.c_void_p * 4)(*([g.data_ptr() for g in gs] + [None] * (4 - n)))
    dyd = dev(dy)
    ffi.call("seg_add_act_bwd_f32", P(dyd), P(yd), gp, n, yd.numel(), ACT[act], 0.3, 0, stream())
    torch.cuda.synchronize()
    assert rel(val(yd, 16), y.detach()) < TOL
    for g, x in zip(gs, xs):
        assert rel(val(g, 16), x.grad) < TOL


def test_attention_multiply_and_convlstm_gates():
    N, H, W, Cc = 2, 8, 8, 24
    a = torch.randn(N, H, W, Cc, requires_grad=True)
    r = torch.rand(N, H, W, 1, requires_grad=True)
    y = a * r
    dy = torch.randn_like(y)
    (y * dy).sum().backward()
    ad, rd = dev(a.detach()), dev(r.detach())
    yd = torch.zeros_like(ad)
    ffi.call("seg_mul_bcast_fwd_f32", P(ad), P(rd), P(yd), N * H * W, Cc, 8, stream())
    da, dr = torch.zeros_like(ad), torch.zeros_like(rd)
    dyd = dev(dy)
    ffi.call("seg_mul_bcast_bwd_f32", P(dyd), P(ad), P(rd), P(da), P(dr), N * H * W, Cc, 8, 0, 0, stream())
    torch.cuda.synchronize()
    assert rel(val(yd, Cc), y.detach()) < TOL and rel(val(da, Cc), a.grad) < TOL and rel(val(dr, 1), r.grad) < 2e-6
    assert float(U.dev_values(dr)[..., 1:].abs().max()) == 0.0
    # ConvLSTM (T = 1) gates: h = hs(o) tanh(hs(i) tanh(c)), the forget gate gets a zero gradient
    Fh = 16
    z = (torch.randn(N, H, W, 4 * Fh) * 2).requires_grad_(True)
    h = K.convlstm_gates(z, Fh) if hasattr(K, "convlstm_gates") else None
    if h is None:
        hs = lambda v: torch.clamp(0.2 * v + 0.5, 0.0, 1.0)
        zi, _zf, zc, zo = z[..., :Fh], z[..., Fh:2 * Fh], z[..., 2 * Fh:3 * Fh], z[..., 3 * Fh:]
        h = hs(zo) * torch.tanh(hs(zi) * torch.tanh(zc))
    dh = torch.randn_like(h)
    (h * dh).sum().backward()
    zd, hd = dev(z.detach()), torch.zeros((N, H, W, Fh), dtype=F32, device="cuda")
    ffi.call("seg_convlstm_gate_fwd_f32", P(zd), 4 * Fh, P(hd), Fh, N * H * W, Fh, stream())
    dz = torch.full_like(zd, 9.0)
    dhd = dev(dh)
    ffi.call("seg_convlstm_gate_bwd_f32", P(dhd), Fh, P(zd), 4 * Fh, P(dz), N * H * W, Fh, stream())
    torch.cuda.synchronize()
    assert rel(val(hd, Fh), h.detach()) < 4e-6 and rel(val(dz, 4 * Fh), z.grad) < 1e-5
    assert float(val(dz, 4 * Fh)[..., Fh:2 * Fh].abs().max()) == 0.0


@pytest.mark.parametrize("Cin,Cout,act", [(64, 1, "sigmoid"), (24, 4, "softmax"), (16, 3, None)])
def test_output_head(Cin, Cout, act):
    N, H, W = 2, 8, 16
    cs = U.r8(Cin)
    x = torch.randn(N, H, W, Cin, requires_grad=True)
    kern = (torch.randn(1, 1, Cin, Cout) * 0.2).requires_grad_(True)
    b = torch.randn(Cout, requires_grad=True)
    pre = K.conv2d_same(x, kern, b)
    y = torch.softmax(pre, -1) if act == "softmax" else K.activation(pre, act)
    dy = torch.randn_like(y)
    (y * dy).sum().backward()
    xd = dev(x.detach())
    E = np.zeros((Cout, cs), np.float32)
    E[:, :Cin] = kern.detach().numpy().reshape(Cin, Cout).T
    wd, bd = fdev(E), fdev(b.detach().numpy())
    yd = torch.zeros((N, H, W, Cout), device="cuda")
    ffi.call("seg_head_fwd_f32", P(xd), cs, Cin, N, H, W, 1, 1, P(wd), cs, P(bd), Cout, ACT[act], P(yd), stream())
    dx = torch.zeros_like(xd)
    dw, db = torch.zeros(Cout * cs, device="cuda"), torch.zeros(Cout, device="cuda")
    dyd = dy.cuda().contiguous()
    ffi.call("seg_head_bwd_f32", P(dyd), P(yd), P(xd), cs, Cin, N, H, W, 1, 1, P(wd), cs, Cout, ACT[act], P(dx), 0, None,
             P(dw), P(db), None, 0, stream())
    torch.cuda.synchronize()
    assert rel(yd.cpu(), y.detach()) < 2e-6 and rel(val(dx, Cin), x.grad) < 4e-6
    assert rel(dw.reshape(Cout, cs)[:, :Cin].cpu(), kern.grad.reshape(Cin, Cout).T) < 1e-5 and rel(db.cpu(), b.grad) < 1e-5


def test_element_granular_copies_and_the_boundary_conversions():
    N, H, W = 2, 4, 8
    parts = [torch.randn(N, H, W, c) for c in (5, 10, 16)]
    dst = torch.zeros((N, H, W, 32), dtype=F32, device="cuda")
    off = 0
    for t in parts:                                   # materialised concatenate at element-granular channel offsets (MultiResBlock)
        d = dev(t)
        ffi.call("seg_copy_channels_f32", P(d), d.shape[3], 0, P(dst), 32, off, t.shape[3], N * H * W, 0, stream())
        off += t.shape[3]
    torch.cuda.synchronize()
    got = U.dev_values(dst)
    assert rel(got[..., :31], torch.cat(parts, -1)) < TOL and float(got[..., 31:].abs().max()) == 0.0
    back = dev(torch.zeros(N, H, W, 10))
    ffi.call("seg_copy_channels_f32", P(dst), 32, 5, P(back), 16, 0, 10, N * H * W, 0, stream())
    x = torch.rand(N * H * W, 3)
    xd = x.cuda()
    packed = torch.zeros((N * H * W, 8), dtype=F32, device="cuda")
    ffi.call("seg_pack_input_f32", P(xd), P(packed), N * H * W, 3, 8, None, stream())
    out = torch.zeros((N * H * W, 3), device="cuda")
    ffi.call("seg_unpack_f32_f32", P(packed), P(out), N * H * W, 3, 8, 0, stream())
    torch.cuda.synchronize()
    assert rel(val(back, 10), parts[1]) < TOL
    assert rel(out.cpu(), x) < TOL and torch.equal(packed, U.SF.split_pack(torch.cat([xd, torch.zeros(N * H * W, 5, device="cuda")], 1)))


def test_transposed_weight_operand_is_the_split_image_of_the_transpose():
    rows, T, K = 24, 9, 64
    rows_pad = 32
    w = torch.randn(rows, T, K)
    wt = torch.zeros(K * T * rows_pad, dtype=F32, device="cuda")
    wd = w.cuda().contiguous()
    ffi.call("seg_transpose_w_f32", P(wd), P(wt), rows, T, K, rows_pad, stream())
    torch.cuda.synchronize()
    want = torch.zeros(K, T, rows_pad)
    want[:, :, :rows] = w.permute(2, 1, 0)
    assert torch.equal(wt.reshape(K, T, rows_pad).cpu().view(torch.int32), U.SF.split_pack(want).view(torch.int32))
