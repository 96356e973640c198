"""Micro-benchmark of the parity-mode streaming kernels of the MultiResBlock / ResPath tails through the C ABI at C4's level-0
size (8 x 512 x 512 x 32 channels, split pairs: 268 MB per tensor): achieved GB/s = algorithmic bytes / time.
    python tools/tail_bench.py [parts ...]        (partial-row / block counts to sweep; default 512)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_util as U
from gpu_util import P, ffi, stream

N, H, W, Cc = 8, 512, 512, int(os.environ.get("EW_C", "32"))
npix = N * H * W
F32 = torch.float32
t4 = lambda c=Cc: U.SF.split_pack(torch.randn((N, H, W, c), device="cuda"))
f32 = lambda n: torch.rand(n, device="cuda") + 0.5
T = npix * Cc * 4


def timeit(name, fn, nbytes, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:44s} {ms*1e3:8.1f} us   {nbytes/ms/1e6:8.1f} GB/s", flush=True)


z1, z2, y, dy, g, dz1, dz2 = (t4() for _ in range(7))
sc1, sh1, sc2, sh2, mean, invstd, c1, c2 = (f32(Cc) for _ in range(8))
for parts in [int(a) for a in sys.argv[1:]] or [512]:
    print(f"---- {parts} blocks / partial rows")
    stats = torch.zeros(parts * 2 * Cc, device="cuda")
    xs = (C.c_void_p * 4)(z1.data_ptr(), z2.data_ptr(), None, None)
    scs = (C.c_void_p * 4)(sc1.data_ptr(), sc2.data_ptr(), None, None)
    shs = (C.c_void_p * 4)(sh1.data_ptr(), sh2.data_ptr(), None, None)
    relus = (C.c_int32 * 4)(1, 1, 0, 0)
    timeit("tail_fwd (ResPath: 2 lazy inputs)", lambda: ffi.call("seg_add_act_stats_fwd_f32", xs, scs, shs, relus, 2, None, None, None, P(y), npix, Cc, Cc, 1, 0.0,
                                                                P(stats), parts, 0, 0, stream()), 3 * T)
    part = torch.zeros(parts * 2 * Cc, device="cuda")
    timeit("bn_act_bwd_reduce (BN_out: dOut, s)", lambda: ffi.call("seg_bn_act_bwd_reduce_f32", P(dy), Cc, 0, P(y), Cc, 0, npix, Cc, P(sc1), P(sh1), P(mean), P(invstd), 0, 0.0,
                                                                   P(part), parts, stream()), 2 * T)
    gp = (C.c_void_p * 4)(g.data_ptr(), None, None, None)
    lz = (ffi.LazyBN * 2)()
    lp = [torch.zeros(parts * 2 * Cc, device="cuda") for _ in range(2)]
    for d, zz, pp in zip(lz, (z1, z2), lp):
        d.z, d.scale, d.shift, d.mean, d.invstd, d.relu, d.partial = zz.data_ptr(), sc1.data_ptr(), sh1.data_ptr(), mean.data_ptr(), invstd.data_ptr(), 1, pp.data_ptr()
    timeit("fanout (LZ = 2: + z1, z2 -> g + sums)", lambda: ffi.call("seg_bn_bwd_apply_fanout_f32", P(dy), P(y), gp, 2, 0, npix, Cc, P(sc1), P(mean), P(invstd), P(c1), P(c2), 1,
                                                                    None, None, None, None, lz, 2, parts, stream()), 5 * T)
    timeit("fanout (LZ = 0)", lambda: ffi.call("seg_bn_bwd_apply_fanout_f32", P(dy), P(y), gp, 2, 0, npix, Cc, P(sc1), P(mean), P(invstd), P(c1), P(c2), 1,
                                              None, None, None, None, None, 0, 0, stream()), 3 * T)
    sets = (ffi.BnBwdSet * 2)()
    for d, zz, dd in zip(sets, (z1, z2), (dz1, dz2)):
        d.x, d.dx, d.scale, d.shift, d.mean, d.invstd, d.c1, d.c2, d.act, d.alpha = (zz.data_ptr(), dd.data_ptr(), sc1.data_ptr(), sh1.data_ptr(), mean.data_ptr(),
                                                                                     invstd.data_ptr(), c1.data_ptr(), c2.data_ptr(), 1, 0.0)
    timeit("pair apply (g, z1, z2 -> dz1, dz2)", lambda: ffi.call("seg_bn_act_bwd_apply2_f32", P(g), sets, npix, Cc, stream()), 5 * T)
    timeit("bn_act_bwd_apply (single)", lambda: ffi.call("seg_bn_act_bwd_apply_f32", P(g), Cc, 0, P(z1), Cc, 0, P(dz1), Cc, 0, npix, Cc, P(sc1), P(sh1), P(mean), P(invstd), P(c1), P(c2),
                                                        1, 0.0, 0, stream()), 3 * T)
    timeit("bn_act_apply", lambda: ffi.call("seg_bn_act_apply_f32", P(z1), Cc, 0, P(y), Cc, 0, N, H, W, Cc, Cc, P(sc1), P(sh1), 1, 0.0, None, 1, 1, stream()), 2 * T)
    # MultiResBlock: 5 + 10 + 16 channels as a virtual concatenation
    chans = (5, 10, 16)
    pcs = [t4(U.r8(c)) for c in chans]
    pgr = [torch.zeros_like(t) for t in pcs]
    cat = ffi.CatDesc()
    cat.n, off = 3, 0
    for k, (t, c) in enumerate(zip(pcs, chans)):
        cat.p[k].x, cat.p[k].dx, cat.p[k].cs, cat.p[k].C, cat.p[k].coff, cat.p[k].acc = t.data_ptr(), pgr[k].data_ptr(), t.shape[3], c, off, 0
        off += c
    pbytes = sum(t.numel() * 4 for t in pcs)
    xs1 = (C.c_void_p * 4)(z1.data_ptr(), None, None, None)
    timeit("cat statistics (gather only)", lambda: ffi.call("seg_add_act_stats_fwd_f32", None, None, None, None, 0, C.byref(cat), None, None, None, npix, Cc, 31, 0, 0.0,
                                                           P(stats), parts, 0, 0, stream()), pbytes)
    timeit("tail_fwd (MRB: 1 lazy + cat)", lambda: ffi.call("seg_add_act_stats_fwd_f32", xs1, scs, shs, relus, 1, C.byref(cat), P(sc2), P(sh2), P(y), npix, Cc, 31, 1, 0.0,
                                                           P(stats), parts, 0, 0, stream()), 2 * T + pbytes)
    timeit("fanout (MRB: cat sums + LZ = 1)", lambda: ffi.call("seg_bn_bwd_apply_fanout_f32", P(dy), P(y), gp, 2, 0, npix, Cc, P(sc1), P(mean), P(invstd), P(c1), P(c2), 1,
                                                              C.byref(cat), P(mean), P(invstd), P(part), lz, 1, parts, stream()), 4 * T + pbytes)
    timeit("cat scatter (g, pieces -> piece grads)", lambda: ffi.call("seg_cat_bn_bwd_scatter_f32", P(g), C.byref(cat), npix, Cc, P(sc1), P(mean), P(invstd), P(c1), P(c2), stream()),
           T + 2 * pbytes)
