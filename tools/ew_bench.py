"""Micro-benchmark of the HBM-bound kernels through the C ABI: achieved GB/s (algorithmic bytes / time) on a
level-0-sized tensor of the C3 workload (16 x 256 x 256 x 64 fp16 = 134 MB)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_util as U
from gpu_util import P, ffi, stream

N, H, W, Cc = 16, 256, 256, int(os.environ.get("EW_C", "64"))
npix = N * H * W
h = lambda: torch.randn((N, H, W, Cc), device="cuda").half()
f32 = lambda n: torch.rand(n, device="cuda") + 0.5


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
    print(f"{name:28s} {ms*1e3:8.1f} us   {nbytes/ms/1e6:8.1f} GB/s", flush=True)


x, y, dy, dx = h(), h(), h(), h()
T = x.numel() * 2
scale, shift, mean, invstd, c1, c2 = (f32(Cc) for _ in range(6))
NPARTS = int(os.environ.get("EW_PARTS", "512"))
part = torch.zeros(NPARTS * 2 * Cc, device="cuda")
pooled = torch.zeros((N, H // 2, W // 2, Cc), dtype=torch.float16, device="cuda")
gs = torch.tensor([8.0, 0.125], device="cuda")
db = torch.zeros(Cc, device="cuda")
timeit("bn_act_apply", lambda: ffi.call("seg_bn_act_apply", P(x), Cc, 0, P(y), Cc, 0, N, H, W, Cc, Cc, P(scale), P(shift), 1, 0.0, None, 1, 1, stream()), 2 * T)
timeit("bn_act_apply+pool", lambda: ffi.call("seg_bn_act_apply", P(x), Cc, 0, P(y), Cc, 0, N, H, W, Cc, Cc, P(scale), P(shift), 1, 0.0, P(pooled), 2, 2, stream()), 2.25 * T)
timeit("bn_act_bwd_reduce", lambda: ffi.call("seg_bn_act_bwd_reduce", P(dy), Cc, 0, P(x), Cc, 0, npix, Cc, P(scale), P(shift), P(mean), P(invstd), 1, 0.0, P(part), NPARTS, stream()), 2 * T)
timeit("bn_act_bwd_apply", lambda: ffi.call("seg_bn_act_bwd_apply", P(dy), Cc, 0, P(x), Cc, 0, P(dx), Cc, 0, npix, Cc, P(scale), P(shift), P(mean), P(invstd), P(c1), P(c2), 1, 0.0, 0, stream()), 3 * T)
timeit("act_bwd(+dbias)", lambda: ffi.call("seg_act_bwd", P(dy), P(y), P(dx), npix, Cc, Cc, 2, 0.3, P(db), 1.0, C.c_void_p(gs.data_ptr() + 4), stream()), 3 * T)
timeit("bn_stats", lambda: ffi.call("seg_bn_stats", P(x), npix, Cc, 0, Cc, P(part), NPARTS, Cc, 0, stream()), T)
timeit("maxpool_bwd", lambda: ffi.call("seg_maxpool_bwd", P(pooled), P(x), P(pooled), P(dx), N, H, W, Cc, 2, 2, 1, stream()), 3.5 * T)
w = torch.rand(1, Cc, device="cuda")
b = torch.zeros(1, device="cuda")
yo = torch.zeros((N, H, W, 1), device="cuda")
dyo = torch.rand((N, H, W, 1), device="cuda")
dw, dbb = torch.zeros(1, Cc, device="cuda"), torch.zeros(1, device="cuda")
hwork = torch.zeros(1024 * (Cc + 1), device="cuda")
timeit("head_fwd (64->1 sigmoid)", lambda: ffi.call("seg_head_fwd", P(x), Cc, Cc, N, H, W, 1, 1, P(w), Cc, P(b), 1, 3, P(yo), stream()), T + npix * 4)
timeit("head_bwd (dx acc + dw)", lambda: ffi.call("seg_head_bwd", P(dyo), P(yo), P(x), Cc, Cc, N, H, W, 1, 1, P(w), Cc, 1, 3, P(dx), 1, P(gs), P(dw), P(dbb), P(hwork), hwork.numel(), stream()), 3 * T + 2 * npix * 4)
pf = torch.rand(199_000_000, device="cuda")
g_, m_, v_ = torch.rand_like(pf), torch.zeros_like(pf), torch.zeros_like(pf)
p16 = torch.zeros(pf.numel(), dtype=torch.float16, device="cuda")
timeit("adam (199M params)", lambda: ffi.call("seg_adam_step", P(pf), P(g_), P(m_), P(v_), P(p16), pf.numel(), 1e-4, 0.9, 0.999, 1e-7, 1.0, stream()), pf.numel() * 30, iters=3)
rows, Tt, K = 2048, 9, 2048
wsrc = torch.rand(rows * Tt * K, device="cuda")
wt = torch.zeros(K * Tt * rows, dtype=torch.float16, device="cuda")
timeit("transpose_w 2048x9x2048", lambda: ffi.call("seg_transpose_w", P(wsrc), P(wt), rows, Tt, K, rows, stream()), rows * Tt * K * 6, iters=3)
am = torch.zeros(1, device="cuda")
timeit("absmax 1M x1 fp32", lambda: ffi.call("seg_absmax", P(dyo), dyo.numel(), P(am), stream()), dyo.numel() * 4)
timeit("act_bwd (no dbias)", lambda: ffi.call("seg_act_bwd", P(dy), P(y), P(dx), npix, Cc, Cc, 2, 0.3, None, 1.0, None, stream()), 3 * T)
timeit("act_bwd in place", lambda: ffi.call("seg_act_bwd", P(dy), P(y), P(dy), npix, Cc, Cc, 2, 0.3, None, 1.0, None, stream()), 3 * T)
timeit("act_bwd relu", lambda: ffi.call("seg_act_bwd", P(dy), P(y), P(dx), npix, Cc, Cc, 1, 0.0, None, 1.0, None, stream()), 3 * T)
