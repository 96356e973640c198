"""Bit-reproducibility stress of the transposed-conv data-gradient launch at the deep level (16 x 16 x 16 images, 1024 <- 512
channels: X32/tconv of C3), alone and with weight-gradient launches running beside it on a second stream."""
import sys, ctypes as C, torch, numpy as np
sys.path.insert(0, "tests"); import conftest
import gpu_util as U
from gpu_util import h16, ffi, P, to_dev_nhwc, stream
def dev_f32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(3)
M = U.M
def setup(N, H, W, cin, cout, tile, variant=3):
    kshape = (4, 4, cout, cin)
    kern = (rng.standard_normal(kshape).astype(np.float32) * 0.05).astype(np.float16).astype(np.float32)
    dz = h16(torch.randn(N, 2 * H, 2 * W, cout))
    n_rows, T, Kpad = U.r8(cout), 16, U.r32(U.r8(cin))
    E = M.pack_gemm_weight(kern, [cin], [0], n_rows, Kpad, True)
    rows_pad = U.r32(n_rows)
    wt = torch.zeros(Kpad * T * rows_pad, dtype=torch.float16, device="cuda")
    Ed = dev_f32(E.reshape(-1))
    ffi.call("seg_transpose_w", P(Ed), P(wt), n_rows, T, Kpad, rows_pad, stream())
    g = M.geom_convT_dgrad(N, H, W, 4, 4, 2, 2)
    dzd = to_dev_nhwc(dz)
    cs_in = U.r8(cin)
    dx = torch.zeros((N, H, W, cs_in), dtype=torch.float16, device="cuda")
    a = ffi.IgemmFwdArgs()
    a.g, a.nseg = g, 1
    a.seg_c[0], a.seg_cs[0], a.seg_woff[0], a.seg_ptr[0] = n_rows, dzd.shape[3], 0, dzd.data_ptr()
    a.w, a.Kpad, a.n_rows, a.n_valid = wt.data_ptr(), rows_pad, cs_in, cs_in
    a.out, a.out_cs, a.out_coff = dx.data_ptr(), cs_in, 0
    a.tile_m, a.tile_n, a.variant = tile[0], tile[1], variant
    return a, dx, (wt, dzd, Ed)
def stress(name, a, dx, beside):
    side = torch.cuda.Stream()
    ffi.call("seg_igemm_fwd", C.byref(a), stream()); torch.cuda.synchronize()
    ref = dx.clone()
    bad = 0
    for rep in range(REPS):
        dx.zero_()
        if beside is not None:
            with torch.cuda.stream(side):
                beside(side.cuda_stream)
        ffi.call("seg_igemm_fwd", C.byref(a), stream())
        torch.cuda.synchronize()
        bad += 0 if torch.equal(dx, ref) else 1
    print(f"{name:56s} differing launches {bad}/{REPS}", flush=True)
# a weight-gradient launch to run beside (deep-level conv, wave-specialised kernel)
xs = [h16(torch.randn(16, 16, 16, 1024))]
dzw = h16(torch.randn(16, 16, 16, 1024))
def make_wgrad():
    import types
    holder = {}
    def launch(s):
        U.igemm_wgrad(xs, dzw, (3, 3, 1024, 1024), "conv", (1, 1), 2, 1, None, raw=True) if False else None
    return launch
a, dx, keep = setup(16, 16, 16, 1024, 512, (256, 128))
stress("tconv dgrad 16x16 1024<-512, alone", a, dx, None)
# beside: another copy of a heavy forward-type launch on the second stream
a2, dx2, keep2 = setup(16, 32, 32, 512, 256, (256, 128))
stress("tconv dgrad 16x16 1024<-512, a 32x32 launch beside", a, dx, lambda s: ffi.call("seg_igemm_fwd", C.byref(a2), C.c_void_p(s)))

for tile in ((128, 128), (128, 64)):
    a3, dx3, keep3 = setup(16, 16, 16, 1024, 512, tile, 0)
    stress(f"16-tap v2 kernel {tile[0]}x{tile[1]}, 16x16 1024<-512, alone", a3, dx3, None)
    stress(f"16-tap v2 kernel {tile[0]}x{tile[1]}, 16x16 1024<-512, launch beside", a3, dx3, lambda s: ffi.call("seg_igemm_fwd", C.byref(a2), C.c_void_p(s)))
a4, dx4, keep4 = setup(16, 8, 8, 2048, 1024, (128, 64), 0)
stress("16-tap v2 kernel 128x64, 8x8 2048<-1024, launch beside", a4, dx4, lambda s: ffi.call("seg_igemm_fwd", C.byref(a2), C.c_void_p(s)))
