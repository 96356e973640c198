"""GPU bring-up diagnostics: runs every check, never stops at the first failure, prints error magnitudes.
Usage on the GPU box:  python tools/gpu_diag.py [section ...]   (sections: igemm wgrad model)

The `model` section compares the fp16 throughput mode element-wise with the fp32 oracle at the exact-mode thresholds:
it is a magnitude report, NOT a pass/fail gate - ReLU / max-pool masks flip on values that differ by one fp16 ulp, so
max-abs gradient errors there say nothing about correctness.  The gates are tests/test_gpu_ops.py (kernels on identical
fp16 inputs), tests/test_gpu_models.py (calibrated fp16 metrics) and tests/test_gpu_fp32.py (north-star tolerance)."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import segamd
import gpu_util as U

torch.manual_seed(0)
rng = np.random.default_rng(0)
FAILS = []
VERBOSE = bool(int(os.environ.get('DIAG_VERBOSE', '0')))


def report(name, err, tol):
    ok = err == err and err <= tol
    print(f"[{'ok' if ok else 'FAIL'}] {name}: err {err:.3e} (tol {tol:.1e})", flush=True)
    if not ok:
        FAILS.append(name)


def guard(fn):
    def w(*a, **k):
        try:
            fn(*a, **k)
        except Exception:
            traceback.print_exc()
            FAILS.append(fn.__name__ + str(a))
            try:
                torch.cuda.synchronize()
            except Exception:
                traceback.print_exc()
    return w


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@guard
def igemm_case(tag, N, H, W, cins, cout, k, kind, strides, tile, act=None, bias=True, stats=False):
    xs = [torch.randn(N, H, W, c) for c in cins]
    if kind == "convT":
        kern = rng.standard_normal((k[0], k[1], cout, sum(cins))).astype(np.float32) * 0.1
    else:
        kern = rng.standard_normal((k[0], k[1], sum(cins), cout)).astype(np.float32) * 0.1
    b = rng.standard_normal(cout).astype(np.float32) if bias else None
    out, st, pad = U.igemm_conv(xs, kern, b, kind, strides, act, 0.3, stats, tile)
    ref = U.ref_conv(xs, kern, b, kind, strides)
    pre = ref
    ref = U.K.activation(ref, act, 0.3)
    report(f"igemm {tag} tile={tile}", rel(out, ref), 4e-3)
    if pad != 0.0:
        report(f"igemm {tag} pad channels nonzero", pad, 0.0)
    if stats:
        s_ref = torch.stack([pre.sum((0, 1, 2)), (pre * pre).sum((0, 1, 2))])
        report(f"igemm {tag} stats", rel(st[:, :cout], s_ref), 1e-3)


@guard
def wgrad_case(tag, N, H, W, cins, cout, k, kind, strides, variant, splitk=None, tile=None):
    xs = [U.h16(torch.randn(N, H, W, c)) for c in cins]
    kshape = (k[0], k[1], cout, sum(cins)) if kind == "convT" else (k[0], k[1], sum(cins), cout)
    kern = torch.zeros(kshape, requires_grad=True)
    y = (U.K.conv2d_transpose_same if kind == "convT" else U.K.conv2d_same)(torch.cat(xs, -1), kern, None, strides)
    dz = U.h16(torch.randn_like(y))
    (y * dz).sum().backward()
    got = U.igemm_wgrad(xs, dz, kshape, kind, strides, variant, splitk, tile)
    report(f"wgrad {tag} v{variant} sk={splitk} tile={tile}", rel(torch.from_numpy(got), kern.grad), 3e-3)


def sec_igemm():
    for tile in [(128, 128), (128, 64), (64, 128), (64, 64)]:
        igemm_case("conv3x3 32->64", 2, 16, 16, [32], 64, (3, 3), "conv", (1, 1), tile, "relu", True, True)
    igemm_case("conv3x3 3->64 (first layer)", 2, 16, 16, [3], 64, (3, 3), "conv", (1, 1), None, None, True, True)
    igemm_case("conv3x3 [64,32,64]->128 concat", 2, 16, 24, [64, 32, 64], 128, (3, 3), "conv", (1, 1), None, "relu", True, True)
    igemm_case("conv3x3 ragged 13x11 40->24", 3, 13, 11, [40], 24, (3, 3), "conv", (1, 1), None, "leaky_relu")
    igemm_case("conv1x1 s2 64->32", 2, 16, 16, [64], 32, (1, 1), "conv", (2, 2), None)
    igemm_case("conv1x1 64->1", 2, 16, 16, [64], 1, (1, 1), "conv", (1, 1), None, "sigmoid")
    igemm_case("conv 1D k3 16->16", 4, 1, 128, [16], 16, (1, 3), "conv", (1, 1), None, None, True, True)
    igemm_case("convT4x4s2 64->32", 2, 8, 8, [64], 32, (4, 4), "convT", (2, 2), None, "leaky_relu", True, True)
    igemm_case("convT4x4s2 128->64 big", 2, 16, 16, [128], 64, (4, 4), "convT", (2, 2), (128, 64), "leaky_relu")
    igemm_case("convT 1D k2s2 32->16", 4, 1, 64, [32], 16, (1, 2), "convT", (1, 2), None, None, True, True)
    igemm_case("conv3x3 256->256 8x8", 4, 8, 8, [256], 256, (3, 3), "conv", (1, 1), None, "relu", True, True)


def sec_wgrad():
    for v in (0, 1):
        wgrad_case("conv3x3 32->64", 2, 16, 16, [32], 64, (3, 3), "conv", (1, 1), v, 1, (64, 64))
        wgrad_case("conv3x3 32->64", 2, 16, 16, [32], 64, (3, 3), "conv", (1, 1), v, 4, (64, 64))
        wgrad_case("conv3x3 [64,32,64]->128", 2, 16, 24, [64, 32, 64], 128, (3, 3), "conv", (1, 1), v, 3, (128, 128))
        wgrad_case("conv3x3 [64,32]->128 t128x64", 2, 16, 16, [64, 32], 128, (3, 3), "conv", (1, 1), v, 2, (128, 64))
        wgrad_case("conv3x3 3->64", 2, 16, 16, [3], 64, (3, 3), "conv", (1, 1), v)
        wgrad_case("conv3x3 ragged 40->24", 3, 13, 11, [40], 24, (3, 3), "conv", (1, 1), v)
        wgrad_case("convT4x4s2 64->32", 2, 8, 8, [64], 32, (4, 4), "convT", (2, 2), v)
        wgrad_case("conv1x1s2 64->32", 2, 16, 16, [64], 32, (1, 1), "conv", (2, 2), v)
        wgrad_case("convT 1D k2s2 32->16", 4, 1, 64, [32], 16, (1, 2), "convT", (1, 2), v)


@guard
def model_case(tag, model, x, targets, loss, fwd_tol=2e-3, grad_tol=3e-2):
    t0 = time.time()
    model.build(x.shape[0], dtype="f16")
    res32 = U.compare_model(model, x, targets, loss, verbose=False, emulate_fp16=False)
    report(f"model {tag} vs FP32 oracle: forward outs", max(res32["outs"]), 5e-3)
    print("   fp32-oracle outs err", ["%.3e" % e for e in res32["outs"]], " worst pgrad rel", "%.3e" % U.grad_rel_err(res32))
    res = U.compare_model(model, x, targets, loss, verbose=VERBOSE, emulate_fp16=True)
    fwd_tol = 1e-3
    worst_node = max((e for _n, e, _m in res["nodes"]), default=0.0)
    report(f"model {tag} forward outs", max(res["outs"]), fwd_tol)
    report(f"model {tag} worst node (abs)", worst_node, 5e-2)
    report(f"model {tag} loss", abs(res["loss"][0] - res["loss"][1]) / max(1e-6, abs(res["loss"][1])), 2e-3)
    report(f"model {tag} worst param-grad rel err", U.grad_rel_err(res), grad_tol)
    wa = max((e / (m + 1e-30) for _n, e, m in res["agrads"]), default=0.0)
    report(f"model {tag} worst activation-grad rel err", wa, 5e-2)
    if not VERBOSE:
        worst = sorted(res["grads"].items(), key=lambda kv: -(kv[1][0] / (kv[1][1] + 1e-4 * res["gmax"])))[:4]
        for k, (e, d) in worst:
            print(f"   pgrad {k:40s} abs err {e:.3e} / max {d:.3e}")
        for n_, e, m_ in sorted(res["agrads"], key=lambda t: -t[1] / (t[2] + 1e-30))[:4]:
            print(f"   agrad {n_:36s} max|err| {e:.3e}  (ref max {m_:.3e})")
    print(f"   ({time.time() - t0:.1f}s)")


def sec_model():
    B = lambda *a, **k: segamd.unet_model_builder(*a, train_mode="from_scratch", **k).ResNet50()
    x = torch.rand(2, 32, 32, 3)
    m = B("UNet", 32, 32, 16, 2, num_channels=3, output_nums=4, final_activation="softmax")
    tg = torch.nn.functional.one_hot(torch.randint(0, 4, (2, 32, 32)), 4).float()
    model_case("UNet d2 w16 softmax/cce", m, x, [tg], "cce")
    m = B("UNetPP", 32, 32, 16, 3, num_channels=3, ds=1)
    model_case("UNetPP d3 w16 ds mse", m, x, [torch.rand(2, 32, 32, 1) for _ in range(4)], "mse")
    m = B("UNet", 32, 32, 16, 2, num_channels=1, lstm=1)
    model_case("UNet+ConvLSTM d2 w16 bce", m, torch.rand(2, 32, 32, 1), [(torch.rand(2, 32, 32, 1) > 0.5).float()], "bce")
    m = B("UNet", 32, 32, 16, 2, num_channels=3, ag=1, ds=1)
    model_case("UNet+AG+DS d2 w16 mse", m, x, [torch.rand(2, 32, 32, 1), torch.rand(2, 16, 16, 1), torch.rand(2, 8, 8, 1)], "mse")
    m = B("UNet", 64, 64, 64, 3, num_channels=3)
    model_case("UNet d3 w64 64x64 bce", m, torch.rand(2, 64, 64, 3), [(torch.rand(2, 64, 64, 1) > 0.5).float()], "bce")
    m1 = segamd.UNet1D(256, 2, 1, 16, 3, "Classification", output_nums=2, ds=0).UNet()
    model_case("1D UNet d2 w16 cce", m1, torch.rand(4, 256, 1),
               [torch.nn.functional.one_hot(torch.randint(0, 2, (4, 256)), 2).float()], "cce")


if __name__ == "__main__":
    secs = sys.argv[1:] or ["igemm", "wgrad", "model"]
    print("device:", torch.cuda.get_device_name(0))
    for s in secs:
        print(f"===== {s} =====", flush=True)
        globals()["sec_" + s]()
    print("over threshold (informational for the model section, see the module docstring):", FAILS if FAILS else "none")
    sys.exit(1 if FAILS else 0)
