import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import segamd
import gpu_util as U
torch.set_num_threads(16)
torch.manual_seed(100003 + 42)
m = segamd.BCDUNet(40, 3, 1, 12, 3, 'Regression', output_nums=1, ds=0, ae=0, ag=0, lstm=0, feature_number=40, is_transconv=True, dense_loop=1).BCDUNet()
x = torch.rand(2, 40, 1)
m.build(2, dtype="f32")
tg = [torch.rand(2, 40, 1)]
k = "enc1/a/conv/kernel"
for rep in range(3):
    r = U.compare_model(m, x, tg, "mse", impose_decisions=True)
    g = m.get_grads()[k]
    print("rep", rep, "engine grad", k, g.shape, "finite:", np.isfinite(g).all(), "gl2_imposed", r["gl2_imposed"][k], "gl2", r["gl2"][k])
    print("   engine:", g.reshape(-1)[:12])
op = next(o for o in m.ops if isinstance(o, U.M.ConvOp) and o.wname == k)
wg = op.wg_args
print("wgrad args: variant", wg.variant, "tile", wg.tile_m, wg.tile_n, "splitk", wg.splitk, "partial", bool(wg.partial), "overwrite", wg.overwrite,
      "n_rows", wg.n_rows, "Kpad", wg.Kpad, "nseg", wg.nseg, "seg_c", wg.seg_c[0], "seg_cs", wg.seg_cs[0], "T", op.T, "M", op.M)
G = m.store.G
sl = m.store.slots[k]
raw = G[sl.off:sl.off + sl.numel].cpu().numpy() if hasattr(sl, "off") else None
if raw is not None:
    print("raw G slice finite:", np.isfinite(raw).all(), "non-finite count", int((~np.isfinite(raw)).sum()), "of", raw.size)
    bad = np.where(~np.isfinite(raw))[0]
    print("   first bad flat indices:", bad[:20], " -> (row, tap, k) with Kpad", wg.Kpad, [(int(b // (op.T * wg.Kpad)), int((b // wg.Kpad) % op.T), int(b % wg.Kpad)) for b in bad[:10]])
inp = op.segs[0][0]
print("input tensor", inp.name, "cs", inp.cs, "C", inp.C, "finite:", bool(torch.isfinite(inp.values()).all()))
rawin = U.SF.split_unpack(inp.buf) if inp.buf.dtype == torch.float32 else inp.buf.float()
print("   padded channels of the input finite:", bool(torch.isfinite(rawin).all()), "max |pad|", float(rawin[..., inp.C:].abs().max()))
print("dz finite:", bool(torch.isfinite(U.SF.split_unpack(op.out.grad)).all()))
dzv = U.SF.split_unpack(op.out.grad).cpu()
bad = (~torch.isfinite(dzv)).nonzero()
print("dz shape", tuple(dzv.shape), "non-finite", bad.shape[0], "first:", bad[:12].tolist())
print("   channels with non-finite:", sorted(set(bad[:, 3].tolist())), "pixels:", sorted(set(bad[:, 2].tolist()))[:20])
for nm in ("scale", "shift", "mean", "invstd", "c1", "c2"):
    t = getattr(op, nm).cpu()
    print("  ", nm, "finite", bool(torch.isfinite(t).all()), t[:16].tolist())
zv = op.z.values().cpu()
print("z finite", bool(torch.isfinite(zv).all()), "z[0,0,:3]", zv[0, 0, :3].tolist())
print("bpart finite", bool(torch.isfinite(op.bpart).all()))
print("ops after enc1/a:", [(type(o).__name__, getattr(getattr(o, 'node', None), 'name', getattr(o, 'bn_name', ''))) for o in m.ops[:6]])
print("grad_src", op.grad_src, "tail_sums", op.tail_sums, "bn_bwd_done", op.bn_bwd_done, "head_inj", op.head_inj, "has_dout", op.has_dout, "lazy", op.lazy)
