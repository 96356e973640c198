"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/seg_hip.h declares
(no compute calls - there is no GPU here)."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "tf-1d-2d-segmentation-end2endpipelines_amd"


@pytest.fixture(scope="module")
def built():
    be = importlib.import_module(PKG + ".build_ext")
    return be.build()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "seg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(seg_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(built):
    lib = ctypes.CDLL(built)
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/seg_hip.h but not exported by libseg_hip.so"


def test_ffi_table_matches_header(built):
    ffi = importlib.import_module(PKG + "._ffi")
    assert sorted(ffi.EXPORTED_SYMBOLS) == declared_symbols()
    L = ffi.lib()
    be = importlib.import_module(PKG + ".build_ext")
    assert L.seg_abi_version() == be.abi_version()          # 28 bits of the content hash of include/seg_hip.h, compiled in
    assert L.seg_last_error() is not None


def test_abi_version_ignores_comments_and_survives_a_missing_header(built, tmp_path, monkeypatch):
    """The ABI version is the hash of the header's DECLARATIONS (a comment or white-space edit must not invalidate every prebuilt
    SEG_LIB); a deployment that ships the package without include/ falls back to the version the build recorded in _abi.py, and
    with neither the loader says which file is missing (an ImportError, not a bare FileNotFoundError from open())."""
    be = importlib.import_module(PKG + ".build_ext")
    want = be.abi_version()
    hdr = open(be.ABI_HEADER).read()
    edited = tmp_path / "seg_hip.h"
    edited.write_text("/* a new comment */\n" + hdr.replace("\n", "\n   ", 3) + "\n// trailing remark\n")
    monkeypatch.setattr(be, "ABI_HEADER", str(edited))
    assert be.abi_version() == want
    edited.write_text(hdr.replace("int32_t seg_tail_row_limit(void);", "int64_t seg_tail_row_limit(void);"))
    assert be.abi_version() != want                              # a signature change IS a new version
    monkeypatch.setattr(be, "ABI_HEADER", str(tmp_path / "absent.h"))
    assert os.path.exists(be.ABI_STAMP) and be.abi_version() == want == be.recorded_abi_version()
    monkeypatch.setattr(be, "ABI_STAMP", str(tmp_path / "absent.py"))
    with pytest.raises(ImportError, match="seg_hip.h"):
        be.abi_version()


def test_struct_layout_matches_header():
    """ctypes mirrors of seg_geom / seg_igemm_*_args: field order and sizes as declared in the header."""
    ffi = importlib.import_module(PKG + "._ffi")
    assert ctypes.sizeof(ffi.Geom) == 4 * (14 + 2 * 4 + 3 * 32)
    hdr = open(os.path.join(ROOT, "include", "seg_hip.h")).read()
    for struct, cls in (("seg_igemm_fwd_args", ffi.IgemmFwdArgs), ("seg_igemm_wgrad_args", ffi.IgemmWgradArgs), ("seg_geom", ffi.Geom),
                        ("seg_transpose_desc", ffi.TransposeDesc)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r"^(const\s+)?(seg_geom|int32_t|float|void\s*\*|float\s*\*|const void\s*\*|const float\s*\*)\s*", "", decl)
            for part in decl.split(","):
                nm = re.sub(r"\[.*?\]", "", part).replace("*", "").strip()
                if nm:
                    names.append(nm)
        assert names == [f[0] for f in cls._fields_], (struct, names, [f[0] for f in cls._fields_])


def test_host_and_library_agree_on_wgrad_taps_per_block(built):
    ffi = importlib.import_module(PKG + "._ffi")
    M = importlib.import_module(PKG + ".model")
    L = ffi.lib()
    for tm in (64, 128):
        for tn in (64, 128):
            for nt in (1, 2, 3, 4, 9, 16):
                for v in (0, 1):
                    assert L.seg_wgrad_taps_per_block(tm, tn, nt, v) == M.wgrad_taps_per_block(tm, tn, nt, v)


def test_host_and_library_agree_on_wgrad_halo_eligibility(built):
    ffi = importlib.import_module(PKG + "._ffi")
    M = importlib.import_module(PKG + ".model")
    L = ffi.lib()
    geoms = [M.geom_conv(2, h, w, kh, kw, s, s) for (h, w) in ((2, 32), (4, 16), (8, 8), (6, 48), (3, 32), (16, 24), (1, 64))
             for (kh, kw) in ((3, 3), (1, 3), (1, 1)) for s in (1, 2)]
    geoms += [M.geom_convT(2, 8, 32, 4, 4, 2, 2), M.geom_conv_dgrad(2, 8, 32, 3, 3, 1, 1), M.geom_convT_dgrad(2, 8, 32, 4, 4, 2, 2)]
    seen = set()
    for g in geoms:
        got = L.seg_wgrad_halo_patch(ctypes.byref(g))
        assert got == M.wgrad_halo_patch(g)
        seen.add(got)
    assert seen == {0, 8, 16, 32}
    geoms += [M.geom_conv(2, h, w, 3, 3, 1, 1) for (h, w) in ((8, 32), (16, 16), (24, 64), (32, 48), (12, 32))]
    geoms += [M.geom_conv_dgrad(2, h, w, 3, 3, 1, 1) for (h, w) in ((8, 32), (16, 16), (4, 32))]
    geoms += [M.geom_convT(2, h, w, k, k, 2, 2) for (h, w) in ((8, 32), (16, 16), (4, 32)) for k in (4, 2)]
    geoms += [M.geom_convT_dgrad(2, h, w, k, k, 2, 2) for (h, w) in ((8, 32), (16, 16), (4, 32)) for k in (4, 2)]
    assert {L.seg_wgrad_halo_class_groups(ctypes.byref(g)) for g in geoms} == {1, 2, 4}
    assert all(L.seg_wgrad_halo_class_groups(ctypes.byref(g)) == M.wgrad_halo_class_groups(g) for g in geoms)
    seen = set()
    for g in geoms:
        got = L.seg_fwd_halo_patch(ctypes.byref(g))
        assert got == M.fwd_halo_patch(g)
        seen.add(got)
    assert seen == {0, 16, 32}
    # operand size limit of the LDS-DMA kernels: the planner routes larger operands to the register-staged kernels
    assert L.seg_lds_dma_operand_limit() == M.LDS_DMA_LIMIT == 0x78000000
    # a row of Win * cs >= 32 M elements: a tap's byte shift would carry a masked lane's offset (2 GiB) back below the operand's end
    wide = M.geom_conv(1, 8, 16384, 3, 3, 1, 1)
    assert M.fits_lds_dma(wide, [1024], 9 * 1024 * 64, False) and not M.fits_lds_dma(wide, [2048], 9 * 2048 * 64, False)
    assert L.seg_tail_row_limit() == M.TAIL_ROW_LIMIT == 2048      # widest fused BN(ReLU(a + b)) tail; wider ones take the separate passes
    g = M.geom_conv(119, 256, 256, 3, 3, 1, 1)
    assert M.fits_lds_dma(g, [64], 64 * 9 * 64, False) and not M.fits_lds_dma(M.geom_conv(120, 256, 256, 3, 3, 1, 1), [64], 64 * 9 * 64, False)
    assert M.plan_fwd(119 * 65536, 64, g, 9 * 64, False, [64], 64 * 9 * 64)[2] == 3
    assert M.plan_fwd(120 * 65536, 64, M.geom_conv(120, 256, 256, 3, 3, 1, 1), 9 * 64, False, [64], 64 * 9 * 64) == (128, 64, 0, 1)


def test_missing_library_fails_loudly(monkeypatch):
    ffi = importlib.import_module(PKG + "._ffi")
    monkeypatch.setattr(ffi, "_lib", None)
    monkeypatch.setattr(ffi, "LIB_PATH", "/nonexistent/libseg_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        ffi.lib()


def test_no_gpu_no_model():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import segamd
    m = segamd.unet_model_builder("UNet", 32, 32, 8, 2, train_mode="from_scratch").ResNet50()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.build(1, dtype="f16")


def test_workspace_queries_agree_with_the_engines_own_sizing(built):
    """the library allocates nothing - the caller asks how much a launch needs.  The engine's buffers (model.ConvOp:
    BN partial statistics, the partial tiles of the deterministic pixel split) are exactly what the queries return."""
    ffi = importlib.import_module(PKG + "._ffi")
    M = importlib.import_module(PKG + ".model")
    L = ffi.lib()
    M._DryRun.on = True
    try:
        m = importlib.import_module(PKG).unet_model_builder("UNetPP", 64, 64, 16, 2, ds=1, train_mode="from_scratch").ResNet50().build(2)
        seen = 0
        for op in m.ops:
            if not isinstance(op, M.ConvOp):
                continue
            if op.has_bn:
                assert L.seg_igemm_fwd_stats_floats(ctypes.byref(op.fwd_args)) == op.stats.numel()
            want = op.wg_partial.numel() * 4 if op.wg_partial is not None else 0
            assert L.seg_igemm_wgrad_workspace_bytes(ctypes.byref(op.wg_args)) == want
            seen += 1
        assert seen > 5
        for a in m.fwd_ws_users:             # split reductions: accumulator images of the slices
            assert L.seg_igemm_fwd_workspace_bytes(ctypes.byref(a)) == M.fwd_workspace_bytes(a)
        a = ffi.IgemmFwdArgs.from_buffer_copy(m.fwd_ws_users[0])
        a.tile_m, a.tile_n, a.splitk = 128, 128, 4
        Mtot = a.g.N * a.g.Hl * a.g.Wl
        assert L.seg_igemm_fwd_workspace_bytes(ctypes.byref(a)) == 4 * a.g.ncls * (-(-Mtot // 128)) * (-(-a.n_rows // 128)) * 128 * 128 * 4
    finally:
        M._DryRun.on = False
