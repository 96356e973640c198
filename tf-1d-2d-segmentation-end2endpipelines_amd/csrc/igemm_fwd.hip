// Implicit-GEMM convolution kernels for gfx950 (MI355X): forward / data-gradient (seg_igemm_fwd) and weight
// gradient (seg_igemm_wgrad).  fp16 operands, fp32 accumulate on v_mfma_f32_32x32x16_f16.
//
// Replaces the TensorFlow kernels behind tf.keras.layers.Conv2D / Conv2DTranspose / ConvLSTM2D (input conv)
// of the reference (TensorFlow/2DCNN/models/unet_variants.py:9,19,148; 1D: 1DCNN/Models/unet_variants.py:55,104).
//
// Tiling (v1): 256 threads = 4 waves in a 2x2 arrangement; block tile BM x BN (128/64 each) x BK = 32 per step,
// register-staged global->LDS double buffering (issue loads for step k+1, MFMA on step k, then write LDS);
// LDS rows are 64 B (32 halves) with the 16-B chunk index XOR-swizzled by (row >> 2) & 3 so that every
// ds_read_b128 16-lane group touches 16 distinct bank quads.  The epilogue stages each 32x32 accumulator
// tile through a per-wave LDS slab so that global stores are 16 B per lane along the channel axis.
// Parity mode (SPLIT): operands are stored as split pairs (igemm_common.h) - the loaders see an fp16 tensor with twice the
// channels, a 64-B LDS row is 16 real K values (hi / lo chunks), three fp16 MFMAs per product into two accumulator sets.
#include "igemm_common.h"
#include "igemm_fwd_kernels.h"

namespace {
// explicit instantiations (the launchers below live outside this namespace)
template __global__ void igemm_fwd2_kernel<128, 128, 1, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 128, 1, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 128, 2, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 128, 2, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 128, 3, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 128, 3, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 128, 4, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 128, 4, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 128, 9, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 128, 9, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 128, 16, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 128, 16, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<256, 64, 1, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<256, 64, 1, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<256, 64, 2, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<256, 64, 2, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<256, 64, 3, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<256, 64, 3, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<256, 64, 4, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<256, 64, 4, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<256, 64, 9, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<256, 64, 9, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<256, 64, 16, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<256, 64, 16, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 64, 1, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 64, 1, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 64, 2, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 64, 2, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 64, 3, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 64, 3, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 64, 4, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 64, 4, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 64, 9, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 64, 9, 64, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 64, 16, 32, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 64, 16, 64, 2>(const seg_igemm_fwd_args);
// small grids (about one block per CU: the 8x8 / 16x16 levels with 1024-2048 channels): 64-deep K steps AND a 3-stage ring
template __global__ void igemm_fwd2_kernel<128, 64, 9, 64, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd2_kernel<128, 64, 16, 64, 3>(const seg_igemm_fwd_args);

}  // namespace

// ---- v2 (LDS-DMA) dispatch
template <int BM, int BN, int BKE, int NS>
static void launch_fwd2_nt(const seg_igemm_fwd_args& a, dim3 grid, size_t lds, hipStream_t st) {
  switch (a.g.ntaps) {
    case 1: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 1, BKE, NS>), grid, dim3(256), lds, st, a); break;
    case 2: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 2, BKE, NS>), grid, dim3(256), lds, st, a); break;
    case 3: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 3, BKE, NS>), grid, dim3(256), lds, st, a); break;
    case 4: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 4, BKE, NS>), grid, dim3(256), lds, st, a); break;
    case 9: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 9, BKE, NS>), grid, dim3(256), lds, st, a); break;
    default: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 16, BKE, NS>), grid, dim3(256), lds, st, a); break;
  }
}

// K-step depth of the LDS-DMA kernel.  64 channels = 128-B rows (whole cache lines per pixel), half the barriers, but a
// 2-stage ring (one K step of loads in flight) and, for the 128x128 tile, 2 instead of 3 blocks per CU.  Measured on
// MI355X (tools/op_bench.py): it wins when the K loop is long and the grid deep enough that other blocks of the CU cover a
// block's load latency; 32 wins for short K loops (transposed convs, data gradients of 64-channel layers, whose
// read-modify-write epilogue wants the third block) and for grids of about one block per CU.  SEG_V2_BK=32|64 overrides.
static int fwd2_bk(const seg_igemm_fwd_args& a, long nblocks) {
  const char* env = std::getenv("SEG_V2_BK");     // read per call: tools/op_bench.py flips it between launches
  if (env != nullptr) return std::atoi(env) == 64 ? 64 : 32;
  // (the 64-deep step runs a two-stage ring; SEG_V2_TWO_STAGE=0 keeps every launch on the 32-deep three-stage form - the
  // bisecting switch of the LDS write-after-read hazard described at the barrier of igemm_fwd2_kernel)
  static const bool two_stage = std::getenv("SEG_V2_TWO_STAGE") == nullptr || std::atoi(std::getenv("SEG_V2_TWO_STAGE")) != 0;
  if (!two_stage) return 32;
  if (a.Kpad & 63) return 32;
  int cmin = 1 << 30;
  for (int s = 0; s < a.nseg; ++s) cmin = a.seg_c[s] < cmin ? a.seg_c[s] : cmin;
  if (a.tile_m == 128 && a.tile_n == 128) return cmin >= 128 ? 64 : 32;
  if (cmin < 64) return 32;
  return (nblocks >= 512 && (long)a.g.ntaps * a.Kpad >= 1024) ? 64 : 32;
}

static int launch_fwd2(const seg_igemm_fwd_args& a, int mblocks, int nblocks, hipStream_t st) {
  const int ksl = a.splitk > 1 ? a.splitk : 1;
  dim3 grid((unsigned)(mblocks * nblocks), (unsigned)a.g.ncls, (unsigned)ksl);
  auto finish = [&](const char* what) -> int {
    SEG_CHECK_LAUNCH(what);
    if (ksl > 1) {
      launch_fwd2_splitk_epilogue<false>(a, grid, st);
      SEG_CHECK_LAUNCH("seg_igemm_fwd (split reduction, epilogue)");
    }
    return SEG_OK;
  };
  // deep-and-small layers: with ~1 block per CU nothing else hides a block's barriers and load latency, so take both
  // the 64-deep K step (8 instead of 4 MFMAs per wave and barrier) and two steps of loads in flight (72 KB of LDS)
  constexpr long deep_max = 1024;          // measured: 384 -> 1024 blocks -0.14 ms per C3 step, 4096 the same
  if (std::getenv("SEG_V2_BK") == nullptr && a.tile_m == 128 && a.tile_n == 64 && (long)mblocks * nblocks * a.g.ncls <= deep_max &&
      (a.g.ntaps == 9 || a.g.ntaps == 16) && (a.Kpad & 63) == 0) {
    bool ok = true;
    for (int s = 0; s < a.nseg; ++s) ok = ok && a.seg_c[s] >= 64 && (a.seg_c[s] & 63) == 0;
    if (ok) {
      const size_t lds = (size_t)3 * (128 + 64) * 128;
      if (a.g.ntaps == 9) hipLaunchKernelGGL((igemm_fwd2_kernel<128, 64, 9, 64, 3>), grid, dim3(256), lds, st, a);
      else hipLaunchKernelGGL((igemm_fwd2_kernel<128, 64, 16, 64, 3>), grid, dim3(256), lds, st, a);
      return finish("seg_igemm_fwd (LDS-DMA, deep ring)");
    }
  }
  if (fwd2_bk(a, (long)mblocks * nblocks * a.g.ncls) == 64) {
    const size_t lds = (size_t)2 * (a.tile_m + a.tile_n) * 128;     // >= epilogue staging (20.5 KB)
    if (a.tile_m == 128 && a.tile_n == 128) launch_fwd2_nt<128, 128, 64, 2>(a, grid, lds, st);
    else if (a.tile_m == 128) launch_fwd2_nt<128, 64, 64, 2>(a, grid, lds, st);
    else launch_fwd2_nt<256, 64, 64, 2>(a, grid, lds, st);
  } else {
    const size_t lds = (size_t)3 * (a.tile_m + a.tile_n) * 64;
    if (a.tile_m == 128 && a.tile_n == 128) launch_fwd2_nt<128, 128, 32, 3>(a, grid, lds, st);
    else if (a.tile_m == 128) launch_fwd2_nt<128, 64, 32, 3>(a, grid, lds, st);
    else launch_fwd2_nt<256, 64, 32, 3>(a, grid, lds, st);
  }
  return finish("seg_igemm_fwd (LDS-DMA)");
}

// bytes of the `partial` workspace of a split reduction (seg_hip.h): one accumulator image per slice, class and output tile
extern "C" int64_t seg_igemm_fwd_workspace_bytes(const seg_igemm_fwd_args* a) {
  if (a == nullptr || a->tile_m <= 0 || a->tile_n <= 0) return -1;
  if (a->splitk <= 1) return 0;
  const int64_t M = (int64_t)a->g.N * a->g.Hl * a->g.Wl;
  const int64_t tiles = ((M + a->tile_m - 1) / a->tile_m) * ((a->n_rows + a->tile_n - 1) / a->tile_n) * a->g.ncls;
  return (int64_t)a->splitk * tiles * a->tile_m * a->tile_n * (int64_t)sizeof(float);
}

extern "C" int seg_igemm_fwd(const seg_igemm_fwd_args* a, seg_stream_t stream) {
  if (const int rc = check_fwd_args(a)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if ((a->variant & 255) == 3) return seg_launch_fwd3(*a, st);      // (bits 8.. : timing-experiment switches of the pipelined halo kernel, igemm_halo2.hip)
  const long M = (long)a->g.N * a->g.Hl * a->g.Wl;
  const int mblocks = (int)((M + a->tile_m - 1) / a->tile_m);
  const int nblocks = (a->n_rows + a->tile_n - 1) / a->tile_n;
  if (fwd2_eligible(*a, 2)) return launch_fwd2(*a, mblocks, nblocks, st);
  SEG_CHECK_ARG(a->splitk <= 1, "seg_igemm_fwd: splitk needs the LDS-DMA kernel (tiles 128x128 / 128x64 / 256x64, taps in {1,2,3,4,9,16}, operands < 1.875 GiB)");
  SEG_CHECK_ARG(a->tile_m <= 128, "seg_igemm_fwd: a 256-row tile needs the LDS-DMA kernel (taps in {1,2,3,4,9,16}, operands < 1.875 GiB)");
  dim3 grid((unsigned)(mblocks * nblocks), (unsigned)a->g.ncls, 1), block(256, 1, 1);
  const int lds_main = 2 * (a->tile_m + a->tile_n) * ROW_BYTES;
  const int lds_epi = STG_BYTES + 2 * a->tile_n * 2 * 4;
  const size_t lds = (size_t)(lds_main > lds_epi ? lds_main : lds_epi);
  if (a->tile_m == 128 && a->tile_n == 128) hipLaunchKernelGGL((igemm_fwd_kernel<false, 128, 128>), grid, block, lds, st, *a);
  else if (a->tile_m == 128 && a->tile_n == 64) hipLaunchKernelGGL((igemm_fwd_kernel<false, 128, 64>), grid, block, lds, st, *a);
  else if (a->tile_m == 64 && a->tile_n == 128) hipLaunchKernelGGL((igemm_fwd_kernel<false, 64, 128>), grid, block, lds, st, *a);
  else hipLaunchKernelGGL((igemm_fwd_kernel<false, 64, 64>), grid, block, lds, st, *a);
  SEG_CHECK_LAUNCH("seg_igemm_fwd");
  return SEG_OK;
}
