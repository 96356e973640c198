// Parity mode (suffix _f32) of seg_igemm_fwd: operands and outputs are split pairs (seg_common.h split_t, igemm_common.h).
// The loaders are the fp16 kernels' (a split tensor = an fp16 tensor with twice the channels); this unit holds the SPLIT
// instantiations of the variant-0 kernels and the entry point; the halo-tile kernels' are in igemm_halo.hip.
#include "igemm_common.h"
#include "igemm_fwd_kernels.h"

template <int BM, int BN>
static void launch_fwd2_split_nt(const seg_igemm_fwd_args& v, dim3 grid, size_t lds, hipStream_t st) {
  switch (v.g.ntaps) {
    case 1: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 1, 32, 3, true>), grid, dim3(256), lds, st, v); break;
    case 2: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 2, 32, 3, true>), grid, dim3(256), lds, st, v); break;
    case 3: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 3, 32, 3, true>), grid, dim3(256), lds, st, v); break;
    case 4: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 4, 32, 3, true>), grid, dim3(256), lds, st, v); break;
    case 9: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 9, 32, 3, true>), grid, dim3(256), lds, st, v); break;
    default: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 16, 32, 3, true>), grid, dim3(256), lds, st, v); break;
  }
}

// data gradients with the weight operand as its hi half (SEG_VARIANT_B_HI): the tap counts data gradients come with - 3x3 (9), the
// parity planes of a transposed conv (16), 1x1 (1)
template <int BM, int BN>
static bool launch_fwd2_split_bhi(const seg_igemm_fwd_args& v, dim3 grid, size_t lds, hipStream_t st) {
  switch (v.g.ntaps) {
    case 1: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 1, 32, 3, true, true>), grid, dim3(256), lds, st, v); return true;
    case 9: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 9, 32, 3, true, true>), grid, dim3(256), lds, st, v); return true;
    case 16: hipLaunchKernelGGL((igemm_fwd2_kernel<BM, BN, 16, 32, 3, true, true>), grid, dim3(256), lds, st, v); return true;
    default: return false;
  }
}

extern "C" int seg_igemm_fwd_f32(const seg_igemm_fwd_args* a, seg_stream_t stream) {
  if (const int rc = check_fwd_args(a)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const seg_igemm_fwd_args v = split_view(*a);        // what the loaders see: pseudo-channels
  if ((a->variant & 255) == 3) return seg_launch_fwd3_split(v, st);
  const long M = (long)a->g.N * a->g.Hl * a->g.Wl;
  const int mblocks = (int)((M + a->tile_m - 1) / a->tile_m);
  const int nblocks = (a->n_rows + a->tile_n - 1) / a->tile_n;
  dim3 grid((unsigned)(mblocks * nblocks), (unsigned)a->g.ncls, 1), block(256, 1, 1);
  if (fwd2_eligible(*a, 4)) {
    const size_t lds = (size_t)3 * (a->tile_m + a->tile_n) * 64;          // >= the epilogue's staging (20.5 KB)
    if (a->splitk > 1) grid.z = (unsigned)a->splitk;
    bool done = false;
    if (a->variant & SEG_VARIANT_B_HI) {         // (tap counts without a hi-only instance take the three-MFMA kernel: more exact, never wrong)
      if (a->tile_m == 128 && a->tile_n == 128) done = launch_fwd2_split_bhi<128, 128>(v, grid, lds, st);
      else if (a->tile_m == 128) done = launch_fwd2_split_bhi<128, 64>(v, grid, lds, st);
      else done = launch_fwd2_split_bhi<256, 64>(v, grid, lds, st);
    }
    if (done) { /* launched */ }
    else if (a->tile_m == 128 && a->tile_n == 128) launch_fwd2_split_nt<128, 128>(v, grid, lds, st);
    else if (a->tile_m == 128) launch_fwd2_split_nt<128, 64>(v, grid, lds, st);
    else launch_fwd2_split_nt<256, 64>(v, grid, lds, st);
    SEG_CHECK_LAUNCH("seg_igemm_fwd_f32 (LDS-DMA)");
    if (a->splitk > 1) {
      launch_fwd2_splitk_epilogue<true>(v, grid, st);
      SEG_CHECK_LAUNCH("seg_igemm_fwd_f32 (split reduction, epilogue)");
    }
    return SEG_OK;
  }
  SEG_CHECK_ARG(a->splitk <= 1, "seg_igemm_fwd_f32: splitk needs the LDS-DMA kernel (tiles 128x128 / 128x64 / 256x64, taps in {1,2,3,4,9,16}, operands < 1.875 GiB)");
  SEG_CHECK_ARG(a->tile_m <= 128, "seg_igemm_fwd_f32: a 256-row tile needs the LDS-DMA kernel (taps in {1,2,3,4,9,16}, operands < 1.875 GiB)");
  const int lds_main = 2 * (a->tile_m + a->tile_n) * ROW_BYTES * 2;      // two sub-tiles per K step
  const int lds_epi = STG_BYTES + 2 * a->tile_n * 2 * 4;
  const size_t lds = (size_t)(lds_main > lds_epi ? lds_main : lds_epi);
  if (a->tile_m == 128 && a->tile_n == 128) hipLaunchKernelGGL((igemm_fwd_kernel<true, 128, 128>), grid, block, lds, st, v);
  else if (a->tile_m == 128 && a->tile_n == 64) hipLaunchKernelGGL((igemm_fwd_kernel<true, 128, 64>), grid, block, lds, st, v);
  else if (a->tile_m == 64 && a->tile_n == 128) hipLaunchKernelGGL((igemm_fwd_kernel<true, 64, 128>), grid, block, lds, st, v);
  else hipLaunchKernelGGL((igemm_fwd_kernel<true, 64, 64>), grid, block, lds, st, v);
  SEG_CHECK_LAUNCH("seg_igemm_fwd_f32");
  return SEG_OK;
}
