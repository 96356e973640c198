// Weight gradient of the 3x3 / stride-1 convolutions, tap-split generation of the halo-tile kernel (variant 2 of
// seg_igemm_wgrad; igemm_wgrad.hip keeps the round-1 kernel for the transposed convs, igemm_wgrad2.hip the wave-specialised
// form for small grids).  Same block tile (64 output channels x 64 input channels x 9 taps per 64-pixel patch), same LDS
// image, same LDS-DMA and transposing reads - a different split of the 36 accumulator tiles over the four waves.
//
// With the loads made asynchronous (igemm_common.h: lds_dma16) the round-1 kernel is bound by its LDS READS: a wave owns
// one 32x32 tile per tap, so every MFMA needs a fresh input-halo fragment - 80 ds_read_b64_tr_b16 (512 B each) per 36 MFMAs
// and wave; the four SIMDs of a CU share one 128 B/clk LDS, which those reads keep busy for 1280 clocks per 1152 clocks of
// MFMA.  Here a wave owns BOTH 32-row tiles of output channels against one 32-column slice of input channels, for half of
// the taps: tap group 0 = taps 0-3 plus the centre tap for row tile 0, tap group 1 = taps 5-8 plus the centre tap for row
// tile 1 (nine tiles each, as before).  An input fragment now feeds two MFMAs: 56 reads per 36 MFMAs (-30 %).
#include "igemm_common.h"

namespace {

template <int C>
__global__ __launch_bounds__(256, 2) void igemm_wgrad_halo_ts_kernel(const seg_igemm_wgrad_args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int R = 64 / C, HC = C + 2, HROWS = (R + 2) * HC;
  constexpr int NI_IN = (HROWS + 7) / 8;                  // LDS-DMA wave instructions per halo tile (8 rows of 128 B each)
  constexpr int NI_W = (NI_IN + 3) / 4;                   // ... per wave
  constexpr int Z_BYTES = 64 * 128, IN_BYTES = NI_IN * 1024, STAGE = Z_BYTES + IN_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
  typedef __attribute__((address_space(3))) s4v* lds_s4v_ptr;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tgrp = wave & 1, wn = wave >> 1;               // tap group, 32-column half of the input-channel plane
  const seg_geom& g = p.g;
  const int H = g.Hin, W = g.Win;
  // timing experiments (tools/op_bench.py --dbg on the wgrad phase; results are WRONG): 1 no LDS-DMA, 4 no barrier, 16 no epilogue
  const int dbg = __builtin_amdgcn_readfirstlane(p.variant >> 8);

  // ---- which (64-row tile, 64-channel plane, patch range) is this block: patch range (z) slowest, then (row tile, plane) -
  // blocks that read the same pixels sit on the same XCD and meet the data in its L2
  const int rtiles = (p.n_rows + 63) / 64;
  const int lin = xcd_remap(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z);
  const int bz = lin / gridDim.x, bx = lin - bz * gridDim.x;
  const int rt = bx % rtiles;
  int pl = bx / rtiles;
  int seg = 0;
  for (; seg < p.nseg; ++seg) {
    const int t = (p.seg_c[seg] + 63) / 64;
    if (pl < t) break;
    pl -= t;
  }
  if (seg >= p.nseg) return;
  const int r0 = rt * 64, c0 = pl * 64;
  const int sc = p.seg_c[seg], cs2 = p.seg_cs[seg] * 2, dzcs2 = p.dz_cs * 2;
  const int PW = W / C, PH = H / R;
  const int npatch = g.N * PH * PW;
  const int per = (npatch + p.splitk - 1) / p.splitk;
  const int q0 = bz * per, q1 = min(npatch, q0 + per);
  if (q0 >= q1) return;

  const i4v rsrc_in = make_rsrc(p.seg_ptr[seg], g.N * H * W * cs2);
  const i4v rsrc_dz = make_rsrc(p.dz, g.N * H * W * dzcs2);
  const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);

  // ---- LDS-DMA lane constants.  A wave instruction fills 8 LDS rows (lane >> 3) x 8 chunks (lane & 7); the chunk a lane
  // fetches from memory is its LDS chunk XOR the row's swizzle
  const int lrow = lane >> 3, lch = lane & 7;
  unsigned z_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 8 + lrow;
    const int gch = lch ^ (((row >> 1) & 1) << 2);
    const int yy = row / C, xx = row % C;
    z_voff[i] = (r0 + gch * 8 < p.n_rows) ? (unsigned)((yy * W + xx) * dzcs2 + (p.dz_coff + r0 + gch * 8) * 2) : OOB_OFF;
  }
  int in_dy[NI_W], in_dx[NI_W], in_cb[NI_W];      // halo pixel relative to the patch origin, channel byte offset (-1: no load)
#pragma unroll
  for (int k = 0; k < NI_W; ++k) {
    const int i = wave + 4 * k;
    const int row = i * 8 + lrow;
    const int gch = lch ^ (((row >> 1) & 1) << 2);
    const int yy = row / HC, xx = row - yy * HC;
    in_dy[k] = yy - 1;
    in_dx[k] = xx - 1;
    in_cb[k] = (i < NI_IN && row < HROWS && c0 + gch * 8 < sc) ? (c0 + gch * 8) * 2 : -1;
  }
  int pn_ = __builtin_amdgcn_readfirstlane(q0 / (PH * PW));
  int py_ = __builtin_amdgcn_readfirstlane((q0 - pn_ * (PH * PW)) / PW);
  int px_ = __builtin_amdgcn_readfirstlane(q0 - pn_ * (PH * PW) - py_ * PW);
  auto issue = [&](int stage) {
    if (dbg & 1) return;
    const int n = pn_, y0 = py_ * R, x0 = px_ * C;
    if (++px_ == PW) {
      px_ = 0;
      if (++py_ == PH) { py_ = 0; ++pn_; }
    }
    const unsigned sb = lds_base + (unsigned)(stage * STAGE);
    const int zs = __builtin_amdgcn_readfirstlane(((n * H + y0) * W + x0) * dzcs2);
#pragma unroll
    for (int i = 0; i < 2; ++i) lds_dma16(rsrc_dz, sb + (wave * 2 + i) * 1024, z_voff[i], zs);
#pragma unroll
    for (int k = 0; k < NI_W; ++k) {
      if (wave + 4 * k < NI_IN) {
        const int y = y0 + in_dy[k], x = x0 + in_dx[k];
        const bool ok = in_cb[k] >= 0 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const unsigned voff = ok ? (unsigned)(((n * H + y) * W + x) * cs2 + in_cb[k]) : OOB_OFF;
        lds_dma16(rsrc_in, sb + Z_BYTES + (wave + 4 * k) * 1024, voff, 0);
      }
    }
  };

  // ---- transposed-read lane offsets: pixel j = 8 (tg >> 1) + tq of a 16-pixel MFMA K block, 8 bytes at column
  // 16 (tg & 1) + 4 tp of a 32-channel slice.  Row = base + j: bit 1 of the row (the swizzle) depends on (base & 3) + j
  // only, so four lane offsets cover every base.
  const int tg = lane >> 4, tt = lane & 15, tq = tt >> 2, tp = tt & 3;
  const int j = 8 * (tg >> 1) + tq;
  int offA[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int cbA = (a * 32 + 16 * (tg & 1) + 4 * tp) * 2;
    offA[a] = j * 128 + (cbA ^ (((j >> 1) & 1) << 6));
  }
  const int cbB = (wn * 32 + 16 * (tg & 1) + 4 * tp) * 2;
  // 8-pixel-wide patches (the 8x8 level): a 16-pixel K block is two image rows, pixels 8..15 sit one halo row further
  const int jr = C >= 16 ? j : j + (j >> 3) * (HC - 8);
  int offB[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) offB[r] = Z_BYTES + jr * 128 + (cbB ^ ((((r + jr) >> 1) & 1) << 6));
  auto trfrag = [&](const unsigned char* ad) -> h8 {
    s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(ad));
    s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(ad + 4 * 128));
    h4 l4 = __builtin_bit_cast(h4, lo), h4v = __builtin_bit_cast(h4, hi);
    return h8{l4[0], l4[1], l4[2], l4[3], h4v[0], h4v[1], h4v[2], h4v[3]};
  };

  // acc[2 i + a]: tap TG * 5 + i (i < 4), row tile a;  acc[8]: the centre tap (4), row tile TG
  f16v acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

  auto compute = [&](auto tgc, int stage) {
    constexpr int TG = decltype(tgc)::value;
    const unsigned char* sb = smem + stage * STAGE;
    {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        constexpr int KPR = C >= 16 ? C / 16 : 1;                    // K blocks per patch row (C = 8: one K block = two rows)
        const int yy = C >= 16 ? kb / KPR : 2 * kb, xx0 = C >= 16 ? (kb % KPR) * 16 : 0;
        const h8 af0 = trfrag(sb + kb * 16 * 128 + offA[0]);
        const h8 af1 = trfrag(sb + kb * 16 * 128 + offA[1]);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const int t = i < 4 ? TG * 5 + i : 4;
          const int hb = (yy + t / 3) * HC + xx0 + t % 3;            // halo row of pixel 0 of this K block under tap t
          const h8 bf = trfrag(sb + hb * 128 + offB[hb & 3]);
          if (i < 4) {
            acc[2 * i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af0, bf, acc[2 * i], 0, 0, 0);
            acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af1, bf, acc[2 * i + 1], 0, 0, 0);
          } else {
            acc[8] = __builtin_amdgcn_mfma_f32_32x32x16_f16(TG == 0 ? af0 : af1, bf, acc[8], 0, 0, 0);
          }
        }
      }
    }
  };
  auto run = [&](auto tgc) {
    int stage = 0;
    issue(0);
    for (int q = q0; q < q1; ++q) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ring_release_barrier(!(dbg & 4));
      if (q + 1 < q1) issue(stage ^ 1);
      compute(tgc, stage);
      stage ^= 1;
    }
  };
  if (tgrp == 0) run(std::integral_constant<int, 0>{});         // wave-uniform: the tap offsets stay compile-time constants
  else run(std::integral_constant<int, 1>{});

  float sc_mul = p.scale;
  if (p.scale_ptr != nullptr) sc_mul *= *p.scale_ptr;
  float* dw = p.dw;
  float* part = (p.partial != nullptr && p.splitk > 1) ? p.partial + (size_t)bz * ((size_t)p.n_rows * g.T_total * p.Kpad) : nullptr;
  if (dbg & 16) {
    float keep = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) keep += acc[t][0] + acc[t][15];
    if (keep != 12345.678f) return;
  }
  const int stride = g.T_total * p.Kpad;
  float* const wroot = (part != nullptr ? part : dw) + p.seg_woff[seg];
  wgrad_mode_dispatch(wgrad_mode(p), [&](auto mc) {
    constexpr int MODE = decltype(mc)::value;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      const int t = s < 8 ? tgrp * 5 + (s >> 1) : 4;               // tap and row tile of accumulator s
      const int a = s < 8 ? (s & 1) : tgrp;
      const int row0 = r0 + a * 32;
      float* const tbase = wroot + (size_t)row0 * stride + g.tap_w[t] * p.Kpad;
      {
        const int kcol = c0 + wn * 32 + (lane & 31);
        if (kcol < sc) wgrad_store_tile<MODE>(tbase + (size_t)(4 * (lane >> 5)) * stride + kcol, acc[s], p.n_rows - row0, stride, sc_mul);
      }
    }
  });
#endif
}

template __global__ void igemm_wgrad_halo_ts_kernel<32>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_ts_kernel<16>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_ts_kernel<8>(const seg_igemm_wgrad_args);

}  // namespace

// called by seg_igemm_wgrad (igemm_wgrad.hip) for 3x3 convs after the argument checks; pc = patch width, grid as the round-1 kernel's
int seg_launch_wgrad_halo_ts(const seg_igemm_wgrad_args& a, int pc, dim3 grid, hipStream_t st) {
  const int ni = pc == 32 ? 17 : pc == 16 ? 14 : 13;
  const size_t lds = (size_t)2 * (8192 + ni * 1024);
  if (pc == 32) hipLaunchKernelGGL((igemm_wgrad_halo_ts_kernel<32>), grid, dim3(256), lds, st, a);
  else if (pc == 16) hipLaunchKernelGGL((igemm_wgrad_halo_ts_kernel<16>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((igemm_wgrad_halo_ts_kernel<8>), grid, dim3(256), lds, st, a);
  return 0;
}
