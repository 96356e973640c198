// Parity mode (suffix _f32) of seg_igemm_wgrad for gfx950:  dW[n][tap][k] = sum_pix dz[pix][n] * in[pix + tap][k]  with both
// operands stored as split pairs (seg_common.h split_t: 8 channels = [8 hi | 8 lo] fp16 halves, x = hi + lo / 2048).  Each
// product is three v_mfma_f32_32x32x16_f16 (hi hi, lo hi, hi lo) into two fp32 accumulator sets; the operands are fetched in
// their stored form (LDS-DMA for the halo kernel) and read with the transposing ds_read_b64_tr_b16 - a 16-byte chunk of a
// pixel row is 8 channels of ONE of the two planes, so the hi and the lo fragment are two reads at different chunk addresses.
// dW is plain fp32 (the flat gradient buffer of the optimiser).
#include "igemm_common.h"

namespace {

typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s4v* lds_s4v_ptr;
// 16 pixels (two groups of four rows, `rstride` bytes apart) x this lane's channel of one plane -> an 8-deep MFMA fragment
__device__ __forceinline__ h8 trfrag(const unsigned char* ad, int rstride4) {
  s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(ad));
  s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(ad + rstride4));
  h4 l4 = __builtin_bit_cast(h4, lo), h4v = __builtin_bit_cast(h4, hi);
  return h8{l4[0], l4[1], l4[2], l4[3], h4v[0], h4v[1], h4v[2], h4v[3]};
}
#define SEG_MMA_SPLIT(ACC, ACCL, AH, AL, BH, BL)                                  \
  do {                                                                            \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, ACC, 0, 0, 0);           \
    ACCL = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH, ACCL, 0, 0, 0);         \
    ACCL = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BL, ACCL, 0, 0, 0);         \
  } while (0)

// SEG_VARIANT_IN_HI (SegModel.build(backward="...w2"): the default's weight gradients): the INPUT operand read as its hi half - a b ~= ah bh + al bh / 2048
// with a = dz whole, b = the activations' hi half.  The lo fragment of every tap's halo read goes (5 of the 12 fragment reads per K
// block of the 3x3 kernel) and a third of the MFMAs; the activations lose 2^-11, elementwise (measured 3.8e-4 on the worst parameter tensor of C2 / C5, 2.8e-4 on C4).
#define SEG_MMA_SPLIT_BHI(ACC, ACCL, AH, AL, BH)                                  \
  do {                                                                            \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, ACC, 0, 0, 0);           \
    ACCL = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH, ACCL, 0, 0, 0);         \
  } while (0)

// SEG_VARIANT_IN_HI | SEG_VARIANT_DZ_HI: BOTH operands as their hi halves - one MFMA per product, one accumulator set, two of the four
// fragment reads (a b ~= ah bh).  The weight gradient is the END of a chain: its rounding (2^-12 rms on each operand, elementwise,
// unbiased) lands once in dW and is not carried into any other gradient - unlike a data gradient's, which every layer below inherits.
#define SEG_MMA_HI(ACC, AH, BH)                                                   \
  do {                                                                            \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, ACC, 0, 0, 0);           \
  } while (0)

// ---------------------------------------------------------------------------------------------------------
// per-tap kernel (1x1, strided, 1-D convolutions, images the halo patches do not tile): register-staged double buffer,
// 32 pixels per step.  LDS per operand: a hi tile and a lo tile [32 pixels][BM (BN) halves], rows padded by 64 B.
// A thread's 16-byte global chunk q of a pixel row is group q >> 1 of plane q & 1 and goes to that plane's tile unchanged.
// ---------------------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void igemm_wgrad_split_kernel(const seg_igemm_wgrad_args p) {
  constexpr int TM = BM / 64, TN = BN / 64, PK = 32;
  constexpr int ZLD = BM * 2 + 64, ALD = BN * 2 + 64;
  constexpr int Z_BYTES = 2 * PK * ZLD, A_BYTES = 2 * PK * ALD;
  constexpr int ZCH = BM / 4, ACH = BN / 4;                        // 16-B chunks per pixel row (hi and lo)
  constexpr int ZPT = (PK * ZCH) / 256, APT = (PK * ACH) / 256;   // chunks per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const seg_geom& g = p.g;
  const int HW = g.Hl * g.Wl;
  const int M = g.N * HW;

  const int rtiles = (p.n_rows + BM - 1) / BM;
  const int lin = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
  const int bz = lin / (gridDim.x * gridDim.y), by = (lin / gridDim.x) % gridDim.y, bx = lin % gridDim.x;
  const int rt = bx % rtiles;
  int kt = bx / rtiles;
  int seg = 0;
  for (; seg < p.nseg; ++seg) {
    int t = (p.seg_c[seg] + BN - 1) / BN;
    if (kt < t) break;
    kt -= t;
  }
  if (seg >= p.nseg) return;
  const int e = by;                                // cls * ntaps + tap
  const int cls = e / g.ntaps;
  const int dy = g.tap_dy[e], dx = g.tap_dx[e], widx = g.tap_w[e];
  const int oy = g.oy[cls], ox = g.ox[cls];
  const int r0 = rt * BM, k0 = kt * BN;
  const int sc = p.seg_c[seg], cs = p.seg_cs[seg];
  const split_t* sp = (const split_t*)p.seg_ptr[seg];
  const split_t* dz = (const split_t*)p.dz;

  int per = (M + p.splitk - 1) / p.splitk;
  per = (per + PK - 1) / PK * PK;
  const int pbeg = bz * per;
  const int pend = min(M, pbeg + per);
  if (pbeg >= pend) return;

  int z_p[ZPT], z_c[ZPT], zn[ZPT], zi[ZPT], zj[ZPT];
  int a_p[APT], a_c[APT], an[APT], ai[APT], aj[APT];
#pragma unroll
  for (int r = 0; r < ZPT; ++r) {
    int q = tid + 256 * r;
    z_p[r] = q / ZCH; z_c[r] = q % ZCH;
    int m = pbeg + z_p[r];
    zn[r] = m / HW; int rem = m - zn[r] * HW; zi[r] = rem / g.Wl; zj[r] = rem - zi[r] * g.Wl;
  }
#pragma unroll
  for (int r = 0; r < APT; ++r) {
    int q = tid + 256 * r;
    a_p[r] = q / ACH; a_c[r] = q % ACH;
    int m = pbeg + a_p[r];
    an[r] = m / HW; int rem = m - an[r] * HW; ai[r] = rem / g.Wl; aj[r] = rem - ai[r] * g.Wl;
  }
  auto advance = [&](int& n, int& i, int& j) {
    j += PK;
    while (j >= g.Wl) { j -= g.Wl; ++i; if (i >= g.Hl) { i = 0; ++n; } }
  };

  f16v acc[TM][TN], accl[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) { acc[a][b][q] = 0.f; accl[a][b][q] = 0.f; }

  u4 rz[ZPT], ra[APT];
  auto gload = [&](int pb) {
#pragma unroll
    for (int r = 0; r < ZPT; ++r) {
      u4 v = {0u, 0u, 0u, 0u};
      int m = pb + z_p[r];
      int col = r0 + (z_c[r] >> 1) * 8;
      if (m < pend && col < p.n_rows) {
        size_t off = (((size_t)zn[r] * g.Hout + (zi[r] * g.osy + oy)) * g.Wout + (zj[r] * g.osx + ox)) * p.dz_cs + p.dz_coff + col + (z_c[r] & 1) * 4;
        v = ld16(dz + off);
      }
      rz[r] = v;
      advance(zn[r], zi[r], zj[r]);
    }
#pragma unroll
    for (int r = 0; r < APT; ++r) {
      u4 v = {0u, 0u, 0u, 0u};
      int m = pb + a_p[r];
      int col = k0 + (a_c[r] >> 1) * 8;
      int y = ai[r] * g.isy + dy, x = aj[r] * g.isx + dx;
      if (m < pend && col < sc && (unsigned)y < (unsigned)g.Hin && (unsigned)x < (unsigned)g.Win) {
        size_t off = (((size_t)an[r] * g.Hin + y) * g.Win + x) * cs + col + (a_c[r] & 1) * 4;
        v = ld16(sp + off);
      }
      ra[r] = v;
      advance(an[r], ai[r], aj[r]);
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* Z = smem + buf * (Z_BYTES + A_BYTES);
    unsigned char* A = Z + Z_BYTES;
#pragma unroll
    for (int r = 0; r < ZPT; ++r) st16(Z + (z_c[r] & 1) * (PK * ZLD) + z_p[r] * ZLD + (z_c[r] >> 1) * 16, rz[r]);
#pragma unroll
    for (int r = 0; r < APT; ++r) st16(A + (a_c[r] & 1) * (PK * ALD) + a_p[r] * ALD + (a_c[r] >> 1) * 16, ra[r]);
  };
  const int fr = lane & 31, fh = lane >> 5;
  const int tg = lane >> 4, tt = lane & 15, tq = tt >> 2, tp = tt & 3;
  auto compute = [&](int buf) {
    const unsigned char* Z = smem + buf * (Z_BYTES + A_BYTES);
    const unsigned char* A = Z + Z_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h8 af[TM], afl[TM], bf[TN], bfl[TN];
      const int krow = ks * 16 + 8 * (tg >> 1) + tq;
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const unsigned char* ad = Z + krow * ZLD + (wm * (BM / 2) + a * 32 + 16 * (tg & 1) + 4 * tp) * 2;
        af[a] = trfrag(ad, 4 * ZLD);
        afl[a] = trfrag(ad + PK * ZLD, 4 * ZLD);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const unsigned char* ad = A + krow * ALD + (wn * (BN / 2) + b * 32 + 16 * (tg & 1) + 4 * tp) * 2;
        bf[b] = trfrag(ad, 4 * ALD);
        bfl[b] = trfrag(ad + PK * ALD, 4 * ALD);
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) SEG_MMA_SPLIT(acc[a][b], accl[a][b], af[a], afl[a], bf[b], bfl[b]);
    }
  };

  const int niter = (pend - pbeg + PK - 1) / PK;
  gload(pbeg);
  lstore(0);
  __syncthreads();
  for (int it = 0; it < niter; ++it) {
    const bool has_next = it + 1 < niter;
    if (has_next) gload(pbeg + (it + 1) * PK);
    compute(it & 1);
    if (has_next) lstore((it + 1) & 1);
    __syncthreads();
  }
  split_combine(acc, accl);

  float sc_mul = p.scale;
  if (p.scale_ptr != nullptr) sc_mul *= *p.scale_ptr;
  float* dw = p.dw;
  float* part = (p.partial != nullptr && p.splitk > 1) ? p.partial + (size_t)bz * ((size_t)p.n_rows * g.T_total * p.Kpad) : nullptr;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int kcol = k0 + wn * (BN / 2) + b * 32 + fr;
      if (kcol < sc) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = r0 + wm * (BM / 2) + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh;
          if (row < p.n_rows) {
            size_t off = ((size_t)row * g.T_total + widx) * p.Kpad + p.seg_woff[seg] + kcol;
            if (part != nullptr) part[off] = acc[a][b][q];          // deterministic split: reduced in order by the reduce launch
            else atomicAdd(dw + off, acc[a][b][q] * sc_mul);
          }
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------------
// halo-tile kernel (variant 2): 3x3 / stride-1 / 'same' convolutions and the sub-pixel classes of the 4x4 / stride-2
// transposed convolution.  A block owns 64 output channels x one 32-channel plane of one concat segment x all taps and walks a
// range of 64-pixel patches (R rows x C columns).  Per patch the dz tile [64 px][256 B = 64 channels, hi / lo chunks] and ONE
// input halo tile [(R+2)(C+2) px][128 B = 32 channels] come in by LDS-DMA (2-stage ring, one barrier per patch); a tap is a row
// offset into the halo tile.  Waves = (row tile wm) x (tap half th): a 3x3 conv gives th = 0 the taps 0-3 and the first two
// 16-pixel K blocks of tap 4, th = 1 the taps 5-8 and the last two K blocks of tap 4 (the two partial sums of tap 4 meet in LDS
// at the end) - 18 MFMA triples per wave and patch for everybody, 5 x 2 accumulator tiles.  A transposed-conv class has four taps:
// two per wave.
// Bank conflicts of the transposing reads (a 32-lane half reads 4 rows x 4 chunks of ONE plane, i.e. every second 16-B slot):
//   dz rows are one 256-B bank row each: slot s of row r holds global chunk s ^ ((r & 1) | ((r >> 1) & 1) << 3) - rows r .. r+3 then
//     put their four hi (or lo) chunks on 16 distinct slots;
//   halo rows are half a bank row: the row's parity picks the half, ((r >> 1) & 1) swaps hi and lo chunks (slot s holds chunk
//     s ^ ((r >> 1) & 1)) - four consecutive rows at any tap shift hit 16 distinct slots.
// With LDS-DMA the LDS image is lane-linear, so the permutation is applied to the chunk a lane FETCHES and undone on the read.
// NARROW (layers of at most 32 output channels - MultiResUNet's full-resolution convs): the upper row tile would contract zeros, so
// the four waves split the TAPS instead: wave w owns taps 2 w and 2 w + 1 and K block w of tap 8 (the four partial sums of tap 8
// meet in LDS at the end, added in wave order) - 9 MFMA triples per wave and patch instead of 18; a transposed-conv class gives
// each wave one of its four taps.
// ---------------------------------------------------------------------------------------------------------
template <int C, bool TC, bool NARROW = false, int BHI = 0>      // BHI: 0 three MFMAs per product, 1 the input operand's hi half (two), 2 both hi halves (one)
__global__ __launch_bounds__(256, 2) void igemm_wgrad_halo_split_kernel(const seg_igemm_wgrad_args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int R = 64 / C, HC = C + 2, HROWS = (R + 2) * HC;
  constexpr int NI_IN = (HROWS + 7) / 8;                  // LDS-DMA wave instructions per halo tile (8 rows of 128 B each)
  constexpr int NI_W = (NI_IN + 3) / 4;                   // ... per wave
  constexpr int Z_BYTES = 64 * 256, IN_BYTES = NI_IN * 1024, STAGE = Z_BYTES + IN_BYTES;
  constexpr int NACC = NARROW ? (TC ? 1 : 3) : (TC ? 2 : 5);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = NARROW ? 0 : wave >> 1, th = NARROW ? 0 : wave & 1;
  const seg_geom& g = p.g;
  const int H = g.Hin, W = g.Win;

  // logical order: patch range (z) slowest, then sub-pixel class (y), then (row tile, plane) - blocks that read the same pixels
  // are handed to the same XCD
  const int rtiles = (p.n_rows + 63) / 64;
  const int lin = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
  const int bz = lin / (gridDim.x * gridDim.y);
  const int lrem = lin - bz * (gridDim.x * gridDim.y);
  const int bcls = lrem / gridDim.x, bx = lrem - bcls * gridDim.x;
  const int rt = bx % rtiles;
  int pl = bx / rtiles;
  int seg = 0;
  for (; seg < p.nseg; ++seg) {
    const int t = (p.seg_c[seg] + 31) / 32;
    if (pl < t) break;
    pl -= t;
  }
  if (seg >= p.nseg) return;
  const int r0 = rt * 64, c0 = pl * 32;
  const int sc = p.seg_c[seg], cs4 = p.seg_cs[seg] * 4, dzcs4 = p.dz_cs * 4;
  const int PW = W / C, PH = H / R;
  const int npatch = g.N * PH * PW;
  const int per = (npatch + p.splitk - 1) / p.splitk;
  const int q0 = bz * per, q1 = min(npatch, q0 + per);
  const int cls = TC ? bcls : 0;
  // transposed conv: the bias gradient (column sums of dz) rides on this launch - the blocks of the first input plane contract their
  // dz fragments with a ones operand (two more MFMAs per K block in the th = 0 waves) and write one partial row per (class, patch
  // range), the layout seg_bn_bwd_finalize reduces - instead of a separate pass over the gradient tensor
  const bool bias_blk = TC && p.bias_part != nullptr && seg == 0 && c0 == 0;
  if (q0 >= q1) {
    if (bias_blk && tid < 64 && r0 + tid < p.n_rows) {             // an empty patch range still owns its partial row
      float* rowp = p.bias_part + ((size_t)(cls * p.splitk + bz) * 2) * p.n_rows;
      rowp[r0 + tid] = 0.f;
      rowp[p.n_rows + r0 + tid] = 0.f;
    }
    return;
  }
  // transposed conv: this wave's two taps (halo shift index 3 (dy + 1) + dx + 1 and weight tap) - wave-uniform run-time values
  // (NARROW conv: the same table holds this wave's taps 2 w, 2 w + 1 and the shared tap 8)
  int tc_t[3] = {0, 0, 0}, tc_w[3] = {0, 0, 0};
  if constexpr (TC) {
#pragma unroll
    for (int i = 0; i < (NARROW ? 1 : 2); ++i) {
      const int e = cls * 4 + (NARROW ? wave : 2 * th + i);
      tc_t[i] = __builtin_amdgcn_readfirstlane((g.tap_dy[e] + 1) * 3 + g.tap_dx[e] + 1);
      tc_w[i] = __builtin_amdgcn_readfirstlane(g.tap_w[e]);
    }
  } else if constexpr (NARROW) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = i < 2 ? 2 * wave + i : 8;                        // (tap e of a 3x3 conv sits at halo shift e: checked by the launcher)
      tc_t[i] = e;
      tc_w[i] = __builtin_amdgcn_readfirstlane(g.tap_w[e]);
    }
  }
  const int Ho = g.Hout, Wo = g.Wout, osy = g.osy, osx = g.osx;

  const i4v rsrc_in = make_rsrc(p.seg_ptr[seg], g.N * H * W * cs4);
  const i4v rsrc_dz = make_rsrc(p.dz, g.N * Ho * Wo * dzcs4);
  const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);

  // ---- LDS-DMA lane constants.  dz: an instruction fills 4 rows (lane >> 4) x 16 chunks (lane & 15), 16 instructions per tile
  unsigned z_voff[4];
  {
    const int lrow = lane >> 4, lch = lane & 15;
    const int gch = lch ^ ((lrow & 1) | (((lrow >> 1) & 1) << 3));      // (rows of an instruction start at a multiple of 4)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave * 4 + i) * 4 + lrow;
      const int yy = row / C, xx = row % C;
      z_voff[i] = (r0 + (gch >> 1) * 8 < p.n_rows) ? (unsigned)((yy * osy * Wo + xx * osx) * dzcs4 + (p.dz_coff + r0) * 4 + gch * 16) : OOB_OFF;
    }
  }
  // halo: an instruction fills 8 rows (lane >> 3) x 8 chunks (lane & 7)
  int in_dy[NI_W], in_dx[NI_W], in_cb[NI_W];      // halo pixel relative to the patch origin, channel byte offset (-1: no load)
  {
    const int lrow = lane >> 3, lch = lane & 7;
#pragma unroll
    for (int k = 0; k < NI_W; ++k) {
      const int i = wave + 4 * k;
      const int row = i * 8 + lrow;
      const int gch = lch ^ ((row >> 1) & 1);
      const int yy = row / HC, xx = row - yy * HC;
      in_dy[k] = yy - 1;
      in_dx[k] = xx - 1;
      // (BHI: the lo chunks of the input rows - odd chunk indices - are never read: no fetch)
      in_cb[k] = (i < NI_IN && row < HROWS && c0 + (gch >> 1) * 8 < sc && !(BHI != 0 && (gch & 1))) ? c0 * 4 + gch * 16 : -1;
    }
  }
  int pn_ = __builtin_amdgcn_readfirstlane(q0 / (PH * PW));
  int py_ = __builtin_amdgcn_readfirstlane((q0 - pn_ * (PH * PW)) / PW);
  int px_ = __builtin_amdgcn_readfirstlane(q0 - pn_ * (PH * PW) - py_ * PW);
  auto issue = [&](int stage) {
    const int n = pn_, y0 = py_ * R, x0 = px_ * C;
    if (++px_ == PW) {
      px_ = 0;
      if (++py_ == PH) { py_ = 0; ++pn_; }
    }
    const unsigned sb = lds_base + (unsigned)(stage * STAGE);
    const int zs = __builtin_amdgcn_readfirstlane(((n * Ho + y0 * osy + g.oy[cls]) * Wo + x0 * osx + g.ox[cls]) * dzcs4);
#pragma unroll
    for (int i = 0; i < 4; ++i) lds_dma16(rsrc_dz, sb + (wave * 4 + i) * 1024, z_voff[i], zs);
#pragma unroll
    for (int k = 0; k < NI_W; ++k) {
      if (wave + 4 * k < NI_IN) {
        const int y = y0 + in_dy[k], x = x0 + in_dx[k];
        const bool ok = in_cb[k] >= 0 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const unsigned voff = ok ? (unsigned)(((n * H + y) * W + x) * cs4 + in_cb[k]) : OOB_OFF;
        lds_dma16(rsrc_in, sb + Z_BYTES + (wave + 4 * k) * 1024, voff, 0);
      }
    }
  };

  // ---- transposed-read lane offsets: pixel j = 8 (tg >> 1) + tq of a 16-pixel K block, 8 bytes = 4 channels at channel
  // 16 (tg & 1) + 4 tp of the wave's 32-channel slice, i.e. group 2 (tg & 1) + (tp >> 1), halves 4 (tp & 1) ..
  const int tg = lane >> 4, tt = lane & 15, tq = tt >> 2, tp = tt & 3;
  const int j = 8 * (tg >> 1) + tq;
  const int grp = 2 * (tg & 1) + (tp >> 1), sub = (tp & 1) * 8;
  const int zsw = (tq & 1) | (((tq >> 1) & 1) << 3);                    // dz rows kb * 16 + j (+ 4): row & 3 == tq
  const int offAh = j * 256 + (((2 * (wm * 4 + grp)) ^ zsw) << 4) + sub;
  const int offAl = j * 256 + (((2 * (wm * 4 + grp) + 1) ^ zsw) << 4) + sub;
  // 8-pixel-wide patches (the 8x8 level): a 16-pixel K block is two image rows, pixels 8..15 sit one halo row further
  const int jr = C >= 16 ? j : j + (j >> 3) * (HC - 8);
  int offB[4];        // hi fragment for halo rows hb + jr with hb & 3 == r; the lo fragment is the neighbouring chunk (^ 16)
#pragma unroll
  for (int r = 0; r < 4; ++r) offB[r] = jr * 128 + (((2 * grp) ^ (((r + jr) >> 1) & 1)) << 4) + sub;

  f16v acc[NACC], accl[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc[t][q] = 0.f; accl[t][q] = 0.f; }

  const bool do_bias = bias_blk && (NARROW ? wave == 0 : th == 0);
  f16v accb, accbl;
#pragma unroll
  for (int q = 0; q < 16; ++q) { accb[q] = 0.f; accbl[q] = 0.f; }
  const h8 ones = h8{(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};

  // K block kb of a patch: image row / first column inside the patch
  auto kb_yy = [](int kb) { constexpr int KPR = C >= 16 ? C / 16 : 1; return C >= 16 ? kb / KPR : 2 * kb; };
  auto kb_xx = [](int kb) { constexpr int KPR = C >= 16 ? C / 16 : 1; return C >= 16 ? (kb % KPR) * 16 : 0; };
  auto compute_conv = [&](auto thc, int stage) {
    constexpr int TH = decltype(thc)::value;
    const unsigned char* Zs = smem + stage * STAGE;
    const unsigned char* Is = Zs + Z_BYTES;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int yy = kb_yy(kb), xx0 = kb_xx(kb);
      const h8 ah = trfrag(Zs + kb * 16 * 256 + offAh, 4 * 256);
      h8 al;
      if constexpr (BHI != 2) al = trfrag(Zs + kb * 16 * 256 + offAl, 4 * 256);
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int t = s < 4 ? TH * 5 + s : 4;
        if (s == 4 && (kb >> 1) != TH) continue;
        const int hb = (yy + t / 3) * HC + xx0 + t % 3;            // halo row of pixel 0 of this K block under tap t
        const int off = hb * 128 + offB[hb & 3];
        const h8 bh = trfrag(Is + off, 4 * 128);
        if constexpr (BHI == 2) {
          SEG_MMA_HI(acc[s], ah, bh);
        } else if constexpr (BHI == 1) {
          SEG_MMA_SPLIT_BHI(acc[s], accl[s], ah, al, bh);
        } else {
          const h8 bl = trfrag(Is + (off ^ 16), 4 * 128);
          SEG_MMA_SPLIT(acc[s], accl[s], ah, al, bh, bl);
        }
      }
    }
  };
  auto compute_tc = [&](int stage) {                                 // taps from the run-time table: transposed conv, NARROW conv
    const unsigned char* Zs = smem + stage * STAGE;
    const unsigned char* Is = Zs + Z_BYTES;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int yy = kb_yy(kb), xx0 = kb_xx(kb);
      const h8 ah = trfrag(Zs + kb * 16 * 256 + offAh, 4 * 256);
      h8 al;
      if constexpr (BHI != 2) al = trfrag(Zs + kb * 16 * 256 + offAl, 4 * 256);
      if (do_bias) {                                                 // wave-uniform (the bias gradient keeps both halves of dz in every mode)
        if constexpr (BHI == 2) al = trfrag(Zs + kb * 16 * 256 + offAl, 4 * 256);
        accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ones, accb, 0, 0, 0);
        accbl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ones, accbl, 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < NACC; ++s) {
        if (!TC && NARROW && s == 2 && kb != wave) continue;          // tap 8: K block `wave` only (wave-uniform)
        const int t = tc_t[s];
        const int hb = (yy + t / 3) * HC + xx0 + t % 3;            // wave-uniform, run time
        const int swz = (((hb & 3) + jr) >> 1) & 1;
        const int off = (hb + jr) * 128 + (((2 * grp) ^ swz) << 4) + sub;
        const h8 bh = trfrag(Is + off, 4 * 128);
        if constexpr (BHI == 2) {
          SEG_MMA_HI(acc[s], ah, bh);
        } else if constexpr (BHI == 1) {
          SEG_MMA_SPLIT_BHI(acc[s], accl[s], ah, al, bh);
        } else {
          const h8 bl = trfrag(Is + (off ^ 16), 4 * 128);
          SEG_MMA_SPLIT(acc[s], accl[s], ah, al, bh, bl);
        }
      }
    }
  };
  auto run = [&](auto thc) {
    int stage = 0;
    issue(0);
    for (int q = q0; q < q1; ++q) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the loads of this patch have landed
      ring_release_barrier();
      if (q + 1 < q1) issue(stage ^ 1);
      if constexpr (TC || NARROW) compute_tc(stage);
      else compute_conv(thc, stage);
      stage ^= 1;
    }
  };
  if (TC || NARROW || th == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});

  if constexpr (BHI != 2) {
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][q] += accl[t][q] * SPLIT_INV;
  }
  if constexpr (!TC && NARROW) {      // the four K blocks of tap 8: waves 1-3 hand their partial tiles to wave 0 through LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    float* xch = reinterpret_cast<float*>(smem);
    if (wave != 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) xch[((wave - 1) * 16 + q) * 64 + lane] = acc[2][q];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[2][q] += xch[(w * 16 + q) * 64 + lane];
    }
  } else if constexpr (!TC) {      // the two halves of tap 4: th = 1 hands its partial tile to th = 0 through LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    float* xch = reinterpret_cast<float*>(smem) + wm * (16 * 64);
    if (th == 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) xch[q * 64 + lane] = acc[4][q];
    }
    __syncthreads();
    if (th == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[4][q] += xch[q * 64 + lane];
    }
  }

  float sc_mul = p.scale;
  if (p.scale_ptr != nullptr) sc_mul *= *p.scale_ptr;
  float* dw = p.dw;
  float* part = (p.partial != nullptr && p.splitk > 1) ? p.partial + (size_t)bz * ((size_t)p.n_rows * g.T_total * p.Kpad) : nullptr;
  const int fr = lane & 31, fh = lane >> 5;
  if (do_bias && fr == 0) {          // every column of accb holds the same sums: the lanes of column 0 write the partial row
    float* rowp = p.bias_part + ((size_t)(cls * p.splitk + bz) * 2) * p.n_rows;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = r0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh;
      if (row < p.n_rows) {
        rowp[row] = accb[q] + accbl[q] * SPLIT_INV;
        rowp[p.n_rows + row] = 0.f;
      }
    }
  }
  const int kcol = c0 + fr;
  if (kcol < sc) {
    const int stride = g.T_total * p.Kpad, row0 = r0 + wm * 32;
    float* const wbase = (part != nullptr ? part : dw) + (size_t)(row0 + 4 * fh) * stride + p.seg_woff[seg] + kcol;
    wgrad_mode_dispatch(wgrad_mode(p), [&](auto mc) {
      if constexpr (TC || NARROW) {
#pragma unroll
        for (int s = 0; s < NACC; ++s) {
          if (!TC && s == 2 && wave != 0) continue;
          wgrad_store_tile<decltype(mc)::value>(wbase + tc_w[s] * p.Kpad, acc[s], p.n_rows - row0, stride, sc_mul);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          if (s == 4 && th != 0) continue;
          const int t = s < 4 ? th * 5 + s : 4;
          wgrad_store_tile<decltype(mc)::value>(wbase + g.tap_w[t] * p.Kpad, acc[s], p.n_rows - row0, stride, sc_mul);
        }
      }
    });
  }
#endif
}
template __global__ void igemm_wgrad_halo_split_kernel<32, false>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<16, false>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<8, false>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<32, true>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<16, true>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<8, true>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<32, false, true>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<16, false, true>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<8, false, true>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<32, true, true>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<16, true, true>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<8, true, true>(const seg_igemm_wgrad_args);
// input operand as its hi half: the 3x3 layers and the per-class transposed-conv form, 64-channel tiles
template __global__ void igemm_wgrad_halo_split_kernel<32, false, false, 1>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<16, false, false, 1>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<8, false, false, 1>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<32, true, false, 1>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<16, true, false, 1>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<8, true, false, 1>(const seg_igemm_wgrad_args);
// both operands as their hi halves (one MFMA per product)
template __global__ void igemm_wgrad_halo_split_kernel<32, false, false, 2>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<16, false, false, 2>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<8, false, false, 2>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<32, true, false, 2>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<16, true, false, 2>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_kernel<8, true, false, 2>(const seg_igemm_wgrad_args);

// ---------------------------------------------------------------------------------------------------------
// Conv2DTranspose(4x4, stride 2), parity mode, two sub-pixel classes per block (the class row py: classes (py, 0) and (py, 1);
// tap table checked by seg_wgrad_halo_class_groups() == 2).  The per-class form above is bound by its LDS reads (6 fragment reads
// for 6 MFMAs per K block and wave = 128 B per clock and CU at full MFMA rate).  Here a block owns 64 output channels x one
// 32-channel input plane x BOTH classes of a class row and walks 32-pixel patches (R = 32 / C rows x C columns): per patch two dz
// tiles [32 px][256 B] and ONE halo tile [(R+1)(C+2) px][128 B] - a class row needs only the input rows dy in {0,-1} (py = 0) or
// {1,0} (py = 1) - in a 2-stage ring of 50 KB (a 3-stage ring with counted waits was slower alone and in the step: 282 against
// 260 us on X05, +0.1 ms per C3 step).  Waves = (row tile wm) x (tap row iy): a wave reads both classes' dz
// fragments and the three column shifts of its input row (10 fragment reads) for 4 MFMA triples per K block: dx = 0 feeds
// both classes.  Per MFMA: 0.83 fragment reads instead of 1.0, 25 KB of LDS-DMA per 32 px and class pair instead of 2 x 16.5 KB.
// ---------------------------------------------------------------------------------------------------------
template <int C, int BHI = 0>
__global__ __launch_bounds__(256, 2) void igemm_wgrad_halo_split_tc2_kernel(const seg_igemm_wgrad_args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int PX = 32, R = PX / C, HC = C + 2, HROWS = (R + 1) * HC;
  constexpr int NI_IN = (HROWS + 7) / 8;                  // LDS-DMA wave instructions per halo tile (8 rows of 128 B each)
  constexpr int NI_W = (NI_IN + 3) / 4;                   // ... per wave, at most
  constexpr int Z_BYTES = PX * 256, IN_BYTES = NI_IN * 1024, STAGE = 2 * Z_BYTES + IN_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, iy = wave & 1;
  const seg_geom& g = p.g;
  const int H = g.Hin, W = g.Win;

  const int rtiles = (p.n_rows + 63) / 64;
  const int lin = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
  const int bz = lin / (gridDim.x * gridDim.y);
  const int lrem = lin - bz * (gridDim.x * gridDim.y);
  const int py = lrem / gridDim.x, bx = lrem - py * gridDim.x;       // class row
  const int rt = bx % rtiles;
  int pl = bx / rtiles;
  int seg = 0;
  for (; seg < p.nseg; ++seg) {
    const int t = (p.seg_c[seg] + 31) / 32;
    if (pl < t) break;
    pl -= t;
  }
  if (seg >= p.nseg) return;
  const int r0 = rt * 64, c0 = pl * 32;
  const int sc = p.seg_c[seg], cs4 = p.seg_cs[seg] * 4, dzcs4 = p.dz_cs * 4;
  const int PW = W / C, PH = H / R;
  const int npatch = g.N * PH * PW;
  const int per = (npatch + p.splitk - 1) / p.splitk;
  const int q0 = bz * per, q1 = min(npatch, q0 + per);
  // bias gradient: both classes of the row are summed into the partial row of class (py, 0); the row of class (py, 1) is zero
  const bool bias_blk = p.bias_part != nullptr && seg == 0 && c0 == 0;
  float* const brow0 = bias_blk ? p.bias_part + ((size_t)((py * 2) * p.splitk + bz) * 2) * p.n_rows : nullptr;
  float* const brow1 = bias_blk ? p.bias_part + ((size_t)((py * 2 + 1) * p.splitk + bz) * 2) * p.n_rows : nullptr;
  if (bias_blk && tid < 64 && r0 + tid < p.n_rows) {
    brow1[r0 + tid] = 0.f;
    brow1[p.n_rows + r0 + tid] = 0.f;
    if (q0 >= q1) {                                                  // an empty patch range still owns its partial rows
      brow0[r0 + tid] = 0.f;
      brow0[p.n_rows + r0 + tid] = 0.f;
    }
  }
  if (q0 >= q1) return;
  const int Ho = g.Hout, Wo = g.Wout, osy = g.osy, osx = g.osx;

  const i4v rsrc_in = make_rsrc(p.seg_ptr[seg], g.N * H * W * cs4);
  const i4v rsrc_dz = make_rsrc(p.dz, g.N * Ho * Wo * dzcs4);
  const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);

  // ---- LDS-DMA lane constants.  dz: an instruction fills 4 rows (lane >> 4) x 16 chunks (lane & 15); 8 instructions per class
  // tile, waves 0-1 fetch class (py, 0), waves 2-3 class (py, 1)
  const int zcls = wave >> 1;
  unsigned z_voff[4];
  {
    const int lrow = lane >> 4, lch = lane & 15;
    const int gch = lch ^ ((lrow & 1) | (((lrow >> 1) & 1) << 3));      // (rows of an instruction start at a multiple of 4)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = ((wave & 1) * 4 + i) * 4 + lrow;
      const int yy = row / C, xx = row % C;
      z_voff[i] = (r0 + (gch >> 1) * 8 < p.n_rows) ? (unsigned)((yy * osy * Wo + xx * osx) * dzcs4 + (p.dz_coff + r0) * 4 + gch * 16) : OOB_OFF;
    }
  }
  // halo: an instruction fills 8 rows (lane >> 3) x 8 chunks (lane & 7); halo row hr is input row y0 + hr - 1 + py
  int in_dy[NI_W], in_dx[NI_W], in_cb[NI_W];
  {
    const int lrow = lane >> 3, lch = lane & 7;
#pragma unroll
    for (int k = 0; k < NI_W; ++k) {
      const int i = wave + 4 * k;
      const int row = i * 8 + lrow;
      const int gch = lch ^ ((row >> 1) & 1);
      const int yy = row / HC, xx = row - yy * HC;
      in_dy[k] = yy - 1 + py;
      in_dx[k] = xx - 1;
      // (BHI: the lo chunks of the input rows - odd chunk indices - are never read: no fetch)
      in_cb[k] = (i < NI_IN && row < HROWS && c0 + (gch >> 1) * 8 < sc && !(BHI != 0 && (gch & 1))) ? c0 * 4 + gch * 16 : -1;
    }
  }
  int pn_ = __builtin_amdgcn_readfirstlane(q0 / (PH * PW));
  int py_ = __builtin_amdgcn_readfirstlane((q0 - pn_ * (PH * PW)) / PW);
  int px_ = __builtin_amdgcn_readfirstlane(q0 - pn_ * (PH * PW) - py_ * PW);
  auto issue = [&](int stage) {
    const int n = pn_, y0 = py_ * R, x0 = px_ * C;
    if (++px_ == PW) {
      px_ = 0;
      if (++py_ == PH) { py_ = 0; ++pn_; }
    }
    const unsigned sb = lds_base + (unsigned)(stage * STAGE);
    const int cls = py * 2 + zcls;
    const int zs = __builtin_amdgcn_readfirstlane(((n * Ho + y0 * osy + g.oy[cls]) * Wo + x0 * osx + g.ox[cls]) * dzcs4);
#pragma unroll
    for (int i = 0; i < 4; ++i) lds_dma16(rsrc_dz, sb + (wave * 4 + i) * 1024, z_voff[i], zs);
#pragma unroll
    for (int k = 0; k < NI_W; ++k) {
      if (wave + 4 * k < NI_IN) {
        const int y = y0 + in_dy[k], x = x0 + in_dx[k];
        const bool ok = in_cb[k] >= 0 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const unsigned voff = ok ? (unsigned)(((n * H + y) * W + x) * cs4 + in_cb[k]) : OOB_OFF;
        lds_dma16(rsrc_in, sb + 2 * Z_BYTES + (wave + 4 * k) * 1024, voff, 0);
      }
    }
  };

  // ---- transposed-read lane offsets (as in the per-class kernel)
  const int tg = lane >> 4, tt = lane & 15, tq = tt >> 2, tp = tt & 3;
  const int j = 8 * (tg >> 1) + tq;
  const int grp = 2 * (tg & 1) + (tp >> 1), sub = (tp & 1) * 8;
  const int zsw = (tq & 1) | (((tq >> 1) & 1) << 3);
  const int offAh = j * 256 + (((2 * (wm * 4 + grp)) ^ zsw) << 4) + sub;
  const int offAl = j * 256 + (((2 * (wm * 4 + grp) + 1) ^ zsw) << 4) + sub;
  const int jr = C >= 16 ? j : j + (j >> 3) * (HC - 8);
  int offB[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) offB[r] = jr * 128 + (((2 * grp) ^ (((r + jr) >> 1) & 1)) << 4) + sub;

  f16v acc[4], accl[4];        // [class px][ix]: (px 0: dx = 0, -1), (px 1: dx = 1, 0)
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc[t][q] = 0.f; accl[t][q] = 0.f; }
  const bool do_bias = bias_blk && iy == 0;
  f16v accb, accbl;
#pragma unroll
  for (int q = 0; q < 16; ++q) { accb[q] = 0.f; accbl[q] = 0.f; }
  const h8 ones = h8{(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};

  auto compute = [&](auto iyc, int stage) {
    constexpr int IY = decltype(iyc)::value;
    const unsigned char* Zs = smem + stage * STAGE;
    const unsigned char* Is = Zs + 2 * Z_BYTES;
#pragma unroll
    for (int kb = 0; kb < PX / 16; ++kb) {
      constexpr int KPR = C >= 16 ? C / 16 : 1;
      const int yy = C >= 16 ? kb / KPR : 2 * kb, xx0 = C >= 16 ? (kb % KPR) * 16 : 0;
      const h8 a0h = trfrag(Zs + kb * 16 * 256 + offAh, 4 * 256), a1h = trfrag(Zs + Z_BYTES + kb * 16 * 256 + offAh, 4 * 256);
      h8 a0l, a1l;
      if constexpr (BHI != 2) {
        a0l = trfrag(Zs + kb * 16 * 256 + offAl, 4 * 256);
        a1l = trfrag(Zs + Z_BYTES + kb * 16 * 256 + offAl, 4 * 256);
      }
      if (do_bias) {                                                 // wave-uniform (the bias gradient keeps both halves of dz in every mode)
        if constexpr (BHI == 2) {
          a0l = trfrag(Zs + kb * 16 * 256 + offAl, 4 * 256);
          a1l = trfrag(Zs + Z_BYTES + kb * 16 * 256 + offAl, 4 * 256);
        }
        accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, ones, accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, ones, accb, 0, 0, 0);
        accbl = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, ones, accbl, 0, 0, 0);
        accbl = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, ones, accbl, 0, 0, 0);
      }
      // tap row iy reads halo row yy + 1 - iy for either class row (dy = {0,-1} + py, halo row 0 = input row y0 - 1 + py)
      const int hb0 = (yy + 1 - IY) * HC + xx0 + 1;                  // halo row of pixel 0 of this K block at dx = 0
      const int om = (hb0 - 1) * 128 + offB[(hb0 - 1) & 3], oz = hb0 * 128 + offB[hb0 & 3], op = (hb0 + 1) * 128 + offB[(hb0 + 1) & 3];
      if constexpr (BHI == 2) {
        const h8 bzh = trfrag(Is + oz, 4 * 128), bmh = trfrag(Is + om, 4 * 128), bph = trfrag(Is + op, 4 * 128);
        SEG_MMA_HI(acc[0], a0h, bzh);
        SEG_MMA_HI(acc[3], a1h, bzh);
        SEG_MMA_HI(acc[1], a0h, bmh);
        SEG_MMA_HI(acc[2], a1h, bph);
      } else if constexpr (BHI == 1) {
        const h8 bzh = trfrag(Is + oz, 4 * 128), bmh = trfrag(Is + om, 4 * 128), bph = trfrag(Is + op, 4 * 128);
        SEG_MMA_SPLIT_BHI(acc[0], accl[0], a0h, a0l, bzh);
        SEG_MMA_SPLIT_BHI(acc[3], accl[3], a1h, a1l, bzh);
        SEG_MMA_SPLIT_BHI(acc[1], accl[1], a0h, a0l, bmh);
        SEG_MMA_SPLIT_BHI(acc[2], accl[2], a1h, a1l, bph);
      } else {
        const h8 bzh = trfrag(Is + oz, 4 * 128), bzl = trfrag(Is + (oz ^ 16), 4 * 128);
        SEG_MMA_SPLIT(acc[0], accl[0], a0h, a0l, bzh, bzl);
        SEG_MMA_SPLIT(acc[3], accl[3], a1h, a1l, bzh, bzl);
        const h8 bmh = trfrag(Is + om, 4 * 128), bml = trfrag(Is + (om ^ 16), 4 * 128);
        SEG_MMA_SPLIT(acc[1], accl[1], a0h, a0l, bmh, bml);
        const h8 bph = trfrag(Is + op, 4 * 128), bpl = trfrag(Is + (op ^ 16), 4 * 128);
        SEG_MMA_SPLIT(acc[2], accl[2], a1h, a1l, bph, bpl);
      }
    }
  };
  auto run = [&](auto iyc) {
    int stage = 0;
    issue(0);
    for (int q = q0; q < q1; ++q) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the loads of this patch have landed
      ring_release_barrier();
      if (q + 1 < q1) issue(stage ^ 1);
      compute(iyc, stage);
      stage ^= 1;
    }
  };
  if (iy == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});

  if constexpr (BHI != 2) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][q] += accl[t][q] * SPLIT_INV;
  }

  float sc_mul = p.scale;
  if (p.scale_ptr != nullptr) sc_mul *= *p.scale_ptr;
  float* part = (p.partial != nullptr && p.splitk > 1) ? p.partial + (size_t)bz * ((size_t)p.n_rows * g.T_total * p.Kpad) : nullptr;
  const int fr = lane & 31, fh = lane >> 5;
  if (do_bias && fr == 0) {          // every column of accb holds the same sums: the lanes of column 0 write the partial row
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = r0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh;
      if (row < p.n_rows) {
        brow0[row] = accb[q] + accbl[q] * SPLIT_INV;
        brow0[p.n_rows + row] = 0.f;
      }
    }
  }
  const int kcol = c0 + fr;
  if (kcol < sc) {
    const int stride = g.T_total * p.Kpad, row0 = r0 + wm * 32;
    float* const wbase = (part != nullptr ? part : p.dw) + (size_t)(row0 + 4 * fh) * stride + p.seg_woff[seg] + kcol;
    wgrad_mode_dispatch(wgrad_mode(p), [&](auto mc) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        wgrad_store_tile<decltype(mc)::value>(wbase + g.tap_w[(py * 2 + (t >> 1)) * 4 + iy * 2 + (t & 1)] * p.Kpad, acc[t], p.n_rows - row0, stride, sc_mul);
    });
  }
#endif
}
template __global__ void igemm_wgrad_halo_split_tc2_kernel<32>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_tc2_kernel<16>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_tc2_kernel<8>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_tc2_kernel<32, 1>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_tc2_kernel<16, 1>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_tc2_kernel<8, 1>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_tc2_kernel<32, 2>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_tc2_kernel<16, 2>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_split_tc2_kernel<8, 2>(const seg_igemm_wgrad_args);

}  // namespace

template <bool TC, int HI>
static void launch_wgrad_halo_split(int pc, dim3 grid, size_t lds, hipStream_t st, const seg_igemm_wgrad_args& a) {
  if (pc == 32) hipLaunchKernelGGL((igemm_wgrad_halo_split_kernel<32, TC, false, HI>), grid, dim3(256), lds, st, a);
  else if (pc == 16) hipLaunchKernelGGL((igemm_wgrad_halo_split_kernel<16, TC, false, HI>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((igemm_wgrad_halo_split_kernel<8, TC, false, HI>), grid, dim3(256), lds, st, a);
}
template <int HI>
static void launch_wgrad_tc2(int pc, dim3 grid2, hipStream_t st, const seg_igemm_wgrad_args& a) {
  if (pc == 32) hipLaunchKernelGGL((igemm_wgrad_halo_split_tc2_kernel<32, HI>), grid2, dim3(256), (size_t)2 * (16384 + 9 * 1024), st, a);
  else if (pc == 16) hipLaunchKernelGGL((igemm_wgrad_halo_split_tc2_kernel<16, HI>), grid2, dim3(256), (size_t)2 * (16384 + 7 * 1024), st, a);
  else hipLaunchKernelGGL((igemm_wgrad_halo_split_tc2_kernel<8, HI>), grid2, dim3(256), (size_t)2 * (16384 + 7 * 1024), st, a);
}

extern "C" int seg_igemm_wgrad_f32(const seg_igemm_wgrad_args* a, seg_stream_t stream) {
  if (const int rc = check_wgrad_args(a, "seg_igemm_wgrad_f32")) return rc;
  SEG_CHECK_ARG(a->bias_part == nullptr || ((a->variant & 255) == 2 && a->g.ncls == 4),
                "seg_igemm_wgrad_f32: bias_part needs variant 2 on a 4x4 / stride-2 transposed conv (one partial row per class and patch range)");
  hipStream_t st = (hipStream_t)stream;
  if ((a->variant & 255) == 2) {
    const int pc = seg_wgrad_halo_patch(&a->g);
    SEG_CHECK_ARG(pc != 0, "seg_igemm_wgrad_f32: variant 2 needs a 3x3 / stride-1 / 'same' conv or a 4x4 / stride-2 transposed conv, with W %% 32 == 0 and H %% 2 == 0, W %% 16 == 0 and H %% 4 == 0, or W and H %% 8 == 0");
    const long pix = (long)a->g.N * a->g.Hin * a->g.Win;
    SEG_CHECK_ARG((long)a->g.N * a->g.Hout * a->g.Wout * a->dz_cs * 4 < LDS_DMA_LIMIT, "seg_igemm_wgrad_f32: variant 2 needs operands below 1.875 GiB (seg_lds_dma_operand_limit)");
    int planes = 0;
    for (int s = 0; s < a->nseg; ++s) {
      SEG_CHECK_ARG(pix * a->seg_cs[s] * 4 < LDS_DMA_LIMIT, "seg_igemm_wgrad_f32: variant 2 needs operands below 1.875 GiB (seg_lds_dma_operand_limit)");
      planes += (a->seg_c[s] + 31) / 32;
    }
    dim3 grid((unsigned)(((a->n_rows + 63) / 64) * planes), (unsigned)a->g.ncls, (unsigned)a->splitk);
    const int ni = pc == 32 ? 17 : pc == 16 ? 14 : 13;              // LDS-DMA pieces of the halo tile
    const size_t lds = (size_t)2 * (16384 + ni * 1024);
    static const char* narrow_env = std::getenv("SEG_WGRAD_NARROW");
    const bool narrow = a->n_rows <= 32 && (narrow_env == nullptr || std::atoi(narrow_env) != 0);
    // transposed conv whose taps follow the class-pair table: both classes of a class row per block
    static const char* tc2_env = std::getenv("SEG_WGRAD_TC2");
    const bool tc2 = !narrow && seg_wgrad_halo_class_groups(&a->g) == 2 && (tc2_env == nullptr || std::atoi(tc2_env) != 0);
    // operand halves (64-channel tiles): 0 both operands whole (three MFMAs per product), 1 the input as its hi half (two), 2 dz as well (one)
    const int hi = narrow ? 0 : ((a->variant & SEG_VARIANT_IN_HI) ? ((a->variant & SEG_VARIANT_DZ_HI) ? 2 : 1) : 0);
    SEG_CHECK_ARG(!(a->variant & SEG_VARIANT_DZ_HI) || (a->variant & SEG_VARIANT_IN_HI), "seg_igemm_wgrad_f32: SEG_VARIANT_DZ_HI comes with SEG_VARIANT_IN_HI (one MFMA per product)");
    if (narrow && a->g.ncls == 1) {
      if (pc == 32) hipLaunchKernelGGL((igemm_wgrad_halo_split_kernel<32, false, true>), grid, dim3(256), lds, st, *a);
      else if (pc == 16) hipLaunchKernelGGL((igemm_wgrad_halo_split_kernel<16, false, true>), grid, dim3(256), lds, st, *a);
      else hipLaunchKernelGGL((igemm_wgrad_halo_split_kernel<8, false, true>), grid, dim3(256), lds, st, *a);
    } else if (narrow) {
      if (pc == 32) hipLaunchKernelGGL((igemm_wgrad_halo_split_kernel<32, true, true>), grid, dim3(256), lds, st, *a);
      else if (pc == 16) hipLaunchKernelGGL((igemm_wgrad_halo_split_kernel<16, true, true>), grid, dim3(256), lds, st, *a);
      else hipLaunchKernelGGL((igemm_wgrad_halo_split_kernel<8, true, true>), grid, dim3(256), lds, st, *a);
    } else if (tc2) {
      dim3 grid2(grid.x, 2, grid.z);
      if (hi == 2) launch_wgrad_tc2<2>(pc, grid2, st, *a);
      else if (hi == 1) launch_wgrad_tc2<1>(pc, grid2, st, *a);
      else launch_wgrad_tc2<0>(pc, grid2, st, *a);
    } else if (a->g.ncls == 1) {
      if (hi == 2) launch_wgrad_halo_split<false, 2>(pc, grid, lds, st, *a);
      else if (hi == 1) launch_wgrad_halo_split<false, 1>(pc, grid, lds, st, *a);
      else launch_wgrad_halo_split<false, 0>(pc, grid, lds, st, *a);
    } else {
      if (hi == 2) launch_wgrad_halo_split<true, 2>(pc, grid, lds, st, *a);
      else if (hi == 1) launch_wgrad_halo_split<true, 1>(pc, grid, lds, st, *a);
      else launch_wgrad_halo_split<true, 0>(pc, grid, lds, st, *a);
    }
    SEG_CHECK_LAUNCH("seg_igemm_wgrad_f32 (halo)");
    return seg_launch_wgrad_reduce(*a, st);
  }
  const int rtiles = (a->n_rows + a->tile_m - 1) / a->tile_m;
  int ktiles = 0;
  for (int s = 0; s < a->nseg; ++s) ktiles += (a->seg_c[s] + a->tile_n - 1) / a->tile_n;
  dim3 grid((unsigned)(rtiles * ktiles), (unsigned)(a->g.ncls * a->g.ntaps), (unsigned)a->splitk), block(256, 1, 1);
  const int bm = a->tile_m, bn = a->tile_n;
  const size_t lds = (size_t)2 * 2 * 32 * ((bm * 2 + 64) + (bn * 2 + 64));
  if (bm == 128 && bn == 128) hipLaunchKernelGGL((igemm_wgrad_split_kernel<128, 128>), grid, block, lds, st, *a);
  else if (bm == 128 && bn == 64) hipLaunchKernelGGL((igemm_wgrad_split_kernel<128, 64>), grid, block, lds, st, *a);
  else if (bm == 64 && bn == 128) hipLaunchKernelGGL((igemm_wgrad_split_kernel<64, 128>), grid, block, lds, st, *a);
  else hipLaunchKernelGGL((igemm_wgrad_split_kernel<64, 64>), grid, block, lds, st, *a);
  SEG_CHECK_LAUNCH("seg_igemm_wgrad_f32");
  return seg_launch_wgrad_reduce(*a, st);
}
