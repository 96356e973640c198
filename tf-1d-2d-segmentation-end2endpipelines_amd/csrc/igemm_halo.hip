// Halo-tile forward / data-gradient kernel (variant 3 of seg_igemm_fwd) for gfx950: 3x3 / stride-1 convolutions, the four
// sub-pixel classes of the 4x4 / stride-2 transposed convolution and its data gradient.  See igemm_fwd.hip for the entry point.
#include "igemm_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// forward / data gradient of the 3x3 / stride-1 / 'same' convolutions, v3 (fp16): halo tiles.  A block owns a
// 256-pixel patch (PH rows x PW columns) x BN output channels.  Per 32-channel chunk of the (virtual) concat ONE
// input halo tile [(PH+2)(PW+2) px][32 ch] is brought in by LDS-DMA and serves all nine taps (a tap is a row offset
// into the tile), instead of nine separately fetched, shifted copies; only the weights are streamed per step of TG
// taps (3-deep ring, two steps of loads in flight, counted vmcnt, one barrier per step).  That cuts the L2 -> LDS
// traffic per MFMA 2-3x and, with TG = 3, the barriers per MFMA 3x against the per-tap kernel above.
// Waves: BN = 128: 2 (M) x 2 (N), each 128 px x 64 channels; BN = 64: 4 x 1, each 64 px x 64 channels.
// ---------------------------------------------------------------------------------------------------------
// Parity mode (SPLIT, BN = 64 or 32): the tensors are split pairs and the loaders see twice the channels (igemm_common.h), so a
// "32-channel chunk" is 16 real channels as hi / lo 16-B chunks; per tap and tile pair three MFMAs into two accumulator sets.
// BN = 32 (parity mode, layers of at most 32 output channels - MultiResUNet's 8 / 17 / 26-filter convs at full resolution): a wave owns
// 64 px x ONE 32-channel tile - half the MFMAs and B-fragment reads of the 64-channel tile, whose upper half would multiply zeros - and
// the weight stage has 32 rows per tap (round 6; until then 64 with the upper half out of range: no memory traffic, but the LDS-DMA
// stream is bound by instructions issued - 8 per stage now, 6 of them real, against 12).
// HI (parity mode, data gradients only): two MFMAs per product instead of three.
//   HI = 1 (variant bit SEG_VARIANT_A_HI): the A operand - the incoming gradient dz - is read as its hi half alone,
//           a b ~= ah bh + ah bl / 2048: the `al` fragment reads and the `al bh` MFMA go (a third of both).  dz spans the whole fp16 range
//           below the loss scale: elements under 2^-14 lose bits in their hi half, under 2^-24 everything.
//   HI = 2 (SEG_VARIANT_B_HI): the B operand - the transposed weights, O(1) numbers in fp16's normal range - as its hi half,
//           a b ~= ah bh + al bh / 2048: the `bl` reads and the `ah bl` MFMA go; dz keeps its 22 bits and its range.
// NOT the forward pass.
template <int PW, int BN, int NT, int TG, int NPL, bool SPLIT = false, int HI = 0>
__global__ __launch_bounds__(256, 2) void igemm_fwd3_kernel(const seg_igemm_fwd_args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(SPLIT ? (BN == 64 || BN == 32) : (BN == 64 || BN == 128), "parity mode: 64 x 64 (or 64 x 32) accumulators per wave, twice");
  static_assert(SPLIT || HI == 0, "the hi-only operands are parity-mode forms");
  constexpr bool AHI = HI == 1, BHI = HI == 2;
  typedef std::conditional_t<SPLIT, split_t, half_t> TO;
  constexpr int PH = 256 / PW, HC = PW + 2, HROWS = (PH + 2) * HC;
  constexpr int NI_A = (HROWS + 15) / 16;                  // LDS-DMA wave instructions per halo tile (16 rows of 64 B)
  constexpr int NA_HI = (NI_A + 3) / 4, NA_REM = NI_A % 4;  // per wave: NA_HI for waves < NA_REM (all when NA_REM == 0), else NA_HI - 1
  constexpr int A_BYTES = NI_A * 1024;
  constexpr int BROWS = BN;                                // rows of a weight stage per tap (BN = 32: a stage is padded with masked LDS-DMA instructions, below)
  // BHI: the stage holds the hi chunks only - rows of 32 B (chunk 0 = channels 0-7, chunk 1 = channels 8-15 of the K chunk), an LDS-DMA
  // instruction fills 32 rows instead of 16: half the weight instructions of the stream that bounds these kernels (the masked-lane
  // form - same instructions, lo lanes out of range - measured neutral: the stream is bound by instructions, not bytes).  A stage whose
  // instructions do not divide by the four waves is padded with fully masked ones (3x3: 6 real of 8, against 12).
  constexpr int BROWB = BHI ? 32 : 64;                     // bytes per weight row in LDS
  constexpr int B_REAL = TG * BROWS * BROWB;               // bytes of a stage that are read
  constexpr int NB_W = (B_REAL + 4095) / 4096;             // LDS-DMA wave instructions per wave and stage
  constexpr int B_BYTES = NB_W * 4096;                     // ... and the LDS a stage occupies
  constexpr int NSTEP = NT / TG;
  constexpr int WMv = BN <= 64 ? 4 : 2, WNv = 4 / WMv, TMw = 8 / WMv;   // A tiles (32 px) per wave
  constexpr int TNw = BN == 32 ? 1 : 2;                    // B tiles (32 channels) per wave
  static_assert(TG * NSTEP == NT && NSTEP >= 2 && (BHI || BN == 32 || B_REAL == B_BYTES), "tap groups / weight stage");
  static_assert(NPL == 1 || (NPL == 4 && NT == 4), "parity planes: the data gradient of the 4x4 / stride-2 transposed conv");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  unsigned char* const Abuf = smem;                         // 2 halo tiles
  unsigned char* const Bbuf = smem + 2 * A_BYTES;           // 3 weight stages

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNv, wn = wave % WNv;
  const seg_geom& g = p.g;
  // NPL = 4: the input (dY of a 4x4 / stride-2 transposed conv) is read as four parity planes (py, px) of the logical
  // size Hl x Wl, plane pixel (i, j) = input pixel (2 i + py, 2 j + px); the K loop runs plane by plane and every
  // plane brings 2 x 2 of the 16 taps, at halo shifts floor((ky - 1) / 2), floor((kx - 1) / 2) in {-1, 0, 1}
  const int H = g.Hl, W = g.Wl, IS = NPL == 4 ? 2 : 1;
  const int PWn = W / PW, PHn = H / PH;
  const int npatch = g.N * PHn * PWn;
  const int lin0 = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  // sub-pixel class of a transposed conv (0 otherwise).  Parity mode: class-fastest - the four classes of a patch run next to each
  // other on one XCD and share the halo tile in its L2 (as igemm_halo2.hip does for the fp16 mode)
  const int ncls_ = (int)gridDim.y;
  const int cls = SPLIT ? lin0 % ncls_ : lin0 / (int)gridDim.x, lin = SPLIT ? lin0 / ncls_ : lin0 - cls * (int)gridDim.x;
  const int mb = lin % npatch, nb = lin / npatch;
  const int n0 = nb * BN;
  const int pn = mb / (PHn * PWn);
  const int prem = mb - pn * (PHn * PWn);
  const int ppy = prem / PWn;
  const int y0 = ppy * PH, x0 = (prem - ppy * PWn) * PW;

  // tap t of the geometry = halo row shift dy * HC + dx (offsets in {-1, 0, 1}) and weight tap index
  int t_shift[NT], t_w[NT * NPL];
  // geometry index of tap i = 2 iy + ix of plane pl = 2 py + px (the tap table of model.geom_convT_dgrad, checked by
  // the launcher): ky = {1,3}[iy] for py = 0, {0,2}[iy] for py = 1, likewise kx
  auto pl_e = [](int pl, int i) { return (2 * (i >> 1) + 1 - (pl >> 1)) * 4 + (2 * (i & 1) + 1 - (pl & 1)); };
  auto pl_shift = [&](int pl, int i) {
    const int e = pl_e(pl, i), dy = (e >> 2) - 1, dx = (e & 3) - 1;
    return (dy >> 1) * HC + (dx >> 1);
  };
#pragma unroll
  for (int t = 0; t < NT; ++t) t_shift[t] = NPL == 1 ? g.tap_dy[cls * NT + t] * HC + g.tap_dx[cls * NT + t] : pl_shift(0, t);
#pragma unroll
  for (int t = 0; t < NT * NPL; ++t) t_w[t] = g.tap_w[NPL == 1 ? cls * NT + t : pl_e(t / NT, t % NT)] * p.Kpad * 2;

  // ---- halo-tile LDS-DMA lane constants: instruction i fills LDS rows 16 i .. 16 i + 15, lane -> (row, 16-B chunk)
  int a_pix[NA_HI], a_gch[NA_HI];        // pixel index relative to image 0 (or -1: outside the image / the tile), fetched chunk
  {
    const int lrow = lane >> 2, lch = lane & 3;
#pragma unroll
    for (int k = 0; k < NA_HI; ++k) {
      const int row = (wave + 4 * k) * 16 + lrow;
      const int yy = row / HC, xx = row - yy * HC;
      const int y = y0 + yy - 1, x = x0 + xx - 1;
      const bool ok = (wave + 4 * k) < NI_A && row < HROWS && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
      a_pix[k] = ok ? (pn * g.Hin + y * IS) * g.Win + x * IS : -1;
      a_gch[k] = lch ^ ((row >> 2) & 3);
    }
  }
  // ---- weight-stage lane constants: instruction j of this wave fills rows of tap (j*4+wave)*16 / BN of the group
  unsigned b_voff[NB_W];
  int b_tap[NB_W];
  {
    const int lrow = BHI ? lane >> 1 : lane >> 2, lch = BHI ? lane & 1 : lane & 3;
    constexpr int RPI = BHI ? 32 : 16;                     // rows per instruction
#pragma unroll
    for (int j = 0; j < NB_W; ++j) {
      const int r = (j * 4 + wave) * RPI + lrow;           // row inside the stage: tap-in-group * BN + channel row
      const int tt = r / BROWS, row = r - tt * BROWS;
      b_tap[j] = __builtin_amdgcn_readfirstlane(tt < TG ? tt : 0);
      // the 16-B chunk this lane fetches: [hi 0-7 | lo 0-7 | hi 8-15 | lo 8-15] of the row's K chunk; BHI: the hi chunks, unswizzled
      // (64 lanes then read a contiguous KiB of fragments: conflict-free as it lies)
      const int gch = BHI ? 2 * lch : lch ^ ((row >> 2) & 3);
      b_voff[j] = (tt < TG && row < BN && n0 + row < p.n_rows) ? (unsigned)(((n0 + row) * g.T_total * p.Kpad + gch * 8) * 2) : OOB_OFF;
    }
  }
  auto uniform_ptr = [](const void* q) -> void* {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
  };
  const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.w), 0, __builtin_amdgcn_readfirstlane(p.n_rows * g.T_total * p.Kpad * 2), 0x00020000);
  const int imgpix = g.N * g.Hin * g.Win;

  int nchunks = 0;
  for (int s = 0; s < p.nseg; ++s) nchunks += (p.seg_c[s] + 31) / 32;
  nchunks *= NPL;

  // (plane, segment, first channel) of a chunk index, advanced incrementally
  struct Cur { int seg, c0, pl; };
  auto advance = [&](Cur& c) {
    c.c0 += 32;
    if (c.seg < p.nseg && c.c0 >= p.seg_c[c.seg]) {
      c.c0 = 0;
      ++c.seg;
      if (NPL > 1 && c.seg == p.nseg && c.pl + 1 < NPL) { c.seg = 0; ++c.pl; }
    }
  };
  auto issueA = [&](const Cur& c, int buf) {
    if (c.seg >= p.nseg) return;
    const int cs2 = p.seg_cs[c.seg] * 2, sc = p.seg_c[c.seg];
    const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.seg_ptr[c.seg]), 0, __builtin_amdgcn_readfirstlane(imgpix * cs2), 0x00020000);
    unsigned char* sb = Abuf + buf * A_BYTES;
    const int ploff = NPL == 1 ? 0 : (c.pl >> 1) * g.Win + (c.pl & 1);
#pragma unroll
    for (int k = 0; k < NA_HI; ++k) {
      if (wave + 4 * k < NI_A) {
        const int cc = c.c0 + a_gch[k] * 8;
        const unsigned voff = (a_pix[k] >= 0 && cc < sc) ? (unsigned)((a_pix[k] + ploff) * cs2 + cc * 2) : OOB_OFF;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sb + (wave + 4 * k) * 1024), 16, voff, 0, 0, 0);
      }
    }
  };
  // weights of step (chunk c, tap group tg): rows [tap-in-group][BN channels] x 32 K columns
  auto issueB = [&](const Cur& c, int tg, int slot) {
    if (c.seg >= p.nseg) return;
    unsigned char* sb = Bbuf + slot * B_BYTES;
    const int kof = (p.seg_woff[c.seg] + c.c0) * 2;
#pragma unroll
    for (int j = 0; j < NB_W; ++j) {
      int tw = 0;
#pragma unroll
      for (int t = 0; t < NT * NPL; ++t)
        if (t == c.pl * NT + tg * TG + b_tap[j]) tw = t_w[t];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(sb + (j * 4 + wave) * 1024), 16, b_voff[j], __builtin_amdgcn_readfirstlane(tw + kof), 0, 0);
    }
  };

  f16v acc[TMw][TNw];
#pragma unroll
  for (int a = 0; a < TMw; ++a)
#pragma unroll
    for (int b = 0; b < TNw; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
  f16v accl[SPLIT ? TMw : 1][SPLIT ? TNw : 1];
  if constexpr (SPLIT) {
#pragma unroll
    for (int a = 0; a < TMw; ++a)
#pragma unroll
      for (int b = 0; b < TNw; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) accl[a][b][q] = 0.f;
  }

  // ---- fragment addressing.  A: halo row of (tile a, lane row fr) under tap t = arow[a] + t_shift[t] + lrow
  const int fr = lane & 31, fh = lane >> 5;
  const int lrowA = PW == 32 ? fr : (fr >> 4) * HC + (fr & 15);
  int arow[TMw];
#pragma unroll
  for (int a = 0; a < TMw; ++a) {
    const int yy = PW == 32 ? wm * TMw + a : (wm * TMw + a) * 2;
    arow[a] = (yy + 1) * HC + 1;
  }
  int boff[TNw][2];     // weight rows are tile-aligned: the swizzle depends on fr only
#pragma unroll
  for (int b = 0; b < TNw; ++b)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      boff[b][ks] = BHI ? (wn * 64 + b * 32 + fr) * 32 + fh * 16 : lds_off(wn * 64 + b * 32 + fr, SPLIT ? 2 * fh + ks : ks * 2 + fh);      // (split: ks = 0 hi, 1 lo)

  auto compute = [&](auto tgc, int abuf, int slot) {
    constexpr int tg = decltype(tgc)::value;
    const unsigned char* A = Abuf + abuf * A_BYTES;
    const unsigned char* B = Bbuf + slot * B_BYTES;
    int lr = lrowA;
    asm volatile("" : "+v"(lr));      // opaque per step: keeps the 36 (tap, tile) row offsets from being hoisted out of the K loop into registers
#pragma unroll
    for (int tt = 0; tt < TG; ++tt) {
      const int t = tg * TG + tt;
      int o0[TMw];
#pragma unroll
      for (int a = 0; a < TMw; ++a) {
        const int row = arow[a] + t_shift[t] + lr;
        o0[a] = (row << 6) + (((SPLIT ? 2 * fh : fh) ^ ((row >> 2) & 3)) << 4);
      }
      if constexpr (SPLIT) {
        h8 ah[TMw], al[AHI ? 1 : TMw], bh[TNw], bl[BHI ? 1 : TNw];
#pragma unroll
        for (int a = 0; a < TMw; ++a) {
          ah[a] = *reinterpret_cast<const h8*>(A + o0[a]);
          if constexpr (!AHI) al[a] = *reinterpret_cast<const h8*>(A + (o0[a] ^ 16));
        }
#pragma unroll
        for (int b = 0; b < TNw; ++b) {
          bh[b] = *reinterpret_cast<const h8*>(B + tt * (BROWS * BROWB) + boff[b][0]);
          if constexpr (!BHI) bl[b] = *reinterpret_cast<const h8*>(B + tt * (BROWS * 64) + boff[b][1]);
        }
#pragma unroll
        for (int a = 0; a < TMw; ++a)
#pragma unroll
          for (int b = 0; b < TNw; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
            if constexpr (!AHI) accl[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], accl[a][b], 0, 0, 0);
            if constexpr (!BHI) accl[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], accl[a][b], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          h8 af[TMw], bf[TNw];
#pragma unroll
          for (int a = 0; a < TMw; ++a) af[a] = *reinterpret_cast<const h8*>(A + (o0[a] ^ (ks * 32)));
#pragma unroll
          for (int b = 0; b < TNw; ++b) bf[b] = *reinterpret_cast<const h8*>(B + tt * (BROWS * 64) + boff[b][ks]);
#pragma unroll
          for (int a = 0; a < TMw; ++a)
#pragma unroll
            for (int b = 0; b < TNw; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
      }
    }
  };

  // ---- pipeline.  Step s = (chunk, tap group).  In flight at the top of step s: weights of s and s + 1 (ring slots
  // s % 3, (s + 1) % 3) and, when it was issued during step s - 1, the next chunk's halo tile (behind them in order)
  const int nsteps = nchunks * NSTEP;
  Cur cA{0, 0, 0};         // chunk whose halo tile is issued next
  Cur cB{0, 0, 0};         // chunk of the weight stage issued next
  Cur cC{0, 0, 0};         // chunk being computed (its plane selects the tap shifts)
  int tgB = 0;             // ... and its tap group
  auto nextB = [&]() {
    if (++tgB == NSTEP) { tgB = 0; advance(cB); }
  };
  issueA(cA, 0);
  advance(cA);
  issueB(cB, tgB, 0);
  nextB();
  if (nsteps > 1) issueB(cB, tgB, 1);
  nextB();

  int s = 0, slot = 0, abuf = 0;
  for (int ck = 0; ck < nchunks; ++ck) {
    static_for<0, NSTEP>([&](auto tgc) {
      constexpr int tg = decltype(tgc)::value;
      // everything older than the newest weight stage (and the halo tile issued after it) must have landed
      if (s + 1 < nsteps) {
        if (tg == 1 % NSTEP && ck + 1 < nchunks) {     // the next chunk's halo tile was issued at the previous step
          if (NA_REM == 0 || wave < NA_REM) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB_W + NA_HI) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB_W + NA_HI - 1) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB_W) : "memory");
        }
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      ring_release_barrier();
      if (s + 2 < nsteps) {
        int s2 = slot + 2;
        if (s2 >= 3) s2 -= 3;
        issueB(cB, tgB, s2);
      }
      nextB();
      if (tg == 0 && ck + 1 < nchunks) {
        issueA(cA, abuf ^ 1);
        advance(cA);
      }
      if (NPL > 1 && tg == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          int sh = pl_shift(0, t);
#pragma unroll
          for (int pl = 1; pl < NPL; ++pl)
            if (cC.pl == pl) sh = pl_shift(pl, t);
          t_shift[t] = sh;
        }
      }
      compute(tgc, abuf, slot);
      slot = slot == 2 ? 0 : slot + 1;
      ++s;
    });
    abuf ^= 1;
    advance(cC);
  }
  __syncthreads();
  if constexpr (SPLIT) split_combine(acc, accl);
  conv_epilogue<TO, TMw, TNw, WMv, BN, PW>(p, acc, smem, tid, PW == 32 ? wm * TMw : wm * TMw * 2, wn * 64, n0, wm, cls, npatch, mb, 0, 0, pn, y0, x0);
#endif
}
template __global__ void igemm_fwd3_kernel<32, 128, 9, 1, 1>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 128, 9, 1, 1>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 64, 9, 3, 1>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 64, 9, 3, 1>(const seg_igemm_fwd_args);
// the four sub-pixel classes of a 4x4 / stride-2 transposed conv: 2x2 taps each out of the same halo tile
template __global__ void igemm_fwd3_kernel<32, 128, 4, 1, 1>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 128, 4, 1, 1>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 64, 4, 2, 1>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 64, 4, 2, 1>(const seg_igemm_fwd_args);
// ... and its data gradient: four parity planes of dY, 2x2 taps each, accumulated into one output
template __global__ void igemm_fwd3_kernel<32, 128, 4, 1, 4>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 128, 4, 1, 4>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 64, 4, 2, 4>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 64, 4, 2, 4>(const seg_igemm_fwd_args);
// parity mode
template __global__ void igemm_fwd3_kernel<32, 64, 9, 3, 1, true>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 64, 9, 3, 1, true>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 64, 4, 2, 1, true>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 64, 4, 2, 1, true>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 64, 4, 2, 4, true>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 64, 4, 2, 4, true>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 32, 9, 3, 1, true>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 32, 9, 3, 1, true>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 32, 4, 2, 1, true>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 32, 4, 2, 1, true>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 32, 4, 2, 4, true>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 32, 4, 2, 4, true>(const seg_igemm_fwd_args);
// parity mode, data gradients with the hi-only A operand (two MFMAs per product): the 3x3 layers and the transposed conv's planes form
template __global__ void igemm_fwd3_kernel<32, 64, 9, 3, 1, true, 1>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 64, 9, 3, 1, true, 1>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 64, 4, 2, 4, true, 1>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 64, 4, 2, 4, true, 1>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 32, 9, 3, 1, true, 1>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 32, 9, 3, 1, true, 1>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 32, 4, 2, 4, true, 1>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 32, 4, 2, 4, true, 1>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 64, 9, 3, 1, true, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 64, 9, 3, 1, true, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 64, 4, 2, 4, true, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 64, 4, 2, 4, true, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 32, 9, 3, 1, true, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 32, 9, 3, 1, true, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<32, 32, 4, 2, 4, true, 2>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd3_kernel<16, 32, 4, 2, 4, true, 2>(const seg_igemm_fwd_args);

}  // namespace

// ---- v3 (halo tiles) dispatch
static int fwd3_planes_patch(const seg_geom* g);
extern "C" int seg_fwd_halo_patch(const seg_geom* g) {
  if (g == nullptr) return 0;
  if (g->isy == 2) return fwd3_planes_patch(g);
  if (g->isy != 1 || g->isx != 1 || g->Hl != g->Hin || g->Wl != g->Win) return 0;
  if (g->ncls == 1 && g->ntaps == 9) {                    // 3x3 stride-1 'same' conv or its data gradient: all nine offsets
    if (g->osy != 1 || g->osx != 1 || g->Hout != g->Hin || g->Wout != g->Win || g->oy[0] != 0 || g->ox[0] != 0) return 0;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
      const int dy = g->tap_dy[t], dx = g->tap_dx[t];
      if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return 0;
      seen |= 1u << ((dy + 1) * 3 + dx + 1);
    }
    if (seen != 0x1ffu) return 0;
  } else if (g->ncls == 4 && g->ntaps == 4) {             // 4x4 stride-2 transposed conv: four classes of 2x2 taps
    if (g->osy != 2 || g->osx != 2 || g->Hout != 2 * g->Hin || g->Wout != 2 * g->Win) return 0;
    for (int c = 0; c < 4; ++c) {
      if (g->oy[c] < 0 || g->oy[c] > 1 || g->ox[c] < 0 || g->ox[c] > 1) return 0;
      for (int t = 0; t < 4; ++t) {
        const int dy = g->tap_dy[c * 4 + t], dx = g->tap_dx[c * 4 + t];
        if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return 0;
      }
    }
  } else {
    return 0;
  }
  if (g->Win % 32 == 0 && g->Hin % 8 == 0) return 32;
  if (g->Win % 16 == 0 && g->Hin % 16 == 0) return 16;
  return 0;
}

// data gradient of the 4x4 / stride-2 transposed conv (model.geom_convT_dgrad): one class, 16 taps (ky - 1, kx - 1) in
// ky-major order over the stride-2 input
static int fwd3_planes_patch(const seg_geom* g) {
  if (g->ncls != 1 || g->ntaps != 16 || g->isy != 2 || g->isx != 2 || g->osy != 1 || g->osx != 1) return 0;
  if (g->Hin != 2 * g->Hl || g->Win != 2 * g->Wl || g->Hout != g->Hl || g->Wout != g->Wl || g->oy[0] != 0 || g->ox[0] != 0) return 0;
  for (int t = 0; t < 16; ++t)
    if (g->tap_dy[t] != (t >> 2) - 1 || g->tap_dx[t] != (t & 3) - 1) return 0;
  if (g->Wl % 32 == 0 && g->Hl % 8 == 0) return 32;
  if (g->Wl % 16 == 0 && g->Hl % 16 == 0) return 16;
  return 0;
}

template <int NT, int TG128, int TG64, int NPL = 1>
static void launch_fwd3_nt(const seg_igemm_fwd_args& a, int pw, dim3 grid, hipStream_t st) {
  const int ni_a = pw == 32 ? 22 : 21;
  if (a.tile_n == 128) {
    const size_t lds = (size_t)2 * ni_a * 1024 + 3 * (TG128 * 8192);
    if (pw == 32) hipLaunchKernelGGL((igemm_fwd3_kernel<32, 128, NT, TG128, NPL>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((igemm_fwd3_kernel<16, 128, NT, TG128, NPL>), grid, dim3(256), lds, st, a);
  } else {
    const size_t lds = (size_t)2 * ni_a * 1024 + 3 * (TG64 * 4096);
    if (pw == 32) hipLaunchKernelGGL((igemm_fwd3_kernel<32, 64, NT, TG64, NPL>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((igemm_fwd3_kernel<16, 64, NT, TG64, NPL>), grid, dim3(256), lds, st, a);
  }
}

int seg_launch_fwd3(const seg_igemm_fwd_args& a, hipStream_t st) {
  const int pw = seg_fwd_halo_patch(&a.g);
  SEG_CHECK_ARG(pw != 0, "seg_igemm_fwd: variant 3 needs a stride-1 3x3 'same' conv, a 4x4 / stride-2 transposed conv (or the data gradient of either), and W %% 32 == 0, H %% 8 == 0 (or W, H %% 16 == 0)");
  SEG_CHECK_ARG(a.tile_m == 256 && (a.tile_n == 64 || a.tile_n == 128), "seg_igemm_fwd: variant 3 tiles are 256x64 and 256x128, not %dx%d", a.tile_m, a.tile_n);
  const long pix = (long)a.g.N * a.g.Hin * a.g.Win;
  for (int s = 0; s < a.nseg; ++s) SEG_CHECK_ARG(pix * a.seg_cs[s] * 2 < LDS_DMA_LIMIT, "seg_igemm_fwd: variant 3 needs operands below 1.875 GiB (seg_lds_dma_operand_limit)");
  SEG_CHECK_ARG((long)a.n_rows * a.g.T_total * a.Kpad * 2 < LDS_DMA_LIMIT, "seg_igemm_fwd: variant 3 needs operands below 1.875 GiB (seg_lds_dma_operand_limit)");
  const int npatch = (int)((long)a.g.N * a.g.Hl * a.g.Wl / 256);
  const int nblocks = (a.n_rows + a.tile_n - 1) / a.tile_n;
  dim3 grid((unsigned)(npatch * nblocks), (unsigned)a.g.ncls, 1);
  // the software-pipelined generation (igemm_halo2.hip).  Measured on MI355X (tools/op_bench.py, same box, C3 layers):
  // transposed conv forward 4-6 % faster (its blocks run class-fastest, so the four classes of a patch share the halo tile
  // in L2) and its data gradient 0-3 %; the 3x3 layers 1-3 % SLOWER on the 128-channel tile and equal on the 64-channel
  // tile - reading fragments ahead does not pay there: the in-order issue stream (LDS-DMA, ds_read, address VALU) is what
  // bounds the kernel, not the exposed ds_read latency (ablations in docs/experiments.md section 4).
  const bool pipe = a.g.ntaps != 9;
  if (pipe || (a.variant >> 8) != 0) {
    seg_launch_fwd4(a, pw, grid, st);
    SEG_CHECK_LAUNCH("seg_igemm_fwd (halo tiles, pipelined)");
    return SEG_OK;
  }
  if (a.g.ntaps == 16) launch_fwd3_nt<4, 1, 2, 4>(a, pw, grid, st);
  else if (a.g.ntaps == 9) launch_fwd3_nt<9, 1, 3>(a, pw, grid, st);
  else launch_fwd3_nt<4, 1, 2>(a, pw, grid, st);
  SEG_CHECK_LAUNCH("seg_igemm_fwd (halo tiles)");
  return SEG_OK;
}

// ---- parity mode: `v` is the split view of the arguments (igemm_common.h: the loaders' pseudo-channels); 256 x 64 and 256 x 32 tiles
template <int NT, int TG, int NPL, int AHI = 0>
static void launch_fwd3_split_nt(const seg_igemm_fwd_args& v, int pw, dim3 grid, hipStream_t st) {
  const int ni_a = pw == 32 ? 22 : 21;
  const size_t lds = (size_t)2 * ni_a * 1024 + 3 * (TG * 4096);
  if (v.tile_n == 32) {
    if (pw == 32) hipLaunchKernelGGL((igemm_fwd3_kernel<32, 32, NT, TG, NPL, true, AHI>), grid, dim3(256), lds, st, v);
    else hipLaunchKernelGGL((igemm_fwd3_kernel<16, 32, NT, TG, NPL, true, AHI>), grid, dim3(256), lds, st, v);
    return;
  }
  if (pw == 32) hipLaunchKernelGGL((igemm_fwd3_kernel<32, 64, NT, TG, NPL, true, AHI>), grid, dim3(256), lds, st, v);
  else hipLaunchKernelGGL((igemm_fwd3_kernel<16, 64, NT, TG, NPL, true, AHI>), grid, dim3(256), lds, st, v);
}
int seg_launch_fwd3_split(const seg_igemm_fwd_args& v, hipStream_t st) {
  const int pw = seg_fwd_halo_patch(&v.g);
  SEG_CHECK_ARG(pw != 0, "seg_igemm_fwd_f32: variant 3 needs a stride-1 3x3 'same' conv, a 4x4 / stride-2 transposed conv (or the data gradient of either), and W %% 32 == 0, H %% 8 == 0 (or W, H %% 16 == 0)");
  SEG_CHECK_ARG(v.tile_m == 256 && (v.tile_n == 64 || v.tile_n == 32), "seg_igemm_fwd_f32: the variant 3 tiles are 256x64 and 256x32, not %dx%d", v.tile_m, v.tile_n);
  const long pix = (long)v.g.N * v.g.Hin * v.g.Win;
  for (int s = 0; s < v.nseg; ++s) SEG_CHECK_ARG(pix * v.seg_cs[s] * 2 < LDS_DMA_LIMIT, "seg_igemm_fwd_f32: variant 3 needs operands below 1.875 GiB (seg_lds_dma_operand_limit)");
  SEG_CHECK_ARG((long)v.n_rows * v.g.T_total * v.Kpad * 2 < LDS_DMA_LIMIT, "seg_igemm_fwd_f32: variant 3 needs operands below 1.875 GiB (seg_lds_dma_operand_limit)");
  const int npatch = (int)((long)v.g.N * v.g.Hl * v.g.Wl / 256);
  const int nblocks = (v.n_rows + v.tile_n - 1) / v.tile_n;
  dim3 grid((unsigned)(npatch * nblocks), (unsigned)v.g.ncls, 1);
  // data gradients: the A operand (dz) or the B operand (the transposed weights) as its hi half - two MFMAs per product
  const int hi = (v.variant & SEG_VARIANT_A_HI) ? 1 : (v.variant & SEG_VARIANT_B_HI) ? 2 : 0;
  SEG_CHECK_ARG(!((v.variant & SEG_VARIANT_A_HI) && (v.variant & SEG_VARIANT_B_HI)), "seg_igemm_fwd_f32: SEG_VARIANT_A_HI and SEG_VARIANT_B_HI exclude each other");
  SEG_CHECK_ARG(hi == 0 || v.g.ntaps != 4, "seg_igemm_fwd_f32: SEG_VARIANT_A_HI / _B_HI are data-gradient forms (3x3 layers, the transposed conv's planes), not the transposed conv's forward pass");
  if (v.g.ntaps == 16) { if (hi == 1) launch_fwd3_split_nt<4, 2, 4, 1>(v, pw, grid, st); else if (hi == 2) launch_fwd3_split_nt<4, 2, 4, 2>(v, pw, grid, st); else launch_fwd3_split_nt<4, 2, 4>(v, pw, grid, st); }
  else if (v.g.ntaps == 9) { if (hi == 1) launch_fwd3_split_nt<9, 3, 1, 1>(v, pw, grid, st); else if (hi == 2) launch_fwd3_split_nt<9, 3, 1, 2>(v, pw, grid, st); else launch_fwd3_split_nt<9, 3, 1>(v, pw, grid, st); }
  else launch_fwd3_split_nt<4, 2, 1>(v, pw, grid, st);
  SEG_CHECK_LAUNCH("seg_igemm_fwd_f32 (halo tiles)");
  return SEG_OK;
}
