// Halo-tile forward / data-gradient kernel, wave-specialised generation for layers with <= 64 output channels (tile 512 px x
// 64 channels, variant 3 of seg_igemm_fwd with tile_m = 512): the full-resolution level of the UNet family, where 36 % of
// the UNet++ MACs sit and where the round-1 kernel is furthest from the MFMA roofline.
//
// Why (timing experiments on the round-1 structure, tools/op_bench.py --dbg, DESIGN.md section 4, X05 of C3):
//   618 us as it runs; 473 us without the halo-tile LDS-DMA, 519 us without the weight LDS-DMA, 419 us without either.
// With 64 output channels a halo tile is contracted against only 64 x 9 x 32 weights, so the LDS-DMA instructions (1 KiB
// each, 60-185 cycles of the issuing wave's in-order stream, MI355X guide) are twice as dense per MFMA as on the 128-channel
// tile; a third of the kernel is the cost of issuing loads from the waves that should be issuing MFMAs.
//
// Here a 768-thread block has EIGHT compute waves - two per SIMD, each 64 px x 64 channels (2 x 2 MFMA tiles, 64 accumulator
// registers: three waves per SIMD fit the register file) - and FOUR loader waves that do nothing but address arithmetic,
// LDS-DMA and the counted vmcnt waits.  The block owns a 512-pixel patch (16 x 32 or 32 x 16): one weight stage now serves
// twice the pixels, the halo overhead falls from 1.33 to 1.20, and one s_barrier per step of TG taps joins all twelve waves:
// it publishes stage s to the compute waves and hands the ring slot of stage s - 1 (and, at a chunk's first step, the other
// halo buffer) back to the loaders.  LDS: 2 x 39 KB halo tiles + 3 weight stages = 114 KB, one block per CU.
#include "igemm_common.h"

namespace {

template <int PW, int NT, int TG, int NSLOT>
__global__ __launch_bounds__(768, 3) void igemm_fwd5_kernel(const seg_igemm_fwd_args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BN = 64, NCW = 8, PIX = 64 * NCW;          // compute waves along the pixels, one block of 64 output channels
  constexpr int PH = PIX / PW, HC = PW + 2, HROWS = (PH + 2) * HC;
  constexpr int NI_A = (HROWS + 15) / 16;                  // LDS-DMA wave instructions per halo tile (16 rows of 64 B)
  constexpr int NA_HI = (NI_A + 3) / 4, NA_REM = NI_A % 4;  // per loader wave: NA_HI for waves < NA_REM (all when NA_REM == 0), else NA_HI - 1
  constexpr int A_BYTES = NI_A * 1024;
  constexpr int B_BYTES = TG * BN * 64, NB_W = B_BYTES / 4096;     // weight stage; LDS-DMA instructions per loader wave and stage
  constexpr int NSTEP = NT / TG;                           // NSLOT weight stages in the LDS ring: NSLOT - 1 issued ahead
  static_assert(TG * NSTEP == NT && NSTEP >= 2 && NB_W * 4096 == B_BYTES, "tap groups / weight stage");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  unsigned char* const Abuf = smem;                         // 2 halo tiles
  unsigned char* const Bbuf = smem + 2 * A_BYTES;           // NSLOT weight stages

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave12 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave12 >= NCW;
  const int wave = loader ? wave12 - NCW : wave12;          // loader index 0..3 / compute wave 0..7 (= its 64-pixel slice)
  const seg_geom& g = p.g;
  const int H = g.Hl, W = g.Wl;
  const int PWn = W / PW, PHn = H / PH;
  const int npatch = g.N * PHn * PWn;
  const int ncls = gridDim.y;
  const int lin0 = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int cls = lin0 % ncls, mb = lin0 / ncls;            // class fastest: the classes of a transposed conv share the halo tile in L2
  const int pn = mb / (PHn * PWn);
  const int prem = mb - pn * (PHn * PWn);
  const int ppy = prem / PWn;
  const int y0 = ppy * PH, x0 = (prem - ppy * PWn) * PW;

  int nchunks = 0;
  for (int s = 0; s < p.nseg; ++s) nchunks += (p.seg_c[s] + 31) / 32;
  const int nsteps = nchunks * NSTEP;
  const bool do_stats = p.stats != nullptr;
  // timing experiments (tools/op_bench.py --dbg, wrong results): 8 = compute waves only keep the barriers (what the LDS-DMA
  // pipeline alone sustains), 1 = no weight LDS-DMA, 2 = no halo LDS-DMA
  const int dbg = __builtin_amdgcn_readfirstlane(p.variant >> 8);

  if (loader) {
    // =========================================================================================== loader waves
    int t_w[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) t_w[t] = g.tap_w[cls * NT + t] * p.Kpad * 2;
    int a_pix[NA_HI], a_gch[NA_HI];
    {
      const int lrow = lane >> 2, lch = lane & 3;
#pragma unroll
      for (int k = 0; k < NA_HI; ++k) {
        const int row = (wave + 4 * k) * 16 + lrow;
        const int yy = row / HC, xx = row - yy * HC;
        const int y = y0 + yy - 1, x = x0 + xx - 1;
        const bool ok = (wave + 4 * k) < NI_A && row < HROWS && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        a_pix[k] = ok ? (pn * g.Hin + y) * g.Win + x : -1;
        a_gch[k] = lch ^ ((row >> 2) & 3);
      }
    }
    unsigned b_voff[NB_W];
    int b_tap[NB_W];
    {
      const int lrow = lane >> 2, lch = lane & 3;
#pragma unroll
      for (int j = 0; j < NB_W; ++j) {
        const int r = (j * 4 + wave) * 16 + lrow;            // row inside the stage: tap-in-group * BN + channel row
        const int tt = r / BN, row = r - tt * BN;
        b_tap[j] = __builtin_amdgcn_readfirstlane(tt);
        const int gch = lch ^ ((row >> 2) & 3);
        b_voff[j] = (row < p.n_rows) ? (unsigned)((row * g.T_total * p.Kpad + gch * 8) * 2) : OOB_OFF;
        // timing experiment (dbg & 64, wrong data): every weight LDS-DMA instruction reads ONE contiguous KiB (eight whole
        // cache lines) instead of 16 rows x 64 B, as a [chunk][tap][row][32] weight layout would allow
        if (dbg & 64) b_voff[j] = (unsigned)(((j * 4 + wave) * 64 + lane) * 16);
      }
    }
    auto uniform_ptr = [](const void* q) -> void* {
      const unsigned long long v = (unsigned long long)q;
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      return (void*)(((unsigned long long)hi << 32) | lo);
    };
    const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.w), 0, __builtin_amdgcn_readfirstlane(p.n_rows * g.T_total * p.Kpad * 2), 0x00020000);
    const int imgpix = g.N * g.Hin * g.Win;
    struct Cur { int seg, c0; };
    auto advance = [&](Cur& c) {
      c.c0 += 32;
      if (c.seg < p.nseg && c.c0 >= p.seg_c[c.seg]) { c.c0 = 0; ++c.seg; }
    };
    auto issueA = [&](const Cur& c, int buf) {
      if (c.seg >= p.nseg || (dbg & 2)) return;
      const int cs2 = p.seg_cs[c.seg] * 2, sc = p.seg_c[c.seg];
      const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.seg_ptr[c.seg]), 0, __builtin_amdgcn_readfirstlane(imgpix * cs2), 0x00020000);
      unsigned char* sb = Abuf + buf * A_BYTES;
#pragma unroll
      for (int k = 0; k < NA_HI; ++k) {
        if (wave + 4 * k < NI_A) {
          const int cc = c.c0 + a_gch[k] * 8;
          const unsigned voff = (a_pix[k] >= 0 && cc < sc) ? (unsigned)(a_pix[k] * cs2 + cc * 2) : OOB_OFF;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sb + (wave + 4 * k) * 1024), 16, voff, 0, 0, 0);
        }
      }
    };
    auto issueB = [&](const Cur& c, int tg, int slot) {
      if (c.seg >= p.nseg || (dbg & 1)) return;
      unsigned char* sb = Bbuf + slot * B_BYTES;
      const int kof = (p.seg_woff[c.seg] + c.c0) * 2;
#pragma unroll
      for (int j = 0; j < NB_W; ++j) {
        int tw = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (t == tg * TG + b_tap[j]) tw = t_w[t];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(sb + (j * 4 + wave) * 1024), 16, b_voff[j], __builtin_amdgcn_readfirstlane(tw + kof), 0, 0);
      }
    };
    Cur cA{0, 0}, cB{0, 0};
    int tgB = 0;
    auto nextB = [&]() {
      if (++tgB == NSTEP) { tgB = 0; advance(cB); }
    };
    issueA(cA, 0);
    advance(cA);
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i) {
      if (i < nsteps) issueB(cB, tgB, i);
      nextB();
    }
    int s = 0, slot = 0, abuf = 0;
    for (int ck = 0; ck < nchunks; ++ck) {
      static_for<0, NSTEP>([&](auto tgc) {
        constexpr int tg = decltype(tgc)::value;
        // Before barrier s stage s must have landed; the NSLOT - 2 younger stages may still fly, and so may the next chunk's
        // halo tile while it sits among them in issue order (it is issued at a chunk's first step, BEFORE that step's weight
        // stage, and has to be complete at the next chunk's first barrier - where it is older than every stage still allowed)
        constexpr int FLY = NSLOT - 2;
        if (s + FLY < nsteps) {
          if (tg >= 1 && tg <= FLY && ck + 1 < nchunks) {
            if (NA_REM == 0 || wave < NA_REM) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FLY * NB_W + NA_HI) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FLY * NB_W + NA_HI - 1) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FLY * NB_W) : "memory");
          }
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();          // loader waves: no LDS reads of their own
        asm volatile("" ::: "memory");
        if (tg == 0 && ck + 1 < nchunks) {
          issueA(cA, abuf ^ 1);
          advance(cA);
        }
        if (s + (NSLOT - 1) < nsteps) {
          int s2 = slot + (NSLOT - 1);
          if (s2 >= NSLOT) s2 -= NSLOT;
          issueB(cB, tgB, s2);
        }
        nextB();
        slot = slot == NSLOT - 1 ? 0 : slot + 1;
        ++s;
      });
      abuf ^= 1;
    }
    // the barriers of the epilogue (the one in front of it, and the one inside its BN-statistics reduction)
    __syncthreads();
    if (do_stats) __syncthreads();
    return;
  }

  // ============================================================================================= compute waves
  int t_shift[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) t_shift[t] = g.tap_dy[cls * NT + t] * HC + g.tap_dx[cls * NT + t];
  f16v acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
  const int fr = lane & 31, fh = lane >> 5;
  const int lrowA = PW == 32 ? fr : (fr >> 4) * HC + (fr & 15);
  int arow[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int yy = PW == 32 ? wave * 2 + a : (wave * 2 + a) * 2;
    arow[a] = (yy + 1) * HC + 1;
  }
  int boff[2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) boff[b][ks] = lds_off(b * 32 + fr, ks * 2 + fh);

  int slot = 0, abuf = 0;
  for (int ck = 0; ck < nchunks; ++ck) {
    static_for<0, NSTEP>([&](auto tgc) {
      constexpr int tg = decltype(tgc)::value;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the loaders refill the slot of the step before right behind this barrier: its fragment reads must have returned)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (dbg & 8) { slot = slot == NSLOT - 1 ? 0 : slot + 1; return; }
      const unsigned char* A = Abuf + abuf * A_BYTES;
      const unsigned char* B = Bbuf + slot * B_BYTES;
      int lr = lrowA;
      asm volatile("" : "+v"(lr));      // opaque per step: keeps the (tap, tile) row offsets out of long-lived registers
#pragma unroll
      for (int tt = 0; tt < TG; ++tt) {
        constexpr int dummy = 0;
        const int t = tg * TG + tt;
        int o0[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int row = arow[a] + t_shift[t] + lr;
          o0[a] = (row << 6) + ((fh ^ ((row >> 2) & 3)) << 4);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          h8 af[2], bf[2];
#pragma unroll
          for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const h8*>(A + (o0[a] ^ (ks * 32)));
#pragma unroll
          for (int b = 0; b < 2; ++b) bf[b] = *reinterpret_cast<const h8*>(B + tt * (BN * 64) + boff[b][ks]);
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        (void)dummy;
      }
      slot = slot == NSLOT - 1 ? 0 : slot + 1;
    });
    abuf ^= 1;
  }
  __syncthreads();
  conv_epilogue<half_t, 2, 2, NCW, BN, PW, NCW>(p, acc, smem, tid, PW == 32 ? wave * 2 : wave * 4, 0, 0, wave, cls, npatch, mb, 0, 0, pn, y0, x0);
#endif
}

template __global__ void igemm_fwd5_kernel<32, 9, 3, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd5_kernel<16, 9, 3, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd5_kernel<32, 9, 3, 5>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd5_kernel<16, 9, 3, 5>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd5_kernel<32, 4, 2, 3>(const seg_igemm_fwd_args);
template __global__ void igemm_fwd5_kernel<16, 4, 2, 3>(const seg_igemm_fwd_args);

}  // namespace

// 512-pixel patches the wave-specialised kernel can tile this geometry with: 32 (16 rows x 32), 16 (32 rows x 16) or 0.
// Same eligibility as seg_fwd_halo_patch (3x3 stride-1 conv / its data gradient, 4x4 stride-2 transposed conv forward).
extern "C" int seg_fwd_halo_patch512(const seg_geom* g) {
  if (g == nullptr || seg_fwd_halo_patch(g) == 0 || g->isy != 1) return 0;       // (not the parity-plane data gradient)
  if (g->Wl % 32 == 0 && g->Hl % 16 == 0) return 32;
  if (g->Wl % 16 == 0 && g->Hl % 32 == 0) return 16;
  return 0;
}

// called by seg_launch_fwd3 (igemm_halo.hip) for tile 512 x 64 after the common argument checks
int seg_launch_fwd5(const seg_igemm_fwd_args& a, hipStream_t st) {
  const int pw = seg_fwd_halo_patch512(&a.g);
  SEG_CHECK_ARG(pw != 0 && a.n_rows <= 64 && (a.g.ntaps == 9 || a.g.ntaps == 4),
                "seg_igemm_fwd: the 512 x 64 halo tile needs <= 64 output rows, 9 or 4 taps and an image tiled by 16 x 32 or 32 x 16 patches");
  const int npatch = (int)((long)a.g.N * a.g.Hl * a.g.Wl / 512);
  dim3 grid((unsigned)npatch, (unsigned)a.g.ncls, 1);
  const int ni_a = pw == 32 ? 39 : 39;                 // (16 + 2) x 34 = 612 rows, (32 + 2) x 18 = 612 rows
  static const int nslot = std::getenv("SEG_FWD5_NSLOT") ? std::atoi(std::getenv("SEG_FWD5_NSLOT")) : 5;
  if (a.g.ntaps == 9 && nslot == 5) {         // four weight stages issued ahead: 2 x 39 + 5 x 12 = 138 KB of LDS
    const size_t lds = (size_t)2 * ni_a * 1024 + 5 * (3 * 4096);
    if (pw == 32) hipLaunchKernelGGL((igemm_fwd5_kernel<32, 9, 3, 5>), grid, dim3(768), lds, st, a);
    else hipLaunchKernelGGL((igemm_fwd5_kernel<16, 9, 3, 5>), grid, dim3(768), lds, st, a);
  } else if (a.g.ntaps == 9) {
    const size_t lds = (size_t)2 * ni_a * 1024 + 3 * (3 * 4096);
    if (pw == 32) hipLaunchKernelGGL((igemm_fwd5_kernel<32, 9, 3, 3>), grid, dim3(768), lds, st, a);
    else hipLaunchKernelGGL((igemm_fwd5_kernel<16, 9, 3, 3>), grid, dim3(768), lds, st, a);
  } else {
    const size_t lds = (size_t)2 * ni_a * 1024 + 3 * (2 * 4096);
    if (pw == 32) hipLaunchKernelGGL((igemm_fwd5_kernel<32, 4, 2, 3>), grid, dim3(768), lds, st, a);
    else hipLaunchKernelGGL((igemm_fwd5_kernel<16, 4, 2, 3>), grid, dim3(768), lds, st, a);
  }
  SEG_CHECK_LAUNCH("seg_igemm_fwd (halo tiles, wave-specialised)");
  return SEG_OK;
}
