// Halo-tile forward / data-gradient kernel, software-pipelined generation (variant 3 of seg_igemm_fwd, the default since
// round 2 for the transposed convs; igemm_halo.hip keeps the round-1 kernel, which the 3x3 layers run on).
//
// Same decomposition as igemm_fwd3_kernel - a block owns a 256-pixel patch x BN output channels, ONE input halo tile per
// 32-channel chunk serves every tap, the weights stream through an LDS ring by LDS-DMA - but the MFMA operands are read
// from LDS one HALF STEP (one tap x 16 channels of K) ahead of the MFMAs that consume them, into the other of two register
// sets.  After a barrier the wave issues MFMAs at once (their fragments are already in registers) instead of first
// waiting for ~12 ds_read_b128: the round-1 kernel exposed that latency once per 16 MFMAs and relied on the second
// resident block to cover it (PMC: 34 % of wave cycles waiting, 29 % issuing; profiles/r01i_c3_pmc_sq1.csv).
//   * the halo tile of a chunk is resident for all its taps, so its fragments can always be read ahead - across chunk
//     boundaries too (the next tile was issued at the first step of the current chunk);
//   * weight fragments of the NEXT step can be read ahead only if that stage has landed and been published by the
//     current step's barrier: the ring gets a 4th slot (three stages issued ahead, the wait before barrier s retires
//     stage s + 1).  NSLOT = 3 keeps the round-1 ring and reads the first weight fragments of a step after its barrier.
//   * blocks of the four sub-pixel classes of a transposed conv that share an input patch are adjacent in the XCD-local
//     order (class fastest): the 2nd-4th read the halo tile from L2 instead of HBM (r01i: 270 MB fetched per launch for
//     a 67 MB input).
#include "igemm_common.h"

namespace {

template <int PW, int BN, int NT, int TG, int NPL, int NSLOT, bool DBG = false>
__global__ __launch_bounds__(256, 2) void igemm_fwd4_kernel(const seg_igemm_fwd_args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int PH = 256 / PW, HC = PW + 2, HROWS = (PH + 2) * HC;
  constexpr int NI_A = (HROWS + 15) / 16;                  // LDS-DMA wave instructions per halo tile (16 rows of 64 B)
  constexpr int NA_HI = (NI_A + 3) / 4, NA_REM = NI_A % 4;  // per wave: NA_HI for waves < NA_REM (all when NA_REM == 0), else NA_HI - 1
  constexpr int A_BYTES = NI_A * 1024;
  constexpr int B_BYTES = TG * BN * 64, NB_W = B_BYTES / 4096;     // weight stage, wave instructions per wave and stage
  constexpr int NSTEP = NT / TG;
  constexpr int WMv = BN == 64 ? 4 : 2, WNv = 4 / WMv, TMw = 8 / WMv;   // A tiles (32 px) per wave; 2 B tiles (64 channels) per wave
  constexpr int NH = 2 * NT;                                // half steps per chunk: (tap, 16-deep MFMA K block)
  constexpr bool XSTEP = NSLOT == 4;                        // weight fragments are read ahead across step boundaries
  constexpr bool XCHUNK = NSTEP >= 3;                       // halo fragments are read ahead across chunk boundaries
  static_assert(TG * NSTEP == NT && NSTEP >= 2 && NB_W * 4096 == B_BYTES, "tap groups / weight stage");
  static_assert(NSLOT == 3 || NSLOT == 4, "weight ring of 3 or 4 stages");
  static_assert(NPL == 1 || (NPL == 4 && NT == 4), "parity planes: the data gradient of the 4x4 / stride-2 transposed conv");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  unsigned char* const Abuf = smem;                         // 2 halo tiles
  unsigned char* const Bbuf = smem + 2 * A_BYTES;           // NSLOT weight stages

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNv, wn = wave % WNv;
  const seg_geom& g = p.g;
  // timing experiments (tools/op_bench.py --dbg; results are WRONG with any bit set): 1 no weight DMA, 2 no halo DMA,
  // 4 no barrier, 8 no MFMA, 16 no epilogue
  // (a separate instantiation: in the production kernel dbg is the constant 0 and none of the switches leaves a branch behind)
  const int dbg = DBG ? __builtin_amdgcn_readfirstlane(p.variant >> 8) : 0;
  const int H = g.Hl, W = g.Wl, IS = NPL == 4 ? 2 : 1;
  const int PWn = W / PW, PHn = H / PH;
  const int npatch = g.N * PHn * PWn;
  // logical block order inside an XCD's contiguous range: sub-pixel class fastest (shares the input patch), then patch,
  // then the block of output channels (shares nothing but the weights' absence from L2)
  const int ncls = gridDim.y;
  const int lin0 = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int cls = lin0 % ncls, lin = lin0 / ncls;
  const int mb = lin % npatch, nb = lin / npatch;
  const int n0 = nb * BN;
  const int pn = mb / (PHn * PWn);
  const int prem = mb - pn * (PHn * PWn);
  const int ppy = prem / PWn;
  const int y0 = ppy * PH, x0 = (prem - ppy * PWn) * PW;

  int t_shift[NT], t_w[NT * NPL];
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
  int a_pix[NA_HI], a_gch[NA_HI];
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
      b_voff[j] = (n0 + row < p.n_rows) ? (unsigned)(((n0 + row) * g.T_total * p.Kpad + gch * 8) * 2) : OOB_OFF;
      if (dbg & 64) b_voff[j] = (unsigned)(((j * 4 + wave) * 64 + lane) * 16);      // timing experiment: one contiguous KiB per weight LDS-DMA
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
    if (c.seg >= p.nseg || (dbg & 2)) return;
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
  auto issueB = [&](const Cur& c, int tg, int slot) {
    if (c.seg >= p.nseg || (dbg & 1)) return;
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

  f16v acc[TMw][2];
#pragma unroll
  for (int a = 0; a < TMw; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

  // ---- fragment addressing.  A: halo row of (tile a, lane row fr) under tap t = arow[a] + t_shift[t] + lrow
  const int fr = lane & 31, fh = lane >> 5;
  const int lrowA = PW == 32 ? fr : (fr >> 4) * HC + (fr & 15);
  int arow[TMw];
#pragma unroll
  for (int a = 0; a < TMw; ++a) {
    const int yy = PW == 32 ? wm * TMw + a : (wm * TMw + a) * 2;
    arow[a] = (yy + 1) * HC + 1;
  }
  int boff[2][2];     // weight rows are tile-aligned: the swizzle depends on fr only
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) boff[b][ks] = lds_off(wn * 64 + b * 32 + fr, ks * 2 + fh);

  // two register sets of MFMA operands: set (h & 1) feeds half step h while set ((h + 1) & 1) is being filled
  h8 fa[2][TMw], fb[2][2];
  // timing experiment (dbg & 32, wrong results): every halo fragment address = one of two precomputed lane registers + an
  // immediate, as a halo tile whose row pitch is a multiple of 16 pixels would allow (the swizzle then does not depend on
  // the tap shift) - measures what the ~6 VALU instructions per fragment read cost
  int xbase[2];
  {
    const int row = arow[0] + lrowA - HC - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) xbase[ks] = (row << 6) + (((fh ^ ((row >> 2) & 3)) << 4) ^ (ks * 32));
  }
  auto loadA = [&](int set, int abuf, int shift, int ks, auto tc) {
    const unsigned char* A = Abuf + abuf * A_BYTES;
    if (dbg & 32) {
      constexpr int t = decltype(tc)::value;
      constexpr int sh = (NT == 9 ? (t / 3) * HC + (t % 3) : (t >> 1) * HC + (t & 1)) * 64;       // (dy + 1) * HC + dx + 1 rows
      const unsigned char* Ax = A + xbase[ks];
#pragma unroll
      for (int a = 0; a < TMw; ++a) fa[set][a] = *reinterpret_cast<const h8*>(Ax + sh + a * (PW == 32 ? HC : 2 * HC) * 64);
      return;
    }
    int lr = lrowA;
    asm volatile("" : "+v"(lr));      // opaque: keeps the (tap, tile) row offsets from being hoisted out of the K loop into registers
#pragma unroll
    for (int a = 0; a < TMw; ++a) {
      const int row = arow[a] + shift + lr;
      const int o0 = (row << 6) + ((fh ^ ((row >> 2) & 3)) << 4);
      fa[set][a] = *reinterpret_cast<const h8*>(A + (o0 ^ (ks * 32)));
    }
  };
  auto loadB = [&](int set, int slot, int tt, int ks) {
    const unsigned char* B = Bbuf + slot * B_BYTES;
#pragma unroll
    for (int b = 0; b < 2; ++b) fb[set][b] = *reinterpret_cast<const h8*>(B + tt * (BN * 64) + boff[b][ks]);
  };
  auto mfma = [&](int set) {
    if (dbg & 8) {
#pragma unroll
      for (int a = 0; a < TMw; ++a) asm volatile("" ::"v"(fa[set][a]));
#pragma unroll
      for (int b = 0; b < 2; ++b) asm volatile("" ::"v"(fb[set][b]));
      return;
    }
#pragma unroll
    for (int a = 0; a < TMw; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][a], fb[set][b], acc[a][b], 0, 0, 0);
  };

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
#pragma unroll
  for (int i = 0; i < NSLOT - 1; ++i) {
    if (i < nsteps) issueB(cB, tgB, i);
    nextB();
  }

  int s = 0, slot = 0, abuf = 0;
  for (int ck = 0; ck < nchunks; ++ck) {
    Cur cN = cC;
    advance(cN);
    static_for<0, NH>([&](auto hc) {
      constexpr int h = decltype(hc)::value;
      constexpr int t = h >> 1, ks = h & 1, tg = t / TG, tt = t % TG;
      if constexpr (tt == 0 && ks == 0) {
        // ---- step start: retire every LDS-DMA older than the newest weight stage (and, while it may still fly, the
        // next chunk's halo tile issued right after the weight stage of this chunk's first step), publish, issue ahead
        if (s + (NSLOT - 2) < nsteps) {
          if (tg >= 1 && tg <= 2 && tg <= NSTEP - (XCHUNK ? 2 : 1) && ck + 1 < nchunks) {
            if (NA_REM == 0 || wave < NA_REM) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB_W + NA_HI) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB_W + NA_HI - 1) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB_W) : "memory");
          }
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ring_release_barrier(!(dbg & 4));
        if (s + (NSLOT - 1) < nsteps) {
          int s2 = slot + (NSLOT - 1);
          if (s2 >= NSLOT) s2 -= NSLOT;
          issueB(cB, tgB, s2);
        }
        nextB();
        if (tg == 0 && ck + 1 < nchunks) {
          issueA(cA, abuf ^ 1);
          advance(cA);
        }
        // fragments nobody could read ahead
        if (h == 0 && (ck == 0 || !XCHUNK)) loadA(0, abuf, t_shift[0], 0, std::integral_constant<int, 0>{});
        if ((h == 0 && ck == 0) || !XSTEP || (h == 0 && !XCHUNK)) loadB(h & 1, slot, 0, 0);
      }
      // ---- read ahead: the operands of half step h + 1
      if constexpr (h + 1 < NH) {
        constexpr int t2 = (h + 1) >> 1, ks2 = (h + 1) & 1, tg2 = t2 / TG, tt2 = t2 % TG;
        loadA((h + 1) & 1, abuf, t_shift[t2], ks2, std::integral_constant<int, t2>{});
        if constexpr (tg2 == tg) {
          loadB((h + 1) & 1, slot, tt2, ks2);
        } else if constexpr (XSTEP) {
          int s1 = slot + 1;
          if (s1 >= NSLOT) s1 -= NSLOT;
          loadB((h + 1) & 1, s1, 0, 0);
        }
      } else if constexpr (XCHUNK) {
        if (ck + 1 < nchunks) {
          int sh0 = t_shift[0];
          if (NPL > 1) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
              if (cN.pl == pl) sh0 = pl_shift(pl, 0);
          }
          loadA(0, abuf ^ 1, sh0, 0, std::integral_constant<int, 0>{});
          if constexpr (XSTEP) {
            int s1 = slot + 1;
            if (s1 >= NSLOT) s1 -= NSLOT;
            loadB(0, s1, 0, 0);
          }
        }
      }
      mfma(h & 1);
      if constexpr (tt == TG - 1 && ks == 1) {
        slot = slot == NSLOT - 1 ? 0 : slot + 1;
        ++s;
      }
    });
    abuf ^= 1;
    cC = cN;
    if (NPL > 1) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        int sh = pl_shift(0, t);
#pragma unroll
        for (int pl = 1; pl < NPL; ++pl)
          if (cC.pl == pl) sh = pl_shift(pl, t);
        t_shift[t] = sh;
      }
    }
  }
  __syncthreads();
  if (dbg & 16) {
    if (acc[0][0][0] == 12345.678f) ((half_t*)p.out)[tid] = (half_t)acc[TMw - 1][1][3];     // keeps the accumulators alive
    return;
  }
  conv_epilogue<half_t, TMw, 2, WMv, BN, PW>(p, acc, smem, tid, PW == 32 ? wm * TMw : wm * TMw * 2, wn * 64, n0, wm, cls, npatch, mb, 0, 0, pn, y0, x0);
#endif
}

#define SEG_FWD4_INST(NT, TG, NPL, NSLOT)                                                                     \
  template __global__ void igemm_fwd4_kernel<32, 128, NT, TG, NPL, NSLOT>(const seg_igemm_fwd_args);          \
  template __global__ void igemm_fwd4_kernel<16, 128, NT, TG, NPL, NSLOT>(const seg_igemm_fwd_args);          \
  template __global__ void igemm_fwd4_kernel<32, 64, NT, TG, NPL, NSLOT>(const seg_igemm_fwd_args);           \
  template __global__ void igemm_fwd4_kernel<16, 64, NT, TG, NPL, NSLOT>(const seg_igemm_fwd_args);           \
  template __global__ void igemm_fwd4_kernel<32, 128, NT, TG, NPL, NSLOT, true>(const seg_igemm_fwd_args);    \
  template __global__ void igemm_fwd4_kernel<32, 64, NT, TG, NPL, NSLOT, true>(const seg_igemm_fwd_args);
SEG_FWD4_INST(9, 1, 1, 4)      // 3x3 conv and its data gradient
SEG_FWD4_INST(4, 1, 1, 4)      // one sub-pixel class of the 4x4 / stride-2 transposed conv
SEG_FWD4_INST(4, 1, 4, 4)      // ... and its data gradient (four parity planes)
// (round 4: the 3-taps-per-barrier and 2-taps-per-barrier groupings of the 64-channel tile - SEG_FWD4_TG3 / SEG_FWD4_TG2 - lost
// their A/Bs in round 2 and are gone with their instances)
#undef SEG_FWD4_INST

template <int NT, int NPL>
void launch_fwd4_nt(const seg_igemm_fwd_args& a, int pw, dim3 grid, hipStream_t st) {
  const int ni_a = pw == 32 ? 22 : 21;
  const bool dbgk = (a.variant >> 8) != 0 && pw == 32;     // timing experiments: the instantiation that keeps the switches
  if (a.tile_n == 128) {
    const size_t lds = (size_t)2 * ni_a * 1024 + 4 * 8192;
    if (dbgk) hipLaunchKernelGGL((igemm_fwd4_kernel<32, 128, NT, 1, NPL, 4, true>), grid, dim3(256), lds, st, a);
    else if (pw == 32) hipLaunchKernelGGL((igemm_fwd4_kernel<32, 128, NT, 1, NPL, 4>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((igemm_fwd4_kernel<16, 128, NT, 1, NPL, 4>), grid, dim3(256), lds, st, a);
  } else {
    const size_t lds = (size_t)2 * ni_a * 1024 + 4 * 4096;
    if (dbgk) hipLaunchKernelGGL((igemm_fwd4_kernel<32, 64, NT, 1, NPL, 4, true>), grid, dim3(256), lds, st, a);
    else if (pw == 32) hipLaunchKernelGGL((igemm_fwd4_kernel<32, 64, NT, 1, NPL, 4>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((igemm_fwd4_kernel<16, 64, NT, 1, NPL, 4>), grid, dim3(256), lds, st, a);
  }
}

}  // namespace

// called by seg_launch_fwd3 (igemm_halo.hip) after the argument checks; grid = (patches x channel blocks, classes)
int seg_launch_fwd4(const seg_igemm_fwd_args& a, int pw, dim3 grid, hipStream_t st) {
  if (a.g.ntaps == 16) launch_fwd4_nt<4, 4>(a, pw, grid, st);
  else if (a.g.ntaps == 9) launch_fwd4_nt<9, 1>(a, pw, grid, st);
  else launch_fwd4_nt<4, 1>(a, pw, grid, st);
  return 0;
}
