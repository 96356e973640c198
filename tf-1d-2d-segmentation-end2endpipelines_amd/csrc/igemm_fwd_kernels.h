// Forward / data-gradient kernels of variant 0 - register-staged (64-row tiles, odd tap counts) and per-tap LDS-DMA (128- and
// 256-row tiles): the templates, shared by the fp16 instantiations (igemm_fwd.hip) and the parity mode's (igemm_fwd_split.hip).
#pragma once
#include "igemm_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// forward / dgrad
// ---------------------------------------------------------------------------------------------------------
template <bool SPLIT, int BM, int BN>
__global__ __launch_bounds__(256, (SPLIT && BM * BN < 128 * 128) ? 2 : 1) void igemm_fwd_kernel(const seg_igemm_fwd_args p) {
  typedef half_t T;                                   // operand element as the loaders see it (split: pseudo-channels)
  typedef std::conditional_t<SPLIT, split_t, half_t> TO;      // output element
  constexpr int TM = BM / 64, TN = BN / 64;
  // split mode: KS = 2 sub-tiles (16 real channels each) per K step - 24 MFMAs per wave and barrier instead of 12, eight 16-B
  // loads per thread in flight instead of four
  constexpr int KS = SPLIT ? 2 : 1;
  constexpr int EPC = ET<T>::EPC, BKS = 4 * EPC, BK = KS * BKS;      // elements per 16-B chunk / per sub-tile / per K step
  constexpr int A_SUB = BM * ROW_BYTES, B_SUB = BN * ROW_BYTES;
  constexpr int A_BYTES = KS * A_SUB, B_BYTES = KS * B_SUB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const seg_geom& g = p.g;
  const int HW = g.Hl * g.Wl;
  const int M = g.N * HW;
  const int mblocks = (M + BM - 1) / BM;
  const int lin = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int cls = lin / gridDim.x, bx = lin - cls * gridDim.x;
  const int mb = bx % mblocks, nb = bx / mblocks;
  const int m0 = mb * BM, n0 = nb * BN;

  // ---- per-thread load coordinates: row = (tid >> 2) + 64 r, 16-B chunk = tid & 3
  const int ld_row = tid >> 2, ld_ch = tid & 3;
  int a_pix[TM], a_y[TM], a_x[TM];
  bool a_ok[TM];
#pragma unroll
  for (int r = 0; r < TM; ++r) {
    int m = m0 + ld_row + 64 * r;
    bool ok = m < M;
    a_ok[r] = ok;
    int mm = ok ? m : 0;
    int n = mm / HW;
    int rem = mm - n * HW;
    int i = rem / g.Wl;
    int j = rem - i * g.Wl;
    a_pix[r] = n * g.Hin * g.Win;
    a_y[r] = i * g.isy;
    a_x[r] = j * g.isx;
  }
  size_t b_rowoff[TN];
  bool b_ok[TN];
#pragma unroll
  for (int r = 0; r < TN; ++r) {
    int row = n0 + ld_row + 64 * r;
    b_ok[r] = row < p.n_rows;
    b_rowoff[r] = (size_t)(b_ok[r] ? row : 0) * g.T_total * p.Kpad;
  }

  f16v acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
  f16v accl[SPLIT ? TM : 1][SPLIT ? TN : 1];      // split mode: the (lo x hi + hi x lo) sums, scaled by 2048
  if constexpr (SPLIT) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) accl[a][b][q] = 0.f;
  }

  int niter = 0;
  for (int s = 0; s < p.nseg; ++s) niter += ((p.seg_c[s] + BK - 1) / BK) * g.ntaps;

  u4 ra[KS][TM], rb[KS][TN];
  const T* wbase = (const T*)p.w;

  auto gload = [&](int seg, int c0, int tap) {
    const T* sp = (const T*)p.seg_ptr[seg];
    const int cs = p.seg_cs[seg];
    const int sc = p.seg_c[seg];
    const int e = cls * g.ntaps + tap;
    const int dy = g.tap_dy[e], dx = g.tap_dx[e], widx = g.tap_w[e];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cc = c0 + ks * BKS + ld_ch * EPC;
      const bool cok = cc < sc;
#pragma unroll
      for (int r = 0; r < TM; ++r) {
        int y = a_y[r] + dy, x = a_x[r] + dx;
        bool ok = cok && a_ok[r] && (unsigned)y < (unsigned)g.Hin && (unsigned)x < (unsigned)g.Win;
        size_t off = ((size_t)(a_pix[r] + y * g.Win + x)) * cs + cc;
        u4 v = {0u, 0u, 0u, 0u};
        if (ok) v = ld16(sp + off);
        ra[ks][r] = v;
      }
      // (a sub-tile past the segment's channels meets zero activations; its weights are not read either - they would come from
      // the next segment's or the next tap's columns, or from beyond the last row)
      const bool wok = KS == 1 || cok;
      const T* wp = wbase + (size_t)widx * p.Kpad + p.seg_woff[seg] + cc;
#pragma unroll
      for (int r = 0; r < TN; ++r) {
        u4 v = {0u, 0u, 0u, 0u};
        if (b_ok[r] && wok) v = ld16(wp + b_rowoff[r]);
        rb[ks][r] = v;
      }
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* A = smem + buf * (A_BYTES + B_BYTES);
    unsigned char* B = A + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int r = 0; r < TM; ++r) st16(A + ks * A_SUB + lds_off(ld_row + 64 * r, ld_ch), ra[ks][r]);
#pragma unroll
      for (int r = 0; r < TN; ++r) st16(B + ks * B_SUB + lds_off(ld_row + 64 * r, ld_ch), rb[ks][r]);
    }
  };
  const int fr = lane & 31, fh = lane >> 5;
  auto compute = [&](int buf) {
    const unsigned char* A = smem + buf * (A_BYTES + B_BYTES);
    if constexpr (SPLIT) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        mma_step_split<TM, TN>(A + ks * A_SUB, A + A_BYTES + ks * B_SUB, wm * (BM / 2), wn * (BN / 2), fr, fh, acc, accl);
    } else mma_step<TM, TN>(A, A + A_BYTES, wm * (BM / 2), wn * (BN / 2), fr, fh, acc);
  };

  // ---- main loop: iteration order (segment, channel chunk, tap) with taps innermost
  int seg = 0, c0 = 0, tap = 0;
  gload(seg, c0, tap);
  lstore(0);
  __syncthreads();
  for (int it = 0; it < niter; ++it) {
    const bool has_next = it + 1 < niter;
    if (has_next) {   // advance to the next (seg, c0, tap)
      ++tap;
      if (tap == g.ntaps) {
        tap = 0;
        c0 += BK;
        if (c0 >= p.seg_c[seg]) { c0 = 0; ++seg; }
      }
      gload(seg, c0, tap);
    }
    compute(it & 1);
    if (has_next) lstore((it + 1) & 1);
    __syncthreads();
  }

  if constexpr (SPLIT) split_combine(acc, accl);
  conv_epilogue<TO, TM, TN, 2, BN>(p, acc, smem, tid, m0 + wm * (BM / 2), wn * (BN / 2), n0, wm, cls, mblocks, mb, M, HW);
}

// ---------------------------------------------------------------------------------------------------------
// forward / dgrad, v2 (fp16): LDS-DMA staging (buffer_load_dwordx4 ... lds) into a 3-stage LDS ring, counted
// vmcnt + raw s_barrier (one barrier per K step, two steps of loads always in flight), kernel taps unrolled at
// compile time so that per-step address work is one scalar multiply per operand and two VALU ops per LDS-DMA
// (tap mask select + add).  Out-of-image taps, rows past M and channel padding use the buffer range check:
// an offset beyond num_records makes the hardware write zeros.  Tiles: 128x128 (2x2 waves) and 256x64 (4x1
// waves), every wave owns a 64x64 accumulator block.
// ---------------------------------------------------------------------------------------------------------
template <int AL>
struct ChunkDesc {      // one (segment, 32-channel chunk) of the K loop
  const void* ptr;      // segment base
  int bytes;            // segment size
  int cs2;              // bytes per pixel
  int wsoff;            // byte offset of this chunk inside a weight row's K axis
  int seg, c0;
  bool valid;
  unsigned vbase[AL];   // per-lane byte offset of (centre pixel, this lane's 16-B chunk), OOB_OFF when masked
};

// BHI (parity mode, data gradients: SEG_VARIANT_B_HI): the weight operand as its hi half - two MFMAs per product, and the lo chunks of
// the weight rows are not fetched (their lanes load out of range: zeros, no memory traffic).
template <int BM, int BN, int NT, int BKE, int NSTAGE, bool SPLIT = false, bool BHI = false>
__global__ __launch_bounds__(256, ((!SPLIT && BM == 128 && (BKE == 32 || BN == 64)) ? 3 : 2)) void igemm_fwd2_kernel(const seg_igemm_fwd_args p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource type only exists for the gfx950 pass; the host pass needs just the stub
  constexpr int WMv = BM / 64, WNv = 4 / WMv;     // waves along M / N; every wave owns 64 rows x (BN / WNv) columns
  constexpr int TNw = BN / (32 * WNv);            // 32-column accumulator tiles per wave (2, or 1 for the 128x64 tile)
  static_assert(WMv * WNv == 4 && TNw >= 1 && TNw * 32 * WNv == BN, "tile / wave layout");
  static_assert(BKE == 32 || BKE == 64, "K step of 32 or 64 channels");
  static_assert(!SPLIT || BKE == 32, "parity mode: one 64-B row = 16 real K values as hi / lo chunks");
  static_assert(SPLIT || !BHI, "the hi-only weight operand is a parity-mode form");
  typedef std::conditional_t<SPLIT, split_t, half_t> TO;
  constexpr int ROWB = BKE * 2;                  // bytes per LDS row (one K step of one pixel / one output channel)
  constexpr int CPR = ROWB / 16;                 // 16-B chunks per row
  constexpr int RPI = 1024 / ROWB;               // rows per LDS-DMA wave instruction (64 lanes x 16 B)
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int AL = BM / (4 * RPI), BL = BN / (4 * RPI);     // LDS-DMA instructions per wave and stage
  constexpr int LPI = AL + BL;                   // vector-memory operations per wave and K step
  constexpr int DIST = NSTAGE - 1;               // K steps of loads in flight
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNv, wn = wave % WNv;
  const seg_geom& g = p.g;
  const int HW = g.Hl * g.Wl;
  const int M = g.N * HW;
  const int mblocks = (M + BM - 1) / BM;
  const int lin = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int cls = lin / gridDim.x, bx = lin - cls * gridDim.x;
  const int mb = bx % mblocks, nb = bx / mblocks;
  const int m0 = mb * BM, n0 = nb * BN;

  int t_delta[NT], t_w[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int e = cls * NT + t;
    t_delta[t] = g.tap_dy[e] * g.Win + g.tap_dx[e];
    t_w[t] = g.tap_w[e] * p.Kpad * 2;
  }

  // ---- per-lane rows of the A tile: wave w, instruction i covers rows (w*AL + i)*16 .. +15, lane -> (row, chunk)
  const int lrow = lane / CPR, lch = lane % CPR;
  auto swz = [](int row) { return CPR == 4 ? (row >> 2) & 3 : (row >> 1) & 7; };
  int a_pix[AL], a_chs[AL];
  unsigned a_mask[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int row = (wave * AL + i) * RPI + lrow;
    const int m = m0 + row;
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    const int n = mm / HW;
    const int rem = mm - n * HW;
    const int ii = rem / g.Wl;
    const int jj = rem - ii * g.Wl;
    const int cy = ii * g.isy, cx = jj * g.isx;
    a_pix[i] = (n * g.Hin + cy) * g.Win + cx;
    a_chs[i] = lch ^ swz(row);
    unsigned msk = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int e = cls * NT + t;
      const int y = cy + g.tap_dy[e], x = cx + g.tap_dx[e];
      if (ok && (unsigned)y < (unsigned)g.Hin && (unsigned)x < (unsigned)g.Win) msk |= 1u << t;
    }
    a_mask[i] = msk;
  }
  unsigned b_voff[BL];
#pragma unroll
  for (int i = 0; i < BL; ++i) {
    const int row = (wave * BL + i) * RPI + lrow;
    const int nrow = n0 + row;
    const int chs = lch ^ swz(row);
    b_voff[i] = (nrow < p.n_rows && !(BHI && (chs & 1))) ? (unsigned)((nrow * g.T_total * p.Kpad + chs * 8) * 2) : OOB_OFF;
  }
  const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.n_rows * g.T_total * p.Kpad * 2, 0x00020000);

  f16v acc[2][TNw];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TNw; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
  f16v accl[SPLIT ? 2 : 1][SPLIT ? TNw : 1];        // parity mode: the (lo x hi + hi x lo) sums, scaled by 2048
  if constexpr (SPLIT) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TNw; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) accl[a][b][q] = 0.f;
  }

  int nchunks = 0;
  for (int s = 0; s < p.nseg; ++s) nchunks += (p.seg_c[s] + BKE - 1) / BKE;
  // split of the reduction: slice blockIdx.z of p.splitk contracts the chunks [ck0, ck1) and leaves its accumulators in p.partial
  const int ksl = p.splitk > 1 ? p.splitk : 1;
  const int kz = ksl > 1 ? (int)blockIdx.z : 0;
  const int kper = (nchunks + ksl - 1) / ksl;
  const int ck0 = min(kz * kper, nchunks), ck1 = min(nchunks, ck0 + kper);
  const int nloc = ck1 - ck0;
  const int niter = nloc * NT;
  const int pixbytes = g.N * g.Hin * g.Win;

  auto setup = [&](ChunkDesc<AL>& d, int seg, int c0, int cidx) {
    d.seg = seg;
    d.c0 = c0;
    d.valid = seg < p.nseg && cidx < ck1;
    if (d.valid) {
      const int cs = p.seg_cs[seg], sc = p.seg_c[seg];
      d.ptr = p.seg_ptr[seg];
      d.cs2 = cs * 2;
      d.bytes = pixbytes * d.cs2;
      d.wsoff = (p.seg_woff[seg] + c0) * 2;
#pragma unroll
      for (int i = 0; i < AL; ++i) {
        const int cc = c0 + a_chs[i] * 8;
        d.vbase[i] = cc < sc ? (unsigned)(a_pix[i] * d.cs2 + cc * 2) : OOB_OFF;
      }
    }
  };
  auto advance = [&](int& seg, int& c0) {
    c0 += BKE;
    if (seg < p.nseg && c0 >= p.seg_c[seg]) { c0 = 0; ++seg; }
  };
  auto issue = [&](auto tc, const ChunkDesc<AL>& d, int stage) {
    constexpr int t = decltype(tc)::value;
    if (!d.valid) return;
    unsigned char* sb = smem + stage * STAGE;
    // make the descriptor provably wave-uniform (otherwise hipcc wraps every buffer op in a waterfall loop)
    const unsigned long long pa = (unsigned long long)d.ptr;
    const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pa), phi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
    void* ubase = (void*)(((unsigned long long)phi << 32) | plo);
    const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(ubase, 0, __builtin_amdgcn_readfirstlane(d.bytes), 0x00020000);
    const int sdelta = t_delta[t] * __builtin_amdgcn_readfirstlane(d.cs2);
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const unsigned voff = ((a_mask[i] >> t) & 1u) ? d.vbase[i] + (unsigned)sdelta : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sb + (wave * AL + i) * 1024), 16, voff, 0, 0, 0);   // RPI rows
    }
    const int soff = t_w[t] + __builtin_amdgcn_readfirstlane(d.wsoff);
#pragma unroll
    for (int i = 0; i < BL; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(sb + A_BYTES + (wave * BL + i) * 1024), 16, b_voff[i], soff, 0, 0);
  };

  ChunkDesc<AL> D[3];
  int cseg = 0, cc0 = 0;
  for (int i = 0; i < ck0; ++i) advance(cseg, cc0);
  setup(D[0], cseg, cc0, ck0);
  advance(cseg, cc0);
  setup(D[1], cseg, cc0, ck0 + 1);
  advance(cseg, cc0);
  setup(D[2], cseg, cc0, ck0 + 2);

  // prologue: the first DIST K steps
  issue(std::integral_constant<int, 0>{}, D[0], 0);
  if constexpr (DIST == 2) issue(std::integral_constant<int, 1 % NT>{}, D[1 / NT], 1);

  const int fr = lane & 31, fh = lane >> 5;
  int it = 0, stage = 0;
  for (int ck = 0; ck < nloc; ++ck) {
    static_for<0, NT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int t2 = (t + DIST) % NT, k2 = (t + DIST) / NT;
      if (DIST == 2 && it + 1 < niter) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // the stage the loads issued below go into is the one this wave read LAST step (two-stage and three-stage ring alike:
      // the prefetch distance is the ring depth minus one)
      ring_release_barrier();
      int st2 = stage + DIST;
      if (st2 >= NSTAGE) st2 -= NSTAGE;
      issue(std::integral_constant<int, t2>{}, D[k2], st2);
      const unsigned char* A = smem + stage * STAGE;
      if constexpr (SPLIT) mma_step_split<2, TNw, BHI>(A, A + A_BYTES, wm * 64, wn * (32 * TNw), fr, fh, acc, accl);
      else if constexpr (BKE == 32) mma_step<2, TNw>(A, A + A_BYTES, wm * 64, wn * (32 * TNw), fr, fh, acc);
      else mma_step64<2, TNw>(A, A + A_BYTES, wm * 64, wn * (32 * TNw), fr, fh, acc);
      stage = stage == NSTAGE - 1 ? 0 : stage + 1;
      ++it;
    });
    D[0] = D[1];
    D[1] = D[2];
    advance(cseg, cc0);
    setup(D[2], cseg, cc0, ck0 + ck + 3);
  }
  __syncthreads();
  if constexpr (SPLIT) split_combine(acc, accl);
  if (ksl > 1) {          // this slice's accumulator image: [slice][class][tile][2 x TNw x 16 values][256 threads], coalesced
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float* img = p.partial + ((size_t)(kz * gridDim.y + cls) * gridDim.x + bx) * (size_t)(256 * 2 * TNw * 16) + tid;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TNw; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) img[((a * TNw + b) * 16 + q) * 256] = acc[a][b][q];
    return;
  }
  conv_epilogue<TO, 2, TNw, WMv, BN>(p, acc, smem, tid, m0 + wm * 64, wn * (32 * TNw), n0, wm, cls, mblocks, mb, M, HW);
#endif
}

// second launch of a split reduction: one block per output tile adds the slices' accumulator images in order (deterministic) and
// runs the epilogue of the kernel above - bias, activation, BN partial statistics, spans, accumulate, fused activation backward
template <int BM, int BN, bool SPLIT>
__global__ __launch_bounds__(256) void igemm_fwd2_splitk_epilogue_kernel(const seg_igemm_fwd_args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WMv = BM / 64, WNv = 4 / WMv, TNw = BN / (32 * WNv);
  typedef std::conditional_t<SPLIT, split_t, half_t> TO;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNv, wn = wave % WNv;
  const seg_geom& g = p.g;
  const int HW = g.Hl * g.Wl;
  const int M = g.N * HW;
  const int mblocks = (M + BM - 1) / BM;
  const int lin = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int cls = lin / gridDim.x, bx = lin - cls * gridDim.x;
  const int mb = bx % mblocks, nb = bx / mblocks;
  const int m0 = mb * BM, n0 = nb * BN;
  f16v acc[2][TNw];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TNw; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
  for (int kz = 0; kz < p.splitk; ++kz) {
    const float* img = p.partial + ((size_t)(kz * gridDim.y + cls) * gridDim.x + bx) * (size_t)(256 * 2 * TNw * 16) + tid;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TNw; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[a][b][q] += img[((a * TNw + b) * 16 + q) * 256];
  }
  conv_epilogue<TO, 2, TNw, WMv, BN>(p, acc, smem, tid, m0 + wm * 64, wn * (32 * TNw), n0, wm, cls, mblocks, mb, M, HW);
#endif
}
template <bool SPLIT>
inline void launch_fwd2_splitk_epilogue(const seg_igemm_fwd_args& a, dim3 grid, hipStream_t st) {
  const size_t lds = STG_BYTES + 4 * 128 * 2 * sizeof(float);
  dim3 g2(grid.x, grid.y, 1);
  if (a.tile_m == 128 && a.tile_n == 128) hipLaunchKernelGGL((igemm_fwd2_splitk_epilogue_kernel<128, 128, SPLIT>), g2, dim3(256), lds, st, a);
  else if (a.tile_m == 128) hipLaunchKernelGGL((igemm_fwd2_splitk_epilogue_kernel<128, 64, SPLIT>), g2, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((igemm_fwd2_splitk_epilogue_kernel<256, 64, SPLIT>), g2, dim3(256), lds, st, a);
}


// argument checks shared by seg_igemm_fwd and seg_igemm_fwd_f32 (returns SEG_OK or sets the error)
inline int check_fwd_args(const seg_igemm_fwd_args* a) {
  SEG_CHECK_ARG(a != nullptr, "seg_igemm_fwd: null args");
  if (!check_geom(a->g, "seg_igemm_fwd") || !check_segs(*a, "seg_igemm_fwd")) return SEG_ERR_ARG;
  SEG_CHECK_ARG(a->w != nullptr && a->n_rows > 0 && (a->n_rows & 7) == 0, "seg_igemm_fwd: null weight pointer / n_rows=%d", a->n_rows);
  if (a->nout == 0) {
    SEG_CHECK_ARG(a->out && (a->out_cs & 7) == 0 && (a->out_coff & 7) == 0 && a->out_coff + a->n_rows <= a->out_cs,
                  "seg_igemm_fwd: n_rows=%d out_cs=%d out_coff=%d", a->n_rows, a->out_cs, a->out_coff);
  } else {
    SEG_CHECK_ARG(a->nout > 0 && a->nout <= SEG_MAX_SEG && a->stats == nullptr, "seg_igemm_fwd: nout=%d (no statistics with a split output)", a->nout);
    int tot = 0;
    for (int o = 0; o < a->nout; ++o) {
      SEG_CHECK_ARG(a->out_span[o] > 0 && (a->out_span[o] & 31) == 0 && (a->out_cs_o[o] & 7) == 0 && a->out_cs_o[o] <= a->out_span[o],
                    "seg_igemm_fwd: output span %d: span=%d cs=%d", o, a->out_span[o], a->out_cs_o[o]);
      tot += a->out_span[o];
    }
    SEG_CHECK_ARG(tot >= a->n_rows, "seg_igemm_fwd: output spans cover %d of %d columns", tot, a->n_rows);
    for (int o = 0; o < a->nout; ++o)
      SEG_CHECK_ARG(a->out_y[o] == nullptr || (a->out_ptr[o] != nullptr && a->out_dact[o] >= SEG_ACT_RELU && a->out_dact[o] <= SEG_ACT_SIGMOID),
                    "seg_igemm_fwd: span %d: fused activation backward needs an output tensor and act in RELU..SIGMOID (got %d)", o, a->out_dact[o]);
  }
  SEG_CHECK_ARG((a->Kpad & 31) == 0, "seg_igemm_fwd: Kpad=%d not a multiple of 32", a->Kpad);
  SEG_CHECK_ARG((a->variant & 255) == 0 || (a->variant & 255) == 3, "seg_igemm_fwd: variant=%d", a->variant);
  SEG_CHECK_ARG(a->act >= SEG_ACT_NONE && a->act <= SEG_ACT_SIGMOID, "seg_igemm_fwd: act=%d", a->act);
  SEG_CHECK_ARG(a->splitk <= 1 || ((a->variant & 255) == 0 && a->partial != nullptr && a->splitk <= 64),
                "seg_igemm_fwd: splitk=%d needs variant 0 and the partial workspace (seg_igemm_fwd_workspace_bytes)", a->splitk);
  if ((a->variant & 255) != 3)
    SEG_CHECK_ARG((a->tile_m == 256 && a->tile_n == 64) || ((a->tile_m == 128 || a->tile_m == 64) && (a->tile_n == 128 || a->tile_n == 64)),
                  "seg_igemm_fwd: tile %dx%d", a->tile_m, a->tile_n);
  return SEG_OK;
}
// the per-tap LDS-DMA kernel takes 128- and 256-row tiles, tap counts it is instantiated for, operands below 1.875 GiB (seg_lds_dma_operand_limit)
// (esz = bytes per element of the operands: 2, parity mode 4)
inline bool fwd2_eligible(const seg_igemm_fwd_args& a, int esz) {
  if (!((a.tile_m == 128 && (a.tile_n == 128 || a.tile_n == 64)) || (a.tile_m == 256 && a.tile_n == 64))) return false;
  const int nt = a.g.ntaps;
  if (!(nt == 1 || nt == 2 || nt == 3 || nt == 4 || nt == 9 || nt == 16)) return false;
  const long lim = LDS_DMA_LIMIT;
  const long pix = (long)a.g.N * a.g.Hin * a.g.Win, shift = max_tap_shift_elems(a.g);
  for (int s = 0; s < a.nseg; ++s)
    if (pix * a.seg_cs[s] * esz >= lim || shift * a.seg_cs[s] * esz >= LDS_DMA_MAX_SHIFT) return false;
  if ((long)a.n_rows * a.g.T_total * a.Kpad * esz >= lim) return false;
  return true;
}

}  // namespace
