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
#include "seg_common.h"

namespace {

constexpr int ROW_BYTES = 64;     // one LDS tile row = one K step = 4 chunks of 16 B (32 halves or 16 floats)
template <typename T> struct ET;
template <> struct ET<half_t> { static constexpr int EPC = 8; };   // elements per 16-B chunk
template <> struct ET<float> { static constexpr int EPC = 4; };

// One K step (64 B per row) of MFMAs for a (TM x TN) grid of 32x32 tiles.  A/B point at the rows of this wave's
// tiles; lane (fr = lane & 31, fh = lane >> 5) reads row fr.  fp16: two v_mfma_f32_32x32x16_f16 (k = 8 fh + j);
// fp32: eight v_mfma_f32_32x32x2_f32 - lane half fh owns k = 8 fh .. 8 fh + 7 of the 16-deep step and MFMA j
// contracts the pair {j, 8 + j} (the sum over k is order-free as long as A and B use the same assignment).
template <typename T, int TM, int TN>
__device__ __forceinline__ void mma_step(const unsigned char* A, const unsigned char* B, int a_row0, int b_row0, int fr, int fh,
                                         f16v (&acc)[TM][TN]) {
  const int fswz = (fr >> 2) & 3;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ((ks * 2 + fh) ^ fswz) << 4;
      h8 af[TM], bf[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const h8*>(A + (a_row0 + a * 32 + fr) * ROW_BYTES + ch);
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const h8*>(B + (b_row0 + b * 32 + fr) * ROW_BYTES + ch);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  } else {
    const int c0 = ((2 * fh) ^ fswz) << 4, c1 = ((2 * fh + 1) ^ fswz) << 4;
    f4 a0[TM], a1[TM], b0[TN], b1[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const unsigned char* r = A + (a_row0 + a * 32 + fr) * ROW_BYTES;
      a0[a] = *reinterpret_cast<const f4*>(r + c0);
      a1[a] = *reinterpret_cast<const f4*>(r + c1);
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const unsigned char* r = B + (b_row0 + b * 32 + fr) * ROW_BYTES;
      b0[b] = *reinterpret_cast<const f4*>(r + c0);
      b1[b] = *reinterpret_cast<const f4*>(r + c1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[a][j], b0[b][j], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[a][j], b1[b][j], acc[a][b], 0, 0, 0);
        }
  }
}
constexpr int STG_LD = 36;        // floats per staging row (32 + 4 pad, keeps 16-B alignment)
constexpr int STG_BYTES = 4 * 32 * STG_LD * 4;   // 4 waves

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ ((row >> 2) & 3)) << 4); }

// ---------------------------------------------------------------------------------------------------------
// forward / dgrad
// ---------------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void igemm_fwd_kernel(const seg_igemm_fwd_args p) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int EPC = ET<T>::EPC, BK = 4 * EPC;      // elements per 16-B chunk / per K step
  constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int cls = blockIdx.y;
  const seg_geom& g = p.g;
  const int HW = g.Hl * g.Wl;
  const int M = g.N * HW;
  const int mblocks = (M + BM - 1) / BM;
  const int mb = blockIdx.x % mblocks, nb = blockIdx.x / mblocks;
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

  int niter = 0;
  for (int s = 0; s < p.nseg; ++s) niter += ((p.seg_c[s] + BK - 1) / BK) * g.ntaps;

  u4 ra[TM], rb[TN];
  const T* wbase = (const T*)p.w;

  auto gload = [&](int seg, int c0, int tap) {
    const T* sp = (const T*)p.seg_ptr[seg];
    const int cs = p.seg_cs[seg];
    const int sc = p.seg_c[seg];
    const int e = cls * g.ntaps + tap;
    const int dy = g.tap_dy[e], dx = g.tap_dx[e], widx = g.tap_w[e];
    const int cc = c0 + ld_ch * EPC;
    const bool cok = cc < sc;
#pragma unroll
    for (int r = 0; r < TM; ++r) {
      int y = a_y[r] + dy, x = a_x[r] + dx;
      bool ok = cok && a_ok[r] && (unsigned)y < (unsigned)g.Hin && (unsigned)x < (unsigned)g.Win;
      size_t off = ((size_t)(a_pix[r] + y * g.Win + x)) * cs + cc;
      u4 v = {0u, 0u, 0u, 0u};
      if (ok) v = ld16(sp + off);
      ra[r] = v;
    }
    const T* wp = wbase + (size_t)widx * p.Kpad + p.seg_woff[seg] + cc;
#pragma unroll
    for (int r = 0; r < TN; ++r) {
      u4 v = {0u, 0u, 0u, 0u};
      if (b_ok[r]) v = ld16(wp + b_rowoff[r]);
      rb[r] = v;
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* A = smem + buf * (A_BYTES + B_BYTES);
    unsigned char* B = A + A_BYTES;
#pragma unroll
    for (int r = 0; r < TM; ++r) st16(A + lds_off(ld_row + 64 * r, ld_ch), ra[r]);
#pragma unroll
    for (int r = 0; r < TN; ++r) st16(B + lds_off(ld_row + 64 * r, ld_ch), rb[r]);
  };
  const int fr = lane & 31, fh = lane >> 5;
  auto compute = [&](int buf) {
    const unsigned char* A = smem + buf * (A_BYTES + B_BYTES);
    mma_step<T, TM, TN>(A, A + A_BYTES, wm * (BM / 2), wn * (BN / 2), fr, fh, acc);
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

  // ---- epilogue
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * STG_LD);
  float* sst = reinterpret_cast<float*>(smem + STG_BYTES);   // [2 (wm)][BN][2]
  const float* bias = p.bias;
  T* out = (T*)p.out;
  const bool do_stats = p.stats != nullptr;
  float s_sum[TN], s_sq[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) { s_sum[b] = 0.f; s_sq[b] = 0.f; }

  // output pixel offsets of the rows this lane stores: row = (lane >> 2) + 16 * pass within each 32-row tile
  const int st_row = lane >> 2, st_ch = lane & 3;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    long opix[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      int m = m0 + wm * (BM / 2) + a * 32 + st_row + 16 * ps;
      if (m < M) {
        int n = m / HW;
        int rem = m - n * HW;
        int i = rem / g.Wl;
        int j = rem - i * g.Wl;
        opix[ps] = ((long)n * g.Hout + (i * g.osy + g.oy[cls])) * g.Wout + (j * g.osx + g.ox[cls]);
      } else {
        opix[ps] = -1;
      }
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int nbase = n0 + wn * (BN / 2) + b * 32;
      const int ncol = nbase + fr;
      float bv = 0.f;
      if (bias != nullptr && ncol < p.n_rows) bv = bias[ncol];
      const bool nzero = ncol >= p.n_valid;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rt = (q & 3) + 8 * (q >> 2) + 4 * fh;
        float v = acc[a][b][q] + bv;
        if (do_stats) {
          const int m = m0 + wm * (BM / 2) + a * 32 + rt;
          if (m < M) { s_sum[b] += v; s_sq[b] += v * v; }
        }
        v = seg_act(v, p.act, p.alpha);
        if (nzero) v = 0.f;
        stg[rt * STG_LD + fr] = v;
      }
      // read back: lane -> (row, 8 consecutive channels)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int rt = st_row + 16 * ps;
        const int nn = nbase + st_ch * 8;
        f4 v0 = *reinterpret_cast<const f4*>(&stg[rt * STG_LD + st_ch * 8]);
        f4 v1 = *reinterpret_cast<const f4*>(&stg[rt * STG_LD + st_ch * 8 + 4]);
        if (opix[ps] >= 0 && nn < p.n_rows) {
          T* dst = out + (size_t)opix[ps] * p.out_cs + p.out_coff + nn;
          float f[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          if (p.accumulate) {
            float o[8];
            load8(dst, o);
#pragma unroll
            for (int t = 0; t < 8; ++t) f[t] += o[t];
          }
          store8(dst, f);
        }
      }
    }
  }
  if (do_stats) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      float s = s_sum[b] + __shfl_xor(s_sum[b], 32);
      float q = s_sq[b] + __shfl_xor(s_sq[b], 32);
      if (fh == 0) {
        int col = wn * (BN / 2) + b * 32 + fr;
        sst[(wm * BN + col) * 2 + 0] = s;
        sst[(wm * BN + col) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < BN) {
      int ncol = n0 + tid;
      if (ncol < p.n_rows) {
        float s = sst[(0 * BN + tid) * 2 + 0] + sst[(1 * BN + tid) * 2 + 0];
        float q = sst[(0 * BN + tid) * 2 + 1] + sst[(1 * BN + tid) * 2 + 1];
        size_t part = (size_t)cls * mblocks + mb;
        p.stats[(part * 2 + 0) * p.n_rows + ncol] = s;
        p.stats[(part * 2 + 1) * p.n_rows + ncol] = q;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient:  D[n][k] = sum_pix dz[pix][n] * in[pix + tap][k]   (reduction dimension = pixels)
// Both operands are "reduction-major" in memory, so their MFMA fragments are read with the hardware
// transposing LDS read ds_read_b64_tr_b16 (VARIANT 0) or transposed while being stored to LDS (VARIANT 1).
// ---------------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int VARIANT>
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(const seg_igemm_wgrad_args p) {
  static_assert(sizeof(T) == 2 || VARIANT == 1, "the transposing LDS read exists for 16-bit elements only");
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int EPC = ET<T>::EPC;
  constexpr int PK = 4 * EPC;                      // pixels per step (one 64-B LDS row in the transposed tile)
  // VARIANT 0 tiles: [32 pixels][BM halves] rows of BM*2 + 64 bytes
  constexpr int ZLD = BM * 2 + 64, ALD = BN * 2 + 64;
  constexpr int Z_BYTES = (VARIANT == 0) ? PK * ZLD : BM * ROW_BYTES;
  constexpr int A_BYTES = (VARIANT == 0) ? PK * ALD : BN * ROW_BYTES;
  constexpr int ZCH = BM / EPC, ACH = BN / EPC;    // 16-B chunks per pixel row
  constexpr int ZPT = (PK * ZCH) / 256, APT = (PK * ACH) / 256;   // chunks per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const seg_geom& g = p.g;
  const int HW = g.Hl * g.Wl;
  const int M = g.N * HW;

  // ---- which (row tile, segment, K tile, tap entry, pixel split) is this block
  const int rtiles = (p.n_rows + BM - 1) / BM;
  const int rt = blockIdx.x % rtiles;
  int kt = blockIdx.x / rtiles;
  int seg = 0;
  for (; seg < p.nseg; ++seg) {
    int t = (p.seg_c[seg] + BN - 1) / BN;
    if (kt < t) break;
    kt -= t;
  }
  if (seg >= p.nseg) return;
  const int e = blockIdx.y;                        // cls * ntaps + tap
  const int cls = e / g.ntaps;
  const int dy = g.tap_dy[e], dx = g.tap_dx[e], widx = g.tap_w[e];
  const int oy = g.oy[cls], ox = g.ox[cls];
  const int r0 = rt * BM, k0 = kt * BN;
  const int sc = p.seg_c[seg], cs = p.seg_cs[seg];
  const T* sp = (const T*)p.seg_ptr[seg];
  const T* dz = (const T*)p.dz;

  int per = (M + p.splitk - 1) / p.splitk;
  per = (per + PK - 1) / PK * PK;
  const int pbeg = blockIdx.z * per;
  const int pend = min(M, pbeg + per);
  if (pbeg >= pend) return;

  // ---- per-thread load bookkeeping: chunk q = tid + 256 r -> pixel row q / CH, chunk q % CH
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

  f16v acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

  u4 rz[ZPT], ra[APT];
  auto gload = [&](int pb) {
#pragma unroll
    for (int r = 0; r < ZPT; ++r) {
      u4 v = {0u, 0u, 0u, 0u};
      int m = pb + z_p[r];
      int col = r0 + z_c[r] * EPC;
      if (m < pend && col < p.n_rows) {
        size_t off = (((size_t)zn[r] * g.Hout + (zi[r] * g.osy + oy)) * g.Wout + (zj[r] * g.osx + ox)) * p.dz_cs + p.dz_coff + col;
        v = ld16(dz + off);
      }
      rz[r] = v;
      advance(zn[r], zi[r], zj[r]);
    }
#pragma unroll
    for (int r = 0; r < APT; ++r) {
      u4 v = {0u, 0u, 0u, 0u};
      int m = pb + a_p[r];
      int col = k0 + a_c[r] * EPC;
      int y = ai[r] * g.isy + dy, x = aj[r] * g.isx + dx;
      if (m < pend && col < sc && (unsigned)y < (unsigned)g.Hin && (unsigned)x < (unsigned)g.Win) {
        size_t off = (((size_t)an[r] * g.Hin + y) * g.Win + x) * cs + col;
        v = ld16(sp + off);
      }
      ra[r] = v;
      advance(an[r], ai[r], aj[r]);
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* Z = smem + buf * (Z_BYTES + A_BYTES);
    unsigned char* A = Z + Z_BYTES;
    if constexpr (VARIANT == 0) {
#pragma unroll
      for (int r = 0; r < ZPT; ++r) st16(Z + z_p[r] * ZLD + z_c[r] * 16, rz[r]);
#pragma unroll
      for (int r = 0; r < APT; ++r) st16(A + a_p[r] * ALD + a_c[r] * 16, ra[r]);
    } else {
      // transposing store: LDS tile [channel row][PK pixels], same swizzled 64-B row format as the fwd kernel
      typedef T tvec __attribute__((ext_vector_type(EPC)));
#pragma unroll
      for (int r = 0; r < ZPT; ++r) {
        tvec v = __builtin_bit_cast(tvec, rz[r]);
        int pp = z_p[r];
#pragma unroll
        for (int t = 0; t < EPC; ++t) {
          int row = z_c[r] * EPC + t;
          *reinterpret_cast<T*>(Z + lds_off(row, pp / EPC) + (pp % EPC) * (int)sizeof(T)) = v[t];
        }
      }
#pragma unroll
      for (int r = 0; r < APT; ++r) {
        tvec v = __builtin_bit_cast(tvec, ra[r]);
        int pp = a_p[r];
#pragma unroll
        for (int t = 0; t < EPC; ++t) {
          int row = a_c[r] * EPC + t;
          *reinterpret_cast<T*>(A + lds_off(row, pp / EPC) + (pp % EPC) * (int)sizeof(T)) = v[t];
        }
      }
    }
  };
  const int fr = lane & 31, fh = lane >> 5;
  // transposed-read lane decomposition: group g16 = lane >> 4, t = lane & 15, q = t >> 2, pp = t & 3
  const int tg = lane >> 4, tt = lane & 15, tq = tt >> 2, tp = tt & 3;
  auto compute = [&](int buf) {
    const unsigned char* Z = smem + buf * (Z_BYTES + A_BYTES);
    const unsigned char* A = Z + Z_BYTES;
    if constexpr (VARIANT == 0) {
      typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
      typedef __attribute__((address_space(3))) s4v* lds_s4v_ptr;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        h8 af[TM], bf[TN];
        const int krow = ks * 16 + 8 * (tg >> 1) + tq;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          int colh = wm * (BM / 2) + a * 32 + 16 * (tg & 1) + 4 * tp;
          const unsigned char* ad = Z + krow * ZLD + colh * 2;
          s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(ad));
          s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(ad + 4 * ZLD));
          h4 l4 = __builtin_bit_cast(h4, lo), h4v = __builtin_bit_cast(h4, hi);
          af[a] = h8{l4[0], l4[1], l4[2], l4[3], h4v[0], h4v[1], h4v[2], h4v[3]};
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          int colh = wn * (BN / 2) + b * 32 + 16 * (tg & 1) + 4 * tp;
          const unsigned char* ad = A + krow * ALD + colh * 2;
          s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(ad));
          s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(ad + 4 * ALD));
          h4 l4 = __builtin_bit_cast(h4, lo), h4v = __builtin_bit_cast(h4, hi);
          bf[b] = h8{l4[0], l4[1], l4[2], l4[3], h4v[0], h4v[1], h4v[2], h4v[3]};
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
      }
    } else {
      mma_step<T, TM, TN>(Z, A, wm * (BM / 2), wn * (BN / 2), fr, fh, acc);
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

  // ---- epilogue: fp32 atomics, lanes 0..31 hit 32 consecutive K columns of one weight row
  float sc_mul = p.scale;
  if (p.scale_ptr != nullptr) sc_mul *= *p.scale_ptr;
  float* dw = p.dw;
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
            atomicAdd(dw + off, acc[a][b][q] * sc_mul);
          }
        }
      }
    }
}

bool check_geom(const seg_geom& g, const char* who) {
  if (g.N <= 0 || g.Hin <= 0 || g.Win <= 0 || g.Hl <= 0 || g.Wl <= 0 || g.Hout <= 0 || g.Wout <= 0) {
    seg_set_error("%s: non-positive dimension", who);
    return false;
  }
  if (g.ncls < 1 || g.ncls > SEG_MAX_CLS || g.ntaps < 1 || g.ncls * g.ntaps > SEG_MAX_TAPS) {
    seg_set_error("%s: ncls=%d ntaps=%d out of range", who, g.ncls, g.ntaps);
    return false;
  }
  for (int c = 0; c < g.ncls; ++c) {
    long ymax = (long)(g.Hl - 1) * g.osy + g.oy[c], xmax = (long)(g.Wl - 1) * g.osx + g.ox[c];
    if (g.oy[c] < 0 || g.ox[c] < 0 || ymax >= g.Hout || xmax >= g.Wout) {
      seg_set_error("%s: class %d output coordinates leave the output tensor", who, c);
      return false;
    }
  }
  for (int e = 0; e < g.ncls * g.ntaps; ++e)
    if (g.tap_w[e] < 0 || g.tap_w[e] >= g.T_total) {
      seg_set_error("%s: tap %d weight index %d outside T_total=%d", who, e, g.tap_w[e], g.T_total);
      return false;
    }
  if ((long)g.N * g.Hl * g.Wl >= (1L << 31) || (long)g.N * g.Hin * g.Win >= (1L << 31)) {
    seg_set_error("%s: pixel count overflows int32", who);
    return false;
  }
  return true;
}

template <typename Args>
bool check_segs(const Args& a, const char* who) {
  if (a.nseg < 1 || a.nseg > SEG_MAX_SEG) { seg_set_error("%s: nseg=%d", who, a.nseg); return false; }
  for (int s = 0; s < a.nseg; ++s) {
    if (a.seg_ptr[s] == nullptr || a.seg_c[s] <= 0 || (a.seg_c[s] & 7) || (a.seg_cs[s] & 7) || a.seg_c[s] > a.seg_cs[s] ||
        (a.seg_woff[s] & 31) || a.seg_woff[s] + ((a.seg_c[s] + 31) & ~31) > a.Kpad) {
      seg_set_error("%s: bad segment %d (c=%d cs=%d woff=%d Kpad=%d)", who, s, a.seg_c[s], a.seg_cs[s], a.seg_woff[s], a.Kpad);
      return false;
    }
  }
  return true;
}

}  // namespace

template <typename T>
static int igemm_fwd_launch(const seg_igemm_fwd_args* a, seg_stream_t stream) {
  SEG_CHECK_ARG(a != nullptr, "seg_igemm_fwd: null args");
  if (!check_geom(a->g, "seg_igemm_fwd") || !check_segs(*a, "seg_igemm_fwd")) return SEG_ERR_ARG;
  SEG_CHECK_ARG(a->w && a->out, "seg_igemm_fwd: null weight/output pointer");
  SEG_CHECK_ARG(a->n_rows > 0 && (a->n_rows & 7) == 0 && (a->out_cs & 7) == 0 && (a->out_coff & 7) == 0 &&
                    a->out_coff + a->n_rows <= a->out_cs,
                "seg_igemm_fwd: n_rows=%d out_cs=%d out_coff=%d", a->n_rows, a->out_cs, a->out_coff);
  SEG_CHECK_ARG((a->Kpad & 31) == 0, "seg_igemm_fwd: Kpad=%d not a multiple of 32", a->Kpad);
  SEG_CHECK_ARG((a->tile_m == 128 || a->tile_m == 64) && (a->tile_n == 128 || a->tile_n == 64),
                "seg_igemm_fwd: tile %dx%d", a->tile_m, a->tile_n);
  SEG_CHECK_ARG(a->act >= SEG_ACT_NONE && a->act <= SEG_ACT_SIGMOID, "seg_igemm_fwd: act=%d", a->act);
  const long M = (long)a->g.N * a->g.Hl * a->g.Wl;
  const int mblocks = (int)((M + a->tile_m - 1) / a->tile_m);
  const int nblocks = (a->n_rows + a->tile_n - 1) / a->tile_n;
  dim3 grid((unsigned)(mblocks * nblocks), (unsigned)a->g.ncls, 1), block(256, 1, 1);
  const int lds_main = 2 * (a->tile_m + a->tile_n) * ROW_BYTES;
  const int lds_epi = STG_BYTES + 2 * a->tile_n * 2 * 4;
  const size_t lds = (size_t)(lds_main > lds_epi ? lds_main : lds_epi);
  hipStream_t st = (hipStream_t)stream;
  if (a->tile_m == 128 && a->tile_n == 128) hipLaunchKernelGGL((igemm_fwd_kernel<T, 128, 128>), grid, block, lds, st, *a);
  else if (a->tile_m == 128 && a->tile_n == 64) hipLaunchKernelGGL((igemm_fwd_kernel<T, 128, 64>), grid, block, lds, st, *a);
  else if (a->tile_m == 64 && a->tile_n == 128) hipLaunchKernelGGL((igemm_fwd_kernel<T, 64, 128>), grid, block, lds, st, *a);
  else hipLaunchKernelGGL((igemm_fwd_kernel<T, 64, 64>), grid, block, lds, st, *a);
  SEG_CHECK_LAUNCH("seg_igemm_fwd");
  return SEG_OK;
}
extern "C" int seg_igemm_fwd(const seg_igemm_fwd_args* a, seg_stream_t stream) { return igemm_fwd_launch<half_t>(a, stream); }
extern "C" int seg_igemm_fwd_f32(const seg_igemm_fwd_args* a, seg_stream_t stream) { return igemm_fwd_launch<float>(a, stream); }

template <typename T>
static int igemm_wgrad_launch(const seg_igemm_wgrad_args* a, seg_stream_t stream) {
  SEG_CHECK_ARG(a != nullptr, "seg_igemm_wgrad: null args");
  if (!check_geom(a->g, "seg_igemm_wgrad") || !check_segs(*a, "seg_igemm_wgrad")) return SEG_ERR_ARG;
  SEG_CHECK_ARG(a->dz && a->dw, "seg_igemm_wgrad: null pointer");
  SEG_CHECK_ARG(a->n_rows > 0 && (a->n_rows & 7) == 0 && (a->dz_cs & 7) == 0 && (a->dz_coff & 7) == 0 &&
                    a->dz_coff + a->n_rows <= a->dz_cs,
                "seg_igemm_wgrad: n_rows=%d dz_cs=%d dz_coff=%d", a->n_rows, a->dz_cs, a->dz_coff);
  SEG_CHECK_ARG((a->Kpad & 31) == 0, "seg_igemm_wgrad: Kpad=%d", a->Kpad);
  SEG_CHECK_ARG((a->tile_m == 128 || a->tile_m == 64) && (a->tile_n == 128 || a->tile_n == 64),
                "seg_igemm_wgrad: tile %dx%d", a->tile_m, a->tile_n);
  SEG_CHECK_ARG(a->splitk >= 1 && a->splitk <= 65535, "seg_igemm_wgrad: splitk=%d", a->splitk);
  SEG_CHECK_ARG(a->variant == 0 || a->variant == 1, "seg_igemm_wgrad: variant=%d", a->variant);
  const int variant = (sizeof(T) == 4) ? 1 : a->variant;      // fp32: transposing-store form only
  const int rtiles = (a->n_rows + a->tile_m - 1) / a->tile_m;
  int ktiles = 0;
  for (int s = 0; s < a->nseg; ++s) ktiles += (a->seg_c[s] + a->tile_n - 1) / a->tile_n;
  dim3 grid((unsigned)(rtiles * ktiles), (unsigned)(a->g.ncls * a->g.ntaps), (unsigned)a->splitk), block(256, 1, 1);
  hipStream_t st = (hipStream_t)stream;
  const int bm = a->tile_m, bn = a->tile_n;
  size_t lds = (variant == 0) ? (size_t)2 * 32 * ((bm * 2 + 64) + (bn * 2 + 64)) : (size_t)2 * (bm + bn) * ROW_BYTES;
#define SEG_WG(BM_, BN_)                                                                                          \
  do {                                                                                                            \
    if constexpr (sizeof(T) == 2) {                                                                               \
      if (variant == 0) hipLaunchKernelGGL((igemm_wgrad_kernel<T, BM_, BN_, 0>), grid, block, lds, st, *a);        \
      else hipLaunchKernelGGL((igemm_wgrad_kernel<T, BM_, BN_, 1>), grid, block, lds, st, *a);                     \
    } else {                                                                                                      \
      hipLaunchKernelGGL((igemm_wgrad_kernel<T, BM_, BN_, 1>), grid, block, lds, st, *a);                          \
    }                                                                                                             \
  } while (0)
  if (bm == 128 && bn == 128) SEG_WG(128, 128);
  else if (bm == 128 && bn == 64) SEG_WG(128, 64);
  else if (bm == 64 && bn == 128) SEG_WG(64, 128);
  else SEG_WG(64, 64);
#undef SEG_WG
  SEG_CHECK_LAUNCH("seg_igemm_wgrad");
  return SEG_OK;
}
extern "C" int seg_igemm_wgrad(const seg_igemm_wgrad_args* a, seg_stream_t stream) { return igemm_wgrad_launch<half_t>(a, stream); }
extern "C" int seg_igemm_wgrad_f32(const seg_igemm_wgrad_args* a, seg_stream_t stream) { return igemm_wgrad_launch<float>(a, stream); }
