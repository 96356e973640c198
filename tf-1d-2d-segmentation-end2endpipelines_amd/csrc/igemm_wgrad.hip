// Weight-gradient kernels of the implicit-GEMM convolutions for gfx950 (seg_igemm_wgrad): per-tap and multi-tap
// register-staged kernels with transposing LDS reads, and the LDS-DMA halo-tile kernels for 3x3 convs / 4x4 transposed convs.
#include "igemm_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// weight gradient:  D[n][k] = sum_pix dz[pix][n] * in[pix + tap][k]   (reduction dimension = pixels)
// Both operands are "reduction-major" in memory, so their MFMA fragments are read with the hardware
// transposing LDS read ds_read_b64_tr_b16 (VARIANT 0) or transposed while being stored to LDS (VARIANT 1).
// (The parity mode's kernels - operands stored as split pairs - are in igemm_wgrad_split.hip.)
// ---------------------------------------------------------------------------------------------------------
template <int BM, int BN, int VARIANT>
__global__ __launch_bounds__(256, 1) void igemm_wgrad_kernel(const seg_igemm_wgrad_args p) {
  typedef half_t T;
  constexpr bool SPLIT = false;       // (the parity mode's kernels are in igemm_wgrad_split.hip)
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int EPC = ET<T>::EPC;
  constexpr int PK = (VARIANT == 0) ? 32 : 4 * EPC;   // pixels per step (VARIANT 1: one 64-B LDS row in the transposed tile)
  // VARIANT 0 tiles: [32 pixels][BM halves] rows of BM*2 + 64 bytes; SPLIT: a hi tile followed by a lo tile of that shape
  constexpr int ZLD = BM * 2 + 64, ALD = BN * 2 + 64;
  constexpr int NPL = 1;
  constexpr int Z_BYTES = (VARIANT == 0) ? NPL * PK * ZLD : BM * ROW_BYTES;
  constexpr int A_BYTES = (VARIANT == 0) ? NPL * PK * ALD : BN * ROW_BYTES;
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
  const T* sp = (const T*)p.seg_ptr[seg];
  const T* dz = (const T*)p.dz;

  int per = (M + p.splitk - 1) / p.splitk;
  per = (per + PK - 1) / PK * PK;
  const int pbeg = bz * per;
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
      mma_step<TM, TN>(Z, A, wm * (BM / 2), wn * (BN / 2), fr, fh, acc);
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
            if (part != nullptr) part[off] = acc[a][b][q];          // deterministic split: reduced in order by wgrad_reduce_kernel
            else atomicAdd(dw + off, acc[a][b][q] * sc_mul);
          }
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient, multi-tap form for layers with <= 64 output channels (the full-resolution level, 36 % of the
// UNet++ MACs): one block owns a 64 (n) x BN (k) tile for NTB taps at once, so the dz tile is read once per
// NTB taps instead of once per tap and the NTB shifted input tiles hit the same L1/L2 lines.  fp16, transposed
// LDS reads, single LDS buffer + register prefetch of the next 32-pixel step.
// ---------------------------------------------------------------------------------------------------------
template <int BN, int NTB>
__global__ __launch_bounds__(256, 2) void igemm_wgrad_mt_kernel(const seg_igemm_wgrad_args p) {
  constexpr int BM = 64, PK = 32, TN = BN / 64;
  constexpr int ZLD = BM * 2 + 64, ALD = BN * 2 + 64;
  constexpr int Z_BYTES = PK * ZLD, A_BYTES = PK * ALD;
  constexpr int ACH = BN / 8, APT = (PK * ACH) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef half_t T;

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
  const int groups = g.ntaps / NTB;
  const int cls = by / groups, tgrp = by % groups;
  const int e0 = cls * g.ntaps + tgrp * NTB;
  int t_dy[NTB], t_dx[NTB], t_w[NTB];
#pragma unroll
  for (int t = 0; t < NTB; ++t) { t_dy[t] = g.tap_dy[e0 + t]; t_dx[t] = g.tap_dx[e0 + t]; t_w[t] = g.tap_w[e0 + t]; }
  const int oy = g.oy[cls], ox = g.ox[cls];
  const int r0 = rt * BM, k0 = kt * BN;
  const int sc = p.seg_c[seg], cs = p.seg_cs[seg];
  const T* sp = (const T*)p.seg_ptr[seg];
  const T* dz = (const T*)p.dz;

  int per = (M + p.splitk - 1) / p.splitk;
  per = (per + PK - 1) / PK * PK;
  const int pbeg = bz * per;
  const int pend = min(M, pbeg + per);
  if (pbeg >= pend) return;

  // ---- per-thread rows: one dz chunk (pixel tid / 8, chunk tid % 8), APT input chunks per tap
  const int z_p = tid >> 3, z_c = tid & 7;
  int zn, zi, zj;
  {
    int m = pbeg + z_p;
    zn = m / HW; int rem = m - zn * HW; zi = rem / g.Wl; zj = rem - zi * g.Wl;
  }
  int a_p[APT], a_c[APT], an[APT], ai[APT], aj[APT];
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

  f16v acc[NTB][TN];
#pragma unroll
  for (int t = 0; t < NTB; ++t)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][b][q] = 0.f;

  u4 rz, ra[NTB][APT];
  auto gload = [&](int pb) {
    {
      u4 v = {0u, 0u, 0u, 0u};
      const int m = pb + z_p, col = r0 + z_c * 8;
      if (m < pend && col < p.n_rows) {
        size_t off = (((size_t)zn * g.Hout + (zi * g.osy + oy)) * g.Wout + (zj * g.osx + ox)) * p.dz_cs + p.dz_coff + col;
        v = ld16(dz + off);
      }
      rz = v;
      advance(zn, zi, zj);
    }
#pragma unroll
    for (int r = 0; r < APT; ++r) {
      const int m = pb + a_p[r], col = k0 + a_c[r] * 8;
      const bool rok = m < pend && col < sc;
      const int cy = ai[r] * g.isy, cx = aj[r] * g.isx;
      const size_t base = ((size_t)an[r] * g.Hin) * g.Win;
#pragma unroll
      for (int t = 0; t < NTB; ++t) {
        u4 v = {0u, 0u, 0u, 0u};
        const int y = cy + t_dy[t], x = cx + t_dx[t];
        if (rok && (unsigned)y < (unsigned)g.Hin && (unsigned)x < (unsigned)g.Win) v = ld16(sp + (base + (size_t)y * g.Win + x) * cs + col);
        ra[t][r] = v;
      }
      advance(an[r], ai[r], aj[r]);
    }
  };
  unsigned char* Z = smem;
  unsigned char* A0 = smem + Z_BYTES;
  auto lstore = [&]() {
    st16(Z + z_p * ZLD + z_c * 16, rz);
#pragma unroll
    for (int t = 0; t < NTB; ++t)
#pragma unroll
      for (int r = 0; r < APT; ++r) st16(A0 + t * A_BYTES + a_p[r] * ALD + a_c[r] * 16, ra[t][r]);
  };
  const int tg = lane >> 4, tt = lane & 15, tq = tt >> 2, tp = tt & 3;
  typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
  typedef __attribute__((address_space(3))) s4v* lds_s4v_ptr;
  auto trfrag = [&](const unsigned char* ad, int ld) -> h8 {
    s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(ad));
    s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(ad + 4 * ld));
    h4 l4 = __builtin_bit_cast(h4, lo), h4v = __builtin_bit_cast(h4, hi);
    return h8{l4[0], l4[1], l4[2], l4[3], h4v[0], h4v[1], h4v[2], h4v[3]};
  };
  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int krow = ks * 16 + 8 * (tg >> 1) + tq;
      const h8 af = trfrag(Z + krow * ZLD + (wm * 32 + 16 * (tg & 1) + 4 * tp) * 2, ZLD);
#pragma unroll
      for (int t = 0; t < NTB; ++t)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const h8 bf = trfrag(A0 + t * A_BYTES + krow * ALD + (wn * (BN / 2) + b * 32 + 16 * (tg & 1) + 4 * tp) * 2, ALD);
          acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[t][b], 0, 0, 0);
        }
    }
  };

  const int niter = (pend - pbeg + PK - 1) / PK;
  gload(pbeg);
  for (int it = 0; it < niter; ++it) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (it + 1 < niter) gload(pbeg + (it + 1) * PK);
    compute();
  }

  float sc_mul = p.scale;
  if (p.scale_ptr != nullptr) sc_mul *= *p.scale_ptr;
  float* dw = p.dw;
  float* part = (p.partial != nullptr && p.splitk > 1) ? p.partial + (size_t)bz * ((size_t)p.n_rows * g.T_total * p.Kpad) : nullptr;
  const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int t = 0; t < NTB; ++t)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int kcol = k0 + wn * (BN / 2) + b * 32 + fr;
      if (kcol < sc) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = r0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh;
          if (row < p.n_rows) {
            size_t off = ((size_t)row * g.T_total + t_w[t]) * p.Kpad + p.seg_woff[seg] + kcol;
            if (part != nullptr) part[off] = acc[t][b][q];
            else atomicAdd(dw + off, acc[t][b][q] * sc_mul);
          }
        }
      }
    }
}

template __global__ void igemm_wgrad_mt_kernel<64, 9>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_mt_kernel<64, 4>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_mt_kernel<64, 3>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_mt_kernel<128, 4>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_mt_kernel<128, 3>(const seg_igemm_wgrad_args);

// ---------------------------------------------------------------------------------------------------------
// weight gradient of the 3x3 / stride-1 / 'same' convolutions (variant 2, fp16) - 95 % of the UNet-family MACs.
// One block owns 64 output channels x one 64-channel plane of one concat segment x all 9 taps, and walks a range of
// 64-pixel patches (R rows x C columns, C = 32 or 16).  Per patch the dz tile [64 px][64 co] and ONE input halo tile
// [(R+2)(C+2) px][64 ci] are brought in by LDS-DMA (2-stage ring, one barrier per patch); the nine shifted input
// operands are nine row offsets into that halo tile, read - like dz - with the transposing ds_read_b64_tr_b16.
// LDS rows are 128 B with the 32-B units XOR-swizzled by bit 1 of the row, which keeps the four rows x two units a
// 32-lane group reads on 64 distinct banks for every tap shift; with the shift a compile-time constant the swizzled
// lane offset is one of four precomputed registers and the rest of the address is the instruction's immediate.
// ---------------------------------------------------------------------------------------------------------
template <int C, bool TC>
__global__ __launch_bounds__(256, 2) void igemm_wgrad_halo_kernel(const seg_igemm_wgrad_args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int R = 64 / C, HC = C + 2, HROWS = (R + 2) * HC;
  constexpr int NI_IN = (HROWS + 7) / 8;                  // LDS-DMA wave instructions per halo tile (8 rows of 128 B each)
  constexpr int NI_W = (NI_IN + 3) / 4;                   // ... per wave
  constexpr int Z_BYTES = 64 * 128, IN_BYTES = NI_IN * 1024, STAGE = Z_BYTES + IN_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
  typedef __attribute__((address_space(3))) s4v* lds_s4v_ptr;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const seg_geom& g = p.g;
  const int H = g.Hin, W = g.Win;
  // timing experiments (tools/op_bench.py --dbg on the wgrad phase; results are WRONG): 1 no LDS-DMA, 4 no barrier, 16 no epilogue
  const int dbg = __builtin_amdgcn_readfirstlane(p.variant >> 8);

  // ---- which (64-row tile, 64-channel plane, patch range) is this block
  // logical order: patch range (z) slowest, then sub-pixel class (y), then (row tile, plane): every block that reads
  // the same pixels - other planes / row tiles re-read dz, the four classes of a transposed conv re-read the input
  // halo - is handed to the same XCD and meets the data in its L2
  const int rtiles = (p.n_rows + 63) / 64;
  const int lin = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
  const int bz = lin / (gridDim.x * gridDim.y);
  const int lrem = lin - bz * (gridDim.x * gridDim.y);
  const int bcls = lrem / gridDim.x, bx = lrem - bcls * gridDim.x;
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
  // taps: a 3x3 conv uses all nine halo shifts t = 3 (dy + 1) + dx + 1 in geometry order; a sub-pixel class of the
  // 4x4 / stride-2 transposed conv (TC) uses four of them against the stride-2 pixels (oy, ox) of dz
  const int cls = TC ? bcls : 0;
  unsigned tmask = TC ? 0u : 0x1ffu;
  int wmap[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wmap[t] = TC ? 0 : g.tap_w[t];
  if constexpr (TC) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = cls * 4 + i;
      const int t = (g.tap_dy[e] + 1) * 3 + g.tap_dx[e] + 1;
      tmask |= 1u << t;
#pragma unroll
      for (int u = 0; u < 9; ++u)
        if (u == t) wmap[u] = g.tap_w[e];
    }
  }
  const int Ho = g.Hout, Wo = g.Wout, osy = g.osy, osx = g.osx;

  const i4v rsrc_in = make_rsrc(p.seg_ptr[seg], g.N * H * W * cs2);
  const i4v rsrc_dz = make_rsrc(p.dz, g.N * Ho * Wo * dzcs2);
  const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);

  // ---- LDS-DMA lane constants.  A wave instruction fills 8 LDS rows (lane >> 3) x 8 chunks (lane & 7); the chunk a
  // lane fetches from memory is its LDS chunk XOR the row's swizzle
  const int lrow = lane >> 3, lch = lane & 7;
  unsigned z_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 8 + lrow;
    const int gch = lch ^ (((row >> 1) & 1) << 2);
    const int yy = row / C, xx = row % C;
    z_voff[i] = (r0 + gch * 8 < p.n_rows) ? (unsigned)((yy * osy * Wo + xx * osx) * dzcs2 + (p.dz_coff + r0 + gch * 8) * 2) : OOB_OFF;
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
  // patch coordinates of the next issue, advanced incrementally (two scalar divisions per patch were 10 % of the
  // wave's issue cycles)
  int pn_ = __builtin_amdgcn_readfirstlane(q0 / (PH * PW));
  int py_ = __builtin_amdgcn_readfirstlane((q0 - pn_ * (PH * PW)) / PW);
  int px_ = __builtin_amdgcn_readfirstlane(q0 - pn_ * (PH * PW) - py_ * PW);
  auto issue = [&](int /*q*/, int stage) {
    if (dbg & 1) return;
    const int n = pn_, y0 = py_ * R, x0 = px_ * C;
    if (++px_ == PW) {
      px_ = 0;
      if (++py_ == PH) { py_ = 0; ++pn_; }
    }
    const unsigned sb = lds_base + (unsigned)(stage * STAGE);
    const int zs = __builtin_amdgcn_readfirstlane(((n * Ho + y0 * osy + g.oy[cls]) * Wo + x0 * osx + g.ox[cls]) * dzcs2);
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
  // 16 (tg & 1) + 4 tp of the wave's 32-channel slice.  Row = base + j: bit 1 of the row (the swizzle) depends on
  // (base & 3) + j only, so four lane offsets cover every base.
  const int tg = lane >> 4, tt = lane & 15, tq = tt >> 2, tp = tt & 3;
  const int j = 8 * (tg >> 1) + tq;
  const int cbA = (wm * 32 + 16 * (tg & 1) + 4 * tp) * 2, cbB = (wn * 32 + 16 * (tg & 1) + 4 * tp) * 2;
  const int offA = j * 128 + (cbA ^ (((j >> 1) & 1) << 6));
  // 8-pixel-wide patches (the 8x8 level): a 16-pixel K block is two image rows, pixels 8..15 sit one halo row further
  const int jr = C >= 16 ? j : j + (j >> 3) * (HC - 8);
  int offB[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) offB[r] = jr * 128 + (cbB ^ ((((r + jr) >> 1) & 1) << 6));
  auto trfrag = [&](const unsigned char* ad) -> h8 {
    s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(ad));
    s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(ad + 4 * 128));
    h4 l4 = __builtin_bit_cast(h4, lo), h4v = __builtin_bit_cast(h4, hi);
    return h8{l4[0], l4[1], l4[2], l4[3], h4v[0], h4v[1], h4v[2], h4v[3]};
  };

  f16v acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

  auto compute = [&](int stage) {
    const unsigned char* Zs = smem + stage * STAGE;
    const unsigned char* Is = Zs + Z_BYTES;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      constexpr int KPR = C >= 16 ? C / 16 : 1;                    // K blocks per patch row (C = 8: one K block = two rows)
      const int yy = C >= 16 ? kb / KPR : 2 * kb, xx0 = C >= 16 ? (kb % KPR) * 16 : 0;
      const h8 af = trfrag(Zs + kb * 16 * 128 + offA);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (TC && !((tmask >> t) & 1u)) continue;                   // wave-uniform
        const int hb = (yy + t / 3) * HC + xx0 + t % 3;            // halo row of pixel 0 of this K block under tap t
        const h8 bf = trfrag(Is + hb * 128 + offB[hb & 3]);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[t], 0, 0, 0);
      }
    }
  };

  int stage = 0;
  issue(q0, 0);
  for (int q = q0; q < q1; ++q) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ring_release_barrier(!(dbg & 4));
    if (q + 1 < q1) issue(q + 1, stage ^ 1);
    compute(stage);
    stage ^= 1;
  }

  float sc_mul = p.scale;
  if (p.scale_ptr != nullptr) sc_mul *= *p.scale_ptr;
  float* dw = p.dw;
  float* part = (p.partial != nullptr && p.splitk > 1) ? p.partial + (size_t)bz * ((size_t)p.n_rows * g.T_total * p.Kpad) : nullptr;
  const int fr = lane & 31, fh = lane >> 5;
  const int kcol = c0 + wn * 32 + fr;
  if (dbg & 16) {
    float keep = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) keep += acc[t][0] + acc[t][15];
    if (keep != 12345.678f) return;
  }
  if (kcol < sc) {
    const int stride = g.T_total * p.Kpad, row0 = r0 + wm * 32;
    float* const wbase = (part != nullptr ? part : dw) + (size_t)(row0 + 4 * fh) * stride + p.seg_woff[seg] + kcol;
    wgrad_mode_dispatch(wgrad_mode(p), [&](auto mc) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (TC && !((tmask >> t) & 1u)) continue;
        wgrad_store_tile<decltype(mc)::value>(wbase + wmap[t] * p.Kpad, acc[t], p.n_rows - row0, stride, sc_mul);
      }
    });
  }
#endif
}
// ---------------------------------------------------------------------------------------------------------
// the same for Conv2DTranspose(4x4, stride 2), two sub-pixel classes per block: class (py, 0) and (py, 1) read
// horizontally adjacent dz pixels and 2 x 2 of the same six halo shifts (dy in {0,-1} for py = 0, {1,0} for py = 1;
// dx in {0,-1} for px = 0, {1,0} for px = 1 - the tap table of model.geom_convT, checked by the launcher).  One input
// halo tile + two dz tiles per 64-pixel patch, 8 accumulator tiles per wave, 32 MFMAs per barrier instead of 16:
// the per-class form is bound by its fixed cost per patch (LDS-DMA pieces, barrier), not by MFMAs.
// ---------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256, 2) void igemm_wgrad_halo_tc2_kernel(const seg_igemm_wgrad_args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int R = 64 / C, HC = C + 2, HROWS = (R + 2) * HC;
  constexpr int NI_IN = (HROWS + 7) / 8, NI_W = (NI_IN + 3) / 4;
  constexpr int Z_BYTES = 64 * 128, IN_BYTES = NI_IN * 1024, STAGE = 2 * Z_BYTES + IN_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
  typedef __attribute__((address_space(3))) s4v* lds_s4v_ptr;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const seg_geom& g = p.g;
  const int H = g.Hin, W = g.Win, Ho = g.Hout, Wo = g.Wout;

  const int rtiles = (p.n_rows + 63) / 64;
  const int lin = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
  const int bz = lin / (gridDim.x * gridDim.y);
  const int lrem = lin - bz * (gridDim.x * gridDim.y);
  const int py = lrem / gridDim.x, bx = lrem - py * gridDim.x;       // class row
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
  if (q0 >= q1) {                      // an empty patch range still owns its row of the bias partials
    if (p.bias_part != nullptr && seg == 0 && c0 == 0 && tid < 64 && r0 + tid < p.n_rows) {
      float* rowp = p.bias_part + ((size_t)(py * p.splitk + bz) * 2) * p.n_rows;
      rowp[r0 + tid] = 0.f;
      rowp[p.n_rows + r0 + tid] = 0.f;
    }
    return;
  }

  const i4v rsrc_in = make_rsrc(p.seg_ptr[seg], g.N * H * W * cs2);
  const i4v rsrc_dz = make_rsrc(p.dz, g.N * Ho * Wo * dzcs2);
  const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);

  const int lrow = lane >> 3, lch = lane & 7;
  unsigned z_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 8 + lrow;
    const int gch = lch ^ (((row >> 1) & 1) << 2);
    const int yy = row / C, xx = row % C;
    z_voff[i] = (r0 + gch * 8 < p.n_rows) ? (unsigned)((yy * 2 * Wo + xx * 2) * dzcs2 + (p.dz_coff + r0 + gch * 8) * 2) : OOB_OFF;
  }
  int in_dy[NI_W], in_dx[NI_W], in_cb[NI_W];
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
    const int n = pn_, y0 = py_ * R, x0 = px_ * C;
    if (++px_ == PW) {
      px_ = 0;
      if (++py_ == PH) { py_ = 0; ++pn_; }
    }
    const unsigned sb = lds_base + (unsigned)(stage * STAGE);
    const int zs = __builtin_amdgcn_readfirstlane(((n * Ho + y0 * 2 + py) * Wo + x0 * 2) * dzcs2);          // class (py, 0); (py, 1) is the next pixel
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int i = 0; i < 2; ++i) lds_dma16(rsrc_dz, sb + c2 * Z_BYTES + (wave * 2 + i) * 1024, z_voff[i], zs + c2 * dzcs2);
#pragma unroll
    for (int k = 0; k < NI_W; ++k) {
      if (wave + 4 * k < NI_IN) {
        const int y = y0 + in_dy[k], x = x0 + in_dx[k];
        const bool ok = in_cb[k] >= 0 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const unsigned voff = ok ? (unsigned)(((n * H + y) * W + x) * cs2 + in_cb[k]) : OOB_OFF;
        lds_dma16(rsrc_in, sb + 2 * Z_BYTES + (wave + 4 * k) * 1024, voff, 0);
      }
    }
  };

  const int tg = lane >> 4, tt = lane & 15, tq = tt >> 2, tp = tt & 3;
  const int j = 8 * (tg >> 1) + tq;
  const int cbA = (wm * 32 + 16 * (tg & 1) + 4 * tp) * 2, cbB = (wn * 32 + 16 * (tg & 1) + 4 * tp) * 2;
  const int offA = j * 128 + (cbA ^ (((j >> 1) & 1) << 6));
  // 8-pixel-wide patches (the 8x8 level): a 16-pixel K block is two image rows, pixels 8..15 sit one halo row further
  const int jr = C >= 16 ? j : j + (j >> 3) * (HC - 8);
  int offB[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) offB[r] = jr * 128 + (cbB ^ ((((r + jr) >> 1) & 1) << 6));
  auto trfrag = [&](const unsigned char* ad) -> h8 {
    s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(ad));
    s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(ad + 4 * 128));
    h4 l4 = __builtin_bit_cast(h4, lo), h4v = __builtin_bit_cast(h4, hi);
    return h8{l4[0], l4[1], l4[2], l4[3], h4v[0], h4v[1], h4v[2], h4v[3]};
  };

  f16v acc[8];        // [class px][iy][ix]
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  // bias gradient = column sums of dz: the blocks of the first input plane contract their dz tiles with a ones operand
  // (one more MFMA per class and K block) instead of a separate pass over the 134 MB gradient tensor
  const bool do_bias = p.bias_part != nullptr && seg == 0 && c0 == 0 && wn == 0;
  f16v accb;
#pragma unroll
  for (int q = 0; q < 16; ++q) accb[q] = 0.f;
  const h8 ones = h8{(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};

  auto compute = [&](auto pyc, int stage) {
    constexpr int PY = decltype(pyc)::value;
    const unsigned char* Zs = smem + stage * STAGE;
    const unsigned char* Is = Zs + 2 * Z_BYTES;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      constexpr int KPR = C >= 16 ? C / 16 : 1;
      const int yy = C >= 16 ? kb / KPR : 2 * kb, xx0 = C >= 16 ? (kb % KPR) * 16 : 0;
      const h8 af0 = trfrag(Zs + kb * 16 * 128 + offA), af1 = trfrag(Zs + Z_BYTES + kb * 16 * 128 + offA);
      if (do_bias) {                                                 // wave-uniform
        accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(af0, ones, accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(af1, ones, accb, 0, 0, 0);
      }
#pragma unroll
      for (int iy = 0; iy < 2; ++iy) {
        const int dy = PY == 0 ? (iy == 0 ? 0 : -1) : (iy == 0 ? 1 : 0);
        const int hb0 = (yy + 1 + dy) * HC + xx0 + 1;               // halo row of pixel 0 of this K block at dx = 0
        const h8 bm = trfrag(Is + (hb0 - 1) * 128 + offB[(hb0 - 1) & 3]);
        const h8 bz_ = trfrag(Is + hb0 * 128 + offB[hb0 & 3]);
        const h8 bp = trfrag(Is + (hb0 + 1) * 128 + offB[(hb0 + 1) & 3]);
        acc[iy * 2 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af0, bz_, acc[iy * 2 + 0], 0, 0, 0);       // px = 0: dx = 0, -1
        acc[iy * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af0, bm, acc[iy * 2 + 1], 0, 0, 0);
        acc[4 + iy * 2 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af1, bp, acc[4 + iy * 2 + 0], 0, 0, 0);  // px = 1: dx = 1, 0
        acc[4 + iy * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af1, bz_, acc[4 + iy * 2 + 1], 0, 0, 0);
      }
    }
  };
  auto run = [&](auto pyc) {
    int stage = 0;
    issue(0);
    for (int q = q0; q < q1; ++q) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ring_release_barrier();
      if (q + 1 < q1) issue(stage ^ 1);
      compute(pyc, stage);
      stage ^= 1;
    }
  };
  if (py == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});

  float sc_mul = p.scale;
  if (p.scale_ptr != nullptr) sc_mul *= *p.scale_ptr;
  float* dw = p.dw;
  float* part = (p.partial != nullptr && p.splitk > 1) ? p.partial + (size_t)bz * ((size_t)p.n_rows * g.T_total * p.Kpad) : nullptr;
  const int fr = lane & 31, fh = lane >> 5;
  if (do_bias && fr == 0) {          // every column of accb holds the same sums: lanes of column 0 write the partial row
    float* rowp = p.bias_part + ((size_t)(py * p.splitk + bz) * 2) * p.n_rows;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = r0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh;
      if (row < p.n_rows) {
        rowp[row] = accb[q];
        rowp[p.n_rows + row] = 0.f;
      }
    }
  }
  const int kcol = c0 + wn * 32 + fr;
  if (kcol < sc) {
    const int stride = g.T_total * p.Kpad, row0 = r0 + wm * 32;
    float* const wbase = (part != nullptr ? part : dw) + (size_t)(row0 + 4 * fh) * stride + p.seg_woff[seg] + kcol;
    wgrad_mode_dispatch(wgrad_mode(p), [&](auto mc) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
        wgrad_store_tile<decltype(mc)::value>(wbase + g.tap_w[(py * 2 + (t >> 2)) * 4 + (t & 3)] * p.Kpad, acc[t], p.n_rows - row0, stride, sc_mul);
    });
  }
#endif
}
template __global__ void igemm_wgrad_halo_tc2_kernel<32>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_tc2_kernel<16>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_tc2_kernel<8>(const seg_igemm_wgrad_args);

template __global__ void igemm_wgrad_halo_kernel<32, false>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_kernel<16, false>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_kernel<32, true>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_kernel<16, true>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_kernel<8, false>(const seg_igemm_wgrad_args);
template __global__ void igemm_wgrad_halo_kernel<8, true>(const seg_igemm_wgrad_args);

// second pass of the deterministic pixel split: dw[i] += scale * sum_z partial[z][i], summed in a FIXED order: G lanes
// share a float4 position (lane g adds z = g, g + G, ... in ascending order), then a butterfly over the G lanes - the same
// tree on every run.  G > 1 keeps layers with few weights and many splits (first layer: 18 K weights x 256 splits) parallel.
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* partial, int splitk, long numel, float* dw, float scale, const float* scale_ptr) {
  if (scale_ptr != nullptr) scale *= *scale_ptr;
  const long n4 = numel >> 2;
  const int g = threadIdx.x % G;
  const long stride = (long)gridDim.x * (256 / G);
  const long iters = (n4 + stride - 1) / stride;                  // every lane runs the same trip count (the shuffles need all G lanes)
  long i = (long)blockIdx.x * (256 / G) + threadIdx.x / G;
  for (long it = 0; it < iters; ++it, i += stride) {
    const bool ok = i < n4;
    f4 s = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      for (int z = g; z < splitk; z += G) {
        const f4 v = *reinterpret_cast<const f4*>(partial + (size_t)z * numel + 4 * i);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
      }
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) {
      s[0] += __shfl_xor(s[0], o); s[1] += __shfl_xor(s[1], o); s[2] += __shfl_xor(s[2], o); s[3] += __shfl_xor(s[3], o);
    }
    if (ok && g == 0) {
      f4 d = *reinterpret_cast<const f4*>(dw + 4 * i);
      d[0] += s[0] * scale; d[1] += s[1] * scale; d[2] += s[2] * scale; d[3] += s[3] * scale;
      *reinterpret_cast<f4*>(dw + 4 * i) = d;
    }
  }
}

}  // namespace
int seg_launch_wgrad_reduce(const seg_igemm_wgrad_args& a, hipStream_t st) {
  if (a.partial == nullptr || a.splitk <= 1) return SEG_OK;
  const long numel = (long)a.n_rows * a.g.T_total * a.Kpad;          // Kpad is a multiple of 32: whole float4 groups
  if (a.splitk >= 32) {
    long blocks = (numel / 4 + 31) / 32;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((wgrad_reduce_kernel<8>), dim3((unsigned)blocks), dim3(256), 0, st, a.partial, a.splitk, numel, a.dw, a.scale, a.scale_ptr);
  } else {
    long blocks = (numel / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL((wgrad_reduce_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, a.partial, a.splitk, numel, a.dw, a.scale, a.scale_ptr);
  }
  SEG_CHECK_LAUNCH("seg_igemm_wgrad (split reduce)");
  return SEG_OK;
}

// largest operand (bytes, exclusive) of the LDS-DMA kernel families - halo tiles (seg_igemm_fwd variant 3, seg_igemm_wgrad variant 2) and
// the per-tap LDS-DMA kernel; beyond it seg_igemm_fwd variant 0 takes its register-staged kernel, seg_igemm_wgrad variants 0 / 1 theirs
extern "C" int64_t seg_lds_dma_operand_limit(void) { return (int64_t)LDS_DMA_LIMIT; }

// ---- workspace queries (no allocation inside the library - the caller asks, allocates, passes the pointer)
// floats of the per-block partial BN statistics seg_igemm_fwd / seg_igemm_fwd_f32 write when args->stats != NULL:
// [ncls * ceil(M / tile_m)][2][n_rows]
extern "C" int64_t seg_igemm_fwd_stats_floats(const seg_igemm_fwd_args* a) {
  if (a == nullptr || a->tile_m <= 0) return -1;
  const int64_t M = (int64_t)a->g.N * a->g.Hl * a->g.Wl;
  return (int64_t)a->g.ncls * ((M + a->tile_m - 1) / a->tile_m) * 2 * a->n_rows;
}
// bytes of the zero-initialised `partial` workspace of the deterministic pixel split of seg_igemm_wgrad / _f32 (0: not needed)
extern "C" int64_t seg_igemm_wgrad_workspace_bytes(const seg_igemm_wgrad_args* a) {
  if (a == nullptr) return -1;
  return a->splitk > 1 ? (int64_t)a->splitk * a->n_rows * a->g.T_total * a->Kpad * (int64_t)sizeof(float) : 0;
}

// taps handled per block by the weight-gradient kernels (1 = the per-tap kernel); exported so that the host can
// size the pixel split with the same rule
extern "C" int seg_wgrad_taps_per_block(int tile_m, int tile_n, int ntaps, int variant) {
  if (variant != 0 || tile_m != 64) return 1;
  if (tile_n == 64) return (ntaps == 9 || ntaps == 4 || ntaps == 3) ? ntaps : 1;
  if (tile_n == 128) return ntaps == 4 ? 4 : (ntaps == 9 || ntaps == 3) ? 3 : 1;
  return 1;
}

// patch width (32 or 16 pixels; the patch is 64 / width rows high) of the halo-tile weight-gradient kernel for this
// geometry, 0 when the geometry is not a 3x3 / stride-1 / 'same' convolution it can tile
extern "C" int seg_wgrad_halo_patch(const seg_geom* g) {
  if (g == nullptr || g->isy != 1 || g->isx != 1 || g->Hl != g->Hin || g->Wl != g->Win) return 0;
  if (g->ncls == 1 && g->ntaps == 9) {
    if (g->osy != 1 || g->osx != 1 || g->Hout != g->Hin || g->Wout != g->Win || g->oy[0] != 0 || g->ox[0] != 0) return 0;
    for (int t = 0; t < 9; ++t)
      if (g->tap_dy[t] != t / 3 - 1 || g->tap_dx[t] != t % 3 - 1) return 0;
  } else if (g->ncls == 4 && g->ntaps == 4) {          // 4x4 / stride-2 transposed conv: 4 distinct halo shifts per sub-pixel class
    if (g->osy != 2 || g->osx != 2 || g->Hout != 2 * g->Hin || g->Wout != 2 * g->Win) return 0;
    for (int c = 0; c < 4; ++c) {
      if (g->oy[c] < 0 || g->oy[c] > 1 || g->ox[c] < 0 || g->ox[c] > 1) return 0;
      unsigned seen = 0;
      for (int t = 0; t < 4; ++t) {
        const int dy = g->tap_dy[c * 4 + t], dx = g->tap_dx[c * 4 + t];
        if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return 0;
        const unsigned bit = 1u << ((dy + 1) * 3 + dx + 1);
        if (seen & bit) return 0;
        seen |= bit;
      }
    }
  } else {
    return 0;
  }
  // 4 x 16 patches first: their halo tile has 108 rows against 136 for 2 x 32 (4-6 % faster, measured), 8 x 8 ties
  if (g->Win % 16 == 0 && g->Hin % 4 == 0) return 16;
  if (g->Win % 32 == 0 && g->Hin % 2 == 0) return 32;
  if (g->Win % 8 == 0 && g->Hin % 8 == 0) return 8;
  return 0;
}

// blocks along the class axis of the variant-2 kernel: 1 for a conv, 2 when the four classes of a transposed conv
// follow the tap table of the class-pair kernel (classes py*2+px; taps iy*2+ix with dy = {0,-1} / {1,0} for py = 0 / 1
// and dx likewise for px), 4 otherwise.  The host sizes splitk with it.
extern "C" int seg_wgrad_halo_class_groups(const seg_geom* g) {
  if (g == nullptr || g->ncls != 4 || g->ntaps != 4) return g != nullptr && g->ncls == 1 ? 1 : (g != nullptr ? g->ncls : 0);
  for (int c = 0; c < 4; ++c) {
    const int py = c >> 1, px = c & 1;
    if (g->oy[c] != py || g->ox[c] != px) return 4;
    for (int t = 0; t < 4; ++t) {
      const int iy = t >> 1, ix = t & 1;
      const int dy = py == 0 ? (iy == 0 ? 0 : -1) : (iy == 0 ? 1 : 0), dx = px == 0 ? (ix == 0 ? 0 : -1) : (ix == 0 ? 1 : 0);
      if (g->tap_dy[c * 4 + t] != dy || g->tap_dx[c * 4 + t] != dx) return 4;
    }
  }
  return 2;
}

extern "C" int seg_igemm_wgrad(const seg_igemm_wgrad_args* a, seg_stream_t stream) {
  typedef half_t T;
  if (const int rc = check_wgrad_args(a, "seg_igemm_wgrad")) return rc;
  SEG_CHECK_ARG(a->bias_part == nullptr || ((a->variant & 255) == 2 && seg_wgrad_halo_class_groups(&a->g) == 2),
                "seg_igemm_wgrad: bias_part needs variant 2 on a transposed conv of the class-pair kernel");
  if ((a->variant & 255) == 2) {      // (bits 8.. : timing-experiment switches of the halo kernel)
    const int pc = seg_wgrad_halo_patch(&a->g);
    SEG_CHECK_ARG(pc != 0, "seg_igemm_wgrad: variant 2 needs a 3x3 / stride-1 / 'same' conv or a 4x4 / stride-2 transposed conv, with W %% 32 == 0 and H %% 2 == 0, W %% 16 == 0 and H %% 4 == 0, or W and H %% 8 == 0");
    const long pix = (long)a->g.N * a->g.Hin * a->g.Win;
    SEG_CHECK_ARG((long)a->g.N * a->g.Hout * a->g.Wout * a->dz_cs * 2 < LDS_DMA_LIMIT, "seg_igemm_wgrad: variant 2 needs operands below 1.875 GiB (seg_lds_dma_operand_limit)");
    int planes = 0;
    for (int s = 0; s < a->nseg; ++s) {
      SEG_CHECK_ARG(pix * a->seg_cs[s] * 2 < LDS_DMA_LIMIT, "seg_igemm_wgrad: variant 2 needs operands below 1.875 GiB (seg_lds_dma_operand_limit)");
      planes += (a->seg_c[s] + 63) / 64;
    }
    if constexpr (sizeof(T) == 2) {
      dim3 gridh((unsigned)(((a->n_rows + 63) / 64) * planes), (unsigned)a->g.ncls, (unsigned)a->splitk);
      hipStream_t sth = (hipStream_t)stream;
      const int ni = pc == 32 ? 17 : pc == 16 ? 14 : 13;              // LDS-DMA pieces of the halo tile
      const size_t lds1 = (size_t)2 * (8192 + ni * 1024), lds2 = (size_t)2 * (2 * 8192 + ni * 1024);
#define SEG_HALO_PC(KERNEL, GRID, LDS, ...)                                                                     \
  do {                                                                                                          \
    if (pc == 32) hipLaunchKernelGGL((KERNEL<32 __VA_ARGS__>), GRID, dim3(256), LDS, sth, *a);                   \
    else if (pc == 16) hipLaunchKernelGGL((KERNEL<16 __VA_ARGS__>), GRID, dim3(256), LDS, sth, *a);              \
    else hipLaunchKernelGGL((KERNEL<8 __VA_ARGS__>), GRID, dim3(256), LDS, sth, *a);                             \
  } while (0)
      // wave-specialised generation (igemm_wgrad2.hip: 4 loader waves + 4 compute waves, 4-stage ring, one block per CU).
      // Measured (tools/op_bench.py, C3 layers, same box): when the grid has at most ~1 block per CU anyway (deep layers:
      // few patches, many weight tiles) it is 25-45 % faster than the round-1 kernel (X23 488 -> 352 us, X32 474 -> 339 us,
      // enc3 92 -> 73 us); when the round-1 kernel can keep two blocks per CU it is 0-12 % slower (X14, X05).
      const long nblk = (long)gridh.x * gridh.y * gridh.z;
      const bool ws = nblk <= 320;
      // larger grids: tap-split generation (igemm_wgrad3.hip: an input fragment feeds two MFMAs, -30 % LDS reads)
      const bool ts = true;
      const int force = a->variant >> 8;                            // tests: 0x40 wave-specialised, 0x80 round 1, 0x100 tap-split
      if (a->g.ncls == 1 && (force == 0x40 || (ws && force != 0x80 && !(force & 0x100)))) {
        seg_launch_wgrad_halo_ws(*a, pc, gridh, sth);
      } else if (a->g.ncls == 1 && ((force & 0x100) || (ts && force != 0x80))) {
        seg_launch_wgrad_halo_ts(*a, pc, gridh, sth);
      } else if (a->g.ncls == 1) {
        SEG_HALO_PC(igemm_wgrad_halo_kernel, gridh, lds1, , false);
      } else if (seg_wgrad_halo_class_groups(&a->g) == 2) {
        dim3 grid2(gridh.x, 2, gridh.z);
        SEG_HALO_PC(igemm_wgrad_halo_tc2_kernel, grid2, lds2);
      } else {
        SEG_HALO_PC(igemm_wgrad_halo_kernel, gridh, lds1, , true);
      }
#undef SEG_HALO_PC
    }
    SEG_CHECK_LAUNCH("seg_igemm_wgrad (halo)");
    return seg_launch_wgrad_reduce(*a, (hipStream_t)stream);
  }
  const int variant = a->variant & 255;
  const int rtiles = (a->n_rows + a->tile_m - 1) / a->tile_m;
  int ktiles = 0;
  for (int s = 0; s < a->nseg; ++s) ktiles += (a->seg_c[s] + a->tile_n - 1) / a->tile_n;
  if constexpr (sizeof(T) == 2) {
    const int ntb = seg_wgrad_taps_per_block(a->tile_m, a->tile_n, a->g.ntaps, a->variant);
    if (ntb > 1) {
      dim3 gridm((unsigned)(rtiles * ktiles), (unsigned)(a->g.ncls * (a->g.ntaps / ntb)), (unsigned)a->splitk);
      const size_t ldsm = (size_t)32 * (64 * 2 + 64) + (size_t)ntb * 32 * (a->tile_n * 2 + 64);
      hipStream_t stm = (hipStream_t)stream;
      if (a->tile_n == 64 && ntb == 9) hipLaunchKernelGGL((igemm_wgrad_mt_kernel<64, 9>), gridm, dim3(256), ldsm, stm, *a);
      else if (a->tile_n == 64 && ntb == 4) hipLaunchKernelGGL((igemm_wgrad_mt_kernel<64, 4>), gridm, dim3(256), ldsm, stm, *a);
      else if (a->tile_n == 64 && ntb == 3) hipLaunchKernelGGL((igemm_wgrad_mt_kernel<64, 3>), gridm, dim3(256), ldsm, stm, *a);
      else if (a->tile_n == 128 && ntb == 4) hipLaunchKernelGGL((igemm_wgrad_mt_kernel<128, 4>), gridm, dim3(256), ldsm, stm, *a);
      else hipLaunchKernelGGL((igemm_wgrad_mt_kernel<128, 3>), gridm, dim3(256), ldsm, stm, *a);
      SEG_CHECK_LAUNCH("seg_igemm_wgrad (multi-tap)");
      return seg_launch_wgrad_reduce(*a, (hipStream_t)stream);
    }
  }
  dim3 grid((unsigned)(rtiles * ktiles), (unsigned)(a->g.ncls * a->g.ntaps), (unsigned)a->splitk), block(256, 1, 1);
  hipStream_t st = (hipStream_t)stream;
  const int bm = a->tile_m, bn = a->tile_n;
  size_t lds = (variant == 0) ? (size_t)2 * 32 * ((bm * 2 + 64) + (bn * 2 + 64)) : (size_t)2 * (bm + bn) * ROW_BYTES;
#define SEG_WG(BM_, BN_)                                                                                  \
  do {                                                                                                    \
    if (variant == 0) hipLaunchKernelGGL((igemm_wgrad_kernel<BM_, BN_, 0>), grid, block, lds, st, *a);     \
    else hipLaunchKernelGGL((igemm_wgrad_kernel<BM_, BN_, 1>), grid, block, lds, st, *a);                  \
  } while (0)
  if (bm == 128 && bn == 128) SEG_WG(128, 128);
  else if (bm == 128 && bn == 64) SEG_WG(128, 64);
  else if (bm == 64 && bn == 128) SEG_WG(64, 128);
  else SEG_WG(64, 64);
#undef SEG_WG
  SEG_CHECK_LAUNCH("seg_igemm_wgrad");
  return seg_launch_wgrad_reduce(*a, st);
}
