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
#include <cstdlib>
#include <type_traits>

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
    // all eight fragment reads of the 32-deep step are issued before the first MFMA: the LDS latency is paid once
    h8 af[2][TM], bf[2][TN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ((ks * 2 + fh) ^ fswz) << 4;
#pragma unroll
      for (int a = 0; a < TM; ++a) af[ks][a] = *reinterpret_cast<const h8*>(A + (a_row0 + a * 32 + fr) * ROW_BYTES + ch);
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[ks][b] = *reinterpret_cast<const h8*>(B + (b_row0 + b * 32 + fr) * ROW_BYTES + ch);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][a], bf[ks][b], acc[a][b], 0, 0, 0);
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

// XCD-aware block order (MI355X: 8 XCDs with private L2s, workgroups dealt round-robin over them): physical
// block b runs on XCD b % 8; give each XCD one contiguous range of logical tiles so that neighbouring tiles (which
// share halo rows / the dz tile / weight panels) meet in the same L2.  Bijective for any block count.
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ ((row >> 2) & 3)) << 4); }

// 64-deep K step on 128-B LDS rows (fp16): 8 chunks per row, chunk index XOR (row >> 1) & 7, so that the 16 rows a
// ds_read_b128 lane group touches land on 16 distinct 16-B slots of the 256-B bank row.  4 x (TM x TN) MFMAs.
template <int TM, int TN>
__device__ __forceinline__ void mma_step64(const unsigned char* A, const unsigned char* B, int a_row0, int b_row0, int fr, int fh,
                                           f16v (&acc)[TM][TN]) {
  const int fswz = (fr >> 1) & 7;      // tile rows start at multiples of 32, so (row >> 1) & 7 depends on fr only
#pragma unroll
  for (int kp = 0; kp < 2; ++kp) {     // two pairs of 16-deep MFMA steps; fragment reads of a pair issued together
    h8 af[2][TM], bf[2][TN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = (((kp * 2 + ks) * 2 + fh) ^ fswz) << 4;
#pragma unroll
      for (int a = 0; a < TM; ++a) af[ks][a] = *reinterpret_cast<const h8*>(A + (a_row0 + a * 32 + fr) * 128 + ch);
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[ks][b] = *reinterpret_cast<const h8*>(B + (b_row0 + b * 32 + fr) * 128 + ch);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][a], bf[ks][b], acc[a][b], 0, 0, 0);
  }
}

// eight stored elements kept in their memory format (16 B of fp16 / 32 B of fp32) until they are needed
template <typename T> struct Raw8;
template <> struct Raw8<half_t> {
  typedef u4 type;
  static __device__ __forceinline__ void zero(type& r) { r = u4{0u, 0u, 0u, 0u}; }
  static __device__ __forceinline__ void load(const half_t* p, type& r) { r = ld16(p); }
  static __device__ __forceinline__ void add(const type& r, float* f) {
    float o[8];
    unpack8(r, o);
#pragma unroll
    for (int t = 0; t < 8; ++t) f[t] += o[t];
  }
};
template <> struct Raw8<float> {
  struct type { f4 a, b; };
  static __device__ __forceinline__ void zero(type& r) { r.a = f4{0.f, 0.f, 0.f, 0.f}; r.b = r.a; }
  static __device__ __forceinline__ void load(const float* p, type& r) {
    r.a = *reinterpret_cast<const f4*>(p);
    r.b = *reinterpret_cast<const f4*>(p + 4);
  }
  static __device__ __forceinline__ void add(const type& r, float* f) {
#pragma unroll
    for (int t = 0; t < 4; ++t) { f[t] += r.a[t]; f[4 + t] += r.b[t]; }
  }
};

// ---------------------------------------------------------------------------------------------------------
// shared epilogue of the forward kernels: bias, BN partial statistics, activation, channel-padding mask, then each
// 32x32 accumulator tile goes through a per-wave LDS slab so that global stores are 8 channels (16 / 32 B) per lane.
// row0 = first logical pixel of this wave's rows, col0 = first column of this wave inside the block's BN columns.
// ---------------------------------------------------------------------------------------------------------
// PWE = 0: this wave's rows are the logical pixels row0 .. ; PWE = 32 / 16: the rows are a 256-pixel patch of width
// PWE at (image pn, row py0, column px0) and row0 is the wave's first patch row (the halo-tile kernel).
template <typename T, int TM, int TN, int NWM, int BN, int PWE = 0>
__device__ __forceinline__ void conv_epilogue(const seg_igemm_fwd_args& p, f16v (&acc)[TM][TN], unsigned char* smem, int tid, int row0,
                                              int col0, int n0, int wm, int cls, int mblocks, int mb, int M, int HW,
                                              int pn = 0, int py0 = 0, int px0 = 0) {
  const seg_geom& g = p.g;
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * STG_LD);
  float* sst = reinterpret_cast<float*>(smem + STG_BYTES);   // [NWM][BN][2]
  const float* bias = p.bias;
  T* out = (T*)p.out;
  const bool do_stats = p.stats != nullptr;
  float s_sum[TN], s_sq[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) { s_sum[b] = 0.f; s_sq[b] = 0.f; }
  // rows this lane stores: (lane >> 2) + 16 * pass within each 32-row tile, 8 channels (lane & 3) * 8 ..
  const int st_row = lane >> 2, st_ch = lane & 3;
  // destination of each 32-column tile: the single output, or one span of the split output (wave-uniform)
  T* obase[TN];
  const T* ybase[TN];            // output of the layer whose activation derivative multiplies this span (or null)
  int ocs[TN], ocol[TN], olim[TN], oacc[TN], dact[TN];
  float dalpha[TN];
  bool any_acc = false, any_y = false;
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int nbase = n0 + col0 + b * 32;
    obase[b] = out;
    ybase[b] = nullptr;
    dact[b] = SEG_ACT_NONE;
    dalpha[b] = 0.f;
    ocs[b] = p.out_cs; ocol[b] = p.out_coff + nbase; olim[b] = p.n_rows - nbase; oacc[b] = p.accumulate;
    if (p.nout > 0) {
      int start = 0;
      obase[b] = nullptr;
      olim[b] = 0;
      for (int o = 0; o < p.nout; ++o) {
        if (nbase >= start && nbase < start + p.out_span[o]) {
          obase[b] = (T*)p.out_ptr[o];
          ocs[b] = p.out_cs_o[o];
          ocol[b] = nbase - start;
          olim[b] = p.out_cs_o[o] - ocol[b];
          oacc[b] = p.out_acc[o];
          ybase[b] = (const T*)p.out_y[o];
          dact[b] = p.out_dact[o];
          dalpha[b] = p.out_dalpha[o];
        }
        start += p.out_span[o];
      }
    }
    any_acc |= oacc[b] != 0;
    any_y |= ybase[b] != nullptr;
  }
  long opix[TM][2];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      if constexpr (PWE > 0) {
        const int r = st_row + 16 * ps;
        const int yy = row0 + (PWE == 32 ? a : 2 * a + (r >> 4)), xx = PWE == 32 ? r : (r & 15);
        opix[a][ps] = ((long)pn * g.Hout + ((py0 + yy) * g.osy + g.oy[cls])) * g.Wout + ((px0 + xx) * g.osx + g.ox[cls]);
      } else {
        int m = row0 + a * 32 + st_row + 16 * ps;
        if (m < M) {
          int n = m / HW;
          int rem = m - n * HW;
          int i = rem / g.Wl;
          int j = rem - i * g.Wl;
          opix[a][ps] = ((long)n * g.Hout + (i * g.osy + g.oy[cls])) * g.Wout + (j * g.osx + g.ox[cls]);
        } else {
          opix[a][ps] = -1;
        }
      }
    }
  // accumulating outputs (data gradients summed over the consumers of a tensor): fetch every old value up front, so
  // the read-modify-write costs one memory latency per block instead of one per 32x32 tile and row pass
  // (in groups of two 32-row tiles: 16 registers per tile row)
  constexpr int AG = TM > 2 ? 2 : TM;
  typename Raw8<T>::type oldv[AG][TN][2], yv[AG][TN][2];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    if ((any_acc || any_y) && a % AG == 0) {
#pragma unroll
      for (int a2 = 0; a2 < AG; ++a2)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            Raw8<T>::zero(oldv[a2][b][ps]);
            Raw8<T>::zero(yv[a2][b][ps]);
            if (opix[a + a2][ps] >= 0 && obase[b] != nullptr && st_ch * 8 < olim[b]) {
              const size_t eo = (size_t)opix[a + a2][ps] * ocs[b] + ocol[b] + st_ch * 8;
              if (oacc[b]) Raw8<T>::load(obase[b] + eo, oldv[a2][b][ps]);
              if (ybase[b] != nullptr) Raw8<T>::load(ybase[b] + eo, yv[a2][b][ps]);
            }
          }
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int nbase = n0 + col0 + b * 32;
      const int ncol = nbase + fr;
      float bv = 0.f;
      if (bias != nullptr && ncol < p.n_rows) bv = bias[ncol];
      const bool nzero = ncol >= p.n_valid;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rt = (q & 3) + 8 * (q >> 2) + 4 * fh;
        float v = acc[a][b][q] + bv;
        if (do_stats) {
          const int m = row0 + a * 32 + rt;
          if (PWE > 0 || m < M) { s_sum[b] += v; s_sq[b] += v * v; }
        }
        v = seg_act(v, p.act, p.alpha);
        if (nzero) v = 0.f;
        stg[rt * STG_LD + fr] = v;
      }
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int rt = st_row + 16 * ps;
        f4 v0 = *reinterpret_cast<const f4*>(&stg[rt * STG_LD + st_ch * 8]);
        f4 v1 = *reinterpret_cast<const f4*>(&stg[rt * STG_LD + st_ch * 8 + 4]);
        if (opix[a][ps] >= 0 && obase[b] != nullptr && st_ch * 8 < olim[b]) {
          T* dst = obase[b] + (size_t)opix[a][ps] * ocs[b] + ocol[b] + st_ch * 8;
          float f[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          if (any_y && ybase[b] != nullptr) {
            float yf[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) yf[t] = 0.f;
            Raw8<T>::add(yv[a % AG][b][ps], yf);
#pragma unroll
            for (int t = 0; t < 8; ++t) f[t] *= seg_act_grad_from_out(yf[t], dact[b], dalpha[b]);
          }
          if (any_acc) Raw8<T>::add(oldv[a % AG][b][ps], f);
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
        int col = col0 + b * 32 + fr;
        sst[(wm * BN + col) * 2 + 0] = s;
        sst[(wm * BN + col) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < BN) {
      int ncol = n0 + tid;
      if (ncol < p.n_rows) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < NWM; ++w) { s += sst[(w * BN + tid) * 2 + 0]; q += sst[(w * BN + tid) * 2 + 1]; }
        size_t part = (size_t)cls * mblocks + mb;
        p.stats[(part * 2 + 0) * p.n_rows + ncol] = s;
        p.stats[(part * 2 + 1) * p.n_rows + ncol] = q;
      }
    }
  }
}

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

  conv_epilogue<T, TM, TN, 2, BN>(p, acc, smem, tid, m0 + wm * (BM / 2), wn * (BN / 2), n0, wm, cls, mblocks, mb, M, HW);
}

// ---------------------------------------------------------------------------------------------------------
// forward / dgrad, v2 (fp16): LDS-DMA staging (buffer_load_dwordx4 ... lds) into a 3-stage LDS ring, counted
// vmcnt + raw s_barrier (one barrier per K step, two steps of loads always in flight), kernel taps unrolled at
// compile time so that per-step address work is one scalar multiply per operand and two VALU ops per LDS-DMA
// (tap mask select + add).  Out-of-image taps, rows past M and channel padding use the buffer range check:
// an offset beyond num_records makes the hardware write zeros.  Tiles: 128x128 (2x2 waves) and 256x64 (4x1
// waves), every wave owns a 64x64 accumulator block.
// ---------------------------------------------------------------------------------------------------------
constexpr unsigned OOB_OFF = 0x60000000u;   // beyond any legal byte offset (operands are checked to be < 1 GiB)

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

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

template <int BM, int BN, int NT, int BKE, int NSTAGE>
__global__ __launch_bounds__(256, ((BM == 128 && (BKE == 32 || BN == 64)) ? 3 : 2)) void igemm_fwd2_kernel(const seg_igemm_fwd_args p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource type only exists for the gfx950 pass; the host pass needs just the stub
  constexpr int WMv = BM / 64, WNv = 4 / WMv;     // waves along M / N; every wave owns 64 rows x (BN / WNv) columns
  constexpr int TNw = BN / (32 * WNv);            // 32-column accumulator tiles per wave (2, or 1 for the 128x64 tile)
  static_assert(WMv * WNv == 4 && TNw >= 1 && TNw * 32 * WNv == BN, "tile / wave layout");
  static_assert(BKE == 32 || BKE == 64, "K step of 32 or 64 channels");
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
    b_voff[i] = nrow < p.n_rows ? (unsigned)((nrow * g.T_total * p.Kpad + chs * 8) * 2) : OOB_OFF;
  }
  const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.n_rows * g.T_total * p.Kpad * 2, 0x00020000);

  f16v acc[2][TNw];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TNw; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

  int nchunks = 0;
  for (int s = 0; s < p.nseg; ++s) nchunks += (p.seg_c[s] + BKE - 1) / BKE;
  const int niter = nchunks * NT;
  const int pixbytes = g.N * g.Hin * g.Win;

  auto setup = [&](ChunkDesc<AL>& d, int seg, int c0) {
    d.seg = seg;
    d.c0 = c0;
    d.valid = seg < p.nseg;
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
  setup(D[0], cseg, cc0);
  advance(cseg, cc0);
  setup(D[1], cseg, cc0);
  advance(cseg, cc0);
  setup(D[2], cseg, cc0);

  // prologue: the first DIST K steps
  issue(std::integral_constant<int, 0>{}, D[0], 0);
  if constexpr (DIST == 2) issue(std::integral_constant<int, 1 % NT>{}, D[1 / NT], 1);

  const int fr = lane & 31, fh = lane >> 5;
  int it = 0, stage = 0;
  for (int ck = 0; ck < nchunks; ++ck) {
    static_for<0, NT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int t2 = (t + DIST) % NT, k2 = (t + DIST) / NT;
      if (DIST == 2 && it + 1 < niter) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int st2 = stage + DIST;
      if (st2 >= NSTAGE) st2 -= NSTAGE;
      issue(std::integral_constant<int, t2>{}, D[k2], st2);
      const unsigned char* A = smem + stage * STAGE;
      if constexpr (BKE == 32) mma_step<half_t, 2, TNw>(A, A + A_BYTES, wm * 64, wn * (32 * TNw), fr, fh, acc);
      else mma_step64<2, TNw>(A, A + A_BYTES, wm * 64, wn * (32 * TNw), fr, fh, acc);
      stage = stage == NSTAGE - 1 ? 0 : stage + 1;
      ++it;
    });
    D[0] = D[1];
    D[1] = D[2];
    advance(cseg, cc0);
    setup(D[2], cseg, cc0);
  }
  __syncthreads();
  conv_epilogue<half_t, 2, TNw, WMv, BN>(p, acc, smem, tid, m0 + wm * 64, wn * (32 * TNw), n0, wm, cls, mblocks, mb, M, HW);
#endif
}

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

// ---------------------------------------------------------------------------------------------------------
// forward / data gradient of the 3x3 / stride-1 / 'same' convolutions, v3 (fp16): halo tiles.  A block owns a
// 256-pixel patch (PH rows x PW columns) x BN output channels.  Per 32-channel chunk of the (virtual) concat ONE
// input halo tile [(PH+2)(PW+2) px][32 ch] is brought in by LDS-DMA and serves all nine taps (a tap is a row offset
// into the tile), instead of nine separately fetched, shifted copies; only the weights are streamed per step of TG
// taps (3-deep ring, two steps of loads in flight, counted vmcnt, one barrier per step).  That cuts the L2 -> LDS
// traffic per MFMA 2-3x and, with TG = 3, the barriers per MFMA 3x against the per-tap kernel above.
// Waves: BN = 128: 2 (M) x 2 (N), each 128 px x 64 channels; BN = 64: 4 x 1, each 64 px x 64 channels.
// ---------------------------------------------------------------------------------------------------------
template <int PW, int BN, int NT, int TG, int NPL>
__global__ __launch_bounds__(256, 2) void igemm_fwd3_kernel(const seg_igemm_fwd_args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int PH = 256 / PW, HC = PW + 2, HROWS = (PH + 2) * HC;
  constexpr int NI_A = (HROWS + 15) / 16;                  // LDS-DMA wave instructions per halo tile (16 rows of 64 B)
  constexpr int NA_HI = (NI_A + 3) / 4, NA_REM = NI_A % 4;  // per wave: NA_HI for waves < NA_REM (all when NA_REM == 0), else NA_HI - 1
  constexpr int A_BYTES = NI_A * 1024;
  constexpr int B_BYTES = TG * BN * 64, NB_W = B_BYTES / 4096;     // weight stage, wave instructions per wave and stage
  constexpr int NSTEP = NT / TG;
  constexpr int WMv = BN == 64 ? 4 : 2, WNv = 4 / WMv, TMw = 8 / WMv;   // A tiles (32 px) per wave; 2 B tiles (64 channels) per wave
  static_assert(TG * NSTEP == NT && NSTEP >= 2 && NB_W * 4096 == B_BYTES, "tap groups / weight stage");
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
  const int cls = lin0 / gridDim.x, lin = lin0 - cls * gridDim.x;      // sub-pixel class of a transposed conv (0 otherwise)
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
    const int lrow = lane >> 2, lch = lane & 3;
#pragma unroll
    for (int j = 0; j < NB_W; ++j) {
      const int r = (j * 4 + wave) * 16 + lrow;            // row inside the stage: tap-in-group * BN + channel row
      const int tt = r / BN, row = r - tt * BN;
      b_tap[j] = __builtin_amdgcn_readfirstlane(tt);
      const int gch = lch ^ ((row >> 2) & 3);
      b_voff[j] = (n0 + row < p.n_rows) ? (unsigned)(((n0 + row) * g.T_total * p.Kpad + gch * 8) * 2) : OOB_OFF;
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
        o0[a] = (row << 6) + ((fh ^ ((row >> 2) & 3)) << 4);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        h8 af[TMw], bf[2];
#pragma unroll
        for (int a = 0; a < TMw; ++a) af[a] = *reinterpret_cast<const h8*>(A + (o0[a] ^ (ks * 32)));
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[b] = *reinterpret_cast<const h8*>(B + tt * (BN * 64) + boff[b][ks]);
#pragma unroll
        for (int a = 0; a < TMw; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
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
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
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
  conv_epilogue<half_t, TMw, 2, WMv, BN, PW>(p, acc, smem, tid, PW == 32 ? wm * TMw : wm * TMw * 2, wn * 64, n0, wm, cls, npatch, mb, 0, 0, pn, y0, x0);
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
            atomicAdd(dw + off, acc[t][b][q] * sc_mul);
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

  // wave-uniform descriptors (readfirstlane returns int: keep the halves unsigned, or a low word with bit 31 set
  // sign-extends into the high word of the base address)
  auto uniform_ptr = [](const void* q) -> void* {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
  };
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.seg_ptr[seg]), 0, __builtin_amdgcn_readfirstlane(g.N * H * W * cs2), 0x00020000);
  const auto rsrc_dz = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.dz), 0, __builtin_amdgcn_readfirstlane(g.N * Ho * Wo * dzcs2), 0x00020000);

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
  int pn_ = q0 / (PH * PW);
  int py_ = (q0 - pn_ * (PH * PW)) / PW;
  int px_ = q0 - pn_ * (PH * PW) - py_ * PW;
  auto issue = [&](int /*q*/, int stage) {
    const int n = pn_, y0 = py_ * R, x0 = px_ * C;
    if (++px_ == PW) {
      px_ = 0;
      if (++py_ == PH) { py_ = 0; ++pn_; }
    }
    unsigned char* sb = smem + stage * STAGE;
    const int zs = ((n * Ho + y0 * osy + g.oy[cls]) * Wo + x0 * osx + g.ox[cls]) * dzcs2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_dz, (lds_ptr_t)(sb + (wave * 2 + i) * 1024), 16, z_voff[i], zs, 0, 0);
#pragma unroll
    for (int k = 0; k < NI_W; ++k) {
      if (wave + 4 * k < NI_IN) {
        const int y = y0 + in_dy[k], x = x0 + in_dx[k];
        const bool ok = in_cb[k] >= 0 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const unsigned voff = ok ? (unsigned)(((n * H + y) * W + x) * cs2 + in_cb[k]) : OOB_OFF;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (lds_ptr_t)(sb + Z_BYTES + (wave + 4 * k) * 1024), 16, voff, 0, 0, 0);
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
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (q + 1 < q1) issue(q + 1, stage ^ 1);
    compute(stage);
    stage ^= 1;
  }

  float sc_mul = p.scale;
  if (p.scale_ptr != nullptr) sc_mul *= *p.scale_ptr;
  float* dw = p.dw;
  const int fr = lane & 31, fh = lane >> 5;
  const int kcol = c0 + wn * 32 + fr;
  if (kcol < sc) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (TC && !((tmask >> t) & 1u)) continue;
      const int widx = wmap[t];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = r0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh;
        if (row < p.n_rows) {
          float* d = dw + ((size_t)row * g.T_total + widx) * p.Kpad + p.seg_woff[seg] + kcol;
          if (p.splitk == 1) {       // this block is the only writer of its tile: no atomic needed, and no read either
            if (p.overwrite) *d = acc[t][q] * sc_mul;          // when the caller vouches that dw holds zeros
            else *d += acc[t][q] * sc_mul;
          }
          else atomicAdd(d, acc[t][q] * sc_mul);
        }
      }
    }
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

  auto uniform_ptr = [](const void* q) -> void* {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
  };
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.seg_ptr[seg]), 0, __builtin_amdgcn_readfirstlane(g.N * H * W * cs2), 0x00020000);
  const auto rsrc_dz = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.dz), 0, __builtin_amdgcn_readfirstlane(g.N * Ho * Wo * dzcs2), 0x00020000);

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
  int pn_ = q0 / (PH * PW);
  int py_ = (q0 - pn_ * (PH * PW)) / PW;
  int px_ = q0 - pn_ * (PH * PW) - py_ * PW;
  auto issue = [&](int stage) {
    const int n = pn_, y0 = py_ * R, x0 = px_ * C;
    if (++px_ == PW) {
      px_ = 0;
      if (++py_ == PH) { py_ = 0; ++pn_; }
    }
    unsigned char* sb = smem + stage * STAGE;
    const int zs = ((n * Ho + y0 * 2 + py) * Wo + x0 * 2) * dzcs2;          // class (py, 0); (py, 1) is the next pixel
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_dz, (lds_ptr_t)(sb + c2 * Z_BYTES + (wave * 2 + i) * 1024), 16, z_voff[i], zs + c2 * dzcs2, 0, 0);
#pragma unroll
    for (int k = 0; k < NI_W; ++k) {
      if (wave + 4 * k < NI_IN) {
        const int y = y0 + in_dy[k], x = x0 + in_dx[k];
        const bool ok = in_cb[k] >= 0 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const unsigned voff = ok ? (unsigned)(((n * H + y) * W + x) * cs2 + in_cb[k]) : OOB_OFF;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (lds_ptr_t)(sb + 2 * Z_BYTES + (wave + 4 * k) * 1024), 16, voff, 0, 0, 0);
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
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
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
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int widx = g.tap_w[(py * 2 + (t >> 2)) * 4 + (t & 3)];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = r0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh;
        if (row < p.n_rows) {
          float* d = dw + ((size_t)row * g.T_total + widx) * p.Kpad + p.seg_woff[seg] + kcol;
          if (p.splitk == 1) {       // this block is the only writer of its tile: no atomic needed, and no read either
            if (p.overwrite) *d = acc[t][q] * sc_mul;          // when the caller vouches that dw holds zeros
            else *d += acc[t][q] * sc_mul;
          }
          else atomicAdd(d, acc[t][q] * sc_mul);
        }
      }
    }
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

static int launch_fwd3(const seg_igemm_fwd_args& a, hipStream_t st) {
  const int pw = seg_fwd_halo_patch(&a.g);
  SEG_CHECK_ARG(pw != 0, "seg_igemm_fwd: variant 3 needs a stride-1 3x3 'same' conv, a 4x4 / stride-2 transposed conv (or the data gradient of either), and W %% 32 == 0, H %% 8 == 0 (or W, H %% 16 == 0)");
  SEG_CHECK_ARG(a.tile_m == 256 && (a.tile_n == 64 || a.tile_n == 128), "seg_igemm_fwd: variant 3 tiles are 256x64 and 256x128, not %dx%d", a.tile_m, a.tile_n);
  const long pix = (long)a.g.N * a.g.Hin * a.g.Win;
  for (int s = 0; s < a.nseg; ++s) SEG_CHECK_ARG(pix * a.seg_cs[s] * 2 < (1L << 30), "seg_igemm_fwd: variant 3 needs operands below 1 GiB");
  SEG_CHECK_ARG((long)a.n_rows * a.g.T_total * a.Kpad * 2 < (1L << 30), "seg_igemm_fwd: variant 3 needs operands below 1 GiB");
  const int npatch = (int)((long)a.g.N * a.g.Hl * a.g.Wl / 256);
  const int nblocks = (a.n_rows + a.tile_n - 1) / a.tile_n;
  dim3 grid((unsigned)(npatch * nblocks), (unsigned)a.g.ncls, 1);
  if (a.g.ntaps == 16) launch_fwd3_nt<4, 1, 2, 4>(a, pw, grid, st);
  else if (a.g.ntaps == 9) launch_fwd3_nt<9, 1, 3>(a, pw, grid, st);
  else launch_fwd3_nt<4, 1, 2>(a, pw, grid, st);
  SEG_CHECK_LAUNCH("seg_igemm_fwd (halo tiles)");
  return SEG_OK;
}

// ---- v2 (LDS-DMA) dispatch
static bool fwd2_eligible(const seg_igemm_fwd_args& a) {
  static const bool force_v1 = std::getenv("SEG_IGEMM_V1") != nullptr;
  if (force_v1) return false;
  if (!((a.tile_m == 128 && (a.tile_n == 128 || a.tile_n == 64)) || (a.tile_m == 256 && a.tile_n == 64))) return false;
  const int nt = a.g.ntaps;
  if (!(nt == 1 || nt == 2 || nt == 3 || nt == 4 || nt == 9 || nt == 16)) return false;
  const long lim = 1L << 30;
  const long pix = (long)a.g.N * a.g.Hin * a.g.Win;
  for (int s = 0; s < a.nseg; ++s)
    if (pix * a.seg_cs[s] * 2 >= lim) return false;
  if ((long)a.n_rows * a.g.T_total * a.Kpad * 2 >= lim) return false;
  return true;
}

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
  if (a.Kpad & 63) return 32;
  int cmin = 1 << 30;
  for (int s = 0; s < a.nseg; ++s) cmin = a.seg_c[s] < cmin ? a.seg_c[s] : cmin;
  if (a.tile_m == 128 && a.tile_n == 128) return cmin >= 128 ? 64 : 32;
  if (cmin < 64) return 32;
  return (nblocks >= 512 && (long)a.g.ntaps * a.Kpad >= 1024) ? 64 : 32;
}

static int launch_fwd2(const seg_igemm_fwd_args& a, int mblocks, int nblocks, hipStream_t st) {
  dim3 grid((unsigned)(mblocks * nblocks), (unsigned)a.g.ncls, 1);
  // deep-and-small layers: with ~1 block per CU nothing else hides a block's barriers and load latency, so take both
  // the 64-deep K step (8 instead of 4 MFMAs per wave and barrier) and two steps of loads in flight (72 KB of LDS)
  static const bool no_deep = std::getenv("SEG_NO_DEEP_RING") != nullptr;
  if (!no_deep && std::getenv("SEG_V2_BK") == nullptr && a.tile_m == 128 && a.tile_n == 64 && (long)mblocks * nblocks * a.g.ncls <= 384 &&
      (a.g.ntaps == 9 || a.g.ntaps == 16) && (a.Kpad & 63) == 0) {
    bool ok = true;
    for (int s = 0; s < a.nseg; ++s) ok = ok && a.seg_c[s] >= 64 && (a.seg_c[s] & 63) == 0;
    if (ok) {
      const size_t lds = (size_t)3 * (128 + 64) * 128;
      if (a.g.ntaps == 9) hipLaunchKernelGGL((igemm_fwd2_kernel<128, 64, 9, 64, 3>), grid, dim3(256), lds, st, a);
      else hipLaunchKernelGGL((igemm_fwd2_kernel<128, 64, 16, 64, 3>), grid, dim3(256), lds, st, a);
      SEG_CHECK_LAUNCH("seg_igemm_fwd (LDS-DMA, deep ring)");
      return SEG_OK;
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
  SEG_CHECK_LAUNCH("seg_igemm_fwd (LDS-DMA)");
  return SEG_OK;
}

template <typename T>
static int igemm_fwd_launch(const seg_igemm_fwd_args* a, seg_stream_t stream) {
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
  SEG_CHECK_ARG(a->variant == 0 || a->variant == 3, "seg_igemm_fwd: variant=%d", a->variant);
  SEG_CHECK_ARG(a->act >= SEG_ACT_NONE && a->act <= SEG_ACT_SIGMOID, "seg_igemm_fwd: act=%d", a->act);
  if (a->variant == 3) {
    SEG_CHECK_ARG(sizeof(T) == 2, "seg_igemm_fwd: variant 3 (halo tiles) is fp16 only");
    if constexpr (sizeof(T) == 2) return launch_fwd3(*a, (hipStream_t)stream);
  }
  SEG_CHECK_ARG((a->tile_m == 256 && a->tile_n == 64) || ((a->tile_m == 128 || a->tile_m == 64) && (a->tile_n == 128 || a->tile_n == 64)),
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
  if constexpr (sizeof(T) == 2) {
    if (fwd2_eligible(*a)) return launch_fwd2(*a, mblocks, nblocks, st);
  }
  SEG_CHECK_ARG(a->tile_m <= 128, "seg_igemm_fwd: a 256-row tile needs the LDS-DMA kernel (fp16, taps in {1,2,3,4,9,16}, operands < 1 GiB)");
  if (a->tile_m == 128 && a->tile_n == 128) hipLaunchKernelGGL((igemm_fwd_kernel<T, 128, 128>), grid, block, lds, st, *a);
  else if (a->tile_m == 128 && a->tile_n == 64) hipLaunchKernelGGL((igemm_fwd_kernel<T, 128, 64>), grid, block, lds, st, *a);
  else if (a->tile_m == 64 && a->tile_n == 128) hipLaunchKernelGGL((igemm_fwd_kernel<T, 64, 128>), grid, block, lds, st, *a);
  else hipLaunchKernelGGL((igemm_fwd_kernel<T, 64, 64>), grid, block, lds, st, *a);
  SEG_CHECK_LAUNCH("seg_igemm_fwd");
  return SEG_OK;
}
extern "C" int seg_igemm_fwd(const seg_igemm_fwd_args* a, seg_stream_t stream) { return igemm_fwd_launch<half_t>(a, stream); }
extern "C" int seg_igemm_fwd_f32(const seg_igemm_fwd_args* a, seg_stream_t stream) { return igemm_fwd_launch<float>(a, stream); }

// taps handled per block by the weight-gradient kernels (1 = the per-tap kernel); exported so that the host can
// size the pixel split with the same rule
extern "C" int seg_wgrad_taps_per_block(int tile_m, int tile_n, int ntaps, int variant) {
  static const bool force_v1 = std::getenv("SEG_WGRAD_V1") != nullptr;
  if (force_v1 || variant != 0 || tile_m != 64) return 1;
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
  SEG_CHECK_ARG(a->variant >= 0 && a->variant <= 2, "seg_igemm_wgrad: variant=%d", a->variant);
  SEG_CHECK_ARG(a->bias_part == nullptr || (a->variant == 2 && seg_wgrad_halo_class_groups(&a->g) == 2),
                "seg_igemm_wgrad: bias_part needs variant 2 on a transposed conv of the class-pair kernel");
  if (a->variant == 2) {
    SEG_CHECK_ARG(sizeof(T) == 2, "seg_igemm_wgrad: variant 2 (LDS-DMA halo tiles) is fp16 only");
    const int pc = seg_wgrad_halo_patch(&a->g);
    SEG_CHECK_ARG(pc != 0, "seg_igemm_wgrad: variant 2 needs a 3x3 / stride-1 / 'same' conv or a 4x4 / stride-2 transposed conv, with W %% 32 == 0 and H %% 2 == 0, W %% 16 == 0 and H %% 4 == 0, or W and H %% 8 == 0");
    const long pix = (long)a->g.N * a->g.Hin * a->g.Win;
    SEG_CHECK_ARG((long)a->g.N * a->g.Hout * a->g.Wout * a->dz_cs * 2 < (1L << 30), "seg_igemm_wgrad: variant 2 needs operands below 1 GiB");
    int planes = 0;
    for (int s = 0; s < a->nseg; ++s) {
      SEG_CHECK_ARG(pix * a->seg_cs[s] * 2 < (1L << 30), "seg_igemm_wgrad: variant 2 needs operands below 1 GiB");
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
      if (a->g.ncls == 1) {
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
    return SEG_OK;
  }
  const int variant = (sizeof(T) == 4) ? 1 : a->variant;      // fp32: transposing-store form only
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
      return SEG_OK;
    }
  }
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
