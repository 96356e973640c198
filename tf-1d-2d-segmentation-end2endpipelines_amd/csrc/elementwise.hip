// HBM-bound kernels of the UNet-family hot path for gfx950: BatchNormalization (statistics, apply + activation
// [+ fused max-pool], backward), activations, pooling, up-sampling, add / gating multiply, ConvLSTM(T=1) gates,
// 1x1 output heads, losses, Adam, weight repacking.  All tensors NHWC fp16 with 16-B (8-channel) vector access:
// consecutive lanes read consecutive 16-B chunks of a pixel row, so every wave instruction touches whole
// 128-B lines.  Per-channel reductions use a (rows x chunk-columns) thread layout with register partials,
// an LDS cross-row reduction and one deterministic partial row per block.
#include "seg_common.h"
#include <cstdlib>
#include <stdarg.h>

// ------------------------------------------------------------------------------------------------ error state
static thread_local char g_err[512] = "";
void seg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* seg_last_error(void) { return g_err; }
#ifndef SEG_ABI_VERSION
#error "build through build_ext.py: SEG_ABI_VERSION (28 bits of the content hash of include/seg_hip.h) must be defined"
#endif
extern "C" int seg_abi_version(void) { return SEG_ABI_VERSION; }

namespace {

constexpr int TPB = 256;
constexpr int MAX_BLOCKS = 4096;     // grid-stride streaming kernels: measured flat from 2048 up, 10-15 % better than 1024 (tools/ew_bench.py)

inline int grid_for(int64_t items) {
  static const int cap = getenv("SEG_EW_MAX_BLOCKS") ? atoi(getenv("SEG_EW_MAX_BLOCKS")) : MAX_BLOCKS;
  int64_t b = (items + TPB - 1) / TPB;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}
inline int pow2ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// (rows x chunk-columns) layout for per-channel reductions over pixels.
struct ColLayout {
  int CH;    // 16-B chunks per pixel that take part
  int CHP;   // chunk columns per block pass (power of two, <= 256)
  int RP;    // pixel rows per block pass = 256 / CHP
  int ngrp;  // ceil(CH / CHP) column groups
};
inline ColLayout col_layout(int C) {
  ColLayout l;
  l.CH = C / 8;
  l.CHP = pow2ceil(l.CH);
  if (l.CHP > TPB) l.CHP = TPB;
  l.RP = TPB / l.CHP;
  l.ngrp = (l.CH + l.CHP - 1) / l.CHP;
  return l;
}

// Reduce NV per-thread vectors of 8 floats across the RP row-groups of the block and hand the result for chunk
// column `col` to thread row 0.  sm must hold TPB * 8 floats.
template <int NV>
__device__ __forceinline__ void block_col_reduce(float (&v)[NV][8], float* sm, int rowi, int col, int CHP, int RP) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 8; ++t) sm[(rowi * CHP + col) * 8 + t] = v[k][t];
    __syncthreads();
    if (rowi == 0) {
      for (int r = 1; r < RP; ++r)
#pragma unroll
        for (int t = 0; t < 8; ++t) v[k][t] += sm[(r * CHP + col) * 8 + t];
    }
  }
}

// ------------------------------------------------------------------------------------------------ BN statistics
template <typename T>
__global__ __launch_bounds__(TPB) void bn_stats_kernel(const T* x, int64_t npix, int cs, int coff, int C, float* stats,
                                                       int nparts, int stats_ld, int stats_coff, ColLayout L) {
  __shared__ float sm[TPB * 8];
  const int col = threadIdx.x % L.CHP, rowi = threadIdx.x / L.CHP;
  const int part = blockIdx.x;
  for (int grp = 0; grp < L.ngrp; ++grp) {
    const int ch = grp * L.CHP + col;
    float acc[2][8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc[0][t] = 0.f; acc[1][t] = 0.f; }
    if (ch < L.CH) {
      for (int64_t pix = (int64_t)part * L.RP + rowi; pix < npix; pix += (int64_t)L.RP * nparts) {   // interleaved rows
        float f[8];
        load8(x + pix * cs + coff + ch * 8, f);
#pragma unroll
        for (int t = 0; t < 8; ++t) { acc[0][t] += f[t]; acc[1][t] += f[t] * f[t]; }
      }
    }
    block_col_reduce<2>(acc, sm, rowi, col, L.CHP, L.RP);
    if (rowi == 0 && ch < L.CH) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        stats[((size_t)part * 2 + 0) * stats_ld + stats_coff + ch * 8 + t] = acc[0][t];
        stats[((size_t)part * 2 + 1) * stats_ld + stats_coff + ch * 8 + t] = acc[1][t];
      }
    }
  }
}

// fold parts {s, s+S, s+2S, ...} into part s, in place (deterministic order)
__global__ __launch_bounds__(TPB) void bn_prereduce_kernel(float* stats, int nparts, int C, int S) {
  __shared__ float sm[2][8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), pl = threadIdx.x >> 5, s = blockIdx.y;
  float a0 = 0.f, a1 = 0.f;
  if (c < C)
    for (int p = s + pl * S; p < nparts; p += 8 * S) {
      a0 += stats[((size_t)p * 2 + 0) * C + c];
      a1 += stats[((size_t)p * 2 + 1) * C + c];
    }
  sm[0][pl][threadIdx.x & 31] = a0;
  sm[1][pl][threadIdx.x & 31] = a1;
  __syncthreads();
  if (pl == 0 && c < C) {
    for (int k = 1; k < 8; ++k) { a0 += sm[0][k][threadIdx.x & 31]; a1 += sm[1][k][threadIdx.x & 31]; }
    stats[((size_t)s * 2 + 0) * C + c] = a0;
    stats[((size_t)s * 2 + 1) * C + c] = a1;
  }
}

// Sum `nparts` partial rows for 16 channels per block with 16 part-lanes per channel (deterministic order),
// result for channel (blockIdx.x*16 + c) lands in lane pl == 0.
__device__ __forceinline__ void reduce_parts16(const float* stats, int nparts, int C, int c, int pl, float& s, float& q,
                                               float (*sm)[16][16]) {
  s = 0.f;
  q = 0.f;
  if (c < C) {
    // four rows per trip, all eight loads issued before the adds (a dependent chain of nparts / 16 L2 round trips is
    // what these tiny launches consist of); the summation order stays fixed
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = pl; p < nparts; p += 64) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = p + 16 * u;
        a[u] = pp < nparts ? stats[((size_t)pp * 2 + 0) * C + c] : 0.f;
        b[u] = pp < nparts ? stats[((size_t)pp * 2 + 1) * C + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s4[u] += a[u]; q4[u] += b[u]; }
    }
    s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
  }
  sm[0][pl][threadIdx.x & 15] = s;
  sm[1][pl][threadIdx.x & 15] = q;
  __syncthreads();
  if (pl == 0)
    for (int k = 1; k < 16; ++k) { s += sm[0][k][threadIdx.x & 15]; q += sm[1][k][threadIdx.x & 15]; }
}

__global__ __launch_bounds__(TPB) void bn_finalize_kernel(const float* stats, int nparts, int C, float inv_count, float unbias,
                                                          const float* gamma, const float* beta, float eps, float momentum,
                                                          float* mmean, float* mvar, int training, float* mean_o, float* invstd_o,
                                                          float* scale_o, float* shift_o) {
  __shared__ float sm[2][16][16];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), pl = threadIdx.x >> 4;
  float s = 0.f, q = 0.f;
  if (training) reduce_parts16(stats, nparts, C, c, pl, s, q, sm);
  if (pl != 0 || c >= C) return;
  float mean, var;
  if (training) {
    mean = s * inv_count;
    var = fmaxf(q * inv_count - mean * mean, 0.f);
    mmean[c] = mmean[c] * momentum + mean * (1.f - momentum);
    mvar[c] = mvar[c] * momentum + var * unbias * (1.f - momentum);
  } else {
    mean = mmean[c];
    var = mvar[c];
  }
  const float invstd = rsqrtf(var + eps);
  const float sc = gamma[c] * invstd;
  if (mean_o) mean_o[c] = mean;
  if (invstd_o) invstd_o[c] = invstd;
  scale_o[c] = sc;
  shift_o[c] = beta[c] - mean * sc;
}

__global__ __launch_bounds__(TPB) void bn_fold_inference_kernel(const float* gamma, const float* beta, const float* mmean, const float* mvar,
                                                                float eps, const float* cbias, int C, float* scale_o, float* shift_o) {
  const int c = blockIdx.x * TPB + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] * rsqrtf(mvar[c] + eps);
  scale_o[c] = sc;
  shift_o[c] = beta[c] + ((cbias != nullptr ? cbias[c] : 0.f) - mmean[c]) * sc;
}

// ------------------------------------------------------------------------------------------------ BN apply (+pool)
// Streaming kernels below process UNR items per thread and iteration with every load issued before the first
// store (the output may alias an input element-for-element, so the compiler cannot hoist loads across stores on
// its own), and keep the per-channel coefficients in registers whenever a thread's channel chunk is loop-invariant
// (grid stride a multiple of the chunks per pixel - always true for power-of-two channel counts).
constexpr int UNR = 4;

// pixel index of flat item idx = pix * CH + chunk.  A 64-bit division costs tens of VALU instructions per item -
// more than the memory system gives these kernels time for - so power-of-two chunk counts (every 8.. 2048-channel
// tensor of the hot path) take a shift.
__device__ __forceinline__ int ch_shift(int CH) { return (CH & (CH - 1)) == 0 ? __ffs(CH) - 1 : -1; }
__device__ __forceinline__ int64_t div_ch(int64_t idx, int CH, int shift) { return shift >= 0 ? (idx >> shift) : idx / CH; }

// HEAD: the tensor also feeds a one-channel 1x1 output head (a deep-supervision level of UNet++, unet_variants.py:300-302, or
// the model output): the head's dot product over the channels rides on this pass - the CH lanes of a pixel reduce their partial
// dots with shuffles and the first one writes act(sum + bias) - instead of a seg_head_fwd launch that reads the tensor again.
struct HeadFwd { int n; const float* w[2]; const float* b[2]; int act[2]; float* y[2]; };      // (a tensor carries at most two: UNet++'s X0d feeds level d and the output)
template <typename T, bool HEAD>
__global__ __launch_bounds__(TPB) void bn_act_apply_kernel(const T* x, int x_cs, int x_coff, T* y, int y_cs, int y_coff,
                                                           int64_t npix, int C, int c_valid, const float* scale, const float* shift,
                                                           int act, float alpha, HeadFwd hd) {
  const int CH = C / 8;
  const int64_t items = npix * CH, stride = (int64_t)gridDim.x * TPB;
  const int64_t it0 = (int64_t)blockIdx.x * TPB + threadIdx.x;
  const bool fixed = HEAD || (stride % CH) == 0;       // (HEAD: CH is a power of two <= 64, checked by the host - always fixed)
  const int chs = ch_shift(CH);
  const int chf = (int)(it0 % CH);
  float sc[8], sh[8], hw[2][8], hb[2];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    sc[t] = scale[chf * 8 + t]; sh[t] = shift[chf * 8 + t];
#pragma unroll
    for (int k = 0; k < 2; ++k) hw[k][t] = (HEAD && k < hd.n && chf * 8 + t < c_valid) ? hd.w[k][chf * 8 + t] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) hb[k] = (HEAD && k < hd.n && hd.b[k] != nullptr) ? hd.b[k][0] : 0.f;
  for (int64_t it = it0; it < items; it += stride * UNR) {
    float f[UNR][8];
    int64_t pix[UNR];
    int ch[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t idx = it + u * stride;
      pix[u] = div_ch(idx, CH, chs);
      ch[u] = fixed ? chf : (int)(idx - pix[u] * CH);
      if (idx < items) load8(x + pix[u] * x_cs + x_coff + ch[u] * 8, f[u]);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (it + u * stride >= items) continue;      // (items is a multiple of CH and a pixel's CH lanes are consecutive: whole groups skip)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int c = ch[u] * 8 + t;
        const float a_ = fixed ? sc[t] : scale[c], b_ = fixed ? sh[t] : shift[c];
        const float v = seg_act(f[u][t] * a_ + b_, act, alpha);
        f[u][t] = c < c_valid ? v : 0.f;
      }
      store8(y + pix[u] * y_cs + y_coff + ch[u] * 8, f[u]);
      if (HEAD) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (k >= hd.n) continue;
          float dot = 0.f;
#pragma unroll
          for (int t = 0; t < 8; ++t) dot += stored<T>(f[u][t]) * hw[k][t];      // the head reads the tensor as stored
          for (int o = CH >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
          if (chf == 0) hd.y[k][pix[u]] = seg_act(dot + hb[k], hd.act[k], 0.f);
        }
      }
    }
  }
}

template <typename T, bool HEAD>
__global__ __launch_bounds__(TPB) void bn_act_apply_pool_kernel(const T* x, int x_cs, int x_coff, T* y, int y_cs,
                                                                int y_coff, int N, int H, int W, int C, int c_valid,
                                                                const float* scale, const float* shift, int act, float alpha,
                                                                T* pooled, int ph, int pw, HeadFwd hd) {
  const int CH = C / 8;
  const int HP = H / ph, WP = W / pw;                  // pooled extent ('valid')
  const int HB = (H + ph - 1) / ph, WB = (W + pw - 1) / pw;   // windows incl. ragged border (no pooled output there)
  const int64_t items = (int64_t)N * HB * WB * CH;
  // (HEAD: CH is a power of two <= 64, so the CH lanes of a window are consecutive lanes of one wave and walk the window together)
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < items; it += (int64_t)gridDim.x * TPB) {
    int64_t r = it;
    const int ch = (int)(r % CH); r /= CH;
    const int wb = (int)(r % WB); r /= WB;
    const int hb = (int)(r % HB);
    const int n = (int)(r / HB);
    float sc[8], sh[8], mx[8], hw[2][8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      sc[t] = scale[ch * 8 + t]; sh[t] = shift[ch * 8 + t]; mx[t] = -3.0e38f;
#pragma unroll
      for (int k = 0; k < 2; ++k) hw[k][t] = (HEAD && k < hd.n && ch * 8 + t < c_valid) ? hd.w[k][ch * 8 + t] : 0.f;
    }
    for (int dy = 0; dy < ph; ++dy)
      for (int dx = 0; dx < pw; ++dx) {
        const int yy = hb * ph + dy, xx = wb * pw + dx;
        if (yy >= H || xx >= W) continue;
        const int64_t pix = ((int64_t)n * H + yy) * W + xx;
        float f[8];
        load8(x + pix * x_cs + x_coff + ch * 8, f);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          float v = seg_act(f[t] * sc[t] + sh[t], act, alpha);
          v = (ch * 8 + t) < c_valid ? v : 0.f;
          f[t] = v;
        }
        store8(y + pix * y_cs + y_coff + ch * 8, f);
#pragma unroll
        for (int t = 0; t < 8; ++t) mx[t] = fmaxf(mx[t], stored<T>(f[t]));   // pool the values as stored (pooled == max of y)
        if (HEAD) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            if (k >= hd.n) continue;
            float dot = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) dot += stored<T>(f[t]) * hw[k][t];
            for (int o = CH >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
            if (ch == 0) hd.y[k][pix] = seg_act(dot + (hd.b[k] != nullptr ? hd.b[k][0] : 0.f), hd.act[k], 0.f);
          }
        }
      }
    if (hb < HP && wb < WP) {
      const int64_t pp = ((int64_t)n * HP + hb) * WP + wb;
      store8(pooled + pp * y_cs + ch * 8, mx);
    }
  }
}

// ------------------------------------------------------------------------------------------------ BN backward
// INJ: the tensor also feeds a one-channel 1x1 output head: the head's data gradient gs * dpre[pix] * w[c] (dpre = dy act'(y) of
// the head) is ADDED ON THE FLY to the gradient read here instead of being written to the gradient tensor by seg_head_bwd first
// (a write of the whole tensor, and a read-modify-write by the data gradient that follows); dy == NULL: the head is the tensor's only
// consumer.
struct HeadInj { int n; const float* dy[2]; const float* y[2]; const float* w[2]; int act[2]; const float* gs; };
__device__ __forceinline__ float head_inj_dpre(const HeadInj& hj, int k, int64_t pix) {
  const float d = hj.dy[k][pix] * seg_act_grad_from_out(hj.y[k][pix], hj.act[k], 0.f);
  return hj.gs != nullptr ? d * hj.gs[0] : d;
}

template <typename T, bool INJ>
__global__ __launch_bounds__(TPB) void bn_act_bwd_reduce_kernel(const T* dy, int dy_cs, int dy_coff, const T* x, int x_cs,
                                                                int x_coff, int64_t npix, int C, const float* scale,
                                                                const float* shift, const float* mean, const float* invstd, int act,
                                                                float alpha, float* partial, int nparts, ColLayout L, HeadInj hj) {
  __shared__ float sm[TPB * 8];
  const int col = threadIdx.x % L.CHP, rowi = threadIdx.x / L.CHP;
  const int part = blockIdx.x;
  for (int grp = 0; grp < L.ngrp; ++grp) {
    const int ch = grp * L.CHP + col;
    float acc[2][8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc[0][t] = 0.f; acc[1][t] = 0.f; }
    if (ch < L.CH) {
      float sc[8], sh[8], mu[8], is[8], hw[2][8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int c = ch * 8 + t;
        sc[t] = scale[c]; sh[t] = shift[c]; mu[t] = mean[c]; is[t] = invstd[c];
#pragma unroll
        for (int k = 0; k < 2; ++k) hw[k][t] = (INJ && k < hj.n) ? hj.w[k][c] : 0.f;
      }
      for (int64_t pix = (int64_t)part * L.RP + rowi; pix < npix; pix += (int64_t)L.RP * nparts) {   // interleaved rows
        float g[8], xv[8];
        if (!INJ || dy != nullptr) {
          load8(dy + pix * dy_cs + dy_coff + ch * 8, g);
        } else {
#pragma unroll
          for (int t = 0; t < 8; ++t) g[t] = 0.f;
        }
        load8(x + pix * x_cs + x_coff + ch * 8, xv);
        const float hp0 = INJ ? head_inj_dpre(hj, 0, pix) : 0.f, hp1 = (INJ && hj.n > 1) ? head_inj_dpre(hj, 1, pix) : 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float yv = seg_act(xv[t] * sc[t] + sh[t], act, alpha);
          const float gg = (g[t] + hp0 * hw[0][t] + hp1 * hw[1][t]) * seg_act_grad_from_out(yv, act, alpha);
          acc[0][t] += gg;
          acc[1][t] += gg * (xv[t] - mu[t]) * is[t];
        }
      }
    }
    block_col_reduce<2>(acc, sm, rowi, col, L.CHP, L.RP);
    if (rowi == 0 && ch < L.CH) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        partial[((size_t)part * 2 + 0) * C + ch * 8 + t] = acc[0][t];
        partial[((size_t)part * 2 + 1) * C + ch * 8 + t] = acc[1][t];
      }
    }
  }
}

__global__ __launch_bounds__(TPB) void bn_bwd_finalize_kernel(const float* partial, int nparts, int C, float inv_count, float* dgamma,
                                                              float* dbeta, float gscale, const float* gscale_ptr, float* c1,
                                                              float* c2) {
  __shared__ float sm[2][16][16];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), pl = threadIdx.x >> 4;
  float s, q;
  reduce_parts16(partial, nparts, C, c, pl, s, q, sm);
  if (pl != 0 || c >= C) return;
  float gs = gscale;
  if (gscale_ptr) gs *= *gscale_ptr;
  if (dbeta) dbeta[c] += s * gs;
  if (dgamma) dgamma[c] += q * gs;
  c1[c] = s * inv_count;
  c2[c] = q * inv_count;
}

template <typename T, bool INJ>
__global__ __launch_bounds__(TPB) void bn_act_bwd_apply_kernel(const T* dy, int dy_cs, int dy_coff, const T* x, int x_cs,
                                                               int x_coff, T* dx, int dx_cs, int dx_coff, int64_t npix, int C,
                                                               const float* scale, const float* shift, const float* mean,
                                                               const float* invstd, const float* c1, const float* c2, int act,
                                                               float alpha, int accumulate, ColLayout L, HeadInj hj) {
  // thread = (pixel row rowi, channel chunk col); blocks walk the tensor in interleaved chunks of RP * UNR pixels (as
  // act_bwd_kernel: concurrently running blocks stream neighbouring addresses - 2.9 -> 4.5 TB/s against a plain
  // grid-stride loop whose four loads per thread are megabytes apart); the channel chunk of a thread is fixed, so
  // the per-channel coefficients live in registers
  const int col = threadIdx.x % L.CHP, rowi = threadIdx.x / L.CHP;
  const bool train = c1 != nullptr;
  const int64_t chunk = (int64_t)L.RP * UNR;
  for (int grp = 0; grp < L.ngrp; ++grp) {
    const int ch = grp * L.CHP + col;
    if (ch >= L.CH) continue;
    // dx = sc * (g - c1 - (x - mu) * is * c2) = g * sc + x * ka + kb   with ka = -sc*is*c2, kb = sc*(mu*is*c2 - c1)
    float sc[8], sh[8], ka[8], kb[8], hw[2][8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int c = ch * 8 + t;
      sc[t] = scale[c];
      sh[t] = shift[c];
#pragma unroll
      for (int k = 0; k < 2; ++k) hw[k][t] = (INJ && k < hj.n) ? hj.w[k][c] : 0.f;
      if (train) {
        const float k = invstd[c] * c2[c];
        ka[t] = -sc[t] * k;
        kb[t] = sc[t] * (mean[c] * k - c1[c]);
      } else {
        ka[t] = 0.f;
        kb[t] = 0.f;
      }
    }
    for (int64_t pb = blockIdx.x * chunk + rowi; pb < npix; pb += chunk * gridDim.x) {
      float g[UNR][8], xv[UNR][8], old[UNR][8], hp[UNR][2];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int64_t pix = pb + (int64_t)u * L.RP;
        hp[u][0] = 0.f;
        hp[u][1] = 0.f;
        if (pix < npix) {
          if (!INJ || dy != nullptr) {
            load8(dy + pix * dy_cs + dy_coff + ch * 8, g[u]);
          } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) g[u][t] = 0.f;
          }
          load8(x + pix * x_cs + x_coff + ch * 8, xv[u]);
          if (accumulate) load8(dx + pix * dx_cs + dx_coff + ch * 8, old[u]);
          if (INJ) {
            hp[u][0] = head_inj_dpre(hj, 0, pix);
            if (hj.n > 1) hp[u][1] = head_inj_dpre(hj, 1, pix);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int64_t pix = pb + (int64_t)u * L.RP;
        if (pix >= npix) continue;
        float o[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float yv = seg_act(xv[u][t] * sc[t] + sh[t], act, alpha);
          const float gg = (g[u][t] + hp[u][0] * hw[0][t] + hp[u][1] * hw[1][t]) * seg_act_grad_from_out(yv, act, alpha);
          o[t] = train ? gg * sc[t] + xv[u][t] * ka[t] + kb[t] : gg * sc[t];
          if (accumulate) o[t] += old[u][t];
        }
        store8(dx + pix * dx_cs + dx_coff + ch * 8, o);
      }
    }
  }
}

// The same pass for TWO Conv_Blocks that received the SAME dOut (the two inputs of a fused ResPath tail, unet_variants.py:110-112:
// the add hands one gradient to the 1x1 shortcut block and to the 3x3 block): dOut is read once, both dz are written.  dx of a set
// may alias dy (each thread reads its elements before it writes them).
struct BwdSet { const void* x; void* dx; const float *scale, *shift, *mean, *invstd, *c1, *c2; int act; float alpha; };
struct BwdSet2 { BwdSet s[2]; };
template <typename T>
__global__ __launch_bounds__(TPB) void bn_act_bwd_apply2_kernel(const T* dy, BwdSet2 bs, int64_t npix, int C, ColLayout L) {
  constexpr int UN = 2;
  const int col = threadIdx.x % L.CHP, rowi = threadIdx.x / L.CHP;
  const int64_t chunk = (int64_t)L.RP * UN;
  for (int grp = 0; grp < L.ngrp; ++grp) {
    const int ch = grp * L.CHP + col;
    if (ch >= L.CH) continue;
    float sc[2][8], sh[2][8], ka[2][8], kb[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int c = ch * 8 + t;
        sc[j][t] = bs.s[j].scale[c];
        sh[j][t] = bs.s[j].shift[c];
        const float k = bs.s[j].invstd[c] * bs.s[j].c2[c];
        ka[j][t] = -sc[j][t] * k;
        kb[j][t] = sc[j][t] * (bs.s[j].mean[c] * k - bs.s[j].c1[c]);
      }
    for (int64_t pb = blockIdx.x * chunk + rowi; pb < npix; pb += chunk * gridDim.x) {
      float g[UN][8], xv[2][UN][8];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int64_t pix = pb + (int64_t)u * L.RP;
        if (pix < npix) {
          load8(dy + pix * C + ch * 8, g[u]);
#pragma unroll
          for (int j = 0; j < 2; ++j) load8((const T*)bs.s[j].x + pix * C + ch * 8, xv[j][u]);
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int64_t pix = pb + (int64_t)u * L.RP;
        if (pix >= npix) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float o[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const float yv = seg_act(xv[j][u][t] * sc[j][t] + sh[j][t], bs.s[j].act, bs.s[j].alpha);
            const float gg = g[u][t] * seg_act_grad_from_out(yv, bs.s[j].act, bs.s[j].alpha);
            o[t] = gg * sc[j][t] + xv[j][u][t] * ka[j][t] + kb[j][t];
          }
          store8((T*)bs.s[j].dx + pix * C + ch * 8, o);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ activation backward
template <typename T>
__global__ __launch_bounds__(TPB) void act_bwd_kernel(const T* dy, const T* y, T* dx, int64_t npix, int cs, int C,
                                                      int act, float alpha, float* dbias, float gscale, const float* gscale_ptr,
                                                      ColLayout L) {
  __shared__ float sm[TPB * 8];
  const int col = threadIdx.x % L.CHP, rowi = threadIdx.x / L.CHP;
  // blocks walk the tensor in interleaved chunks of RP*UNR pixels, so concurrently running blocks stream
  // neighbouring addresses (DRAM page locality) instead of 1000+ far-apart private ranges
  const int64_t p1 = npix;
  const int64_t chunk = (int64_t)L.RP * UNR;
  float gs = gscale;
  if (gscale_ptr) gs *= *gscale_ptr;
  for (int grp = 0; grp < L.ngrp; ++grp) {
    const int ch = grp * L.CHP + col;
    float acc[1][8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[0][t] = 0.f;
    if (ch < L.CH) {
      for (int64_t pb = blockIdx.x * chunk + rowi; pb < p1; pb += chunk * gridDim.x) {
        float g[UNR][8], yv[UNR][8];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int64_t pix = pb + (int64_t)u * L.RP;
          if (pix < p1) {
            load8(dy + pix * cs + ch * 8, g[u]);
            load8(y + pix * cs + ch * 8, yv[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int64_t pix = pb + (int64_t)u * L.RP;
          if (pix >= p1) continue;
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            g[u][t] *= seg_act_grad_from_out(yv[u][t], act, alpha);
            acc[0][t] += g[u][t];
          }
          store8(dx + pix * cs + ch * 8, g[u]);
        }
      }
    }
    if (dbias != nullptr) {
      block_col_reduce<1>(acc, sm, rowi, col, L.CHP, L.RP);
      if (rowi == 0 && ch < L.CH) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (ch * 8 + t < C) atomicAdd(dbias + ch * 8 + t, acc[0][t] * gs);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ max pooling
template <typename T>
__global__ __launch_bounds__(TPB) void maxpool_fwd_kernel(const T* x, T* y, int N, int H, int W, int cs, int ph, int pw) {
  const int CH = cs / 8, HP = H / ph, WP = W / pw;
  const int64_t items = (int64_t)N * HP * WP * CH;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < items; it += (int64_t)gridDim.x * TPB) {
    int64_t r = it;
    const int ch = (int)(r % CH); r /= CH;
    const int wp = (int)(r % WP); r /= WP;
    const int hp = (int)(r % HP);
    const int n = (int)(r / HP);
    float mx[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) mx[t] = -3.0e38f;
    for (int dy = 0; dy < ph; ++dy)
      for (int dx = 0; dx < pw; ++dx) {
        float f[8];
        load8(x + (((int64_t)n * H + hp * ph + dy) * W + wp * pw + dx) * cs + ch * 8, f);
#pragma unroll
        for (int t = 0; t < 8; ++t) mx[t] = fmaxf(mx[t], f[t]);
      }
    store8(y + it * 8, mx);
  }
}

template <typename T>
__global__ __launch_bounds__(TPB) void maxpool_bwd_kernel(const T* dy, const T* x, const T* y, T* dx, int N, int H,
                                                          int W, int cs, int ph, int pw, int accumulate) {
  const int CH = cs / 8, HP = H / ph, WP = W / pw;
  const int HB = (H + ph - 1) / ph, WB = (W + pw - 1) / pw;
  const int64_t items = (int64_t)N * HB * WB * CH;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < items; it += (int64_t)gridDim.x * TPB) {
    int64_t r = it;
    const int ch = (int)(r % CH); r /= CH;
    const int wb = (int)(r % WB); r /= WB;
    const int hb = (int)(r % HB);
    const int n = (int)(r / HB);
    const bool full = hb < HP && wb < WP;
    float g[8], mx[8];
    bool done[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { g[t] = 0.f; mx[t] = 0.f; done[t] = false; }
    if (full) {
      const int64_t pp = ((int64_t)n * HP + hb) * WP + wb;
      load8(dy + pp * cs + ch * 8, g);
      load8(y + pp * cs + ch * 8, mx);
    }
    for (int dyy = 0; dyy < ph; ++dyy)
      for (int dxx = 0; dxx < pw; ++dxx) {
        const int yy = hb * ph + dyy, xx = wb * pw + dxx;
        if (yy >= H || xx >= W) continue;
        T* dst = dx + (((int64_t)n * H + yy) * W + xx) * cs + ch * 8;
        float o[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] = 0.f;
        if (full) {
          float f[8];
          load8(x + (((int64_t)n * H + yy) * W + xx) * cs + ch * 8, f);
#pragma unroll
          for (int t = 0; t < 8; ++t)
            if (!done[t] && f[t] == mx[t]) { o[t] = g[t]; done[t] = true; }
        }
        if (accumulate) {
          float old[8];
          load8(dst, old);
#pragma unroll
          for (int t = 0; t < 8; ++t) o[t] += old[t];
        }
        store8(dst, o);
      }
  }
}

// ------------------------------------------------------------------------------------------------ up-sampling
__device__ __forceinline__ void bilin_src(int o, int f, int n_in, int& i0, int& i1, float& w1) {
  const float src = (o + 0.5f) / (float)f - 0.5f;
  const float fl = floorf(src);
  w1 = src - fl;
  int a = (int)fl;
  i0 = min(max(a, 0), n_in - 1);
  i1 = min(max(a + 1, 0), n_in - 1);
}

template <typename T>
__global__ __launch_bounds__(TPB) void upsample_fwd_kernel(const T* x, T* y, int N, int H, int W, int cs, int fh, int fw,
                                                           int bilinear) {
  const int CH = cs / 8, HO = H * fh, WO = W * fw;
  const int64_t items = (int64_t)N * HO * WO * CH;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < items; it += (int64_t)gridDim.x * TPB) {
    int64_t r = it;
    const int ch = (int)(r % CH); r /= CH;
    const int xo = (int)(r % WO); r /= WO;
    const int yo = (int)(r % HO);
    const int n = (int)(r / HO);
    float o[8];
    if (bilinear) {
      int y0, y1, x0, x1;
      float wy, wx;
      bilin_src(yo, fh, H, y0, y1, wy);
      bilin_src(xo, fw, W, x0, x1, wx);
      float a[8], b[8], c[8], d[8];
      const T* base = x + (int64_t)n * H * W * cs + ch * 8;
      load8(base + ((int64_t)y0 * W + x0) * cs, a);
      load8(base + ((int64_t)y0 * W + x1) * cs, b);
      load8(base + ((int64_t)y1 * W + x0) * cs, c);
      load8(base + ((int64_t)y1 * W + x1) * cs, d);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float top = a[t] + (b[t] - a[t]) * wx, bot = c[t] + (d[t] - c[t]) * wx;
        o[t] = top + (bot - top) * wy;
      }
    } else {
      load8(x + (((int64_t)n * H + yo / fh) * W + xo / fw) * cs + ch * 8, o);
    }
    store8(y + it * 8, o);
  }
}

// gather form of the transpose: every input pixel sums the output pixels whose footprint contains it
template <typename T>
__global__ __launch_bounds__(TPB) void upsample_bwd_kernel(const T* dy, T* dx, int N, int H, int W, int cs, int fh, int fw,
                                                           int bilinear, int accumulate) {
  const int CH = cs / 8, HO = H * fh, WO = W * fw;
  const int64_t items = (int64_t)N * H * W * CH;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < items; it += (int64_t)gridDim.x * TPB) {
    int64_t r = it;
    const int ch = (int)(r % CH); r /= CH;
    const int xi = (int)(r % W); r /= W;
    const int yi = (int)(r % H);
    const int n = (int)(r / H);
    float o[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) o[t] = 0.f;
    const T* base = dy + (int64_t)n * HO * WO * cs + ch * 8;
    if (bilinear) {
      const int ylo = max(0, (yi - 1) * fh), yhi = min(HO, (yi + 2) * fh);
      const int xlo = max(0, (xi - 1) * fw), xhi = min(WO, (xi + 2) * fw);
      for (int yo = ylo; yo < yhi; ++yo) {
        int y0, y1; float wy;
        bilin_src(yo, fh, H, y0, y1, wy);
        const float cy = (y0 == yi ? 1.f - wy : 0.f) + (y1 == yi ? wy : 0.f);
        if (cy == 0.f) continue;
        for (int xo = xlo; xo < xhi; ++xo) {
          int x0, x1; float wx;
          bilin_src(xo, fw, W, x0, x1, wx);
          const float cx = (x0 == xi ? 1.f - wx : 0.f) + (x1 == xi ? wx : 0.f);
          if (cx == 0.f) continue;
          float g[8];
          load8(base + ((int64_t)yo * WO + xo) * cs, g);
#pragma unroll
          for (int t = 0; t < 8; ++t) o[t] += g[t] * (cy * cx);
        }
      }
    } else {
      for (int a = 0; a < fh; ++a)
        for (int b = 0; b < fw; ++b) {
          float g[8];
          load8(base + ((int64_t)(yi * fh + a) * WO + xi * fw + b) * cs, g);
#pragma unroll
          for (int t = 0; t < 8; ++t) o[t] += g[t];
        }
    }
    T* dst = dx + it * 8;
    if (accumulate) {
      float old[8];
      load8(dst, old);
#pragma unroll
      for (int t = 0; t < 8; ++t) o[t] += old[t];
    }
    store8(dst, o);
  }
}

// ------------------------------------------------------------------------------------------------ add / mul
template <typename T> struct Ptr4 { const T* p[4]; };
template <typename T> struct MPtr4 { T* p[4]; };

template <typename T>
__global__ __launch_bounds__(TPB) void add_act_fwd_kernel(Ptr4<T> xs, int n, T* y, int64_t nchunks, int CH, int c_valid, int act, float alpha) {
  // padding channels (>= c_valid) must stay exactly zero: act(0) is not 0 for a sigmoid.  Masked only when the row has padding at all
  const bool mask = c_valid < CH * 8 && act == SEG_ACT_SIGMOID;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < nchunks; it += (int64_t)gridDim.x * TPB) {
    float o[8];
    load8(xs.p[0] + it * 8, o);
    for (int k = 1; k < n; ++k) {
      float f[8];
      load8(xs.p[k] + it * 8, f);
#pragma unroll
      for (int t = 0; t < 8; ++t) o[t] += f[t];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) o[t] = seg_act(o[t], act, alpha);
    if (mask) {
      const int c0 = (int)(it % CH) * 8;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (c0 + t >= c_valid) o[t] = 0.f;
    }
    store8(y + it * 8, o);
  }
}

template <typename T>
__global__ __launch_bounds__(TPB) void add_act_bwd_kernel(const T* dy, const T* y, MPtr4<T> dxs, int n, int64_t nchunks, int act,
                                                          float alpha, int accmask) {
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < nchunks; it += (int64_t)gridDim.x * TPB) {
    float g[8];
    load8(dy + it * 8, g);
    if (act != SEG_ACT_NONE) {
      float yv[8];
      load8(y + it * 8, yv);
#pragma unroll
      for (int t = 0; t < 8; ++t) g[t] *= seg_act_grad_from_out(yv[t], act, alpha);
    }
    for (int k = 0; k < n; ++k) {
      if (dxs.p[k] == nullptr) continue;
      float o[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) o[t] = g[t];
      if (accmask & (1 << k)) {
        float old[8];
        load8(dxs.p[k] + it * 8, old);
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] += old[t];
      }
      store8(dxs.p[k] + it * 8, o);
    }
  }
}

// ---- the tails of MultiResBlock / ResPath: BatchNormalization(ReLU(a + b)) (unet_variants.py:97-99, :118-120) ---------------------
// forward   y = act(sum_k pre_k(x_k) [+ cat_scale * cat + cat_shift]) with the per-channel sums of y and y^2 as per-block partial
//           rows (the statistics of the BatchNormalization that follows): the add, the ReLU and those statistics are ONE pass.
//           pre_k = identity, or relu?(x * scale_k + shift_k): an input that is the output of a Conv_Block (conv -> BN -> ReLU)
//           with no other consumer is normalised ON LOAD from the conv's raw output and never written (no bn_act_apply pass).
//           cat = the VIRTUAL concatenation of up to four tensors whose channel counts are not multiples of 8 (MultiResBlock's
//           cat[c3, c5, c7] = 5 + 10 + 16 channels at W = 32, unet_variants.py:95), normalised on the fly by the BatchNormalization
//           over it: the pieces' 8-channel groups are fetched with whole-group loads (coalesced) and laid into an LDS tile
//           [pixel][cat channel] element by element - the concatenation exists only there.
// backward  dx = [x > 0] * sc * (g - c1 - xhat c2), x = the BatchNormalization's input = the ReLU's output, written ONCE per distinct
//           destination (the add fans the same gradient out to its inputs); with a cat input the launch also delivers the two sums of
//           the BatchNormalization over the cat (sum dx, sum dx * xhat_cat) as partial rows, and seg_cat_bn_bwd_scatter turns
//           (dx, cat) into the gradients of the pieces (whole groups, padding channels zero) through the same LDS tile.
struct Aff4 { const float* sc[4]; const float* sh[4]; int relu[4]; };
struct CatDev {
  int n, ngroups, ctotal;
  const void* x[4];
  void* dx[4];
  int cs[4], C[4], coff[4], g0[4], acc[4];
};

// Lay the pieces' values for the pixels [p0, p0 + TP) into sm[TP][cs] at their channel positions of the concatenation (positions
// >= ctotal are never written: the caller zero-fills the tile once).  Four whole-group loads per thread in flight.
template <typename T>
__device__ __forceinline__ void cat_gather(const CatDev* cd, int64_t p0, int TP, int64_t npix, float* sm, int cs) {
  const int ng = cd->ngroups, total = TP * ng;
  for (int i0 = threadIdx.x; i0 < total; i0 += TPB * 4) {
    float f[4][8];
    int dst[4], lim[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * TPB;
      lim[u] = 0;
      if (i < total) {
        const int r = i / ng, g = i - r * ng;
        int k = 0;
        while (k + 1 < cd->n && g >= cd->g0[k + 1]) ++k;
        const int gg = g - cd->g0[k];
        const int64_t pix = p0 + r;
        if (pix < npix) {
          load8((const T*)cd->x[k] + pix * cd->cs[k] + gg * 8, f[u]);
        } else {
#pragma unroll
          for (int t = 0; t < 8; ++t) f[u][t] = 0.f;
        }
        dst[u] = r * cs + cd->coff[k] + gg * 8;
        lim[u] = min(8, cd->C[k] - gg * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (t < lim[u]) sm[dst[u] + t] = f[u][t];
  }
}

template <typename T, bool GATHER, int NIN>
__global__ __launch_bounds__(TPB) void tail_fwd_kernel(Ptr4<T> xs, Aff4 aff, int n, CatDev cdv, const float* cat_sc, const float* cat_sh, T* y,
                                                       int64_t npix, int cs, int c_valid, int act, float alpha, float* stats, int stats_ld,
                                                       int stats_coff, ColLayout L) {
  // (gathering form: two pixels per thread instead of four - 16 KB instead of 32 KB of LDS tile and ~60 registers less, so that
  // three or four blocks per CU overlap one block's gather phase with another's compute phase)
  constexpr int UN = GATHER ? 2 : UNR;
  __shared__ float sm[GATHER ? TPB * 8 * UN : TPB * 8];
  __shared__ CatDev scd;
  const int col = threadIdx.x % L.CHP, rowi = threadIdx.x / L.CHP;
  const int TP = L.RP * UN;
  const int ch = col;                       // (the host launches this kernel for cs <= 8 * TPB only: one column group)
  if (GATHER) {
    if (threadIdx.x == 0) scd = cdv;
    for (int i = threadIdx.x; i < TP * cs; i += TPB) sm[i] = 0.f;
  }
  float acc[2][8];
#pragma unroll
  for (int t = 0; t < 8; ++t) { acc[0][t] = 0.f; acc[1][t] = 0.f; }
  float sc[NIN][8], sh[NIN][8], csc[8], csh[8];       // (k loops fully unrolled with a guard: a run-time index would put these in scratch)
#pragma unroll
  for (int k = 0; k < NIN; ++k)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      sc[k][t] = (ch < L.CH && k < n && aff.sc[k] != nullptr) ? aff.sc[k][ch * 8 + t] : 1.f;
      sh[k][t] = (ch < L.CH && k < n && aff.sh[k] != nullptr) ? aff.sh[k][ch * 8 + t] : 0.f;
    }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    csc[t] = (GATHER && ch < L.CH && cat_sc != nullptr) ? cat_sc[ch * 8 + t] : 1.f;
    csh[t] = (GATHER && ch < L.CH && cat_sh != nullptr) ? cat_sh[ch * 8 + t] : 0.f;
  }
  for (int64_t p0 = (int64_t)blockIdx.x * TP; p0 < npix; p0 += (int64_t)TP * gridDim.x) {     // interleaved tiles of RP * UNR pixels
    if (GATHER) {
      __syncthreads();
      cat_gather<T>(&scd, p0, TP, npix, sm, cs);
      __syncthreads();
    }
    if (ch >= L.CH) continue;
    float f[NIN][UN][8];
#pragma unroll
    for (int k = 0; k < NIN; ++k)
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int64_t pix = p0 + rowi + (int64_t)u * L.RP;
        if (k < n && pix < npix) load8(xs.p[k] + pix * cs + ch * 8, f[k][u]);
      }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t pix = p0 + rowi + (int64_t)u * L.RP;
      if (pix >= npix) continue;
      float o[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) o[t] = GATHER ? sm[(rowi + u * L.RP) * cs + ch * 8 + t] * csc[t] + csh[t] : 0.f;
#pragma unroll
      for (int k = 0; k < NIN; ++k) {
        if (k >= n) continue;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          float v = f[k][u][t] * sc[k][t] + sh[k][t];
          if (aff.relu[k]) v = fmaxf(v, 0.f);
          o[t] += v;
        }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        o[t] = seg_act(o[t], act, alpha);
        if (ch * 8 + t >= c_valid) o[t] = 0.f;
      }
      if (y != nullptr) store8(y + pix * cs + ch * 8, o);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float v = stored<T>(o[t]);              // the statistics of what the BatchNormalization will read
        acc[0][t] += v;
        acc[1][t] += v * v;
      }
    }
  }
  __syncthreads();
  block_col_reduce<2>(acc, sm, rowi, col, L.CHP, L.RP);
  // (stats_ld / stats_coff: the rows of another tensor's statistics - a piece of a virtual concatenation writes its VALID channels into
  // the columns of the concatenation, the operand of the BatchNormalization over it)
  if (rowi == 0 && ch < L.CH) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (stats_coff >= 0 && ch * 8 + t >= c_valid) continue;      // (foreign rows: only this tensor's valid channels)
      const int sc0 = stats_coff >= 0 ? stats_coff : 0;
      stats[((size_t)blockIdx.x * 2 + 0) * stats_ld + sc0 + ch * 8 + t] = acc[0][t];
      stats[((size_t)blockIdx.x * 2 + 1) * stats_ld + sc0 + ch * 8 + t] = acc[1][t];
    }
  }
}

// LZ > 0: up to two inputs of the add are Conv_Blocks normalised on load (conv -> BN -> ReLU whose only consumer is this add):
// the gradient written here IS their dOut, so the two sums their own BatchNormalization backward needs - sum dOut act'(.),
// sum dOut act'(.) xhat_z - are accumulated in this launch from their raw conv outputs z (one more read each) instead of by a
// seg_bn_act_bwd_reduce pass per block (two reads each).
struct LazyBN { const void* z; const float *sc, *sh, *mu, *is; int relu; float* partial; };
struct LazyBN2 { LazyBN l[2]; };

template <typename T, bool GATHER, int LZ>
__global__ __launch_bounds__(TPB) void tail_bwd_fanout_kernel(const T* dy, const T* x, MPtr4<T> dxs, int n, int accmask, int64_t npix, int cs,
                                                              const float* scale, const float* mean, const float* invstd, const float* c1,
                                                              const float* c2, int relu_mask, CatDev cdv, const float* cat_mean,
                                                              const float* cat_invstd, float* cat_partial, LazyBN2 lz, ColLayout L) {
  constexpr int UN = LZ > 0 ? 2 : UNR;            // (the lazy sums keep 40 more registers per input: two pixels per thread in flight)
  __shared__ float sm[GATHER ? TPB * 8 * UN : (LZ > 0 ? TPB * 8 : 1)];
  __shared__ CatDev scd;
  const int col = threadIdx.x % L.CHP, rowi = threadIdx.x / L.CHP;
  const int TP = L.RP * UN;
  const int ch = col;
  const bool ok = ch < L.CH;
  if (GATHER) {
    if (threadIdx.x == 0) scd = cdv;
    for (int i = threadIdx.x; i < TP * cs; i += TPB) sm[i] = 0.f;
  }
  float sc[8], ka[8], kb[8], cmu[8], cis[8], acc[2][8];
  float lsc[LZ > 0 ? LZ : 1][8], lsh[LZ > 0 ? LZ : 1][8], lmu[LZ > 0 ? LZ : 1][8], lacc[LZ > 0 ? LZ : 1][2][8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int c = ch * 8 + t;
    sc[t] = ok ? scale[c] : 0.f;
    const float k = ok ? invstd[c] * c2[c] : 0.f;
    ka[t] = -sc[t] * k;
    kb[t] = ok ? sc[t] * (mean[c] * k - c1[c]) : 0.f;
    cmu[t] = (GATHER && ok) ? cat_mean[c] : 0.f;
    cis[t] = (GATHER && ok) ? cat_invstd[c] : 0.f;
    acc[0][t] = 0.f;
    acc[1][t] = 0.f;
#pragma unroll
    for (int j = 0; j < LZ; ++j) {
      lsc[j][t] = ok ? lz.l[j].sc[c] : 0.f;
      lsh[j][t] = ok ? lz.l[j].sh[c] : 0.f;
      lmu[j][t] = ok ? lz.l[j].mu[c] : 0.f;
      lacc[j][0][t] = 0.f;
      lacc[j][1][t] = 0.f;
    }
  }
  for (int64_t p0 = (int64_t)blockIdx.x * TP; p0 < npix; p0 += (int64_t)TP * gridDim.x) {
    if (GATHER) {
      __syncthreads();
      cat_gather<T>(&scd, p0, TP, npix, sm, cs);
      __syncthreads();
    }
    if (!ok) continue;
    float g[UN][8], xv[UN][8], zv[LZ > 0 ? LZ : 1][UN][8];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t pix = p0 + rowi + (int64_t)u * L.RP;
      if (pix < npix) {
        load8(dy + pix * cs + ch * 8, g[u]);
        load8(x + pix * cs + ch * 8, xv[u]);
#pragma unroll
        for (int j = 0; j < LZ; ++j) load8((const T*)lz.l[j].z + pix * cs + ch * 8, zv[j][u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t pix = p0 + rowi + (int64_t)u * L.RP;
      if (pix >= npix) continue;
      float o[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float v = g[u][t] * sc[t] + xv[u][t] * ka[t] + kb[t];
        o[t] = (relu_mask && !(xv[u][t] > 0.f)) ? 0.f : v;
      }
      if (GATHER || LZ > 0) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float gg = stored<T>(o[t]);           // the sums of what the consumers of this gradient will read back
          if (GATHER) {
            acc[0][t] += gg;
            acc[1][t] += gg * (sm[(rowi + u * L.RP) * cs + ch * 8 + t] - cmu[t]) * cis[t];
          }
#pragma unroll
          for (int j = 0; j < LZ; ++j) {
            const float pre = zv[j][u][t] * lsc[j][t] + lsh[j][t];
            const float gm = (lz.l[j].relu && !(pre > 0.f)) ? 0.f : gg;
            lacc[j][0][t] += gm;
            lacc[j][1][t] += gm * (zv[j][u][t] - lmu[j][t]);        // (x invstd once, at the end)
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k >= n || dxs.p[k] == nullptr) continue;
        float w[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) w[t] = o[t];
        if (accmask & (1 << k)) {
          float old[8];
          load8(dxs.p[k] + pix * cs + ch * 8, old);
#pragma unroll
          for (int t = 0; t < 8; ++t) w[t] += old[t];
        }
        store8(dxs.p[k] + pix * cs + ch * 8, w);
      }
    }
  }
  if (GATHER) {
    __syncthreads();
    block_col_reduce<2>(acc, sm, rowi, col, L.CHP, L.RP);
    if (rowi == 0 && ok) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        cat_partial[((size_t)blockIdx.x * 2 + 0) * cs + ch * 8 + t] = acc[0][t];
        cat_partial[((size_t)blockIdx.x * 2 + 1) * cs + ch * 8 + t] = acc[1][t];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < LZ; ++j) {
    __syncthreads();
    block_col_reduce<2>(lacc[j], sm, rowi, col, L.CHP, L.RP);
    if (rowi == 0 && ok) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        lz.l[j].partial[((size_t)blockIdx.x * 2 + 0) * cs + ch * 8 + t] = lacc[j][0][t];
        lz.l[j].partial[((size_t)blockIdx.x * 2 + 1) * cs + ch * 8 + t] = lacc[j][1][t] * lz.l[j].is[ch * 8 + t];
      }
    }
  }
}

// d cat = sc (g - c1 - xhat c2) of the BatchNormalization over the virtual concatenation, written to the pieces' gradients
template <typename T>
__global__ __launch_bounds__(TPB) void cat_bn_bwd_scatter_kernel(const T* g, CatDev cdv, int64_t npix, int cs, const float* scale, const float* mean,
                                                                 const float* invstd, const float* c1, const float* c2, ColLayout L) {
  __shared__ float sm[TPB * 8 * UNR];
  __shared__ CatDev scd;
  const int col = threadIdx.x % L.CHP, rowi = threadIdx.x / L.CHP;
  const int TP = L.RP * UNR;
  const int ch = col;
  if (threadIdx.x == 0) scd = cdv;
  for (int i = threadIdx.x; i < TP * cs; i += TPB) sm[i] = 0.f;
  float sc[8], ka[8], kb[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int c = ch * 8 + t;
    const bool ok = ch < L.CH && c < cdv.ctotal;
    sc[t] = ok ? scale[c] : 0.f;
    const float k = ok ? invstd[c] * c2[c] : 0.f;
    ka[t] = -sc[t] * k;
    kb[t] = ok ? sc[t] * (mean[c] * k - c1[c]) : 0.f;
  }
  for (int64_t p0 = (int64_t)blockIdx.x * TP; p0 < npix; p0 += (int64_t)TP * gridDim.x) {
    __syncthreads();
    cat_gather<T>(&scd, p0, TP, npix, sm, cs);
    __syncthreads();
    if (ch < L.CH) {
      float gv[UNR][8];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int64_t pix = p0 + rowi + (int64_t)u * L.RP;
        if (pix < npix) load8(g + pix * cs + ch * 8, gv[u]);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int64_t pix = p0 + rowi + (int64_t)u * L.RP;
        float* cell = sm + (rowi + u * L.RP) * cs + ch * 8;        // (this thread's own cells: read the cat value, leave d cat)
#pragma unroll
        for (int t = 0; t < 8; ++t) cell[t] = pix < npix ? gv[u][t] * sc[t] + cell[t] * ka[t] + kb[t] : 0.f;
      }
    }
    __syncthreads();
    const int ng = scd.ngroups, total = TP * ng;
    for (int i = threadIdx.x; i < total; i += TPB) {
      const int r = i / ng, gi = i - r * ng;
      int k = 0;
      while (k + 1 < scd.n && gi >= scd.g0[k + 1]) ++k;
      const int gg = gi - scd.g0[k];
      const int64_t pix = p0 + r;
      if (pix >= npix || scd.dx[k] == nullptr) continue;
      const int lim = scd.C[k] - gg * 8;
      const float* src = sm + r * cs + scd.coff[k] + gg * 8;
      float o[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) o[t] = t < lim ? src[t] : 0.f;
      T* dst = (T*)scd.dx[k] + pix * scd.cs[k] + gg * 8;
      if (scd.acc[k]) {
        float old[8];
        load8(dst, old);
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] += old[t];
      }
      store8(dst, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(TPB) void mul_bcast_fwd_kernel(const T* a, const T* r, T* y, int64_t npix, int cs, int r_cs) {
  const int CH = cs / 8;
  const int64_t items = npix * CH;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < items; it += (int64_t)gridDim.x * TPB) {
    const int64_t pix = it / CH;
    const float rv = load1(r + pix * r_cs, 0);
    float f[8];
    load8(a + it * 8, f);
#pragma unroll
    for (int t = 0; t < 8; ++t) f[t] *= rv;
    store8(y + it * 8, f);
  }
}

// one group of LPP lanes per pixel (LPP = power of two <= 64); dr needs the sum over all channels of the pixel
template <typename T>
__global__ __launch_bounds__(TPB) void mul_bcast_bwd_kernel(const T* dy, const T* a, const T* r, T* da, T* dr,
                                                            int64_t npix, int cs, int r_cs, int acc_a, int acc_r, int LPP) {
  const int CH = cs / 8;
  const int gpb = TPB / LPP;
  const int sub = threadIdx.x % LPP, grp = threadIdx.x / LPP;
  for (int64_t pix0 = (int64_t)blockIdx.x * gpb; pix0 < npix; pix0 += (int64_t)gridDim.x * gpb) {
    const int64_t pix = pix0 + grp;
    const bool ok = pix < npix;
    float dot = 0.f;
    const float rv = ok ? load1(r + pix * r_cs, 0) : 0.f;
    for (int ch = sub; ch < CH; ch += LPP) {
      if (!ok) break;
      float g[8], av[8], o[8];
      load8(dy + pix * cs + ch * 8, g);
      load8(a + pix * cs + ch * 8, av);
#pragma unroll
      for (int t = 0; t < 8; ++t) { dot += g[t] * av[t]; o[t] = g[t] * rv; }
      if (da != nullptr) {
        T* dst = da + pix * cs + ch * 8;
        if (acc_a) {
          float old[8];
          load8(dst, old);
#pragma unroll
          for (int t = 0; t < 8; ++t) o[t] += old[t];
        }
        store8(dst, o);
      }
    }
    for (int o = LPP >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    if (ok && sub == 0 && dr != nullptr) {
      float f[8] = {dot, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      T* dst = dr + pix * r_cs;
      if (acc_r) f[0] += load1(dst, 0);
      if (r_cs >= 8) store8(dst, f);
      else store1(dst, 0, f[0]);           // (fp16 only: a split tensor always has whole 8-channel groups, host-checked)
    }
  }
}

// ------------------------------------------------------------------------------------------------ ConvLSTM (T = 1) gates
__device__ __forceinline__ float hsig(float v) { return fminf(fmaxf(0.2f * v + 0.5f, 0.f), 1.f); }
// TF clip_by_value (minimum/maximum) passes the gradient AT the knees, so the interval is closed
__device__ __forceinline__ float hsig_grad(float v) { return (v >= -2.5f && v <= 2.5f) ? 0.2f : 0.f; }

template <typename T>
__global__ __launch_bounds__(TPB) void convlstm_gate_fwd_kernel(const T* z, int z_cs, T* h, int h_cs, int64_t npix, int F) {
  const int CH = F / 8;
  const int64_t items = npix * CH;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < items; it += (int64_t)gridDim.x * TPB) {
    const int64_t pix = it / CH;
    const int ch = (int)(it - pix * CH);
    float zi[8], zc[8], zo[8], o[8];
    const T* zp = z + pix * z_cs + ch * 8;
    load8(zp, zi);
    load8(zp + 2 * F, zc);
    load8(zp + 3 * F, zo);
#pragma unroll
    for (int t = 0; t < 8; ++t) o[t] = hsig(zo[t]) * tanhf(hsig(zi[t]) * tanhf(zc[t]));
    store8(h + pix * h_cs + ch * 8, o);
  }
}

template <typename T>
__global__ __launch_bounds__(TPB) void convlstm_gate_bwd_kernel(const T* dh, int h_cs, const T* z, int z_cs, T* dz,
                                                                int64_t npix, int F) {
  const int CH = F / 8;
  const int64_t items = npix * CH;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < items; it += (int64_t)gridDim.x * TPB) {
    const int64_t pix = it / CH;
    const int ch = (int)(it - pix * CH);
    float zi[8], zc[8], zo[8], g[8], di[8], dc[8], dO[8], zero[8];
    const T* zp = z + pix * z_cs + ch * 8;
    load8(zp, zi);
    load8(zp + 2 * F, zc);
    load8(zp + 3 * F, zo);
    load8(dh + pix * h_cs + ch * 8, g);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float a = hsig(zi[t]), b = tanhf(zc[t]), u = a * b, tu = tanhf(u), o = hsig(zo[t]);
      dO[t] = g[t] * tu * hsig_grad(zo[t]);
      const float du = g[t] * o * (1.f - tu * tu);
      di[t] = du * b * hsig_grad(zi[t]);
      dc[t] = du * a * (1.f - b * b);
      zero[t] = 0.f;
    }
    T* dp = dz + pix * z_cs + ch * 8;
    store8(dp, di);
    store8(dp + F, zero);
    store8(dp + 2 * F, dc);
    store8(dp + 3 * F, dO);
  }
}

// ------------------------------------------------------------------------------------------------ 1x1 heads (Cout <= 8)
constexpr int HEAD_MAXC = 8;

// element offset of input pixel (n, yo*sy, xo*sx) for output pixel index pix; the full-resolution heads (stride 1)
// skip the three integer divisions, the strided ones (UNet 3+ deep supervision) use 32-bit ones (npix < 2^31, host-checked)
__device__ __forceinline__ int64_t head_in_off(int64_t pix, int H, int W, int HO, int WO, int sy, int sx, int cs) {
  if (sy == 1 && sx == 1) return pix * cs;
  const unsigned pp = (unsigned)pix;
  const unsigned r = pp / (unsigned)WO, xo = pp - r * (unsigned)WO;
  const unsigned n = r / (unsigned)HO, yo = r - n * (unsigned)HO;
  return (((int64_t)n * H + yo * sy) * W + xo * sx) * cs;
}

template <typename T, int CO>
__global__ __launch_bounds__(TPB) void head_fwd_kernel(const T* x, int cs, int Cin, int N, int H, int W, int sy, int sx,
                                                       const float* w, int w_ld, const float* b, int Cout, int act, float* y, int LPP) {
  const int CH = (Cin + 7) / 8;
  const int HO = (H + sy - 1) / sy, WO = (W + sx - 1) / sx;
  const int64_t npix = (int64_t)N * HO * WO;
  const int gpb = TPB / LPP;
  const int sub = threadIdx.x % LPP, grp = threadIdx.x / LPP;
  const bool one = CH <= LPP;            // one chunk per lane: keep its weights in registers
  float wr[CO][8];
#pragma unroll
  for (int k = 0; k < CO; ++k)
#pragma unroll
    for (int t = 0; t < 8; ++t) wr[k][t] = (one && k < Cout && sub * 8 + t < Cin) ? w[k * w_ld + sub * 8 + t] : 0.f;
  // a block owns chunks of gpb * UNR consecutive pixels, interleaved over the grid (neighbouring blocks stream
  // neighbouring addresses); pixel u of a trip is gpb further on
  const int64_t pstride = gpb;
  for (int64_t pb = (int64_t)blockIdx.x * gpb * UNR; pb < npix; pb += (int64_t)gridDim.x * gpb * UNR) {
    float acc[UNR][CO];
    float f[UNR][8];
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t pix = pb + u * pstride + grp;
      ok[u] = pix < npix;
#pragma unroll
      for (int k = 0; k < CO; ++k) acc[u][k] = 0.f;
      if (ok[u] && one && sub < CH) {
        const int64_t ioff = head_in_off(pix, H, W, HO, WO, sy, sx, cs);
        load8(x + ioff + sub * 8, f[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t pix = pb + u * pstride + grp;
      if (ok[u]) {
        if (one) {
          if (sub < CH) {
#pragma unroll
            for (int k = 0; k < CO; ++k)
              if (k < Cout) {
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[u][k] += f[u][t] * wr[k][t];
              }
          }
        } else {
          const int64_t ioff = head_in_off(pix, H, W, HO, WO, sy, sx, cs);
          const T* xp = x + ioff;
          for (int ch = sub; ch < CH; ch += LPP) {
            float g[8];
            load8(xp + ch * 8, g);
#pragma unroll
            for (int k = 0; k < CO; ++k)
              if (k < Cout) {
#pragma unroll
                for (int t = 0; t < 8; ++t)
                  if (ch * 8 + t < Cin) acc[u][k] += g[t] * w[k * w_ld + ch * 8 + t];
              }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < CO; ++k)
        for (int o = LPP >> 1; o > 0; o >>= 1) acc[u][k] += __shfl_xor(acc[u][k], o);
      if (ok[u] && sub == 0) {
        float v[CO];
#pragma unroll
        for (int k = 0; k < CO; ++k) v[k] = (k < Cout) ? acc[u][k] + (b ? b[k] : 0.f) : -3.0e38f;
        if (act == SEG_ACT_SOFTMAX) {
          float mx = v[0];
#pragma unroll
          for (int k = 1; k < CO; ++k) mx = fmaxf(mx, v[k]);
          float sum = 0.f;
#pragma unroll
          for (int k = 0; k < CO; ++k) { v[k] = (k < Cout) ? __expf(v[k] - mx) : 0.f; sum += v[k]; }
          const float inv = 1.f / sum;
#pragma unroll
          for (int k = 0; k < CO; ++k) v[k] *= inv;
        } else {
#pragma unroll
          for (int k = 0; k < CO; ++k) v[k] = seg_act(v[k], act, 0.f);
        }
#pragma unroll
        for (int k = 0; k < CO; ++k)
          if (k < Cout) y[pix * Cout + k] = v[k];
      }
    }
  }
}

// gradient of the pre-activation given dy (wrt activated y) and y
template <int CO>
__device__ __forceinline__ void head_dpre(const float* dy, const float* y, int Cout, int act, float* d) {
  if (act == SEG_ACT_SOFTMAX) {
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < CO; ++k)
      if (k < Cout) dot += dy[k] * y[k];
#pragma unroll
    for (int k = 0; k < CO; ++k) d[k] = (k < Cout) ? y[k] * (dy[k] - dot) : 0.f;
  } else {
#pragma unroll
    for (int k = 0; k < CO; ++k) d[k] = (k < Cout) ? dy[k] * seg_act_grad_from_out(y[k], act, 0.f) : 0.f;
  }
}

// dx (+)= gs * dpre . w      (one LPP-lane group per pixel, like the forward)
template <typename T, int CO>
__global__ __launch_bounds__(TPB) void head_bwd_dx_kernel(const float* dy, const float* y, int cs, int Cin, int N, int H, int W, int sy,
                                                          int sx, const float* w, int w_ld, int Cout, int act, T* dx,
                                                          int accumulate, const float* gs_ptr, int LPP) {
  const int CH = cs / 8;
  const int HO = (H + sy - 1) / sy, WO = (W + sx - 1) / sx;
  const int64_t npix = (int64_t)N * HO * WO;
  const int gpb = TPB / LPP;
  const int sub = threadIdx.x % LPP, grp = threadIdx.x / LPP;
  const float gs = gs_ptr ? *gs_ptr : 1.f;
  const bool one = CH <= LPP;
  float wr[CO][8];
#pragma unroll
  for (int k = 0; k < CO; ++k)
#pragma unroll
    for (int t = 0; t < 8; ++t) wr[k][t] = (one && k < Cout && sub * 8 + t < Cin) ? w[k * w_ld + sub * 8 + t] * gs : 0.f;
  // a block owns chunks of gpb * UNR consecutive pixels, interleaved over the grid (neighbouring blocks stream
  // neighbouring addresses); pixel u of a trip is gpb further on
  const int64_t pstride = gpb;
  for (int64_t pb = (int64_t)blockIdx.x * gpb * UNR; pb < npix; pb += (int64_t)gridDim.x * gpb * UNR) {
    float d[UNR][CO], old[UNR][8];
    T* dp[UNR];
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t pix = pb + u * pstride + grp;
      ok[u] = pix < npix;
      dp[u] = dx;
      if (ok[u]) {
        float dyv[CO], yv[CO];
#pragma unroll
        for (int k = 0; k < CO; ++k) {
          dyv[k] = (k < Cout) ? dy[pix * Cout + k] : 0.f;
          yv[k] = (k < Cout) ? y[pix * Cout + k] : 0.f;
        }
        head_dpre<CO>(dyv, yv, Cout, act, d[u]);
        const int64_t ioff = head_in_off(pix, H, W, HO, WO, sy, sx, cs);
        dp[u] = dx + ioff;
        if (accumulate && one && sub < CH) load8(dp[u] + sub * 8, old[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (!ok[u]) continue;
      if (one) {
        if (sub < CH) {
          float o[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            float sacc = 0.f;
#pragma unroll
            for (int k = 0; k < CO; ++k)
              if (k < Cout) sacc += d[u][k] * wr[k][t];
            o[t] = accumulate ? sacc + old[u][t] : sacc;
          }
          store8(dp[u] + sub * 8, o);
        }
      } else {
        for (int ch = sub; ch < CH; ch += LPP) {
          float o[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            float sacc = 0.f;
            const int c = ch * 8 + t;
            if (c < Cin) {
#pragma unroll
              for (int k = 0; k < CO; ++k)
                if (k < Cout) sacc += d[u][k] * w[k * w_ld + c];
            }
            o[t] = sacc * gs;
          }
          if (accumulate) {
            float oo[8];
            load8(dp[u] + ch * 8, oo);
#pragma unroll
            for (int t = 0; t < 8; ++t) o[t] += oo[t];
          }
          store8(dp[u] + ch * 8, o);
        }
      }
    }
  }
}

// dw[k][c] += sum_pix dpre[k] * x[c],  db[k] += sum_pix dpre[k]   ((rows x chunk-columns) layout + atomics)
template <typename T, int CO>
__global__ __launch_bounds__(TPB) void head_bwd_dw_kernel(const float* dy, const float* y, const T* x, int cs, int Cin, int N, int H,
                                                          int W, int sy, int sx, int w_ld, int Cout, int act, float* dw, float* db,
                                                          float* work, ColLayout L) {
  __shared__ float sm[TPB * 8];
  const int col = threadIdx.x % L.CHP, rowi = threadIdx.x / L.CHP;
  const int HO = (H + sy - 1) / sy, WO = (W + sx - 1) / sx;
  const int64_t npix = (int64_t)N * HO * WO;
  const int64_t p1 = npix;
  for (int grp = 0; grp < L.ngrp; ++grp) {
    const int ch = grp * L.CHP + col;
    float acc[CO][8];
    float accb[CO];
#pragma unroll
    for (int k = 0; k < CO; ++k) {
      accb[k] = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[k][t] = 0.f;
    }
    if (ch < L.CH) {
      // UNR pixels per trip, every load issued before the first use (a block walks thousands of pixels: one load
      // in flight per thread leaves the kernel latency-bound)
      for (int64_t pb = (int64_t)blockIdx.x * L.RP * UNR + rowi; pb < p1; pb += (int64_t)gridDim.x * L.RP * UNR) {
        float dyv[UNR][CO], yv[UNR][CO], f[UNR][8];
        bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int64_t pix = pb + (int64_t)u * L.RP;
          ok[u] = pix < p1;
#pragma unroll
          for (int k = 0; k < CO; ++k) {
            dyv[u][k] = (ok[u] && k < Cout) ? dy[pix * Cout + k] : 0.f;
            yv[u][k] = (ok[u] && k < Cout) ? y[pix * Cout + k] : 0.f;
          }
#pragma unroll
          for (int t = 0; t < 8; ++t) f[u][t] = 0.f;
          if (ok[u]) {
            const int64_t ioff = head_in_off(pix, H, W, HO, WO, sy, sx, cs);
            load8(x + ioff + ch * 8, f[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          float d[CO];
          head_dpre<CO>(dyv[u], yv[u], Cout, act, d);
#pragma unroll
          for (int k = 0; k < CO; ++k)
            if (k < Cout) {
              accb[k] += d[k];
#pragma unroll
              for (int t = 0; t < 8; ++t) acc[k][t] += d[k] * f[u][t];
            }
        }
      }
    }
    block_col_reduce<CO>(acc, sm, rowi, col, L.CHP, L.RP);
    if (rowi == 0 && ch < L.CH) {
      // with a workspace every block stores its partial row (summed by head_bwd_dw_finalize_kernel); without one the
      // Cout x Cin sums of all blocks meet in the same few cache lines as atomics (~4 ns each, serialised)
#pragma unroll
      for (int k = 0; k < CO; ++k)
        if (k < Cout) {
#pragma unroll
          for (int t = 0; t < 8; ++t)
            if (ch * 8 + t < Cin) {
              if (work != nullptr) work[((size_t)blockIdx.x * Cout + k) * w_ld + ch * 8 + t] = acc[k][t];
              else atomicAdd(dw + k * w_ld + ch * 8 + t, acc[k][t]);
            }
        }
    }
    if (grp == 0 && db != nullptr) {   // bias: column 0 threads of every row-group
      __syncthreads();
      if (col == 0) {
#pragma unroll
        for (int k = 0; k < CO; ++k) sm[rowi * CO + k] = accb[k];
      }
      __syncthreads();
      if (threadIdx.x < Cout) {
        float s = 0.f;
        for (int r = 0; r < L.RP; ++r) s += sm[r * CO + threadIdx.x];
        if (work != nullptr) work[(size_t)gridDim.x * Cout * w_ld + (size_t)blockIdx.x * Cout + threadIdx.x] = s;
        else atomicAdd(db + threadIdx.x, s);
      }
    }
  }
}

// dw[k][c] += sum_b work[b][k][c],  db[k] += sum_b workb[b][k]: 16 outputs x 16 partial lanes per block, the lanes of
// an output stride through the blocks' rows and meet in LDS
__global__ __launch_bounds__(TPB) void head_bwd_dw_finalize_kernel(const float* work, int nblk, int Cin, int w_ld, int Cout, float* dw,
                                                                   float* db) {
  __shared__ float sm[16][17];
  const int o = blockIdx.x * 16 + (threadIdx.x & 15), pl = threadIdx.x >> 4;
  const int nw = Cout * w_ld;
  float s = 0.f;
  if (o < nw) {
    for (int b = pl; b < nblk; b += 16) s += work[(size_t)b * nw + o];
  } else if (o < nw + Cout) {
    for (int b = pl; b < nblk; b += 16) s += work[(size_t)nblk * nw + (size_t)b * Cout + (o - nw)];
  }
  sm[pl][threadIdx.x & 15] = s;
  __syncthreads();
  if (pl != 0) return;
  float t = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) t += sm[j][threadIdx.x];
  if (o < nw) {
    if (o % w_ld < Cin) dw[o] += t;
  } else if (o < nw + Cout && db != nullptr) {
    db[o - nw] += t;
  }
}

// ------------------------------------------------------------------------------------------------ wide output heads
// Heads with more than 8 output channels run their 1x1 conv on the implicit-GEMM path (logits in the storage type);
// these two kernels are the fp32 activation on top: y = act(logits) and d logits = gs * act'(y) dy (padding channels 0).
template <typename T>
__global__ __launch_bounds__(TPB) void head_act_fwd_kernel(const T* x, int cs, int64_t npix, int C, int act, float* y) {
  for (int64_t pix = (int64_t)blockIdx.x * TPB + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * TPB) {
    const T* xp = x + pix * cs;
    float* yp = y + pix * C;
    if (act == SEG_ACT_SOFTMAX) {
      float mx = -3.4e38f;
      for (int k = 0; k < C; ++k) mx = fmaxf(mx, load_elem(xp, k));
      float s = 0.f;
      for (int k = 0; k < C; ++k) s += expf(load_elem(xp, k) - mx);
      const float inv = 1.f / s;
      for (int k = 0; k < C; ++k) yp[k] = expf(load_elem(xp, k) - mx) * inv;
    } else {
      for (int k = 0; k < C; ++k) yp[k] = seg_act(load_elem(xp, k), act, 0.f);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(TPB) void head_act_bwd_kernel(const float* dy, const float* y, int64_t npix, int C, int act, const float* gs_ptr,
                                                           T* dx, int cs, int accumulate) {
  const float gs = gs_ptr ? gs_ptr[0] : 1.f;
  for (int64_t pix = (int64_t)blockIdx.x * TPB + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * TPB) {
    const float* dp = dy + pix * C;
    const float* yp = y + pix * C;
    T* xp = dx + pix * cs;
    float dot = 0.f;
    if (act == SEG_ACT_SOFTMAX)
      for (int k = 0; k < C; ++k) dot += dp[k] * yp[k];
    for (int k = 0; k < cs; ++k) {
      float d = 0.f;
      if (k < C) d = (act == SEG_ACT_SOFTMAX ? yp[k] * (dp[k] - dot) : dp[k] * seg_act_grad_from_out(yp[k], act, 0.f)) * gs;
      store_elem(xp, k, accumulate ? load_elem(xp, k) + d : d);
    }
  }
}

// ------------------------------------------------------------------------------------------------ losses
// The 16 names of utils/tf_losses.py:4-44 with the Keras 2.x formulas (keras/losses.py, keras/backend.py; eps = 1e-7).
// Element-wise kinds average f(pred, target) over every element; per-pixel kinds reduce over the channel axis first and
// average over pixels.  dpred = weight * d loss / d pred, loss_out[0] += weight * loss.
__device__ __forceinline__ bool loss_per_pixel(int kind) {
  return kind == SEG_LOSS_CCE || kind == SEG_LOSS_KLD || kind == SEG_LOSS_CATHINGE || kind == SEG_LOSS_COSINE || kind == SEG_LOSS_SPARSE_CCE;
}

__device__ __forceinline__ float loss_elem(int kind, float p, float t, bool pm1, float& g) {
  const float EPS = 1e-7f;
  switch (kind) {
    case SEG_LOSS_MSE: { const float d = p - t; g = 2.f * d; return d * d; }
    case SEG_LOSS_BCE: {   // probabilities clipped to [eps, 1 - eps], zero gradient outside
      const float pc = fminf(fmaxf(p, EPS), 1.f - EPS);
      const float inside = (p > EPS && p < 1.f - EPS) ? 1.f : 0.f;
      g = inside * (-(t / pc) + (1.f - t) / (1.f - pc));
      return -(t * __logf(pc) + (1.f - t) * __logf(1.f - pc));
    }
    case SEG_LOSS_MAE: { const float d = p - t; g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); return fabsf(d); }
    case SEG_LOSS_MAPE: {  // 100 |(t - p) / max(|t|, eps)|
      const float den = fmaxf(fabsf(t), EPS), d = p - t;
      g = (d > 0.f ? 100.f : (d < 0.f ? -100.f : 0.f)) / den;
      return 100.f * fabsf(d) / den;
    }
    case SEG_LOSS_MSLE: {  // (log(max(p, eps) + 1) - log(max(t, eps) + 1))^2
      const float pc = fmaxf(p, EPS), d = log1pf(pc) - log1pf(fmaxf(t, EPS));
      g = p >= EPS ? 2.f * d / (pc + 1.f) : 0.f;
      return d * d;
    }
    case SEG_LOSS_HUBER: { // delta = 1
      const float d = p - t, ad = fabsf(d);
      g = ad <= 1.f ? d : (d > 0.f ? 1.f : -1.f);
      return ad <= 1.f ? 0.5f * d * d : ad - 0.5f;
    }
    case SEG_LOSS_LOGCOSH: {  // x + softplus(-2x) - log 2
      const float x = p - t, z = -2.f * x;
      g = tanhf(x);
      return x + fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))) - 0.69314718056f;
    }
    case SEG_LOSS_POISSON: { g = 1.f - t / (p + EPS); return p - t * logf(p + EPS); }
    case SEG_LOSS_HINGE: case SEG_LOSS_SQHINGE: {   // labels in {0,1} are read as {-1,+1} (keras _maybe_convert_labels)
      const float tt = pm1 ? 2.f * t - 1.f : t, m = fmaxf(1.f - tt * p, 0.f);
      if (kind == SEG_LOSS_HINGE) { g = m > 0.f ? -tt : 0.f; return m; }
      g = -2.f * tt * m;
      return m * m;
    }
    default: {             // SEG_LOSS_FOCAL: (1 - p_t)^2 * bce, p_t = t p + (1 - t)(1 - p) on the unclipped prediction
      const float pc = fminf(fmaxf(p, EPS), 1.f - EPS);
      const float inside = (p > EPS && p < 1.f - EPS) ? 1.f : 0.f;
      const float bce = -(t * logf(pc) + (1.f - t) * logf(1.f - pc));
      const float dbce = inside * (-(t / pc) + (1.f - t) / (1.f - pc));
      const float q = 1.f - (t * p + (1.f - t) * (1.f - p));
      g = q * q * dbce - 2.f * q * (2.f * t - 1.f) * bce;
      return q * q * bce;
    }
  }
}

__global__ __launch_bounds__(TPB) void loss_kernel(int kind, const float* pred, const float* target, int64_t npix, int C, float inv_n,
                                                   float weight, const int* binary_flag, float* loss_out, float* dpred) {
  const float EPS = 1e-7f;
  const float gw = inv_n * weight;
  float local = 0.f;
  if (loss_per_pixel(kind)) {
    for (int64_t pix = (int64_t)blockIdx.x * TPB + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * TPB) {
      const float* p = pred + pix * C;
      float scratch[1];
      float* d = dpred ? dpred + pix * C : scratch;
      const int ds = dpred ? 1 : 0;          // metric evaluation (dpred == NULL): every gradient store lands in scratch[0]
      if (kind == SEG_LOSS_CCE) {
        const float* t = target + pix * C;
        float s = 0.f;
        for (int k = 0; k < C; ++k) s += p[k];
        const float invs = 1.f / s;
        // L = -sum_k t_k log(clip(p_k / s));  dL/dp_j = -t_j m_j / p_j + sum_k t_k m_k / s   (m = 1 inside the clip range)
        float corr = 0.f;
        for (int k = 0; k < C; ++k) {
          const float q = p[k] * invs;
          const float m = (q > EPS && q < 1.f - EPS) ? 1.f : 0.f;
          corr += t[k] * m;
          local -= t[k] * __logf(fminf(fmaxf(q, EPS), 1.f - EPS));
        }
        for (int k = 0; k < C; ++k) {
          const float q = p[k] * invs;
          const float m = (q > EPS && q < 1.f - EPS) ? 1.f : 0.f;
          d[k * ds] = ((m > 0.f ? -t[k] / p[k] : 0.f) + corr * invs) * gw;
        }
      } else if (kind == SEG_LOSS_SPARSE_CCE) {
        // target: one class index per pixel.  q = clip(p, eps, 1 - eps); L = -log(q_c / sum q)  (the clipped
        // probabilities go through log and a softmax cross-entropy, which renormalises them)
        const int c = (int)target[pix];
        float s = 0.f;
        for (int k = 0; k < C; ++k) s += fminf(fmaxf(p[k], EPS), 1.f - EPS);
        const bool valid = c >= 0 && c < C;
        const float qc = valid ? fminf(fmaxf(p[c], EPS), 1.f - EPS) : 1.f;
        if (valid) local += logf(s) - logf(qc);
        for (int k = 0; k < C; ++k) {
          const float m = (p[k] > EPS && p[k] < 1.f - EPS) ? 1.f : 0.f;
          d[k * ds] = valid ? m * (1.f / s - (k == c ? 1.f / qc : 0.f)) * gw : 0.f;
        }
      } else if (kind == SEG_LOSS_KLD) {
        // sum_k tc_k log(tc_k / pc_k), both clipped to [eps, 1]
        const float* t = target + pix * C;
        for (int k = 0; k < C; ++k) {
          const float tc = fminf(fmaxf(t[k], EPS), 1.f), pc = fminf(fmaxf(p[k], EPS), 1.f);
          local += tc * logf(tc / pc);
          d[k * ds] = (p[k] >= EPS && p[k] <= 1.f) ? -tc / pc * gw : 0.f;
        }
      } else if (kind == SEG_LOSS_CATHINGE) {
        // max(0, max_k((1 - t_k) p_k) - sum_k t_k p_k + 1)
        const float* t = target + pix * C;
        float pos = 0.f, neg = -3.4e38f;
        int arg = 0;
        for (int k = 0; k < C; ++k) {
          pos += t[k] * p[k];
          const float v = (1.f - t[k]) * p[k];
          if (v > neg) { neg = v; arg = k; }
        }
        const float l = neg - pos + 1.f;
        if (l > 0.f) local += l;
        for (int k = 0; k < C; ++k) d[k * ds] = l > 0.f ? (-t[k] + (k == arg ? 1.f - t[k] : 0.f)) * gw : 0.f;
      } else {   // SEG_LOSS_COSINE: -sum_k l2n(t)_k l2n(p)_k, l2n(x) = x rsqrt(max(sum x^2, 1e-12))
        const float* t = target + pix * C;
        float pp = 0.f, tt = 0.f, pt = 0.f;
        for (int k = 0; k < C; ++k) { pp += p[k] * p[k]; tt += t[k] * t[k]; pt += p[k] * t[k]; }
        const float rp = rsqrtf(fmaxf(pp, 1e-12f)), rt = rsqrtf(fmaxf(tt, 1e-12f));
        local -= pt * rp * rt;
        // d/dp_j of (p.v) rp, v = t rt: v_j rp - (p.v) p_j rp^3 when sum p^2 is above the floor, else v_j rp
        const float cub = pp > 1e-12f ? pt * rt * rp * rp * rp : 0.f;
        for (int k = 0; k < C; ++k) d[k * ds] = -(t[k] * rt * rp - cub * p[k]) * gw;
      }
    }
  } else {
    const int64_t n = npix * C;
    const bool pm1 = binary_flag ? (binary_flag[0] != 0) : true;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
      float g;
      local += loss_elem(kind, pred[i], target[i], pm1, g);
      if (dpred) dpred[i] = g * gw;
    }
  }
  local = wave_sum(local);
  __shared__ float sm[TPB / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int k = 0; k < TPB / 64; ++k) s += sm[k];
    atomicAdd(loss_out, s * gw);
  }
}

// counts[0..3] += TP, FP, FN, TN with "pred > thr" (Keras BinaryAccuracy / Precision / Recall / TruePositives ...),
// counts[4..7] += the same with "pred >= thr" (Keras BinaryIoU); a label counts as positive when it is > 0.5
__global__ __launch_bounds__(TPB) void binary_counts_kernel(const float* pred, const float* target, int64_t n, float thr,
                                                            unsigned long long* counts) {
  unsigned c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const float p = pred[i];
    const bool t = target[i] > 0.5f;
    const bool a = p > thr, b = p >= thr;
    c[0] += (a && t);  c[1] += (a && !t);  c[2] += (!a && t);  c[3] += (!a && !t);      // TP FP FN TN
    c[4] += (b && t);  c[5] += (b && !t);  c[6] += (!b && t);  c[7] += (!b && !t);
  }
  __shared__ unsigned sm[8];
  if (threadIdx.x < 8) sm[threadIdx.x] = 0;
  __syncthreads();
  for (int k = 0; k < 8; ++k) {
    unsigned v = c[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sm[k], v);
  }
  __syncthreads();
  if (threadIdx.x < 8 && sm[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)sm[threadIdx.x]);
}

// counts[0] += pixels whose arg-max channel of pred equals that of target (first maximum wins, as tf.argmax)
__global__ __launch_bounds__(TPB) void argmax_match_kernel(const float* pred, const float* target, int64_t npix, int C, int sparse,
                                                           unsigned long long* counts) {
  unsigned c = 0;
  for (int64_t pix = (int64_t)blockIdx.x * TPB + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * TPB) {
    int ap = 0, at = 0;
    for (int k = 1; k < C; ++k) if (pred[pix * C + k] > pred[pix * C + ap]) ap = k;
    if (sparse) at = (int)target[pix];
    else for (int k = 1; k < C; ++k) if (target[pix * C + k] > target[pix * C + at]) at = k;
    c += (ap == at) ? 1u : 0u;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counts[0], (unsigned long long)c);
}

// flag[0] &= every label is 0 or 1  (flag preset to 1 by the launcher's first kernel)
__global__ void set_flag_kernel(int* flag) { flag[0] = 1; }
__global__ __launch_bounds__(TPB) void labels_binary_kernel(const float* t, int64_t n, int* flag) {
  bool ok = true;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) ok = ok && (t[i] == 0.f || t[i] == 1.f);
  if (!ok) atomicAnd(flag, 0);
}

__global__ __launch_bounds__(TPB) void absmax_kernel(const float* x, int64_t n, float* amax) {
  float m = 0.f;
  const int64_t n4 = n / 4;
  const f4* x4 = reinterpret_cast<const f4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
    const f4 v = x4[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = fabsf(v[k]);
      m = (a == a) ? fmaxf(m, a) : m;
    }
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const float a = fabsf(x[i]);
    m = (a == a) ? fmaxf(m, a) : m;
  }
  m = wave_max(m);
  __shared__ float sm[TPB / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < TPB / 64; ++k) m = fmaxf(m, sm[k]);
    atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(m));   // m >= 0: bit order == value order
  }
}

__global__ void grad_scale_kernel(const float* amax, float target, float* gs) {
  const float a = amax[0];
  float s = 1.f;
  if (a > 0.f && a < 3.0e38f) {
    int e;
    frexpf(a, &e);                       // a = f * 2^e, f in [0.5, 1)
    int et;
    frexpf(target, &et);
    s = ldexpf(1.f, (et - 1) - e);       // a*s in [target/2, target) for a power-of-two target
  }
  gs[0] = s;
  gs[1] = 1.f / s;
}

// ------------------------------------------------------------------------------------------------ packing (the optimiser kernels live in optim.hip)
template <typename T>
__global__ __launch_bounds__(TPB) void transpose_w_kernel(const float* w, T* wt, int rows, int Tn, int K, int rows_pad) {
  // out[k][t][r] = in[r][t][k]: 32 x 32 (r, k) tiles through LDS so that both the fp32 reads (k fastest) and the
  // writes (r fastest) are coalesced.  grid = (ceil(K/32) * ceil(rows_pad/32), Tn)
  __shared__ float tile[32][33];
  const int kt = blockIdx.x % ((K + 31) / 32), rt = blockIdx.x / ((K + 31) / 32);
  const int t = blockIdx.y;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = rt * 32 + ty + 8 * j, k = kt * 32 + tx;
    tile[ty + 8 * j][tx] = (r < rows && k < K) ? w[((int64_t)r * Tn + t) * K + k] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = kt * 32 + ty + 8 * j, r = rt * 32 + tx;
    if (k < K && r < rows_pad) store_elem(wt, ((int64_t)k * Tn + t) * rows_pad + r, tile[tx][ty + 8 * j]);
  }
}

// the same transpose for a table of (weight, K-column slice, destination column offset) jobs in ONE launch: the
// data-gradient operands of every layer, and the side-by-side slices of a gathered data gradient, are refreshed
// after each optimiser step; ~50 launches of a few microseconds each otherwise
template <typename T>
__global__ __launch_bounds__(TPB) void transpose_w_multi_kernel(const seg_transpose_desc* descs, int n) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n - 1;                    // last descriptor with tile0 <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const seg_transpose_desc d = descs[lo];
  const int local = blockIdx.x - d.tile0;
  const int ktn = (d.kcount + 31) / 32, rtn = (d.rows_pad + 31) / 32;
  const int t = local / (ktn * rtn), rem = local - t * (ktn * rtn);
  const int kt = rem % ktn, rt = rem / ktn;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = rt * 32 + ty + 8 * j, k = kt * 32 + tx;
    float v = 0.f;
    if (r < d.rows && k < d.kcount && d.k0 + k < d.K) {
      const int64_t i = ((int64_t)r * d.T + t) * d.K + d.k0 + k;
      v = d.src16 ? (float)reinterpret_cast<const half_t*>(d.w)[i] : d.w[i];      // fp16 operand copy: half the read traffic
    }
    tile[ty + 8 * j][tx] = v;
  }
  __syncthreads();
  T* wt = (T*)d.wt;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = kt * 32 + ty + 8 * j, r = rt * 32 + tx;
    // (dst_T: the destination has more taps than the source - a 1x1 kernel laid into the centre tap dst_t0 of a 3x3 operand)
    if (k < d.kcount && r < d.rows_pad)
      store_elem(wt, ((int64_t)k * (d.dst_T > 0 ? d.dst_T : d.T) + t + d.dst_t0) * d.dst_ld + d.dst_coff + r, tile[tx][ty + 8 * j]);
  }
}

__global__ __launch_bounds__(TPB) void cast_kernel(const float* x, half_t* y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) y[i] = (half_t)x[i];
}

// flat fp32 buffer -> its split image (8 consecutive elements -> [8 hi | 8 lo]); n a multiple of 8
__global__ __launch_bounds__(TPB) void cast_split_kernel(const float* x, split_t* y, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n8; i += (int64_t)gridDim.x * TPB) {
    float f[8];
    load8(x + 8 * i, f);
    store8(y + 8 * i, f);
  }
}

template <typename T>
__global__ __launch_bounds__(TPB) void pack_input_kernel(const float* x, T* y, int64_t npix, int C, int cs, int* overflow) {
  const int64_t n = npix * cs;
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const int c = (int)(i % cs);
    const int64_t pix = i / cs;
    const float v = (c < C) ? x[pix * C + c] : 0.f;
    bad += !(fabsf(v) < 65504.f);          // NaN, Inf, or beyond what an fp16 / split-pair element can hold: it would become Inf downstream
    store_elem(y + pix * cs, c, v);
  }
  if (overflow != nullptr && __any(bad != 0)) {      // (rare: one atomic per wave that saw such an element)
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(overflow, bad);
  }
}

template <typename T>
__global__ __launch_bounds__(TPB) void unpack_kernel(const T* x, float* y, int64_t npix, int C, int cs, int coff) {
  const int64_t n = npix * C;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    y[i] = load_elem(x + pix * cs, coff + c);
  }
}

// channel-slice copy at element granularity (materialises a concatenation whose pieces do not start on 8-channel
// boundaries: MultiResBlock's cat[c3x3, c5x5, c7x7] feeding BatchNormalization, unet_variants.py:95-96)
template <typename T>
__global__ __launch_bounds__(TPB) void copy_channels_kernel(const T* src, int src_cs, int src_coff, T* dst, int dst_cs, int dst_coff,
                                                            int C, int64_t npix, int accumulate) {
  const int64_t n = npix * C;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const int64_t pix = i / C;
    const int c = (int)(i - pix * C);
    // (element index from the tensor's base: a row stride that is not a multiple of 8 - the flattened side of the autoencoder's
    // Reshape - still addresses whole 8-element groups of the split layout correctly)
    const int64_t es = pix * src_cs + src_coff + c, ed = pix * dst_cs + dst_coff + c;
    const float v = load_elem(src, es);
    store_elem(dst, ed, accumulate ? v + load_elem(dst, ed) : v);
  }
}

}  // namespace

// ================================================================================================ C ABI
#define ST(s) ((hipStream_t)(s))

template <typename T>
static int seg_bn_stats_impl(const void* x, int64_t npix, int32_t cs, int32_t coff, int32_t C, float* stats, int32_t nparts,
                            int32_t stats_ld, int32_t stats_coff, seg_stream_t stream) {
  SEG_CHECK_ARG(x && stats && npix > 0 && C > 0 && (C & 7) == 0 && (cs & 7) == 0 && (coff & 7) == 0 && coff + C <= cs && nparts > 0 &&
                    stats_coff + C <= stats_ld,
                "seg_bn_stats: bad arguments (npix=%ld C=%d cs=%d coff=%d nparts=%d)", (long)npix, C, cs, coff, nparts);
  hipLaunchKernelGGL((bn_stats_kernel<T>), dim3(nparts), dim3(TPB), 0, ST(stream), (const T*)x, npix, cs, coff, C, stats, nparts,
                     stats_ld, stats_coff, col_layout(C));
  SEG_CHECK_LAUNCH("seg_bn_stats");
  return SEG_OK;
}
extern "C" int seg_bn_stats(const void* x, int64_t npix, int32_t cs, int32_t coff, int32_t C, float* stats, int32_t nparts, int32_t stats_ld, int32_t stats_coff, seg_stream_t stream) { return seg_bn_stats_impl<half_t>(x, npix, cs, coff, C, stats, nparts, stats_ld, stats_coff, stream); }
extern "C" int seg_bn_stats_f32(const void* x, int64_t npix, int32_t cs, int32_t coff, int32_t C, float* stats, int32_t nparts, int32_t stats_ld, int32_t stats_coff, seg_stream_t stream) { return seg_bn_stats_impl<split_t>(x, npix, cs, coff, C, stats, nparts, stats_ld, stats_coff, stream); }

extern "C" int seg_bn_finalize(const float* stats, int32_t nparts, int32_t C, int64_t count, const float* gamma, const float* beta,
                               float eps, float momentum, float* moving_mean, float* moving_var, int32_t unbiased_moving,
                               int32_t training, float* mean, float* invstd, float* scale, float* shift, seg_stream_t stream) {
  SEG_CHECK_ARG(C > 0 && gamma && beta && moving_mean && moving_var && scale && shift, "seg_bn_finalize: null pointer / C=%d", C);
  SEG_CHECK_ARG(!training || (stats && nparts > 0 && count > 0), "seg_bn_finalize: training needs stats");
  int np = nparts;
  if (training && nparts > 64) {
    const int S = 32;
    hipLaunchKernelGGL(bn_prereduce_kernel, dim3((C + 31) / 32, S), dim3(TPB), 0, ST(stream), const_cast<float*>(stats), nparts, C, S);
    np = S;
  }
  const float unbias = (unbiased_moving && count > 1) ? (float)((double)count / (double)(count - 1)) : 1.f;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(TPB), 0, ST(stream), stats, np, C,
                     training ? (float)(1.0 / (double)count) : 0.f, unbias, gamma, beta, eps, momentum, moving_mean, moving_var, training,
                     mean, invstd, scale, shift);
  SEG_CHECK_LAUNCH("seg_bn_finalize");
  return SEG_OK;
}

extern "C" int seg_bn_fold_inference(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var, float eps,
                                     const float* conv_bias, int32_t C, float* scale, float* shift, seg_stream_t stream) {
  SEG_CHECK_ARG(gamma && beta && moving_mean && moving_var && scale && shift && C > 0, "seg_bn_fold_inference: bad arguments");
  hipLaunchKernelGGL(bn_fold_inference_kernel, dim3((C + TPB - 1) / TPB), dim3(TPB), 0, ST(stream), gamma, beta, moving_mean, moving_var, eps,
                     conv_bias, C, scale, shift);
  SEG_CHECK_LAUNCH("seg_bn_fold_inference");
  return SEG_OK;
}

template <typename T>
static int seg_bn_act_apply_impl(const void* x, int32_t x_cs, int32_t x_coff, void* y, int32_t y_cs, int32_t y_coff, int32_t N, int32_t H,
                                int32_t W, int32_t C, int32_t c_valid, const float* scale, const float* shift, int32_t act, float alpha,
                                void* pooled, int32_t ph, int32_t pw, const seg_head_fwd_desc* head, seg_stream_t stream) {
  SEG_CHECK_ARG(x && y && scale && shift && N > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && (x_cs & 7) == 0 && (y_cs & 7) == 0 &&
                    (x_coff & 7) == 0 && (y_coff & 7) == 0 && x_coff + C <= x_cs && y_coff + C <= y_cs,
                "seg_bn_act_apply: bad arguments");
  const int64_t npix = (int64_t)N * H * W;
  HeadFwd hd{0, {nullptr, nullptr}, {nullptr, nullptr}, {0, 0}, {nullptr, nullptr}};
  if (head != nullptr) {
    const int CH = C / 8;
    SEG_CHECK_ARG(head->n >= 1 && head->n <= 2 && CH <= 64 && (CH & (CH - 1)) == 0,
                  "seg_bn_act_apply_head: one or two one-channel heads on a tensor of 8 .. 512 channels (a power of two of 8-channel groups)");
    hd.n = head->n;
    for (int k = 0; k < head->n; ++k) {
      SEG_CHECK_ARG(head->w[k] && head->y[k] && (head->act[k] == SEG_ACT_NONE || head->act[k] == SEG_ACT_SIGMOID || head->act[k] == SEG_ACT_RELU),
                    "seg_bn_act_apply_head: head %d (weights, output, activation none / sigmoid / relu)", k);
      hd.w[k] = head->w[k]; hd.b[k] = head->b[k]; hd.act[k] = head->act[k]; hd.y[k] = head->y[k];
    }
  }
  if (pooled == nullptr) {
    const dim3 grid(grid_for((npix * (C / 8) + UNR - 1) / UNR));
    if (head != nullptr)
      hipLaunchKernelGGL((bn_act_apply_kernel<T, true>), grid, dim3(TPB), 0, ST(stream), (const T*)x, x_cs, x_coff, (T*)y, y_cs, y_coff, npix, C,
                         c_valid, scale, shift, act, alpha, hd);
    else
      hipLaunchKernelGGL((bn_act_apply_kernel<T, false>), grid, dim3(TPB), 0, ST(stream), (const T*)x, x_cs, x_coff, (T*)y, y_cs, y_coff, npix, C,
                         c_valid, scale, shift, act, alpha, hd);
  } else {
    SEG_CHECK_ARG(ph >= 1 && pw >= 1 && y_coff == 0 && C == y_cs, "seg_bn_act_apply: fused pooling needs a dense output (C == y_cs)");
    const int64_t items = (int64_t)N * ((H + ph - 1) / ph) * ((W + pw - 1) / pw) * (C / 8);
    if (head != nullptr)
      hipLaunchKernelGGL((bn_act_apply_pool_kernel<T, true>), dim3(grid_for(items)), dim3(TPB), 0, ST(stream), (const T*)x, x_cs, x_coff,
                         (T*)y, y_cs, y_coff, N, H, W, C, c_valid, scale, shift, act, alpha, (T*)pooled, ph, pw, hd);
    else
      hipLaunchKernelGGL((bn_act_apply_pool_kernel<T, false>), dim3(grid_for(items)), dim3(TPB), 0, ST(stream), (const T*)x, x_cs, x_coff,
                         (T*)y, y_cs, y_coff, N, H, W, C, c_valid, scale, shift, act, alpha, (T*)pooled, ph, pw, hd);
  }
  SEG_CHECK_LAUNCH("seg_bn_act_apply");
  return SEG_OK;
}
extern "C" int seg_bn_act_apply(const void* x, int32_t x_cs, int32_t x_coff, void* y, int32_t y_cs, int32_t y_coff, int32_t N, int32_t H, int32_t W, int32_t C, int32_t c_valid, const float* scale, const float* shift, int32_t act, float alpha, void* pooled, int32_t ph, int32_t pw, seg_stream_t stream) { return seg_bn_act_apply_impl<half_t>(x, x_cs, x_coff, y, y_cs, y_coff, N, H, W, C, c_valid, scale, shift, act, alpha, pooled, ph, pw, nullptr, stream); }
extern "C" int seg_bn_act_apply_f32(const void* x, int32_t x_cs, int32_t x_coff, void* y, int32_t y_cs, int32_t y_coff, int32_t N, int32_t H, int32_t W, int32_t C, int32_t c_valid, const float* scale, const float* shift, int32_t act, float alpha, void* pooled, int32_t ph, int32_t pw, seg_stream_t stream) { return seg_bn_act_apply_impl<split_t>(x, x_cs, x_coff, y, y_cs, y_coff, N, H, W, C, c_valid, scale, shift, act, alpha, pooled, ph, pw, nullptr, stream); }
extern "C" int seg_bn_act_apply_head(const void* x, int32_t x_cs, int32_t x_coff, void* y, int32_t y_cs, int32_t y_coff, int32_t N, int32_t H, int32_t W, int32_t C, int32_t c_valid, const float* scale, const float* shift, int32_t act, float alpha, void* pooled, int32_t ph, int32_t pw, const seg_head_fwd_desc* head, seg_stream_t stream) { return seg_bn_act_apply_impl<half_t>(x, x_cs, x_coff, y, y_cs, y_coff, N, H, W, C, c_valid, scale, shift, act, alpha, pooled, ph, pw, head, stream); }
extern "C" int seg_bn_act_apply_head_f32(const void* x, int32_t x_cs, int32_t x_coff, void* y, int32_t y_cs, int32_t y_coff, int32_t N, int32_t H, int32_t W, int32_t C, int32_t c_valid, const float* scale, const float* shift, int32_t act, float alpha, void* pooled, int32_t ph, int32_t pw, const seg_head_fwd_desc* head, seg_stream_t stream) { return seg_bn_act_apply_impl<split_t>(x, x_cs, x_coff, y, y_cs, y_coff, N, H, W, C, c_valid, scale, shift, act, alpha, pooled, ph, pw, head, stream); }

static int head_inj_check(const seg_head_inject* inj, HeadInj* hj) {
  *hj = HeadInj{0, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {0, 0}, nullptr};
  if (inj == nullptr) return SEG_OK;
  SEG_CHECK_ARG(inj->n >= 1 && inj->n <= 2, "head gradient injection: one or two heads");
  hj->n = inj->n;
  hj->gs = inj->gs_ptr;
  for (int k = 0; k < inj->n; ++k) {
    SEG_CHECK_ARG(inj->dy[k] && inj->y[k] && inj->w[k] && (inj->act[k] == SEG_ACT_NONE || inj->act[k] == SEG_ACT_SIGMOID || inj->act[k] == SEG_ACT_RELU),
                  "head gradient injection: dy, y, w of one-channel head %d (activation none / sigmoid / relu)", k);
    hj->dy[k] = inj->dy[k]; hj->y[k] = inj->y[k]; hj->w[k] = inj->w[k]; hj->act[k] = inj->act[k];
  }
  return SEG_OK;
}

template <typename T>
static int seg_bn_act_bwd_reduce_impl(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs, int32_t x_coff,
                                     int64_t npix, int32_t C, const float* scale, const float* shift, const float* mean,
                                     const float* invstd, int32_t act, float alpha, float* partial, int32_t nparts,
                                     const seg_head_inject* inj, seg_stream_t stream) {
  SEG_CHECK_ARG((dy || inj) && x && scale && shift && mean && invstd && partial && npix > 0 && C > 0 && (C & 7) == 0 && nparts > 0 &&
                    dy_coff + C <= dy_cs && x_coff + C <= x_cs,
                "seg_bn_act_bwd_reduce: bad arguments");
  HeadInj hj;
  if (int rc = head_inj_check(inj, &hj)) return rc;
  if (inj != nullptr)
    hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<T, true>), dim3(nparts), dim3(TPB), 0, ST(stream), (const T*)dy, dy_cs, dy_coff, (const T*)x,
                       x_cs, x_coff, npix, C, scale, shift, mean, invstd, act, alpha, partial, nparts, col_layout(C), hj);
  else
    hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<T, false>), dim3(nparts), dim3(TPB), 0, ST(stream), (const T*)dy, dy_cs, dy_coff, (const T*)x,
                       x_cs, x_coff, npix, C, scale, shift, mean, invstd, act, alpha, partial, nparts, col_layout(C), hj);
  SEG_CHECK_LAUNCH("seg_bn_act_bwd_reduce");
  return SEG_OK;
}
extern "C" int seg_bn_act_bwd_reduce(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs, int32_t x_coff, int64_t npix, int32_t C, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t act, float alpha, float* partial, int32_t nparts, seg_stream_t stream) { return seg_bn_act_bwd_reduce_impl<half_t>(dy, dy_cs, dy_coff, x, x_cs, x_coff, npix, C, scale, shift, mean, invstd, act, alpha, partial, nparts, nullptr, stream); }
extern "C" int seg_bn_act_bwd_reduce_f32(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs, int32_t x_coff, int64_t npix, int32_t C, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t act, float alpha, float* partial, int32_t nparts, seg_stream_t stream) { return seg_bn_act_bwd_reduce_impl<split_t>(dy, dy_cs, dy_coff, x, x_cs, x_coff, npix, C, scale, shift, mean, invstd, act, alpha, partial, nparts, nullptr, stream); }
extern "C" int seg_bn_act_bwd_reduce_inj(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs, int32_t x_coff, int64_t npix, int32_t C, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t act, float alpha, float* partial, int32_t nparts, const seg_head_inject* inj, seg_stream_t stream) { return seg_bn_act_bwd_reduce_impl<half_t>(dy, dy_cs, dy_coff, x, x_cs, x_coff, npix, C, scale, shift, mean, invstd, act, alpha, partial, nparts, inj, stream); }
extern "C" int seg_bn_act_bwd_reduce_inj_f32(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs, int32_t x_coff, int64_t npix, int32_t C, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t act, float alpha, float* partial, int32_t nparts, const seg_head_inject* inj, seg_stream_t stream) { return seg_bn_act_bwd_reduce_impl<split_t>(dy, dy_cs, dy_coff, x, x_cs, x_coff, npix, C, scale, shift, mean, invstd, act, alpha, partial, nparts, inj, stream); }

extern "C" int seg_bn_bwd_finalize(const float* partial, int32_t nparts, int32_t C, int64_t count, float* dgamma, float* dbeta,
                                   float gscale, const float* gscale_ptr, float* c1, float* c2, seg_stream_t stream) {
  SEG_CHECK_ARG(partial && c1 && c2 && nparts > 0 && C > 0 && count > 0, "seg_bn_bwd_finalize: bad arguments");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(TPB), 0, ST(stream), partial, nparts, C,
                     (float)(1.0 / (double)count), dgamma, dbeta, gscale, gscale_ptr, c1, c2);
  SEG_CHECK_LAUNCH("seg_bn_bwd_finalize");
  return SEG_OK;
}

template <typename T>
static int seg_bn_act_bwd_apply_impl(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs, int32_t x_coff, void* dx,
                                    int32_t dx_cs, int32_t dx_coff, int64_t npix, int32_t C, const float* scale, const float* shift,
                                    const float* mean, const float* invstd, const float* c1, const float* c2, int32_t act, float alpha,
                                    int32_t accumulate, const seg_head_inject* inj, seg_stream_t stream) {
  SEG_CHECK_ARG((dy || inj) && x && dx && scale && shift && npix > 0 && C > 0 && (C & 7) == 0 && dy_coff + C <= dy_cs && x_coff + C <= x_cs &&
                    dx_coff + C <= dx_cs && (c1 == nullptr || (c2 && mean && invstd)),
                "seg_bn_act_bwd_apply: bad arguments");
  HeadInj hj;
  if (int rc = head_inj_check(inj, &hj)) return rc;
  const ColLayout L = col_layout(C);
  int64_t blocks = (npix + (int64_t)L.RP * UNR - 1) / ((int64_t)L.RP * UNR);
  if (blocks > 512) blocks = 512;       // two blocks per CU: 96 us per 134 MB tensor against 114 us at 2048 blocks (measured sweep)
  if (inj != nullptr)
    hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, true>), dim3((int)blocks), dim3(TPB), 0, ST(stream), (const T*)dy, dy_cs, dy_coff,
                       (const T*)x, x_cs, x_coff, (T*)dx, dx_cs, dx_coff, npix, C, scale, shift, mean, invstd, c1, c2, act, alpha,
                       accumulate, L, hj);
  else
    hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, false>), dim3((int)blocks), dim3(TPB), 0, ST(stream), (const T*)dy, dy_cs, dy_coff,
                       (const T*)x, x_cs, x_coff, (T*)dx, dx_cs, dx_coff, npix, C, scale, shift, mean, invstd, c1, c2, act, alpha,
                       accumulate, L, hj);
  SEG_CHECK_LAUNCH("seg_bn_act_bwd_apply");
  return SEG_OK;
}
extern "C" int seg_bn_act_bwd_apply(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs, int32_t x_coff, void* dx, int32_t dx_cs, int32_t dx_coff, int64_t npix, int32_t C, const float* scale, const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2, int32_t act, float alpha, int32_t accumulate, seg_stream_t stream) { return seg_bn_act_bwd_apply_impl<half_t>(dy, dy_cs, dy_coff, x, x_cs, x_coff, dx, dx_cs, dx_coff, npix, C, scale, shift, mean, invstd, c1, c2, act, alpha, accumulate, nullptr, stream); }
extern "C" int seg_bn_act_bwd_apply_f32(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs, int32_t x_coff, void* dx, int32_t dx_cs, int32_t dx_coff, int64_t npix, int32_t C, const float* scale, const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2, int32_t act, float alpha, int32_t accumulate, seg_stream_t stream) { return seg_bn_act_bwd_apply_impl<split_t>(dy, dy_cs, dy_coff, x, x_cs, x_coff, dx, dx_cs, dx_coff, npix, C, scale, shift, mean, invstd, c1, c2, act, alpha, accumulate, nullptr, stream); }
extern "C" int seg_bn_act_bwd_apply_inj(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs, int32_t x_coff, void* dx, int32_t dx_cs, int32_t dx_coff, int64_t npix, int32_t C, const float* scale, const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2, int32_t act, float alpha, int32_t accumulate, const seg_head_inject* inj, seg_stream_t stream) { return seg_bn_act_bwd_apply_impl<half_t>(dy, dy_cs, dy_coff, x, x_cs, x_coff, dx, dx_cs, dx_coff, npix, C, scale, shift, mean, invstd, c1, c2, act, alpha, accumulate, inj, stream); }
extern "C" int seg_bn_act_bwd_apply_inj_f32(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs, int32_t x_coff, void* dx, int32_t dx_cs, int32_t dx_coff, int64_t npix, int32_t C, const float* scale, const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2, int32_t act, float alpha, int32_t accumulate, const seg_head_inject* inj, seg_stream_t stream) { return seg_bn_act_bwd_apply_impl<split_t>(dy, dy_cs, dy_coff, x, x_cs, x_coff, dx, dx_cs, dx_coff, npix, C, scale, shift, mean, invstd, c1, c2, act, alpha, accumulate, inj, stream); }

template <typename T>
static int seg_bn_act_bwd_apply2_impl(const void* dy, const seg_bn_bwd_set* sets, int64_t npix, int32_t C, seg_stream_t stream) {
  SEG_CHECK_ARG(dy && sets && npix > 0 && C > 0 && (C & 7) == 0, "seg_bn_act_bwd_apply2: bad arguments");
  BwdSet2 bs;
  for (int j = 0; j < 2; ++j) {
    const seg_bn_bwd_set& q = sets[j];
    SEG_CHECK_ARG(q.x && q.dx && q.scale && q.shift && q.mean && q.invstd && q.c1 && q.c2, "seg_bn_act_bwd_apply2: set %d", j);
    bs.s[j] = BwdSet{q.x, q.dx, q.scale, q.shift, q.mean, q.invstd, q.c1, q.c2, q.act, q.alpha};
  }
  const ColLayout L = col_layout(C);
  int64_t blocks = (npix + (int64_t)L.RP * 2 - 1) / ((int64_t)L.RP * 2);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL((bn_act_bwd_apply2_kernel<T>), dim3((int)blocks), dim3(TPB), 0, ST(stream), (const T*)dy, bs, npix, C, L);
  SEG_CHECK_LAUNCH("seg_bn_act_bwd_apply2");
  return SEG_OK;
}
extern "C" int seg_bn_act_bwd_apply2(const void* dy, const seg_bn_bwd_set* sets, int64_t npix, int32_t C, seg_stream_t stream) { return seg_bn_act_bwd_apply2_impl<half_t>(dy, sets, npix, C, stream); }
extern "C" int seg_bn_act_bwd_apply2_f32(const void* dy, const seg_bn_bwd_set* sets, int64_t npix, int32_t C, seg_stream_t stream) { return seg_bn_act_bwd_apply2_impl<split_t>(dy, sets, npix, C, stream); }

template <typename T>
static int seg_act_bwd_impl(const void* dy, const void* y, void* dx, int64_t npix, int32_t cs, int32_t C, int32_t act, float alpha,
                           float* dbias, float gscale, const float* gscale_ptr, seg_stream_t stream) {
  SEG_CHECK_ARG(dy && y && dx && npix > 0 && cs > 0 && (cs & 7) == 0 && C > 0 && C <= cs, "seg_act_bwd: bad arguments");
  int blocks = (int)((npix + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((act_bwd_kernel<T>), dim3(blocks), dim3(TPB), 0, ST(stream), (const T*)dy, (const T*)y, (T*)dx, npix, cs, C,
                     act, alpha, dbias, gscale, gscale_ptr, col_layout(cs));
  SEG_CHECK_LAUNCH("seg_act_bwd");
  return SEG_OK;
}
extern "C" int seg_act_bwd(const void* dy, const void* y, void* dx, int64_t npix, int32_t cs, int32_t C, int32_t act, float alpha, float* dbias, float gscale, const float* gscale_ptr, seg_stream_t stream) { return seg_act_bwd_impl<half_t>(dy, y, dx, npix, cs, C, act, alpha, dbias, gscale, gscale_ptr, stream); }
extern "C" int seg_act_bwd_f32(const void* dy, const void* y, void* dx, int64_t npix, int32_t cs, int32_t C, int32_t act, float alpha, float* dbias, float gscale, const float* gscale_ptr, seg_stream_t stream) { return seg_act_bwd_impl<split_t>(dy, y, dx, npix, cs, C, act, alpha, dbias, gscale, gscale_ptr, stream); }

template <typename T>
static int seg_maxpool_fwd_impl(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t ph, int32_t pw,
                               seg_stream_t stream) {
  SEG_CHECK_ARG(x && y && N > 0 && H >= ph && W >= pw && ph >= 1 && pw >= 1 && (cs & 7) == 0, "seg_maxpool_fwd: bad arguments");
  const int64_t items = (int64_t)N * (H / ph) * (W / pw) * (cs / 8);
  hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(grid_for(items)), dim3(TPB), 0, ST(stream), (const T*)x, (T*)y, N, H, W, cs, ph, pw);
  SEG_CHECK_LAUNCH("seg_maxpool_fwd");
  return SEG_OK;
}
extern "C" int seg_maxpool_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t ph, int32_t pw, seg_stream_t stream) { return seg_maxpool_fwd_impl<half_t>(x, y, N, H, W, cs, ph, pw, stream); }
extern "C" int seg_maxpool_fwd_f32(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t ph, int32_t pw, seg_stream_t stream) { return seg_maxpool_fwd_impl<split_t>(x, y, N, H, W, cs, ph, pw, stream); }

template <typename T>
static int seg_maxpool_bwd_impl(const void* dy, const void* x, const void* y, void* dx, int32_t N, int32_t H, int32_t W, int32_t cs,
                               int32_t ph, int32_t pw, int32_t accumulate, seg_stream_t stream) {
  SEG_CHECK_ARG(dy && x && y && dx && N > 0 && H >= ph && W >= pw && ph >= 1 && pw >= 1 && (cs & 7) == 0, "seg_maxpool_bwd: bad arguments");
  const int64_t items = (int64_t)N * ((H + ph - 1) / ph) * ((W + pw - 1) / pw) * (cs / 8);
  hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(grid_for(items)), dim3(TPB), 0, ST(stream), (const T*)dy, (const T*)x,
                     (const T*)y, (T*)dx, N, H, W, cs, ph, pw, accumulate);
  SEG_CHECK_LAUNCH("seg_maxpool_bwd");
  return SEG_OK;
}
extern "C" int seg_maxpool_bwd(const void* dy, const void* x, const void* y, void* dx, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t ph, int32_t pw, int32_t accumulate, seg_stream_t stream) { return seg_maxpool_bwd_impl<half_t>(dy, x, y, dx, N, H, W, cs, ph, pw, accumulate, stream); }
extern "C" int seg_maxpool_bwd_f32(const void* dy, const void* x, const void* y, void* dx, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t ph, int32_t pw, int32_t accumulate, seg_stream_t stream) { return seg_maxpool_bwd_impl<split_t>(dy, x, y, dx, N, H, W, cs, ph, pw, accumulate, stream); }

template <typename T>
static int seg_upsample_fwd_impl(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t fh, int32_t fw,
                                int32_t bilinear, seg_stream_t stream) {
  SEG_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && fh >= 1 && fw >= 1 && (cs & 7) == 0, "seg_upsample_fwd: bad arguments");
  const int64_t items = (int64_t)N * H * fh * W * fw * (cs / 8);
  hipLaunchKernelGGL((upsample_fwd_kernel<T>), dim3(grid_for(items)), dim3(TPB), 0, ST(stream), (const T*)x, (T*)y, N, H, W, cs, fh, fw,
                     bilinear);
  SEG_CHECK_LAUNCH("seg_upsample_fwd");
  return SEG_OK;
}
extern "C" int seg_upsample_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t fh, int32_t fw, int32_t bilinear, seg_stream_t stream) { return seg_upsample_fwd_impl<half_t>(x, y, N, H, W, cs, fh, fw, bilinear, stream); }
extern "C" int seg_upsample_fwd_f32(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t fh, int32_t fw, int32_t bilinear, seg_stream_t stream) { return seg_upsample_fwd_impl<split_t>(x, y, N, H, W, cs, fh, fw, bilinear, stream); }

template <typename T>
static int seg_upsample_bwd_impl(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t fh, int32_t fw,
                                int32_t bilinear, int32_t accumulate, seg_stream_t stream) {
  SEG_CHECK_ARG(dy && dx && N > 0 && H > 0 && W > 0 && fh >= 1 && fw >= 1 && (cs & 7) == 0, "seg_upsample_bwd: bad arguments");
  const int64_t items = (int64_t)N * H * W * (cs / 8);
  hipLaunchKernelGGL((upsample_bwd_kernel<T>), dim3(grid_for(items)), dim3(TPB), 0, ST(stream), (const T*)dy, (T*)dx, N, H, W, cs, fh,
                     fw, bilinear, accumulate);
  SEG_CHECK_LAUNCH("seg_upsample_bwd");
  return SEG_OK;
}
extern "C" int seg_upsample_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t fh, int32_t fw, int32_t bilinear, int32_t accumulate, seg_stream_t stream) { return seg_upsample_bwd_impl<half_t>(dy, dx, N, H, W, cs, fh, fw, bilinear, accumulate, stream); }
extern "C" int seg_upsample_bwd_f32(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t fh, int32_t fw, int32_t bilinear, int32_t accumulate, seg_stream_t stream) { return seg_upsample_bwd_impl<split_t>(dy, dx, N, H, W, cs, fh, fw, bilinear, accumulate, stream); }

template <typename T>
static int seg_add_act_fwd_impl(const void* const* xs, int32_t n, void* y, int64_t nelem, int32_t cs, int32_t c_valid, int32_t act, float alpha,
                                seg_stream_t stream) {
  SEG_CHECK_ARG(xs && y && n >= 1 && n <= 4 && nelem > 0 && cs > 0 && (cs & 7) == 0 && nelem % cs == 0 && c_valid > 0 && c_valid <= cs,
                "seg_add_act_fwd: bad arguments (n=%d cs=%d c_valid=%d)", n, cs, c_valid);
  Ptr4<T> p;
  for (int k = 0; k < 4; ++k) p.p[k] = k < n ? (const T*)xs[k] : nullptr;
  hipLaunchKernelGGL((add_act_fwd_kernel<T>), dim3(grid_for(nelem / 8)), dim3(TPB), 0, ST(stream), p, n, (T*)y, nelem / 8, cs / 8, c_valid, act, alpha);
  SEG_CHECK_LAUNCH("seg_add_act_fwd");
  return SEG_OK;
}
extern "C" int seg_add_act_fwd(const void* const* xs, int32_t n, void* y, int64_t nelem, int32_t cs, int32_t c_valid, int32_t act, float alpha, seg_stream_t stream) { return seg_add_act_fwd_impl<half_t>(xs, n, y, nelem, cs, c_valid, act, alpha, stream); }
extern "C" int seg_add_act_fwd_f32(const void* const* xs, int32_t n, void* y, int64_t nelem, int32_t cs, int32_t c_valid, int32_t act, float alpha, seg_stream_t stream) { return seg_add_act_fwd_impl<split_t>(xs, n, y, nelem, cs, c_valid, act, alpha, stream); }

template <typename T>
static int seg_add_act_bwd_impl(const void* dy, const void* y, void* const* dxs, int32_t n, int64_t nelem, int32_t act, float alpha,
                               int32_t accumulate_mask, seg_stream_t stream) {
  SEG_CHECK_ARG(dy && dxs && n >= 1 && n <= 4 && nelem > 0 && (nelem & 7) == 0 && (act == SEG_ACT_NONE || y), "seg_add_act_bwd: bad arguments");
  MPtr4<T> p;
  for (int k = 0; k < 4; ++k) p.p[k] = k < n ? (T*)dxs[k] : nullptr;
  hipLaunchKernelGGL((add_act_bwd_kernel<T>), dim3(grid_for(nelem / 8)), dim3(TPB), 0, ST(stream), (const T*)dy, (const T*)y, p, n,
                     nelem / 8, act, alpha, accumulate_mask);
  SEG_CHECK_LAUNCH("seg_add_act_bwd");
  return SEG_OK;
}
extern "C" int seg_add_act_bwd(const void* dy, const void* y, void* const* dxs, int32_t n, int64_t nelem, int32_t act, float alpha, int32_t accumulate_mask, seg_stream_t stream) { return seg_add_act_bwd_impl<half_t>(dy, y, dxs, n, nelem, act, alpha, accumulate_mask, stream); }
extern "C" int seg_add_act_bwd_f32(const void* dy, const void* y, void* const* dxs, int32_t n, int64_t nelem, int32_t act, float alpha, int32_t accumulate_mask, seg_stream_t stream) { return seg_add_act_bwd_impl<split_t>(dy, y, dxs, n, nelem, act, alpha, accumulate_mask, stream); }

static int cat_to_dev(const seg_cat_desc* cat, int32_t cs, bool backward, CatDev* cd) {
  cd->n = 0; cd->ngroups = 0; cd->ctotal = 0;
  for (int k = 0; k < 4; ++k) { cd->x[k] = nullptr; cd->dx[k] = nullptr; cd->cs[k] = cd->C[k] = cd->coff[k] = cd->g0[k] = cd->acc[k] = 0; }
  if (cat == nullptr) return SEG_OK;
  SEG_CHECK_ARG(cat->n >= 1 && cat->n <= 4, "virtual concatenation: 1 .. 4 pieces (got %d)", cat->n);
  cd->n = cat->n;
  for (int k = 0; k < cat->n; ++k) {
    const seg_cat_piece& q = cat->p[k];
    SEG_CHECK_ARG(q.x != nullptr && q.C > 0 && q.cs >= q.C && (q.cs & 7) == 0 && q.coff == cd->ctotal,
                  "virtual concatenation: piece %d (C=%d cs=%d coff=%d, expected offset %d)", k, q.C, q.cs, q.coff, cd->ctotal);
    cd->x[k] = q.x; cd->dx[k] = backward ? q.dx : nullptr; cd->cs[k] = q.cs; cd->C[k] = q.C; cd->coff[k] = q.coff; cd->acc[k] = q.acc;
    cd->g0[k] = cd->ngroups;
    cd->ngroups += (q.C + 7) / 8;
    cd->ctotal += q.C;
  }
  SEG_CHECK_ARG(cd->ctotal <= cs && cs <= 8 * TPB, "virtual concatenation: %d channels into a row of %d (row limit %d)", cd->ctotal, cs, 8 * TPB);
  return SEG_OK;
}

extern "C" int32_t seg_tail_row_limit(void) { return 8 * TPB; }

template <typename T>
static int seg_add_act_stats_fwd_impl(const void* const* xs, const float* const* scales, const float* const* shifts, const int32_t* relus, int32_t n,
                                      const seg_cat_desc* cat, const float* cat_scale, const float* cat_shift, void* y, int64_t npix, int32_t cs,
                                      int32_t c_valid, int32_t act, float alpha, float* stats, int32_t nparts, int32_t stats_ld, int32_t stats_coff,
                                      seg_stream_t stream) {
  // (the kernel takes stats_coff = -1 for the tensor's own rows - all cs columns written - and >= 0 for another tensor's rows, of
  // which only the c_valid columns of this tensor are written)
  if (stats_ld <= 0) { stats_ld = cs; stats_coff = -1; }
  else SEG_CHECK_ARG(stats_coff >= 0 && stats_coff + c_valid <= stats_ld, "seg_add_act_stats_fwd: statistics columns %d + %d of %d", stats_coff, c_valid, stats_ld);
  SEG_CHECK_ARG(stats && n >= 0 && n <= 4 && (n > 0 || cat != nullptr) && (n == 0 || xs != nullptr) && npix > 0 && cs > 0 && (cs & 7) == 0 &&
                cs <= 8 * TPB && nparts > 0 && c_valid > 0 && c_valid <= cs && (cat_scale == nullptr) == (cat_shift == nullptr),
                "seg_add_act_stats_fwd: bad arguments (n=%d cs=%d)", n, cs);
  Ptr4<T> p;
  Aff4 a;
  for (int k = 0; k < 4; ++k) {
    p.p[k] = k < n ? (const T*)xs[k] : nullptr;
    a.sc[k] = (k < n && scales != nullptr) ? scales[k] : nullptr;
    a.sh[k] = (k < n && shifts != nullptr) ? shifts[k] : nullptr;
    a.relu[k] = (k < n && relus != nullptr) ? relus[k] : 0;
    SEG_CHECK_ARG(k >= n || (p.p[k] != nullptr && (a.sc[k] == nullptr) == (a.sh[k] == nullptr)), "seg_add_act_stats_fwd: input %d", k);
  }
  CatDev cd;
  if (int rc = cat_to_dev(cat, cs, false, &cd)) return rc;
  const ColLayout L = col_layout(cs);
#define LAUNCH_TAIL_FWD(G, NIN)                                                                                                          \
  hipLaunchKernelGGL((tail_fwd_kernel<T, G, NIN>), dim3(nparts), dim3(TPB), 0, ST(stream), p, a, n, cd, cat_scale, cat_shift, (T*)y, npix, cs, \
                     c_valid, act, alpha, stats, stats_ld, stats_coff, L)
  if (cat != nullptr) { if (n <= 2) LAUNCH_TAIL_FWD(true, 2); else LAUNCH_TAIL_FWD(true, 4); }
  else { if (n <= 2) LAUNCH_TAIL_FWD(false, 2); else LAUNCH_TAIL_FWD(false, 4); }
#undef LAUNCH_TAIL_FWD
  SEG_CHECK_LAUNCH("seg_add_act_stats_fwd");
  return SEG_OK;
}
extern "C" int seg_add_act_stats_fwd(const void* const* xs, const float* const* scales, const float* const* shifts, const int32_t* relus, int32_t n, const seg_cat_desc* cat, const float* cat_scale, const float* cat_shift, void* y, int64_t npix, int32_t cs, int32_t c_valid, int32_t act, float alpha, float* stats, int32_t nparts, int32_t stats_ld, int32_t stats_coff, seg_stream_t stream) { return seg_add_act_stats_fwd_impl<half_t>(xs, scales, shifts, relus, n, cat, cat_scale, cat_shift, y, npix, cs, c_valid, act, alpha, stats, nparts, stats_ld, stats_coff, stream); }
extern "C" int seg_add_act_stats_fwd_f32(const void* const* xs, const float* const* scales, const float* const* shifts, const int32_t* relus, int32_t n, const seg_cat_desc* cat, const float* cat_scale, const float* cat_shift, void* y, int64_t npix, int32_t cs, int32_t c_valid, int32_t act, float alpha, float* stats, int32_t nparts, int32_t stats_ld, int32_t stats_coff, seg_stream_t stream) { return seg_add_act_stats_fwd_impl<split_t>(xs, scales, shifts, relus, n, cat, cat_scale, cat_shift, y, npix, cs, c_valid, act, alpha, stats, nparts, stats_ld, stats_coff, stream); }

template <typename T>
static int seg_bn_bwd_apply_fanout_impl(const void* dy, const void* x, void* const* dxs, int32_t n, int32_t accumulate_mask, int64_t npix, int32_t cs,
                                        const float* scale, const float* mean, const float* invstd, const float* c1, const float* c2,
                                        int32_t relu_mask, const seg_cat_desc* cat, const float* cat_mean, const float* cat_invstd,
                                        float* cat_partial, const seg_lazy_bn* lazy, int32_t nlazy, int32_t nparts, seg_stream_t stream) {
  SEG_CHECK_ARG(dy && x && dxs && n >= 1 && n <= 4 && npix > 0 && cs > 0 && (cs & 7) == 0 && cs <= 8 * TPB && scale && mean && invstd && c1 && c2 &&
                (cat == nullptr || (cat_mean && cat_invstd && cat_partial)) && nlazy >= 0 && nlazy <= 2 && (nlazy == 0 || lazy != nullptr) &&
                ((cat == nullptr && nlazy == 0) || nparts > 0), "seg_bn_bwd_apply_fanout: bad arguments");
  MPtr4<T> p;
  for (int k = 0; k < 4; ++k) p.p[k] = k < n ? (T*)dxs[k] : nullptr;
  CatDev cd;
  if (int rc = cat_to_dev(cat, cs, false, &cd)) return rc;
  LazyBN2 lz;
  for (int j = 0; j < 2; ++j) {
    lz.l[j] = LazyBN{nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr};
    if (j < nlazy) {
      const seg_lazy_bn& q = lazy[j];
      SEG_CHECK_ARG(q.z && q.scale && q.shift && q.mean && q.invstd && q.partial, "seg_bn_bwd_apply_fanout: lazy input %d", j);
      lz.l[j] = LazyBN{q.z, q.scale, q.shift, q.mean, q.invstd, q.relu, q.partial};
    }
  }
  const ColLayout L = col_layout(cs);
  int64_t blocks = (npix + (int64_t)L.RP * UNR - 1) / ((int64_t)L.RP * UNR);
  if (blocks > 512) blocks = 512;
  if (cat != nullptr || nlazy > 0) blocks = nparts;      // one partial row per block: the grid IS the row count the finalize launches are told
#define LAUNCH_FANOUT(G, LZN)                                                                                                                \
  hipLaunchKernelGGL((tail_bwd_fanout_kernel<T, G, LZN>), dim3((int)blocks), dim3(TPB), 0, ST(stream), (const T*)dy, (const T*)x, p, n,       \
                     accumulate_mask, npix, cs, scale, mean, invstd, c1, c2, relu_mask, cd, cat_mean, cat_invstd, cat_partial, lz, L)
  if (cat != nullptr) { if (nlazy == 0) LAUNCH_FANOUT(true, 0); else if (nlazy == 1) LAUNCH_FANOUT(true, 1); else LAUNCH_FANOUT(true, 2); }
  else { if (nlazy == 0) LAUNCH_FANOUT(false, 0); else if (nlazy == 1) LAUNCH_FANOUT(false, 1); else LAUNCH_FANOUT(false, 2); }
#undef LAUNCH_FANOUT
  SEG_CHECK_LAUNCH("seg_bn_bwd_apply_fanout");
  return SEG_OK;
}
extern "C" int seg_bn_bwd_apply_fanout(const void* dy, const void* x, void* const* dxs, int32_t n, int32_t accumulate_mask, int64_t npix, int32_t cs, const float* scale, const float* mean, const float* invstd, const float* c1, const float* c2, int32_t relu_mask, const seg_cat_desc* cat, const float* cat_mean, const float* cat_invstd, float* cat_partial, const seg_lazy_bn* lazy, int32_t nlazy, int32_t nparts, seg_stream_t stream) { return seg_bn_bwd_apply_fanout_impl<half_t>(dy, x, dxs, n, accumulate_mask, npix, cs, scale, mean, invstd, c1, c2, relu_mask, cat, cat_mean, cat_invstd, cat_partial, lazy, nlazy, nparts, stream); }
extern "C" int seg_bn_bwd_apply_fanout_f32(const void* dy, const void* x, void* const* dxs, int32_t n, int32_t accumulate_mask, int64_t npix, int32_t cs, const float* scale, const float* mean, const float* invstd, const float* c1, const float* c2, int32_t relu_mask, const seg_cat_desc* cat, const float* cat_mean, const float* cat_invstd, float* cat_partial, const seg_lazy_bn* lazy, int32_t nlazy, int32_t nparts, seg_stream_t stream) { return seg_bn_bwd_apply_fanout_impl<split_t>(dy, x, dxs, n, accumulate_mask, npix, cs, scale, mean, invstd, c1, c2, relu_mask, cat, cat_mean, cat_invstd, cat_partial, lazy, nlazy, nparts, stream); }

template <typename T>
static int seg_cat_bn_bwd_scatter_impl(const void* g, const seg_cat_desc* cat, int64_t npix, int32_t cs, const float* scale, const float* mean,
                                       const float* invstd, const float* c1, const float* c2, seg_stream_t stream) {
  SEG_CHECK_ARG(g && cat && npix > 0 && cs > 0 && (cs & 7) == 0 && scale && mean && invstd && c1 && c2, "seg_cat_bn_bwd_scatter: bad arguments");
  CatDev cd;
  if (int rc = cat_to_dev(cat, cs, true, &cd)) return rc;
  const ColLayout L = col_layout(cs);
  int64_t blocks = (npix + (int64_t)L.RP * UNR - 1) / ((int64_t)L.RP * UNR);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL((cat_bn_bwd_scatter_kernel<T>), dim3((int)blocks), dim3(TPB), 0, ST(stream), (const T*)g, cd, npix, cs, scale, mean, invstd, c1, c2, L);
  SEG_CHECK_LAUNCH("seg_cat_bn_bwd_scatter");
  return SEG_OK;
}
extern "C" int seg_cat_bn_bwd_scatter(const void* g, const seg_cat_desc* cat, int64_t npix, int32_t cs, const float* scale, const float* mean, const float* invstd, const float* c1, const float* c2, seg_stream_t stream) { return seg_cat_bn_bwd_scatter_impl<half_t>(g, cat, npix, cs, scale, mean, invstd, c1, c2, stream); }
extern "C" int seg_cat_bn_bwd_scatter_f32(const void* g, const seg_cat_desc* cat, int64_t npix, int32_t cs, const float* scale, const float* mean, const float* invstd, const float* c1, const float* c2, seg_stream_t stream) { return seg_cat_bn_bwd_scatter_impl<split_t>(g, cat, npix, cs, scale, mean, invstd, c1, c2, stream); }

template <typename T>
static int seg_mul_bcast_fwd_impl(const void* a, const void* r, void* y, int64_t npix, int32_t cs, int32_t r_cs, seg_stream_t stream) {
  SEG_CHECK_ARG(a && r && y && npix > 0 && (cs & 7) == 0 && r_cs >= 1, "seg_mul_bcast_fwd: bad arguments");
  hipLaunchKernelGGL((mul_bcast_fwd_kernel<T>), dim3(grid_for(npix * (cs / 8))), dim3(TPB), 0, ST(stream), (const T*)a, (const T*)r,
                     (T*)y, npix, cs, r_cs);
  SEG_CHECK_LAUNCH("seg_mul_bcast_fwd");
  return SEG_OK;
}
extern "C" int seg_mul_bcast_fwd(const void* a, const void* r, void* y, int64_t npix, int32_t cs, int32_t r_cs, seg_stream_t stream) { return seg_mul_bcast_fwd_impl<half_t>(a, r, y, npix, cs, r_cs, stream); }
extern "C" int seg_mul_bcast_fwd_f32(const void* a, const void* r, void* y, int64_t npix, int32_t cs, int32_t r_cs, seg_stream_t stream) { return seg_mul_bcast_fwd_impl<split_t>(a, r, y, npix, cs, r_cs, stream); }

template <typename T>
static int seg_mul_bcast_bwd_impl(const void* dy, const void* a, const void* r, void* da, void* dr, int64_t npix, int32_t cs, int32_t r_cs,
                                 int32_t acc_a, int32_t acc_r, seg_stream_t stream) {
  SEG_CHECK_ARG(dy && a && r && npix > 0 && (cs & 7) == 0 && r_cs >= 1 && (sizeof(T) == 2 || (r_cs & 7) == 0), "seg_mul_bcast_bwd: bad arguments");
  int LPP = pow2ceil(cs / 8);
  if (LPP > 64) LPP = 64;
  const int gpb = TPB / LPP;
  int64_t blocks = (npix + gpb - 1) / gpb;
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
  hipLaunchKernelGGL((mul_bcast_bwd_kernel<T>), dim3((int)blocks), dim3(TPB), 0, ST(stream), (const T*)dy, (const T*)a, (const T*)r,
                     (T*)da, (T*)dr, npix, cs, r_cs, acc_a, acc_r, LPP);
  SEG_CHECK_LAUNCH("seg_mul_bcast_bwd");
  return SEG_OK;
}
extern "C" int seg_mul_bcast_bwd(const void* dy, const void* a, const void* r, void* da, void* dr, int64_t npix, int32_t cs, int32_t r_cs, int32_t acc_a, int32_t acc_r, seg_stream_t stream) { return seg_mul_bcast_bwd_impl<half_t>(dy, a, r, da, dr, npix, cs, r_cs, acc_a, acc_r, stream); }
extern "C" int seg_mul_bcast_bwd_f32(const void* dy, const void* a, const void* r, void* da, void* dr, int64_t npix, int32_t cs, int32_t r_cs, int32_t acc_a, int32_t acc_r, seg_stream_t stream) { return seg_mul_bcast_bwd_impl<split_t>(dy, a, r, da, dr, npix, cs, r_cs, acc_a, acc_r, stream); }

template <typename T>
static int seg_convlstm_gate_fwd_impl(const void* z, int32_t z_cs, void* h, int32_t h_cs, int64_t npix, int32_t F, seg_stream_t stream) {
  SEG_CHECK_ARG(z && h && npix > 0 && F > 0 && (F & 7) == 0 && 4 * F <= z_cs && F <= h_cs && (z_cs & 7) == 0 && (h_cs & 7) == 0,
                "seg_convlstm_gate_fwd: bad arguments (F=%d z_cs=%d h_cs=%d)", F, z_cs, h_cs);
  hipLaunchKernelGGL((convlstm_gate_fwd_kernel<T>), dim3(grid_for(npix * (F / 8))), dim3(TPB), 0, ST(stream), (const T*)z, z_cs, (T*)h,
                     h_cs, npix, F);
  SEG_CHECK_LAUNCH("seg_convlstm_gate_fwd");
  return SEG_OK;
}
extern "C" int seg_convlstm_gate_fwd(const void* z, int32_t z_cs, void* h, int32_t h_cs, int64_t npix, int32_t F, seg_stream_t stream) { return seg_convlstm_gate_fwd_impl<half_t>(z, z_cs, h, h_cs, npix, F, stream); }
extern "C" int seg_convlstm_gate_fwd_f32(const void* z, int32_t z_cs, void* h, int32_t h_cs, int64_t npix, int32_t F, seg_stream_t stream) { return seg_convlstm_gate_fwd_impl<split_t>(z, z_cs, h, h_cs, npix, F, stream); }

template <typename T>
static int seg_convlstm_gate_bwd_impl(const void* dh, int32_t h_cs, const void* z, int32_t z_cs, void* dz, int64_t npix, int32_t F,
                                     seg_stream_t stream) {
  SEG_CHECK_ARG(dh && z && dz && npix > 0 && F > 0 && (F & 7) == 0 && 4 * F <= z_cs && F <= h_cs, "seg_convlstm_gate_bwd: bad arguments");
  hipLaunchKernelGGL((convlstm_gate_bwd_kernel<T>), dim3(grid_for(npix * (F / 8))), dim3(TPB), 0, ST(stream), (const T*)dh, h_cs,
                     (const T*)z, z_cs, (T*)dz, npix, F);
  SEG_CHECK_LAUNCH("seg_convlstm_gate_bwd");
  return SEG_OK;
}
extern "C" int seg_convlstm_gate_bwd(const void* dh, int32_t h_cs, const void* z, int32_t z_cs, void* dz, int64_t npix, int32_t F, seg_stream_t stream) { return seg_convlstm_gate_bwd_impl<half_t>(dh, h_cs, z, z_cs, dz, npix, F, stream); }
extern "C" int seg_convlstm_gate_bwd_f32(const void* dh, int32_t h_cs, const void* z, int32_t z_cs, void* dz, int64_t npix, int32_t F, seg_stream_t stream) { return seg_convlstm_gate_bwd_impl<split_t>(dh, h_cs, z, z_cs, dz, npix, F, stream); }

static int head_lpp(int Cin) {
  int LPP = pow2ceil((Cin + 7) / 8);
  return LPP > 64 ? 64 : LPP;
}

template <typename T>
static int seg_head_fwd_impl(const void* x, int32_t cs, int32_t Cin, int32_t N, int32_t H, int32_t W, int32_t sy, int32_t sx, const float* w,
                            int32_t w_ld, const float* b, int32_t Cout, int32_t act, float* y, seg_stream_t stream) {
  SEG_CHECK_ARG(x && w && y && Cin > 0 && Cin <= cs && (cs & 7) == 0 && Cout >= 1 && Cout <= HEAD_MAXC && w_ld >= Cin && sy >= 1 && sx >= 1 &&
                    N > 0 && H > 0 && W > 0,
                "seg_head_fwd: bad arguments (Cin=%d cs=%d Cout=%d)", Cin, cs, Cout);
  const int LPP = head_lpp(Cin), gpb = TPB / LPP;
  const int64_t npix = (int64_t)N * ((H + sy - 1) / sy) * ((W + sx - 1) / sx);
  SEG_CHECK_ARG(npix < (1LL << 31), "seg_head_fwd: more than 2^31 output pixels");
  int64_t blocks = (npix + (int64_t)gpb * UNR - 1) / ((int64_t)gpb * UNR);
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
#define SEG_HEAD_CO(KERNEL, ...)                                                                        \
  do {                                                                                                  \
    if (Cout <= 1) hipLaunchKernelGGL((KERNEL<T, 1>), __VA_ARGS__);                                      \
    else if (Cout <= 2) hipLaunchKernelGGL((KERNEL<T, 2>), __VA_ARGS__);                                 \
    else if (Cout <= 4) hipLaunchKernelGGL((KERNEL<T, 4>), __VA_ARGS__);                                 \
    else hipLaunchKernelGGL((KERNEL<T, 8>), __VA_ARGS__);                                                \
  } while (0)
  SEG_HEAD_CO(head_fwd_kernel, dim3((int)blocks), dim3(TPB), 0, ST(stream), (const T*)x, cs, Cin, N, H, W, sy, sx, w, w_ld, b, Cout, act, y,
              LPP);
  SEG_CHECK_LAUNCH("seg_head_fwd");
  return SEG_OK;
}
extern "C" int seg_head_fwd(const void* x, int32_t cs, int32_t Cin, int32_t N, int32_t H, int32_t W, int32_t sy, int32_t sx, const float* w, int32_t w_ld, const float* b, int32_t Cout, int32_t act, float* y, seg_stream_t stream) { return seg_head_fwd_impl<half_t>(x, cs, Cin, N, H, W, sy, sx, w, w_ld, b, Cout, act, y, stream); }
extern "C" int seg_head_fwd_f32(const void* x, int32_t cs, int32_t Cin, int32_t N, int32_t H, int32_t W, int32_t sy, int32_t sx, const float* w, int32_t w_ld, const float* b, int32_t Cout, int32_t act, float* y, seg_stream_t stream) { return seg_head_fwd_impl<split_t>(x, cs, Cin, N, H, W, sy, sx, w, w_ld, b, Cout, act, y, stream); }

template <typename T>
static int seg_head_bwd_impl(const float* dy, const float* y, const void* x, int32_t cs, int32_t Cin, int32_t N, int32_t H, int32_t W,
                            int32_t sy, int32_t sx, const float* w, int32_t w_ld, int32_t Cout, int32_t act, void* dx, int32_t accumulate,
                            const float* gs_ptr, float* dw, float* db, float* work, int64_t work_elems, seg_stream_t stream) {
  SEG_CHECK_ARG(dy && y && x && w && Cin > 0 && Cin <= cs && (cs & 7) == 0 && Cout >= 1 && Cout <= HEAD_MAXC && w_ld >= Cin,
                "seg_head_bwd: bad arguments");
  const int64_t npix = (int64_t)N * ((H + sy - 1) / sy) * ((W + sx - 1) / sx);
  SEG_CHECK_ARG(npix < (1LL << 31), "seg_head_bwd: more than 2^31 output pixels");
  if (dx != nullptr) {
    const int LPP = head_lpp(cs), gpb = TPB / LPP;
    int64_t blocks = (npix + (int64_t)gpb * UNR - 1) / ((int64_t)gpb * UNR);
    if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    SEG_HEAD_CO(head_bwd_dx_kernel, dim3((int)blocks), dim3(TPB), 0, ST(stream), dy, y, cs, Cin, N, H, W, sy, sx, w, w_ld, Cout, act, (T*)dx,
                accumulate, gs_ptr, LPP);
  }
  if (dw != nullptr) {
    // with a workspace: up to work_elems / (Cout * (w_ld + 1)) blocks (1024 fill the chip), partial rows + a finalize
    // launch; without: every block ends with Cout*Cin same-address atomics, so the count is kept low
    int blocks = (int)((npix + 1023) / 1024);
    int cap = 256;
    if (work != nullptr) {
      cap = (int)(work_elems / ((int64_t)Cout * (w_ld + 1)));
      if (cap > 1024) cap = 1024;
      SEG_CHECK_ARG(cap >= 1, "seg_head_bwd: workspace of %ld floats is too small", (long)work_elems);
      blocks = (int)((npix + 255) / 256);
    }
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    SEG_HEAD_CO(head_bwd_dw_kernel, dim3(blocks), dim3(TPB), 0, ST(stream), dy, y, (const T*)x, cs, Cin, N, H, W, sy, sx, w_ld, Cout, act, dw,
                db, work, col_layout(((Cin + 7) / 8) * 8));
    if (work != nullptr)
      hipLaunchKernelGGL(head_bwd_dw_finalize_kernel, dim3((Cout * w_ld + Cout + 15) / 16), dim3(TPB), 0, ST(stream), work, blocks, Cin,
                         w_ld, Cout, dw, db);
  }
  SEG_CHECK_LAUNCH("seg_head_bwd");
  return SEG_OK;
}
extern "C" int seg_head_bwd(const float* dy, const float* y, const void* x, int32_t cs, int32_t Cin, int32_t N, int32_t H, int32_t W, int32_t sy, int32_t sx, const float* w, int32_t w_ld, int32_t Cout, int32_t act, void* dx, int32_t accumulate, const float* gs_ptr, float* dw, float* db, float* work, int64_t work_elems, seg_stream_t stream) { return seg_head_bwd_impl<half_t>(dy, y, x, cs, Cin, N, H, W, sy, sx, w, w_ld, Cout, act, dx, accumulate, gs_ptr, dw, db, work, work_elems, stream); }
extern "C" int seg_head_bwd_f32(const float* dy, const float* y, const void* x, int32_t cs, int32_t Cin, int32_t N, int32_t H, int32_t W, int32_t sy, int32_t sx, const float* w, int32_t w_ld, int32_t Cout, int32_t act, void* dx, int32_t accumulate, const float* gs_ptr, float* dw, float* db, float* work, int64_t work_elems, seg_stream_t stream) { return seg_head_bwd_impl<split_t>(dy, y, x, cs, Cin, N, H, W, sy, sx, w, w_ld, Cout, act, dx, accumulate, gs_ptr, dw, db, work, work_elems, stream); }

template <typename T>
static int seg_head_act_fwd_impl(const void* x, int32_t cs, int64_t npix, int32_t C, int32_t act, float* y, seg_stream_t stream) {
  SEG_CHECK_ARG(x && y && npix > 0 && C >= 1 && cs >= C && act >= SEG_ACT_NONE && act <= SEG_ACT_SOFTMAX, "seg_head_act_fwd: bad arguments");
  hipLaunchKernelGGL((head_act_fwd_kernel<T>), dim3(grid_for(npix)), dim3(TPB), 0, ST(stream), (const T*)x, cs, npix, C, act, y);
  SEG_CHECK_LAUNCH("seg_head_act_fwd");
  return SEG_OK;
}
extern "C" int seg_head_act_fwd(const void* x, int32_t cs, int64_t npix, int32_t C, int32_t act, float* y, seg_stream_t stream) { return seg_head_act_fwd_impl<half_t>(x, cs, npix, C, act, y, stream); }
extern "C" int seg_head_act_fwd_f32(const void* x, int32_t cs, int64_t npix, int32_t C, int32_t act, float* y, seg_stream_t stream) { return seg_head_act_fwd_impl<split_t>(x, cs, npix, C, act, y, stream); }

template <typename T>
static int seg_head_act_bwd_impl(const float* dy, const float* y, int64_t npix, int32_t C, int32_t act, const float* gs_ptr, void* dx, int32_t cs,
                                 int32_t accumulate, seg_stream_t stream) {
  SEG_CHECK_ARG(dy && y && dx && npix > 0 && C >= 1 && cs >= C && act >= SEG_ACT_NONE && act <= SEG_ACT_SOFTMAX, "seg_head_act_bwd: bad arguments");
  hipLaunchKernelGGL((head_act_bwd_kernel<T>), dim3(grid_for(npix)), dim3(TPB), 0, ST(stream), dy, y, npix, C, act, gs_ptr, (T*)dx, cs, accumulate);
  SEG_CHECK_LAUNCH("seg_head_act_bwd");
  return SEG_OK;
}
extern "C" int seg_head_act_bwd(const float* dy, const float* y, int64_t npix, int32_t C, int32_t act, const float* gs_ptr, void* dx, int32_t cs, int32_t accumulate, seg_stream_t stream) { return seg_head_act_bwd_impl<half_t>(dy, y, npix, C, act, gs_ptr, dx, cs, accumulate, stream); }
extern "C" int seg_head_act_bwd_f32(const float* dy, const float* y, int64_t npix, int32_t C, int32_t act, const float* gs_ptr, void* dx, int32_t cs, int32_t accumulate, seg_stream_t stream) { return seg_head_act_bwd_impl<split_t>(dy, y, npix, C, act, gs_ptr, dx, cs, accumulate, stream); }

extern "C" int seg_loss_fwd_bwd_w(int32_t kind, const float* pred, const float* target, int64_t npix, int32_t C, float weight,
                                  const int32_t* binary_flag, float* loss_out, float* dpred, seg_stream_t stream) {
  SEG_CHECK_ARG(pred && target && loss_out && npix > 0 && C >= 1 && kind >= SEG_LOSS_MSE && kind <= SEG_LOSS_SPARSE_CCE,
                "seg_loss_fwd_bwd: bad arguments");
  const bool pp = kind == SEG_LOSS_CCE || kind == SEG_LOSS_KLD || kind == SEG_LOSS_CATHINGE || kind == SEG_LOSS_COSINE || kind == SEG_LOSS_SPARSE_CCE;
  const double n = pp ? (double)npix : (double)npix * C;
  const int64_t items = pp ? npix : npix * C;
  int blocks = grid_for(items);
  if (blocks > 1024) blocks = 1024;      // every block ends with one atomic on the same loss word
  hipLaunchKernelGGL(loss_kernel, dim3(blocks), dim3(TPB), 0, ST(stream), kind, pred, target, npix, C, (float)(1.0 / n), weight,
                     (const int*)binary_flag, loss_out, dpred);
  SEG_CHECK_LAUNCH("seg_loss_fwd_bwd");
  return SEG_OK;
}

extern "C" int seg_loss_fwd_bwd(int32_t kind, const float* pred, const float* target, int64_t npix, int32_t C, float* loss_out, float* dpred,
                                seg_stream_t stream) {
  return seg_loss_fwd_bwd_w(kind, pred, target, npix, C, 1.f, nullptr, loss_out, dpred, stream);
}

extern "C" int seg_binary_counts(const float* pred, const float* target, int64_t n, float threshold, uint64_t* counts, seg_stream_t stream) {
  SEG_CHECK_ARG(pred && target && counts && n > 0, "seg_binary_counts: bad arguments");
  int blocks = grid_for(n);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(binary_counts_kernel, dim3(blocks), dim3(TPB), 0, ST(stream), pred, target, n, threshold, (unsigned long long*)counts);
  SEG_CHECK_LAUNCH("seg_binary_counts");
  return SEG_OK;
}

extern "C" int seg_argmax_match(const float* pred, const float* target, int64_t npix, int32_t C, int32_t sparse_target, uint64_t* counts,
                                seg_stream_t stream) {
  SEG_CHECK_ARG(pred && target && counts && npix > 0 && C >= 1, "seg_argmax_match: bad arguments");
  int blocks = grid_for(npix);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(argmax_match_kernel, dim3(blocks), dim3(TPB), 0, ST(stream), pred, target, npix, C, sparse_target,
                     (unsigned long long*)counts);
  SEG_CHECK_LAUNCH("seg_argmax_match");
  return SEG_OK;
}

extern "C" int seg_labels_binary(const float* target, int64_t n, int32_t* flag, seg_stream_t stream) {
  SEG_CHECK_ARG(target && flag && n > 0, "seg_labels_binary: bad arguments");
  hipLaunchKernelGGL(set_flag_kernel, dim3(1), dim3(1), 0, ST(stream), (int*)flag);
  int blocks = grid_for(n);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(labels_binary_kernel, dim3(blocks), dim3(TPB), 0, ST(stream), target, n, (int*)flag);
  SEG_CHECK_LAUNCH("seg_labels_binary");
  return SEG_OK;
}

extern "C" int seg_absmax(const float* x, int64_t n, float* amax, seg_stream_t stream) {
  SEG_CHECK_ARG(x && amax && n > 0, "seg_absmax: bad arguments");
  int blocks = grid_for(n / 4 + 1);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(TPB), 0, ST(stream), x, n, amax);
  SEG_CHECK_LAUNCH("seg_absmax");
  return SEG_OK;
}

extern "C" int seg_grad_scale(const float* amax, float target, float* gs, seg_stream_t stream) {
  SEG_CHECK_ARG(amax && gs && target > 0.f, "seg_grad_scale: bad arguments");
  hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(1), 0, ST(stream), amax, target, gs);
  SEG_CHECK_LAUNCH("seg_grad_scale");
  return SEG_OK;
}

template <typename T>
static int seg_transpose_w_impl(const float* w, void* wt, int32_t rows, int32_t Tn, int32_t K, int32_t rows_pad, seg_stream_t stream) {
  SEG_CHECK_ARG(w && wt && rows > 0 && Tn > 0 && K > 0 && rows_pad >= rows, "seg_transpose_w: bad arguments");
  hipLaunchKernelGGL((transpose_w_kernel<T>), dim3(((K + 31) / 32) * ((rows_pad + 31) / 32), Tn), dim3(TPB), 0, ST(stream), w, (T*)wt,
                     rows, Tn, K, rows_pad);
  SEG_CHECK_LAUNCH("seg_transpose_w");
  return SEG_OK;
}
extern "C" int seg_transpose_w(const float* w, void* wt, int32_t rows, int32_t Tn, int32_t K, int32_t rows_pad, seg_stream_t stream) { return seg_transpose_w_impl<half_t>(w, wt, rows, Tn, K, rows_pad, stream); }
extern "C" int seg_transpose_w_f32(const float* w, void* wt, int32_t rows, int32_t Tn, int32_t K, int32_t rows_pad, seg_stream_t stream) { return seg_transpose_w_impl<split_t>(w, wt, rows, Tn, K, rows_pad, stream); }

template <typename T>
static int seg_transpose_w_multi_impl(const seg_transpose_desc* descs, int32_t n, int32_t total_tiles, seg_stream_t stream) {
  SEG_CHECK_ARG(descs && n > 0 && total_tiles > 0, "seg_transpose_w_multi: bad arguments");
  hipLaunchKernelGGL((transpose_w_multi_kernel<T>), dim3(total_tiles), dim3(TPB), 0, ST(stream), descs, n);
  SEG_CHECK_LAUNCH("seg_transpose_w_multi");
  return SEG_OK;
}
extern "C" int seg_transpose_w_multi(const seg_transpose_desc* descs, int32_t n, int32_t total_tiles, seg_stream_t stream) { return seg_transpose_w_multi_impl<half_t>(descs, n, total_tiles, stream); }
extern "C" int seg_transpose_w_multi_f32(const seg_transpose_desc* descs, int32_t n, int32_t total_tiles, seg_stream_t stream) { return seg_transpose_w_multi_impl<split_t>(descs, n, total_tiles, stream); }

extern "C" int seg_cast_f32_f16(const float* x, void* y, int64_t n, seg_stream_t stream) {
  SEG_CHECK_ARG(x && y && n > 0, "seg_cast_f32_f16: bad arguments");
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n)), dim3(TPB), 0, ST(stream), x, (half_t*)y, n);
  SEG_CHECK_LAUNCH("seg_cast_f32_f16");
  return SEG_OK;
}

extern "C" int seg_cast_f32_split(const float* x, void* y, int64_t n, seg_stream_t stream) {
  SEG_CHECK_ARG(x && y && n > 0 && (n & 7) == 0 && ((uintptr_t)x & 31) == 0 && ((uintptr_t)y & 31) == 0,
                "seg_cast_f32_split: bad arguments (n a multiple of 8, 32-B aligned buffers)");
  hipLaunchKernelGGL(cast_split_kernel, dim3(grid_for(n / 8)), dim3(TPB), 0, ST(stream), x, (split_t*)y, n / 8);
  SEG_CHECK_LAUNCH("seg_cast_f32_split");
  return SEG_OK;
}

template <typename T>
static int seg_pack_input_impl(const float* x, void* y, int64_t npix, int32_t C, int32_t cs, int32_t* overflow, seg_stream_t stream) {
  SEG_CHECK_ARG(x && y && npix > 0 && C > 0 && C <= cs, "seg_pack_input: bad arguments");
  hipLaunchKernelGGL((pack_input_kernel<T>), dim3(grid_for(npix * cs)), dim3(TPB), 0, ST(stream), x, (T*)y, npix, C, cs, overflow);
  SEG_CHECK_LAUNCH("seg_pack_input");
  return SEG_OK;
}
extern "C" int seg_pack_input(const float* x, void* y, int64_t npix, int32_t C, int32_t cs, int32_t* overflow, seg_stream_t stream) { return seg_pack_input_impl<half_t>(x, y, npix, C, cs, overflow, stream); }
extern "C" int seg_pack_input_f32(const float* x, void* y, int64_t npix, int32_t C, int32_t cs, int32_t* overflow, seg_stream_t stream) { return seg_pack_input_impl<split_t>(x, y, npix, C, cs, overflow, stream); }

template <typename T>
static int seg_unpack_f32_impl(const void* x, float* y, int64_t npix, int32_t C, int32_t cs, int32_t coff, seg_stream_t stream) {
  SEG_CHECK_ARG(x && y && npix > 0 && C > 0 && coff + C <= cs, "seg_unpack_f32: bad arguments");
  hipLaunchKernelGGL((unpack_kernel<T>), dim3(grid_for(npix * C)), dim3(TPB), 0, ST(stream), (const T*)x, y, npix, C, cs, coff);
  SEG_CHECK_LAUNCH("seg_unpack_f32");
  return SEG_OK;
}
extern "C" int seg_unpack_f32(const void* x, float* y, int64_t npix, int32_t C, int32_t cs, int32_t coff, seg_stream_t stream) { return seg_unpack_f32_impl<half_t>(x, y, npix, C, cs, coff, stream); }
extern "C" int seg_unpack_f32_f32(const void* x, float* y, int64_t npix, int32_t C, int32_t cs, int32_t coff, seg_stream_t stream) { return seg_unpack_f32_impl<split_t>(x, y, npix, C, cs, coff, stream); }

template <typename T>
static int seg_copy_channels_impl(const void* src, int32_t src_cs, int32_t src_coff, void* dst, int32_t dst_cs, int32_t dst_coff, int32_t C,
                                  int64_t npix, int32_t accumulate, seg_stream_t stream) {
  SEG_CHECK_ARG(src && dst && C > 0 && npix > 0 && src_coff >= 0 && dst_coff >= 0 && src_coff + C <= src_cs && dst_coff + C <= dst_cs,
                "seg_copy_channels: bad arguments (C=%d src %d+%d/%d dst %d+%d/%d)", C, src_coff, C, src_cs, dst_coff, C, dst_cs);
  hipLaunchKernelGGL((copy_channels_kernel<T>), dim3(grid_for(npix * C)), dim3(TPB), 0, ST(stream), (const T*)src, src_cs, src_coff, (T*)dst,
                     dst_cs, dst_coff, C, npix, accumulate);
  SEG_CHECK_LAUNCH("seg_copy_channels");
  return SEG_OK;
}
extern "C" int seg_copy_channels(const void* src, int32_t src_cs, int32_t src_coff, void* dst, int32_t dst_cs, int32_t dst_coff, int32_t C, int64_t npix, int32_t accumulate, seg_stream_t stream) { return seg_copy_channels_impl<half_t>(src, src_cs, src_coff, dst, dst_cs, dst_coff, C, npix, accumulate, stream); }
extern "C" int seg_copy_channels_f32(const void* src, int32_t src_cs, int32_t src_coff, void* dst, int32_t dst_cs, int32_t dst_coff, int32_t C, int64_t npix, int32_t accumulate, seg_stream_t stream) { return seg_copy_channels_impl<split_t>(src, src_cs, src_coff, dst, dst_cs, dst_coff, C, npix, accumulate, stream); }
