// Shared pieces of the implicit-GEMM translation units (igemm_fwd.hip: register-staged and LDS-DMA per-tap kernels;
// igemm_halo.hip: halo-tile forward / data gradient; igemm_wgrad.hip: weight gradients): MFMA K steps, the XCD-aware block
// order, the forward epilogue (bias, BN partial statistics, activation, split outputs) and the host-side argument checks.
// Everything lives in an unnamed namespace: each translation unit gets its own copy.
#pragma once
#include "seg_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int ROW_BYTES = 64;     // one LDS tile row = one K step = 4 chunks of 16 B (32 halves or 16 floats)
template <typename T> struct ET;
template <> struct ET<half_t> { static constexpr int EPC = 8; };   // elements per 16-B chunk

// One K step (64 B per row = 32 fp16 channels) of MFMAs for a (TM x TN) grid of 32x32 tiles.  A/B point at the rows of this
// wave's tiles; lane (fr = lane & 31, fh = lane >> 5) reads row fr: two v_mfma_f32_32x32x16_f16 (k = 8 fh + j).
template <int TM, int TN>
__device__ __forceinline__ void mma_step(const unsigned char* A, const unsigned char* B, int a_row0, int b_row0, int fr, int fh,
                                         f16v (&acc)[TM][TN]) {
  const int fswz = (fr >> 2) & 3;
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
}
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ ((row >> 2) & 3)) << 4); }

// ---- parity mode: operands STORED as split pairs (seg_common.h, split_t), x = hi + lo / 2048.  To every loader a split
// tensor is an fp16 tensor with twice the channels ("pseudo-channels": an 8-channel group = 16 pseudo-channels, [8 hi | 8 lo]) -
// the launchers double seg_c / seg_cs / seg_woff / Kpad and the fp16 loaders run unchanged.  A 64-B LDS row is then 16 real K
// values as chunks (hi 0-7, lo 0-7, hi 8-15, lo 8-15): lane half fh contracts k = 8 fh .. 8 fh + 7, its hi fragment is chunk
// 2 fh, its lo fragment chunk 2 fh + 1.  Products of fp16 values are exact in the fp32 accumulator:
// a b ~= ah bh + (al bh + ah bl) / 2048 (al bl, 2^-22 relative, is dropped): three MFMAs per product, the two sums in separate
// accumulators that are combined once at the end (split_combine).
// BHI (data gradients, SEG_VARIANT_B_HI): the B operand - the transposed weights - as its hi half: the `bl` reads and the `ah bl` MFMA go.
template <int TM, int TN, bool BHI = false>
__device__ __forceinline__ void mma_step_split(const unsigned char* A, const unsigned char* B, int a_row0, int b_row0, int fr, int fh,
                                               f16v (&acc)[TM][TN], f16v (&accl)[TM][TN]) {
  const int fswz = (fr >> 2) & 3;
  const int ch = ((2 * fh) ^ fswz) << 4, cl = ((2 * fh + 1) ^ fswz) << 4;
  h8 ah[TM], al[TM], bh[TN], bl[BHI ? 1 : TN];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const unsigned char* r = A + (a_row0 + a * 32 + fr) * ROW_BYTES;
    ah[a] = *reinterpret_cast<const h8*>(r + ch);
    al[a] = *reinterpret_cast<const h8*>(r + cl);
  }
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const unsigned char* r = B + (b_row0 + b * 32 + fr) * ROW_BYTES;
    bh[b] = *reinterpret_cast<const h8*>(r + ch);
    if constexpr (!BHI) bl[b] = *reinterpret_cast<const h8*>(r + cl);
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
      accl[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], accl[a][b], 0, 0, 0);
      if constexpr (!BHI) accl[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], accl[a][b], 0, 0, 0);
    }
}
template <int TM, int TN>
__device__ __forceinline__ void split_combine(f16v (&acc)[TM][TN], const f16v (&accl)[TM][TN]) {
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] += accl[a][b][q] * SPLIT_INV;
}
// host side: the argument block as the loaders of a split launch see it (pseudo-channels); the output side (n_rows, out_cs,
// spans: the epilogue's units) stays in real channels
template <typename Args>
inline Args split_view(const Args& a) {
  Args b = a;
  for (int s = 0; s < a.nseg; ++s) { b.seg_c[s] *= 2; b.seg_cs[s] *= 2; b.seg_woff[s] *= 2; }
  b.Kpad *= 2;
  return b;
}

constexpr int STG_LD = 36;        // floats per staging row (32 + 4 pad, keeps 16-B alignment)
constexpr int STG_BYTES = 4 * 32 * STG_LD * 4;   // 4 waves

// XCD-aware block order (MI355X: 8 XCDs with private L2s, workgroups dealt round-robin over them): physical
// block b runs on XCD b % 8; give each XCD one contiguous range of logical tiles so that neighbouring tiles (which
// share halo rows / the dz tile / weight panels) meet in the same L2.  Bijective for any block count.
// ---- LDS-DMA issued from inline assembly.  The compiler models ds_read_b64_tr_b16 as a possible LDS WRITE and orders it
// behind every LDS-DMA it knows to be in flight with an s_waitcnt vmcnt(0) of its own - which makes the loads of the
// weight-gradient kernels synchronous (issue next patch, wait for it, only then contract the current one).  Loads the
// compiler cannot see are waited for where the kernel says so (counted vmcnt + s_barrier).
typedef int i4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i4v make_rsrc(const void* base, int bytes) {       // raw buffer, bounds-checked against `bytes`
  const unsigned long long v = (unsigned long long)base;
  i4v r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) & 0xffff;
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000;
  return r;
}
// 64 lanes x 16 bytes -> 1 KiB of LDS at lds_addr (wave-uniform); lane i fetches rsrc[voff + soff]
__device__ __forceinline__ void lds_dma16(i4v rsrc, unsigned lds_addr, unsigned voff, int soff) {
#if defined(SEG_DMA_BUILTIN)      // bisecting aid: the compiler-visible form (synchronous in the kernels with transposing reads)
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)(unsigned)rsrc.y << 32) | (unsigned)rsrc.x), 0, rsrc.z, rsrc.w), (lds_ptr_t)(size_t)lds_addr, 16, voff, soff, 0, 0);
#else
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
#endif
}

// ---- the ONLY two places of this code base that execute s_barrier in a ring kernel.
// ring_release_barrier: a barrier after which other waves may refill the LDS ring slot this wave has just read.  gfx950 puts no
// implicit wait in front of s_barrier and hipcc is free to leave the last MFMAs of a step - with the wait for their fragment reads -
// behind it, so the wave's own LDS reads must have RETURNED (s_waitcnt lgkmcnt(0)), not just been issued: with the LDS backed up
// by a co-resident kernel a queued read otherwise returns the next step's bytes (one 32 x 32 tile a K step off in 1-10 % of the
// C3 training steps until the end of round 2; docs/experiments.md section 3).  The waits for the wave's LDS-DMA (counted vmcnt) stay with
// the caller, in front of this call.  with_barrier = false: timing experiments only (tools/op_bench.py --dbg).
__device__ __forceinline__ void ring_release_barrier(bool with_barrier = true) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (with_barrier) __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// loader_barrier: the same rendezvous for pure loader waves of the wave-specialised kernels, which read nothing from LDS
__device__ __forceinline__ void loader_barrier(bool with_barrier = true) {
  if (with_barrier) __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}


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
template <> struct Raw8<split_t> {
  struct type { u4 hi, lo; };
  static __device__ __forceinline__ void zero(type& r) { r.hi = u4{0u, 0u, 0u, 0u}; r.lo = r.hi; }
  static __device__ __forceinline__ void load(const split_t* p, type& r) { r.hi = ld16(p); r.lo = ld16(p + 4); }
  static __device__ __forceinline__ void add(const type& r, float* f) {
    const h8 hi = __builtin_bit_cast(h8, r.hi), lo = __builtin_bit_cast(h8, r.lo);
#pragma unroll
    for (int t = 0; t < 8; ++t) f[t] += (float)hi[t] + (float)lo[t] * SPLIT_INV;
  }
};

// ---------------------------------------------------------------------------------------------------------
// shared epilogue of the forward kernels: bias, BN partial statistics, activation, channel-padding mask, then each
// 32x32 accumulator tile goes through a per-wave LDS slab so that global stores are 8 channels (16 / 32 B) per lane.
// row0 = first logical pixel of this wave's rows, col0 = first column of this wave inside the block's BN columns.
// ---------------------------------------------------------------------------------------------------------
// PWE = 0: this wave's rows are the logical pixels row0 .. ; PWE = 32 / 16: the rows are a 256-pixel patch of width
// PWE at (image pn, row py0, column px0) and row0 is the wave's first patch row (the halo-tile kernel).
template <typename T, int TM, int TN, int NWM, int BN, int PWE = 0, int NWV = 4>
__device__ __forceinline__ void conv_epilogue(const seg_igemm_fwd_args& p, f16v (&acc)[TM][TN], unsigned char* smem, int tid, int row0,
                                              int col0, int n0, int wm, int cls, int mblocks, int mb, int M, int HW,
                                              int pn = 0, int py0 = 0, int px0 = 0) {
  const seg_geom& g = p.g;
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  // the last MFMAs of the K loop retire before anything below reads an accumulator (the statistics are accumulated by
  // inline assembly, which the compiler's hazard tracking does not look into)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * STG_LD);
  float* sst = reinterpret_cast<float*>(smem + (STG_BYTES / 4) * NWV);   // [NWM][BN][2], behind the NWV staging slabs
  const float* bias = p.bias;
  T* out = (T*)p.out;
  const bool do_stats = p.stats != nullptr;
  const float act_slope = p.act == SEG_ACT_LEAKY ? p.alpha : 0.f;      // ReLU / LeakyReLU: v > 0 ? v : v * slope
  float s_sum[TN], s_sq[TN], bvs[TN], svs[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) {                // (the bias loads go out first: their latency passes behind the set-up below
    s_sum[b] = 0.f;                             // instead of being waited out in front of every tile)
    s_sq[b] = 0.f;
    const int ncol = n0 + col0 + b * 32 + fr;
    bvs[b] = (bias != nullptr && ncol < p.n_rows) ? bias[ncol] : 0.f;
    svs[b] = (p.ep_scale != nullptr && ncol < p.n_rows) ? p.ep_scale[ncol] : 1.f;      // folded inference-mode BatchNormalization
  }
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
  constexpr int AG = sizeof(T) == 4 ? 1 : (TM > 2 ? 2 : TM);      // (split pairs are twice the registers: one tile row at a time)
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
      const float bv = bvs[b], sv = svs[b];
      const bool nzero = ncol >= p.n_valid;
      // (block-uniform choices - statistics, kind of activation - are taken once per tile, not once per element: the
      // element loop used to carry ~14 scalar branches, a third of the run time of the short-K layers' blocks)
      float vv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) vv[q] = acc[a][b][q] * sv + bv;
      if (do_stats) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = row0 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh;
          const float v = (PWE > 0 || m < M) ? vv[q] : 0.f;
          // written out: left to the compiler this pair becomes v_pk_mul_f32 / v_pk_add_f32 on (v, v * v) register pairs, and
          // with those the SUM OF SQUARES came out up to 0.4 % off in lanes 16-31 / 48-63 of a few blocks per launch,
          // differently from run to run (the sums stayed exact) - tools/stats_probe.py
          asm("v_add_f32 %0, %0, %1" : "+v"(s_sum[b]) : "v"(v));
          asm("v_fmac_f32 %0, %1, %1" : "+v"(s_sq[b]) : "v"(v));
        }
      }
      if (p.act == SEG_ACT_SIGMOID) {
#pragma unroll
        for (int q = 0; q < 16; ++q) vv[q] = 1.f / (1.f + __expf(-vv[q]));
      } else if (p.act != SEG_ACT_NONE) {
#pragma unroll
        for (int q = 0; q < 16; ++q) vv[q] = vv[q] > 0.f ? vv[q] : vv[q] * act_slope;
      }
      if (nzero) {
#pragma unroll
        for (int q = 0; q < 16; ++q) vv[q] = 0.f;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) stg[((q & 3) + 8 * (q >> 2) + 4 * fh) * STG_LD + fr] = vv[q];
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
            if (dact[b] == SEG_ACT_SIGMOID) {
#pragma unroll
              for (int t = 0; t < 8; ++t) f[t] *= yf[t] * (1.f - yf[t]);
            } else if (dact[b] != SEG_ACT_NONE) {
              const float lo = dact[b] == SEG_ACT_LEAKY ? dalpha[b] : 0.f;
#pragma unroll
              for (int t = 0; t < 8; ++t) f[t] *= yf[t] > 0.f ? 1.f : lo;
            }
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

// ---- epilogue of the weight-gradient kernels: one 32 x 32 fp32 accumulator tile -> dw (or the split's partial buffer).
// How the tile is written is a block-uniform choice: made once (wgrad_mode), not per element; the element address is
// a tile base plus a scalar multiple of the row stride.
//   0 store (only writer, dw known to be zero)   1 read-add-store (only writer)   2 store into the partial buffer of this
//   patch range (deterministic split, reduced in order by wgrad_reduce_kernel)    3 atomicAdd
__device__ __forceinline__ int wgrad_mode(const seg_igemm_wgrad_args& p) {
  return p.splitk == 1 ? (p.overwrite ? 0 : 1) : ((p.partial != nullptr) ? 2 : 3);
}
// d0: element (first row of the tile + 4 (lane >> 5), this lane's column); rows_left = n_rows - first row of the tile (n_rows
// is a multiple of 8, so each group of eight rows is inside or outside as a whole); stride = floats between consecutive rows
template <int MODE>
__device__ __forceinline__ void wgrad_store_tile(float* d0, const f16v& acc, int rows_left, int stride, float sc_mul) {
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    if (8 * gq < rows_left) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* d = d0 + (8 * gq + j) * stride;
        const float v = acc[4 * gq + j];
        if constexpr (MODE == 0) *d = v * sc_mul;
        else if constexpr (MODE == 1) *d += v * sc_mul;
        else if constexpr (MODE == 2) *d = v;
        else atomicAdd(d, v * sc_mul);
      }
    }
  }
}
template <typename F>
__device__ __forceinline__ void wgrad_mode_dispatch(int mode, F&& f) {
  if (mode == 0) f(std::integral_constant<int, 0>{});
  else if (mode == 1) f(std::integral_constant<int, 1>{});
  else if (mode == 2) f(std::integral_constant<int, 2>{});
  else f(std::integral_constant<int, 3>{});
}

// Operands the LDS-DMA kernels address through one buffer resource with 32-bit byte offsets must stay below LDS_DMA_LIMIT
// (seg_lds_dma_operand_limit(): 1.875 GiB).  A masked lane fetches at OOB_OFF = 2 GiB: at or beyond every legal num_records even
// after a (negative) tap shift of up to 128 MiB has been added to it, and OOB_OFF + the largest scalar offset still below 4 GiB -
// the hardware range check then returns zeros.  Larger operands take the register-staged kernels (64-bit addresses): the host
// planner (planning.plan_fwd / ops.ConvOp) mirrors the limit.
constexpr unsigned OOB_OFF = 0x80000000u;
constexpr long LDS_DMA_LIMIT = 0x78000000L;
// The per-tap LDS-DMA kernels add a tap's byte shift (tap_dy * Win + tap_dx) * cs * esz to a lane offset that may already be OOB_OFF
// (channel padding): the sum must stay at or beyond every legal num_records, i.e. the largest shift below OOB_OFF - LDS_DMA_LIMIT
// (128 MiB: a row of Win * cs >= 32 M elements) - otherwise a masked lane would fetch live data instead of zeros.  Checked by
// fwd2_eligible() and mirrored by planning.fits_lds_dma(); such a geometry takes the register-staged kernels (64-bit addresses).
constexpr long LDS_DMA_MAX_SHIFT = (long)OOB_OFF - LDS_DMA_LIMIT;
inline long max_tap_shift_elems(const seg_geom& g) {
  long m = 0;
  for (int e = 0; e < g.ncls * g.ntaps && e < SEG_MAX_TAPS; ++e) {
    const long d = (long)g.tap_dy[e] * g.Win + g.tap_dx[e];
    m = d < 0 ? (m < -d ? -d : m) : (m < d ? d : m);
  }
  return m;
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
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

inline int check_wgrad_args(const seg_igemm_wgrad_args* a, const char* who) {
  SEG_CHECK_ARG(a != nullptr, "%s: null args", who);
  if (!check_geom(a->g, who) || !check_segs(*a, who)) return SEG_ERR_ARG;
  SEG_CHECK_ARG(a->dz && a->dw, "%s: null pointer", who);
  SEG_CHECK_ARG(a->n_rows > 0 && (a->n_rows & 7) == 0 && (a->dz_cs & 7) == 0 && (a->dz_coff & 7) == 0 && a->dz_coff + a->n_rows <= a->dz_cs,
                "%s: n_rows=%d dz_cs=%d dz_coff=%d", who, a->n_rows, a->dz_cs, a->dz_coff);
  SEG_CHECK_ARG((a->Kpad & 31) == 0, "%s: Kpad=%d", who, a->Kpad);
  SEG_CHECK_ARG(a->partial == nullptr || (((uintptr_t)a->partial | (uintptr_t)a->dw) & 15) == 0, "%s: partial / dw must be 16-byte aligned for the split reduce", who);
  SEG_CHECK_ARG((a->tile_m == 128 || a->tile_m == 64) && (a->tile_n == 128 || a->tile_n == 64), "%s: tile %dx%d", who, a->tile_m, a->tile_n);
  SEG_CHECK_ARG(a->splitk >= 1 && a->splitk <= 65535, "%s: splitk=%d", who, a->splitk);
  SEG_CHECK_ARG((a->variant & 255) >= 0 && (a->variant & 255) <= 2, "%s: variant=%d", who, a->variant);
  return SEG_OK;
}

}  // namespace

// second pass of the deterministic pixel split of the weight-gradient kernels (igemm_wgrad.hip)
int seg_launch_wgrad_reduce(const seg_igemm_wgrad_args& a, hipStream_t st);
// halo-tile (variant 3) forward launcher, igemm_halo.hip
int seg_launch_fwd3(const seg_igemm_fwd_args& a, hipStream_t st);
int seg_launch_fwd3_split(const seg_igemm_fwd_args& v, hipStream_t st);      // parity mode; v = split_view(args)
// its software-pipelined generation (igemm_halo2.hip); called by seg_launch_fwd3 after the argument checks
int seg_launch_fwd4(const seg_igemm_fwd_args& a, int pw, dim3 grid, hipStream_t st);
// wave-specialised halo-tile weight gradient (igemm_wgrad2.hip); called by seg_igemm_wgrad after the argument checks
int seg_launch_wgrad_halo_ws(const seg_igemm_wgrad_args& a, int pc, dim3 grid, hipStream_t st);
int seg_launch_wgrad_halo_ts(const seg_igemm_wgrad_args& a, int pc, dim3 grid, hipStream_t st);   // igemm_wgrad3.hip
