// Shared device/host helpers for the gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "seg_hip.h"

typedef _Float16 half_t;
typedef half_t h8 __attribute__((ext_vector_type(8)));
typedef half_t h4 __attribute__((ext_vector_type(4)));
typedef half_t h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

void seg_set_error(const char* fmt, ...);

#define SEG_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      seg_set_error(__VA_ARGS__);                \
      return SEG_ERR_ARG;                        \
    }                                            \
  } while (0)

#define SEG_CHECK_LAUNCH(name)                                                        \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) {                                                          \
      seg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));           \
      return SEG_ERR_LAUNCH;                                                          \
    }                                                                                 \
  } while (0)

__device__ __forceinline__ float seg_act(float v, int act, float alpha) {
  switch (act) {
    case SEG_ACT_RELU: return v > 0.f ? v : 0.f;
    case SEG_ACT_LEAKY: return v > 0.f ? v : v * alpha;
    case SEG_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}

// derivative of the activation expressed with the activation OUTPUT y
__device__ __forceinline__ float seg_act_grad_from_out(float y, int act, float alpha) {
  switch (act) {
    case SEG_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case SEG_ACT_LEAKY: return y > 0.f ? 1.f : alpha;
    case SEG_ACT_SIGMOID: return y * (1.f - y);
    default: return 1.f;
  }
}

__device__ __forceinline__ u4 ld16(const void* p) { return *reinterpret_cast<const u4*>(p); }
__device__ __forceinline__ void st16(void* p, u4 v) { *reinterpret_cast<u4*>(p) = v; }

__device__ __forceinline__ void unpack8(u4 v, float* f) {
  h8 h = __builtin_bit_cast(h8, v);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)h[i];
}
__device__ __forceinline__ u4 pack8(const float* f) {
  h8 h;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = (half_t)f[i];
  return __builtin_bit_cast(u4, h);
}

// storage-type generic 8-element vector access (8 x fp16 = 16 B, 8 x fp32 = 32 B)
template <typename T> __device__ __forceinline__ void load8(const T* p, float* f);
template <> __device__ __forceinline__ void load8<half_t>(const half_t* p, float* f) { unpack8(ld16(p), f); }
template <> __device__ __forceinline__ void load8<float>(const float* p, float* f) {
  const f4 a = *reinterpret_cast<const f4*>(p), b = *reinterpret_cast<const f4*>(p + 4);
  f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* f);
template <> __device__ __forceinline__ void store8<half_t>(half_t* p, const float* f) { st16(p, pack8(f)); }
template <> __device__ __forceinline__ void store8<float>(float* p, const float* f) {
  *reinterpret_cast<f4*>(p) = f4{f[0], f[1], f[2], f[3]};
  *reinterpret_cast<f4*>(p + 4) = f4{f[4], f[5], f[6], f[7]};
}
// ---- parity-mode storage: split_t.  An fp32 value x is kept as TWO fp16 numbers, x ~= hi + lo / 2048 with hi = fp16(x) and
// lo = fp16((x - hi) * 2048): the residual is exact in fp32 and scaled back into fp16's normal range, so the pair carries ~22
// significant bits of x (|x| < 65504; below 2^-14 the absolute error floor is ~2^-36).  The pair is what the fp16 MFMA contracts
// (a b ~= ah bh + (al bh + ah bl) / 2048, three MFMAs per product), so tensors are STORED in that form and every GEMM kernel
// fetches MFMA-ready operands: 8 consecutive channels (an 8-aligned group) occupy 32 bytes = [8 hi halves | 8 lo halves].
// sizeof(split_t) == 4: element-granular pointer arithmetic (pix * cs + coff with coff % 8 == 0) addresses a group's 32 bytes
// exactly like an fp32 tensor would; only access INSIDE a group differs (load1 / store1 below).
struct split_t { uint32_t raw; };
constexpr float SPLIT_SCALE = 2048.f;
constexpr float SPLIT_INV = 1.f / 2048.f;

template <> __device__ __forceinline__ void load8<split_t>(const split_t* p, float* f) {
  const h8 hi = __builtin_bit_cast(h8, ld16(p)), lo = __builtin_bit_cast(h8, ld16(p + 4));
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)hi[i] + (float)lo[i] * SPLIT_INV;
}
__device__ __forceinline__ void split8(const float* f, u4& hi_o, u4& lo_o) {
  typedef float f8v __attribute__((ext_vector_type(8)));
  f8v x;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = f[i];
  const h8 hi = __builtin_convertvector(x, h8);          // v_cvt_pk_f16_f32 (round to nearest even), two values per instruction
  f8v r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (x[i] - (float)hi[i]) * SPLIT_SCALE;
  const h8 lo = __builtin_convertvector(r, h8);
  hi_o = __builtin_bit_cast(u4, hi);
  lo_o = __builtin_bit_cast(u4, lo);
}
template <> __device__ __forceinline__ void store8<split_t>(split_t* p, const float* f) {
  u4 hi, lo;
  split8(f, hi, lo);
  st16(p, hi);
  st16(p + 4, lo);
}
// one element: g = address of the element's 8-aligned group, k = index inside the group (0..7)
template <typename T> __device__ __forceinline__ float load1(const T* g, int k) { return (float)g[k]; }
template <> __device__ __forceinline__ float load1<split_t>(const split_t* g, int k) {
  const half_t* h = reinterpret_cast<const half_t*>(g);
  return (float)h[k] + (float)h[8 + k] * SPLIT_INV;
}
template <typename T> __device__ __forceinline__ void store1(T* g, int k, float v) { g[k] = (T)v; }
template <> __device__ __forceinline__ void store1<split_t>(split_t* g, int k, float v) {
  half_t* h = reinterpret_cast<half_t*>(g);
  const half_t hi = (half_t)v;
  h[k] = hi;
  h[8 + k] = (half_t)((v - (float)hi) * SPLIT_SCALE);
}
// element e of a row of elements that starts at an 8-aligned group boundary
template <typename T> __device__ __forceinline__ float load_elem(const T* row, int64_t e) { return load1(row + (e & ~(int64_t)7), (int)(e & 7)); }
template <typename T> __device__ __forceinline__ void store_elem(T* row, int64_t e, float v) { store1(row + (e & ~(int64_t)7), (int)(e & 7), v); }

// the value a stored element will have when read back (fp16 rounding / the split pair / identity)
template <typename T> __device__ __forceinline__ float stored(float v) { return (float)(T)v; }
template <> __device__ __forceinline__ float stored<split_t>(float v) {
  const half_t hi = (half_t)v;
  return (float)hi + (float)(half_t)((v - (float)hi) * SPLIT_SCALE) * SPLIT_INV;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
