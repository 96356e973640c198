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
// the value a stored element will have when read back (fp16 rounding / identity)
template <typename T> __device__ __forceinline__ float stored(float v) { return (float)(T)v; }

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
