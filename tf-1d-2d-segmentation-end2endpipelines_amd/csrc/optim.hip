// Optimiser kernels of the train step (Keras Adam, 2DCNN/Train.py:322 via utils/tf_optimizers.py:11, and the seven other
// registry optimisers) on the flat parameter buffers, plus the bf16 packing of gradient slices for the data-parallel
// exchange.  HBM-bound streaming kernels: 30 B per parameter (26 B when the gradient arrives as bf16).
//
// Gradient operand: fp32 (single GPU; the weight-gradient kernels accumulate there) or bf16 (what travelled over xGMI:
// the sum of the ranks' slices, SURVEY.md 8e mitigation 1) - always widened to fp32 before the update, the optimiser state
// and the master weights stay fp32.
// Non-finite gradients: the update of that element is skipped AND counted into a device counter the caller owns, so
// that an overflow of the fp16 backward pass is visible (SegModel.nonfinite_updates(), harness.fit backs the loss scale off).
#include "seg_common.h"
#include <cstdlib>

namespace {

constexpr int TPB = 256;
#define ST(s) ((hipStream_t)(s))

__device__ __forceinline__ float grad_load(const float* g, int64_t i) { return g[i]; }
__device__ __forceinline__ float grad_load(const uint16_t* g, int64_t i) { return __uint_as_float((uint32_t)g[i] << 16); }     // bf16

// operand copy of an updated parameter for the GEMM kernels: the fp16 image (throughput mode) or the split pair of the parity
// mode (seg_common.h: element i of the flat buffer -> halves [16 (i / 8) + i % 8] = hi and [.. + 8] = lo)
__device__ __forceinline__ void operand_store(half_t* p16, int fmt, int64_t i, float v) {
  if (fmt == SEG_OPERAND_F16) {
    p16[i] = (half_t)v;
  } else {
    const int64_t o = ((i >> 3) << 4) + (i & 7);
    const half_t hi = (half_t)v;
    p16[o] = hi;
    p16[o + 8] = (half_t)((v - (float)hi) * SPLIT_SCALE);
  }
}

__device__ __forceinline__ void count_nonfinite(uint32_t* counter, uint32_t mine) {
  if (counter == nullptr) return;
  // rare path: one atomic per wave that saw any
  const unsigned long long any = __ballot(mine != 0);
  if (any == 0) return;
  uint32_t tot = mine;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(counter, tot);
}

// one element of the Keras Adam update (utils/tf_optimizers.py:11); false = non-finite gradient, nothing touched.  Explicit fused
// multiply-adds: the scalar and the eight-per-thread kernel must not depend on what the compiler contracts in each of them.
__device__ __forceinline__ bool adam_elem(float& p, float& m, float& v, float g, float alpha_t, float b1, float b2, float eps) {
  if (!(g == g && fabsf(g) < 3.0e38f)) return false;
  m = __fmaf_rn(b1, m, __fmul_rn(1.f - b1, g));
  v = __fmaf_rn(b2, v, __fmul_rn(__fmul_rn(1.f - b2, g), g));
  p = __fsub_rn(p, __fdiv_rn(__fmul_rn(alpha_t, m), __fadd_rn(__fsqrt_rn(v), eps)));
  return true;
}

template <typename GT>
__global__ __launch_bounds__(TPB) void adam_kernel(float* p, const GT* g, float* m, float* v, half_t* p16, int p16_fmt, int64_t n, float alpha_t,
                                                   float b1, float b2, float eps, float gmul, const float* alpha_ptr, uint32_t* nonfinite) {
  if (alpha_ptr != nullptr) alpha_t *= *alpha_ptr;      // step size from device memory: a captured hipGraph replays with a fresh value
  uint32_t bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const float gi = grad_load(g, i) * gmul;
    float pi = p[i], mi = m[i], vi = v[i];
    if (adam_elem(pi, mi, vi, gi, alpha_t, b1, b2, eps)) {
      m[i] = mi;
      v[i] = vi;
      p[i] = pi;
    } else {
      ++bad;
    }
    if (p16) operand_store(p16, p16_fmt, i, pi);
  }
  count_nonfinite(nonfinite, bad);
}

// The same update, eight consecutive parameters per thread: 32-byte loads and stores on every stream (a wave moves 2 KB of each
// array per trip), and the eight operand copies of a thread are exactly one group of the GEMM operand - one 16-byte store of fp16
// halves, or the parity mode's 32-byte [8 hi | 8 lo] pair - instead of 2-byte scatters.  The scalar kernel above keeps the
// unaligned cases and the tail.  Measured on the 199 M parameters of C3, alone: 1.76 ms = 3.85 TB/s (scalar) -> 1.23 ms = 5.5 TB/s.
__device__ __forceinline__ void grad_load8(const float* g, int64_t i8, float* o) {
  const f4 a = *reinterpret_cast<const f4*>(g + i8 * 8), b = *reinterpret_cast<const f4*>(g + i8 * 8 + 4);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
__device__ __forceinline__ void grad_load8(const uint16_t* g, int64_t i8, float* o) {
  const u4 r = *reinterpret_cast<const u4*>(g + i8 * 8);
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[2 * k] = __uint_as_float(w[k] << 16);
    o[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
  }
}
__device__ __forceinline__ void load8f(const float* x, int64_t i8, float* o) { grad_load8(x, i8, o); }
__device__ __forceinline__ void store8f(float* x, int64_t i8, const float* v) {
  *reinterpret_cast<f4*>(x + i8 * 8) = f4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f4*>(x + i8 * 8 + 4) = f4{v[4], v[5], v[6], v[7]};
}

template <typename GT>
__global__ __launch_bounds__(TPB) void adam8_kernel(float* p, const GT* g, float* m, float* v, half_t* p16, int p16_fmt, int64_t n8, float alpha_t,
                                                    float b1, float b2, float eps, float gmul, const float* alpha_ptr, uint32_t* nonfinite) {
  if (alpha_ptr != nullptr) alpha_t *= *alpha_ptr;
  uint32_t bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n8; i += (int64_t)gridDim.x * TPB) {
    float gi[8], pi[8], mi[8], vi[8];
    grad_load8(g, i, gi);
    load8f(p, i, pi);
    load8f(m, i, mi);
    load8f(v, i, vi);
    bool any_ok = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (adam_elem(pi[k], mi[k], vi[k], gi[k] * gmul, alpha_t, b1, b2, eps)) any_ok = true;
      else ++bad;
    }
    if (any_ok) {
      store8f(m, i, mi);
      store8f(v, i, vi);
      store8f(p, i, pi);
    }
    if (p16) {
      if (p16_fmt == SEG_OPERAND_F16) {
        h8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (half_t)pi[k];
        *reinterpret_cast<h8*>(p16 + i * 8) = o;
      } else {
        h8 hi, lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          hi[k] = (half_t)pi[k];
          lo[k] = (half_t)((pi[k] - (float)hi[k]) * SPLIT_SCALE);
        }
        *reinterpret_cast<h8*>(p16 + i * 16) = hi;
        *reinterpret_cast<h8*>(p16 + i * 16 + 8) = lo;
      }
    }
  }
  count_nonfinite(nonfinite, bad);
}

// The other optimisers of utils/tf_optimizers.py:12-26 with the Keras (>= 2.11) update rules; s1 / s2 are the two state
// buffers (the flat m / v buffers of the parameter store), c[] the per-step scalars the host computes (seg_hip.h).
template <typename GT>
__global__ __launch_bounds__(TPB) void optimizer_kernel(int kind, float* p, const GT* g, float* s1, float* s2, half_t* p16, int p16_fmt, int64_t n,
                                                        float c0, float c1, float c2, float c3, float c4, float gmul, uint32_t* nonfinite) {
  uint32_t bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const float gi = grad_load(g, i) * gmul;
    float pi = p[i];
    if (gi == gi && fabsf(gi) < 3.0e38f) {
      switch (kind) {
        case SEG_OPT_SGD: pi -= c0 * gi; break;                                       // c0 = lr
        case SEG_OPT_RMSPROP: {                                                        // c0 = lr, c1 = rho, c2 = eps
          const float vv = c1 * s2[i] + (1.f - c1) * gi * gi;
          s2[i] = vv;
          pi -= c0 * gi / sqrtf(vv + c2);
          break;
        }
        case SEG_OPT_ADAGRAD: {                                                        // c0 = lr, c2 = eps; s2 starts at 0.1
          const float a = s2[i] + gi * gi;
          s2[i] = a;
          pi -= c0 * gi / sqrtf(a + c2);
          break;
        }
        case SEG_OPT_ADADELTA: {                                                       // c0 = lr, c1 = rho, c2 = eps
          const float ag = c1 * s1[i] + (1.f - c1) * gi * gi;
          const float d = -sqrtf(s2[i] + c2) * gi / sqrtf(ag + c2);
          s1[i] = ag;
          s2[i] = c1 * s2[i] + (1.f - c1) * d * d;
          pi += c0 * d;
          break;
        }
        case SEG_OPT_ADAMAX: {                                                         // c0 = lr / (1 - b1^t), c1 = b1, c2 = b2, c3 = eps
          const float mm = s1[i] + (gi - s1[i]) * (1.f - c1);
          const float u = fmaxf(c2 * s2[i], fabsf(gi));
          s1[i] = mm;
          s2[i] = u;
          pi -= c0 * mm / (u + c3);
          break;
        }
        case SEG_OPT_NADAM: {   // c0 = lr, c1 = u_{t+1} / (1 - prod u_{1..t+1}), c2 = (1 - u_t) / (1 - prod u_{1..t}), c3 = 1 / (1 - b2^t), c4 = eps
          const float mm = s1[i] + (gi - s1[i]) * (1.f - 0.9f);
          const float vv = s2[i] + (gi * gi - s2[i]) * (1.f - 0.999f);
          s1[i] = mm;
          s2[i] = vv;
          pi -= c0 * (c1 * mm + c2 * gi) / (sqrtf(vv * c3) + c4);
          break;
        }
        default: {              // SEG_OPT_FTRL (lr_power -0.5, no l1 / l2 / shrinkage): c0 = lr; s1 = accumulator (starts at 0.1), s2 = linear
          const float a0 = s1[i], a1 = a0 + gi * gi;
          const float lin = s2[i] + gi - (sqrtf(a1) - sqrtf(a0)) / c0 * pi;
          s1[i] = a1;
          s2[i] = lin;
          pi = -lin / (sqrtf(a1) / c0);
          break;
        }
      }
      p[i] = pi;
    } else {
      ++bad;
    }
    if (p16) operand_store(p16, p16_fmt, i, pi);
  }
  count_nonfinite(nonfinite, bad);
}

// fp32 -> bf16, round to nearest even; NaN stays NaN, +-Inf stays +-Inf (so that the receiving optimiser still sees and
// counts a non-finite gradient).  8 elements per thread: 32-B loads, 16-B stores.
__global__ __launch_bounds__(TPB) void cast_bf16_kernel(const float* x, uint16_t* y, int64_t n) {
  auto cvt = [](float f) -> uint32_t {
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;         // NaN: keep it quiet and non-zero in the mantissa
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;                         // Inf + rounding stays Inf (mantissa is zero)
  };
  const int64_t n8 = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n8; i += (int64_t)gridDim.x * TPB) {
    const f4 a = *reinterpret_cast<const f4*>(x + 8 * i), b = *reinterpret_cast<const f4*>(x + 8 * i + 4);
    u4 o;
    o[0] = cvt(a[0]) | (cvt(a[1]) << 16);
    o[1] = cvt(a[2]) | (cvt(a[3]) << 16);
    o[2] = cvt(b[0]) | (cvt(b[1]) << 16);
    o[3] = cvt(b[2]) | (cvt(b[3]) << 16);
    *reinterpret_cast<u4*>(y + 8 * i) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const int64_t i = (n8 << 3) + threadIdx.x;
    y[i] = (uint16_t)cvt(x[i]);
  }
}

// The optimiser streams 30 B per parameter next to the forward GEMMs of the following step (SegModel runs it on the side
// stream): 1024 blocks keep the HBM busy and leave CUs to the GEMMs - measured on C3 against 256 / 512 / 2048 / 4096 blocks
// (25.2 / 24.1 / 23.7 / 23.8 ms per step; 1024: 23.5).
int optimizer_grid(int64_t n, int cap = 1024) {
  int64_t blocks = (n + TPB - 1) / TPB;
  if (blocks < 1) blocks = 1;
  if (cap > 0 && blocks > cap) blocks = cap;
  return (int)blocks;
}

}  // namespace

extern "C" int seg_adam_step_ex(float* p, const void* g, int32_t g_dtype, float* m, float* v, void* p16, int32_t p16_fmt, int64_t n, float alpha_t,
                                const float* alpha_ptr, float b1, float b2, float eps, float gmul, uint32_t* nonfinite, seg_stream_t stream) {
  SEG_CHECK_ARG(p && g && m && v && n > 0, "seg_adam_step_ex: bad arguments");
  SEG_CHECK_ARG(g_dtype == SEG_GRAD_F32 || g_dtype == SEG_GRAD_BF16, "seg_adam_step_ex: g_dtype=%d", g_dtype);
  SEG_CHECK_ARG(p16_fmt == SEG_OPERAND_F16 || p16_fmt == SEG_OPERAND_SPLIT, "seg_adam_step_ex: p16_fmt=%d", p16_fmt);
  // eight parameters per thread where every stream is 32-byte aligned (slices of the flat parameter store are: slots are padded to
  // whole groups of eight); the scalar kernel takes the tail and everything else
  // One block per CU: 1.23 ms for C3's 199 M parameters alone (5.5 TB/s; 1 024 / 2 048 blocks: 1.31 / 1.32 ms) and neutral beside the
  // forward pass it shares the GPU with (interleaved A/B against the scalar kernel: 49.62 / 49.61 / 49.62 against 49.55 / 49.64 / 49.64
  // ms per C3 step; 512 and 1 024 blocks cost +0.5 / +0.25 ms there).
  const bool vec = n >= 8 &&
                   ((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)p16) & 31) == 0) && (((uintptr_t)g & (g_dtype == SEG_GRAD_F32 ? 31 : 15)) == 0);
  const int64_t n8 = vec ? n >> 3 : 0, done = n8 << 3, rest = n - done;
  const size_t gsz = g_dtype == SEG_GRAD_F32 ? 4 : 2;
  if (n8 > 0) {
    if (g_dtype == SEG_GRAD_F32)
      hipLaunchKernelGGL((adam8_kernel<float>), dim3(optimizer_grid(n8, 256)), dim3(TPB), 0, ST(stream), p, (const float*)g, m, v, (half_t*)p16, p16_fmt, n8,
                         alpha_t, b1, b2, eps, gmul, alpha_ptr, nonfinite);
    else
      hipLaunchKernelGGL((adam8_kernel<uint16_t>), dim3(optimizer_grid(n8, 256)), dim3(TPB), 0, ST(stream), p, (const uint16_t*)g, m, v, (half_t*)p16,
                         p16_fmt, n8, alpha_t, b1, b2, eps, gmul, alpha_ptr, nonfinite);
    SEG_CHECK_LAUNCH("seg_adam_step_ex (8 per thread)");
  }
  if (rest > 0) {
    // (done is a multiple of eight: the operand copy of element done + i is element i of the tail's own groups in either format)
    half_t* p16t = p16 == nullptr ? nullptr : (half_t*)p16 + (p16_fmt == SEG_OPERAND_F16 ? done : 2 * done);
    const void* gt = (const char*)g + done * gsz;
    if (g_dtype == SEG_GRAD_F32)
      hipLaunchKernelGGL((adam_kernel<float>), dim3(optimizer_grid(rest)), dim3(TPB), 0, ST(stream), p + done, (const float*)gt, m + done, v + done, p16t,
                         p16_fmt, rest, alpha_t, b1, b2, eps, gmul, alpha_ptr, nonfinite);
    else
      hipLaunchKernelGGL((adam_kernel<uint16_t>), dim3(optimizer_grid(rest)), dim3(TPB), 0, ST(stream), p + done, (const uint16_t*)gt, m + done, v + done,
                         p16t, p16_fmt, rest, alpha_t, b1, b2, eps, gmul, alpha_ptr, nonfinite);
    SEG_CHECK_LAUNCH("seg_adam_step_ex");
  }
  return SEG_OK;
}

extern "C" int seg_adam_step(float* p, const float* g, float* m, float* v, void* p16, int64_t n, float alpha_t, float b1, float b2, float eps,
                             float gmul, seg_stream_t stream) {
  return seg_adam_step_ex(p, g, SEG_GRAD_F32, m, v, p16, SEG_OPERAND_F16, n, alpha_t, nullptr, b1, b2, eps, gmul, nullptr, stream);
}
extern "C" int seg_adam_step_dev(float* p, const float* g, float* m, float* v, void* p16, int64_t n, const float* alpha_ptr, float b1, float b2,
                                 float eps, float gmul, seg_stream_t stream) {
  SEG_CHECK_ARG(alpha_ptr != nullptr, "seg_adam_step_dev: null alpha_ptr");
  return seg_adam_step_ex(p, g, SEG_GRAD_F32, m, v, p16, SEG_OPERAND_F16, n, 1.0f, alpha_ptr, b1, b2, eps, gmul, nullptr, stream);
}

extern "C" int seg_optimizer_step_ex(int32_t kind, float* p, const void* g, int32_t g_dtype, float* s1, float* s2, void* p16, int32_t p16_fmt, int64_t n,
                                     const float* c, float gmul, uint32_t* nonfinite, seg_stream_t stream) {
  SEG_CHECK_ARG(p && g && s1 && s2 && c && n > 0 && kind >= SEG_OPT_SGD && kind <= SEG_OPT_FTRL, "seg_optimizer_step_ex: bad arguments");
  SEG_CHECK_ARG(g_dtype == SEG_GRAD_F32 || g_dtype == SEG_GRAD_BF16, "seg_optimizer_step_ex: g_dtype=%d", g_dtype);
  SEG_CHECK_ARG(p16_fmt == SEG_OPERAND_F16 || p16_fmt == SEG_OPERAND_SPLIT, "seg_optimizer_step_ex: p16_fmt=%d", p16_fmt);
  if (g_dtype == SEG_GRAD_F32)
    hipLaunchKernelGGL((optimizer_kernel<float>), dim3(optimizer_grid(n)), dim3(TPB), 0, ST(stream), kind, p, (const float*)g, s1, s2, (half_t*)p16,
                       p16_fmt, n, c[0], c[1], c[2], c[3], c[4], gmul, nonfinite);
  else
    hipLaunchKernelGGL((optimizer_kernel<uint16_t>), dim3(optimizer_grid(n)), dim3(TPB), 0, ST(stream), kind, p, (const uint16_t*)g, s1, s2,
                       (half_t*)p16, p16_fmt, n, c[0], c[1], c[2], c[3], c[4], gmul, nonfinite);
  SEG_CHECK_LAUNCH("seg_optimizer_step_ex");
  return SEG_OK;
}
extern "C" int seg_optimizer_step(int32_t kind, float* p, const float* g, float* s1, float* s2, void* p16, int64_t n, const float* c,
                                  float gmul, seg_stream_t stream) {
  return seg_optimizer_step_ex(kind, p, g, SEG_GRAD_F32, s1, s2, p16, SEG_OPERAND_F16, n, c, gmul, nullptr, stream);
}

extern "C" int seg_cast_f32_bf16(const float* x, void* y, int64_t n, seg_stream_t stream) {
  SEG_CHECK_ARG(x && y && n > 0 && ((uintptr_t)x & 31) == 0 && ((uintptr_t)y & 15) == 0, "seg_cast_f32_bf16: bad arguments (x 32-B, y 16-B aligned)");
  int64_t blocks = ((n >> 3) + TPB - 1) / TPB;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((int)blocks), dim3(TPB), 0, ST(stream), x, (uint16_t*)y, n);
  SEG_CHECK_LAUNCH("seg_cast_f32_bf16");
  return SEG_OK;
}
