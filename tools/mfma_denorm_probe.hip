// Does v_mfma_f32_32x32x16_f16 on gfx950 honour fp16 SUBNORMAL inputs?  (Build-authored probe; decides whether the split pair's lo
// half can be stored unscaled - hi + lo with lo = fp16(x - hi), possibly subnormal - and contracted into ONE accumulator set.)
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_denorm_probe.hip -o ab/mfma_denorm_probe && ab/mfma_denorm_probe
// Prints, for a few magnitudes of a and b: the MFMA's result of sum_k a_k b_k over k = 16 against the exact fp64 value.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half_t;
typedef half_t h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// A: 32 x 16 (row m, k), B: 16 x 32 (k, column n); lane (fr = lane & 31, fh = lane >> 5) holds A[fr][8 fh .. 8 fh + 7] and B[8 fh ..][fr]
__global__ void probe(const half_t* A, const half_t* B, float* D) {
  const int lane = threadIdx.x, fr = lane & 31, fh = lane >> 5;
  h8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[fr * 16 + 8 * fh + j];
    b[j] = B[(8 * fh + j) * 32 + fr];
  }
  f16v acc;
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  // acc[q]: row (q & 3) + 8 (q >> 2) + 4 fh, column fr
  for (int q = 0; q < 16; ++q) D[((q & 3) + 8 * (q >> 2) + 4 * fh) * 32 + fr] = acc[q];
}

static void run(const char* what, double amag, double bmag) {
  std::vector<half_t> A(32 * 16), B(16 * 32);
  srand(7);
  for (auto& v : A) v = (half_t)(float)(amag * (0.25 + 0.75 * rand() / (double)RAND_MAX) * ((rand() & 1) ? 1 : -1));
  for (auto& v : B) v = (half_t)(float)(bmag * (0.25 + 0.75 * rand() / (double)RAND_MAX) * ((rand() & 1) ? 1 : -1));
  half_t *dA, *dB;
  float* dD;
  hipMalloc(&dA, A.size() * 2);
  hipMalloc(&dB, B.size() * 2);
  hipMalloc(&dD, 32 * 32 * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  std::vector<float> D(32 * 32);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0, ref_rms = 0;
  int zeros = 0;
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      double ref = 0, mag = 0;
      for (int k = 0; k < 16; ++k) {
        ref += (double)(float)A[m * 16 + k] * (double)(float)B[k * 32 + n];
        mag += fabs((double)(float)A[m * 16 + k] * (double)(float)B[k * 32 + n]);
      }
      worst = fmax(worst, fabs(D[m * 32 + n] - ref) / mag);
      ref_rms += ref * ref;
      zeros += D[m * 32 + n] == 0.f;
    }
  printf("%-44s |a| ~ %.3g  |b| ~ %.3g   max |mfma - exact| / sum |a b| = %.3e   exact zeros %d / 1024   rms(ref) %.3e\n", what, amag, bmag,
         worst, zeros, sqrt(ref_rms / 1024));
  hipFree(dA); hipFree(dB); hipFree(dD);
}

int main() {
  run("normal x normal", 1.0, 1.0);
  run("subnormal a (2^-17) x normal b", ldexp(1.0, -17), 1.0);
  run("subnormal a (2^-22) x normal b", ldexp(1.0, -22), 1.0);
  run("normal a x subnormal b (2^-20)", 1.0, ldexp(1.0, -20));
  run("subnormal a (2^-16) x subnormal b (2^-16)", ldexp(1.0, -16), ldexp(1.0, -16));
  run("normal a (2^-13) x normal b (2^-13): tiny products", ldexp(1.0, -13), ldexp(1.0, -13));
  return 0;
}
