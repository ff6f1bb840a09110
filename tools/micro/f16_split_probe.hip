// f16_split_probe.hip — does v_mfma_f32_32x32x16_f16 keep f16 SUBNORMAL inputs, and how exact is a 2-way f16 split
// (hi + lo, three MFMAs: hh + hl + lh) of float32 operands against a float64 dot product?  (round 4: the `f16x3` mode)
//   build: hipcc --offload-arch=gfx950 -O3 -o f16_split_probe f16_split_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void denorm_kernel(float a, float b, float* out) {
  f16x8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (_Float16)a; B[i] = (_Float16)b; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)A[0]; }
}

// one wave: C[32][32] = A[32][K] . B[32][K]^T with the split applied on the fly; K a multiple of 16
// seg > 0: the MFMA chain is cut every `seg` terms and the partials are summed in float32 (wide = 0) or float64 (wide = 1)
__global__ void split_kernel(const float* __restrict__ A, const float* __restrict__ B, int K, float* C, int terms, int seg = 0, int wide = 0) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 c = {0};
  double td[16] = {0}; float tf[16] = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    if (seg && k0 && k0 % seg == 0) for (int e = 0; e < 16; ++e) { td[e] += (double)c[e]; tf[e] += c[e]; c[e] = 0; }
    f16x8 ah, al, bh, bl;
    for (int i = 0; i < 8; ++i) {
      const float a = A[r * K + k0 + 8 * h + i], b = B[r * K + k0 + 8 * h + i];
      ah[i] = (_Float16)a; al[i] = (_Float16)(a - (float)ah[i]);
      bh[i] = (_Float16)b; bl[i] = (_Float16)(b - (float)bh[i]);
    }
    if (terms >= 3) { c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0); }
    if (terms >= 4) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
  }
  // C layout of 32x32: lane (r = column j? ) -> element e: row = (e & 3) + 8 * (e >> 2) + 4 * h, col = r
  if (seg) for (int e = 0; e < 16; ++e) { td[e] += (double)c[e]; tf[e] += c[e]; c[e] = wide ? (float)td[e] : tf[e]; }
  for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = c[e];
}

int main() {
  float* d; hipMalloc(&d, 64);
  float hst[2];
  const float subs[] = {9.5367431640625e-7f /* 2^-20 */, 5.9604644775390625e-8f /* 2^-24, the smallest */, 3.0517578125e-5f /* 2^-15 */};
  for (float s : subs) {
    denorm_kernel<<<1, 64>>>(s, 1.0f, d);
    hipMemcpy(hst, d, 8, hipMemcpyDeviceToHost);
    printf("subnormal A = %.6e (as f16 %.6e), B = 1: C = %.9e  expected %.9e  -> %s\n", s, hst[1], hst[0], 16.0 * s,
           hst[0] == 16.0f * s ? "KEPT" : "FLUSHED/ALTERED");
    denorm_kernel<<<1, 64>>>(1.0f, s, d);
    hipMemcpy(hst, d, 8, hipMemcpyDeviceToHost);
    printf("subnormal B = %.6e, A = 1: C = %.9e -> %s\n", s, hst[0], hst[0] == 16.0f * s ? "KEPT" : "FLUSHED/ALTERED");
  }
  for (int K : {576, 1152, 4608, 6912}) {
    std::vector<float> A(32 * K), B(32 * K), C(1024);
    srand(K);
    auto rnd = [] { float u = 0; for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX; return u - 6.0f; };
    for (auto& v : A) { float x = rnd() * 1.5f; v = x / (1.0f + std::exp(-x)); }   // SiLU of a normal: the conv inputs' distribution
    for (auto& v : B) v = rnd();                                                   // standardised weights: unit variance
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    struct V { int terms, seg, wide; };
    for (V v : {V{1, 0, 0}, V{3, 0, 0}, V{4, 0, 0}, V{3, 288, 0}, V{3, 288, 1}, V{3, 32, 0}, V{3, 32, 1}, V{3, 96, 1}}) {
      const int terms = v.terms;
      split_kernel<<<1, 64>>>(dA, dB, K, dC, terms, v.seg, v.wide);
      hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
      double worst = 0, rms = 0, worst32 = 0, scale = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double ex = 0; float f = 0;
          for (int k = 0; k < K; ++k) { ex += (double)A[i * K + k] * B[j * K + k]; f = fmaf(A[i * K + k], B[j * K + k], f); }
          const double e = std::fabs(C[i * 32 + j] - ex);
          worst = std::max(worst, e); rms += e * e; worst32 = std::max(worst32, std::fabs((double)f - ex)); scale += ex * ex;
        }
      printf("seg %3d %s | ", v.seg, v.wide ? "f64" : "f32");
      printf("K = %4d, %d MFMA terms: max |err| = %.3e, rms = %.3e ; serial fp32 fmaf chain max |err| = %.3e ; rms(output) = %.3f\n", K, terms,
             worst, std::sqrt(rms / 1024), worst32, std::sqrt(scale / 1024));
    }
  }
  return 0;
}
