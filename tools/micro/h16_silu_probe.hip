// h16_silu8 (conv.h: GroupNorm affine + SiLU on packed f16 pairs, inline-asm transcendentals) against float arithmetic, every
// f16 input in [-24, 24] x a few coefficient pairs; prints the worst absolute / relative error and any non-finite output.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I pointreggpt_amd/csrc -o tools/micro/h16_silu_probe tools/micro/h16_silu_probe.hip
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "conv.h"
using namespace prg;

__global__ void probe(const h16_u32x4* in, h16_u32x4* out, const h16x2* ab, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  h16x2 a[4], b[4];
  for (int j = 0; j < 4; ++j) { a[j] = ab[j]; b[j] = ab[4 + j]; }
  out[i] = h16_silu8(in[i], a, b);
}

int main() {
  std::vector<uint16_t> xs;
  for (uint32_t bits = 0; bits < 65536; ++bits) {
    _Float16 h;
    uint16_t u = (uint16_t)bits;
    std::memcpy(&h, &u, 2);
    const float f = (float)h;
    if (std::isfinite(f) && std::fabs(f) <= 24.0f) xs.push_back(u);
  }
  while (xs.size() % 8) xs.push_back(0);
  const int n = (int)xs.size() / 8;
  const float A[8] = {1.0f, 0.37f, 2.5f, -1.2f, 0.05f, 3.0f, 1.0f, 0.9f}, B[8] = {0.0f, 0.5f, -1.0f, 0.25f, 0.0f, -2.0f, 4.0f, -0.3f};
  _Float16 abh[16];
  for (int j = 0; j < 8; ++j) { abh[j] = (_Float16)A[j]; abh[8 + j] = (_Float16)B[j]; }
  void *din, *dout, *dab;
  hipMalloc(&din, xs.size() * 2); hipMalloc(&dout, xs.size() * 2); hipMalloc(&dab, sizeof(abh));
  hipMemcpy(din, xs.data(), xs.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dab, abh, sizeof(abh), hipMemcpyHostToDevice);
  probe<<<(n + 255) / 256, 256>>>((const h16_u32x4*)din, (h16_u32x4*)dout, (const h16x2*)dab, n);
  std::vector<uint16_t> ys(xs.size());
  hipMemcpy(ys.data(), dout, ys.size() * 2, hipMemcpyDeviceToHost);
  double worst_abs = 0, worst_rel = 0;
  int bad = 0;
  for (size_t i = 0; i < xs.size(); ++i) {
    _Float16 hx, hy;
    std::memcpy(&hx, &xs[i], 2); std::memcpy(&hy, &ys[i], 2);
    const int j = (int)(i % 8);
    const double y = (double)(float)hx * (double)(float)abh[j] + (double)(float)abh[8 + j];
    const double ref = y / (1.0 + std::exp(-y));
    const double got = (double)(float)hy;
    if (!std::isfinite(got)) { ++bad; continue; }
    const double e = std::fabs(got - ref);
    if (e > worst_abs) worst_abs = e;
    if (std::fabs(ref) > 1e-2 && e / std::fabs(ref) > worst_rel) worst_rel = e / std::fabs(ref);
  }
  std::printf("h16_silu8: %zu inputs, non-finite outputs %d, worst abs %.3e, worst rel (|ref| > 1e-2) %.3e (f16 ulp = 4.9e-4)\n", xs.size(), bad,
              worst_abs, worst_rel);
  return bad ? 1 : 0;
}
