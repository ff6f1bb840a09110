// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 operand layout (fp8 e4m3 x fp8 e4m3, E8M0 block scales).
// build: hipcc --offload-arch=gfx950 -O2 -o mx_layout_probe mx_layout_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(const i32x8* a, const i32x8* b, f32x16* c, const int* sa, const int* sb) {
  int l = threadIdx.x;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
  c[l] = acc;
}
static float e4m3(unsigned char v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}
static void *dA, *dB, *dC, *dSA, *dSB;
static std::vector<float> run(const std::vector<unsigned char>& A, const std::vector<unsigned char>& B, const std::vector<int>& SA,
                              const std::vector<int>& SB) {
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
  hipMemcpy(dSA, SA.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>((i32x8*)dA, (i32x8*)dB, (f32x16*)dC, (int*)dSA, (int*)dSB);
  std::vector<float> C(64 * 16), D(32 * 32);
  hipMemcpy(C.data(), dC, 64 * 64, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = C[l * 16 + r];
  return D;
}
int main() {
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 64 * 64); hipMalloc(&dSA, 256); hipMalloc(&dSB, 256);
  const unsigned char ONE = 0x38;
  std::vector<int> S127(64, 127);
  // 1. rows / cols: A one-hot at (lane, byte) against all-ones B -> which output row lights up
  printf("A one-hot (lane, byte) -> output row (expect lane & 31):");
  for (int la : {0, 5, 31, 32, 37, 63})
    for (int ia : {0, 17, 31}) {
      std::vector<unsigned char> A(2048, 0), B(2048, ONE);
      A[la * 32 + ia] = ONE;
      auto D = run(A, B, S127, S127);
      int row = -1, n = 0;
      for (int r = 0; r < 32; ++r) if (D[r * 32] != 0) { row = r; ++n; }
      printf(" (%d,%d)->%d%s", la, ia, row, n == 1 ? "" : "!");
    }
  printf("\n");
  // 2. k pairing: A one-hot (la, ia) x B one-hot (lb, ib): nonzero iff same k.  Print the partner of each A element.
  int mism = 0;
  for (int ha = 0; ha < 2; ++ha)
    for (int ia = 0; ia < 32; ++ia) {
      int found_h = -1, found_i = -1, cnt = 0;
      for (int hb = 0; hb < 2; ++hb)
        for (int ib = 0; ib < 32; ++ib) {
          std::vector<unsigned char> A(2048, 0), B(2048, 0);
          A[(ha * 32 + 3) * 32 + ia] = ONE;      // row 3
          B[(hb * 32 + 7) * 32 + ib] = ONE;      // col 7
          auto D = run(A, B, S127, S127);
          if (D[3 * 32 + 7] != 0) { found_h = hb; found_i = ib; ++cnt; }
        }
      if (!(cnt == 1 && found_h == ha && found_i == ia)) { ++mism; printf("  A(h=%d,i=%d) pairs with B(h=%d,i=%d) x%d\n", ha, ia, found_h, found_i, cnt); }
    }
  printf("k pairing A(h,i) <-> B(h,i): %s (%d mismatches)\n", mism ? "NOT identity" : "identity", mism);
  // 3. scale blocks: all ones, one lane's A scale doubled -> which rows change and by how much (64 = unchanged)
  for (int l0 : {0, 5, 32, 37}) {
    std::vector<unsigned char> A(2048, ONE), B(2048, ONE);
    std::vector<int> SA(64, 127);
    SA[l0] = 128;
    auto D = run(A, B, SA, S127);
    printf("A scale of lane %d doubled: ", l0);
    for (int r = 0; r < 32; ++r) if (D[r * 32] != 64.0f) printf(" row %d = %.0f", r, D[r * 32]);
    printf("\n");
  }
  // which elements does lane l0's scale cover: one-hot A at (la, ia), scale of lane l0 doubled -> result 2 instead of 1
  for (int l0 : {5, 37}) {
    printf("elements (lane,byte) of row 5 scaled by lane %d's scale:", l0);
    for (int ha = 0; ha < 2; ++ha)
      for (int ia = 0; ia < 32; ++ia) {
        std::vector<unsigned char> A(2048, 0), B(2048, ONE);
        A[(ha * 32 + 5) * 32 + ia] = ONE;
        std::vector<int> SA(64, 127);
        SA[l0] = 128;
        auto D = run(A, B, SA, S127);
        if (D[5 * 32] == 2.0f) printf(" (%d,%d)", ha * 32 + 5, ia);
      }
    printf("\n");
  }
  // 4. random check under: row = l&31, consistent k, lane l's scale covers lane l's own 32 bytes
  std::vector<unsigned char> A(2048), B(2048);
  std::vector<int> SA(64), SB(64);
  srand(1);
  for (auto& v : A) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }
  for (auto& v : B) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }
  for (int l = 0; l < 64; ++l) { SA[l] = 120 + rand() % 12; SB[l] = 122 + rand() % 10; }
  auto D = run(A, B, SA, SB);
  double worst = 0, scale = 0;
  for (int row = 0; row < 32; ++row)
    for (int col = 0; col < 32; ++col) {
      double ref = 0;
      for (int h = 0; h < 2; ++h) {
        double blk = 0;
        for (int i = 0; i < 32; ++i) blk += (double)e4m3(A[(row + 32 * h) * 32 + i]) * (double)e4m3(B[(col + 32 * h) * 32 + i]);
        ref += blk * std::ldexp(1.0, SA[row + 32 * h] - 127) * std::ldexp(1.0, SB[col + 32 * h] - 127);
      }
      worst = std::fmax(worst, std::fabs(ref - D[row * 32 + col]));
      scale = std::fmax(scale, std::fabs(ref));
    }
  printf("random operands + random scales vs hypothesis: max |err| %.3e on scale %.3e\n", worst, scale);
  return 0;
}
