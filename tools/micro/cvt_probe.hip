#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short short2v;
__global__ void k(const float* in, const float* sc, unsigned* out, int n) {
  int i = threadIdx.x;
  __builtin_amdgcn_s_setreg((1 | (23 << 6) | (0 << 11)), 1);   // hwreg(HW_REG_MODE = 1, offset 23, size 1): FP16_OVFL
  if (i < n) {
    bf16x2 v = {(__bf16)in[2 * i], (__bf16)in[2 * i + 1]};
    short2v o = {0, 0};
    o = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(o, v, sc[i], false);
    short2v o2 = {0, 0};
    o2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o2, in[2 * i], in[2 * i + 1], sc[i], false);
    out[2 * i] = (unsigned)(unsigned short)o[0] | ((unsigned)(unsigned short)o[1] << 16);
    out[2 * i + 1] = (unsigned)(unsigned short)o2[0] | ((unsigned)(unsigned short)o2[1] << 16);
  }
}
int main() {
  const int n = 12;
  float hin[2 * n] = {1.0f, -1.0f, 448.0f, 1000.0f, 0.3f, 0.35f, 8.0f, 16.0f, 1.0f, 2.0f, 3.0f, 5.0f, 0.0625f, 0.001f, 17.0f, 18.0f, 19.0f, 20.0f, 1e6f, -1e6f, 0.0f, -0.0f, 1.5f, 2.5f};
  float hsc[n] = {1.0f, 1.0f, 1.0f, 1.0f, 2.0f, 0.5f, 1.0f, 1.0f, 4.0f, 1.0f, 1.0f, 8.0f};
  float *din, *dsc; unsigned* dout;
  hipMalloc(&din, sizeof(hin)); hipMalloc(&dsc, sizeof(hsc)); hipMalloc(&dout, 2 * n * 4);
  hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice); hipMemcpy(dsc, hsc, sizeof(hsc), hipMemcpyHostToDevice);
  k<<<1, 64>>>(din, dsc, dout, n);
  unsigned ho[2 * n]; hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("in (%g, %g) scale %g -> bf16 path %08x  f32 path %08x\n", hin[2 * i], hin[2 * i + 1], hsc[i], ho[2 * i], ho[2 * i + 1]);
  return 0;
}
