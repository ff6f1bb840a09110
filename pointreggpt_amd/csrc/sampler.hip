// sampler.hip — one fused elementwise kernel per denoising transition (HBM-bound: 20-24 B/pixel/step):
// x0 from the network output, DDNM known-pixel replacement, clamp, posterior / DDIM combination, noise.
// Built with -ffp-contract=off: the float ops are in the reference's order (sd:1158-1162, 1173-1180, 1210-1218,
// 1250-1251, 1280, 1369-1373) so that, given identical network outputs and noise, the state is bit-identical.
#include "sampler.h"

#include <vector>

namespace prg {

// ---- Philox4x32-10 (counter-based; key = per-scene seed, counter = (pixel quad, draw index)) ----
__device__ inline void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ inline void philox4x32_10(uint32_t (&c)[4], uint64_t key) {
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
// four standard normals for (scene key, draw index, pixel quad): two Box-Muller pairs
__device__ inline float4 philox_normal4(uint64_t key, uint32_t draw, uint32_t quad) {
  uint32_t c[4] = {quad, draw, 0x70726721u, 0u};
  philox4x32_10(c, key);
  const float k = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)c[0] + 0.5f) * k, u1 = ((float)c[1] + 0.5f) * k;
  const float u2 = ((float)c[2] + 0.5f) * k, u3 = ((float)c[3] + 0.5f) * k;
  const float r0 = sqrtf(-2.0f * logf(fminf(fmaxf(u0, 1e-12f), 1.0f)));
  const float r1 = sqrtf(-2.0f * logf(fminf(fmaxf(u2, 1e-12f), 1.0f)));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u1, &s0, &c0);
  sincosf(6.283185307179586f * u3, &s1, &c1);
  return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}

__device__ inline float clamp1(float v) { return fminf(fmaxf(v, -1.0f), 1.0f); }

// grid (chunks, B); each thread handles 4 consecutive pixels
__global__ __launch_bounds__(256) void sampler_step_kernel(SamplerStepArgs a) {
  const int k = *a.step_idx;
  const prg_step st = a.steps[k];
  const int b = blockIdx.y;
  const bool last = k == a.n_steps - 1;
  const size_t img = (size_t)b * a.HW;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q * 4 < a.HW; q += gridDim.x * blockDim.x) {
    const size_t o = img + (size_t)q * 4;
    const float4 xv = *reinterpret_cast<const float4*>(a.x + o);
    const float4 uv = *reinterpret_cast<const float4*>(a.u + o);
    float4 cd = make_float4(0, 0, 0, 0), cm = make_float4(-1, -1, -1, -1), nz = make_float4(0, 0, 0, 0);
    if (a.cond) {
      cd = *reinterpret_cast<const float4*>(a.cond + (size_t)b * 2 * a.HW + (size_t)q * 4);
      cm = *reinterpret_cast<const float4*>(a.cond + (size_t)b * 2 * a.HW + a.HW + (size_t)q * 4);
    }
    if (st.sigma != 0.0f) {
      if (a.noise)
        nz = *reinterpret_cast<const float4*>(a.noise + ((size_t)(k + 1) * a.B + b) * a.HW + (size_t)q * 4);
      else
        nz = philox_normal4(a.seeds[b], (uint32_t)(k + 1), (uint32_t)q);
    }
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, us[4] = {uv.x, uv.y, uv.z, uv.w};
    const float cds[4] = {cd.x, cd.y, cd.z, cd.w}, cms[4] = {cm.x, cm.y, cm.z, cm.w};
    const float nzs[4] = {nz.x, nz.y, nz.z, nz.w};
    float r[4], f[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x0p = (st.clip_pred & 1) ? clamp1(us[e]) : us[e];
      const bool known = a.cond && ((cms[e] + 1.0f) * 0.5f > 0.5f);  // get_mask_from_img_cond (sd:507-508)
      float x0 = known ? cds[e] : x0p;
      if (st.clip_pred & 2) x0 = clamp1(x0);   // p_mean_variance clip_denoised (sd:1250); ddim_sample has no such clamp
      float v = st.c_x0 * x0;
      if (st.clip_pred & 4) {
        // refine row (has_refine_step, sd:1307-1314 / 1374-1388): evaluation without the replacement; only the KNOWN
        // pixels take the clamped network output (posterior mean at t = 0 is exactly x0: coef1 = 1, coef2 = 0)
        r[e] = known ? clamp1(us[e]) : xs[e];
        f[e] = (r[e] + 1.0f) * 0.5f;
        continue;
      }
      if (st.c_x != 0.0f) v = v + st.c_x * xs[e];
      if (st.c_eps != 0.0f) {
        const float eps = (st.sqrt_recip * xs[e] - x0p) / st.sqrt_recipm1;
        v = v + st.c_eps * eps;
      }
      if (st.sigma != 0.0f) v = v + st.sigma * nzs[e];
      r[e] = v;
      f[e] = (v + 1.0f) * 0.5f;
    }
    *reinterpret_cast<float4*>(a.x + o) = make_float4(r[0], r[1], r[2], r[3]);
    if (last) *reinterpret_cast<float4*>(a.final_out + o) = make_float4(f[0], f[1], f[2], f[3]);
  }
  // Every workgroup read the step counter at its first instruction; the last one to arrive here advances it for the
  // next launch and re-arms the ticket (kernel boundaries order this against the neighbouring launches).
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nwg = gridDim.x * gridDim.y;
    if (atomicAdd(a.ticket, 1) == nwg - 1) {
      *a.ticket = 0;
      *a.step_idx = k + 1;
    }
  }
}

__global__ void sampler_init_kernel(float* __restrict__ x, const float* __restrict__ noise,
                                    const uint64_t* __restrict__ seeds, int HW) {
  const int b = blockIdx.y;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q * 4 < HW; q += gridDim.x * blockDim.x) {
    const size_t o = (size_t)b * HW + (size_t)q * 4;
    float4 v;
    if (noise)
      v = *reinterpret_cast<const float4*>(noise + o);
    else
      v = philox_normal4(seeds[b], 0u, (uint32_t)q);
    *reinterpret_cast<float4*>(x + o) = v;
  }
}

static inline dim3 step_grid(int HW, int B) {
  // about one workgroup per CU: the last-arriver ticket of sampler_step_kernel is one atomic per workgroup on a single
  // address (~12 ns each, serialised): 1024 workgroups cost more than the launch the ticket saves
  int gx = ceil_div(HW / 4, 256);
  const int cap = 256 / B > 1 ? 256 / B : 1;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  return dim3(gx, B, 1);
}

int launch_sampler_step(const SamplerStepArgs& a, hipStream_t s) {
  PRG_CHECK(a.HW % 4 == 0, "sampler: H*W must be a multiple of 4");
  sampler_step_kernel<<<step_grid(a.HW, a.B), 256, 0, s>>>(a);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}
int launch_sampler_init(float* x, const float* noise, const uint64_t* seeds, int B, int HW, hipStream_t s) {
  PRG_CHECK(HW % 4 == 0, "sampler: H*W must be a multiple of 4");
  sampler_init_kernel<<<step_grid(HW, B), 256, 0, s>>>(x, noise, seeds, HW);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

}  // namespace prg

using namespace prg;

extern "C" int prg_debug_sampler_step(float* x, const float* u, const float* img_cond, const uint64_t* seeds, const prg_step* step, int B,
                                      int HW, int reps, float* avg_us, void* stream) {
  PRG_CHECK(x && u && seeds && step && B > 0 && HW > 0 && HW % 4 == 0 && reps > 0, "prg_debug_sampler_step: bad argument");
  hipStream_t s = (hipStream_t)stream;
  // a table of `reps` copies of the transition (+1 row so that no launch is the chain's last), the device step counter and the ticket
  std::vector<prg_step> rows((size_t)reps + 1, *step);
  char* scratch = nullptr;
  const size_t tab = sizeof(prg_step) * rows.size();
  PRG_HIP(hipMalloc(&scratch, tab + 2 * sizeof(int)));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = PRG_OK;
  auto fail = [&](hipError_t e) { if (e != hipSuccess && rc == PRG_OK) { set_error(std::string("prg_debug_sampler_step: ") + hipGetErrorString(e)); rc = PRG_E_HIP; } };
  fail(hipMemcpyAsync(scratch, rows.data(), tab, hipMemcpyHostToDevice, s));
  fail(hipMemsetAsync(scratch + tab, 0, 2 * sizeof(int), s));
  fail(hipEventCreate(&e0));
  fail(hipEventCreate(&e1));
  if (rc == PRG_OK) {
    SamplerStepArgs a{};
    a.x = x; a.u = u; a.cond = img_cond; a.noise = nullptr; a.steps = reinterpret_cast<const prg_step*>(scratch);
    a.step_idx = reinterpret_cast<int*>(scratch + tab); a.ticket = a.step_idx + 1; a.seeds = seeds; a.final_out = x;
    a.B = B; a.HW = HW; a.n_steps = reps + 1;
    fail(hipEventRecord(e0, s));
    for (int i = 0; i < reps && rc == PRG_OK; ++i) rc = launch_sampler_step(a, s);
    fail(hipEventRecord(e1, s));
    fail(hipStreamSynchronize(s));
    float ms = 0.0f;
    if (rc == PRG_OK) fail(hipEventElapsedTime(&ms, e0, e1));
    if (avg_us) *avg_us = ms * 1e3f / (float)reps;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(scratch);
  return rc;
}
