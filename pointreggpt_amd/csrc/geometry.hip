// geometry.hip — memory-bound camera-geometry kernels (HBM roofline): pinhole unprojection, SE(3) move,
// z-buffer projection, DepthAugment, mask application.  Built with -ffp-contract=off: every float op below is
// written in the exact order torch / numpy evaluate the reference expressions, fused-multiply-adds only where
// the host BLAS uses them (the 3x3 rotation: fma chain in k order — probed), so the integer
// pixel indices and the float32 depths come out bit-identical to the reference.
//
// sd = denoising_diffusion_pytorch/successive_ddnm_diffusion.py, dc = depth_correction_pytorch/depth_correction.py
#include "common.h"

namespace prg {

static constexpr uint32_t kEmpty = 0xFFFFFFFFu;  // z-buffer sentinel: above every positive float's bit pattern

struct Cam {
  float fx, fy, cx, cy;
};
__device__ inline Cam load_cam(const float* K, int b) {
  const float* k = K + (size_t)b * 9;
  return Cam{k[0], k[4], k[2], k[5]};
}

// rotate + translate exactly like `pc @ R^T + t` on the host: fma chain over k = 0,1,2, then a separate add.
__device__ inline void se3_apply(const float* P, float x, float y, float z, float& ox, float& oy, float& oz) {
  // P row-major 4x4: R[j][k] = P[j*4+k], t[j] = P[j*4+3]
  float v;
  v = x * P[0]; v = __fmaf_rn(y, P[1], v); v = __fmaf_rn(z, P[2], v); ox = v + P[3];
  v = x * P[4]; v = __fmaf_rn(y, P[5], v); v = __fmaf_rn(z, P[6], v); oy = v + P[7];
  v = x * P[8]; v = __fmaf_rn(y, P[9], v); v = __fmaf_rn(z, P[10], v); oz = v + P[11];
}

// project one camera-frame point and min-merge it into the z-buffer (sd:225-258)
__device__ inline void splat(uint32_t* zbuf, const Cam& c, float x, float y, float z, int H, int W) {
  if (!(z > 0.0f)) return;  // also rejects NaN
  float fc = rintf(x * c.fx / z + c.cx);  // torch.round = round-half-even
  float fr = rintf(y * c.fy / z + c.cy);
  if (!(fc >= 0.0f && fc < (float)W && fr >= 0.0f && fr < (float)H)) return;
  int col = (int)fc, row = (int)fr;
  atomicMin(zbuf + (size_t)row * W + col, __float_as_uint(z));
}

// ---- depth2pc_tensor (sd:176-209) ------------------------------------------------------------
__global__ void depth2pc_kernel(const float* __restrict__ depth, const float* __restrict__ K, float* __restrict__ pc,
                                uint8_t* __restrict__ valid, int HW, int W, float lo, float hi, float inval,
                                int clip) {
  int b = blockIdx.y;
  Cam c = load_cam(K, b);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    float d = depth[(size_t)b * HW + i];
    bool ok = clip ? (d > lo && d < hi) : true;
    int r = i / W, col = i - r * W;
    float x = inval, y = inval, z = inval;
    if (ok) {
      z = d;
      x = ((float)col - c.cx) * z / c.fx;
      y = ((float)r - c.cy) * z / c.fy;
    }
    float* o = pc + ((size_t)b * HW + i) * 3;
    o[0] = x; o[1] = y; o[2] = z;
    valid[(size_t)b * HW + i] = ok ? 1 : 0;
  }
}

// ---- pc2depth_tensor scatter (sd:212-258) -----------------------------------------------------
__global__ void pc2depth_scatter_kernel(const float* __restrict__ pc, const uint8_t* __restrict__ valid,
                                        const float* __restrict__ K, uint32_t* __restrict__ zbuf, int N, int H,
                                        int W) {
  int b = blockIdx.y;
  Cam c = load_cam(K, b);
  uint32_t* zb = zbuf + (size_t)b * H * W;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    if (valid && !valid[(size_t)b * N + i]) continue;
    const float* p = pc + ((size_t)b * N + i) * 3;
    splat(zb, c, p[0], p[1], p[2], H, W);
  }
}

// ---- ragged clouds with per-scene pose (sd:2531-2547) -----------------------------------------
__global__ void project_points_kernel(const float* __restrict__ pts, const int64_t* __restrict__ offsets,
                                      const float* __restrict__ pose, const float* __restrict__ K,
                                      uint32_t* __restrict__ zbuf, int H, int W) {
  int b = blockIdx.y;
  int64_t beg = offsets[b], end = offsets[b + 1];
  Cam c = load_cam(K, b);
  uint32_t* zb = zbuf + (size_t)b * H * W;
  const float* P = pose ? pose + (size_t)b * 16 : nullptr;
  for (int64_t i = beg + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (int64_t)gridDim.x * blockDim.x) {
    float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    if (P) {
      float ox, oy, oz;
      se3_apply(P, x, y, z, ox, oy, oz);
      x = ox; y = oy; z = oz;
    }
    splat(zb, c, x, y, z, H, W);
  }
}

// ---- fused reproject_tensor (sd:268-286): one pass over the source depth, no cloud in HBM ------
__global__ void reproject_scatter_kernel(const float* __restrict__ depth, const float* __restrict__ K,
                                         const float* __restrict__ pose, uint32_t* __restrict__ zbuf, int H,
                                         int W, float unit, float lo, float hi) {
  int b = blockIdx.y;
  int HW = H * W;
  Cam c = load_cam(K, b);
  const float* P = pose + (size_t)b * 16;
  uint32_t* zb = zbuf + (size_t)b * HW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    float d = depth[(size_t)b * HW + i] * unit;
    if (!(d > lo && d < hi)) continue;
    int r = i / W, col = i - r * W;
    float x = ((float)col - c.cx) * d / c.fx;
    float y = ((float)r - c.cy) * d / c.fy;
    float ox, oy, oz;
    se3_apply(P, x, y, d, ox, oy, oz);
    splat(zb, c, ox, oy, oz, H, W);
  }
}

// ---- z-buffer resolve: sentinel -> 0 / mask, optional scale (sd:259-263, sd:2552) --------------
__global__ void zbuf_resolve_kernel(const uint32_t* __restrict__ zbuf, float* __restrict__ depth,
                                    uint8_t* __restrict__ mask, size_t n, float scale, int do_scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t u = zbuf[i];
    bool hit = u != kEmpty;
    float z = hit ? __uint_as_float(u) : 0.0f;
    if (do_scale) z = z * scale;
    depth[i] = z;  // NOTE: depth may alias zbuf (same element, read before write)
    if (mask) mask[i] = hit ? 1 : 0;
  }
}

// ---- numpy point_cloud + inverse pose, float64 (sd:122-143, sd:2627-2628) ----------------------
__global__ void unproject_f64_kernel(const float* __restrict__ depth, const float* __restrict__ K,
                                     const float* __restrict__ pose, double* __restrict__ xyz,
                                     uint8_t* __restrict__ valid, int H, int W, float unit, float lo, float hi) {
  int b = blockIdx.y;
  int HW = H * W;
  Cam c = load_cam(K, b);
  const float* P = pose ? pose + (size_t)b * 16 : nullptr;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    float d = depth[(size_t)b * HW + i] * unit;  // float32 product, as numpy keeps float32 * python-int
    bool ok = d > lo && d < hi;
    double X = nan, Y = nan, Z = nan;
    if (ok) {
      int r = i / W, col = i - r * W;
      double z = (double)d;
      double x = ((double)col - (double)c.cx) * z / (double)c.fx;
      double y = ((double)r - (double)c.cy) * z / (double)c.fy;
      if (P) {
        // (p - t) @ R : out[j] = sum_k (p-t)[k] * R[k][j], dgemm fma chain in k order
        double px = x - (double)P[3], py = y - (double)P[7], pz = z - (double)P[11];
        X = fma(pz, (double)P[8], fma(py, (double)P[4], px * (double)P[0]));
        Y = fma(pz, (double)P[9], fma(py, (double)P[5], px * (double)P[1]));
        Z = fma(pz, (double)P[10], fma(py, (double)P[6], px * (double)P[2]));
      } else {
        X = x; Y = y; Z = z;
      }
    }
    double* o = xyz + ((size_t)b * HW + i) * 3;
    o[0] = X; o[1] = Y; o[2] = Z;
    valid[(size_t)b * HW + i] = ok ? 1 : 0;
  }
}

// ---- DepthAugment (dc:577-604) -------------------------------------------------------------------
// min over the valid (non-zero) neighbours of the 3x3 window, or over all in-frame neighbours when none is valid (the pool pads
// with -inf of -x: out-of-frame positions never win).  fminf is exact and order-independent on the finite inputs of this path, so
// the window may be walked in any order: a thread owns FOUR consecutive pixels of a row (round 6: three float4 row loads + six
// edge scalars instead of 36 scalar loads, three float4 stores; 16 Mpx: 2.3 -> see profiles/r06_*), W % 4 == 0, else one pixel.
__device__ inline void aug_min(float v, float& mv, float& mr) {
  mr = fminf(mr, v);
  if (v != 0.0f) mv = fminf(mv, v);
}
__global__ void depth_augment_kernel(const float* __restrict__ depth, float* __restrict__ out, int H, int W) {
  int b = blockIdx.y;
  int HW = H * W;
  const float* d = depth + (size_t)b * HW;
  float* o = out + (size_t)b * 3 * HW;
  const float inf = __uint_as_float(0x7f800000u);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    int r = i / W, col = i - r * W;
    float mv = inf, mr = inf;  // min over valid (non-zero) neighbours / over all neighbours (pool pads with -inf of -x)
    for (int dy = -1; dy <= 1; ++dy) {
      int rr = r + dy;
      if (rr < 0 || rr >= H) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        int cc = col + dx;
        if (cc < 0 || cc >= W) continue;
        aug_min(d[rr * W + cc], mv, mr);
      }
    }
    float m = (mv == inf) ? mr : mv;
    float c0 = d[i];
    o[i] = c0;
    o[HW + i] = m;
    o[2 * HW + i] = m - c0;
  }
}
__global__ __launch_bounds__(256) void depth_augment4_kernel(const float* __restrict__ depth, float* __restrict__ out, int H, int W) {
  const int b = blockIdx.y;
  const int HW = H * W, W4 = W >> 2, Q = HW >> 2;
  const float* d = depth + (size_t)b * HW;
  float* o = out + (size_t)b * 3 * HW;
  const float inf = __uint_as_float(0x7f800000u);
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < Q; q += gridDim.x * blockDim.x) {
    const int r = q / W4, c4 = (q - r * W4) << 2;
    float mv[4] = {inf, inf, inf, inf}, mr[4] = {inf, inf, inf, inf};
    float4 ctr = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int rr = r + dy;
      if (rr < 0 || rr >= H) continue;
      const float* row = d + (size_t)rr * W + c4;
      const float4 v = *reinterpret_cast<const float4*>(row);
      if (dy == 0) ctr = v;
      const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        aug_min(x[k], mv[k], mr[k]);
        if (k > 0) aug_min(x[k - 1], mv[k], mr[k]);
        if (k < 3) aug_min(x[k + 1], mv[k], mr[k]);
      }
      if (c4 > 0) aug_min(row[-1], mv[0], mr[0]);
      if (c4 + 4 < W) aug_min(row[4], mv[3], mr[3]);
    }
    const float c[4] = {ctr.x, ctr.y, ctr.z, ctr.w};
    float m[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = (mv[k] == inf) ? mr[k] : mv[k];
    const size_t i = (size_t)q << 2;
    *reinterpret_cast<float4*>(o + i) = ctr;
    *reinterpret_cast<float4*>(o + HW + i) = make_float4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<float4*>(o + 2 * (size_t)HW + i) = make_float4(m[0] - c[0], m[1] - c[1], m[2] - c[2], m[3] - c[3]);
  }
}

// ---- occlusion filter of the successive multi-view path (sd:446-463) -----------------------------
// min over the VALID pixels of the 3x3 window (the pool pads with -inf of -x: out-of-frame neighbours never win);
// a pixel more than `thr` behind that minimum is replaced by it.  (depth - min) < thr keeps the pixel; invalid pixels
// hold 0 and compare as 0 - min (or 0 - inf): always kept, exactly like the reference's expression.
__global__ void occlusion_filter_kernel(const float* __restrict__ depth, const uint8_t* __restrict__ mask,
                                        float* __restrict__ out, int H, int W, float thr) {
  const int b = blockIdx.y, HW = H * W;
  const float* d = depth + (size_t)b * HW;
  const uint8_t* m = mask + (size_t)b * HW;
  const float inf = __uint_as_float(0x7f800000u);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    const int r = i / W, col = i - r * W;
    float mn = inf;
    for (int dy = -1; dy <= 1; ++dy) {
      const int rr = r + dy;
      if (rr < 0 || rr >= H) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int cc = col + dx;
        if (cc < 0 || cc >= W) continue;
        if (m[rr * W + cc]) mn = fminf(mn, d[rr * W + cc]);
      }
    }
    const float c0 = d[i];
    out[(size_t)b * HW + i] = (c0 - mn) < thr ? c0 : mn;
  }
}

// ---- overlap ratio of generate_gt.py:68-102 ---------------------------------------------------------
// For every point of cloud A of a pair: is some point of cloud B strictly within `radius` (squared distance in
// float64, < r^2 like the KD-tree radius search)?  Clouds are a few thousand points after the 0.025 voxel grid, so
// the exact all-pairs test with B streamed through LDS is microseconds per pair on one CU and needs no grid or tree;
// one workgroup per (pair, direction, 256-query slab), a count per (pair, direction) by atomicAdd of integers (exact).
__global__ __launch_bounds__(256) void overlap_count_kernel(const double* __restrict__ pts, const int64_t* __restrict__ offs,
                                                            double r2, int32_t* __restrict__ counts) {
  __shared__ double tile[256 * 3];
  const int pair = blockIdx.y, dir = blockIdx.z;
  const int64_t a0 = offs[2 * pair + dir], a1 = offs[2 * pair + dir + 1];
  const int64_t b0 = dir == 0 ? offs[2 * pair + 1] : offs[2 * pair], b1 = dir == 0 ? offs[2 * pair + 2] : offs[2 * pair + 1];
  const int64_t q = a0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (a0 + (int64_t)blockIdx.x * 256 >= a1) return;            // whole slab beyond this cloud (uniform exit)
  const bool live = q < a1;
  double qx = 0, qy = 0, qz = 0;
  if (live) { qx = pts[3 * q]; qy = pts[3 * q + 1]; qz = pts[3 * q + 2]; }
  bool found = false;
  for (int64_t t0 = b0; t0 < b1; t0 += 256) {
    const int64_t n = min((int64_t)256, b1 - t0);
    __syncthreads();
    if (threadIdx.x < n) {
      tile[threadIdx.x] = pts[3 * (t0 + threadIdx.x)];
      tile[256 + threadIdx.x] = pts[3 * (t0 + threadIdx.x) + 1];
      tile[512 + threadIdx.x] = pts[3 * (t0 + threadIdx.x) + 2];
    }
    __syncthreads();
    if (!found) {
      for (int j = 0; j < (int)n; ++j) {
        const double dx = tile[j] - qx, dy = tile[256 + j] - qy, dz = tile[512 + j] - qz;
        if (dx * dx + dy * dy + dz * dz < r2) found = true;
      }
    }
  }
  const unsigned long long hits = __ballot(live && found);
  if ((threadIdx.x & 63) == 0 && hits) atomicAdd(counts + 2 * pair + dir, (int32_t)__popcll(hits));
}

// ---- mask application + condition assembly (sd:2564-2570, sd:2579-2581) --------------------------
__global__ void apply_mask_kernel(const float* __restrict__ prob, const float* __restrict__ depth,
                                  const uint8_t* __restrict__ hit, float thr, float* __restrict__ depth_out,
                                  uint8_t* __restrict__ hit_out, float* __restrict__ cond, int HW) {
  int b = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    size_t g = (size_t)b * HW + i;
    bool keep = prob[g] > thr;
    float d = keep ? depth[g] : 0.0f;
    bool m = keep && (hit ? hit[g] != 0 : true);
    if (depth_out) depth_out[g] = d;
    if (hit_out) hit_out[g] = m ? 1 : 0;
    if (cond) {
      cond[(size_t)b * 2 * HW + i] = d * 2.0f - 1.0f;
      cond[(size_t)b * 2 * HW + HW + i] = (m ? 1.0f : 0.0f) * 2.0f - 1.0f;
    }
  }
}

static inline dim3 grid_for(int n, int B, int threads = 256, int max_x = 1024) {
  int gx = ceil_div(n, threads);
  if (gx > max_x) gx = max_x;
  if (gx < 1) gx = 1;
  return dim3(gx, B, 1);
}

}  // namespace prg

using namespace prg;

extern "C" {

int prg_depth2pc(const float* depth, const float* K, float* pc, uint8_t* valid, int B, int H, int W, float clip_lo,
                 float clip_hi, float invalid_value, void* stream) {
  PRG_CHECK(depth && K && pc && valid, "prg_depth2pc: null pointer");
  PRG_CHECK(B > 0 && H > 0 && W > 0, "prg_depth2pc: bad shape");
  hipStream_t s = (hipStream_t)stream;
  int clip = clip_lo <= clip_hi;
  depth2pc_kernel<<<grid_for(H * W, B), 256, 0, s>>>(depth, K, pc, valid, H * W, W, clip_lo, clip_hi, invalid_value,
                                                    clip);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

int prg_pc2depth(const float* pc, const uint8_t* valid, const float* K, float* depth, uint8_t* mask, int B, int N,
                 int H, int W, void* stream) {
  PRG_CHECK((pc || N == 0) && K && depth, "prg_pc2depth: null pointer");
  PRG_CHECK(B > 0 && N >= 0 && H > 0 && W > 0, "prg_pc2depth: bad shape");
  hipStream_t s = (hipStream_t)stream;
  size_t n = (size_t)B * H * W;
  PRG_HIP(hipMemsetAsync(depth, 0xFF, n * sizeof(float), s));
  if (N > 0) {
    pc2depth_scatter_kernel<<<grid_for(N, B), 256, 0, s>>>(pc, valid, K, (uint32_t*)depth, N, H, W);
    PRG_LAUNCH_CHECK();
  }
  zbuf_resolve_kernel<<<grid_for((int)((n + 0) > 262144 * 4 ? 262144 * 4 : n), 1), 256, 0, s>>>(
      (const uint32_t*)depth, depth, mask, n, 1.0f, 0);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

int prg_project_points_zbuffer(const float* points, const int64_t* offsets, const float* pose, const float* K,
                               float* depth, uint8_t* mask, int B, int H, int W, float depth_scale, void* stream) {
  PRG_CHECK(points && offsets && K && depth, "prg_project_points_zbuffer: null pointer");
  PRG_CHECK(B > 0 && H > 0 && W > 0, "prg_project_points_zbuffer: bad shape");
  hipStream_t s = (hipStream_t)stream;
  size_t n = (size_t)B * H * W;
  PRG_HIP(hipMemsetAsync(depth, 0xFF, n * sizeof(float), s));
  project_points_kernel<<<dim3(256, B, 1), 256, 0, s>>>(points, offsets, pose, K, (uint32_t*)depth, H, W);
  PRG_LAUNCH_CHECK();
  zbuf_resolve_kernel<<<grid_for((int)(n > 262144 * 4 ? 262144 * 4 : n), 1), 256, 0, s>>>(
      (const uint32_t*)depth, depth, mask, n, depth_scale, depth_scale != 1.0f);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

int prg_reproject_zbuffer(const float* depth, const float* K, const float* pose, float* depth_out, uint8_t* mask_out,
                          int B, int H, int W, float depth_unit, float clip_lo, float clip_hi, float out_scale,
                          void* stream) {
  PRG_CHECK(depth && K && pose && depth_out, "prg_reproject_zbuffer: null pointer");
  PRG_CHECK(depth != depth_out, "prg_reproject_zbuffer: in-place not supported");
  PRG_CHECK(B > 0 && H > 0 && W > 0, "prg_reproject_zbuffer: bad shape");
  hipStream_t s = (hipStream_t)stream;
  size_t n = (size_t)B * H * W;
  PRG_HIP(hipMemsetAsync(depth_out, 0xFF, n * sizeof(float), s));
  reproject_scatter_kernel<<<grid_for(H * W, B), 256, 0, s>>>(depth, K, pose, (uint32_t*)depth_out, H, W, depth_unit,
                                                             clip_lo, clip_hi);
  PRG_LAUNCH_CHECK();
  zbuf_resolve_kernel<<<grid_for((int)(n > 262144 * 4 ? 262144 * 4 : n), 1), 256, 0, s>>>(
      (const uint32_t*)depth_out, depth_out, mask_out, n, out_scale, out_scale != 1.0f);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

int prg_unproject_f64(const float* depth, const float* K, const float* pose, double* xyz, uint8_t* valid, int B, int H,
                      int W, float depth_unit, float clip_lo, float clip_hi, void* stream) {
  PRG_CHECK(depth && K && xyz && valid, "prg_unproject_f64: null pointer");
  PRG_CHECK(B > 0 && H > 0 && W > 0, "prg_unproject_f64: bad shape");
  unproject_f64_kernel<<<grid_for(H * W, B), 256, 0, (hipStream_t)stream>>>(depth, K, pose, xyz, valid, H, W,
                                                                           depth_unit, clip_lo, clip_hi);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

int prg_depth_augment(const float* depth, float* out, int B, int H, int W, void* stream) {
  PRG_CHECK(depth && out, "prg_depth_augment: null pointer");
  PRG_CHECK(B > 0 && H > 0 && W > 0, "prg_depth_augment: bad shape");
  if (W % 4 == 0 && (reinterpret_cast<uintptr_t>(depth) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
    depth_augment4_kernel<<<grid_for(H * W / 4, B), 256, 0, (hipStream_t)stream>>>(depth, out, H, W);
  else
    depth_augment_kernel<<<grid_for(H * W, B), 256, 0, (hipStream_t)stream>>>(depth, out, H, W);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

int prg_apply_mask(const float* prob, const float* depth, const uint8_t* hit, float thr, float* depth_out,
                   uint8_t* hit_out, float* img_cond, int B, int H, int W, void* stream) {
  PRG_CHECK(prob && depth, "prg_apply_mask: null pointer");
  PRG_CHECK(B > 0 && H > 0 && W > 0, "prg_apply_mask: bad shape");
  apply_mask_kernel<<<grid_for(H * W, B), 256, 0, (hipStream_t)stream>>>(prob, depth, hit, thr, depth_out, hit_out,
                                                                        img_cond, H * W);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

int prg_occlusion_filter(const float* depth, const uint8_t* mask, float* out, int B, int H, int W, float threshold,
                         void* stream) {
  PRG_CHECK(depth && mask && out && depth != out, "prg_occlusion_filter: null or aliased pointer");
  PRG_CHECK(B > 0 && H > 0 && W > 0, "prg_occlusion_filter: bad shape");
  occlusion_filter_kernel<<<grid_for(H * W, B), 256, 0, (hipStream_t)stream>>>(depth, mask, out, H, W, threshold);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

int prg_overlap_counts(const double* pts, const int64_t* offsets, int n_pairs, int64_t max_cloud, double radius,
                       int32_t* counts, void* stream) {
  PRG_CHECK(pts && offsets && counts, "prg_overlap_counts: null pointer");
  PRG_CHECK(n_pairs > 0 && n_pairs <= 65535 && max_cloud > 0 && radius > 0, "prg_overlap_counts: bad arguments");
  PRG_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * 2 * (size_t)n_pairs, (hipStream_t)stream));
  const dim3 grid((unsigned)((max_cloud + 255) / 256), (unsigned)n_pairs, 2);
  overlap_count_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(pts, offsets, radius * radius, counts);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

}  // extern "C"
