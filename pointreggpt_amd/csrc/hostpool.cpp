// hostpool.cpp — the host half of Generator.generate's per-scene output (sd:2484-2500, 2586-2685): compaction of the
// unprojected views, rigid moves, bounding-box crop, voxel-grid mean down-sampling, and the PLY / PNG / text writers, as
// plain C++ behind the C-ABI plus a worker pool so that a batch's files are produced while the GPU samples the next
// batch.  (The reference does all of this serially in Python through open3d / torchvision / cv2 after every batch; at
// the GPU rates of this build that, not the sampler, would bound generate_dataset.py.)
//
// Semantics follow pointreggpt_amd/postprocess.py (its numpy forms are the specification and stay as the test
// reference): crop bounds inclusive; voxel index = floor((p - (min - voxel/2)) / voxel), output = per-voxel mean in
// ascending voxel-key order, points of a voxel summed in input order; PLY binary_little_endian `double x y z`;
// save_image = clamp(x*255 + 0.5) as 8-bit grey replicated to RGB; depth png = uint16(depth * 1e4); np.savetxt's
// "%.18e" rows.  open3d / torchvision / cv2 are absent from the image: parity-unpinned, as DESIGN.md states.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/prg.h"

namespace prg {
int fail(int code, const std::string& msg);
}
using prg::fail;

namespace {

struct P3 {
  double x, y, z;
};

// ---- geometry --------------------------------------------------------------------------------------------------------
void crop_aabb(std::vector<P3>& pts, const double lo[3], const double hi[3]) {
  size_t w = 0;
  for (size_t i = 0; i < pts.size(); ++i) {
    const P3& p = pts[i];
    if (p.x >= lo[0] && p.x <= hi[0] && p.y >= lo[1] && p.y <= hi[1] && p.z >= lo[2] && p.z <= hi[2]) pts[w++] = p;
  }
  pts.resize(w);
}

// p' = R p + t with the products summed left to right (numpy's `pts @ R.T + t` up to the BLAS's use of FMA)
void transform(std::vector<P3>& pts, const double* T /* 4x4 row-major */) {
  for (P3& p : pts) {
    const double x = p.x, y = p.y, z = p.z;
    p.x = x * T[0] + y * T[1] + z * T[2] + T[3];
    p.y = x * T[4] + y * T[5] + z * T[6] + T[7];
    p.z = x * T[8] + y * T[9] + z * T[10] + T[11];
  }
}

int voxel_down_sample(const std::vector<P3>& pts, double voxel, std::vector<P3>& out) {
  out.clear();
  const size_t n = pts.size();
  if (n == 0) return PRG_OK;
  if (!(voxel > 0)) return fail(PRG_E_INVALID, "voxel_down_sample: voxel_size <= 0");
  double mn[3] = {pts[0].x, pts[0].y, pts[0].z};
  for (const P3& p : pts) {
    if (!(std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z))) return fail(PRG_E_INVALID, "voxel_down_sample: non-finite point");
    mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z);
  }
  const double org[3] = {mn[0] - voxel * 0.5, mn[1] - voxel * 0.5, mn[2] - voxel * 0.5};
  std::vector<int64_t> ix(n), iy(n), iz(n);
  int64_t dx = 0, dy = 0, dz = 0;
  for (size_t i = 0; i < n; ++i) {
    ix[i] = (int64_t)std::floor((pts[i].x - org[0]) / voxel);
    iy[i] = (int64_t)std::floor((pts[i].y - org[1]) / voxel);
    iz[i] = (int64_t)std::floor((pts[i].z - org[2]) / voxel);
    dx = std::max(dx, ix[i] + 1); dy = std::max(dy, iy[i] + 1); dz = std::max(dz, iz[i] + 1);
  }
  if ((double)dx * (double)dy * (double)dz >= 4.611686018427388e18) return fail(PRG_E_INVALID, "voxel_down_sample: voxel_size is too small");
  std::vector<int64_t> key(n);
  for (size_t i = 0; i < n; ++i) key[i] = (ix[i] * dy + iy[i]) * dz + iz[i];
  std::vector<uint32_t> order(n);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
  size_t i = 0;
  while (i < n) {
    const int64_t k = key[order[i]];
    double sx = pts[order[i]].x, sy = pts[order[i]].y, sz = pts[order[i]].z;   // np.add.reduceat: first element, then += in order
    size_t j = i + 1;
    for (; j < n && key[order[j]] == k; ++j) {
      sx += pts[order[j]].x; sy += pts[order[j]].y; sz += pts[order[j]].z;
    }
    const double c = (double)(j - i);
    out.push_back(P3{sx / c, sy / c, sz / c});
    i = j;
  }
  return PRG_OK;
}

// ---- writers ---------------------------------------------------------------------------------------------------------
int write_file_atomic(const std::string& path, const std::string& head, const void* body, size_t nbytes) {
  const std::string tmp = path + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return fail(PRG_E_INVALID, "cannot open " + tmp);
  bool ok = std::fwrite(head.data(), 1, head.size(), f) == head.size();
  if (ok && nbytes) ok = std::fwrite(body, 1, nbytes, f) == nbytes;
  ok = std::fclose(f) == 0 && ok;
  if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) return fail(PRG_E_INVALID, "cannot write " + path);
  return PRG_OK;
}

int write_ply(const std::string& path, const std::vector<P3>& pts) {
  char head[256];
  std::snprintf(head, sizeof(head),
                "ply\nformat binary_little_endian 1.0\ncomment Created by pointreggpt_amd\nelement vertex %zu\n"
                "property double x\nproperty double y\nproperty double z\nend_header\n",
                pts.size());
  static_assert(sizeof(P3) == 24, "packed xyz");
  return write_file_atomic(path, head, pts.data(), pts.size() * sizeof(P3));   // little-endian host (x86-64)
}

void png_chunk(std::string& out, const char type[4], const unsigned char* data, size_t n) {
  unsigned char len[4] = {(unsigned char)(n >> 24), (unsigned char)(n >> 16), (unsigned char)(n >> 8), (unsigned char)n};
  out.append(reinterpret_cast<char*>(len), 4);
  uLong crc = crc32(0L, reinterpret_cast<const Bytef*>(type), 4);
  if (n) crc = crc32(crc, data, (uInt)n);
  out.append(type, 4);
  if (n) out.append(reinterpret_cast<const char*>(data), n);
  unsigned char c[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
  out.append(reinterpret_cast<char*>(c), 4);
}

// kind 0: 8-bit RGB (grey replicated), kind 1: 16-bit grey
int write_png(const std::string& path, const float* img, int H, int W, int kind) {
  const int bpp = kind == 0 ? 3 : 2;
  std::vector<unsigned char> raw((size_t)H * (1 + (size_t)W * bpp));
  for (int y = 0; y < H; ++y) {
    unsigned char* row = raw.data() + (size_t)y * (1 + (size_t)W * bpp);
    row[0] = 0;   // filter: none
    for (int x = 0; x < W; ++x) {
      const float v = img[(size_t)y * W + x];
      if (kind == 0) {
        float q = v * 255.0f + 0.5f;
        q = q < 0.0f ? 0.0f : (q > 255.0f ? 255.0f : q);
        const unsigned char u = (unsigned char)q;
        row[1 + 3 * x] = row[2 + 3 * x] = row[3 + 3 * x] = u;
      } else {
        const float q = v * 1e4f;
        const unsigned u = (unsigned)(q < 0.0f ? 0.0f : (q > 65535.0f ? 65535.0f : q));
        row[1 + 2 * x] = (unsigned char)(u >> 8);
        row[2 + 2 * x] = (unsigned char)u;
      }
    }
  }
  uLongf clen = compressBound((uLong)raw.size());
  std::vector<unsigned char> comp(clen);
  if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 3) != Z_OK) return fail(PRG_E_INVALID, "png: deflate failed");
  std::string out("\x89PNG\r\n\x1a\n", 8);
  unsigned char ihdr[13] = {(unsigned char)(W >> 24), (unsigned char)(W >> 16), (unsigned char)(W >> 8), (unsigned char)W,
                            (unsigned char)(H >> 24), (unsigned char)(H >> 16), (unsigned char)(H >> 8), (unsigned char)H,
                            (unsigned char)(kind == 0 ? 8 : 16), (unsigned char)(kind == 0 ? 2 : 0), 0, 0, 0};
  png_chunk(out, "IHDR", ihdr, 13);
  png_chunk(out, "IDAT", comp.data(), clen);
  png_chunk(out, "IEND", nullptr, 0);
  return write_file_atomic(path, out, nullptr, 0);
}

int write_text(const std::string& path, const double* v, int rows, int cols) {
  std::string out;
  char buf[64];
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      std::snprintf(buf, sizeof(buf), "%.18e", v[(size_t)r * cols + c]);   // np.savetxt's default fmt, ' ' delimiter
      out += buf;
      out += c + 1 < cols ? ' ' : '\n';
    }
  return write_file_atomic(path, out, nullptr, 0);
}

}  // namespace

// ---- pool ------------------------------------------------------------------------------------------------------------
struct prg_pool {
  std::vector<std::thread> workers;
  std::deque<std::function<int()>> queue;
  std::mutex mu;
  std::condition_variable cv_job, cv_idle;
  int in_flight = 0;
  bool stop = false;
  int first_error = PRG_OK;
  std::string first_msg;
  int64_t jobs_done = 0;

  void run() {
    for (;;) {
      std::function<int()> job;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return stop || !queue.empty(); });
        if (queue.empty()) return;
        job = std::move(queue.front());
        queue.pop_front();
        ++in_flight;
      }
      const int rc = job();
      {
        std::lock_guard<std::mutex> lk(mu);
        if (rc != PRG_OK && first_error == PRG_OK) {
          first_error = rc;
          first_msg = prg_last_error();   // thread-local of THIS worker: copied while still current
        }
        --in_flight;
        ++jobs_done;
        if (queue.empty() && in_flight == 0) cv_idle.notify_all();
      }
    }
  }
  void submit(std::function<int()> job) {
    {
      std::lock_guard<std::mutex> lk(mu);
      queue.push_back(std::move(job));
    }
    cv_job.notify_one();
  }
};

extern "C" {

int prg_host_crop_aabb(const double* pts, int64_t n, const double* lo, const double* hi, double* out, int64_t* n_out) {
  if (!pts && n > 0) return fail(PRG_E_INVALID, "prg_host_crop_aabb: null pointer");
  if (!lo || !hi || !out || !n_out || n < 0) return fail(PRG_E_INVALID, "prg_host_crop_aabb: bad arguments");
  std::vector<P3> v(reinterpret_cast<const P3*>(pts), reinterpret_cast<const P3*>(pts) + n);
  crop_aabb(v, lo, hi);
  if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(P3));
  *n_out = (int64_t)v.size();
  return PRG_OK;
}

int prg_host_voxel_down_sample(const double* pts, int64_t n, double voxel, double* out, int64_t* n_out) {
  if ((!pts && n > 0) || !out || !n_out || n < 0) return fail(PRG_E_INVALID, "prg_host_voxel_down_sample: bad arguments");
  std::vector<P3> v(reinterpret_cast<const P3*>(pts), reinterpret_cast<const P3*>(pts) + n), o;
  const int rc = voxel_down_sample(v, voxel, o);
  if (rc) return rc;
  if (!o.empty()) std::memcpy(out, o.data(), o.size() * sizeof(P3));
  *n_out = (int64_t)o.size();
  return PRG_OK;
}

int prg_host_write_ply(const char* path, const double* pts, int64_t n) {
  if (!path || (!pts && n > 0) || n < 0) return fail(PRG_E_INVALID, "prg_host_write_ply: bad arguments");
  std::vector<P3> v(reinterpret_cast<const P3*>(pts), reinterpret_cast<const P3*>(pts) + n);
  return write_ply(path, v);
}

int prg_pool_create(int n_threads, prg_pool** out) {
  if (!out || n_threads < 1 || n_threads > 256) return fail(PRG_E_INVALID, "prg_pool_create: bad arguments");
  std::unique_ptr<prg_pool> p(new prg_pool());
  for (int i = 0; i < n_threads; ++i) p->workers.emplace_back([q = p.get()] { q->run(); });
  *out = p.release();
  return PRG_OK;
}

int prg_pool_destroy(prg_pool* p) {
  if (!p) return PRG_OK;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->stop = true;
  }
  p->cv_job.notify_all();
  for (auto& t : p->workers) t.join();
  delete p;
  return PRG_OK;
}

int prg_pool_wait(prg_pool* p, int64_t* jobs_done) {
  if (!p) return fail(PRG_E_INVALID, "prg_pool_wait: null pool");
  std::unique_lock<std::mutex> lk(p->mu);
  p->cv_idle.wait(lk, [&] { return p->queue.empty() && p->in_flight == 0; });
  if (jobs_done) *jobs_done = p->jobs_done;
  if (p->first_error != PRG_OK) {
    const int rc = p->first_error;
    const std::string m = p->first_msg;
    p->first_error = PRG_OK;
    return fail(rc, "worker: " + m);
  }
  return PRG_OK;
}

int prg_pool_submit_cloud(prg_pool* p, const char* path, const double* xyz, int64_t n, const uint8_t* valid,
                          const double* T_pre, int crop, const double* lo, const double* hi, double voxel,
                          const double* T_post) {
  if (!p || !path || (!xyz && n > 0) || n < 0 || (crop && (!lo || !hi))) return fail(PRG_E_INVALID, "prg_pool_submit_cloud: bad arguments");
  auto pts = std::make_shared<std::vector<P3>>();
  pts->reserve((size_t)n);
  const P3* src = reinterpret_cast<const P3*>(xyz);
  for (int64_t i = 0; i < n; ++i)
    if (!valid || valid[i]) pts->push_back(src[i]);        // compaction in row-major order (sd:122-143)
  std::vector<double> pre(T_pre ? T_pre : nullptr, T_pre ? T_pre + 16 : nullptr), post(T_post ? T_post : nullptr, T_post ? T_post + 16 : nullptr);
  std::vector<double> blo(crop ? lo : nullptr, crop ? lo + 3 : nullptr), bhi(crop ? hi : nullptr, crop ? hi + 3 : nullptr);
  const std::string spath(path);
  p->submit([pts, pre, post, blo, bhi, crop, voxel, spath]() -> int {
    if (!pre.empty()) transform(*pts, pre.data());
    if (crop) crop_aabb(*pts, blo.data(), bhi.data());
    std::vector<P3> ds;
    const std::vector<P3>* outp = pts.get();
    if (voxel > 0) {
      const int rc = voxel_down_sample(*pts, voxel, ds);
      if (rc) return rc;
      outp = &ds;
    }
    if (!post.empty()) {
      if (outp != &ds) ds = *pts;
      transform(ds, post.data());
      outp = &ds;
    }
    return write_ply(spath, *outp);
  });
  return PRG_OK;
}

int prg_pool_submit_image(prg_pool* p, const char* path, const float* img, int H, int W, int kind) {
  if (!p || !path || !img || H <= 0 || W <= 0 || (kind != 0 && kind != 1)) return fail(PRG_E_INVALID, "prg_pool_submit_image: bad arguments");
  auto data = std::make_shared<std::vector<float>>(img, img + (size_t)H * W);
  const std::string spath(path);
  p->submit([data, spath, H, W, kind]() -> int { return write_png(spath, data->data(), H, W, kind); });
  return PRG_OK;
}

int prg_pool_submit_text(prg_pool* p, const char* path, const double* values, int rows, int cols) {
  if (!p || !path || !values || rows <= 0 || cols <= 0) return fail(PRG_E_INVALID, "prg_pool_submit_text: bad arguments");
  auto data = std::make_shared<std::vector<double>>(values, values + (size_t)rows * cols);
  const std::string spath(path);
  p->submit([data, spath, rows, cols]() -> int { return write_text(spath, data->data(), rows, cols); });
  return PRG_OK;
}

}  // extern "C"
