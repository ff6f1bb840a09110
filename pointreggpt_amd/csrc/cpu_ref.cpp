// cpu_ref.cpp — libprg_cpu.so: the prg_cpu_* twins of include/prg_cpu.h.  Plain C++ / OpenMP
// restatement of the hot path with HOST pointers: what BASELINE configs[0] ("on CPU, 64x64, 50-step DDIM, no GPU") runs when
// it is asked for with `--device cpu`, and a native second opinion next to the torch oracle.  NOT a fallback: nothing in
// libprg_hip.so or in pointreggpt_amd selects it implicitly.
//
// Built with g++ -O2 -fopenmp -ffp-contract=off: the geometry functions evaluate the reference's float expressions in the
// reference's order (fused multiply-adds only where the host BLAS uses them, exactly like geometry.hip) and are bit-exact
// against the fixtures; the networks accumulate in float64 and round once (closer to exact arithmetic than oneDNN's fp32,
// compared at 1e-4 like the HIP parity mode).
// sd = denoising_diffusion_pytorch/successive_ddnm_diffusion.py, dc = depth_correction_pytorch/depth_correction.py
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/prg_cpu.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& m) {
  g_err = m;
  return code;
}
#define CPU_CHECK(cond, msg) \
  do {                       \
    if (!(cond)) return fail(PRG_E_INVALID, std::string(msg) + " [" #cond "]"); \
  } while (0)

// ------------------------------------------------------------------------------------------------------------------
// geometry (arithmetic of geometry.hip, which is the reference's: sd:122-286)
// ------------------------------------------------------------------------------------------------------------------
struct Cam {
  float fx, fy, cx, cy;
};
inline Cam load_cam(const float* K, int b) {
  const float* k = K + (size_t)b * 9;
  return Cam{k[0], k[4], k[2], k[5]};
}
// `pc @ R^T + t` as the host BLAS evaluates it: fma chain over k, then a separate add
inline void se3_apply(const float* P, float x, float y, float z, float& ox, float& oy, float& oz) {
  float v;
  v = x * P[0]; v = std::fmaf(y, P[1], v); v = std::fmaf(z, P[2], v); ox = v + P[3];
  v = x * P[4]; v = std::fmaf(y, P[5], v); v = std::fmaf(z, P[6], v); oy = v + P[7];
  v = x * P[8]; v = std::fmaf(y, P[9], v); v = std::fmaf(z, P[10], v); oz = v + P[11];
}
constexpr float kInf = std::numeric_limits<float>::infinity();
// z-buffer of one image: min z per pixel (sd:225-258); `zb` holds +inf where nothing landed yet
inline void splat(float* zb, const Cam& c, float x, float y, float z, int H, int W) {
  if (!(z > 0.0f)) return;
  const float fc = std::nearbyintf(x * c.fx / z + c.cx);   // torch.round: half to even (default rounding mode)
  const float fr = std::nearbyintf(y * c.fy / z + c.cy);
  if (!(fc >= 0.0f && fc < (float)W && fr >= 0.0f && fr < (float)H)) return;
  float& d = zb[(size_t)(int)fr * W + (int)fc];
  if (z < d) d = z;
}
inline void resolve(float* zb, uint8_t* mask, size_t n, float scale, bool do_scale) {
  for (size_t i = 0; i < n; ++i) {
    const bool hit = zb[i] != kInf;
    float z = hit ? zb[i] : 0.0f;
    if (do_scale) z = z * scale;
    zb[i] = z;
    if (mask) mask[i] = hit ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// network building blocks, NCHW float32 tensors as flat vectors
// ------------------------------------------------------------------------------------------------------------------
using Vec = std::vector<float>;

// out (B,Cout,Ho,Wo) = conv(in (B,Cin,H,W) [nearest x2 when ups], w (Cout,Cin,K,K)) + bias; float64 accumulation
void conv2d(const float* in, int B, int Cin, int H, int W, const float* w, const float* bias, int Cout, int K, int stride,
            int pad, int ups, Vec& out, int& Ho, int& Wo) {
  const int Hl = ups ? 2 * H : H, Wl = ups ? 2 * W : W;
  Ho = (Hl + 2 * pad - K) / stride + 1;
  Wo = (Wl + 2 * pad - K) / stride + 1;
  out.assign((size_t)B * Cout * Ho * Wo, 0.0f);
  const int ho = Ho, wo = Wo;
#pragma omp parallel
  {
    std::vector<double> acc((size_t)ho * wo);
#pragma omp for collapse(2) schedule(dynamic)
    for (int b = 0; b < B; ++b)
      for (int co = 0; co < Cout; ++co) {
        std::fill(acc.begin(), acc.end(), bias ? (double)bias[co] : 0.0);
        for (int ci = 0; ci < Cin; ++ci) {
          const float* src = in + ((size_t)b * Cin + ci) * H * W;
          const float* wk = w + ((size_t)co * Cin + ci) * K * K;
          for (int kh = 0; kh < K; ++kh)
            for (int kw = 0; kw < K; ++kw) {
              const double wv = (double)wk[kh * K + kw];
              for (int oy = 0; oy < ho; ++oy) {
                const int iy = oy * stride + kh - pad;
                if (iy < 0 || iy >= Hl) continue;
                const float* row = src + (size_t)(ups ? iy >> 1 : iy) * W;
                double* arow = acc.data() + (size_t)oy * wo;
                // ox range with 0 <= ox*stride + kw - pad < Wl
                int ox0 = 0;
                while (ox0 < wo && ox0 * stride + kw - pad < 0) ++ox0;
                int ox1 = wo;
                while (ox1 > ox0 && (ox1 - 1) * stride + kw - pad >= Wl) --ox1;
                if (stride == 1 && !ups) {
                  const float* r0 = row + kw - pad;
                  for (int ox = ox0; ox < ox1; ++ox) arow[ox] += wv * (double)r0[ox];
                } else {
                  for (int ox = ox0; ox < ox1; ++ox) {
                    const int ix = ox * stride + kw - pad;
                    arow[ox] += wv * (double)row[ups ? ix >> 1 : ix];
                  }
                }
              }
            }
        }
        float* o = out.data() + ((size_t)b * Cout + co) * ho * wo;
        for (size_t i = 0; i < (size_t)ho * wo; ++i) o[i] = (float)acc[i];
      }
  }
}

// weight standardisation per output channel, biased variance, eps 1e-5 (sd:601-616)
Vec standardize(const float* w, int Cout, int K) {
  Vec o((size_t)Cout * K);
  for (int c = 0; c < Cout; ++c) {
    double m = 0;
    for (int k = 0; k < K; ++k) m += w[(size_t)c * K + k];
    m /= K;
    double v = 0;
    for (int k = 0; k < K; ++k) { const double d = w[(size_t)c * K + k] - m; v += d * d; }
    v /= K;
    const double rs = 1.0 / std::sqrt(v + 1e-5);
    for (int k = 0; k < K; ++k) o[(size_t)c * K + k] = (float)((w[(size_t)c * K + k] - m) * rs);
  }
  return o;
}

inline float silu(float x) { return x / (1.0f + std::exp(-x)); }
inline float gelu(float x) { return 0.5f * x * (1.0f + std::erf(x * 0.70710678118654752440f)); }

// GroupNorm (eps 1e-5) [+ (scale+1, shift) per (image, channel)] + SiLU, in place (sd:681-697)
void group_norm_silu(Vec& x, int B, int C, int HW, int G, const float* gamma, const float* beta, const float* ss, int ss_stride) {
  const int cpg = C / G;
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int g = 0; g < G; ++g) {
      float* p = x.data() + ((size_t)b * C + (size_t)g * cpg) * HW;
      const size_t n = (size_t)cpg * HW;
      double s = 0, q = 0;
      for (size_t i = 0; i < n; ++i) { s += p[i]; q += (double)p[i] * p[i]; }
      const double mean = s / n;
      double var = q / n - mean * mean;
      if (var < 0) var = 0;
      const float fm = (float)mean, rs = (float)(1.0 / std::sqrt(var + 1e-5));
      for (int c = 0; c < cpg; ++c) {
        const int ch = g * cpg + c;
        const float ga = gamma[ch], be = beta[ch];
        float sc = 0.0f, sh = 0.0f;
        if (ss) { sc = ss[(size_t)b * ss_stride + ch]; sh = ss[(size_t)b * ss_stride + C + ch]; }
        float* pc = p + (size_t)c * HW;
        for (int i = 0; i < HW; ++i) {
          float y = (pc[i] - fm) * rs * ga + be;
          if (ss) y = y * (sc + 1.0f) + sh;
          pc[i] = silu(y);
        }
      }
    }
}

// per-pixel LayerNorm over channels, gain only (sd:619-628)
void channel_layernorm(const Vec& x, Vec& y, int B, int C, int HW, const float* g) {
  y.resize(x.size());
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < HW; ++i) {
      const float* p = x.data() + (size_t)b * C * HW + i;
      double s = 0, q = 0;
      for (int c = 0; c < C; ++c) { const double v = p[(size_t)c * HW]; s += v; q += v * v; }
      const double mean = s / C;
      double var = q / C - mean * mean;
      if (var < 0) var = 0;
      const float fm = (float)mean, rs = (float)(1.0 / std::sqrt(var + 1e-5));
      float* o = y.data() + (size_t)b * C * HW + i;
      for (int c = 0; c < C; ++c) o[(size_t)c * HW] = (p[(size_t)c * HW] - fm) * rs * g[c];
    }
}

// y[r][o] = sum_i x[r][i] W[o][i] + b[o]
void linear(const float* x, int R, int I, const float* W, const float* b, int O, float* y) {
  for (int r = 0; r < R; ++r)
    for (int o = 0; o < O; ++o) {
      double a = b ? (double)b[o] : 0.0;
      for (int i = 0; i < I; ++i) a += (double)x[(size_t)r * I + i] * W[(size_t)o * I + i];
      y[(size_t)r * O + o] = (float)a;
    }
}

constexpr int kHeads = 4, kDh = 32, kHid = 128;

// ------------------------------------------------------------------------------------------------------------------
// parameter walk over the flat state dict (order of weights.param_spec / the reference's state_dict)
// ------------------------------------------------------------------------------------------------------------------
struct ConvP { int64_t w = -1, b = -1; int Cout = 0, Cin = 0, K = 0; };
struct ResP { int cin = 0, cout = 0; int64_t mlp_w = -1, mlp_b = -1; ConvP c1, c2, res; int64_t g1 = 0, b1 = 0, g2 = 0, b2 = 0; bool has_res = false; };
struct AttnP { int C = 0; bool linear = true; ConvP qkv, out; int64_t out_g = -1, norm_g = -1; };
struct LevelP { ResP r0, r1; AttnP at; ConvP resample; bool strided = false; };
struct Cursor {
  int64_t pos = 0;
  int64_t take(int64_t n) { const int64_t p = pos; pos += n; return p; }
};
void walk_conv(Cursor& c, ConvP& p, int Cout, int Cin, int K, bool bias) {
  p.Cout = Cout; p.Cin = Cin; p.K = K;
  p.w = c.take((int64_t)Cout * Cin * K * K);
  p.b = bias ? c.take(Cout) : -1;
}
void walk_res(Cursor& c, ResP& r, int cin, int cout, bool cond, int emb) {
  r.cin = cin; r.cout = cout;
  if (cond) { r.mlp_w = c.take((int64_t)2 * cout * 2 * emb); r.mlp_b = c.take(2 * cout); }
  walk_conv(c, r.c1, cout, cin, 3, true);
  r.g1 = c.take(cout); r.b1 = c.take(cout);
  walk_conv(c, r.c2, cout, cout, 3, true);
  r.g2 = c.take(cout); r.b2 = c.take(cout);
  r.has_res = cin != cout;
  if (r.has_res) walk_conv(c, r.res, cout, cin, 1, true);
}
void walk_attn(Cursor& c, AttnP& a, int C, bool lin) {
  a.C = C; a.linear = lin;
  walk_conv(c, a.qkv, 3 * kHid, C, 1, false);
  walk_conv(c, a.out, C, kHid, 1, true);
  if (lin) a.out_g = c.take(C);
  a.norm_g = c.take(C);
}

}  // namespace

struct prg_cpu_unet {
  prg_unet_config cfg;
  int L = 0, emb = 0;
  std::vector<int> dims;
  int64_t stem_w = 0, stem_b = 0, tm1_w = 0, tm1_b = 0, tm3_w = 0, tm3_b = 0, pm0_w = 0, pm0_b = 0, pm2_w = 0, pm2_b = 0;
  std::vector<LevelP> downs, ups;
  ResP mid1, mid2, fin;
  AttnP mid_at;
  int64_t head_w = 0, head_b = 0, total = 0;
  Vec flat;               // the state dict; the Block conv weights are standardised in place at create
  Vec freqs;              // SinusoidalPosEmb frequencies

  const float* F(int64_t o) const { return o < 0 ? nullptr : flat.data() + o; }

  void conv(const ConvP& p, const Vec& in, int B, int H, int W, int stride, int pad, int ups, Vec& out, int& Ho, int& Wo) const {
    conv2d(in.data(), B, p.Cin, H, W, F(p.w), F(p.b), p.Cout, p.K, stride, pad, ups, out, Ho, Wo);
  }

  // ResnetBlock (sd:700-734 / dc:726-740): x (B,cin,H,W) -> (B,cout,H,W); cond (B, 2 emb) or null
  void resblock(const ResP& r, const Vec& x, const float* cond, int B, int H, int W, Vec& out) const {
    const int HW = H * W, G = cfg.groups;
    Vec ss;
    if (cond && r.mlp_w >= 0) {
      Vec sc((size_t)B * 2 * emb);
      for (size_t i = 0; i < sc.size(); ++i) sc[i] = silu(cond[i]);
      ss.resize((size_t)B * 2 * r.cout);
      linear(sc.data(), B, 2 * emb, F(r.mlp_w), F(r.mlp_b), 2 * r.cout, ss.data());
    }
    Vec h1, h2;
    int ho, wo;
    conv(r.c1, x, B, H, W, 1, 1, 0, h1, ho, wo);
    group_norm_silu(h1, B, r.cout, HW, G, F(r.g1), F(r.b1), ss.empty() ? nullptr : ss.data(), 2 * r.cout);
    conv(r.c2, h1, B, H, W, 1, 1, 0, h2, ho, wo);
    group_norm_silu(h2, B, r.cout, HW, G, F(r.g2), F(r.b2), nullptr, 0);
    if (r.has_res) {
      Vec rs;
      conv(r.res, x, B, H, W, 1, 0, 0, rs, ho, wo);
      for (size_t i = 0; i < h2.size(); ++i) h2[i] += rs[i];
    } else {
      for (size_t i = 0; i < h2.size(); ++i) h2[i] += x[i];
    }
    out.swap(h2);
  }

  // Residual(PreNorm(LinearAttention | Attention)) (sd:583-589, 631-639, 737-796)
  void attention(const AttnP& a, const Vec& x, int B, int H, int W, Vec& out) const {
    const int N = H * W, C = a.C;
    Vec xn, qkv, o((size_t)B * kHid * N);
    channel_layernorm(x, xn, B, C, N, F(a.norm_g));
    int ho, wo;
    conv(a.qkv, xn, B, H, W, 1, 0, 0, qkv, ho, wo);
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
      for (int h = 0; h < kHeads; ++h) {
        const float* q = qkv.data() + ((size_t)b * 3 * kHid + (size_t)h * kDh) * N;
        const float* k = q + (size_t)kHid * N;
        const float* v = k + (size_t)kHid * N;
        float* oo = o.data() + ((size_t)b * kHid + (size_t)h * kDh) * N;
        if (a.linear) {
          // softmax(q) over d (per pixel) * 32^-1/2; softmax(k) over pixels; v / N; ctx[d][e] = sum_n k v; out[e][n] = sum_d ctx q
          std::vector<double> ks((size_t)kDh * N), ctx((size_t)kDh * kDh, 0.0);
          for (int d = 0; d < kDh; ++d) {
            float m = -kInf;
            for (int n = 0; n < N; ++n) m = std::max(m, k[(size_t)d * N + n]);
            double s = 0;
            for (int n = 0; n < N; ++n) { const double e = std::exp((double)(k[(size_t)d * N + n] - m)); ks[(size_t)d * N + n] = e; s += e; }
            for (int n = 0; n < N; ++n) ks[(size_t)d * N + n] /= s;
          }
          for (int d = 0; d < kDh; ++d)
            for (int e = 0; e < kDh; ++e) {
              double acc = 0;
              for (int n = 0; n < N; ++n) acc += ks[(size_t)d * N + n] * ((double)v[(size_t)e * N + n] / (double)N);
              ctx[(size_t)d * kDh + e] = acc;
            }
          for (int n = 0; n < N; ++n) {
            float m = -kInf;
            for (int d = 0; d < kDh; ++d) m = std::max(m, q[(size_t)d * N + n]);
            double qs[kDh], s = 0;
            for (int d = 0; d < kDh; ++d) { qs[d] = std::exp((double)(q[(size_t)d * N + n] - m)); s += qs[d]; }
            for (int d = 0; d < kDh; ++d) qs[d] = qs[d] / s * 0.17677669529663687;
            for (int e = 0; e < kDh; ++e) {
              double acc = 0;
              for (int d = 0; d < kDh; ++d) acc += ctx[(size_t)d * kDh + e] * qs[d];
              oo[(size_t)e * N + n] = (float)acc;
            }
          }
        } else {
          // softmax_j(q_i . k_j / sqrt(32)) v_j
          std::vector<double> sim(N);
          for (int i = 0; i < N; ++i) {
            double m = -1e300;
            for (int j = 0; j < N; ++j) {
              double acc = 0;
              for (int d = 0; d < kDh; ++d) acc += (double)(q[(size_t)d * N + i] * 0.17677669529663687f) * k[(size_t)d * N + j];
              sim[j] = acc;
              m = std::max(m, acc);
            }
            double s = 0;
            for (int j = 0; j < N; ++j) { sim[j] = std::exp(sim[j] - m); s += sim[j]; }
            for (int d = 0; d < kDh; ++d) {
              double acc = 0;
              for (int j = 0; j < N; ++j) acc += sim[j] * v[(size_t)d * N + j];
              oo[(size_t)d * N + i] = (float)(acc / s);
            }
          }
        }
      }
    Vec y;
    conv(a.out, o, B, H, W, 1, 0, 0, y, ho, wo);
    if (a.linear) {
      Vec yn;
      channel_layernorm(y, yn, B, C, N, F(a.out_g));
      y.swap(yn);
    }
    out.resize(y.size());
    for (size_t i = 0; i < y.size(); ++i) out[i] = y[i] + x[i];
  }

  // trunk shared by the two networks; x (B,in_channels,S,S); cond (B, 2 emb) or null -> (B,1,S,S) before the optional sigmoid
  int forward(const float* x_in, const float* cond, float* out, int B, int S) const {
    const int nl = L, d0 = cfg.dim;
    if (S % (1 << (nl - 1)) != 0 || (S >> (nl - 1)) < 2) return fail(PRG_E_INVALID, "cpu forward: image size too small for the level count");
    Vec xin(x_in, x_in + (size_t)B * cfg.in_channels * S * S), x0, x;
    int H = S, ho, wo;
    ConvP stem;
    stem.w = stem_w; stem.b = stem_b; stem.Cout = d0; stem.Cin = cfg.in_channels; stem.K = 7;
    conv(stem, xin, B, S, S, 1, 3, 0, x0, ho, wo);
    x = x0;
    std::vector<std::pair<Vec, int>> skips;
    for (int i = 0; i < nl; ++i) {
      const LevelP& lv = downs[i];
      Vec s1, t, s2, xd;
      resblock(lv.r0, x, cond, B, H, H, s1);
      resblock(lv.r1, s1, cond, B, H, H, t);
      attention(lv.at, t, B, H, H, s2);
      conv(lv.resample, s2, B, H, H, lv.strided ? 2 : 1, 1, 0, xd, ho, wo);
      skips.push_back({std::move(s1), dims[i]});
      skips.push_back({std::move(s2), dims[i]});
      x.swap(xd);
      H = ho;
    }
    {
      Vec m1, m2, m3;
      resblock(mid1, x, cond, B, H, H, m1);
      attention(mid_at, m1, B, H, H, m2);
      resblock(mid2, m2, cond, B, H, H, m3);
      x.swap(m3);
    }
    auto cat = [&](const Vec& a, int Ca, const Vec& b, int Cb, int HW) {
      Vec o((size_t)B * (Ca + Cb) * HW);
      for (int bb = 0; bb < B; ++bb) {
        std::memcpy(o.data() + (size_t)bb * (Ca + Cb) * HW, a.data() + (size_t)bb * Ca * HW, sizeof(float) * Ca * HW);
        std::memcpy(o.data() + ((size_t)bb * (Ca + Cb) + Ca) * HW, b.data() + (size_t)bb * Cb * HW, sizeof(float) * Cb * HW);
      }
      return o;
    };
    for (int i = 0; i < nl; ++i) {
      const LevelP& lv = ups[i];
      const int Co = dims[nl - i];
      Vec u1, u2, u3, xu;
      {
        auto sk = std::move(skips.back()); skips.pop_back();
        resblock(lv.r0, cat(x, Co, sk.first, sk.second, H * H), cond, B, H, H, u1);
      }
      {
        auto sk = std::move(skips.back()); skips.pop_back();
        resblock(lv.r1, cat(u1, Co, sk.first, sk.second, H * H), cond, B, H, H, u2);
      }
      attention(lv.at, u2, B, H, H, u3);
      conv(lv.resample, u3, B, H, H, 1, 1, lv.strided ? 1 : 0, xu, ho, wo);
      x.swap(xu);
      H = ho;
    }
    Vec fr, y;
    resblock(fin, cat(x, d0, x0, d0, H * H), cond, B, H, H, fr);
    ConvP head;
    head.w = head_w; head.b = head_b; head.Cout = 1; head.Cin = d0; head.K = 1;
    conv(head, fr, B, H, H, 1, 0, 0, y, ho, wo);
    for (size_t i = 0; i < y.size(); ++i) out[i] = cfg.sigmoid_out ? 1.0f / (1.0f + std::exp(-y[i])) : y[i];
    return PRG_OK;
  }

  // cond = cat[time_mlp(t), param_mlp(K)] (sd:845-856, 925-932): Linear - GELU(erf) - Linear each
  void conditioning(const int64_t* time, const float* pc, int B, Vec& cond) const {
    const int d0 = cfg.dim, e = emb, half = d0 / 2;
    cond.assign((size_t)B * 2 * e, 0.0f);
    Vec sinu((size_t)B * d0), h1((size_t)B * e), te((size_t)B * e), h2((size_t)B * e), pe((size_t)B * e);
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < half; ++i) {
        const float a = (float)time[b] * freqs[i];
        sinu[(size_t)b * d0 + i] = std::sin(a);
        sinu[(size_t)b * d0 + half + i] = std::cos(a);
      }
    linear(sinu.data(), B, d0, F(tm1_w), F(tm1_b), e, h1.data());
    for (auto& v : h1) v = gelu(v);
    linear(h1.data(), B, e, F(tm3_w), F(tm3_b), e, te.data());
    linear(pc, B, cfg.param_cond_dim, F(pm0_w), F(pm0_b), e, h2.data());
    for (auto& v : h2) v = gelu(v);
    linear(h2.data(), B, e, F(pm2_w), F(pm2_b), e, pe.data());
    for (int b = 0; b < B; ++b) {
      std::memcpy(cond.data() + (size_t)b * 2 * e, te.data() + (size_t)b * e, sizeof(float) * e);
      std::memcpy(cond.data() + (size_t)b * 2 * e + e, pe.data() + (size_t)b * e, sizeof(float) * e);
    }
  }
};

namespace {

int build(prg_cpu_unet& u, const prg_unet_config& cfg) {
  CPU_CHECK(cfg.dim >= 8 && cfg.dim % 8 == 0, "config: dim must be a multiple of 8");
  CPU_CHECK(cfg.n_levels >= 1 && cfg.n_levels <= 8, "config: n_levels out of range");
  CPU_CHECK(cfg.in_channels == 1 || cfg.in_channels == 3, "config: in_channels must be 1 or 3");
  CPU_CHECK(cfg.groups >= 1 && cfg.groups <= 64, "config: groups out of range");
  u.cfg = cfg;
  u.L = cfg.n_levels;
  u.emb = cfg.dim * 4;
  u.dims.assign(1, cfg.dim);
  for (int i = 0; i < cfg.n_levels; ++i) u.dims.push_back(cfg.dim * cfg.dim_mults[i]);
  for (int d : u.dims) CPU_CHECK(d % cfg.groups == 0, "config: width not divisible by groups");
  const bool cond = cfg.conditional != 0;
  Cursor c;
  const int d0 = cfg.dim, e = u.emb;
  u.stem_w = c.take((int64_t)d0 * cfg.in_channels * 49);
  u.stem_b = c.take(d0);
  if (cond) {
    u.tm1_w = c.take((int64_t)e * d0); u.tm1_b = c.take(e);
    u.tm3_w = c.take((int64_t)e * e); u.tm3_b = c.take(e);
    u.pm0_w = c.take((int64_t)e * cfg.param_cond_dim); u.pm0_b = c.take(e);
    u.pm2_w = c.take((int64_t)e * e); u.pm2_b = c.take(e);
  }
  u.downs.resize(u.L);
  u.ups.resize(u.L);
  for (int i = 0; i < u.L; ++i) {
    const int ci = u.dims[i], co = u.dims[i + 1];
    LevelP& lv = u.downs[i];
    walk_res(c, lv.r0, ci, ci, cond, e);
    walk_res(c, lv.r1, ci, ci, cond, e);
    walk_attn(c, lv.at, ci, true);
    lv.strided = i != u.L - 1;
    walk_conv(c, lv.resample, co, ci, lv.strided ? 4 : 3, true);
  }
  for (int i = 0; i < u.L; ++i) {
    const int ci = u.dims[u.L - 1 - i], co = u.dims[u.L - i];
    LevelP& lv = u.ups[i];
    walk_res(c, lv.r0, co + ci, co, cond, e);
    walk_res(c, lv.r1, co + ci, co, cond, e);
    walk_attn(c, lv.at, co, true);
    lv.strided = i != u.L - 1;
    walk_conv(c, lv.resample, ci, co, 3, true);
  }
  const int mid = u.dims.back();
  walk_res(c, u.mid1, mid, mid, cond, e);
  walk_attn(c, u.mid_at, mid, false);
  walk_res(c, u.mid2, mid, mid, cond, e);
  walk_res(c, u.fin, 2 * d0, d0, cond, e);
  u.head_w = c.take(d0);
  u.head_b = c.take(1);
  u.total = c.pos;
  return PRG_OK;
}

// ---- Philox4x32-10, the device generator's counter layout (sampler.hip) ----
inline void philox4x32_10(uint32_t (&c)[4], uint64_t key) {
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
inline void philox_normal4(uint64_t key, uint32_t draw, uint32_t quad, float (&o)[4]) {
  uint32_t c[4] = {quad, draw, 0x70726721u, 0u};
  philox4x32_10(c, key);
  const float k = 2.3283064365386963e-10f;
  const float u0 = ((float)c[0] + 0.5f) * k, u1 = ((float)c[1] + 0.5f) * k, u2 = ((float)c[2] + 0.5f) * k, u3 = ((float)c[3] + 0.5f) * k;
  const float r0 = std::sqrt(-2.0f * std::log(std::min(std::max(u0, 1e-12f), 1.0f)));
  const float r1 = std::sqrt(-2.0f * std::log(std::min(std::max(u2, 1e-12f), 1.0f)));
  o[0] = r0 * std::cos(6.283185307179586f * u1); o[1] = r0 * std::sin(6.283185307179586f * u1);
  o[2] = r1 * std::cos(6.283185307179586f * u3); o[3] = r1 * std::sin(6.283185307179586f * u3);
}
inline float clamp1(float v) { return std::min(std::max(v, -1.0f), 1.0f); }

}  // namespace

extern "C" {

const char* prg_cpu_last_error(void) { return g_err.c_str(); }

int prg_cpu_depth2pc(const float* depth, const float* K, float* pc, uint8_t* valid, int B, int H, int W, float lo, float hi,
                     float inval) {
  CPU_CHECK(depth && K && pc && valid && B > 0 && H > 0 && W > 0, "prg_cpu_depth2pc: bad arguments");
  const int HW = H * W;
  const bool clip = lo <= hi;
  for (int b = 0; b < B; ++b) {
    const Cam c = load_cam(K, b);
    for (int i = 0; i < HW; ++i) {
      const float d = depth[(size_t)b * HW + i];
      const bool ok = clip ? (d > lo && d < hi) : true;
      const int r = i / W, col = i - r * W;
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
  return PRG_OK;
}

int prg_cpu_pc2depth(const float* pc, const uint8_t* valid, const float* K, float* depth, uint8_t* mask, int B, int N, int H,
                     int W) {
  CPU_CHECK((pc || N == 0) && K && depth && B > 0 && N >= 0 && H > 0 && W > 0, "prg_cpu_pc2depth: bad arguments");
  const size_t HW = (size_t)H * W;
  std::fill(depth, depth + (size_t)B * HW, kInf);
  for (int b = 0; b < B; ++b) {
    const Cam c = load_cam(K, b);
    for (int i = 0; i < N; ++i) {
      if (valid && !valid[(size_t)b * N + i]) continue;
      const float* p = pc + ((size_t)b * N + i) * 3;
      splat(depth + (size_t)b * HW, c, p[0], p[1], p[2], H, W);
    }
  }
  resolve(depth, mask, (size_t)B * HW, 1.0f, false);
  return PRG_OK;
}

int prg_cpu_project_points_zbuffer(const float* pts, const int64_t* offs, const float* pose, const float* K, float* depth,
                                   uint8_t* mask, int B, int H, int W, float depth_scale) {
  CPU_CHECK(pts && offs && K && depth && B > 0 && H > 0 && W > 0, "prg_cpu_project_points_zbuffer: bad arguments");
  const size_t HW = (size_t)H * W;
  std::fill(depth, depth + (size_t)B * HW, kInf);
  for (int b = 0; b < B; ++b) {
    const Cam c = load_cam(K, b);
    const float* P = pose ? pose + (size_t)b * 16 : nullptr;
    for (int64_t i = offs[b]; i < offs[b + 1]; ++i) {
      float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
      if (P) {
        float ox, oy, oz;
        se3_apply(P, x, y, z, ox, oy, oz);
        x = ox; y = oy; z = oz;
      }
      splat(depth + (size_t)b * HW, c, x, y, z, H, W);
    }
  }
  resolve(depth, mask, (size_t)B * HW, depth_scale, depth_scale != 1.0f);
  return PRG_OK;
}

int prg_cpu_reproject_zbuffer(const float* depth, const float* K, const float* pose, float* out, uint8_t* mask, int B, int H,
                              int W, float unit, float lo, float hi, float out_scale) {
  CPU_CHECK(depth && K && pose && out && depth != out && B > 0 && H > 0 && W > 0, "prg_cpu_reproject_zbuffer: bad arguments");
  const int HW = H * W;
  std::fill(out, out + (size_t)B * HW, kInf);
  for (int b = 0; b < B; ++b) {
    const Cam c = load_cam(K, b);
    const float* P = pose + (size_t)b * 16;
    for (int i = 0; i < HW; ++i) {
      const float d = depth[(size_t)b * HW + i] * unit;
      if (!(d > lo && d < hi)) continue;
      const int r = i / W, col = i - r * W;
      const float x = ((float)col - c.cx) * d / c.fx, y = ((float)r - c.cy) * d / c.fy;
      float ox, oy, oz;
      se3_apply(P, x, y, d, ox, oy, oz);
      splat(out + (size_t)b * HW, c, ox, oy, oz, H, W);
    }
  }
  resolve(out, mask, (size_t)B * HW, out_scale, out_scale != 1.0f);
  return PRG_OK;
}

int prg_cpu_unproject_f64(const float* depth, const float* K, const float* pose, double* xyz, uint8_t* valid, int B, int H,
                          int W, float unit, float lo, float hi) {
  CPU_CHECK(depth && K && xyz && valid && B > 0 && H > 0 && W > 0, "prg_cpu_unproject_f64: bad arguments");
  const int HW = H * W;
  const double nan = std::numeric_limits<double>::quiet_NaN();
  for (int b = 0; b < B; ++b) {
    const Cam c = load_cam(K, b);
    const float* P = pose ? pose + (size_t)b * 16 : nullptr;
    for (int i = 0; i < HW; ++i) {
      const float d = depth[(size_t)b * HW + i] * unit;
      const bool ok = d > lo && d < hi;
      double X = nan, Y = nan, Z = nan;
      if (ok) {
        const int r = i / W, col = i - r * W;
        const double z = (double)d;
        const double x = ((double)col - (double)c.cx) * z / (double)c.fx;
        const double y = ((double)r - (double)c.cy) * z / (double)c.fy;
        if (P) {
          const double px = x - (double)P[3], py = y - (double)P[7], pz = z - (double)P[11];
          X = std::fma(pz, (double)P[8], std::fma(py, (double)P[4], px * (double)P[0]));
          Y = std::fma(pz, (double)P[9], std::fma(py, (double)P[5], px * (double)P[1]));
          Z = std::fma(pz, (double)P[10], std::fma(py, (double)P[6], px * (double)P[2]));
        } else {
          X = x; Y = y; Z = z;
        }
      }
      double* o = xyz + ((size_t)b * HW + i) * 3;
      o[0] = X; o[1] = Y; o[2] = Z;
      valid[(size_t)b * HW + i] = ok ? 1 : 0;
    }
  }
  return PRG_OK;
}

int prg_cpu_depth_augment(const float* depth, float* out, int B, int H, int W) {
  CPU_CHECK(depth && out && B > 0 && H > 0 && W > 0, "prg_cpu_depth_augment: bad arguments");
  const int HW = H * W;
  for (int b = 0; b < B; ++b) {
    const float* d = depth + (size_t)b * HW;
    float* o = out + (size_t)b * 3 * HW;
    for (int i = 0; i < HW; ++i) {
      const int r = i / W, col = i - r * W;
      float mv = kInf, mr = kInf;
      for (int dy = -1; dy <= 1; ++dy) {
        const int rr = r + dy;
        if (rr < 0 || rr >= H) continue;
        for (int dx = -1; dx <= 1; ++dx) {
          const int cc = col + dx;
          if (cc < 0 || cc >= W) continue;
          const float v = d[rr * W + cc];
          mr = std::fmin(mr, v);
          if (v != 0.0f) mv = std::fmin(mv, v);
        }
      }
      const float m = (mv == kInf) ? mr : mv;
      o[i] = d[i];
      o[HW + i] = m;
      o[2 * HW + i] = m - d[i];
    }
  }
  return PRG_OK;
}

int prg_cpu_apply_mask(const float* prob, const float* depth, const uint8_t* hit, float thr, float* depth_out, uint8_t* hit_out,
                       float* cond, int B, int H, int W) {
  CPU_CHECK(prob && depth && B > 0 && H > 0 && W > 0, "prg_cpu_apply_mask: bad arguments");
  const int HW = H * W;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < HW; ++i) {
      const size_t g = (size_t)b * HW + i;
      const bool keep = prob[g] > thr;
      const float d = keep ? depth[g] : 0.0f;
      const bool m = keep && (hit ? hit[g] != 0 : true);
      if (depth_out) depth_out[g] = d;
      if (hit_out) hit_out[g] = m ? 1 : 0;
      if (cond) {
        cond[(size_t)b * 2 * HW + i] = d * 2.0f - 1.0f;
        cond[(size_t)b * 2 * HW + HW + i] = (m ? 1.0f : 0.0f) * 2.0f - 1.0f;
      }
    }
  return PRG_OK;
}

int prg_cpu_unet_create(const prg_unet_config* cfg, const float* weights, int64_t n, prg_cpu_unet** out) {
  CPU_CHECK(cfg && weights && out, "prg_cpu_unet_create: null pointer");
  prg_cpu_unet* u = new prg_cpu_unet();
  int rc = build(*u, *cfg);
  if (rc == PRG_OK && u->total != n)
    rc = fail(PRG_E_INVALID, "prg_cpu_unet_create: expected " + std::to_string(u->total) + " floats, got " + std::to_string(n));
  if (rc) { delete u; return rc; }
  u->flat.assign(weights, weights + n);
  auto ws = [&](const ConvP& p) {       // Block.proj: standardised once (the reference does it every forward, sd:601-616)
    Vec s = standardize(u->flat.data() + p.w, p.Cout, p.Cin * p.K * p.K);
    std::memcpy(u->flat.data() + p.w, s.data(), sizeof(float) * s.size());
  };
  auto res = [&](const ResP& r) { ws(r.c1); ws(r.c2); };
  for (auto& lv : u->downs) { res(lv.r0); res(lv.r1); }
  for (auto& lv : u->ups) { res(lv.r0); res(lv.r1); }
  res(u->mid1); res(u->mid2); res(u->fin);
  const int half = cfg->dim / 2;
  u->freqs.resize(half);
  const float stepf = -(float)(9.210340371976184 / (double)(half - 1));
  for (int i = 0; i < half; ++i) u->freqs[i] = std::exp((float)i * stepf);
  *out = u;
  return PRG_OK;
}

int prg_cpu_unet_destroy(prg_cpu_unet* h) {
  delete h;
  return PRG_OK;
}

int prg_cpu_unet_set_time_freqs(prg_cpu_unet* h, const float* freqs, int n) {
  CPU_CHECK(h && freqs && n == h->cfg.dim / 2, "prg_cpu_unet_set_time_freqs: need dim/2 frequencies");
  h->freqs.assign(freqs, freqs + n);
  return PRG_OK;
}

int prg_cpu_unet_forward(prg_cpu_unet* h, const float* x, const int64_t* time, const float* pc, float* out, int B, int S) {
  CPU_CHECK(h && x && time && pc && out && B > 0 && S > 0, "prg_cpu_unet_forward: bad arguments");
  CPU_CHECK(h->cfg.conditional && h->cfg.in_channels == 1, "prg_cpu_unet_forward: handle is not a conditional U-Net");
  Vec cond;
  h->conditioning(time, pc, B, cond);
  return h->forward(x, cond.data(), out, B, S);
}

int prg_cpu_maskunet_forward(prg_cpu_unet* h, const float* depth, float* prob, int B, int S) {
  CPU_CHECK(h && depth && prob && B > 0 && S > 0, "prg_cpu_maskunet_forward: bad arguments");
  CPU_CHECK(!h->cfg.conditional && h->cfg.in_channels == 3, "prg_cpu_maskunet_forward: handle is not a MaskUnet");
  Vec aug((size_t)B * 3 * S * S);
  int rc = prg_cpu_depth_augment(depth, aug.data(), B, S, S);
  if (rc) return rc;
  return h->forward(aug.data(), nullptr, prob, B, S);
}

int prg_cpu_sampler_run(prg_cpu_unet* h, const prg_step* steps, int n_steps, const float* pc, const float* cond,
                        const float* noise, int64_t noise_slabs, const uint64_t* seeds, float* out, int B, int S) {
  CPU_CHECK(h && steps && pc && out && n_steps > 0 && B > 0 && S > 0, "prg_cpu_sampler_run: bad arguments");
  CPU_CHECK(noise || seeds, "prg_cpu_sampler_run: need stored noise or per-scene seeds");
  CPU_CHECK((S * S) % 4 == 0, "prg_cpu_sampler_run: H*W must be a multiple of 4");
  const int HW = S * S;
  if (noise) {
    int64_t need = 1;
    for (int k = 0; k < n_steps; ++k)
      if (steps[k].sigma != 0.0f) need = k + 2;
    CPU_CHECK(noise_slabs >= need, "prg_cpu_sampler_run: stored noise has too few slabs for this transition table");
  }
  auto draw = [&](int k, int b, int q, float (&o)[4]) {        // slab k (0 = start image), pixel quad q of image b
    if (noise) std::memcpy(o, noise + ((size_t)k * B + b) * HW + (size_t)q * 4, sizeof(float) * 4);
    else philox_normal4(seeds[b], (uint32_t)k, (uint32_t)q, o);
  };
  Vec x((size_t)B * HW), u((size_t)B * HW);
  for (int b = 0; b < B; ++b)
    for (int q = 0; q * 4 < HW; ++q) {
      float o[4];
      draw(0, b, q, o);
      std::memcpy(x.data() + (size_t)b * HW + (size_t)q * 4, o, sizeof(o));
    }
  std::vector<int64_t> tt(B);
  for (int k = 0; k < n_steps; ++k) {
    const prg_step st = steps[k];
    for (int b = 0; b < B; ++b) tt[b] = st.t;
    int rc = prg_cpu_unet_forward(h, x.data(), tt.data(), pc, u.data(), B, S);
    if (rc) return rc;
    for (int b = 0; b < B; ++b)
      for (int q = 0; q * 4 < HW; ++q) {
        float nz[4] = {0, 0, 0, 0};
        if (st.sigma != 0.0f) draw(k + 1, b, q, nz);
        for (int e = 0; e < 4; ++e) {
          const size_t o = (size_t)b * HW + (size_t)q * 4 + e;
          const float xs = x[o], us = u[o];
          const float cd = cond ? cond[(size_t)b * 2 * HW + (size_t)q * 4 + e] : 0.0f;
          const float cm = cond ? cond[(size_t)b * 2 * HW + HW + (size_t)q * 4 + e] : -1.0f;
          // arithmetic of sampler_step_kernel (sampler.hip) = the reference's order (sd:1158-1218, 1250-1280, 1369-1373)
          const float x0p = (st.clip_pred & 1) ? clamp1(us) : us;
          const bool known = cond && ((cm + 1.0f) * 0.5f > 0.5f);
          float x0 = known ? cd : x0p;
          if (st.clip_pred & 2) x0 = clamp1(x0);
          float v = st.c_x0 * x0;
          if (st.clip_pred & 4) {
            x[o] = known ? clamp1(us) : xs;
            continue;
          }
          if (st.c_x != 0.0f) v = v + st.c_x * xs;
          if (st.c_eps != 0.0f) {
            const float eps = (st.sqrt_recip * xs - x0p) / st.sqrt_recipm1;
            v = v + st.c_eps * eps;
          }
          if (st.sigma != 0.0f) v = v + st.sigma * nz[e];
          x[o] = v;
        }
      }
  }
  for (size_t i = 0; i < x.size(); ++i) out[i] = (x[i] + 1.0f) * 0.5f;
  return PRG_OK;
}

}  // extern "C"
