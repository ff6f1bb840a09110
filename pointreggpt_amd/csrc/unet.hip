// unet.hip — host side of the two U-Nets and of the sampler: weight arena, layer walk, workspace, hipGraph.
//
// Mirrors (structure only) Unet.forward sd:920-964, MaskUnet.forward dc:871-906, GaussianDiffusion.sample
// sd:1283-1409.  Activations live as NHWC tensors of T (bf16_t or float) in a stack arena owned by the handle;
// the skip `torch.cat` never materialises (two source pointers into the conv), `nn.Upsample` is folded into the
// following conv's gather, weight standardisation is folded into the packed weights at load time.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include "blocks.h"
#include "conv.h"
#include "sampler.h"

namespace prg {

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
int device_cu_count() {
  static std::atomic<int> cus[64];          // zero-initialised; per device ordinal
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::atomic<int>& c = cus[dev & 63];
  int n = c.load(std::memory_order_acquire);
  if (n <= 0) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    n = p.multiProcessorCount;
    c.store(n, std::memory_order_release);
  }
  return n;
}
int fail(int code, const std::string& m) {
  g_err = m;
  return code;
}
const char* last_error() { return g_err.c_str(); }

// PRG_GN_FOLD=1: the conv kernels that can (conv_c64.hip) fold the GroupNorm statistics into coefficients themselves (last
// workgroup of an image, per-image ticket) instead of a gn_coeff_kernel launch.  Measured net-neutral (every workgroup of a
// persistent launch finishes at the same time, so the fold lands on the kernel's tail: +5-6 us per conv against the 5 us
// launch + 1.7 us boundary it removes): off by default, kept as a tested option.
static bool head_fuse_enabled() {
  static const int on = [] { const char* e = std::getenv("PRG_HEAD_FUSE"); return e ? std::atoi(e) : 1; }();
  return on != 0;
}
static bool res_epilogue_enabled() {
  static const int on = [] { const char* e = std::getenv("PRG_RES_EPILOGUE"); return e ? std::atoi(e) : 1; }();
  return on != 0;
}
// PRG_GN_ACC=0: GroupNorm statistics as per-tile slabs + one gn_coeff_kernel launch per norm (rounds 1-2) instead of the
// fixed-point accumulators folded by the consumers (common.h, GnFold).  bf16 / mxfp8 handles only; fp32 always uses slabs.
static bool gn_acc_enabled() {
  static const int on = [] { const char* e = std::getenv("PRG_GN_ACC"); return e ? std::atoi(e) : 1; }();
  return on != 0;
}
static bool gn_fold_enabled() {
  static const int on = [] { const char* e = std::getenv("PRG_GN_FOLD"); return e ? std::atoi(e) : 0; }();
  return on != 0;
}

// ---------------------------------------------------------------------------------------------
// device stack arena
// ---------------------------------------------------------------------------------------------
struct Arena {
  char* base = nullptr;
  size_t cap = 0, top = 0, high = 0;
  bool dry = false;  // dry run: only measure
  void* alloc(size_t bytes) {
    size_t a = (top + 255) & ~(size_t)255;
    top = a + bytes;
    if (top > high) high = top;
    if (dry) return reinterpret_cast<void*>((uintptr_t)0x1000 + a);  // never dereferenced
    return (top <= cap) ? base + a : nullptr;
  }
  size_t mark() const { return top; }
  void reset(size_t m) { top = m; }
};

struct ProfileSink {  // per-launch conv timing (bench roofline)
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  size_t used = 0;
  double conv_ms = 0, conv_flops = 0, conv_bytes = 0;   // conv_bytes: algorithmic input + output + weight bytes
  double conv_flops_exec = 0;                           // 2 * MACs the kernels executed (sub-pixel Upsample convs: 4 / 9 of the algorithmic count)
  int64_t launches = 0;
  // per-SHAPE table (round 5, bench.py roofline.per_kernel): what pool[i] timed, and the totals per distinct conv shape
  struct Rec { prg_profile_shape key; double flops, flops_exec; };
  std::vector<Rec> recs;                                       // recs[i] <-> pool[i] of the step being harvested
  std::vector<prg_profile_shape> shapes;                       // accumulated (launches, ms, flops) per distinct key
  void add_shape(const Rec& r, double ms) {
    for (auto& sh : shapes)
      if (sh.cin == r.key.cin && sh.cout == r.key.cout && sh.k == r.key.k && sh.stride == r.key.stride && sh.ups == r.key.ups &&
          sh.hout == r.key.hout && sh.wout == r.key.wout && sh.two_source == r.key.two_source && sh.prologue == r.key.prologue &&
          sh.mx == r.key.mx) {
        sh.launches += 1; sh.ms += ms; sh.flops += r.flops; sh.flops_executed += r.flops_exec;
        return;
      }
    prg_profile_shape sh = r.key;
    sh.launches = 1; sh.ms = ms; sh.flops = r.flops; sh.flops_executed = r.flops_exec;
    shapes.push_back(sh);
  }
  std::vector<std::pair<hipEvent_t, hipEvent_t>> step_pool;   // the per-transition update kernel (HBM-bound)
  size_t step_used = 0;
  double step_ms = 0;
  int64_t step_launches = 0;
};

// ---------------------------------------------------------------------------------------------
// parameter walk: the flat float32 array is the reference state_dict in order (weights.param_spec)
// ---------------------------------------------------------------------------------------------
struct ConvP {
  int Cout = 0, Cin = 0, KH = 0, KW = 0, CoutPad = 0, kchunks = 0;
  size_t w_off = 0;        // element offset into the packed-T arena
  int64_t b_off = -1;      // float offset of the bias in the flat array (-1: none)
  int64_t w_flat = -1;     // float offset of the raw OIHW weight in the flat array
  bool ws = false;         // weight-standardised (Block.proj)
  int64_t mx_off = -1;     // byte offset of the MX-fp8 copy of the weights (-1: none) and of its block scales
  int64_t mx_soff = -1;
  int64_t s2d_off = -1;    // 4x4 / stride 2 convs (bf16): element offset of the equivalent 2x2-tap packing (ConvLaunch::w_s2d)
  int s2d_kchunks = 0;
  int64_t sp_off = -1;     // f16x3 mode: element offset of the hi / lo f16 split packing (ConvLaunch::w_split)
  int sp_kchunks = 0;
  int64_t h16_off = -1;    // bf16 mode, second conv of a ResnetBlock: element offset of the f16 twin of the packing (ConvLaunch::w_f16)
  int64_t up_off = -1;     // bf16 mode, Upsample convs: element offset of the four pre-summed 2 x 2-tap packings (ConvLaunch::w_up)
  int64_t sp_scale_off = -1, up_sp_scale_off = -1;   // f16x3 mode: offsets into d_split_scale (ConvLaunch::split_scale / split_scale_up)
  int64_t up_sp_off = -1;  // f16x3 mode: the same in the split layout (ConvLaunch::w_up_split)
};
struct ResP {
  int cin = 0, cout = 0;
  int64_t mlp_w = -1, mlp_b = -1;  // Linear(2*emb -> 2*cout)
  int ss_off = 0;                  // column offset of this block's (scale|shift) in the conditioning row
  ConvP c1, c2, res;
  int64_t g1 = 0, b1 = 0, g2 = 0, b2 = 0;
  int64_t fw_res = -1;             // bf16 element offset of the raw res_conv weight in the fused-kernel arena (-1: unfused)
  bool has_res = false;
  int64_t pq1 = -1, pq2 = -1;      // float offsets into d_pq_static of (P = gamma | Q = beta) of norm 1 / 2, each cpad floats
  int cpad = 0;                    // cout rounded up to 4 floats (16-byte aligned rows)
};
struct AttnP {
  int C = 0;
  bool linear = true;
  ConvP qkv, out;
  int64_t out_g = -1, norm_g = -1;
  int64_t fw_qkv = -1, fw_out = -1;   // bf16 element offsets into the fused-attention weight arena (-1: unfused path)
  int64_t kshift = -1;                // float offset of the 128 static softmax shifts in d_kshift (-1: measure the maxima)
  int64_t sp_qkv = -1, sp_out = -1;   // f16x3 mode: element offsets into d_attn_split of the hi halves (the lo halves follow)
};
struct LevelP {
  ResP r0, r1;
  AttnP at;
  ConvP resample;
  bool strided = false;  // down: 4x4 s2 ; up: nearest x2 + 3x3
};

struct Cursor {
  int64_t pos = 0;
  int64_t take(int64_t n) {
    int64_t p = pos;
    pos += n;
    return p;
  }
};

static void walk_conv(Cursor& c, ConvP& p, int Cout, int Cin, int K, bool bias, bool ws) {
  p.Cout = Cout; p.Cin = Cin; p.KH = K; p.KW = K; p.ws = ws;
  p.w_flat = c.take((int64_t)Cout * Cin * K * K);
  p.b_off = bias ? c.take(Cout) : -1;
}
static void walk_res(Cursor& c, ResP& r, int cin, int cout, bool cond, int emb, int& ss_total) {
  r.cin = cin; r.cout = cout;
  if (cond) {
    r.mlp_w = c.take((int64_t)2 * cout * 2 * emb);
    r.mlp_b = c.take(2 * cout);
    r.ss_off = ss_total;
    ss_total += 2 * cout;
  }
  walk_conv(c, r.c1, cout, cin, 3, true, true);
  r.g1 = c.take(cout); r.b1 = c.take(cout);
  walk_conv(c, r.c2, cout, cout, 3, true, true);
  r.g2 = c.take(cout); r.b2 = c.take(cout);
  r.has_res = cin != cout;
  if (r.has_res) walk_conv(c, r.res, cout, cin, 1, true, false);
}
static void walk_attn(Cursor& c, AttnP& a, int C, bool linear) {
  a.C = C; a.linear = linear;
  walk_conv(c, a.qkv, 3 * kHidden, C, 1, false, false);
  walk_conv(c, a.out, C, kHidden, 1, true, false);
  if (linear) a.out_g = c.take(C);
  a.norm_g = c.take(C);
}

struct Layout {
  prg_unet_config cfg;
  int emb = 0, ss_total = 0, L = 0;
  std::vector<int> dims;
  int64_t stem_w = 0, stem_b = 0;
  int64_t tm1_w = 0, tm1_b = 0, tm3_w = 0, tm3_b = 0, pm0_w = 0, pm0_b = 0, pm2_w = 0, pm2_b = 0;
  std::vector<LevelP> downs, ups;
  ResP mid1, mid2, fin;
  AttnP mid_at;
  int64_t head_w = 0, head_b = 0;
  int64_t total = 0;
};

static int build_layout(const prg_unet_config& cfg, Layout& L) {
  PRG_CHECK(cfg.dim >= 8 && cfg.dim % 8 == 0, "config: dim must be a multiple of 8");
  PRG_CHECK(cfg.n_levels >= 1 && cfg.n_levels <= 8, "config: n_levels out of range");
  PRG_CHECK(cfg.in_channels == 1 || cfg.in_channels == 3, "config: in_channels must be 1 or 3");
  PRG_CHECK(cfg.groups >= 1 && cfg.groups <= 64, "config: groups out of range");
  L.cfg = cfg;
  L.L = cfg.n_levels;
  L.emb = cfg.dim * 4;
  L.dims.assign(1, cfg.dim);
  for (int i = 0; i < cfg.n_levels; ++i) {
    PRG_CHECK(cfg.dim_mults[i] >= 1, "config: bad dim_mult");
    L.dims.push_back(cfg.dim * cfg.dim_mults[i]);
  }
  for (size_t i = 0; i < L.dims.size(); ++i) PRG_CHECK(L.dims[i] % cfg.groups == 0, "config: width not divisible by groups");
  for (size_t i = 0; i < L.dims.size(); ++i)
    if (L.dims[i] > 1024)
      return fail(PRG_E_INVALID, "config: dim * dim_mult = " + std::to_string(L.dims[i]) + " exceeds 1024 channels, the widest "
                  "GroupNorm the coefficient kernels handle (gn_coeff_kernel / affine_silu_fold_kernel: four channels per thread)");
  const bool cond = cfg.conditional != 0;
  Cursor c;
  const int d0 = cfg.dim, e = L.emb;
  L.stem_w = c.take((int64_t)d0 * cfg.in_channels * 49);
  L.stem_b = c.take(d0);
  if (cond) {
    L.tm1_w = c.take((int64_t)e * d0); L.tm1_b = c.take(e);
    L.tm3_w = c.take((int64_t)e * e);  L.tm3_b = c.take(e);
    L.pm0_w = c.take((int64_t)e * cfg.param_cond_dim); L.pm0_b = c.take(e);
    L.pm2_w = c.take((int64_t)e * e);  L.pm2_b = c.take(e);
  }
  L.downs.resize(L.L);
  L.ups.resize(L.L);
  for (int i = 0; i < L.L; ++i) {
    const int ci = L.dims[i], co = L.dims[i + 1];
    LevelP& lv = L.downs[i];
    walk_res(c, lv.r0, ci, ci, cond, e, L.ss_total);
    walk_res(c, lv.r1, ci, ci, cond, e, L.ss_total);
    walk_attn(c, lv.at, ci, true);
    lv.strided = i != L.L - 1;
    walk_conv(c, lv.resample, co, ci, lv.strided ? 4 : 3, true, false);
  }
  for (int i = 0; i < L.L; ++i) {
    const int ci = L.dims[L.L - 1 - i], co = L.dims[L.L - i];
    LevelP& lv = L.ups[i];
    walk_res(c, lv.r0, co + ci, co, cond, e, L.ss_total);
    walk_res(c, lv.r1, co + ci, co, cond, e, L.ss_total);
    walk_attn(c, lv.at, co, true);
    lv.strided = i != L.L - 1;  // here: "followed by x2 upsample"
    walk_conv(c, lv.resample, ci, co, 3, true, false);
  }
  const int mid = L.dims.back();
  walk_res(c, L.mid1, mid, mid, cond, e, L.ss_total);
  walk_attn(c, L.mid_at, mid, false);
  walk_res(c, L.mid2, mid, mid, cond, e, L.ss_total);
  walk_res(c, L.fin, 2 * d0, d0, cond, e, L.ss_total);
  L.head_w = c.take(d0);
  L.head_b = c.take(1);
  L.total = c.pos;
  return PRG_OK;
}

// conditioning source for the ResnetBlocks of one forward
struct CondSrc {
  const float* ss_a = nullptr;
  const float* ss_b = nullptr;
  int64_t ss_a_stride = 0, ss_b_stride = 0;
  const int* row = nullptr;
  int64_t row_stride = 0;
};

struct Tap {
  const void* ptr;
  int C, H, W, B;
  bool nchw_f32;
};

}  // namespace prg

using namespace prg;

// ---------------------------------------------------------------------------------------------
// handle types
// ---------------------------------------------------------------------------------------------
struct prg_unet {
  Layout lay;
  int dtype = PRG_F32;
  float* d_flat = nullptr;      // the float32 state_dict on device (biases, norm gains, MLPs read in place)
  void* d_packed = nullptr;     // packed conv weights of T
  float* d_stem = nullptr;      // stem weights [49*Cin][dim]
  bf16_t* d_stem_frag = nullptr; // stem weights as MFMA fragments (bf16 path, Cin 1 -> 64)
  uint16_t* d_stem_split = nullptr;    // f16x3 mode: the same fragments as f16 hi / lo halves (stem_mfma_kernel<CIN, true>)
  float* d_stem_split_scale = nullptr; // ... and the inverse of the packer's per-channel power-of-two scale [64]
  bf16_t* d_attn = nullptr;     // fused linear attention: gain-folded to_qkv and to_out weights (bf16 path only)
  float* d_kshift = nullptr;    // fused linear attention: static softmax shifts of the k columns
  uint8_t* d_mx = nullptr;      // MX-fp8 conv weights (dtype PRG_MXFP8): e4m3 data and E8M0 block scales
  uint8_t* d_mx_scale = nullptr;
  uint16_t* d_attn_split = nullptr;   // f16x3 mode: fused linear attention weights as f16 hi / lo halves (attn_split.hip)
  float* d_split_scale = nullptr;   // f16x3 mode: the packer's per-output-channel power-of-two factors, undone in the epilogues
  uint16_t* d_split = nullptr;  // f16x3 mode (dtype PRG_F16X3): every conv weight as f16 hi / lo halves (conv_split.hip)
  uint16_t* d_h16 = nullptr;    // bf16 mode: f16 twins of the ResnetBlocks' second convs (the h16 format, conv.h)
  float* d_freqs = nullptr;     // SinusoidalPosEmb frequencies [dim/2] (sd:645-657), see prg_unet_set_time_freqs
  int* d_tickets = nullptr;     // [kMaxTicketImages] per-image arrival counters of the conv kernels that fold GroupNorm coefficients (self-resetting)
  // fixed-point GroupNorm statistics (common.h, GnFold; bf16 / mxfp8 handles)
  long long* d_gnacc = nullptr; // [slots][resB][groups][2], zeroed by ONE memset at the start of every forward
  size_t gnacc_bytes = 0;
  int gn_slots = 0, gn_slot = 0;
  float* d_pq_static = nullptr; // (gamma | beta) of every norm, 16-byte aligned rows: P / Q of the unconditioned norms
  CondFoldEntry* d_cond_entries = nullptr;   // one entry per conditioned norm (Block 1 of every ResnetBlock)
  int n_cond_entries = 0;
  Arena arena;
  uint64_t arena_gen = 0;       // bumped whenever the workspace is reallocated: captured graphs bake its pointers in
  int resB = 0, resS = 0;
  bool taps_on = false;
  std::map<std::string, Tap> taps;
  ProfileSink* prof = nullptr;
  virtual ~prg_unet() {}
  virtual int measure(int B, int S, size_t* bytes) = 0;
  virtual int forward(const float* x_nchw, const CondSrc& cond, float* out, int B, int S, hipStream_t s) = 0;
  virtual int cond_general(const int64_t* time, const float* param_cond, int B, CondSrc* out, hipStream_t s) = 0;
  virtual int tap_copy(const Tap& t, float* out, hipStream_t s) = 0;
};

namespace prg {

template <typename T>
struct UnetImpl : prg_unet {
  // -------- small helpers --------
  const float* F(int64_t off) const { return off < 0 ? nullptr : d_flat + off; }
  const T* W(const ConvP& p) const { return reinterpret_cast<const T*>(d_packed) + p.w_off; }
  template <typename U>
  U* alloc(size_t n) { return reinterpret_cast<U*>(arena.alloc(n * sizeof(U))); }

  // set around the final block only: its fused tail also applies the head (launch_resblock_tail_fused)
  const float* head_w = nullptr;
  const float* head_b = nullptr;
  float* head_out = nullptr;
  int head_sigmoid = 0;
  bool head_done = false;

  struct ConvOpt {           // optional fusions of one conv launch
    const T* residual = nullptr;
    const float* res_a = nullptr;   // activated residual: + SiLU(residual * res_a + res_b) (ConvLaunch::res_a)
    const float* res_b = nullptr;
    float* gn_partials = nullptr;   // fused GroupNorm statistics of the output
    int* gn_nsplit = nullptr;       // out: slabs written (0 = not fused for this shape)
    const float* pro_a = nullptr;   // fused GroupNorm+cond+SiLU on the input (halo kernel)
    const float* pro_b = nullptr;
    const GnApply* gn = nullptr;    // fold the output's statistics into coefficients inside the conv kernel when it can
    float* coef_a = nullptr;
    float* coef_b = nullptr;
    int* coef_done = nullptr;       // out: 1 = coef_a / coef_b are written by the conv launch (no gn_coeff launch needed)
    long long* gn_acc = nullptr;    // fixed-point statistics of the output (instead of gn_partials when the kernel can)
    int* acc_done = nullptr;        // out: 1 = the launch accumulated into gn_acc
    const GnFold* fold = nullptr;   // prologue coefficients folded by the consumer (pro_a / pro_b = scratch tables)
    const GnFold* res_fold = nullptr;   // activated residual folded in the 1x1 conv's epilogue (instead of res_a / res_b)
    bool out_f16 = false;           // h16 (conv.h): the output is stored as f16 / the input is f16 and the conv takes its f16 weights
    bool in_f16 = false;
  };

  ConvDesc make_desc(const ConvP& p, int C0, int C1, int B, int Hin, int Win, int stride, int pad, int ups) const {
    ConvDesc d;
    d.B = B; d.Hin = Hin; d.Win = Win; d.C0 = C0; d.C1 = C1; d.ups = ups;
    d.KH = p.KH; d.KW = p.KW; d.stride = stride; d.pad = pad;
    const int Hl = ups ? 2 * Hin : Hin, Wl = ups ? 2 * Win : Win;
    d.Hout = (Hl + 2 * pad - p.KH) / stride + 1;
    d.Wout = (Wl + 2 * pad - p.KW) / stride + 1;
    d.Cout = p.Cout; d.CoutPad = p.CoutPad; d.kchunks = p.kchunks;
    return d;
  }

  ConvLaunch<T> make_launch(const ConvP& p, const T* s0, int C0, const T* s1, int C1, int B, int Hin, int Win, int stride, int pad,
                            int ups, const ConvOpt& o, T* out) const {
    ConvLaunch<T> L;
    L.d = make_desc(p, C0, C1, B, Hin, Win, stride, pad, ups);
    L.src0 = s0; L.src1 = s1; L.w = W(p); L.bias = F(p.b_off); L.residual = o.residual; L.out = out;
    L.res_a = o.res_a; L.res_b = o.res_b;
    L.gn_partials = o.gn_partials; L.gn_groups = lay.cfg.groups; L.pro_a = o.pro_a; L.pro_b = o.pro_b;
    L.w_mx = (d_mx && p.mx_off >= 0) ? d_mx + p.mx_off : nullptr;
    L.w_mx_scale = (d_mx_scale && p.mx_soff >= 0) ? d_mx_scale + p.mx_soff : nullptr;
    L.mx_pure = 0;
    L.w_s2d = (p.s2d_off >= 0 && stride == 2 && pad == 1) ? reinterpret_cast<const T*>(d_packed) + p.s2d_off : nullptr;
    L.w_up = (p.up_off >= 0 && ups && stride == 1 && pad == 1) ? reinterpret_cast<const T*>(d_packed) + p.up_off : nullptr;
    L.s2d_kchunks = p.s2d_kchunks;
    L.w_split = (d_split && p.sp_off >= 0) ? d_split + p.sp_off : nullptr;
    L.w_up_split = (d_split && p.up_sp_off >= 0 && ups && stride == 1 && pad == 1) ? d_split + p.up_sp_off : nullptr;
    L.split_kchunks = p.sp_kchunks;
    L.split_scale = (L.w_split && d_split_scale && p.sp_scale_off >= 0) ? d_split_scale + p.sp_scale_off : nullptr;
    L.split_scale_up = (L.w_up_split && d_split_scale && p.up_sp_scale_off >= 0) ? d_split_scale + p.up_sp_scale_off : nullptr;
    L.gn = o.gn ? *o.gn : GnApply{};
    L.gn_coef_a = o.gn ? o.coef_a : nullptr; L.gn_coef_b = o.gn ? o.coef_b : nullptr;
    L.gn_tickets = (o.gn && gn_fold_enabled()) ? d_tickets : nullptr;
    L.gn_acc = o.gn_acc;
    L.pro_fold = o.fold ? *o.fold : GnFold{};
    L.res_fold = o.res_fold ? *o.res_fold : GnFold{};
    L.out_f16 = o.out_f16 ? 1 : 0;
    L.in_f16 = o.in_f16 ? 1 : 0;
    L.w_f16 = (d_h16 && p.h16_off >= 0) ? d_h16 + p.h16_off : nullptr;
    return L;
  }

  int conv(const ConvP& p, const T* s0, int C0, const T* s1, int C1, int B, int Hin, int Win, int stride, int pad,
           int ups, const ConvOpt& o, T* out, hipStream_t s) {
    const ConvLaunch<T> L = make_launch(p, s0, C0, s1, C1, B, Hin, Win, stride, pad, ups, o, out);
    PRG_CHECK(C0 + C1 == p.Cin, "conv: channel mismatch");
    if (o.gn_nsplit) *o.gn_nsplit = 0;
    if (o.coef_done) *o.coef_done = 0;
    if (o.acc_done) *o.acc_done = 0;
    if (arena.dry) return PRG_OK;
    if (prof && prof->on) {
      if (prof->used == prof->pool.size()) {
        hipEvent_t a, b;
        PRG_HIP(hipEventCreate(&a));
        PRG_HIP(hipEventCreate(&b));
        prof->pool.push_back({a, b});
      }
      auto& ev = prof->pool[prof->used++];
      PRG_HIP(hipEventRecord(ev.first, s));
      int rc = launch_conv<T>(L, s, o.gn_nsplit, o.coef_done, o.acc_done);
      PRG_HIP(hipEventRecord(ev.second, s));
      prof->conv_flops += conv_flops(L.d);
      prof->conv_flops_exec += conv_flops(L.d) * conv_last_exec_scale();
      {
        ProfileSink::Rec r{};
        r.key.cin = L.d.C0 + L.d.C1; r.key.cout = L.d.Cout; r.key.k = L.d.KH; r.key.stride = L.d.stride; r.key.ups = L.d.ups;
        r.key.hout = L.d.Hout; r.key.wout = L.d.Wout; r.key.two_source = L.d.C1 > 0; r.key.prologue = (L.pro_a != nullptr || L.pro_fold.acc != nullptr);
        r.key.mx = conv_last_was_mx();
        r.flops = conv_flops(L.d); r.flops_exec = conv_flops(L.d) * conv_last_exec_scale();
        if (prof->recs.size() < prof->used) prof->recs.resize(prof->used);
        prof->recs[prof->used - 1] = r;
      }
      prof->conv_bytes += ((double)L.d.B * L.d.Hin * L.d.Win * (L.d.C0 + L.d.C1) + (double)L.d.B * L.d.Hout * L.d.Wout * L.d.Cout +
                           (double)L.d.Cout * (L.d.C0 + L.d.C1) * L.d.KH * L.d.KW) * sizeof(T);
      prof->launches += 1;
      return rc;
    }
    return launch_conv<T>(L, s, o.gn_nsplit, o.coef_done, o.acc_done);
  }

  // ---- fixed-point GroupNorm statistics (bf16 path) ----
  const float* pq_dyn = nullptr;     // [B][ss_total] P | Q of the conditioned norms of THIS forward (cond_fold_kernel)
  bool acc_on() const { return std::is_same<T, bf16_t>::value && d_gnacc && gn_acc_enabled(); }
  long long* next_acc(int B) {
    if (!acc_on() || gn_slot >= gn_slots) return nullptr;
    return d_gnacc + (size_t)(gn_slot++) * resB * lay.cfg.groups * 2;
  }
  GnFold make_fold(const long long* acc, int C, int HW, bool conditioned, int ss_off, int64_t pq_static) const {
    GnFold f{};
    f.acc = acc;
    f.G = lay.cfg.groups;
    f.cpg = C / f.G;
    f.inv_n = 1.0f / ((float)HW * (float)f.cpg);
    if (conditioned && pq_dyn) {
      f.P = pq_dyn + ss_off; f.Q = f.P + C; f.pq_stride = lay.ss_total;
    } else {
      f.P = d_pq_static + pq_static; f.Q = f.P + ((C + 3) & ~3); f.pq_stride = 0;
    }
    return f;
  }

  GnApply gn_params(int64_t g_off, int64_t b_off, const CondSrc* cs, int ss_off) const {
    GnApply p{};
    p.gamma = F(g_off); p.beta = F(b_off);
    if (cs && cs->ss_a) {
      p.ss_a = cs->ss_a + ss_off;
      p.ss_a_stride = cs->ss_a_stride;
      p.ss_b = cs->ss_b ? cs->ss_b + ss_off : nullptr;
      p.ss_b_stride = cs->ss_b_stride;
      p.ss_a_row = cs->row;
      p.ss_a_row_stride = cs->row_stride;
    }
    return p;
  }

  // ResnetBlock (sd:700-734 / dc:726-740): out <- block2(block1(cat[s0,s1])) + res(cat[s0,s1]).
  //   conv1 (+ fused GN statistics) -> [GN + cond + SiLU folded into conv2's halo load] -> conv2 (+ statistics)
  //   -> GN + SiLU + skip in one elementwise pass.
  int resblock(const ResP& r, const T* s0, int C0, const T* s1, int C1, const CondSrc* cs, T* out, int B, int H, int Wd,
               hipStream_t s) {
    const size_t m = arena.mark();
    const size_t M = (size_t)B * H * Wd;
    const int G = lay.cfg.groups, HW = H * Wd;
    T* h1 = alloc<T>(M * r.cout);
    T* res = r.has_res ? alloc<T>(M * r.cout) : nullptr;
    float* part1 = alloc<float>((size_t)B * kGnMaxSplit * G * 2);
    float* part2 = alloc<float>((size_t)B * kGnMaxSplit * G * 2);
    float* coefA = alloc<float>((size_t)B * r.cout);
    float* coefB = alloc<float>((size_t)B * r.cout);
    float* coefA2 = alloc<float>((size_t)B * r.cout);
    float* coefB2 = alloc<float>((size_t)B * r.cout);
    PRG_CHECK(arena.dry || (h1 && part1 && part2 && coefA && coefB && coefA2 && coefB2 && (!r.has_res || res)),
              "workspace exhausted (resblock)");
    const CondSrc* c1 = lay.cfg.conditional ? cs : nullptr;
    int rc, ns1 = 0, ns2 = 0, cd1 = 0, cd2 = 0, ad1 = 0, ad2 = 0;
    const GnApply g1 = gn_params(r.g1, r.b1, c1, r.ss_off);
    const GnApply g2 = gn_params(r.g2, r.b2, nullptr, 0);
    // bf16: statistics as fixed-point accumulators, coefficients folded by whoever consumes them (no gn_coeff launches)
    long long* acc1 = next_acc(B);
    long long* acc2 = next_acc(B);
    ConvOpt o1;
    o1.gn_partials = part1; o1.gn_nsplit = &ns1;
    o1.gn = &g1; o1.coef_a = coefA; o1.coef_b = coefB; o1.coef_done = &cd1;
    o1.gn_acc = acc1; o1.acc_done = &ad1;
    // conv2's own statistics fold into a second coefficient pair (coefA / coefB are still being read by its prologue)
    ConvOpt o2;
    o2.gn_partials = part2; o2.gn_nsplit = &ns2;
    o2.gn = &g2; o2.coef_a = coefA2; o2.coef_b = coefB2; o2.coef_done = &cd2;
    o2.gn_acc = acc2; o2.acc_done = &ad2;
    // PRG_FUSE_PRO: -1 (default) fuse wherever the conv supports it; 0 never; N > 0 only for widths >= N
    static const int fuse_min = [] { const char* e = std::getenv("PRG_FUSE_PRO"); return e ? std::atoi(e) : -1; }();
    // PRG_FUSE_PRO_MAX=N: widths above N apply GroupNorm + SiLU in one flat pass over h1 instead (a conv with several
    // output-channel tiles transforms every input pixel once per tile in its prologue)
    static const int fuse_max = [] { const char* e = std::getenv("PRG_FUSE_PRO_MAX"); return e ? std::atoi(e) : 1 << 30; }();
    const bool fuse_pro = conv_supports_prologue<T>(make_desc(r.c2, r.cout, 0, B, H, Wd, 1, 1, 0)) &&
                          (fuse_min < 0 || (fuse_min > 0 && r.cout >= fuse_min)) && r.cout <= fuse_max;
    GnFold f1{}, f2{};
    // h16 (conv.h): h1 only ever feeds conv2's fused prologue — stored as f16 when both launches land on kernels that implement it
    // (probed with the launches' own descriptors: PRG_H16=0, PRG_CONV_C64=0, ... and uncovered shapes fall back to bf16 h1)
    bool h16 = false;
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (!arena.dry && acc1 && fuse_pro && d_h16 && r.c2.h16_off >= 0) {
        const GnFold fp = make_fold(acc1, r.cout, HW, c1 && c1->ss_a, r.ss_off, r.pq1);
        ConvOpt p2 = o2;
        p2.fold = &fp; p2.pro_a = coefA; p2.pro_b = coefB;
        h16 = conv_h16_pair_ok(make_launch(r.c1, s0, C0, s1, C1, B, H, Wd, 1, 1, 0, o1, h1),
                               make_launch(r.c2, h1, r.cout, nullptr, 0, B, H, Wd, 1, 1, 0, p2, out));
      }
    }
    o1.out_f16 = h16;
    o2.in_f16 = h16;
    if ((rc = conv(r.c1, s0, C0, s1, C1, B, H, Wd, 1, 1, 0, o1, h1, s))) return rc;
    if (!arena.dry && ns1 == 0 && (rc = launch_gn_stats<T>(h1, part1, B, HW, r.cout, G, &ns1, s))) return rc;
    PRG_CHECK(!h16 || ad1, "resblock: the h16 probe promised fixed-point statistics");
    if (!arena.dry) {
      if (ad1) {
        f1 = make_fold(acc1, r.cout, HW, c1 && c1->ss_a, r.ss_off, r.pq1);
        if (fuse_pro) {
          o2.fold = &f1; o2.pro_a = coefA; o2.pro_b = coefB;    // (tables: scratch for kernels without the in-kernel fold)
        } else {
          if constexpr (std::is_same<T, bf16_t>::value) {
            if ((rc = launch_affine_silu_fold(h1, f1, nullptr, h1, B, HW, r.cout, s))) return rc;
          }
        }
      } else {
        if (!cd1 && (rc = launch_gn_coeff(part1, ns1, g1, coefA, coefB, B, HW, r.cout, G, s))) return rc;
        if (fuse_pro) {
          o2.pro_a = coefA; o2.pro_b = coefB;
        } else if ((rc = launch_affine_silu<T>(h1, coefA, coefB, nullptr, h1, B, HW, r.cout, s))) {
          return rc;
        }
      }
    }
    if ((rc = conv(r.c2, h1, r.cout, nullptr, 0, B, H, Wd, 1, 1, 0, o2, out, s))) return rc;
    if (!arena.dry && ns2 == 0 && (rc = launch_gn_stats<T>(out, part2, B, HW, r.cout, G, &ns2, s))) return rc;
    const T* skip = s0;
    bool fused_tail = false;
    if constexpr (std::is_same<T, bf16_t>::value)
      fused_tail = r.has_res && r.fw_res >= 0 && d_attn && resblock_tail_fused_supported(C0, C1, r.cout);
    // bf16: a res_conv the fused tail kernel does not cover (up levels 0-1: 768 -> 512, 384 -> 256) takes the tail into ITS
    // epilogue instead: out = Wres . cat[s0, s1] + b + SiLU(GroupNorm(h)), one launch, `res` never materialised either
    // (f16x3 too, round 4: the split-operand 1x1 kernel shares the transposing epilogue — `res` is not written and re-read, the
    //  GroupNorm + SiLU + skip pass of these blocks is gone; the exact-f32 parity mode keeps its separate passes)
    const bool epi_tail = (std::is_same<T, bf16_t>::value || (std::is_same<T, float>::value && d_split != nullptr)) && r.has_res && !fused_tail &&
                          r.cout % 8 == 0 && res_epilogue_enabled();
    if (fused_tail || epi_tail) {
      // res_conv folded into the tail pass below: `res` is never materialised
    } else if (r.has_res) {
      if ((rc = conv(r.res, s0, C0, s1, C1, B, H, Wd, 1, 0, 0, ConvOpt(), res, s))) return rc;
      skip = res;
    } else {
      PRG_CHECK(C1 == 0 && C0 == r.cout, "resblock: identity skip needs equal widths");
    }
    if (!arena.dry) {
      // GroupNorm + SiLU + skip.  Accumulator path: the tail kernels fold the coefficients themselves (the 1x1 res_conv's
      // epilogue reads tables: one gn_coeff_acc launch for those four blocks); slab path: gn_coeff launch, then the tables
      if (ad2) f2 = make_fold(acc2, r.cout, HW, false, 0, r.pq2);
      else if (!cd2 && (rc = launch_gn_coeff(part2, ns2, g2, coefA2, coefB2, B, HW, r.cout, G, s))) return rc;
      if constexpr (std::is_same<T, bf16_t>::value) {
        if (fused_tail) {
          rc = launch_resblock_tail_fused(out, coefA2, coefB2, s0, C0, s1, C1, d_attn + r.fw_res, F(r.res.b_off), out, B, HW,
                                          r.cout, s, head_w, head_b, head_out, head_sigmoid, ad2 ? &f2 : nullptr);
          head_done = head_out != nullptr;
          arena.reset(m);
          return rc;
        }
      }
      if (epi_tail) {
        ConvOpt ro;
        ro.residual = out;
        const bool fold_ok = ad2 && f2.cpg % 8 == 0;
        if (fold_ok) {
          ro.res_fold = &f2;                                  // coefficients folded in the res_conv's epilogue
        } else {
          if constexpr (std::is_same<T, bf16_t>::value) {
            if (ad2 && (rc = launch_gn_coeff_acc(f2, coefA2, coefB2, B, r.cout, s))) return rc;
          }
          ro.res_a = coefA2; ro.res_b = coefB2;
        }
        rc = conv(r.res, s0, C0, s1, C1, B, H, Wd, 1, 0, 0, ro, out, s);
        arena.reset(m);
        return rc;
      }
      if constexpr (std::is_same<T, bf16_t>::value) {
        if (ad2) {
          rc = launch_affine_silu_fold(out, f2, skip, out, B, HW, r.cout, s);
          arena.reset(m);
          return rc;
        }
      }
      if ((rc = launch_affine_silu<T>(out, coefA2, coefB2, skip, out, B, HW, r.cout, s))) return rc;
    } else if (epi_tail) {
      ConvOpt ro;                                            // dry run (workspace sizing): the same conv call
      if ((rc = conv(r.res, s0, C0, s1, C1, B, H, Wd, 1, 0, 0, ro, out, s))) return rc;
    }
    arena.reset(m);
    return PRG_OK;
  }

  // Residual(PreNorm(LinearAttention | Attention)) (sd:583-589, 631-639, 737-796)
  int attention(const AttnP& a, const T* x, T* out, int B, int H, int Wd, hipStream_t s) {
    const size_t m = arena.mark();
    const int N = H * Wd;
    const size_t M = (size_t)B * N;
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (a.linear && a.fw_qkv >= 0 && d_attn) {
        float* ws = alloc<float>(linattn_fused_ws_floats(B, N));
        PRG_CHECK(arena.dry || ws, "workspace exhausted (fused attention)");
        int rc = PRG_OK;
        if (!arena.dry)
          rc = launch_linear_attention_fused(x, d_attn + a.fw_qkv, d_attn + a.fw_out, F(a.out.b_off), F(a.out_g), out, ws, B, N,
                                             a.C, a.kshift >= 0 ? d_kshift + a.kshift : nullptr, s);
        arena.reset(m);
        return rc;
      }
    }
    if constexpr (std::is_same<T, float>::value) {
      if (a.linear && a.sp_qkv >= 0 && d_attn_split && linattn_split_supported(a.C, N)) {
        float* ws = alloc<float>(linattn_split_ws_floats(B, N));
        PRG_CHECK(arena.dry || ws, "workspace exhausted (split attention)");
        int rc = PRG_OK;
        if (!arena.dry) {
          const uint16_t* qh = d_attn_split + a.sp_qkv;
          const uint16_t* oh = d_attn_split + a.sp_out;
          rc = launch_linear_attention_split(x, qh, qh + (size_t)3 * kHidden * a.C, oh, oh + (size_t)a.C * kHidden, F(a.out.b_off),
                                             F(a.out_g), out, ws, B, N, a.C, s);
        }
        arena.reset(m);
        return rc;
      }
    }
    T* xn = alloc<T>(M * a.C);
    T* qkv = alloc<T>(M * 3 * kHidden);
    T* o = alloc<T>(M * kHidden);
    T* y = a.linear ? alloc<T>(M * a.C) : nullptr;
    float* ws = a.linear ? alloc<float>(linattn_ws_floats(B, N)) : nullptr;
    PRG_CHECK(arena.dry || (xn && qkv && o && (!a.linear || (y && ws))), "workspace exhausted (attention)");
    int rc;
    if (!arena.dry && (rc = launch_layernorm<T>(x, F(a.norm_g), nullptr, xn, (int64_t)M, a.C, s))) return rc;
    if ((rc = conv(a.qkv, xn, a.C, nullptr, 0, B, H, Wd, 1, 0, 0, ConvOpt(), qkv, s))) return rc;
    if (a.linear) {
      if (!arena.dry && (rc = launch_linear_attention<T>(qkv, o, ws, B, N, s))) return rc;
      if ((rc = conv(a.out, o, kHidden, nullptr, 0, B, H, Wd, 1, 0, 0, ConvOpt(), y, s))) return rc;
      if (!arena.dry && (rc = launch_layernorm<T>(y, F(a.out_g), x, out, (int64_t)M, a.C, s))) return rc;
    } else {
      bool done = false;
      if constexpr (std::is_same<T, bf16_t>::value) {
        if (!arena.dry && full_attention_mfma_supported(N)) {
          if ((rc = launch_full_attention_mfma(qkv, o, B, N, s))) return rc;
          done = true;
        }
      }
      if constexpr (std::is_same<T, float>::value) {
        if (!arena.dry && d_split && full_attention_split_supported(N)) {          // f16x3: split-f16 MFMAs (attn_split.hip)
          if ((rc = launch_full_attention_split(qkv, o, B, N, s))) return rc;
          done = true;
        }
      }
      if (!arena.dry && !done && (rc = launch_full_attention<T>(qkv, o, B, N, s))) return rc;
      { ConvOpt ro; ro.residual = x;
        if ((rc = conv(a.out, o, kHidden, nullptr, 0, B, H, Wd, 1, 0, 0, ro, out, s))) return rc; }
    }
    arena.reset(m);
    return PRG_OK;
  }

  void tap(const char* name, const void* p, int B, int C, int H, int Wd, bool nchw = false) {
    if (taps_on && !arena.dry) taps[name] = Tap{p, C, H, Wd, B, nchw};
  }

  int forward(const float* x_nchw, const CondSrc& cond, float* out, int B, int S, hipStream_t s) override {
    const Layout& L = lay;
    const int nl = L.L;
    PRG_CHECK(S % (1 << (nl - 1)) == 0 && (S >> (nl - 1)) >= 2, "forward: image size too small for the level count");
    PRG_CHECK((S * S) % 4 == 0, "forward: H*W must be a multiple of 4");
    arena.reset(0);
    if (taps_on) taps.clear();
    const CondSrc* cs = L.cfg.conditional ? &cond : nullptr;
    int rc;
    const int d0 = L.cfg.dim;
    // fixed-point GroupNorm statistics (bf16 path): one memset for every norm of the evaluation, one launch that folds the
    // conditioning (scale + 1, shift) of every ResnetBlock into P / Q — instead of one gn_coeff launch per norm
    gn_slot = 0;
    pq_dyn = nullptr;
    if (std::is_same<T, bf16_t>::value && gn_acc_enabled()) {
      // (the conditioned network clears the accumulators inside its cond_fold launch below: one launch fewer per evaluation)
      const bool fold_clears = acc_on() && !arena.dry && L.cfg.conditional && n_cond_entries > 0 && cs && cs->ss_a;
      if (acc_on() && !arena.dry && !fold_clears) PRG_HIP(hipMemsetAsync(d_gnacc, 0, gnacc_bytes, s));
      if (L.cfg.conditional && n_cond_entries > 0) {
        float* pq = alloc<float>((size_t)B * L.ss_total);
        PRG_CHECK(arena.dry || pq, "workspace exhausted (conditioning fold)");
        if (acc_on() && !arena.dry && cs && cs->ss_a) {
          GnApply ss{};
          ss.ss_a = cs->ss_a; ss.ss_a_stride = cs->ss_a_stride; ss.ss_b = cs->ss_b; ss.ss_b_stride = cs->ss_b_stride;
          ss.ss_a_row = cs->row; ss.ss_a_row_stride = cs->row_stride;
          if ((rc = launch_cond_fold(d_cond_entries, n_cond_entries, d_flat, ss, pq, L.ss_total, B, s, d_gnacc, (int64_t)(gnacc_bytes / sizeof(long long))))) return rc;
          pq_dyn = pq;
        }
      }
    }
    T* x0 = alloc<T>((size_t)B * S * S * d0);
    PRG_CHECK(arena.dry || x0, "workspace exhausted (stem)");
    bool stem_done = false;
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (!arena.dry && d_stem_frag && stem_conv_mfma_supported(L.cfg.in_channels, d0, S, S)) {
        if ((rc = launch_stem_conv_mfma(x_nchw, d_stem_frag, F(L.stem_b), x0, B, L.cfg.in_channels, S, S, s))) return rc;
        stem_done = true;
      }
    }
    if constexpr (std::is_same<T, float>::value) {
      if (!arena.dry && d_stem_split && stem_conv_mfma_supported(L.cfg.in_channels, d0, S, S)) {
        if ((rc = launch_stem_conv_mfma_split(x_nchw, d_stem_split, F(L.stem_b), d_stem_split_scale, x0, B, L.cfg.in_channels, S, S, s))) return rc;
        stem_done = true;
      }
    }
    if (!arena.dry && !stem_done &&
        (rc = launch_stem_conv<T>(x_nchw, d_stem, F(L.stem_b), x0, B, L.cfg.in_channels, S, S, d0, s)))
      return rc;
    tap("init_conv", x0, B, d0, S, S);
    std::vector<std::pair<const T*, int>> skips;
    const T* x = x0;
    int H = S;
    for (int i = 0; i < nl; ++i) {
      const LevelP& lv = L.downs[i];
      const int C = L.dims[i], Co = L.dims[i + 1];
      const size_t M = (size_t)B * H * H;
      T* s1 = alloc<T>(M * C);
      PRG_CHECK(arena.dry || s1, "workspace exhausted (down)");
      if ((rc = resblock(lv.r0, x, C, nullptr, 0, cs, s1, B, H, H, s))) return rc;
      skips.push_back({s1, C});
      if (i == 0) tap("down0_block0", s1, B, C, H, H);
      T* s2 = alloc<T>(M * C);
      const size_t mk = arena.mark();
      T* t = alloc<T>(M * C);
      PRG_CHECK(arena.dry || (s2 && t), "workspace exhausted (down)");
      if ((rc = resblock(lv.r1, s1, C, nullptr, 0, cs, t, B, H, H, s))) return rc;
      if ((rc = attention(lv.at, t, s2, B, H, H, s))) return rc;
      arena.reset(mk);
      skips.push_back({s2, C});
      if (i == 0) tap("down0_attn", s2, B, C, H, H);
      const int Ho = lv.strided ? H / 2 : H;
      T* xd = alloc<T>((size_t)B * Ho * Ho * Co);
      PRG_CHECK(arena.dry || xd, "workspace exhausted (downsample)");
      if ((rc = conv(lv.resample, s2, C, nullptr, 0, B, H, H, lv.strided ? 2 : 1, 1, 0, ConvOpt(), xd, s))) return rc;
      if (i == 0) tap("down0_out", xd, B, Co, Ho, Ho);
      x = xd;
      H = Ho;
    }
    {
      const int C = L.dims.back();
      const size_t M = (size_t)B * H * H;
      T* m2 = alloc<T>(M * C);
      T* m3 = alloc<T>(M * C);
      const size_t mk = arena.mark();
      T* m1 = alloc<T>(M * C);
      PRG_CHECK(arena.dry || (m1 && m2 && m3), "workspace exhausted (mid)");
      if ((rc = resblock(L.mid1, x, C, nullptr, 0, cs, m1, B, H, H, s))) return rc;
      if ((rc = attention(L.mid_at, m1, m2, B, H, H, s))) return rc;
      arena.reset(mk);
      tap("mid_attn", m2, B, C, H, H);
      if ((rc = resblock(L.mid2, m2, C, nullptr, 0, cs, m3, B, H, H, s))) return rc;
      x = m3;
    }
    for (int i = 0; i < nl; ++i) {
      const LevelP& lv = L.ups[i];
      const int Ci = L.dims[nl - 1 - i], Co = L.dims[nl - i];  // block width Co, resample to Ci
      const size_t M = (size_t)B * H * H;
      const int Ho = lv.strided ? 2 * H : H;
      T* xu = alloc<T>((size_t)B * Ho * Ho * Ci);
      const size_t mk = arena.mark();
      T* u1 = alloc<T>(M * Co);
      T* u2 = alloc<T>(M * Co);
      T* u3 = alloc<T>(M * Co);
      PRG_CHECK(arena.dry || (xu && u1 && u2 && u3), "workspace exhausted (up)");
      auto sk = skips.back(); skips.pop_back();
      if ((rc = resblock(lv.r0, x, Co, sk.first, sk.second, cs, u1, B, H, H, s))) return rc;
      sk = skips.back(); skips.pop_back();
      if ((rc = resblock(lv.r1, u1, Co, sk.first, sk.second, cs, u2, B, H, H, s))) return rc;
      if ((rc = attention(lv.at, u2, u3, B, H, H, s))) return rc;
      if ((rc = conv(lv.resample, u3, Co, nullptr, 0, B, H, H, 1, 1, lv.strided ? 1 : 0, ConvOpt(), xu, s))) return rc;
      arena.reset(mk);
      if (i == 0) tap("up0_out", xu, B, Ci, Ho, Ho);
      x = xu;
      H = Ho;
    }
    {
      const size_t M = (size_t)B * H * H;
      T* fr = alloc<T>(M * d0);
      PRG_CHECK(arena.dry || fr, "workspace exhausted (final)");
      // the final block's fused tail applies the 1x1 head to its LDS tile and writes only the network output (bf16 path,
      // no taps requested): `fr` is then never written
      head_done = false;
      if (!taps_on && !arena.dry && d0 == 64 && head_fuse_enabled()) {
        head_w = F(L.head_w); head_b = F(L.head_b); head_out = out; head_sigmoid = L.cfg.sigmoid_out;
      }
      rc = resblock(L.fin, x, d0, x0, d0, cs, fr, B, H, H, s);
      head_w = head_b = nullptr; head_out = nullptr;
      if (rc) return rc;
      tap("final_res", fr, B, d0, H, H);
      if (!arena.dry && !head_done && (rc = launch_head_conv<T>(fr, F(L.head_w), F(L.head_b), out, (int64_t)M, d0,
                                                                 L.cfg.sigmoid_out, s)))
        return rc;
    }
    return PRG_OK;
  }

  int measure(int B, int S, size_t* bytes) override {
    Arena saved = arena;
    arena = Arena();
    arena.dry = true;
    CondSrc cs;
    int rc = forward(nullptr, cs, nullptr, B, S, nullptr);
    // general-path conditioning scratch lives in the same arena, after the activations
    size_t extra = 0;
    if (lay.cfg.conditional) extra = ((size_t)B * (lay.cfg.dim + 5 * lay.emb + lay.ss_total) * sizeof(float) + 4096);
    if (lay.cfg.in_channels == 3) extra += (size_t)B * 3 * S * S * sizeof(float) + 4096;  // DepthAugment output
    *bytes = arena.high + extra + (1 << 20);
    arena = saved;
    return rc;
  }

  // general conditioning path (per-image timesteps): cond = cat[time_mlp(t), param_mlp(K)] -> every block's Linear
  int cond_general(const int64_t* time, const float* param_cond, int B, CondSrc* out, hipStream_t s) override {
    const Layout& L = lay;
    const int e = L.emb, d0 = L.cfg.dim;
    // placed at the top of the arena so the forward's stack (which starts at 0) cannot reach it
    size_t need = (size_t)B * (d0 + 5 * e + L.ss_total) * sizeof(float) + 4096;
    PRG_CHECK(arena.cap >= need, "workspace too small for conditioning");
    float* base = reinterpret_cast<float*>(arena.base + ((arena.cap - need) & ~(size_t)255));
    float* sinu = base;                       // [B][d0]
    float* h1 = sinu + (size_t)B * d0;        // [B][e]
    float* cat = h1 + (size_t)B * e;          // [B][2e]  = [t_emb | p_emb]
    float* h2 = cat + (size_t)B * 2 * e;      // [B][e]
    float* ss = h2 + (size_t)B * e;           // [B][ss_total]
    int rc;
    if ((rc = launch_sinusoidal(time, d_freqs, sinu, B, d0, s))) return rc;
    if ((rc = launch_linear(sinu, d0, 0, F(L.tm1_w), d0, 0, F(L.tm1_b), h1, e, B, d0, e, ACT_NONE, ACT_GELU, s))) return rc;
    if ((rc = launch_linear(h1, e, 0, F(L.tm3_w), e, 0, F(L.tm3_b), cat, 2 * e, B, e, e, ACT_NONE, ACT_NONE, s))) return rc;
    const int pc = L.cfg.param_cond_dim;
    if ((rc = launch_linear(param_cond, pc, 0, F(L.pm0_w), pc, 0, F(L.pm0_b), h2, e, B, pc, e, ACT_NONE, ACT_GELU, s))) return rc;
    if ((rc = launch_linear(h2, e, 0, F(L.pm2_w), e, 0, F(L.pm2_b), cat + e, 2 * e, B, e, e, ACT_NONE, ACT_NONE, s))) return rc;
    auto one = [&](const ResP& r) -> int {
      return launch_linear(cat, 2 * e, 0, F(r.mlp_w), 2 * e, 0, F(r.mlp_b), ss + r.ss_off, L.ss_total, B, 2 * e,
                           2 * r.cout, ACT_SILU, ACT_NONE, s);
    };
    for (auto& lv : L.downs) { if ((rc = one(lv.r0))) return rc; if ((rc = one(lv.r1))) return rc; }
    for (auto& lv : L.ups) { if ((rc = one(lv.r0))) return rc; if ((rc = one(lv.r1))) return rc; }
    if ((rc = one(L.mid1))) return rc;
    if ((rc = one(L.mid2))) return rc;
    if ((rc = one(L.fin))) return rc;
    out->ss_a = ss;
    out->ss_a_stride = L.ss_total;
    out->ss_b = nullptr;
    out->row = nullptr;
    return PRG_OK;
  }

  int tap_copy(const Tap& t, float* out, hipStream_t s) override {
    if (t.nchw_f32) {
      PRG_HIP(hipMemcpyAsync(out, t.ptr, (size_t)t.B * t.C * t.H * t.W * sizeof(float), hipMemcpyDeviceToDevice, s));
      return PRG_OK;
    }
    return launch_nhwc_to_nchw_f32<T>(reinterpret_cast<const T*>(t.ptr), out, t.B, t.H * t.W, t.C, s);
  }
};

// ---- weight preparation (host) ---------------------------------------------------------------
static bool fused_attention_enabled() {
  static const int on = [] { const char* e = std::getenv("PRG_FUSED_ATTN"); return e ? std::atoi(e) : 1; }();
  return on != 0;
}

static void standardize(const float* w, int Cout, int K, std::vector<float>& out) {
  out.resize((size_t)Cout * K);
  for (int o = 0; o < Cout; ++o) {
    double m = 0;
    for (int k = 0; k < K; ++k) m += w[(size_t)o * K + k];
    m /= K;
    double v = 0;
    for (int k = 0; k < K; ++k) { double d = w[(size_t)o * K + k] - m; v += d * d; }
    v /= K;
    const double rs = 1.0 / std::sqrt(v + 1e-5);
    for (int k = 0; k < K; ++k) out[(size_t)o * K + k] = (float)((w[(size_t)o * K + k] - m) * rs);
  }
}

template <typename T>
static void pack_all(Layout& L, const float* flat, std::vector<T>& packed) {
  std::vector<ConvP*> convs;
  auto add_res = [&](ResP& r) { convs.push_back(&r.c1); convs.push_back(&r.c2); if (r.has_res) convs.push_back(&r.res); };
  auto add_at = [&](AttnP& a) { convs.push_back(&a.qkv); convs.push_back(&a.out); };
  for (auto& lv : L.downs) { add_res(lv.r0); add_res(lv.r1); add_at(lv.at); convs.push_back(&lv.resample); }
  for (auto& lv : L.ups) { add_res(lv.r0); add_res(lv.r1); add_at(lv.at); convs.push_back(&lv.resample); }
  add_res(L.mid1); add_at(L.mid_at); add_res(L.mid2); add_res(L.fin);
  std::vector<float> tmp;
  std::vector<T> one;
  for (ConvP* p : convs) {
    const float* w = flat + p->w_flat;
    if (p->ws) { standardize(w, p->Cout, p->Cin * p->KH * p->KW, tmp); w = tmp.data(); }
    pack_conv_weight<T>(w, p->Cout, p->Cin, p->KH, p->KW, one, &p->CoutPad, &p->kchunks);
    size_t off = (packed.size() + 127) / 128 * 128;  // 256-byte aligned tiles
    packed.resize(off + one.size());
    std::memcpy(packed.data() + off, one.data(), one.size() * sizeof(T));
    p->w_off = off;
    if (std::is_same<T, bf16_t>::value && p->KH == 4 && p->KW == 4 && p->Cin % 64 == 0 && p->Cout % 64 == 0) {
      // Downsample: second packing for the 256-pixel kernel's 2 x 2-tap mode (conv_w256.hip)
      std::vector<float> eq;
      s2d_equivalent_weights(w, p->Cout, p->Cin, eq);
      int cp = 0;
      pack_conv_weight<T>(eq.data(), p->Cout, 4 * p->Cin, 3, 3, one, &cp, &p->s2d_kchunks);
      off = (packed.size() + 127) / 128 * 128;
      packed.resize(off + one.size());
      std::memcpy(packed.data() + off, one.data(), one.size() * sizeof(T));
      p->s2d_off = (int64_t)off;
    }
  }
  if (std::is_same<T, bf16_t>::value) {
    // Upsample convs (up levels whose resample is nearest x2 + 3 x 3): third packing, the sub-pixel decomposition (conv_w256.hip MODE 2)
    for (auto& lv : L.ups) {
      ConvP* p = &lv.resample;
      if (!lv.strided || !(p->KH == 3 && p->KW == 3 && p->Cin % 64 == 0 && (p->Cout == 64 || p->Cout % 128 == 0))) continue;
      const float* w = flat + p->w_flat;
      if (p->ws) { standardize(w, p->Cout, p->Cin * 9, tmp); w = tmp.data(); }
      std::vector<float> eq;
      up_equivalent_weights(w, p->Cout, p->Cin, eq);
      size_t off = (packed.size() + 127) / 128 * 128;
      p->up_off = (int64_t)off;
      for (int ph = 0; ph < 4; ++ph) {
        int cp = 0, kc = 0;
        pack_conv_weight<T>(eq.data() + (size_t)ph * p->Cout * p->Cin * 4, p->Cout, p->Cin, 2, 2, one, &cp, &kc);
        packed.resize(off + one.size());
        std::memcpy(packed.data() + off, one.data(), one.size() * sizeof(T));
        off += one.size();
      }
    }
  }
}

static void collect_convs(Layout& L, std::vector<ConvP*>& convs) {
  auto add_res = [&](ResP& r) { convs.push_back(&r.c1); convs.push_back(&r.c2); if (r.has_res) convs.push_back(&r.res); };
  auto add_at = [&](AttnP& a) { convs.push_back(&a.qkv); convs.push_back(&a.out); };
  for (auto& lv : L.downs) { add_res(lv.r0); add_res(lv.r1); add_at(lv.at); convs.push_back(&lv.resample); }
  for (auto& lv : L.ups) { add_res(lv.r0); add_res(lv.r1); add_at(lv.at); convs.push_back(&lv.resample); }
  add_res(L.mid1); add_at(L.mid_at); add_res(L.mid2); add_res(L.fin);
}

template <typename T>
static int create_impl(const prg_unet_config* cfg, const float* weights, int64_t n, prg_unet** out, bool mx = false, bool split = false) {
  std::unique_ptr<UnetImpl<T>> u(new UnetImpl<T>());
  int rc = build_layout(*cfg, u->lay);
  if (rc) return rc;
  if (u->lay.total != n)
    return fail(PRG_E_INVALID, "prg_unet_create: expected " + std::to_string(u->lay.total) + " floats, got " +
                                   std::to_string(n));
  std::vector<T> packed;
  pack_all<T>(u->lay, weights, packed);
  const Layout& L = u->lay;
  std::vector<float> stem((size_t)49 * L.cfg.in_channels * L.cfg.dim);
  for (int o = 0; o < L.cfg.dim; ++o)
    for (int c = 0; c < L.cfg.in_channels; ++c)
      for (int t = 0; t < 49; ++t)
        stem[((size_t)t * L.cfg.in_channels + c) * L.cfg.dim + o] = weights[L.stem_w + ((size_t)o * L.cfg.in_channels + c) * 49 + t];
  if (hipMalloc(&u->d_flat, (size_t)n * sizeof(float)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(flat weights)");
  if (hipMalloc(&u->d_packed, packed.size() * sizeof(T)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(packed weights)");
  if (hipMalloc(&u->d_stem, stem.size() * sizeof(float)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(stem weights)");
  PRG_HIP(hipMemcpy(u->d_flat, weights, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  PRG_HIP(hipMemcpy(u->d_packed, packed.data(), packed.size() * sizeof(T), hipMemcpyHostToDevice));
  PRG_HIP(hipMemcpy(u->d_stem, stem.data(), stem.size() * sizeof(float), hipMemcpyHostToDevice));
  {
    // default frequency table: the reference's expression in float32 with this host's libm (the Python front-end
    // replaces it with torch's own evaluation, which is what the reference would compute on the same host)
    const int half = L.cfg.dim / 2;
    std::vector<float> fr(half);
    const float stepf = -(float)(9.210340371976184 / (double)(half - 1));
    for (int i = 0; i < half; ++i) fr[i] = std::exp((float)i * stepf);
    if (hipMalloc(&u->d_freqs, half * sizeof(float)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(time frequencies)");
    PRG_HIP(hipMemcpy(u->d_freqs, fr.data(), half * sizeof(float), hipMemcpyHostToDevice));
  }
  if (mx) {
    // MX-fp8 copies of every 3x3 conv weight whose widths are 64-channel multiples (what conv3x3_mx_kernel covers); the
    // 1x1 / 4x4 / stem convs and everything that is not a convolution stay bf16.
    std::vector<ConvP*> convs;
    collect_convs(u->lay, convs);
    std::vector<uint8_t> data, scales, one, ones;
    std::vector<float> tmp;
    for (ConvP* p : convs) {
      if (!(p->KH == 3 && p->KW == 3 && p->Cin % 64 == 0 && p->Cout % 64 == 0)) continue;
      const float* w = weights + p->w_flat;
      if (p->ws) { standardize(w, p->Cout, p->Cin * 9, tmp); w = tmp.data(); }
      int cp = 0, kc = 0;
      pack_conv_weight_mxfp8(w, p->Cout, p->Cin, 3, 3, one, ones, &cp, &kc);
      p->mx_off = (int64_t)((data.size() + 255) / 256 * 256);
      data.resize((size_t)p->mx_off + one.size());
      std::memcpy(data.data() + p->mx_off, one.data(), one.size());
      p->mx_soff = (int64_t)((scales.size() + 255) / 256 * 256);
      scales.resize((size_t)p->mx_soff + ones.size());
      std::memcpy(scales.data() + p->mx_soff, ones.data(), ones.size());
    }
    if (!data.empty()) {
      if (hipMalloc(&u->d_mx, data.size()) != hipSuccess || hipMalloc(&u->d_mx_scale, scales.size()) != hipSuccess)
        return fail(PRG_E_NOMEM, "hipMalloc(MX-fp8 weights)");
      PRG_HIP(hipMemcpy(u->d_mx, data.data(), data.size(), hipMemcpyHostToDevice));
      PRG_HIP(hipMemcpy(u->d_mx_scale, scales.data(), scales.size(), hipMemcpyHostToDevice));
    }
  }
  if (std::is_same<T, bf16_t>::value) {
    // h16 (conv.h): f16 twins of the ResnetBlocks' second convs (3 x 3, Cin = Cout, 64-channel multiples), standardised like the
    // bf16 packing; conv2 takes them when the block's h1 tensor is stored as f16 (mxfp8 handles: the 64-channel pairs only —
    // conv_h16_pair_ok — the wide convs run on their MX copies)
    Layout& Lm = u->lay;
    std::vector<ResP*> rs;
    for (auto& lv : Lm.downs) { rs.push_back(&lv.r0); rs.push_back(&lv.r1); }
    for (auto& lv : Lm.ups) { rs.push_back(&lv.r0); rs.push_back(&lv.r1); }
    rs.push_back(&Lm.mid1); rs.push_back(&Lm.mid2); rs.push_back(&Lm.fin);
    std::vector<uint16_t> data, one;
    std::vector<float> tmp;
    for (ResP* r : rs) {
      ConvP* p = &r->c2;
      if (!(p->KH == 3 && p->KW == 3 && p->Cin == p->Cout && p->Cin % 64 == 0)) continue;
      const float* w = weights + p->w_flat;
      if (p->ws) { standardize(w, p->Cout, p->Cin * 9, tmp); w = tmp.data(); }
      pack_conv_weight_f16(w, p->Cout, p->Cin, 3, 3, one);
      p->h16_off = (int64_t)((data.size() + 127) / 128 * 128);
      data.resize((size_t)p->h16_off + one.size());
      std::memcpy(data.data() + p->h16_off, one.data(), one.size() * sizeof(uint16_t));
    }
    if (!data.empty()) {
      if (hipMalloc(&u->d_h16, data.size() * sizeof(uint16_t)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(h16 weights)");
      PRG_HIP(hipMemcpy(u->d_h16, data.data(), data.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
  }
  if (split) {
    // f16x3 mode: hi / lo f16 halves of every conv weight (standardised first where the Block does), conv_split.hip
    std::vector<ConvP*> convs;
    collect_convs(u->lay, convs);
    std::vector<uint16_t> data, one;
    std::vector<float> tmp, scales, sc1;
    for (ConvP* p : convs) {
      const float* w = weights + p->w_flat;
      if (p->ws) { standardize(w, p->Cout, p->Cin * p->KH * p->KW, tmp); w = tmp.data(); }
      int cp = 0;
      pack_conv_weight_split(w, p->Cout, p->Cin, p->KH, p->KW, one, &cp, &p->sp_kchunks, &sc1);
      p->sp_off = (int64_t)((data.size() + 127) / 128 * 128);
      data.resize((size_t)p->sp_off + one.size());
      std::memcpy(data.data() + p->sp_off, one.data(), one.size() * sizeof(uint16_t));
      p->sp_scale_off = (int64_t)scales.size();
      scales.insert(scales.end(), sc1.begin(), sc1.end());
    }
    // Upsample convs: the sub-pixel decomposition's four 2 x 2-tap packings (conv_split.hip, UP form of the wave-specialised kernel)
    // (built only when the option is on — it is off by default, conv_split.hip: try_launch_conv_split — ADVICE round 4)
    static const int split_up_on = [] { const char* e = std::getenv("PRG_SPLIT_UP2X2"); return e ? std::atoi(e) : 0; }();
    for (auto& lv : u->lay.ups) {
      ConvP* p = &lv.resample;
      if (!split_up_on || !lv.strided || !(p->KH == 3 && p->KW == 3 && p->Cin % 32 == 0 && p->Cout % 128 == 0)) continue;
      const float* w = weights + p->w_flat;
      if (p->ws) { standardize(w, p->Cout, p->Cin * p->KH * p->KW, tmp); w = tmp.data(); }   // (as the bf16 twin in pack_all; no Upsample conv is)
      std::vector<float> eq;
      up_equivalent_weights(w, p->Cout, p->Cin, eq);
      p->up_sp_off = (int64_t)((data.size() + 127) / 128 * 128);
      p->up_sp_scale_off = (int64_t)scales.size();
      size_t off = (size_t)p->up_sp_off;
      for (int ph = 0; ph < 4; ++ph) {
        int cp = 0, kc = 0;
        pack_conv_weight_split(eq.data() + (size_t)ph * p->Cout * p->Cin * 4, p->Cout, p->Cin, 2, 2, one, &cp, &kc, &sc1);
        data.resize(off + one.size());
        std::memcpy(data.data() + off, one.data(), one.size() * sizeof(uint16_t));
        off += one.size();
        scales.insert(scales.end(), sc1.begin(), sc1.end());     // [phase][CoutPad]
      }
    }
    if (hipMalloc(&u->d_split_scale, scales.size() * sizeof(float)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(split scales)");
    PRG_HIP(hipMemcpy(u->d_split_scale, scales.data(), scales.size() * sizeof(float), hipMemcpyHostToDevice));
    if (hipMalloc(&u->d_split, data.size() * sizeof(uint16_t)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(split weights)");
    PRG_HIP(hipMemcpy(u->d_split, data.data(), data.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    // fused linear attention (attn_split.hip): to_qkv with the PreNorm gain folded in (q and k rows times log2 e: both only ever
    // enter a softmax, evaluated with exp2) and to_out, each as f16 hi halves followed by the lo halves.  PRG_SPLIT_ATTN=0: unfused
    static const int sa_on = [] { const char* e = std::getenv("PRG_SPLIT_ATTN"); return e ? std::atoi(e) : 1; }();
    std::vector<uint16_t> aw;
    auto f16bits = [](float v, uint16_t& h, uint16_t& l) {
      const _Float16 a = (_Float16)v, b = (_Float16)(v - (float)a);
      std::memcpy(&h, &a, 2);
      std::memcpy(&l, &b, 2);
    };
    auto add_attn = [&](AttnP& a) {
      if (!sa_on || !a.linear || (a.C != 64 && a.C != 128)) return;   // (linattn_split_supported: C = 64 / 128; the token count is checked per call)
      aw.resize((aw.size() + 63) / 64 * 64);
      a.sp_qkv = (int64_t)aw.size();
      const size_t nq = (size_t)3 * kHidden * a.C;
      aw.resize(aw.size() + 2 * nq);
      for (int o = 0; o < 3 * kHidden; ++o)
        for (int c = 0; c < a.C; ++c) {
          const float v = weights[a.qkv.w_flat + (size_t)o * a.C + c] * weights[a.norm_g + c] * (o < 2 * kHidden ? 1.4426950408889634f : 1.0f);
          f16bits(v, aw[a.sp_qkv + (size_t)o * a.C + c], aw[a.sp_qkv + nq + (size_t)o * a.C + c]);
        }
      aw.resize((aw.size() + 63) / 64 * 64);
      a.sp_out = (int64_t)aw.size();
      const size_t no = (size_t)a.C * kHidden;
      aw.resize(aw.size() + 2 * no);
      for (size_t i = 0; i < no; ++i) f16bits(weights[a.out.w_flat + i], aw[a.sp_out + i], aw[a.sp_out + no + i]);
    };
    for (auto& lv : u->lay.downs) add_attn(lv.at);
    for (auto& lv : u->lay.ups) add_attn(lv.at);
    if (!aw.empty()) {
      if (hipMalloc(&u->d_attn_split, aw.size() * sizeof(uint16_t)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(split attention weights)");
      PRG_HIP(hipMemcpy(u->d_attn_split, aw.data(), aw.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
  }
  if (std::is_same<T, bf16_t>::value) {
    // fixed-point GroupNorm statistics (common.h, GnFold): P = gamma, Q = beta of every norm in 16-byte aligned rows (what
    // the unconditioned norms use directly) and the table cond_fold_kernel walks for the conditioned ones
    std::vector<float> pq;
    std::vector<CondFoldEntry> ent;
    auto add = [&](ResP& r) {
      r.cpad = (r.cout + 3) & ~3;
      auto put = [&](int64_t g_off, int64_t b_off) {
        const int64_t o = (int64_t)pq.size();
        pq.resize(pq.size() + 2 * (size_t)r.cpad, 0.0f);
        std::memcpy(pq.data() + o, weights + g_off, sizeof(float) * r.cout);
        std::memcpy(pq.data() + o + r.cpad, weights + b_off, sizeof(float) * r.cout);
        return o;
      };
      r.pq1 = put(r.g1, r.b1);
      r.pq2 = put(r.g2, r.b2);
      if (L.cfg.conditional) ent.push_back(CondFoldEntry{r.ss_off, r.cout, (long long)r.g1, (long long)r.b1});
    };
    for (auto& lv : u->lay.downs) { add(lv.r0); add(lv.r1); }
    for (auto& lv : u->lay.ups) { add(lv.r0); add(lv.r1); }
    add(u->lay.mid1); add(u->lay.mid2); add(u->lay.fin);
    if (hipMalloc(&u->d_pq_static, pq.size() * sizeof(float)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(norm gains)");
    PRG_HIP(hipMemcpy(u->d_pq_static, pq.data(), pq.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!ent.empty()) {
      if (hipMalloc(&u->d_cond_entries, ent.size() * sizeof(CondFoldEntry)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(cond entries)");
      PRG_HIP(hipMemcpy(u->d_cond_entries, ent.data(), ent.size() * sizeof(CondFoldEntry), hipMemcpyHostToDevice));
      u->n_cond_entries = (int)ent.size();
    }
  }
  if (hipMalloc(&u->d_tickets, kMaxTicketImages * sizeof(int)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(tickets)");
  PRG_HIP(hipMemset(u->d_tickets, 0, kMaxTicketImages * sizeof(int)));
  static const int stem3_on = [] { const char* e = std::getenv("PRG_STEM_MFMA3"); return e ? std::atoi(e) : 1; }();   // 0: MaskUnet's stem on the direct kernel
  if (std::is_same<T, bf16_t>::value && (L.cfg.in_channels == 1 || (L.cfg.in_channels == 3 && stem3_on)) && L.cfg.dim == 64) {
    std::vector<bf16_t> sf;
    pack_stem_mfma_weights(weights + L.stem_w, L.cfg.in_channels, sf);
    if (hipMalloc(&u->d_stem_frag, sf.size() * sizeof(bf16_t)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(stem fragments)");
    PRG_HIP(hipMemcpy(u->d_stem_frag, sf.data(), sf.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
  }
  // f16x3 (round 5): the stem on the same MFMA kernel with f16 hi / lo halves (PRG_SPLIT_STEM=0: the direct fmaf kernel of the parity mode)
  static const int split_stem_on = [] { const char* e = std::getenv("PRG_SPLIT_STEM"); return e ? std::atoi(e) : 1; }();
  if (split && split_stem_on && (L.cfg.in_channels == 1 || L.cfg.in_channels == 3) && L.cfg.dim == 64) {
    std::vector<uint16_t> sf;
    std::vector<float> sc;
    pack_stem_mfma_weights_split(weights + L.stem_w, L.cfg.in_channels, sf, sc);
    if (hipMalloc(&u->d_stem_split, sf.size() * sizeof(uint16_t)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(split stem fragments)");
    PRG_HIP(hipMemcpy(u->d_stem_split, sf.data(), sf.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    if (hipMalloc(&u->d_stem_split_scale, sc.size() * sizeof(float)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(split stem scales)");
    PRG_HIP(hipMemcpy(u->d_stem_split_scale, sc.data(), sc.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if (std::is_same<T, bf16_t>::value && fused_attention_enabled()) {
    // fused linear attention (attn_fused.hip): to_qkv with the PreNorm gain folded in, to_out as is, both [out][in] bf16
    std::vector<bf16_t> aw;
    std::vector<float> ks;
    // PRG_LA_KSHIFT=0 forces the measured column maxima (the la_kmax pass) for every block
    static const int kshift_on = [] { const char* e = std::getenv("PRG_LA_KSHIFT"); return e ? std::atoi(e) : 1; }();
    auto add = [&](AttnP& a) {
      if (!a.linear || !linattn_fused_supported(a.C)) return;
      aw.resize((aw.size() + 63) / 64 * 64);
      a.fw_qkv = (int64_t)aw.size();
      for (int o = 0; o < 3 * kHidden; ++o)
        for (int c = 0; c < a.C; ++c)
          // q and k only ever enter a softmax: their rows carry log2(e), so the kernels exponentiate with a bare v_exp_f32
          aw.push_back(f32_to_bf16(weights[a.qkv.w_flat + (size_t)o * a.C + c] * weights[a.norm_g + c] *
                                   (o < 2 * kHidden ? 1.4426950408889634f : 1.0f)));
      // Softmax over pixels of k[n][d] = w_d . LN(x_n): a LayerNorm output has norm <= sqrt(C), so |k| <= ||w_d|| sqrt(C)
      // (Cauchy-Schwarz; w_d = the bf16 weights the kernel multiplies with, 2 % slack for the bf16 rounding of LN(x)).
      // exp(k - bound) >= exp(-2 bound): with bound <= 40 nothing underflows and the column maxima need not be measured.
      // (k, hence the bound, in units of 1 / log2(e): the rows above are pre-scaled.)
      float shifts[kHidden];
      bool ok = kshift_on != 0;
      for (int d = 0; d < kHidden; ++d) {
        double n2 = 0;
        for (int c = 0; c < a.C; ++c) {
          const double w = bf16_to_f32(aw[(size_t)a.fw_qkv + (size_t)(kHidden + d) * a.C + c]);
          n2 += w * w;
        }
        shifts[d] = (float)(1.02 * std::sqrt(n2 * a.C));
        ok = ok && shifts[d] <= 40.0f * 1.4426950408889634f;
      }
      // the same bound for the q rows (softmax over the 32 d of a head, per pixel): one shift per head
      float qsh[kHeads];
      for (int h = 0; h < kHeads; ++h) {
        double worst = 0;
        for (int d = 0; d < kDimHead; ++d) {
          double n2 = 0;
          for (int c = 0; c < a.C; ++c) {
            const double w = bf16_to_f32(aw[(size_t)a.fw_qkv + (size_t)(h * kDimHead + d) * a.C + c]);
            n2 += w * w;
          }
          worst = std::max(worst, 1.02 * std::sqrt(n2 * a.C));
        }
        qsh[h] = (float)worst;
        ok = ok && qsh[h] <= 40.0f * 1.4426950408889634f;
      }
      if (ok) {
        a.kshift = (int64_t)ks.size();
        ks.insert(ks.end(), shifts, shifts + kHidden);
        ks.insert(ks.end(), qsh, qsh + kHeads);
        ks.resize((ks.size() + 3) & ~(size_t)3);
      }
      a.fw_out = (int64_t)aw.size();
      for (int c = 0; c < a.C; ++c)
        for (int j = 0; j < kHidden; ++j) aw.push_back(f32_to_bf16(weights[a.out.w_flat + (size_t)c * kHidden + j]));
    };
    for (auto& lv : u->lay.downs) add(lv.at);
    for (auto& lv : u->lay.ups) add(lv.at);
    // fused ResnetBlock tail: raw res_conv weights [Cout][Cin] as bf16
    auto add_res = [&](ResP& r) {
      if (!r.has_res || r.res.b_off < 0 || !resblock_tail_fused_supported(r.cin / 2, r.cin - r.cin / 2, r.cout)) return;
      aw.resize((aw.size() + 63) / 64 * 64);
      r.fw_res = (int64_t)aw.size();
      for (size_t i = 0; i < (size_t)r.cout * r.cin; ++i) aw.push_back(f32_to_bf16(weights[r.res.w_flat + i]));
    };
    for (auto& lv : u->lay.ups) { add_res(lv.r0); add_res(lv.r1); }
    add_res(u->lay.fin);
    if (!aw.empty()) {
      if (hipMalloc(&u->d_attn, aw.size() * sizeof(bf16_t)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(attention weights)");
      PRG_HIP(hipMemcpy(u->d_attn, aw.data(), aw.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
    }
    if (!ks.empty()) {
      if (hipMalloc(&u->d_kshift, ks.size() * sizeof(float)) != hipSuccess) return fail(PRG_E_NOMEM, "hipMalloc(softmax shifts)");
      PRG_HIP(hipMemcpy(u->d_kshift, ks.data(), ks.size() * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  *out = u.release();
  return PRG_OK;
}

static int reserve(prg_unet* h, int B, int S) {
  if (B <= h->resB && S <= h->resS && h->arena.base) return PRG_OK;
  size_t bytes = 0;
  const int nb = B > h->resB ? B : h->resB, ns = S > h->resS ? S : h->resS;
  int rc = h->measure(nb, ns, &bytes);
  if (rc) return rc;
  PRG_HIP(hipDeviceSynchronize());
  if (h->arena.base) PRG_HIP(hipFree(h->arena.base));
  h->arena = Arena();
  if (hipMalloc(reinterpret_cast<void**>(&h->arena.base), bytes) != hipSuccess) {
    h->arena.base = nullptr;
    h->resB = h->resS = 0;
    return fail(PRG_E_NOMEM, "hipMalloc(workspace " + std::to_string(bytes >> 20) + " MiB)");
  }
  h->arena.cap = bytes;
  ++h->arena_gen;
  h->resB = nb;
  h->resS = ns;
  if (h->d_pq_static) {   // (bf16 / mxfp8 handles only)
    // fixed-point GroupNorm accumulators: two norms per ResnetBlock, [slots][resB][groups][2] int64
    if (h->d_gnacc) PRG_HIP(hipFree(h->d_gnacc));
    h->d_gnacc = nullptr;
    h->gn_slots = 2 * (4 * h->lay.L + 3);
    h->gnacc_bytes = (size_t)h->gn_slots * nb * h->lay.cfg.groups * 2 * sizeof(long long);
    if (hipMalloc(reinterpret_cast<void**>(&h->d_gnacc), h->gnacc_bytes) != hipSuccess) {
      h->d_gnacc = nullptr;
      return fail(PRG_E_NOMEM, "hipMalloc(GroupNorm accumulators)");
    }
  }
  return PRG_OK;
}

}  // namespace prg

// =============================================================================================
// sampler handle
// =============================================================================================
struct prg_sampler {
  prg_unet* unet = nullptr;
  int B = 0, S = 0, n_steps = 0;
  std::vector<prg_step> steps;
  prg_step* d_steps = nullptr;
  float* d_tpart = nullptr;   // [n_steps][ss_total]  time half of every block's conditioning (+ bias)
  float* d_ppart = nullptr;   // [B][ss_total]        camera-parameter half
  float* d_scratch = nullptr; // embedding scratch
  float* d_x = nullptr;       // state (B, S*S)
  float* d_u = nullptr;       // network output
  int* d_step = nullptr;
  uint64_t* d_seeds = nullptr;
  hipStream_t own_stream = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  const float* g_cond = nullptr;   // pointers baked into the captured graph
  const float* g_noise = nullptr;
  float* g_out = nullptr;
  hipStream_t g_stream = nullptr;
  uint64_t g_arena_gen = 0;        // workspace generation the graph was captured against
  bool use_graph = true;
  ProfileSink prof;
  double last_total_ms = 0;
};

namespace prg {

static void sampler_free(prg_sampler* h) {
  if (h->exec) (void)hipGraphExecDestroy(h->exec);
  if (h->graph) (void)hipGraphDestroy(h->graph);
  h->exec = nullptr;
  h->graph = nullptr;
}

static int sampler_one_step(prg_sampler* h, const float* cond, const float* noise, float* out, hipStream_t s) {
  CondSrc cs;
  cs.ss_a = h->d_tpart;
  cs.ss_a_stride = 0;
  cs.row = h->d_step;
  cs.row_stride = h->unet->lay.ss_total;
  cs.ss_b = h->d_ppart;
  cs.ss_b_stride = h->unet->lay.ss_total;
  int rc = h->unet->forward(h->d_x, cs, h->d_u, h->B, h->S, s);
  if (rc) return rc;
  SamplerStepArgs a;
  a.x = h->d_x; a.u = h->d_u; a.cond = cond; a.noise = noise; a.steps = h->d_steps; a.step_idx = h->d_step;
  a.seeds = h->d_seeds; a.final_out = out; a.B = h->B; a.HW = h->S * h->S; a.n_steps = h->n_steps;
  a.ticket = h->d_step + 1;
  if (h->prof.on) {
    ProfileSink& p = h->prof;
    if (p.step_used == p.step_pool.size()) {
      hipEvent_t e0, e1;
      PRG_HIP(hipEventCreate(&e0));
      PRG_HIP(hipEventCreate(&e1));
      p.step_pool.push_back({e0, e1});
    }
    auto& ev = p.step_pool[p.step_used++];
    PRG_HIP(hipEventRecord(ev.first, s));
    rc = launch_sampler_step(a, s);
    PRG_HIP(hipEventRecord(ev.second, s));
    p.step_launches += 1;
    return rc;
  }
  return launch_sampler_step(a, s);   // its last workgroup advances the step counter
}

}  // namespace prg

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

int prg_abi_version(void) { return PRG_ABI_VERSION; }
const char* prg_last_error(void) { return prg::last_error(); }

int prg_device_info(char* name, size_t name_len, int* compute_units) {
  int dev = 0;
  PRG_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  PRG_HIP(hipGetDeviceProperties(&p, dev));
  if (name && name_len) {
    std::strncpy(name, p.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  if (compute_units) *compute_units = p.multiProcessorCount;
  return PRG_OK;
}

int64_t prg_unet_param_count(const prg_unet_config* cfg) {
  if (!cfg) return PRG_E_INVALID;
  Layout L;
  if (build_layout(*cfg, L)) return PRG_E_INVALID;
  return L.total;
}

int prg_unet_create(const prg_unet_config* cfg, const float* weights, int64_t n_floats, int dtype, prg_unet** out) {
  PRG_CHECK(cfg && weights && out, "prg_unet_create: null pointer");
  PRG_CHECK(dtype == PRG_F32 || dtype == PRG_BF16 || dtype == PRG_MXFP8 || dtype == PRG_F16X3,
            "prg_unet_create: dtype must be PRG_F32, PRG_BF16, PRG_MXFP8 or PRG_F16X3");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(PRG_E_HIP, "prg_unet_create: no HIP device");
  *out = nullptr;
  int rc = (dtype == PRG_F32 || dtype == PRG_F16X3) ? create_impl<float>(cfg, weights, n_floats, out, false, dtype == PRG_F16X3)
                                                     : create_impl<bf16_t>(cfg, weights, n_floats, out, dtype == PRG_MXFP8);
  if (rc == PRG_OK) (*out)->dtype = dtype;
  return rc;
}

int prg_unet_destroy(prg_unet* h) {
  if (!h) return PRG_OK;
  (void)hipDeviceSynchronize();
  if (h->d_flat) (void)hipFree(h->d_flat);
  if (h->d_packed) (void)hipFree(h->d_packed);
  if (h->d_stem) (void)hipFree(h->d_stem);
  if (h->d_attn) (void)hipFree(h->d_attn);
  if (h->d_kshift) (void)hipFree(h->d_kshift);
  if (h->d_freqs) (void)hipFree(h->d_freqs);
  if (h->d_mx) (void)hipFree(h->d_mx);
  if (h->d_mx_scale) (void)hipFree(h->d_mx_scale);
  if (h->d_split) (void)hipFree(h->d_split);
  if (h->d_split_scale) (void)hipFree(h->d_split_scale);
  if (h->d_h16) (void)hipFree(h->d_h16);
  if (h->d_attn_split) (void)hipFree(h->d_attn_split);
  if (h->d_tickets) (void)hipFree(h->d_tickets);
  if (h->d_gnacc) (void)hipFree(h->d_gnacc);
  if (h->d_pq_static) (void)hipFree(h->d_pq_static);
  if (h->d_cond_entries) (void)hipFree(h->d_cond_entries);
  if (h->d_stem_frag) (void)hipFree(h->d_stem_frag);
  if (h->d_stem_split) (void)hipFree(h->d_stem_split);
  if (h->d_stem_split_scale) (void)hipFree(h->d_stem_split_scale);
  if (h->arena.base) (void)hipFree(h->arena.base);
  delete h;
  return PRG_OK;
}

int prg_unet_set_time_freqs(prg_unet* h, const float* freqs, int n) {
  PRG_CHECK(h && freqs, "prg_unet_set_time_freqs: null pointer");
  PRG_CHECK(n == h->lay.cfg.dim / 2, "prg_unet_set_time_freqs: need dim/2 frequencies");
  PRG_HIP(hipDeviceSynchronize());
  PRG_HIP(hipMemcpy(h->d_freqs, freqs, sizeof(float) * n, hipMemcpyHostToDevice));
  return PRG_OK;
}

int prg_unet_reserve(prg_unet* h, int B, int S) {
  PRG_CHECK(h && B > 0 && S > 0, "prg_unet_reserve: bad arguments");
  return reserve(h, B, S);
}

int prg_unet_set_taps(prg_unet* h, int enable) {
  PRG_CHECK(h, "prg_unet_set_taps: null handle");
  h->taps_on = enable != 0;
  return PRG_OK;
}

int prg_unet_get_tap(prg_unet* h, const char* name, float* out, int64_t cap, int* C, int* H, int* W, void* stream) {
  PRG_CHECK(h && name && out, "prg_unet_get_tap: null pointer");
  auto it = h->taps.find(name);
  if (it == h->taps.end()) return fail(PRG_E_STATE, std::string("no such tap recorded: ") + name);
  const Tap& t = it->second;
  PRG_CHECK((int64_t)t.B * t.C * t.H * t.W <= cap, "prg_unet_get_tap: output buffer too small");
  if (C) *C = t.C;
  if (H) *H = t.H;
  if (W) *W = t.W;
  return h->tap_copy(t, out, (hipStream_t)stream);
}

int prg_unet_forward(prg_unet* h, const float* x, const int64_t* time, const float* param_cond, float* out, int B,
                     int S, void* stream) {
  PRG_CHECK(h && x && out, "prg_unet_forward: null pointer");
  PRG_CHECK(h->lay.cfg.conditional && time && param_cond, "prg_unet_forward: handle is not a conditional U-Net");
  PRG_CHECK(h->lay.cfg.in_channels == 1, "prg_unet_forward: in_channels must be 1");
  PRG_CHECK(B > 0 && S > 0, "prg_unet_forward: bad shape");
  int rc = reserve(h, B, S);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  CondSrc cs;
  if ((rc = h->cond_general(time, param_cond, B, &cs, s))) return rc;
  return h->forward(x, cs, out, B, S, s);
}

int prg_maskunet_forward(prg_unet* h, const float* depth, float* prob, int B, int S, void* stream) {
  PRG_CHECK(h && depth && prob, "prg_maskunet_forward: null pointer");
  PRG_CHECK(!h->lay.cfg.conditional && h->lay.cfg.in_channels == 3, "prg_maskunet_forward: handle is not a MaskUnet");
  PRG_CHECK(B > 0 && S > 0, "prg_maskunet_forward: bad shape");
  int rc = reserve(h, B, S);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  // DepthAugment output (B,3,S,S) float32 sits at the top of the arena, out of the forward stack's reach
  const size_t aug_bytes = (size_t)B * 3 * S * S * sizeof(float);
  PRG_CHECK(h->arena.cap > aug_bytes + 4096, "workspace too small");
  float* aug = reinterpret_cast<float*>(h->arena.base + ((h->arena.cap - aug_bytes - 256) & ~(size_t)255));
  if ((rc = prg_depth_augment(depth, aug, B, S, S, stream))) return rc;
  CondSrc cs;
  rc = h->forward(aug, cs, prob, B, S, s);
  if (h->taps_on) h->taps["augment"] = Tap{aug, 3, S, S, B, true};
  return rc;
}

// ---------------------------------------------------------------------------------------------
// float32-storage handles (PRG_F32: the exact-f32 kernels; PRG_F16X3: the split-operand kernels of conv_split.hip)
static int debug_conv_f32(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                          int dtype, int K, int stride, int pad, hipStream_t s, int ups = 0) {
  const size_t M = (size_t)B * H * W;
  const int Ho = ups ? 2 * H : (H + 2 * pad - K) / stride + 1, Wo = ups ? 2 * W : (W + 2 * pad - K) / stride + 1;
  const size_t Mo = (size_t)B * Ho * Wo;
  std::vector<float> packed;
  std::vector<uint16_t> sp;
  int cp = 0, kc = 0, cp2 = 0, kc32 = 0;
  pack_conv_weight<float>(w, Cout, Cin, K, K, packed, &cp, &kc);
  std::vector<float> spsc;
  if (dtype == PRG_F16X3) pack_conv_weight_split(w, Cout, Cin, K, K, sp, &cp2, &kc32, &spsc);
  // Upsample (nearest x2, then the 3x3): the sub-pixel form's four 2 x 2-tap packings, as create_impl builds them (ADVICE round 4:
  // the UP form of conv3x3_split_ws_kernel gets a kernel-level test)
  std::vector<uint16_t> spu;
  std::vector<float> spusc;
  if (ups && dtype == PRG_F16X3 && K == 3 && Cin % 32 == 0 && Cout % 128 == 0) {
    std::vector<float> eq, sc1;
    std::vector<uint16_t> one;
    up_equivalent_weights(w, Cout, Cin, eq);
    for (int ph = 0; ph < 4; ++ph) {
      int cpu_ = 0, kcu_ = 0;
      pack_conv_weight_split(eq.data() + (size_t)ph * Cout * Cin * 4, Cout, Cin, 2, 2, one, &cpu_, &kcu_, &sc1);
      spu.insert(spu.end(), one.begin(), one.end());
      spusc.insert(spusc.end(), sc1.begin(), sc1.end());
    }
  }
  std::vector<float> zb(Cout, 0.0f);
  void *d_in = nullptr, *d_out = nullptr, *d_w = nullptr, *d_b = nullptr, *d_sp = nullptr, *d_sc = nullptr, *d_spu = nullptr, *d_scu = nullptr;
  auto cleanup = [&]() { for (void* p : {d_in, d_out, d_w, d_b, d_sp, d_sc, d_spu, d_scu}) if (p) (void)hipFree(p); };
  if (hipMalloc(&d_in, M * Cin * 4) != hipSuccess || hipMalloc(&d_out, Mo * Cout * 4) != hipSuccess ||
      hipMalloc(&d_w, packed.size() * 4) != hipSuccess || hipMalloc(&d_b, Cout * 4) != hipSuccess ||
      (!sp.empty() && (hipMalloc(&d_sp, sp.size() * 2) != hipSuccess || hipMalloc(&d_sc, spsc.size() * 4) != hipSuccess)) ||
      (!spu.empty() && (hipMalloc(&d_spu, spu.size() * 2) != hipSuccess || hipMalloc(&d_scu, spusc.size() * 4) != hipSuccess))) {
    cleanup();
    return fail(PRG_E_NOMEM, "prg_debug_conv: hipMalloc failed");
  }
  if (d_spu && (hipMemcpy(d_spu, spu.data(), spu.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(d_scu, spusc.data(), spusc.size() * 4, hipMemcpyHostToDevice) != hipSuccess)) {
    cleanup();
    return fail(PRG_E_HIP, "prg_debug_conv: hipMemcpy failed");
  }
  if (hipMemcpy(d_w, packed.data(), packed.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_b, bias ? bias : zb.data(), Cout * 4, hipMemcpyHostToDevice) != hipSuccess ||
      (d_sp && (hipMemcpy(d_sp, sp.data(), sp.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(d_sc, spsc.data(), spsc.size() * 4, hipMemcpyHostToDevice) != hipSuccess))) {
    cleanup();
    return fail(PRG_E_HIP, "prg_debug_conv: hipMemcpy failed");
  }
  int rc = launch_nchw_f32_to_nhwc<float>(x, reinterpret_cast<float*>(d_in), B, H * W, Cin, s);
  if (rc == PRG_OK) {
    ConvLaunch<float> L{};
    L.d.B = B; L.d.Hin = H; L.d.Win = W; L.d.C0 = Cin; L.d.C1 = 0; L.d.ups = ups; L.d.KH = K; L.d.KW = K; L.d.stride = stride; L.d.pad = pad;
    L.d.Hout = Ho; L.d.Wout = Wo; L.d.Cout = Cout; L.d.CoutPad = cp; L.d.kchunks = kc;
    L.src0 = reinterpret_cast<const float*>(d_in); L.w = reinterpret_cast<const float*>(d_w);
    L.bias = reinterpret_cast<const float*>(d_b); L.out = reinterpret_cast<float*>(d_out);
    L.gn_groups = 8;
    L.w_split = reinterpret_cast<const uint16_t*>(d_sp); L.split_kchunks = kc32;
    L.split_scale = reinterpret_cast<const float*>(d_sc);
    L.w_up_split = reinterpret_cast<const uint16_t*>(d_spu);
    L.split_scale_up = reinterpret_cast<const float*>(d_scu);
    rc = launch_conv<float>(L, s, nullptr);
  }
  if (rc == PRG_OK) rc = launch_nhwc_to_nchw_f32<float>(reinterpret_cast<const float*>(d_out), out, B, Ho * Wo, Cout, s);
  if (hipStreamSynchronize(s) != hipSuccess && rc == PRG_OK) rc = fail(PRG_E_HIP, "prg_debug_conv: stream synchronise failed");
  cleanup();
  return rc;
}

static int debug_conv(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                      int dtype, int K, int stride, void* stream, int ups = 0) {
  PRG_CHECK(x && w && out, "prg_debug_conv3x3: null pointer");
  PRG_CHECK(B > 0 && H > 0 && W > 0 && Cin % 8 == 0 && Cout % 8 == 0, "prg_debug_conv3x3: bad shape");
  PRG_CHECK(dtype == PRG_BF16 || dtype == PRG_MXFP8 || dtype == PRG_F32 || dtype == PRG_F16X3, "prg_debug_conv3x3: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PRG_F32 || dtype == PRG_F16X3) return debug_conv_f32(x, w, bias, out, B, Cin, Cout, H, W, dtype, K, stride, K == 1 ? 0 : 1, s, ups);
  const size_t M = (size_t)B * H * W;
  const int Ho = ups ? 2 * H : H / stride, Wo = ups ? 2 * W : W / stride;
  const size_t Mo = (size_t)B * Ho * Wo;
  std::vector<bf16_t> packed;
  int cp = 0, kc = 0;
  pack_conv_weight<bf16_t>(w, Cout, Cin, K, K, packed, &cp, &kc);
  std::vector<bf16_t> packed_up;
  if (ups && Cin % 64 == 0 && (Cout == 64 || Cout % 128 == 0)) {
    std::vector<float> eq;
    std::vector<bf16_t> one;
    up_equivalent_weights(w, Cout, Cin, eq);
    for (int ph = 0; ph < 4; ++ph) {
      int cp2 = 0, kc2 = 0;
      pack_conv_weight<bf16_t>(eq.data() + (size_t)ph * Cout * Cin * 4, Cout, Cin, 2, 2, one, &cp2, &kc2);
      packed_up.insert(packed_up.end(), one.begin(), one.end());
    }
  }
  std::vector<bf16_t> packed_s2d;
  int kc_s2d = 0;
  if (K == 4 && Cin % 64 == 0 && Cout % 64 == 0) {
    std::vector<float> eq;
    int cp2 = 0;
    s2d_equivalent_weights(w, Cout, Cin, eq);
    pack_conv_weight<bf16_t>(eq.data(), Cout, 4 * Cin, 3, 3, packed_s2d, &cp2, &kc_s2d);
  }
  std::vector<uint8_t> mxd, mxs;
  if (dtype == PRG_MXFP8) {
    PRG_CHECK(Cin % 64 == 0 && Cout % 64 == 0, "prg_debug_conv3x3: MX-fp8 needs 64-channel multiples");
    int cp2 = 0, kc2 = 0;
    pack_conv_weight_mxfp8(w, Cout, Cin, 3, 3, mxd, mxs, &cp2, &kc2);
  }
  std::vector<float> zb(Cout, 0.0f);
  void *d_in = nullptr, *d_out = nullptr, *d_w = nullptr, *d_b = nullptr, *d_mxd = nullptr, *d_mxs = nullptr, *d_w2 = nullptr, *d_wu = nullptr;
  auto cleanup = [&]() { for (void* p : {d_in, d_out, d_w, d_b, d_mxd, d_mxs, d_w2, d_wu}) if (p) (void)hipFree(p); };
  if (hipMalloc(&d_in, M * Cin * 2) != hipSuccess || hipMalloc(&d_out, Mo * Cout * 2) != hipSuccess ||
      (!packed_up.empty() && (hipMalloc(&d_wu, packed_up.size() * 2) != hipSuccess ||
                              hipMemcpy(d_wu, packed_up.data(), packed_up.size() * 2, hipMemcpyHostToDevice) != hipSuccess)) ||
      (!packed_s2d.empty() && hipMalloc(&d_w2, packed_s2d.size() * 2) != hipSuccess) ||
      hipMalloc(&d_w, packed.size() * 2) != hipSuccess || hipMalloc(&d_b, Cout * 4) != hipSuccess ||
      (dtype == PRG_MXFP8 && (hipMalloc(&d_mxd, mxd.size()) != hipSuccess || hipMalloc(&d_mxs, mxs.size()) != hipSuccess))) {
    cleanup();
    return fail(PRG_E_NOMEM, "prg_debug_conv3x3: hipMalloc failed");
  }
  bool up = hipMemcpy(d_w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice) == hipSuccess &&
            (!d_w2 || hipMemcpy(d_w2, packed_s2d.data(), packed_s2d.size() * 2, hipMemcpyHostToDevice) == hipSuccess) &&
            hipMemcpy(d_b, bias ? bias : zb.data(), Cout * 4, hipMemcpyHostToDevice) == hipSuccess;
  if (up && dtype == PRG_MXFP8)
    up = hipMemcpy(d_mxd, mxd.data(), mxd.size(), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_mxs, mxs.data(), mxs.size(), hipMemcpyHostToDevice) == hipSuccess;
  if (!up) {
    cleanup();
    return fail(PRG_E_HIP, "prg_debug_conv3x3: hipMemcpy failed");
  }
  int rc = launch_nchw_f32_to_nhwc<bf16_t>(x, reinterpret_cast<bf16_t*>(d_in), B, H * W, Cin, s);
  if (rc == PRG_OK) {
    ConvLaunch<bf16_t> L{};
    L.d.B = B; L.d.Hin = H; L.d.Win = W; L.d.C0 = Cin; L.d.C1 = 0; L.d.ups = ups; L.d.KH = K; L.d.KW = K; L.d.stride = stride; L.d.pad = 1;
    L.d.Hout = Ho; L.d.Wout = Wo; L.d.Cout = Cout; L.d.CoutPad = cp; L.d.kchunks = kc;
    L.src0 = reinterpret_cast<const bf16_t*>(d_in); L.w = reinterpret_cast<const bf16_t*>(d_w);
    L.bias = reinterpret_cast<const float*>(d_b); L.out = reinterpret_cast<bf16_t*>(d_out);
    L.gn_groups = 8;
    L.w_up = reinterpret_cast<const bf16_t*>(d_wu);
    L.w_s2d = reinterpret_cast<const bf16_t*>(d_w2); L.s2d_kchunks = kc_s2d;
    L.w_mx = reinterpret_cast<const uint8_t*>(d_mxd); L.w_mx_scale = reinterpret_cast<const uint8_t*>(d_mxs);
    L.mx_pure = 1;
    rc = launch_conv<bf16_t>(L, s, nullptr);
  }
  if (rc == PRG_OK) rc = launch_nhwc_to_nchw_f32<bf16_t>(reinterpret_cast<const bf16_t*>(d_out), out, B, Ho * Wo, Cout, s);
  if (hipStreamSynchronize(s) != hipSuccess && rc == PRG_OK) rc = fail(PRG_E_HIP, "prg_debug_conv3x3: stream synchronise failed");
  cleanup();
  return rc;
}

// The two convolutions of a ResnetBlock's Block pair in bf16 mode, through the library's own dispatch (prg.h).
int prg_debug_block_pair(const float* x, const float* w1, const float* b1, const float* gamma, const float* beta, const float* w2,
                         const float* b2, float* out, int B, int Cin, int C, int H, int W, int groups, int h16, void* stream) {
  PRG_CHECK(x && w1 && b1 && gamma && beta && w2 && b2 && out, "prg_debug_block_pair: null pointer");
  PRG_CHECK(B > 0 && H > 0 && W > 0 && Cin % 64 == 0 && C % 64 == 0 && groups > 0 && C % groups == 0 && (C / groups) % 8 == 0,
            "prg_debug_block_pair: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const size_t M = (size_t)B * H * W;
  std::vector<bf16_t> p1, p2;
  std::vector<uint16_t> p2h;
  int cp1 = 0, kc1 = 0, cp2 = 0, kc2 = 0;
  pack_conv_weight<bf16_t>(w1, C, Cin, 3, 3, p1, &cp1, &kc1);
  pack_conv_weight<bf16_t>(w2, C, C, 3, 3, p2, &cp2, &kc2);
  pack_conv_weight_f16(w2, C, C, 3, 3, p2h);
  std::vector<float> pq(2 * (size_t)C);
  for (int c = 0; c < C; ++c) { pq[c] = gamma[c]; pq[C + c] = beta[c]; }
  void *d_x = nullptr, *d_h = nullptr, *d_y = nullptr, *d_w1 = nullptr, *d_w2 = nullptr, *d_w2h = nullptr, *d_b1 = nullptr, *d_b2 = nullptr,
       *d_pq = nullptr, *d_acc = nullptr, *d_part = nullptr, *d_coef = nullptr;
  auto cleanup = [&]() { for (void* p : {d_x, d_h, d_y, d_w1, d_w2, d_w2h, d_b1, d_b2, d_pq, d_acc, d_part, d_coef}) if (p) (void)hipFree(p); };
  const size_t acc_bytes = (size_t)B * groups * 2 * sizeof(long long);
  if (hipMalloc(&d_x, M * Cin * 2) != hipSuccess || hipMalloc(&d_h, M * C * 2) != hipSuccess || hipMalloc(&d_y, M * C * 2) != hipSuccess ||
      hipMalloc(&d_w1, p1.size() * 2) != hipSuccess || hipMalloc(&d_w2, p2.size() * 2) != hipSuccess || hipMalloc(&d_w2h, p2h.size() * 2) != hipSuccess ||
      hipMalloc(&d_b1, C * 4) != hipSuccess || hipMalloc(&d_b2, C * 4) != hipSuccess || hipMalloc(&d_pq, pq.size() * 4) != hipSuccess ||
      hipMalloc(&d_acc, acc_bytes) != hipSuccess || hipMalloc(&d_part, (size_t)B * kGnMaxSplit * groups * 2 * 4) != hipSuccess ||
      hipMalloc(&d_coef, (size_t)2 * B * C * 4) != hipSuccess) {
    cleanup();
    return fail(PRG_E_NOMEM, "prg_debug_block_pair: hipMalloc failed");
  }
  if (hipMemcpy(d_w1, p1.data(), p1.size() * 2, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_w2, p2.data(), p2.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_w2h, p2h.data(), p2h.size() * 2, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_b1, b1, C * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_b2, b2, C * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_pq, pq.data(), pq.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemsetAsync(d_acc, 0, acc_bytes, s) != hipSuccess) {
    cleanup();
    return fail(PRG_E_HIP, "prg_debug_block_pair: upload failed");
  }
  int rc = launch_nchw_f32_to_nhwc<bf16_t>(x, reinterpret_cast<bf16_t*>(d_x), B, H * W, Cin, s);
  ConvLaunch<bf16_t> L1{}, L2{};
  auto desc = [&](ConvLaunch<bf16_t>& L, int cin, int cp, int kc) {
    L.d.B = B; L.d.Hin = H; L.d.Win = W; L.d.C0 = cin; L.d.C1 = 0; L.d.ups = 0; L.d.KH = 3; L.d.KW = 3; L.d.stride = 1; L.d.pad = 1;
    L.d.Hout = H; L.d.Wout = W; L.d.Cout = C; L.d.CoutPad = cp; L.d.kchunks = kc;
    L.gn_groups = groups;
  };
  desc(L1, Cin, cp1, kc1);
  L1.src0 = reinterpret_cast<const bf16_t*>(d_x); L1.w = reinterpret_cast<const bf16_t*>(d_w1); L1.bias = reinterpret_cast<const float*>(d_b1);
  L1.out = reinterpret_cast<bf16_t*>(d_h);
  L1.gn_partials = reinterpret_cast<float*>(d_part); L1.gn_acc = reinterpret_cast<long long*>(d_acc);
  desc(L2, C, cp2, kc2);
  L2.src0 = reinterpret_cast<const bf16_t*>(d_h); L2.w = reinterpret_cast<const bf16_t*>(d_w2); L2.w_f16 = reinterpret_cast<const uint16_t*>(d_w2h);
  L2.bias = reinterpret_cast<const float*>(d_b2); L2.out = reinterpret_cast<bf16_t*>(d_y);
  GnFold f{};
  f.acc = reinterpret_cast<const long long*>(d_acc); f.P = reinterpret_cast<const float*>(d_pq); f.Q = f.P + C; f.pq_stride = 0;
  f.G = groups; f.cpg = C / groups; f.inv_n = 1.0f / ((float)(H * W) * (float)f.cpg);
  L2.pro_fold = f;
  L2.pro_a = reinterpret_cast<const float*>(d_coef); L2.pro_b = L2.pro_a + (size_t)B * C;
  if (rc == PRG_OK && h16) {
    if (!conv_h16_pair_ok(L1, L2)) rc = fail(PRG_E_INVALID, "prg_debug_block_pair: the kernels this shape dispatches to do not implement the f16 format");
    L1.out_f16 = 1;
    L2.in_f16 = 1;
  }
  int ns = 0, cd = 0, ad = 0;
  if (rc == PRG_OK) rc = launch_conv<bf16_t>(L1, s, &ns, &cd, &ad);
  if (rc == PRG_OK && !ad) rc = fail(PRG_E_INVALID, "prg_debug_block_pair: conv1's kernel does not accumulate fixed-point statistics for this shape");
  if (rc == PRG_OK) rc = launch_conv<bf16_t>(L2, s, nullptr);
  if (rc == PRG_OK) rc = launch_nhwc_to_nchw_f32<bf16_t>(reinterpret_cast<const bf16_t*>(d_y), out, B, H * W, C, s);
  if (hipStreamSynchronize(s) != hipSuccess && rc == PRG_OK) rc = fail(PRG_E_HIP, "prg_debug_block_pair: stream synchronise failed");
  cleanup();
  return rc;
}

int prg_debug_conv3x3(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                      int dtype, void* stream) {
  return debug_conv(x, w, bias, out, B, Cin, Cout, H, W, dtype, 3, 1, stream);
}

int prg_debug_conv(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                   int dtype, int K, int stride, void* stream) {
  PRG_CHECK((K == 1 || K == 3 || K == 4) && (stride == 1 || stride == 2), "prg_debug_conv: K must be 1, 3 or 4, stride 1 or 2");
  PRG_CHECK(dtype == PRG_F32 || dtype == PRG_F16X3 || K != 1, "prg_debug_conv: 1x1 convs only in the float32-storage modes");
  PRG_CHECK(stride == 1 || (H % 2 == 0 && W % 2 == 0), "prg_debug_conv: odd image size");
  return debug_conv(x, w, bias, out, B, Cin, Cout, H, W, dtype, K, stride, stream);
}

int prg_debug_upsample_conv3x3(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                               void* stream) {
  return debug_conv(x, w, bias, out, B, Cin, Cout, H, W, PRG_BF16, 3, 1, stream, 1);
}

int prg_debug_upsample_conv(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                            int dtype, void* stream) {
  return debug_conv(x, w, bias, out, B, Cin, Cout, H, W, dtype, 3, 1, stream, 1);
}

int prg_debug_conv4x4s2(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                        void* stream) {
  PRG_CHECK(H % 2 == 0 && W % 2 == 0, "prg_debug_conv4x4s2: odd image size");
  return debug_conv(x, w, bias, out, B, Cin, Cout, H, W, PRG_BF16, 4, 2, stream);
}

// ---------------------------------------------------------------------------------------------
int prg_sampler_create(prg_unet* unet, const prg_step* steps, int n_steps, int B, int S, prg_sampler** out) {
  PRG_CHECK(unet && steps && out, "prg_sampler_create: null pointer");
  PRG_CHECK(unet->lay.cfg.conditional && unet->lay.cfg.in_channels == 1, "prg_sampler_create: needs the conditional U-Net");
  PRG_CHECK(n_steps > 0 && B > 0 && S > 0, "prg_sampler_create: bad sizes");
  int rc = reserve(unet, B, S);
  if (rc) return rc;
  std::unique_ptr<prg_sampler> h(new prg_sampler());
  h->unet = unet; h->B = B; h->S = S; h->n_steps = n_steps;
  h->steps.assign(steps, steps + n_steps);
  const Layout& L = unet->lay;
  const int e = L.emb, d0 = L.cfg.dim, W = L.ss_total;
  const size_t HW = (size_t)S * S;
  const int R = n_steps > B ? n_steps : B;
  if (hipMalloc(&h->d_steps, sizeof(prg_step) * n_steps) != hipSuccess ||
      hipMalloc(&h->d_tpart, sizeof(float) * (size_t)n_steps * W) != hipSuccess ||
      hipMalloc(&h->d_ppart, sizeof(float) * (size_t)B * W) != hipSuccess ||
      hipMalloc(&h->d_scratch, sizeof(float) * (size_t)R * (d0 + 2 * e) + sizeof(int32_t) * n_steps) != hipSuccess ||
      hipMalloc(&h->d_x, sizeof(float) * B * HW) != hipSuccess || hipMalloc(&h->d_u, sizeof(float) * B * HW) != hipSuccess ||
      hipMalloc(&h->d_step, 2 * sizeof(int)) != hipSuccess || hipMalloc(&h->d_seeds, sizeof(uint64_t) * B) != hipSuccess)
    return fail(PRG_E_NOMEM, "prg_sampler_create: hipMalloc failed");
  PRG_HIP(hipMemcpy(h->d_steps, steps, sizeof(prg_step) * n_steps, hipMemcpyHostToDevice));
  PRG_HIP(hipStreamCreateWithFlags(&h->own_stream, hipStreamDefault));
  // time half of the conditioning for every transition: Tpart[k] = W_t . SiLU(time_mlp(t_k)) + bias
  {
    hipStream_t s = h->own_stream;
    float* sinu = h->d_scratch;
    float* h1 = sinu + (size_t)R * d0;
    float* temb = h1 + (size_t)R * e;
    int32_t* tdev = reinterpret_cast<int32_t*>(temb + (size_t)R * e);
    std::vector<int32_t> tt(n_steps);
    for (int k = 0; k < n_steps; ++k) tt[k] = steps[k].t;
    PRG_HIP(hipMemcpyAsync(tdev, tt.data(), sizeof(int32_t) * n_steps, hipMemcpyHostToDevice, s));
    const float* F = unet->d_flat;
    if ((rc = launch_sinusoidal_i32(tdev, unet->d_freqs, sinu, n_steps, d0, s))) return rc;
    if ((rc = launch_linear(sinu, d0, 0, F + L.tm1_w, d0, 0, F + L.tm1_b, h1, e, n_steps, d0, e, ACT_NONE, ACT_GELU, s))) return rc;
    if ((rc = launch_linear(h1, e, 0, F + L.tm3_w, e, 0, F + L.tm3_b, temb, e, n_steps, e, e, ACT_NONE, ACT_NONE, s))) return rc;
    auto one = [&](const ResP& r) -> int {
      return launch_linear(temb, e, 0, F + r.mlp_w, 2 * e, 0, F + r.mlp_b, h->d_tpart + r.ss_off, W, n_steps, e,
                           2 * r.cout, ACT_SILU, ACT_NONE, s);
    };
    for (auto& lv : L.downs) { if ((rc = one(lv.r0))) return rc; if ((rc = one(lv.r1))) return rc; }
    for (auto& lv : L.ups) { if ((rc = one(lv.r0))) return rc; if ((rc = one(lv.r1))) return rc; }
    if ((rc = one(L.mid1))) return rc;
    if ((rc = one(L.mid2))) return rc;
    if ((rc = one(L.fin))) return rc;
    PRG_HIP(hipStreamSynchronize(s));
  }
  *out = h.release();
  return PRG_OK;
}

int prg_sampler_destroy(prg_sampler* h) {
  if (!h) return PRG_OK;
  (void)hipDeviceSynchronize();
  sampler_free(h);
  for (auto& ev : h->prof.pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  for (auto& ev : h->prof.step_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  void* ptrs[] = {h->d_steps, h->d_tpart, h->d_ppart, h->d_scratch, h->d_x, h->d_u, h->d_step, h->d_seeds};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  delete h;
  return PRG_OK;
}

int prg_sampler_set_graph(prg_sampler* h, int enable) {
  PRG_CHECK(h, "prg_sampler_set_graph: null handle");
  h->use_graph = enable != 0;
  return PRG_OK;
}

int prg_sampler_set_profile(prg_sampler* h, int enable) {
  PRG_CHECK(h, "prg_sampler_set_profile: null handle");
  h->prof.on = enable != 0;
  return PRG_OK;
}

int prg_sampler_get_profile_executed(prg_sampler* h, double* conv_flops_executed) {
  PRG_CHECK(h && conv_flops_executed, "prg_sampler_get_profile_executed: null argument");
  *conv_flops_executed = h->prof.conv_flops_exec;
  return PRG_OK;
}

int prg_sampler_get_profile_bytes(prg_sampler* h, double* conv_bytes) {
  PRG_CHECK(h && conv_bytes, "prg_sampler_get_profile_bytes: null argument");
  *conv_bytes = h->prof.conv_bytes;
  return PRG_OK;
}

int prg_sampler_get_profile_shapes(prg_sampler* h, prg_profile_shape* rows, int32_t max_rows, int32_t* n_rows) {
  PRG_CHECK(h && n_rows && (rows || max_rows == 0), "prg_sampler_get_profile_shapes: null argument");
  const int n = (int)h->prof.shapes.size();
  *n_rows = n;
  for (int i = 0; i < n && i < max_rows; ++i) rows[i] = h->prof.shapes[i];
  return PRG_OK;
}

int prg_sampler_get_profile_step(prg_sampler* h, double* step_ms, int64_t* step_launches) {
  PRG_CHECK(h && step_ms && step_launches, "prg_sampler_get_profile_step: null argument");
  *step_ms = h->prof.step_ms;
  *step_launches = h->prof.step_launches;
  return PRG_OK;
}

int prg_sampler_get_profile(prg_sampler* h, double* conv_ms, int64_t* conv_launches, double* conv_flops,
                            double* total_ms) {
  PRG_CHECK(h, "prg_sampler_get_profile: null handle");
  if (conv_ms) *conv_ms = h->prof.conv_ms;
  if (conv_launches) *conv_launches = h->prof.launches;
  if (conv_flops) *conv_flops = h->prof.conv_flops;
  if (total_ms) *total_ms = h->last_total_ms;
  return PRG_OK;
}

int prg_sampler_run(prg_sampler* h, const float* param_cond, const float* img_cond, const float* noise,
                    int64_t noise_slabs, const uint64_t* seeds, float* out, void* stream) {
  PRG_CHECK(h && param_cond && out, "prg_sampler_run: null pointer");
  PRG_CHECK(noise || seeds, "prg_sampler_run: need stored noise or per-scene seeds");
  if (noise) {
    int64_t need = 1;
    for (int k = 0; k < h->n_steps; ++k)
      if (h->steps[k].sigma != 0.0f) need = k + 2;
    PRG_CHECK(noise_slabs >= need, "prg_sampler_run: stored noise has too few slabs for this transition table");
  }
  prg_unet* u = h->unet;
  PRG_CHECK(u->resB >= h->B && u->resS >= h->S, "prg_sampler_run: U-Net workspace was shrunk");
  hipStream_t s = stream ? (hipStream_t)stream : h->own_stream;
  const Layout& L = u->lay;
  const int e = L.emb, W = L.ss_total, B = h->B;
  int rc;
  if (seeds) PRG_HIP(hipMemcpyAsync(h->d_seeds, seeds, sizeof(uint64_t) * B, hipMemcpyHostToDevice, s));
  PRG_HIP(hipMemsetAsync(h->d_step, 0, 2 * sizeof(int), s));   // [step counter, arrival ticket]
  // camera half of the conditioning: Ppart[b] = W_p . SiLU(param_mlp(K_b))
  {
    float* h2 = h->d_scratch;
    float* pemb = h2 + (size_t)B * e;
    const float* F = u->d_flat;
    const int pc = L.cfg.param_cond_dim;
    if ((rc = launch_linear(param_cond, pc, 0, F + L.pm0_w, pc, 0, F + L.pm0_b, h2, e, B, pc, e, ACT_NONE, ACT_GELU, s))) return rc;
    if ((rc = launch_linear(h2, e, 0, F + L.pm2_w, e, 0, F + L.pm2_b, pemb, e, B, e, e, ACT_NONE, ACT_NONE, s))) return rc;
    auto one = [&](const ResP& r) -> int {
      return launch_linear(pemb, e, 0, F + r.mlp_w, 2 * e, e, nullptr, h->d_ppart + r.ss_off, W, B, e, 2 * r.cout,
                           ACT_SILU, ACT_NONE, s);
    };
    for (auto& lv : L.downs) { if ((rc = one(lv.r0))) return rc; if ((rc = one(lv.r1))) return rc; }
    for (auto& lv : L.ups) { if ((rc = one(lv.r0))) return rc; if ((rc = one(lv.r1))) return rc; }
    if ((rc = one(L.mid1))) return rc;
    if ((rc = one(L.mid2))) return rc;
    if ((rc = one(L.fin))) return rc;
  }
  if ((rc = launch_sampler_init(h->d_x, noise, h->d_seeds, B, h->S * h->S, s))) return rc;

  const bool profiling = h->prof.on;
  u->prof = profiling ? &h->prof : nullptr;
  h->prof.conv_ms = 0; h->prof.conv_flops = 0; h->prof.conv_flops_exec = 0; h->prof.conv_bytes = 0; h->prof.launches = 0; h->prof.used = 0;
  h->prof.step_ms = 0; h->prof.step_launches = 0; h->prof.step_used = 0;
  h->prof.shapes.clear(); h->prof.recs.clear();
  hipEvent_t t0 = nullptr, t1 = nullptr;
  if (profiling) {
    PRG_HIP(hipEventCreate(&t0));
    PRG_HIP(hipEventCreate(&t1));
    PRG_HIP(hipEventRecord(t0, s));
  }
  if (h->use_graph && !profiling && !u->taps_on) {
    const bool stale = !h->exec || h->g_cond != img_cond || h->g_noise != noise || h->g_out != out || h->g_stream != s ||
                       h->g_arena_gen != u->arena_gen;
    if (stale) {
      sampler_free(h);
      PRG_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      rc = sampler_one_step(h, img_cond, noise, out, s);
      hipGraph_t g = nullptr;
      hipError_t ce = hipStreamEndCapture(s, &g);
      if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
      if (ce != hipSuccess) return fail(PRG_E_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
      h->graph = g;
      PRG_HIP(hipGraphInstantiate(&h->exec, h->graph, nullptr, nullptr, 0));
      h->g_cond = img_cond; h->g_noise = noise; h->g_out = out; h->g_stream = s; h->g_arena_gen = u->arena_gen;
    }
    for (int k = 0; k < h->n_steps; ++k) PRG_HIP(hipGraphLaunch(h->exec, s));
  } else {
    for (int k = 0; k < h->n_steps; ++k) {
      if ((rc = sampler_one_step(h, img_cond, noise, out, s))) { u->prof = nullptr; return rc; }
      if (profiling) {  // harvest this step's conv events (keeps the pool small)
        PRG_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < h->prof.used; ++i) {
          float ms = 0;
          PRG_HIP(hipEventElapsedTime(&ms, h->prof.pool[i].first, h->prof.pool[i].second));
          h->prof.conv_ms += ms;
          if (i < h->prof.recs.size()) h->prof.add_shape(h->prof.recs[i], ms);
        }
        h->prof.used = 0;
        for (size_t i = 0; i < h->prof.step_used; ++i) {
          float ms = 0;
          PRG_HIP(hipEventElapsedTime(&ms, h->prof.step_pool[i].first, h->prof.step_pool[i].second));
          h->prof.step_ms += ms;
        }
        h->prof.step_used = 0;
      }
    }
  }
  u->prof = nullptr;
  if (profiling) {
    PRG_HIP(hipEventRecord(t1, s));
    PRG_HIP(hipEventSynchronize(t1));
    float ms = 0;
    PRG_HIP(hipEventElapsedTime(&ms, t0, t1));
    h->last_total_ms = ms;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
  }
  return PRG_OK;
}

}  // extern "C"
