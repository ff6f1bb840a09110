// conv.hip — convolutions of the two U-Nets as implicit GEMM on the CDNA4 matrix cores (MFMA roofline).
//
//   out[m][n] = bias[n] + sum_{tap,c} in[pixel(m) + tap][c] * w[n][tap][c]        m = (b, oy, ox), NHWC
//
// One kernel covers every MFMA-shaped conv on the path (sd:592-616, 681-796, 864-918): 3x3 (Block proj, the last
// down/up convs), 4x4 stride 2 (Downsample), 1x1 (res_conv, to_qkv, to_out), the x2 nearest Upsample folded into
// the gather, and the skip concat as two source pointers.  T = bf16_t uses v_mfma_f32_32x32x16_bf16 (fp32
// accumulate); T = float uses v_mfma_f32_32x32x2_f32, which is an exact k-ordered fmaf chain — the parity mode.
//
// Tiling: a 256-thread workgroup (4 waves) owns a BM x BN output tile; the main loop walks (tap, 64-byte channel
// chunk).  Both operands are staged global -> registers -> LDS (register staging lets the gather zero-fill the
// padding halo and select the concat source per 16-byte vector), with the next chunk's global loads issued
// before the MFMAs of the current one and a double-buffered LDS image, so there is one barrier per chunk.
// LDS rows are padded by 16 B: the 32 rows a wave reads with one ds_read_b128 fall on distinct bank slots.
#include "conv.h"

namespace prg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <typename T>
struct Mma;

template <>
struct Mma<bf16_t> {
  static constexpr int KSTEP = 16;
  using Frag = bf16x8;
  __device__ static inline Frag load(const bf16_t* row_base, int kk, int hi) {
    return *reinterpret_cast<const Frag*>(row_base + kk * 16 + hi * 8);
  }
  __device__ static inline f32x16 mma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

template <>
struct Mma<float> {
  static constexpr int KSTEP = 2;
  using Frag = float;
  __device__ static inline Frag load(const float* row_base, int kk, int hi) { return row_base[kk * 2 + hi]; }
  __device__ static inline f32x16 mma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvLaunch<T> L, const int M, const int tiles_m,
                                                         const int tiles_n) {
  constexpr int BK = ConvTile<T>::BK;
  constexpr int VEC = Elem<T>::kVec;
  constexpr int VPR = BK / VEC;              // 16-byte vectors per tile row (4)
  constexpr int BKP = BK + VEC;              // row pitch: +16 B
  constexpr int RPP = 256 / VPR;             // rows staged per pass (64)
  constexpr int AP = BM / RPP, BP = BN / RPP;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* As = reinterpret_cast<T*>(smem);        // [2][BM][BKP]
  T* Bs = As + 2 * BM * BKP;                 // [2][BN][BKP]

  const ConvDesc& d = L.d;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed); give each XCD a contiguous run of tiles so
  // the tiles that share input halos / the same A tile meet in one L2.  Bijective for any grid size.
  const int nblk = tiles_m * tiles_n;
  const int q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int tn = lin % tiles_n, tm = lin / tiles_n;

  const int tid = threadIdx.x;
  const int lrow = tid / VPR, kvec = tid % VPR;
  const int Cin = d.C0 + d.C1;
  const int Hl = d.ups ? 2 * d.Hin : d.Hin, Wl = d.ups ? 2 * d.Win : d.Win;
  const int HWo = d.Hout * d.Wout;

  int a_iy0[AP], a_ix0[AP];
  int64_t a_base[AP];
  bool a_ok[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    int m = tm * BM + lrow + i * RPP;
    a_ok[i] = m < M;
    int mm = a_ok[i] ? m : 0;
    int b = mm / HWo, rem = mm - b * HWo;
    int oy = rem / d.Wout, ox = rem - oy * d.Wout;
    a_iy0[i] = oy * d.stride - d.pad;
    a_ix0[i] = ox * d.stride - d.pad;
    a_base[i] = (int64_t)b * d.Hin * d.Win;
  }

  Vec16<T> ra[AP], rb[BP];
  const int ntaps = d.KH * d.KW;
  const int niter = ntaps * d.kchunks;

  auto gload = [&](int tap, int kc) {
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
    const int c = kc * BK + kvec * VEC;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
      bool ok = a_ok[i] && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl && c < Cin;
      if (d.ups) { iy >>= 1; ix >>= 1; }
      if (ok) {
        int64_t pix = a_base[i] + (int64_t)iy * d.Win + ix;
        const T* p = (c < d.C0) ? L.src0 + pix * d.C0 + c : L.src1 + pix * d.C1 + (c - d.C0);
        ra[i] = vec_load(p);
      } else {
        ra[i] = vec_zero<T>();
      }
    }
    const T* wt = L.w + ((size_t)(tap * d.kchunks + kc) * d.CoutPad + (size_t)tn * BN) * BK;
#pragma unroll
    for (int j = 0; j < BP; ++j) rb[j] = vec_load(wt + (size_t)(j * 256 + tid) * VEC);
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, hi = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  int tap = 0, kc = 0;
  gload(0, 0);
  for (int it = 0; it < niter; ++it) {
    const int buf = it & 1;
    T* Ab = As + buf * BM * BKP;
    T* Bb = Bs + buf * BN * BKP;
#pragma unroll
    for (int i = 0; i < AP; ++i) vec_store(Ab + (lrow + i * RPP) * BKP + kvec * VEC, ra[i]);
#pragma unroll
    for (int j = 0; j < BP; ++j) vec_store(Bb + (lrow + j * RPP) * BKP + kvec * VEC, rb[j]);
    __syncthreads();
    if (++kc == d.kchunks) { kc = 0; ++tap; }
    if (it + 1 < niter) gload(tap, kc);   // in flight while the MFMAs below run
#pragma unroll
    for (int kk = 0; kk < BK / Mma<T>::KSTEP; ++kk) {
      typename Mma<T>::Frag fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = Mma<T>::load(Ab + (wm * WM + i * 32 + l31) * BKP, kk, hi);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = Mma<T>::load(Bb + (wn * WN + j * 32 + l31) * BKP, kk, hi);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::mma(fa[i], fb[j], acc[i][j]);
    }
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = tn * BN + wn * WN + j * 32 + l31;
    if (col >= d.Cout) continue;
    const float bv = L.bias ? L.bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = tm * BM + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
        if (row < M) {
          float v = acc[i][j][e] + bv;
          const size_t o = (size_t)row * d.Cout + col;
          if (L.residual) v += Elem<T>::load(L.residual[o]);
          L.out[o] = Elem<T>::store(v);
        }
      }
    }
  }
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_cfg(const ConvLaunch<T>& L, int M, hipStream_t s) {
  constexpr int BK = ConvTile<T>::BK;
  constexpr int BKP = BK + Elem<T>::kVec;
  const int tiles_m = ceil_div(M, BM), tiles_n = L.d.CoutPad / BN;
  const size_t lds = (size_t)2 * (BM + BN) * BKP * sizeof(T);
  conv_igemm_kernel<T, BM, BN, WM, WN><<<dim3(tiles_m * tiles_n), 256, lds, s>>>(L, M, tiles_m, tiles_n);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

template <typename T>
int launch_conv(const ConvLaunch<T>& L, hipStream_t s) {
  const ConvDesc& d = L.d;
  constexpr int VEC = Elem<T>::kVec;
  PRG_CHECK(L.src0 && L.w && L.out, "conv: null pointer");
  PRG_CHECK(d.C0 % VEC == 0 && d.C1 % VEC == 0, "conv: channel counts must be multiples of the 16-byte vector");
  PRG_CHECK(d.C1 == 0 || L.src1, "conv: second source missing");
  PRG_CHECK(d.CoutPad % 64 == 0 && d.CoutPad >= d.Cout, "conv: bad CoutPad");
  const int64_t M64 = (int64_t)d.B * d.Hout * d.Wout;
  PRG_CHECK(M64 > 0 && M64 < (int64_t)1 << 31, "conv: M out of range");
  const int M = (int)M64;
  if (d.CoutPad % 128 == 0) return launch_cfg<T, 128, 128, 64, 64>(L, M, s);
  if (M >= 256 * 64) return launch_cfg<T, 256, 64, 64, 64>(L, M, s);
  return launch_cfg<T, 128, 64, 32, 64>(L, M, s);
}

template int launch_conv<float>(const ConvLaunch<float>&, hipStream_t);
template int launch_conv<bf16_t>(const ConvLaunch<bf16_t>&, hipStream_t);

// ---------------------------------------------------------------------------------------------
// weight packing (host)
// ---------------------------------------------------------------------------------------------
template <typename T>
void pack_conv_weight(const float* w, int Cout, int Cin, int KH, int KW, std::vector<T>& out, int* CoutPad,
                      int* kchunks) {
  constexpr int BK = ConvTile<T>::BK;
  const int cp = (Cout + 63) / 64 * 64;
  const int kcn = (Cin + BK - 1) / BK;
  out.assign((size_t)KH * KW * kcn * cp * BK, Elem<T>::store(0.0f));
  for (int kh = 0; kh < KH; ++kh)
    for (int kw = 0; kw < KW; ++kw)
      for (int kc = 0; kc < kcn; ++kc)
        for (int n = 0; n < Cout; ++n)
          for (int k = 0; k < BK; ++k) {
            int c = kc * BK + k;
            if (c >= Cin) break;
            float v = w[(((size_t)n * Cin + c) * KH + kh) * KW + kw];
            out[((((size_t)(kh * KW + kw)) * kcn + kc) * cp + n) * BK + k] = Elem<T>::store(v);
          }
  *CoutPad = cp;
  *kchunks = kcn;
}
template void pack_conv_weight<float>(const float*, int, int, int, int, std::vector<float>&, int*, int*);
template void pack_conv_weight<bf16_t>(const float*, int, int, int, int, std::vector<bf16_t>&, int*, int*);

// ---------------------------------------------------------------------------------------------
// stem: 7x7 pad 3 direct conv, float32 NCHW in (Cin 1 or 3) -> T NHWC out (sd:824 / dc:822)
// ---------------------------------------------------------------------------------------------
template <typename T, int CIN>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ wk,
                                                        const float* __restrict__ bias, T* __restrict__ out, int H,
                                                        int W, int Cout) {
  constexpr int TS = 16, HALO = 3, TW = TS + 2 * HALO;  // 22
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);          // [CIN][TW][TW]
  float* wl = tile + CIN * TW * TW;                      // [49*CIN][Cout]
  const int b = blockIdx.z, ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS;
  const int tid = threadIdx.x;
  for (int i = tid; i < CIN * TW * TW; i += 256) {
    int c = i / (TW * TW), rem = i - c * TW * TW;
    int yy = ty0 + rem / TW - HALO, xx = tx0 + rem % TW - HALO;
    tile[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[((size_t)b * CIN + c) * H * W + (size_t)yy * W + xx] : 0.0f;
  }
  for (int i = tid; i < 49 * CIN * Cout; i += 256) wl[i] = wk[i];
  __syncthreads();
  const int ly = tid / TS, lx = tid % TS;
  const int oy = ty0 + ly, ox = tx0 + lx;
  if (oy >= H || ox >= W) return;
  T* o = out + (((size_t)b * H + oy) * W + ox) * Cout;
#pragma unroll 1
  for (int co = 0; co < Cout; co += 8) {
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0f;
#pragma unroll 1
    for (int c = 0; c < CIN; ++c)
#pragma unroll 1
      for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) {
          float v = tile[c * TW * TW + (ly + kh) * TW + lx + kw];
          const float* wp = wl + ((kh * 7 + kw) * CIN + c) * Cout + co;
#pragma unroll
          for (int u = 0; u < 8; ++u) acc[u] = fmaf(v, wp[u], acc[u]);
        }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (co + u < Cout) o[co + u] = Elem<T>::store(acc[u] + bias[co + u]);
  }
}

template <typename T>
int launch_stem_conv(const float* x, const float* wk, const float* bias, T* out, int B, int Cin, int H, int W,
                     int Cout, hipStream_t s) {
  PRG_CHECK(Cin == 1 || Cin == 3, "stem conv: Cin must be 1 or 3");
  PRG_CHECK(Cout % 8 == 0, "stem conv: Cout must be a multiple of 8");
  dim3 grid(ceil_div(W, 16), ceil_div(H, 16), B);
  size_t lds = ((size_t)Cin * 22 * 22 + (size_t)49 * Cin * Cout) * sizeof(float);
  PRG_CHECK(lds <= 64 * 1024, "stem conv: weights do not fit LDS");
  if (Cin == 1)
    stem_conv_kernel<T, 1><<<grid, 256, lds, s>>>(x, wk, bias, out, H, W, Cout);
  else
    stem_conv_kernel<T, 3><<<grid, 256, lds, s>>>(x, wk, bias, out, H, W, Cout);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}
template int launch_stem_conv<float>(const float*, const float*, const float*, float*, int, int, int, int, int,
                                     hipStream_t);
template int launch_stem_conv<bf16_t>(const float*, const float*, const float*, bf16_t*, int, int, int, int, int,
                                      hipStream_t);

// ---------------------------------------------------------------------------------------------
// head: 1x1 conv to one channel (+ sigmoid), T NHWC -> float32 (sd:918 / dc:868-869)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void head_conv_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                 float* __restrict__ out, int64_t M, int C, int sigmoid) {
  constexpr int VEC = Elem<T>::kVec;
  // LP lanes cooperate on one pixel: each takes a 16-byte vector, then a shuffle tree sums the partials
  const int LP = C / VEC >= 64 ? 64 : (C / VEC >= 32 ? 32 : (C / VEC >= 16 ? 16 : (C / VEC >= 8 ? 8 : (C / VEC >= 4 ? 4 : (C / VEC >= 2 ? 2 : 1)))));
  const int per_block = 256 / LP;
  const int sub = threadIdx.x % LP;
  for (int64_t m = (int64_t)blockIdx.x * per_block + threadIdx.x / LP; m < M; m += (int64_t)gridDim.x * per_block) {
    float acc = 0.0f;
    for (int c = sub * VEC; c < C; c += LP * VEC) {
      Vec16<T> v = vec_load(x + m * C + c);
#pragma unroll
      for (int u = 0; u < VEC; ++u) acc = fmaf(Elem<T>::load(v.e[u]), w[c + u], acc);
    }
    for (int o = LP >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (sub == 0) {
      float v = acc + bias[0];
      out[m] = sigmoid ? sigmoid_f(v) : v;
    }
  }
}

template <typename T>
int launch_head_conv(const T* x, const float* w, const float* bias, float* out, int64_t M, int C, int sigmoid,
                     hipStream_t s) {
  PRG_CHECK(C % Elem<T>::kVec == 0, "head conv: C must be a multiple of the vector width");
  int grid = (int)((M + 31) / 32);
  if (grid > 8192) grid = 8192;
  head_conv_kernel<T><<<grid, 256, 0, s>>>(x, w, bias, out, M, C, sigmoid);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}
template int launch_head_conv<float>(const float*, const float*, const float*, float*, int64_t, int, int, hipStream_t);
template int launch_head_conv<bf16_t>(const bf16_t*, const float*, const float*, float*, int64_t, int, int,
                                      hipStream_t);

}  // namespace prg
