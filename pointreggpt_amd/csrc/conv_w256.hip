// conv_w256.hip — wave-specialised 3x3 convolution with a 256-pixel x 128-channel workgroup tile (bf16 throughput path).
//
// The 128-pixel tiles of conv_ws.hip stage one 16 KB weight tile and meet one workgroup barrier per 512 MFMA cycles, and
// their consumers read one LDS fragment per MFMA; measured, a third of that kernel's time is interference between the two
// wave classes.  Here every ratio is halved: the workgroup (512 threads, two waves per SIMD, up to 256
// VGPRs) computes 256 pixels x 128 output channels per tile,
//
//   waves 0-3  CONSUMERS (2 pixel halves x 2 channel halves): a wave owns 128 pixels x 64 channels = eight 32x32
//              accumulators (128 VGPRs); per tap and 16-channel k-step it reads 4 pixel + 2 weight fragments for 8 MFMAs
//              (0.75 ds_read_b128 per MFMA), prefetched one call (8 MFMAs) ahead, across the phase barrier included.
//              A phase (one tap of one 64-channel chunk) is 32 MFMAs = 1024 cycles per wave.  Tile end: bias, GroupNorm
//              partial sums (the halving butterfly of conv_ws.hip), bf16, and DIRECT stores — a v_permlane32_swap pairs
//              the two lane halves' channel quads, so every lane stores 16 contiguous bytes; no LDS stage, no drain;
//   waves 4-7  PRODUCERS: weight tile ph+2 from registers into a 3-slot LDS ring and the loads of tile ph+5 (three
//              register sets, three phases = ~3000 cycles between a load and its use); the halo of step s+1 written
//              during step s from registers loaded a whole step earlier (optional fused GroupNorm + (scale+1, shift) +
//              SiLU prologue of the previous Block, sd:690-696), each register re-issued at once for halo s+2.
//              Plain loads: a step is straight-line code (no branch between a load and its use), so the compiler's
//              s_waitcnt vmcnt(N) counts are exact; the barrier waits for LDS only, never for VMEM.
//
// LDS: 2 halos x 352 rows x 144 B (padded rows: conflict-free ds_read_b128, immediate tap offsets; 12 spare rows take the
// writes of the units past the halo end) + 3 weight slots x 128 rows x 144 B + bias = 157 KB, one workgroup per CU.
// Covers the 3x3 / stride 1 / pad 1 convs with Cout % 128 == 0 and 64-channel sources whose launch has at least one tile
// per CU: tiles of 8 x 32 pixels (image widths 32, 64, 128, ...) or 16 x 16 (the 16 x 16 level).
//
// MODE 1 — Downsample, Conv2d(C, Cout, 4, stride 2, pad 1) (sd:596-597), as the same kernel: with the input shifted by
// (1, 1) the 4 x 4 window of output pixel (oy, ox) is exactly the 2 x 2 block of 2 x 2-pixel blocks (oy + {0,1},
// ox + {0,1}), i.e. a 2 x 2-tap convolution over the space-to-depth view [H/2][W/2][4 C] of the input.  Nothing is
// rearranged in memory: a 64-channel chunk of the 4 C virtual channels is one sub-pixel (dy, dx) of every block, so the
// producers gather "virtual pixel" (y', x') of chunk (dy, dx) from source pixel (2 y' + dy - 1, 2 x' + dx - 1) — a
// constant stride of two source pixels — and the step has four taps ((1,1), (1,2), (2,1), (2,2) of the 3 x 3 frame)
// instead of nine; the weights are the standard 3 x 3 packing of the equivalent [Cout][4 C][3][3] tensor.  Cout = 64
// runs with the second channel half of the consumers idle (the shape is HBM-bound).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "conv.h"

namespace prg {

typedef __attribute__((ext_vector_type(8))) __bf16 w2_bf16x8;
typedef __attribute__((ext_vector_type(16))) float w2_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int w2_u32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 w2_f16x8;

namespace {

// Timing experiments only (results become garbage): 1 consumers skip reads + MFMAs, 2 producers skip the halo, 4 no
// epilogue, 8 producers skip the weights, 16 no prologue arithmetic.
#ifndef PRG_W256_EXP
#define PRG_W256_EXP 0
#endif

constexpr int kCH = 64, BN = 128, ROWB = 144;

template <int TW>
struct W2Geom {
  static constexpr int TH = 256 / TW, HP = TW + 2, HALO = (TH + 2) * HP;
  static constexpr int NPT = 256, RPP = NPT / 8;             // producer threads; halo rows per pass
  static constexpr int KU = (HALO + RPP - 1) / RPP;          // halo units per producer thread
  static constexpr int HROWS = KU * RPP;                     // LDS rows per halo buffer (units past HALO land in the spare rows)
  static constexpr size_t AH_BYTES = (size_t)HROWS * ROWB;
  static constexpr size_t BW_BYTES = (size_t)BN * ROWB;
  static constexpr size_t LDS = 2 * AH_BYTES + 3 * BW_BYTES + BN * sizeof(float);
  static_assert(KU == 11, "unit schedule below");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

__device__ inline float w2_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ inline float w2_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ inline uint32_t w2_pack(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ inline float w2_silu(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
template <int CTRL>
__device__ inline float w2_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int XOR>
__device__ inline float w2_swz(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (XOR << 10) | 0x1F));
}

// barrier that never waits for VMEM; LDS_DONE: this wave's LDS writes are complete first (producers)
template <bool LDS_DONE>
__device__ __forceinline__ void w2_barrier() {
  if constexpr (LDS_DONE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  else asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Tile schedule of one workgroup (workgroup b runs on XCD b % 8: observed placement, speed only).  With several output
// channel tiles every XCD is pinned to ONE of them: its CUs stream the same weight slice, which stays in the XCD's L2.
struct W2Tiles {
  int tiles_x, tiles_y, first, stride, count, tn, psh;
  // psh = 2 (MODE 2, the Upsample conv as four 2 x 2-tap sub-pixel convolutions): the two low bits of a tile index are the
  // sub-pixel phase (dy, dx) — the four phases of one source tile run next to each other and share its halo in the L2
  __device__ __forceinline__ void init(int bid, int GR, int tx, int ty, int tiles_n, int nb, int phase_bits = 0) {
    tiles_x = tx;
    tiles_y = ty;
    psh = phase_bits;
    const int npix = (tx * ty * nb) << phase_bits, per_xcd = GR >> 3, xcd = bid & 7, idx = bid >> 3;
    tn = xcd % tiles_n;
    const int gx = 8 / tiles_n, member = xcd / tiles_n;
    first = member * per_xcd + idx;
    stride = gx * per_xcd;
    count = first < npix ? (npix - first + stride - 1) / stride : 0;
  }
  __device__ __forceinline__ int phase(int it) const { return (first + it * stride) & ((1 << psh) - 1); }
  __device__ __forceinline__ void decode(int it, int& b, int& y0, int& x0, int TH, int TW) const {
    int t = (first + it * stride) >> psh;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    b = t / tiles_y;
    y0 = ty * TH;
    x0 = tx * TW;
  }
};

// phase p of a step: LDS row offset of its tap inside the halo, and the tap's index in the packed 3 x 3 weights
template <int MODE, int HP>
__device__ constexpr int w2_toff(int p) { return MODE == 1 ? (1 + p / 2) * HP + 1 + p % 2 : MODE == 2 ? (p / 2) * HP + p % 2 : (p / 3) * HP + p % 3; }
template <int MODE>
__device__ constexpr int w2_tapid(int p) { return MODE == 1 ? (1 + p / 2) * 3 + 1 + p % 2 : p; }

// PRO 0: no prologue, 1: coefficient tables, 2: coefficients folded in-kernel (pro_fold), 3: as 2 on an f16 input with f16
// weights (the h16 format of conv.h: packed-f16 prologue, v_mfma_f32_32x32x16_f16).  L.out_f16: f16 output (run-time flag).
// MODE 0: 3 x 3 / stride 1; 1: Downsample (4 x 4 / stride 2 as 2 x 2 taps over the space-to-depth view); 2 (round 4): Upsample —
// nn.Upsample(x2, nearest) + Conv2d(3, pad 1) (sd:592-594) as FOUR 2 x 2-tap convolutions of the SOURCE image, one per output
// sub-pixel (dy, dx): output row 2y + dy reads upsampled rows 2y + dy - 1 .. + 1 = source rows {y - 1, y, y} (dy = 0) or
// {y, y, y + 1} (dy = 1), so its three kernel rows collapse onto two source rows with weights (w0, w1 + w2) / (w0 + w1, w2); the
// same in x.  4 instead of 9 taps per output: 2.25x fewer MACs, and the zero padding of the upsampled image is exactly the zero
// padding of the source image.  Tiles are 256 SOURCE pixels x one phase; L.w_up holds the four pre-summed 2 x 2 packings.
template <int TW, int PRO, int MODE>
__global__ __launch_bounds__(512) void conv3x3_w256_kernel(const ConvLaunch<bf16_t> L, const int tiles_x, const int tiles_y,
                                                           const int tiles_n, const int fuse_stats) {
  using G = W2Geom<TW>;
  constexpr int TH = G::TH, HP = G::HP, KU = G::KU, RPP = G::RPP;
  constexpr int AH = (int)G::AH_BYTES, BW = (int)G::BW_BYTES;
  // phases (taps) per step; steps per loop iteration: the weight ring slot / register set of global tile NPH g + p is
  // (NPH g + p) % 3, a compile-time value once g % UNR is one (9 = 0, 4 = 1 mod 3)
  constexpr int NPH = MODE ? 4 : 9, UNR = MODE ? 3 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const bias_lds = reinterpret_cast<float*>(smem + 2 * G::AH_BYTES + 3 * G::BW_BYTES);
  const ConvDesc& d = L.d;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nchunks = MODE == 1 ? 4 * d.C0 / kCH : (d.C0 + d.C1) / kCH;
  // MODE 2 with Cout = 64 ("dual"): the two channel halves of the consumers take the two x-phases (dy, 0) / (dy, 1) of the SAME
  // source tile — 2 x 64 virtual output channels on one halo — instead of one half idling; a tile index then carries dy only
  const bool dual = MODE == 2 && d.Cout == 64;
  const int nwn = (d.Cout >= BN || dual) ? 2 : 1;           // active channel halves (Cout = 64: the second half idles)
  W2Tiles tm;
  tm.init((int)blockIdx.x, (int)gridDim.x, tiles_x, tiles_y, tiles_n, d.B, MODE == 2 ? (dual ? 1 : 2) : 0);
  const int nsteps = tm.count * nchunks;
  if (nsteps == 0) return;
  const int nsteps_pad = (nsteps + UNR - 1) / UNR * UNR;    // the producers' loop body covers UNR steps (clamped reloads past the end)

  // ---------------------------------------------------------------------------------------------------
  if (wave < 4) {
    __builtin_amdgcn_s_setprio(3);
    const int wm = wave >> 1, wn = wave & 1;                // pixel half, channel half of the tile
    const int l31 = lane & 31, hi = lane >> 5;
    // pixel of the wave's 32-pixel group `pt` that this lane owns.  16-pixel tile rows: a group spans two halo rows; the
    // second row's columns are rotated by two so that every ds_read_b128 lane group stays bank-conflict free (conv_ws.hip).
    const int lpx = (TW == 16 && l31 >= 16) ? 16 + ((l31 - 2) & 15) : l31;
    constexpr int GROWS = 32 / TW;                          // tile rows per 32-pixel group
    const int prow = wm * (TH / 2) + (TW == 16 ? (lpx >> 4) : 0), pcol = TW == 16 ? (lpx & 15) : lpx;
    constexpr int PTB = GROWS * HP * ROWB;                  // LDS bytes between the wave's pixel groups
    const char* xa = smem + (prow * HP + pcol) * ROWB + hi * 16;
    const char* xn = xa + AH;
    // MODE 2: the 2 x 2 window of phase (dy, dx) starts at halo position (dy, dx): the phase's byte offset rides on xa / xn
    // (pho_a / pho_n = the offsets they currently carry; step s + 2's tile is tracked by (chunk2, it2))
    auto pho_of = [&](int tile) {
      const int ph = tm.phase(tile);
      return dual ? (ph * HP + wn) * ROWB : ((ph >> 1) * HP + (ph & 1)) * ROWB;
    };
    int pho_a = 0, pho_n = 0, chunk2 = 0, it2 = 0;
    if constexpr (MODE == 2) {
      pho_a = pho_of(0);
      pho_n = nchunks > 1 ? pho_a : pho_of(1);
      xa += pho_a;
      xn += pho_n;
      chunk2 = 2 % nchunks;
      it2 = 2 / nchunks;
    }
    const char* const wr = smem + 2 * AH + (wn * 64 + l31) * ROWB + hi * 16;
    if (dual) { if (tid < BN) bias_lds[tid] = L.bias[tid & 63]; }
    else if (tid < BN && tid < d.Cout) bias_lds[tid] = L.bias[tm.tn * BN + tid];
    const int gn_per = fuse_stats ? (d.Cout / L.gn_groups) >> 3 : 1;   // 8-channel chunks per GroupNorm group
    const int gn_per_sh = 31 - __builtin_clz(gn_per);
    if (wn >= nwn) {                                         // idle channel half: only the barriers
      for (int i = 0; i < 1 + nsteps_pad * NPH; ++i) w2_barrier<false>();
      return;
    }
    w2_f32x16 acc[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ct][pt][e] = 0.0f;
    w2_bf16x8 fw[2][2], fx[2][4];
#define W2_LW(SET, CT, RING, CALL) fw[SET][CT] = *reinterpret_cast<const w2_bf16x8*>(wr + (RING) * BW + (CT) * 32 * ROWB + (CALL) * 32)
#define W2_LX(SET, PT, BASE, TOFF, CALL) fx[SET][PT] = *reinterpret_cast<const w2_bf16x8*>(BASE + (PT) * PTB + (TOFF) * ROWB + (CALL) * 32)
#define W2_MM(SET, CT, PT)                                                                                                              \
  do {                                                                                                                                    \
    if constexpr (PRO == 3)                                                                                                               \
      acc[CT][PT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(w2_f16x8, fw[SET][CT]), __builtin_bit_cast(w2_f16x8, fx[SET][PT]), \
                                                           acc[CT][PT], 0, 0, 0);                                                         \
    else                                                                                                                                  \
      acc[CT][PT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[SET][CT], fx[SET][PT], acc[CT][PT], 0, 0, 0);                              \
  } while (0)
#define W2_SB() __builtin_amdgcn_sched_barrier(0)
    w2_barrier<true>();                                      // halo 0 and weight tiles 0, 1 are in LDS; the bias too
    // Round 5: the accumulators START at the bias (here, and again at the end of every tile's epilogue, where the bias registers are
    // live anyway) instead of at zero: the epilogue's 128 bias additions per tile and wave are gone; the sum's rounding order changes
    // (bias first), far below the bf16 / f16 output rounding.  PRG_W256_EXP & 128: the old form (A/B builds).
    if constexpr (!(PRG_W256_EXP & 128)) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + wn * 64 + ct * 32 + 8 * q + 4 * hi);
#pragma unroll
          for (int pt = 0; pt < 4; ++pt) { acc[ct][pt][4 * q] = b4.x; acc[ct][pt][4 * q + 1] = b4.y; acc[ct][pt][4 * q + 2] = b4.z; acc[ct][pt][4 * q + 3] = b4.w; }
        }
    }
    {
      constexpr int t0 = w2_toff<MODE, HP>(0);
      W2_LW(0, 0, 0, 0); W2_LX(0, 0, xa, t0, 0); W2_LX(0, 1, xa, t0, 0); W2_LX(0, 2, xa, t0, 0); W2_LX(0, 3, xa, t0, 0); W2_LW(0, 1, 0, 0);
    }
    int chunk = 0, it = 0;
    // one step (NPH taps of one 64-channel chunk); GP = g % UNR fixes the ring slots at compile time
    auto cstep = [&](auto GPc) {
      constexpr int GP = decltype(GPc)::value;
      const bool tile_end = chunk == nchunks - 1;
#pragma unroll
      for (int p = 0; p < NPH; ++p) {
        constexpr int dummy = 0;
        (void)dummy;
        const int ring = (GP * NPH + p) % 3, ringN = (ring + 1) % 3;
        const int toff = w2_toff<MODE, HP>(p);
        const int toffN = w2_toff<MODE, HP>(p == NPH - 1 ? 0 : p + 1);
#pragma unroll
        for (int call = 0; call < ((PRG_W256_EXP & 1) ? 0 : 4); ++call) {
          const int cur = call & 1, nxt = cur ^ 1;           // 4 NPH calls per step (even): the set parity is the call parity
          // the next call's six fragments, one load between two MFMAs, earliest-needed first
          if (call < 3) {
            W2_LW(nxt, 0, ring, call + 1); W2_MM(cur, 0, 0); W2_SB();
            W2_LX(nxt, 0, xa, toff, call + 1); W2_MM(cur, 0, 1); W2_SB();
            W2_LX(nxt, 1, xa, toff, call + 1); W2_MM(cur, 0, 2); W2_SB();
            W2_LX(nxt, 2, xa, toff, call + 1); W2_MM(cur, 0, 3); W2_SB();
            W2_LX(nxt, 3, xa, toff, call + 1); W2_MM(cur, 1, 0); W2_SB();
            W2_LW(nxt, 1, ring, call + 1); W2_MM(cur, 1, 1); W2_SB();
          } else if (p < NPH - 1) {                          // next tap: its weight tile is in the next ring slot since the last barrier
            W2_LW(nxt, 0, ringN, 0); W2_MM(cur, 0, 0); W2_SB();
            W2_LX(nxt, 0, xa, toffN, 0); W2_MM(cur, 0, 1); W2_SB();
            W2_LX(nxt, 1, xa, toffN, 0); W2_MM(cur, 0, 2); W2_SB();
            W2_LX(nxt, 2, xa, toffN, 0); W2_MM(cur, 0, 3); W2_SB();
            W2_LX(nxt, 3, xa, toffN, 0); W2_MM(cur, 1, 0); W2_SB();
            W2_LW(nxt, 1, ringN, 0); W2_MM(cur, 1, 1); W2_SB();
          } else {                                           // next step: the other halo buffer (complete since the previous barrier)
            W2_LW(nxt, 0, ringN, 0); W2_MM(cur, 0, 0); W2_SB();
            W2_LX(nxt, 0, xn, toffN, 0); W2_MM(cur, 0, 1); W2_SB();
            W2_LX(nxt, 1, xn, toffN, 0); W2_MM(cur, 0, 2); W2_SB();
            W2_LX(nxt, 2, xn, toffN, 0); W2_MM(cur, 0, 3); W2_SB();
            W2_LX(nxt, 3, xn, toffN, 0); W2_MM(cur, 1, 0); W2_SB();
            W2_LW(nxt, 1, ringN, 0); W2_MM(cur, 1, 1); W2_SB();
          }
          W2_MM(cur, 1, 2);
          W2_MM(cur, 1, 3);
          W2_SB();
        }
        if (p == NPH - 1 && tile_end) {
          // tile finished.  Lane holds pixel (group pt, lpx), channels ct*32 + 8q + 4hi + {0..3} of the wave's 64.
          int tb, ty0, tx0;
          tm.decode(it, tb, ty0, tx0, TH, TW);
          if (PRG_W256_EXP & 4) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
              for (int pt = 0; pt < 4; ++pt) asm volatile("" ::"v"(acc[ct][pt]));
          } else {
            const int oph = MODE == 2 ? tm.phase(it) : 0, osc = MODE == 2 ? 2 : 1;   // MODE 2: output pixel (2 y + dy, 2 x + dx)
            const int ody = dual ? oph : oph >> 1, odx = dual ? wn : oph & 1, och = dual ? 0 : tm.tn * BN + wn * 64;
            char* const obase = reinterpret_cast<char*>(L.out) +
                                ((((size_t)tb * d.Hout + osc * (ty0 + prow) + ody) * d.Wout + osc * (tx0 + pcol) + odx) * d.Cout + och + 8 * hi) * 2;
            const size_t optb = (size_t)GROWS * osc * d.Wout * d.Cout * 2;   // bytes between the wave's pixel groups
            float V[16];                                     // [sum | sum of squares][ct][q]
            const bool o16 = L.out_f16 != 0;                  // (wave-uniform; MODE 1 = Downsample never stores f16)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
              float bv[4][4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + wn * 64 + ct * 32 + 8 * q + 4 * hi);
                bv[q][0] = b4.x; bv[q][1] = b4.y; bv[q][2] = b4.z; bv[q][3] = b4.w;
                V[ct * 4 + q] = 0.0f;
                V[8 + ct * 4 + q] = 0.0f;
              }
#pragma unroll
              for (int pt = 0; pt < 4; ++pt) {
                uint32_t pk[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  float v[4];
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    if constexpr ((PRG_W256_EXP & 128) != 0) {
                      v[r] = acc[ct][pt][4 * q + r] + bv[q][r];
                      acc[ct][pt][4 * q + r] = 0.0f;
                    } else {
                      v[r] = acc[ct][pt][4 * q + r];
                      acc[ct][pt][4 * q + r] = bv[q][r];          // the next tile's initial value
                    }
                    V[ct * 4 + q] += v[r];
                    V[8 + ct * 4 + q] = fmaf(v[r], v[r], V[8 + ct * 4 + q]);
                  }
                  if (!MODE && o16) {
                    pk[2 * q] = h16_pack(v[0], v[1]);
                    pk[2 * q + 1] = h16_pack(v[2], v[3]);
                  } else {
                    pk[2 * q] = w2_pack(v[0], v[1]);
                    pk[2 * q + 1] = w2_pack(v[2], v[3]);
                  }
                }
                // lanes l and l + 32 hold the two channel quads of the same pixel and 8-channel chunk q: swapping the upper
                // half of chunk 2m with the lower half of chunk 2m+1 leaves lane half 0 with all 8 channels of chunk 2m and
                // half 1 with those of chunk 2m+1 — one 16-byte store each.
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                  const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * m], pk[4 * m + 2], false, false);
                  const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * m + 1], pk[4 * m + 3], false, false);
                  const w2_u32x4 o = {(uint32_t)s0[0], (uint32_t)s1[0], (uint32_t)s0[1], (uint32_t)s1[1]};
                  *reinterpret_cast<w2_u32x4*>(obase + pt * optb + ct * 64 + m * 32) = o;
                }
              }
            }
            if (fuse_stats) {
              // 16 full-wave sums with 17 lane exchanges (the halving butterfly of conv_ws.hip): fixed order, deterministic
              const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
              float A8[8], B4[4], C2[2];
#pragma unroll
              for (int j = 0; j < 8; ++j) A8[j] = (b0 ? V[8 + j] : V[j]) + w2_dpp<0xB1>(b0 ? V[j] : V[8 + j]);          // lane ^ 1
#pragma unroll
              for (int j = 0; j < 4; ++j) B4[j] = (b1 ? A8[4 + j] : A8[j]) + w2_dpp<0x4E>(b1 ? A8[j] : A8[4 + j]);    // lane ^ 2
#pragma unroll
              for (int j = 0; j < 2; ++j) C2[j] = (b2 ? B4[2 + j] : B4[j]) + w2_swz<4>(b2 ? B4[j] : B4[2 + j]);
              float D = (b3 ? C2[1] : C2[0]) + w2_swz<8>(b3 ? C2[0] : C2[1]);
              D += w2_swz<16>(D);
              D += __shfl_xor(D, 32, 64);
              // lane (< 16) holds the wave total of value i = 8 b0 + 4 b1 + 2 b2 + b3 = [sq][ct][q]
              if (gn_per >= 2) D += w2_swz<8>(D);
              if (gn_per >= 4) D += w2_swz<4>(D);
              if (gn_per >= 8) D += w2_dpp<0x4E>(D);
              const int i = (lane & 1) * 8 + (lane & 2) * 2 + ((lane >> 2) & 1) * 2 + ((lane >> 3) & 1);
              const int cc = i & 7;
              if (lane < 16 && (cc & (gn_per - 1)) == 0) {
                const int nsplit = tiles_x * tiles_y * 2;
                const int slab = ((ty0 / TH) * tiles_x + tx0 / TW) * 2 + wm;
                const int grp = (((tm.tn * BN + wn * 64) >> 3) + cc) >> gn_per_sh;
                if (L.gn_acc) gn_acc_add(L.gn_acc, L.gn_groups, tb, grp, i >> 3, D);   // fixed-point accumulators (common.h)
                else L.gn_partials[(((size_t)tb * nsplit + slab) * L.gn_groups + grp) * 2 + (i >> 3)] = D;
              }
            }
          }
        }
        w2_barrier<false>();   // every LDS read of this phase has been consumed by an MFMA above
      }
      { const char* t = xa; xa = xn; xn = t; }
      if constexpr (MODE == 2) {                             // xn (the buffer just left) serves step s + 2 next: its tile's phase
        const int t = pho_a; pho_a = pho_n; pho_n = t;
        const int want = pho_of(it2);
        xn += want - pho_n;
        pho_n = want;
        if (++chunk2 == nchunks) { chunk2 = 0; ++it2; }
      }
      if (++chunk == nchunks) {
        chunk = 0;
        ++it;
      }
    };
    for (int g = 0; g < nsteps_pad; g += UNR) {
      cstep(std::integral_constant<int, 0>{});
      if constexpr (UNR == 3) {
        if (g + 1 < nsteps) cstep(std::integral_constant<int, 1>{});
        else for (int i = 0; i < NPH; ++i) w2_barrier<false>();
        if (g + 2 < nsteps) cstep(std::integral_constant<int, 2>{});
        else for (int i = 0; i < NPH; ++i) w2_barrier<false>();
      }
    }
#undef W2_LW
#undef W2_LX
#undef W2_MM
#undef W2_SB
    return;
  }

  // ---------------------------------------------------------------------------------------------------
  {
    const int ptid = tid - 256, slot = ptid & 7, row = ptid >> 3;   // 16-byte unit of a 128-byte row; rows row + 32 k
    char* const Ah0 = smem + row * ROWB + slot * 16;
    char* const Bw0 = smem + 2 * AH + row * ROWB + slot * 16;
    const int Hl = MODE == 2 ? d.Hin : d.Hout, Wl = MODE == 2 ? d.Win : d.Wout;   // (MODE 2 tiles the SOURCE image)
    // tile-independent part of every halo unit's address and validity
    unsigned hpix[KU], hedge[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int hp = k * RPP + row;
      const int hy = hp / HP, hx = hp - hy * HP;
      int ry = hy - 1, rx = hx - 1;                          // tile origins are even: the x2 gather is (origin / 2) + (r >> 1)
      if (!MODE && d.ups) { ry >>= 1; rx >>= 1; }
      hpix[k] = (unsigned)((ry + 1) * d.Win + (rx + 1));     // MODE 1: in units of two source pixels (a virtual row = 2 Win pixels)
      hedge[k] = (hy == 0 ? 1u : 0u) | (hy == TH + 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == TW + 1 ? 8u : 0u) |
                 (hp >= G::HALO ? 16u : 0u) | (hy == 1 ? 32u : 0u) | (hx == 1 ? 64u : 0u);
    }
    const bf16_t* const wbase = MODE == 1 ? L.w_s2d : MODE == 2 ? L.w_up : PRO == 3 ? reinterpret_cast<const bf16_t*>(L.w_f16) : L.w;
    const int wkch = MODE == 1 ? L.s2d_kchunks : d.kchunks;
    // this thread's first unit of a weight tile inside a tap's [2][CoutPad][32] slab; rows row + 32 j are 2048 bytes apart
    const unsigned w_voff = (unsigned)((((slot >> 2) * d.CoutPad + row) * 32 + (slot & 3) * 8) * 2);
    const int wj_mask = (nwn == 2 && !dual) ? 3 : 1;         // Cout = 64: only 64 weight rows exist (the upper ones are re-read)
    const size_t wsub = (size_t)4 * wkch * d.CoutPad * 64;   // dual: bytes between the packings of two phases (rows 64-127 = phase dx = 1)
    struct StepInfo { int chunk, it, b, y0, x0, ph; };
    int gC = 0;
    auto advance = [&](StepInfo& si) {                       // past the last step it stays there: harmless reloads
      if (gC + 1 < nsteps) {
        ++gC;
        if (++si.chunk == nchunks) {
          si.chunk = 0;
          ++si.it;
          tm.decode(si.it, si.b, si.y0, si.x0, TH, TW);
          si.ph = tm.phase(si.it);
        }
      }
    };
    StepInfo sA;
    sA.chunk = 0;
    sA.it = 0;
    tm.decode(0, sA.b, sA.y0, sA.x0, TH, TW);
    sA.ph = tm.phase(0);
    StepInfo sB = sA;
    advance(sB);
    StepInfo sC = sB;
    advance(sC);

    w2_u32x4 wset[3][4], hreg[KU];
    float4 cf[4], nf[4];                                     // a0..3, a4..7, b0..3, b4..7 of the halo being written / issued
    unsigned hvalid = 0, hvalid_nxt = 0;
    const char* ld_base = nullptr;
    __amdgpu_buffer_rsrc_t ld_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(L.src0), 0, 0, 0x00020000);
    unsigned ld_cs2 = 0, ld_tedge = 0;
    const float* ld_ca = nullptr;
    const float* ld_cb = nullptr;
    // pro_fold (common.h, GnFold): ld_ca / ld_cb point at P / Q and the coefficients are folded here from the image group's
    // fixed-point statistics (ld_acc) when a halo's coefficients are adopted: A = rstd P, B = Q - mean A
    constexpr bool pfold = PRO >= 2;
    const long long* ld_acc = nullptr;
    longlong2 nacc = make_longlong2(0, 0);
    h16x2 ah2[4], bh2[4];                                    // PRO == 3: the adopted coefficients as packed f16 channel pairs
    auto fold_inplace = [&](float4* c) {
      if constexpr (PRO) {
        if (pfold) {
          float mean, rstd;
          gn_fold_stats_raw(nacc.x, nacc.y, L.pro_fold.inv_n, mean, rstd);
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            c[h2] = make_float4(rstd * c[h2].x, rstd * c[h2].y, rstd * c[h2].z, rstd * c[h2].w);
            c[2 + h2] = make_float4(fmaf(-mean, c[h2].x, c[2 + h2].x), fmaf(-mean, c[h2].y, c[2 + h2].y),
                                    fmaf(-mean, c[h2].z, c[2 + h2].z), fmaf(-mean, c[h2].w, c[2 + h2].w));
          }
        }
        if constexpr (PRO == 3) {
          typedef __attribute__((ext_vector_type(2))) float f32x2;
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            ah2[2 * h2] = __builtin_convertvector((f32x2){c[h2].x, c[h2].y}, h16x2);
            ah2[2 * h2 + 1] = __builtin_convertvector((f32x2){c[h2].z, c[h2].w}, h16x2);
            bh2[2 * h2] = __builtin_convertvector((f32x2){c[2 + h2].x, c[2 + h2].y}, h16x2);
            bh2[2 * h2 + 1] = __builtin_convertvector((f32x2){c[2 + h2].z, c[2 + h2].w}, h16x2);
          }
        }
      }
    };
    auto issue_setup = [&](const StepInfo& si) {
      if constexpr (MODE == 1) {
        // chunk = sub-pixel (dy, dx) of the 2 x 2 blocks, then 64 of its C0 channels.  Virtual pixel (y', x') of the chunk
        // is source pixel (2 y' + dy - 1, 2 x' + dx - 1): outside the image for y' = 0 with dy = 0 and for y' = Hout
        // with dy = 1 (x alike); halo row / column 0 is never multiplied (its taps carry zero weights) and written as zeros
        const int nsub = d.C0 / kCH, sp = si.chunk / nsub, cc = (si.chunk - sp * nsub) * kCH;
        const int dy = sp >> 1, dx = sp & 1;
        ld_cs2 = (unsigned)(d.C0 * 4);
        ld_tedge = 1u | 4u | 16u | (dy == 0 && si.y0 == 0 ? 32u : 0u) | (dy == 1 && si.y0 + TH == Hl ? 2u : 0u) |
                   (dx == 0 && si.x0 == 0 ? 64u : 0u) | (dx == 1 && si.x0 + TW == Wl ? 8u : 0u);
        const int64_t horg = ((int64_t)si.b * d.Hin + 2 * si.y0 + dy - 3) * d.Win + 2 * si.x0 + dx - 3;   // halo position (0, 0)
        ld_base = reinterpret_cast<const char*>(L.src0) + (horg * d.C0 + cc) * 2;
      } else {
        const int c = si.chunk * kCH;
        const bool first = c < d.C0;
        const bf16_t* src = first ? L.src0 : L.src1;
        const int cs = first ? d.C0 : d.C1;
        const int cc = first ? c : c - d.C0;
        ld_cs2 = (unsigned)(cs * 2);
        ld_tedge = (si.y0 == 0 ? 1u : 0u) | (si.y0 + TH == Hl ? 2u : 0u) | (si.x0 == 0 ? 4u : 0u) | (si.x0 + TW == Wl ? 8u : 0u) | 16u;
        const int sh = MODE == 2 ? 0 : d.ups;                 // (MODE 2: tile origins are source coordinates already)
        const int64_t horg = ((int64_t)si.b * d.Hin + (si.y0 >> sh)) * d.Win + (si.x0 >> sh) - (d.Win + 1);
        ld_base = reinterpret_cast<const char*>(src) + (horg * cs + cc) * 2;
      }
      ld_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ld_base), 0, 0x7ffffff0, 0x00020000);   // (base may lie before the tensor: only in-image offsets are ever in range AND valid)
      if constexpr (PRO) {
        const int c0 = si.chunk * kCH + slot * 8;
        if (pfold) {
          const GnFold& f = L.pro_fold;
          ld_ca = f.P + (size_t)si.b * f.pq_stride + c0;
          ld_cb = f.Q + (size_t)si.b * f.pq_stride + c0;
          ld_acc = f.acc + ((size_t)si.b * f.G + c0 / f.cpg) * 2;
        } else {
          const size_t o = (size_t)si.b * d.C0 + c0;
          ld_ca = L.pro_a + o;
          ld_cb = L.pro_b + o;
        }
      }
      hvalid_nxt = 0;
    };
    auto issue_coeffs = [&](float4* dst) {
      if constexpr (PRO) {
        dst[0] = *reinterpret_cast<const float4*>(ld_ca);
        dst[1] = *reinterpret_cast<const float4*>(ld_ca + 4);
        dst[2] = *reinterpret_cast<const float4*>(ld_cb);
        dst[3] = *reinterpret_cast<const float4*>(ld_cb + 4);
        if (pfold) nacc = *reinterpret_cast<const longlong2*>(ld_acc);
      }
    };
    // Raw BUFFER loads from a descriptor based at the halo origin (round 3): a padding unit gets offset 0xffffffff and the
    // hardware's range check returns zeros — no always-mapped stand-in pixel, and without a prologue no select before the
    // LDS write either (conv_c64.hip's producers, same idea).
    auto issue_unit = [&](int k) {                           // k is a compile-time constant at every call site
      const bool ok = (hedge[k] & ld_tedge) == 0 && !(PRG_W256_EXP & 2);
      const unsigned voff = (__umul24(hpix[k], ld_cs2) + (unsigned)(slot * 16)) | (ok ? 0u : 0xffffffffu);   // (branch-free)
      hreg[k] = __builtin_amdgcn_raw_buffer_load_b128(ld_rsrc, (int)voff, 0, 0);
      if constexpr (PRO) hvalid_nxt |= (ok ? 1u : 0u) << k;
    };
    auto write_unit = [&](int k, int bufoff) {
      w2_u32x4 v = hreg[k];
      if constexpr (PRO == 3 && !(PRG_W256_EXP & 16)) {
        v = h16_silu8(v, ah2, bh2);
      } else if constexpr (PRO && !(PRG_W256_EXP & 16)) {
        const float a8[8] = {cf[0].x, cf[0].y, cf[0].z, cf[0].w, cf[1].x, cf[1].y, cf[1].z, cf[1].w};
        const float b8[8] = {cf[2].x, cf[2].y, cf[2].z, cf[2].w, cf[3].x, cf[3].y, cf[3].z, cf[3].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = w2_silu(fmaf(w2_lo(v[j]), a8[2 * j], b8[2 * j]));
          const float hh = w2_silu(fmaf(w2_hi(v[j]), a8[2 * j + 1], b8[2 * j + 1]));
          v[j] = w2_pack(lo, hh);
        }
      }
      if constexpr (PRO) {                                   // (padding must be zero AFTER the transform; without one it arrived as zeros)
        if (!((hvalid >> k) & 1u)) v = w2_u32x4{0u, 0u, 0u, 0u};
      }
      *reinterpret_cast<w2_u32x4*>(Ah0 + bufoff + k * RPP * ROWB) = v;
    };
    // weight tile of phase `ph` (its tap in the packed 3 x 3 layout) and 64-channel chunk; MODE 2: of sub-pixel `sub`'s 2 x 2 packing
    auto w_tile = [&](int ph, int chunk, int sub) -> const char* {
      const int tap = MODE == 2 ? (dual ? 2 * sub : sub) * 4 + ph : w2_tapid<MODE>(ph);
      return reinterpret_cast<const char*>(wbase) + ((size_t)(tap * wkch + 2 * chunk) * d.CoutPad + tm.tn * BN) * 64 + w_voff;
    };
    auto w_issue = [&](int set, const char* p) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        wset[set][j] = *reinterpret_cast<const w2_u32x4*>(p + ((PRG_W256_EXP & 8) ? 0 : (j & wj_mask) * 2048) + ((dual && j >= 2) ? wsub : (size_t)0));
    };
    auto w_write = [&](int set, int ring) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<w2_u32x4*>(Bw0 + ring * BW + j * 32 * ROWB) = wset[set][j];
    };
    // global weight tile t (t < 3 NPH) belongs to step t / NPH of (sA, sB, sC)
    auto chunk_of = [&](int t) { return t < NPH ? sA.chunk : t < 2 * NPH ? sB.chunk : sC.chunk; };
    auto sub_of = [&](int t) { return MODE == 2 ? (t < NPH ? sA.ph : t < 2 * NPH ? sB.ph : sC.ph) : 0; };

    // ---- prologue: halo 0 and weight tiles 0, 1 into LDS; halo 1 and tiles 2, 3, 4 into registers
    issue_setup(sA);
    issue_coeffs(cf);
    fold_inplace(cf);
#pragma unroll
    for (int k = 0; k < KU; ++k) issue_unit(k);
    hvalid = hvalid_nxt;
    w_issue(0, w_tile(0, chunk_of(0), sub_of(0)));
    w_issue(1, w_tile(1, chunk_of(1), sub_of(1)));
#pragma unroll
    for (int k = 0; k < KU; ++k) write_unit(k, 0);
    w_write(0, 0);
    w_write(1, 1);
    issue_setup(sB);
    issue_coeffs(nf);
#pragma unroll
    for (int k = 0; k < KU; ++k) issue_unit(k);
    hvalid = hvalid_nxt;
#pragma unroll
    for (int j = 0; j < 4; ++j) cf[j] = nf[j];
    fold_inplace(cf);
    w_issue(2, w_tile(2 % NPH, chunk_of(2), sub_of(2)));
    w_issue(0, w_tile(3 % NPH, chunk_of(3), sub_of(3)));
    w_issue(1, w_tile(4 % NPH, chunk_of(4), sub_of(4)));
    issue_setup(sC);
    w2_barrier<true>();

    // units of halo s+1 written (and re-issued for halo s+2) in phases 0 .. NPH-2: 2 2 2 1 1 1 1 1 (nine taps), 4 4 3 (four)
    constexpr int US9[10] = {0, 2, 4, 6, 7, 8, 9, 10, 11, 11};
    constexpr int US4[5] = {0, 4, 8, 11, 11};
    auto pstep = [&](auto GPc, int g) {
      constexpr int GP = decltype(GPc)::value;
      const int bufoff = ((g + 1) & 1) * AH;
#pragma unroll
      for (int p = 0; p < NPH; ++p) {
        const int set = (GP * NPH + p + 2) % 3;
        const int u0 = MODE ? US4[p] : US9[p], u1 = MODE ? US4[p + 1] : US9[p + 1];
        w_write(set, set);                                   // weight tile NPH g + p + 2 (loaded three phases ago)
#pragma unroll
        for (int k = u0; k < u1; ++k) write_unit(k, bufoff);
        if (p == 0) issue_coeffs(nf);                        // coefficients of halo g + 2
        w_issue(set, w_tile((p + 5) % NPH, chunk_of(p + 5), sub_of(p + 5)));   // weight tile NPH g + p + 5: this step's or a later one's chunk
#pragma unroll
        for (int k = u0; k < u1; ++k) issue_unit(k);
        if (p == NPH - 1) {                                  // step bookkeeping (wave-uniform) in the phase without halo work
          hvalid = hvalid_nxt;
#pragma unroll
          for (int j = 0; j < 4; ++j) cf[j] = nf[j];
          fold_inplace(cf);
          sA = sB;
          sB = sC;
          advance(sC);
          issue_setup(sC);
        }
        w2_barrier<true>();
      }
    };
#pragma unroll 1
    for (int g = 0; g < nsteps_pad; g += UNR) {
      pstep(std::integral_constant<int, 0>{}, g);
      if constexpr (UNR == 3) {
        pstep(std::integral_constant<int, 1>{}, g + 1);
        pstep(std::integral_constant<int, 2>{}, g + 2);
      }
    }
  }
}

// =====================================================================================================================
// MX-fp8 operands (BASELINE configs[4]) on the same 256-pixel x 128-channel structure.
// Both MFMA operands are OCP e4m3 with one E8M0 scale per 32 channels; v_mfma_scale_f32_32x32x64_f8f6f4 contracts a whole
// 64-channel chunk of a tap per instruction (64 cycles for four times the MACs of the bf16 instruction's 32: measured
// 3.7-3.9 PFLOP/s on random data against 1.45-1.65 for bf16, tools/micro/mfma_power_fp8.hip).  A phase = 8 MFMAs = 512
// cycles per consumer wave; its six operands (two 16-byte reads + one scale byte each) are prefetched during the previous
// phase.  The producers quantise the bf16 activations while they write the halo (after the optional fused prologue, whose
// result is rounded to bf16 first, like the tensor it replaces): the four lanes that hold one pixel's 32 consecutive channels
// agree on the block maximum with two DPP exchanges, scale = 2^(floor(log2 max) - 8), elements by v_cvt_pk_fp8_f32
// (arithmetic of conv3x3_mx_kernel in conv.hip and of oracle/mx.py).  Weights arrive quantised (pack_conv_weight_mxfp8).
// LDS rows are 64 bytes padded to 80 (conflict-free ds_read_b128), scales sit in byte arrays beside them: 90 KB.
// Operand layout (probed, tools/micro/mx_layout_probe.hip): lane l holds row / column l & 31; its 32 bytes are
// k = 16 h + (0..15) and 32 + 16 h + (0..15), h = l >> 5; its scale covers k in [32 h, 32 h + 32).
// =====================================================================================================================
typedef __attribute__((ext_vector_type(8))) int w2_i32x8;
constexpr int MXROW = 80;

template <int TW>
struct W2MxGeom {
  static constexpr int TH = 256 / TW, HP = TW + 2, HALO = (TH + 2) * HP;
  static constexpr int RPP = 32, KU = (HALO + RPP - 1) / RPP, HROWS = KU * RPP;
  static constexpr int AH = HROWS * MXROW, BW = BN * MXROW, HS = HROWS * 2, WS = BN * 2;
  static constexpr int OFF_B = 2 * AH, OFF_HS = OFF_B + 3 * BW, OFF_WS = OFF_HS + 2 * HS, OFF_BIAS = OFF_WS + 3 * WS;
  static constexpr size_t LDS = OFF_BIAS + BN * sizeof(float);
  static_assert(KU == 11 && HS % 16 == 0 && WS % 16 == 0, "layout");
};

__device__ inline uint32_t w2_cvt4(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}

template <int TW, int PRO>   // PRO as in conv3x3_w256_kernel
__global__ __launch_bounds__(512) void conv3x3_w256mx_kernel(const ConvLaunch<bf16_t> L, const int tiles_x, const int tiles_y,
                                                             const int tiles_n, const int fuse_stats) {
  using G = W2MxGeom<TW>;
  constexpr int TH = G::TH, HP = G::HP, KU = G::KU, RPP = G::RPP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const bias_lds = reinterpret_cast<float*>(smem + G::OFF_BIAS);
  const ConvDesc& d = L.d;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nchunks = (d.C0 + d.C1) / kCH;
  W2Tiles tm;
  tm.init((int)blockIdx.x, (int)gridDim.x, tiles_x, tiles_y, tiles_n, d.B);
  const int nsteps = tm.count * nchunks;
  if (nsteps == 0) return;

  // ---------------------------------------------------------------------------------------------------
  if (wave < 4) {
    __builtin_amdgcn_s_setprio(3);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int lpx = (TW == 16 && l31 >= 16) ? 16 + ((l31 - 2) & 15) : l31;
    constexpr int GROWS = 32 / TW;
    const int prow = wm * (TH / 2) + (TW == 16 ? (lpx >> 4) : 0), pcol = TW == 16 ? (lpx & 15) : lpx;
    constexpr int PTB = GROWS * HP * MXROW, PTS = GROWS * HP * 2;
    const char* xa = smem + (prow * HP + pcol) * MXROW + hi * 16;          // data of halo buffer 0 / 1
    const char* xn = xa + G::AH;
    const char* sa = smem + G::OFF_HS + (prow * HP + pcol) * 2 + hi;       // its block scales
    const char* sn = sa + G::HS;
    const char* const wr = smem + G::OFF_B + (wn * 64 + l31) * MXROW + hi * 16;
    const char* const ws = smem + G::OFF_WS + (wn * 64 + l31) * 2 + hi;
    if (tid < BN) bias_lds[tid] = L.bias[tm.tn * BN + tid];
    const int gn_per = fuse_stats ? (d.Cout / L.gn_groups) >> 3 : 1;
    const int gn_per_sh = 31 - __builtin_clz(gn_per);
    w2_f32x16 acc[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ct][pt][e] = 0.0f;
    // Pixel operands: ONE set, MFMAs in pixel-group-major order, each group's registers reloaded for the next phase right
    // after its second (last) MFMA of this phase: six MFMAs = 384 cycles until the next use.  Weight operands are needed by
    // all eight MFMAs of a phase: TWO sets, the next phase's loaded at the start of this one.  (Two full sets spill.)
    w2_i32x8 fw[2][2], fx[4];
    int sw[2][2], sx[4];
    auto ldw = [&](int set, int ct, int ring) {
      const char* p = wr + ring * G::BW + ct * 32 * MXROW;
      const uint4 lo = *reinterpret_cast<const uint4*>(p), up = *reinterpret_cast<const uint4*>(p + 32);
      fw[set][ct] = w2_i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)up.x, (int)up.y, (int)up.z, (int)up.w};
      sw[set][ct] = *reinterpret_cast<const unsigned char*>(ws + ring * G::WS + ct * 64);
    };
    auto ldx = [&](int pt, const char* bd, const char* bs, int toff) {
      const char* p = bd + pt * PTB + toff * MXROW;
      const uint4 lo = *reinterpret_cast<const uint4*>(p), up = *reinterpret_cast<const uint4*>(p + 32);
      fx[pt] = w2_i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)up.x, (int)up.y, (int)up.z, (int)up.w};
      sx[pt] = *reinterpret_cast<const unsigned char*>(bs + pt * PTS + toff * 2);
    };
#define W2X_MM(SET, CT, PT) \
  acc[CT][PT] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fw[SET][CT], fx[PT], acc[CT][PT], 0, 0, 0, sw[SET][CT], 0, sx[PT])
#define W2X_SB() __builtin_amdgcn_sched_barrier(0)
    w2_barrier<true>();                                      // halo 0, weight tiles 0 and 1, the bias
    ldw(0, 0, 0); ldx(0, xa, sa, 0); ldw(0, 1, 0); ldx(1, xa, sa, 0); ldx(2, xa, sa, 0); ldx(3, xa, sa, 0);
    int chunk = 0, it = 0;
    // nine phases per step: the weight set of phase p alternates with the step parity GP -> two steps per loop iteration
    auto cstep = [&](auto GPc) {
      constexpr int GP = decltype(GPc)::value;
      const bool tile_end = chunk == nchunks - 1;
#pragma unroll
      for (int p = 0; p < 9; ++p) {
        const int cur = (GP + p) & 1, nxt = cur ^ 1;
        const int ringN = (p + 1) % 3;
        const int toffN = p == 8 ? 0 : ((p + 1) / 3) * HP + (p + 1) % 3;
        const char* const bd = p == 8 ? xn : xa;
        const char* const bs = p == 8 ? sn : sa;
        if (!(PRG_W256_EXP & 1)) {
          ldw(nxt, 0, ringN); W2X_MM(cur, 0, 0); W2X_SB();
          ldw(nxt, 1, ringN); W2X_MM(cur, 1, 0); W2X_SB();
          ldx(0, bd, bs, toffN); W2X_MM(cur, 0, 1); W2X_MM(cur, 1, 1); W2X_SB();
          ldx(1, bd, bs, toffN); W2X_MM(cur, 0, 2); W2X_MM(cur, 1, 2); W2X_SB();
          ldx(2, bd, bs, toffN); W2X_MM(cur, 0, 3); W2X_MM(cur, 1, 3); W2X_SB();
          ldx(3, bd, bs, toffN); W2X_SB();
        }
        if (p == 8 && tile_end) {
          // tile finished.  Lane holds pixel (group pt, lpx), channels ct*32 + 8q + 4hi + {0..3} of the wave's 64.
          int tb, ty0, tx0;
          tm.decode(it, tb, ty0, tx0, TH, TW);
          if (PRG_W256_EXP & 4) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
              for (int pt = 0; pt < 4; ++pt) asm volatile("" ::"v"(acc[ct][pt]));
          } else {
            char* const obase = reinterpret_cast<char*>(L.out) +
                                ((((size_t)tb * d.Hout + ty0 + prow) * d.Wout + tx0 + pcol) * d.Cout + tm.tn * BN + wn * 64 + 8 * hi) * 2;
            const size_t optb = (size_t)GROWS * d.Wout * d.Cout * 2;   // bytes between the wave's pixel groups
            float V[16];                                     // [sum | sum of squares][ct][q]
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
              float bv[4][4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + wn * 64 + ct * 32 + 8 * q + 4 * hi);
                bv[q][0] = b4.x; bv[q][1] = b4.y; bv[q][2] = b4.z; bv[q][3] = b4.w;
                V[ct * 4 + q] = 0.0f;
                V[8 + ct * 4 + q] = 0.0f;
              }
#pragma unroll
              for (int pt = 0; pt < 4; ++pt) {
                uint32_t pk[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  float v[4];
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    v[r] = acc[ct][pt][4 * q + r] + bv[q][r];
                    V[ct * 4 + q] += v[r];
                    V[8 + ct * 4 + q] = fmaf(v[r], v[r], V[8 + ct * 4 + q]);
                    acc[ct][pt][4 * q + r] = 0.0f;
                  }
                  pk[2 * q] = w2_pack(v[0], v[1]);
                  pk[2 * q + 1] = w2_pack(v[2], v[3]);
                }
                // lanes l and l + 32 hold the two channel quads of the same pixel and 8-channel chunk q: swapping the upper
                // half of chunk 2m with the lower half of chunk 2m+1 leaves lane half 0 with all 8 channels of chunk 2m and
                // half 1 with those of chunk 2m+1 — one 16-byte store each.
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                  const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * m], pk[4 * m + 2], false, false);
                  const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * m + 1], pk[4 * m + 3], false, false);
                  const w2_u32x4 o = {(uint32_t)s0[0], (uint32_t)s1[0], (uint32_t)s0[1], (uint32_t)s1[1]};
                  *reinterpret_cast<w2_u32x4*>(obase + pt * optb + ct * 64 + m * 32) = o;
                }
              }
            }
            if (fuse_stats) {
              // 16 full-wave sums with 17 lane exchanges (the halving butterfly of conv_ws.hip): fixed order, deterministic
              const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
              float A8[8], B4[4], C2[2];
#pragma unroll
              for (int j = 0; j < 8; ++j) A8[j] = (b0 ? V[8 + j] : V[j]) + w2_dpp<0xB1>(b0 ? V[j] : V[8 + j]);          // lane ^ 1
#pragma unroll
              for (int j = 0; j < 4; ++j) B4[j] = (b1 ? A8[4 + j] : A8[j]) + w2_dpp<0x4E>(b1 ? A8[j] : A8[4 + j]);    // lane ^ 2
#pragma unroll
              for (int j = 0; j < 2; ++j) C2[j] = (b2 ? B4[2 + j] : B4[j]) + w2_swz<4>(b2 ? B4[j] : B4[2 + j]);
              float D = (b3 ? C2[1] : C2[0]) + w2_swz<8>(b3 ? C2[0] : C2[1]);
              D += w2_swz<16>(D);
              D += __shfl_xor(D, 32, 64);
              // lane (< 16) holds the wave total of value i = 8 b0 + 4 b1 + 2 b2 + b3 = [sq][ct][q]
              if (gn_per >= 2) D += w2_swz<8>(D);
              if (gn_per >= 4) D += w2_swz<4>(D);
              if (gn_per >= 8) D += w2_dpp<0x4E>(D);
              const int i = (lane & 1) * 8 + (lane & 2) * 2 + ((lane >> 2) & 1) * 2 + ((lane >> 3) & 1);
              const int cc = i & 7;
              if (lane < 16 && (cc & (gn_per - 1)) == 0) {
                const int nsplit = tiles_x * tiles_y * 2;
                const int slab = ((ty0 / TH) * tiles_x + tx0 / TW) * 2 + wm;
                const int grp = (((tm.tn * BN + wn * 64) >> 3) + cc) >> gn_per_sh;
                if (L.gn_acc) gn_acc_add(L.gn_acc, L.gn_groups, tb, grp, i >> 3, D);   // fixed-point accumulators (common.h)
                else L.gn_partials[(((size_t)tb * nsplit + slab) * L.gn_groups + grp) * 2 + (i >> 3)] = D;
              }
            }
          }
        }
        w2_barrier<false>();
      }
      { const char* t = xa; xa = xn; xn = t; t = sa; sa = sn; sn = t; }
      if (++chunk == nchunks) {
        chunk = 0;
        ++it;
      }
    };
    for (int g = 0; g < nsteps; g += 2) {
      cstep(std::integral_constant<int, 0>{});
      if (g + 1 < nsteps) cstep(std::integral_constant<int, 1>{});
    }
#undef W2X_MM
#undef W2X_SB
    return;
  }

  // ---------------------------------------------------------------------------------------------------
  {
    const int ptid = tid - 256, slot = ptid & 7, row = ptid >> 3;
    char* const Ah0 = smem + row * MXROW + slot * 8;                       // 8 fp8 bytes of halo row `row`
    unsigned char* const Hs0 = reinterpret_cast<unsigned char*>(smem) + G::OFF_HS + row * 2 + (slot >> 2);
    const int Hl = d.Hout, Wl = d.Wout;
    unsigned hpix[KU], hedge[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int hp = k * RPP + row;
      const int hy = hp / HP, hx = hp - hy * HP;
      int ry = hy - 1, rx = hx - 1;
      if (d.ups) { ry >>= 1; rx >>= 1; }
      hpix[k] = (unsigned)((ry + 1) * d.Win + (rx + 1));
      hedge[k] = (hy == 0 ? 1u : 0u) | (hy == TH + 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == TW + 1 ? 8u : 0u) |
                 (hp >= G::HALO ? 16u : 0u);
    }
    // weight tile = 128 rows x 64 bytes: unit u = ptid + 256 j -> row u >> 2, 16-byte piece u & 3; scales: 256 bytes
    const unsigned w_voff = (unsigned)((ptid >> 2) * 64 + (ptid & 3) * 16);
    char* const Bw0 = smem + G::OFF_B + (ptid >> 2) * MXROW + (ptid & 3) * 16;
    char* const Ws0 = smem + G::OFF_WS + (ptid & 15) * 16;
    const unsigned ld_dummy = (unsigned)(d.Win + 1);
    struct StepInfo { int chunk, it, b, y0, x0; };
    int gC = 0;
    auto advance = [&](StepInfo& si) {
      if (gC + 1 < nsteps) {
        ++gC;
        if (++si.chunk == nchunks) {
          si.chunk = 0;
          ++si.it;
          tm.decode(si.it, si.b, si.y0, si.x0, TH, TW);
        }
      }
    };
    StepInfo sA;
    sA.chunk = 0;
    sA.it = 0;
    tm.decode(0, sA.b, sA.y0, sA.x0, TH, TW);
    StepInfo sB = sA;
    advance(sB);
    StepInfo sC = sB;
    advance(sC);

    w2_u32x4 wset[3][2], wsc[3], hreg[KU];
    float4 cf[4], nf[4];
    unsigned hvalid = 0, hvalid_nxt = 0;
    const char* ld_base = nullptr;
    unsigned ld_cs2 = 0, ld_tedge = 0;
    const float* ld_ca = nullptr;
    const float* ld_cb = nullptr;
    // pro_fold (common.h, GnFold): ld_ca / ld_cb point at P / Q and the coefficients are folded here from the image group's
    // fixed-point statistics (ld_acc) when a halo's coefficients are adopted: A = rstd P, B = Q - mean A
    constexpr bool pfold = PRO == 2;
    const long long* ld_acc = nullptr;
    longlong2 nacc = make_longlong2(0, 0);
    auto fold_inplace = [&](float4* c) {
      if constexpr (PRO) {
        if (pfold) {
          float mean, rstd;
          gn_fold_stats_raw(nacc.x, nacc.y, L.pro_fold.inv_n, mean, rstd);
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            c[h2] = make_float4(rstd * c[h2].x, rstd * c[h2].y, rstd * c[h2].z, rstd * c[h2].w);
            c[2 + h2] = make_float4(fmaf(-mean, c[h2].x, c[2 + h2].x), fmaf(-mean, c[h2].y, c[2 + h2].y),
                                    fmaf(-mean, c[h2].z, c[2 + h2].z), fmaf(-mean, c[h2].w, c[2 + h2].w));
          }
        }
      }
    };
    auto issue_setup = [&](const StepInfo& si) {
      const int c = si.chunk * kCH;
      const bool first = c < d.C0;
      const bf16_t* src = first ? L.src0 : L.src1;
      const int cs = first ? d.C0 : d.C1;
      const int cc = first ? c : c - d.C0;
      ld_cs2 = (unsigned)(cs * 2);
      ld_tedge = (si.y0 == 0 ? 1u : 0u) | (si.y0 + TH == Hl ? 2u : 0u) | (si.x0 == 0 ? 4u : 0u) | (si.x0 + TW == Wl ? 8u : 0u) | 16u;
      const int64_t horg = ((int64_t)si.b * d.Hin + (si.y0 >> d.ups)) * d.Win + (si.x0 >> d.ups) - (d.Win + 1);
      ld_base = reinterpret_cast<const char*>(src) + (horg * cs + cc) * 2;
      if constexpr (PRO) {
        const int c0 = si.chunk * kCH + slot * 8;
        if (pfold) {
          const GnFold& f = L.pro_fold;
          ld_ca = f.P + (size_t)si.b * f.pq_stride + c0;
          ld_cb = f.Q + (size_t)si.b * f.pq_stride + c0;
          ld_acc = f.acc + ((size_t)si.b * f.G + c0 / f.cpg) * 2;
        } else {
          const size_t o = (size_t)si.b * d.C0 + c0;
          ld_ca = L.pro_a + o;
          ld_cb = L.pro_b + o;
        }
      }
      hvalid_nxt = 0;
    };
    auto issue_coeffs = [&](float4* dst) {
      if constexpr (PRO) {
        dst[0] = *reinterpret_cast<const float4*>(ld_ca);
        dst[1] = *reinterpret_cast<const float4*>(ld_ca + 4);
        dst[2] = *reinterpret_cast<const float4*>(ld_cb);
        dst[3] = *reinterpret_cast<const float4*>(ld_cb + 4);
        if (pfold) nacc = *reinterpret_cast<const longlong2*>(ld_acc);
      }
    };
    auto issue_unit = [&](int k) {
      const bool ok = (hedge[k] & ld_tedge) == 0 && !(PRG_W256_EXP & 2);   // 2: timing experiment, every unit loads the tile origin
      const unsigned pix = ok ? hpix[k] : ld_dummy;
      const unsigned voff = __umul24(pix, ld_cs2) + (unsigned)(slot * 16);
      if constexpr (!PRO && (PRG_W256_EXP & 64)) {
        // 64: CAP EXPERIMENT (tools/gpu_r5_mxcap.sh): what "MX activations in memory" could buy at most — the unit gathers 8 bytes
        // (as if the producer had stored e4m3) plus one dword standing in for the block scales, and write_unit stores them as they
        // are (masked to finite e4m3, scale 1): HALF the gather bytes, NO quantisation arithmetic.  Results are meaningless.
        const uint2 h = *reinterpret_cast<const uint2*>(ld_base + (voff >> 1));
        const unsigned sc = *reinterpret_cast<const unsigned*>(ld_base + ((voff >> 5) << 2));
        hreg[k] = w2_u32x4{h.x, h.y, sc, 0u};
      } else {
        hreg[k] = *reinterpret_cast<const w2_u32x4*>(ld_base + voff);
      }
      hvalid_nxt |= (ok ? 1u : 0u) << k;
    };
    // one unit = 8 channels of one halo pixel: optional prologue (its result rounded to bf16, like the tensor it replaces),
    // then everything on the packed bf16 words: block maximum of the 15-bit magnitudes (integer order = magnitude order)
    // over the pixel's 32 channels (four lanes, two DPP exchanges), E8M0 scale from the maximum's exponent field, and
    // v_cvt_scalef32_pk_fp8_bf16 (divides by the scale, rounds to nearest even; MODE.FP16_OVFL makes it saturate at +-448
    // instead of returning NaN for the (448, 512) part of a block — probed, tools/micro/cvt_probe.hip).
    __builtin_amdgcn_s_setreg((1 | (23 << 6) | (0 << 11)), 1);          // hwreg(MODE, offset 23, 1 bit) = FP16_OVFL
    auto write_unit = [&](int k, int buf) {
      w2_u32x4 v = hreg[k];
      if constexpr (PRO && !(PRG_W256_EXP & 16)) {
        const float a8[8] = {cf[0].x, cf[0].y, cf[0].z, cf[0].w, cf[1].x, cf[1].y, cf[1].z, cf[1].w};
        const float b8[8] = {cf[2].x, cf[2].y, cf[2].z, cf[2].w, cf[3].x, cf[3].y, cf[3].z, cf[3].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          v[j] = w2_pack(w2_silu(fmaf(w2_lo(v[j]), a8[2 * j], b8[2 * j])), w2_silu(fmaf(w2_hi(v[j]), a8[2 * j + 1], b8[2 * j + 1])));
      }
      if (!((hvalid >> k) & 1u)) v = w2_u32x4{0u, 0u, 0u, 0u};
      if constexpr (!PRO && (PRG_W256_EXP & 64)) {          // cap experiment (see issue_unit): finite e4m3 bytes, unit scale
        *reinterpret_cast<uint2*>(Ah0 + buf * G::AH + k * RPP * MXROW) = make_uint2(v[0] & 0x77777777u, v[1] & 0x77777777u);
        Hs0[buf * G::HS + k * RPP * 2] = (unsigned char)(127u + (v[2] & 0u));
        return;
      }
      if (PRG_W256_EXP & 32) {                               // timing experiment: no quantisation arithmetic
        *reinterpret_cast<uint2*>(Ah0 + buf * G::AH + k * RPP * MXROW) = make_uint2(v[0], v[1]);
        return;
      }
      uint32_t m = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t a = v[j] & 0x7fff7fffu;
        m = max(m, max(a & 0xffffu, a >> 16));
      }
      m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, false));   // lane ^ 1
      m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, false));   // lane ^ 2: the block's four lanes agree
      const int E = (int)(m >> 7);                           // exponent field of the block maximum
      const int S = E > 8 ? E - 8 : 0;                       // E8M0 scale 2^(S - 127); an all-zero block has S = 0
      // (S = 0 means a maximum below 2^-118: the divisor is taken as 2^-126, the smallest normal float — such elements
      // contribute nothing either way)
      const float scale = __uint_as_float((uint32_t)(S > 1 ? S : 1) << 23);
      // (inline asm: hipcc 7.2 folds the chained __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16 calls onto the first source word)
      uint2 w = make_uint2(0u, 0u);
      asm("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(w.x) : "v"(v[0]), "v"(scale));
      asm("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(w.x) : "v"(v[1]), "v"(scale));
      asm("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(w.y) : "v"(v[2]), "v"(scale));
      asm("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(w.y) : "v"(v[3]), "v"(scale));
      *reinterpret_cast<uint2*>(Ah0 + buf * G::AH + k * RPP * MXROW) = w;
      Hs0[buf * G::HS + k * RPP * 2] = (unsigned char)S;    // the block's four lanes store the same byte
    };
    auto w_issue = [&](int set, int tap, int chunk) {
      const size_t tile = ((size_t)(tap * nchunks + chunk) * d.CoutPad + (size_t)tm.tn * BN);
      const char* p = reinterpret_cast<const char*>(L.w_mx) + ((PRG_W256_EXP & 8) ? 0 : tile * 64) + w_voff;   // 8: timing experiment, one L1-resident tile
      wset[set][0] = *reinterpret_cast<const w2_u32x4*>(p);
      wset[set][1] = *reinterpret_cast<const w2_u32x4*>(p + 64 * 64);                     // rows 64 .. 127
      wsc[set] = *reinterpret_cast<const w2_u32x4*>(reinterpret_cast<const char*>(L.w_mx_scale) + tile * 2 + (ptid & 15) * 16);
    };
    auto w_write = [&](int set, int ring) {
      *reinterpret_cast<w2_u32x4*>(Bw0 + ring * G::BW) = wset[set][0];
      *reinterpret_cast<w2_u32x4*>(Bw0 + ring * G::BW + 64 * MXROW) = wset[set][1];
      *reinterpret_cast<w2_u32x4*>(Ws0 + ring * G::WS) = wsc[set];   // 16 copies of the same 256 bytes: no branch
    };

    issue_setup(sA);
    issue_coeffs(cf);
    fold_inplace(cf);
#pragma unroll
    for (int k = 0; k < KU; ++k) issue_unit(k);
    hvalid = hvalid_nxt;
    w_issue(0, 0, sA.chunk);
    w_issue(1, 1, sA.chunk);
#pragma unroll
    for (int k = 0; k < KU; ++k) write_unit(k, 0);
    w_write(0, 0);
    w_write(1, 1);
    issue_setup(sB);
    issue_coeffs(nf);
#pragma unroll
    for (int k = 0; k < KU; ++k) issue_unit(k);
    hvalid = hvalid_nxt;
#pragma unroll
    for (int j = 0; j < 4; ++j) cf[j] = nf[j];
    fold_inplace(cf);
    w_issue(2, 2, sA.chunk);
    w_issue(0, 3, sA.chunk);
    w_issue(1, 4, sA.chunk);
    issue_setup(sC);
    w2_barrier<true>();

    constexpr int US[10] = {0, 2, 4, 6, 7, 8, 9, 10, 11, 11};
#pragma unroll 1
    for (int g = 0; g < nsteps; ++g) {
      const int buf = (g + 1) & 1;
      const int ch0 = sA.chunk, ch1 = sB.chunk;
#pragma unroll
      for (int p = 0; p < 9; ++p) {
        const int set = (p + 2) % 3;
        w_write(set, set);
#pragma unroll
        for (int k = US[p]; k < US[p + 1]; ++k) write_unit(k, buf);
        if (p == 0) issue_coeffs(nf);
        w_issue(set, (p + 5) % 9, p + 5 < 9 ? ch0 : ch1);
#pragma unroll
        for (int k = US[p]; k < US[p + 1]; ++k) issue_unit(k);
        if (p == 8) {
          hvalid = hvalid_nxt;
#pragma unroll
          for (int j = 0; j < 4; ++j) cf[j] = nf[j];
          fold_inplace(cf);
          sA = sB;
          sB = sC;
          advance(sC);
          issue_setup(sC);
        }
        w2_barrier<true>();
      }
    }
  }
}

}  // namespace

// Returns 1 when it launched, 0 when the shape is not covered (caller falls back), negative on error.
int materialize_prologue(ConvLaunch<bf16_t>& L, hipStream_t s);   // conv.hip

// in-kernel GroupNorm fold of the prologue (common.h, GnFold): valid when an 8-channel unit lies inside one group
static int w256_prepare_fold(ConvLaunch<bf16_t>& Lk, hipStream_t s) {
  if (!Lk.pro_fold.acc) return PRG_OK;
  const GnFold& f = Lk.pro_fold;
  if (f.P && f.Q && f.cpg % 8 == 0 && f.G * f.cpg == Lk.d.C0 && Lk.d.C1 == 0) return PRG_OK;
  return materialize_prologue(Lk, s);
}

int try_launch_conv3x3_w256(const ConvLaunch<bf16_t>& L, hipStream_t s, int* gn_nsplit_out, int* acc_done) {
  static const int enabled = [] {
    const char* e = std::getenv("PRG_CONV_W256");
    return e ? std::atoi(e) : 1;
  }();
  if (!enabled) return 0;
  const ConvDesc& d = L.d;
  if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1)) return 0;
  if (d.C0 % kCH || d.C1 % kCH || d.C0 == 0 || d.Cout % BN || d.CoutPad != d.Cout) return 0;
  if (L.residual || !L.bias) return 0;
  if (L.pro_a && d.C1) return 0;                            // the fused prologue is defined for a single source
  const int tiles_n = d.Cout / BN;
  if (tiles_n != 1 && tiles_n != 2 && tiles_n != 4 && tiles_n != 8) return 0;   // an XCD is pinned to one channel tile
  const int H = d.Hout, W = d.Wout;
  int tw = 0;
  if (W % 32 == 0 && H % 8 == 0) tw = 32;
  else if (W % 16 == 0 && H % 16 == 0) tw = 16;
  else return 0;
  if (d.ups && ((H | W) & 1)) return 0;
  const int th = 256 / tw, tiles_x = W / tw, tiles_y = H / th;
  const long total = (long)tiles_x * tiles_y * tiles_n * d.B;
  const int num_cus = device_cu_count();
  if (num_cus <= 0) return 0;
  // fewer tiles than CUs: the 128-pixel tiles of conv_ws.hip fill the chip better
  static const int min_fill = [] { const char* e = std::getenv("PRG_W256_MIN_TILES"); return e ? std::atoi(e) : 0; }();
  const int grid = num_cus & ~7;
  if (grid < 8 || total < (min_fill > 0 ? min_fill : grid)) return 0;
  const int cpg = L.gn_groups > 0 ? d.Cout / L.gn_groups : 0;
  const int fuse = L.gn_partials != nullptr && cpg % 8 == 0 && cpg <= 64 && (cpg & (cpg - 1)) == 0 &&
                   tiles_x * tiles_y * 2 <= kGnMaxSplit;
  if (L.gn_partials && !fuse) return 0;
  ConvLaunch<bf16_t> Lk = L;
  if (L.in_f16) {                                            // h16 input: only through the in-kernel fold, with f16 weights
    const GnFold& f = L.pro_fold;
    if (!(f.acc && f.P && f.Q && f.cpg % 8 == 0 && f.G * f.cpg == d.C0 && d.C1 == 0 && L.w_f16)) return 0;
  }
  if (L.out_f16 && (L.pro_a || L.pro_fold.acc)) return 0;   // h16 output: conv1 of a ResnetBlock (no prologue)
  if (!L.probe)
    if (int rc = w256_prepare_fold(Lk, s)) return rc;
  const int pro = L.in_f16 ? 3 : Lk.pro_fold.acc ? 2 : (L.pro_a ? 1 : 0);
  const void* fns[2][4] = {{reinterpret_cast<const void*>(&conv3x3_w256_kernel<16, 0, 0>), reinterpret_cast<const void*>(&conv3x3_w256_kernel<16, 1, 0>),
                            reinterpret_cast<const void*>(&conv3x3_w256_kernel<16, 2, 0>), reinterpret_cast<const void*>(&conv3x3_w256_kernel<16, 3, 0>)},
                           {reinterpret_cast<const void*>(&conv3x3_w256_kernel<32, 0, 0>), reinterpret_cast<const void*>(&conv3x3_w256_kernel<32, 1, 0>),
                            reinterpret_cast<const void*>(&conv3x3_w256_kernel<32, 2, 0>), reinterpret_cast<const void*>(&conv3x3_w256_kernel<32, 3, 0>)}};
  const size_t lds = tw == 32 ? W2Geom<32>::LDS : W2Geom<16>::LDS;
  static DeviceOnce attr_done[2][4];   // zero-initialised; atomic: lanes launch from several host threads
  if (!attr_done[tw == 32][pro].done()) {
    hipError_t e = hipFuncSetAttribute(fns[tw == 32][pro], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(PRG_E_HIP, std::string("hipFuncSetAttribute(w256 conv): ") + hipGetErrorString(e));
    attr_done[tw == 32][pro].mark();
  }
  if (gn_nsplit_out) *gn_nsplit_out = fuse ? tiles_x * tiles_y * 2 : 0;
  if (acc_done) *acc_done = (fuse && L.gn_acc) ? 1 : 0;
  if (L.probe) return 1;
  if (tw == 32) {
    if (pro == 3) conv3x3_w256_kernel<32, 3, 0><<<dim3(grid), 512, lds, s>>>(Lk, tiles_x, tiles_y, tiles_n, fuse);
    else if (pro == 2) conv3x3_w256_kernel<32, 2, 0><<<dim3(grid), 512, lds, s>>>(Lk, tiles_x, tiles_y, tiles_n, fuse);
    else if (pro == 1) conv3x3_w256_kernel<32, 1, 0><<<dim3(grid), 512, lds, s>>>(Lk, tiles_x, tiles_y, tiles_n, fuse);
    else conv3x3_w256_kernel<32, 0, 0><<<dim3(grid), 512, lds, s>>>(Lk, tiles_x, tiles_y, tiles_n, fuse);
  } else {
    if (pro == 3) conv3x3_w256_kernel<16, 3, 0><<<dim3(grid), 512, lds, s>>>(Lk, tiles_x, tiles_y, tiles_n, fuse);
    else if (pro == 2) conv3x3_w256_kernel<16, 2, 0><<<dim3(grid), 512, lds, s>>>(Lk, tiles_x, tiles_y, tiles_n, fuse);
    else if (pro == 1) conv3x3_w256_kernel<16, 1, 0><<<dim3(grid), 512, lds, s>>>(Lk, tiles_x, tiles_y, tiles_n, fuse);
    else conv3x3_w256_kernel<16, 0, 0><<<dim3(grid), 512, lds, s>>>(Lk, tiles_x, tiles_y, tiles_n, fuse);
  }
  PRG_LAUNCH_CHECK();
  return 1;
}

// Upsample (nearest x2) + 3x3 conv as four 2 x 2-tap sub-pixel convolutions of the source image (MODE 2).  Returns 1 / 0 / negative.
int try_launch_conv3x3_up_w256(const ConvLaunch<bf16_t>& L, hipStream_t s) {
  static const int enabled = [] {
    const char* e = std::getenv("PRG_UP2X2");
    return e ? std::atoi(e) : 1;
  }();
  if (!enabled || !L.w_up) return 0;
  const ConvDesc& d = L.d;
  if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.ups == 1 && d.C1 == 0)) return 0;
  if (d.C0 % kCH || d.C0 == 0 || (d.Cout != 64 && d.Cout % BN) || d.CoutPad != d.Cout) return 0;
  if (L.residual || !L.bias || L.pro_a || L.pro_fold.acc || L.gn_partials || L.out_f16 || L.in_f16) return 0;
  if (d.Hout != 2 * d.Hin || d.Wout != 2 * d.Win) return 0;
  const int tiles_n = d.Cout == 64 ? 1 : d.Cout / BN;
  if (tiles_n != 1 && tiles_n != 2 && tiles_n != 4 && tiles_n != 8) return 0;
  const int H = d.Hin, W = d.Win;                            // tiles of 256 SOURCE pixels, four phases each
  int tw = 0;
  if (W % 32 == 0 && H % 8 == 0) tw = 32;
  else if (W % 16 == 0 && H % 16 == 0) tw = 16;
  else return 0;
  if ((size_t)d.B * d.Hin * d.Win * d.C0 * 2 >= ((size_t)1 << 31)) return 0;   // (32-bit halo offsets, as the other modes)
  const int th = 256 / tw, tiles_x = W / tw, tiles_y = H / th;
  const long total = (long)tiles_x * tiles_y * tiles_n * d.B * (d.Cout == 64 ? 2 : 4);   // (Cout = 64: two x-phases per tile)
  const int num_cus = device_cu_count();
  if (num_cus <= 0) return 0;
  const int grid = num_cus & ~7;
  static const int min_fill = [] { const char* e = std::getenv("PRG_W256_MIN_TILES"); return e ? std::atoi(e) : 0; }();
  if (grid < 8 || total < (min_fill > 0 ? min_fill : grid / 2)) return 0;
  const void* fn = tw == 32 ? reinterpret_cast<const void*>(&conv3x3_w256_kernel<32, 0, 2>)
                            : reinterpret_cast<const void*>(&conv3x3_w256_kernel<16, 0, 2>);
  const size_t lds = tw == 32 ? W2Geom<32>::LDS : W2Geom<16>::LDS;
  static DeviceOnce attr_done[2];
  if (!attr_done[tw == 32].done()) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(PRG_E_HIP, std::string("hipFuncSetAttribute(w256 upsample): ") + hipGetErrorString(e));
    attr_done[tw == 32].mark();
  }
  if (L.probe) return 1;
  if (tw == 32) conv3x3_w256_kernel<32, 0, 2><<<dim3(grid), 512, lds, s>>>(L, tiles_x, tiles_y, tiles_n, 0);
  else conv3x3_w256_kernel<16, 0, 2><<<dim3(grid), 512, lds, s>>>(L, tiles_x, tiles_y, tiles_n, 0);
  PRG_LAUNCH_CHECK();
  return 1;
}

// MX-fp8 operands (handles of dtype PRG_MXFP8): same shapes as the bf16 entry.  Returns 1 / 0 / negative.
int try_launch_conv3x3_w256mx(const ConvLaunch<bf16_t>& L, hipStream_t s, int* gn_nsplit_out, int* acc_done) {
  static const int enabled = [] {
    const char* e = std::getenv("PRG_CONV_W256MX");
    return e ? std::atoi(e) : 1;
  }();
  if (!enabled || !L.w_mx || !L.w_mx_scale) return 0;
  const ConvDesc& d = L.d;
  if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1)) return 0;
  if (d.C0 % kCH || d.C1 % kCH || d.C0 == 0 || d.Cout % BN || d.CoutPad != d.Cout) return 0;
  if (L.residual || !L.bias) return 0;
  if (L.pro_a && d.C1) return 0;
  const int tiles_n = d.Cout / BN;
  if (tiles_n != 1 && tiles_n != 2 && tiles_n != 4 && tiles_n != 8) return 0;
  const int H = d.Hout, W = d.Wout;
  int tw = 0;
  if (W % 32 == 0 && H % 8 == 0) tw = 32;
  else if (W % 16 == 0 && H % 16 == 0) tw = 16;
  else return 0;
  if (d.ups && ((H | W) & 1)) return 0;
  const int th = 256 / tw, tiles_x = W / tw, tiles_y = H / th;
  const long total = (long)tiles_x * tiles_y * tiles_n * d.B;
  const int num_cus = device_cu_count();
  if (num_cus <= 0) return 0;
  static const int min_fill = [] { const char* e = std::getenv("PRG_W256_MIN_TILES"); return e ? std::atoi(e) : 0; }();
  const int grid = num_cus & ~7;
  // Round 5: a launch with fewer tiles than CUs stays on the bf16 kernels (the network's level-3 convs: 128 tiles on 256 CUs — measured
  // at the configs[4] shape, same box, same positions: 25 / 29 us on conv3x3_ws_kernel against 28-29 / 38-40 us here,
  // profiles/r05_configs4_conv_per_launch_bf16_vs_mxfp8.txt); mx_pure (the conv-level tests) keeps the old half-a-wave threshold
  if (grid < 8 || total < (min_fill > 0 ? min_fill : (L.mx_pure ? grid / 2 : grid))) return 0;
  const int cpg = L.gn_groups > 0 ? d.Cout / L.gn_groups : 0;
  const int fuse = L.gn_partials != nullptr && cpg % 8 == 0 && cpg <= 64 && (cpg & (cpg - 1)) == 0 &&
                   tiles_x * tiles_y * 2 <= kGnMaxSplit;
  if (L.gn_partials && !fuse) return 0;
  ConvLaunch<bf16_t> Lk = L;
  if (int rc = w256_prepare_fold(Lk, s)) return rc;
  const int pro = Lk.pro_fold.acc ? 2 : (L.pro_a ? 1 : 0);
  const void* fns[2][3] = {{reinterpret_cast<const void*>(&conv3x3_w256mx_kernel<16, 0>), reinterpret_cast<const void*>(&conv3x3_w256mx_kernel<16, 1>),
                            reinterpret_cast<const void*>(&conv3x3_w256mx_kernel<16, 2>)},
                           {reinterpret_cast<const void*>(&conv3x3_w256mx_kernel<32, 0>), reinterpret_cast<const void*>(&conv3x3_w256mx_kernel<32, 1>),
                            reinterpret_cast<const void*>(&conv3x3_w256mx_kernel<32, 2>)}};
  const size_t lds = tw == 32 ? W2MxGeom<32>::LDS : W2MxGeom<16>::LDS;
  static DeviceOnce attr_done[2][3];   // zero-initialised; atomic: lanes launch from several host threads
  if (!attr_done[tw == 32][pro].done()) {
    hipError_t e = hipFuncSetAttribute(fns[tw == 32][pro], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(PRG_E_HIP, std::string("hipFuncSetAttribute(w256 mx conv): ") + hipGetErrorString(e));
    attr_done[tw == 32][pro].mark();
  }
  if (gn_nsplit_out) *gn_nsplit_out = fuse ? tiles_x * tiles_y * 2 : 0;
  if (acc_done) *acc_done = (fuse && L.gn_acc) ? 1 : 0;
  if (tw == 32) {
    if (pro == 2) conv3x3_w256mx_kernel<32, 2><<<dim3(grid), 512, lds, s>>>(Lk, tiles_x, tiles_y, tiles_n, fuse);
    else if (pro == 1) conv3x3_w256mx_kernel<32, 1><<<dim3(grid), 512, lds, s>>>(Lk, tiles_x, tiles_y, tiles_n, fuse);
    else conv3x3_w256mx_kernel<32, 0><<<dim3(grid), 512, lds, s>>>(Lk, tiles_x, tiles_y, tiles_n, fuse);
  } else {
    if (pro == 2) conv3x3_w256mx_kernel<16, 2><<<dim3(grid), 512, lds, s>>>(Lk, tiles_x, tiles_y, tiles_n, fuse);
    else if (pro == 1) conv3x3_w256mx_kernel<16, 1><<<dim3(grid), 512, lds, s>>>(Lk, tiles_x, tiles_y, tiles_n, fuse);
    else conv3x3_w256mx_kernel<16, 0><<<dim3(grid), 512, lds, s>>>(Lk, tiles_x, tiles_y, tiles_n, fuse);
  }
  PRG_LAUNCH_CHECK();
  return 1;
}

// Conv2d(C, Cout, 4, stride 2, pad 1) through the same kernel (MODE 1).  Returns 1 / 0 / negative like the entry above.
int try_launch_conv4x4s2_w256(const ConvLaunch<bf16_t>& L, hipStream_t s) {
  static const int enabled = [] {
    const char* e = std::getenv("PRG_CONV_DOWN_W256");
    return e ? std::atoi(e) : 1;
  }();
  if (!enabled || !L.w_s2d) return 0;
  const ConvDesc& d = L.d;
  if (!(d.KH == 4 && d.KW == 4 && d.stride == 2 && d.pad == 1 && d.ups == 0 && d.C1 == 0)) return 0;
  if (d.C0 % kCH || d.C0 == 0 || d.C0 > 128 || (d.Cout != 64 && d.Cout % BN) || d.CoutPad != d.Cout) return 0;
  if (L.residual || !L.bias || L.pro_a || L.gn_partials) return 0;
  if (d.Hin != 2 * d.Hout || d.Win != 2 * d.Wout) return 0;
  const int tiles_n = d.Cout == 64 ? 1 : d.Cout / BN;
  if (tiles_n != 1 && tiles_n != 2 && tiles_n != 4 && tiles_n != 8) return 0;
  const int H = d.Hout, W = d.Wout;
  int tw = 0;
  if (W % 32 == 0 && H % 8 == 0) tw = 32;
  else if (W % 16 == 0 && H % 16 == 0) tw = 16;
  else return 0;
  const int th = 256 / tw, tiles_x = W / tw, tiles_y = H / th;
  const long total = (long)tiles_x * tiles_y * tiles_n * d.B;
  const int num_cus = device_cu_count();
  if (num_cus <= 0) return 0;
  const int grid = num_cus & ~7;
  static const int min_fill = [] { const char* e = std::getenv("PRG_W256_MIN_TILES"); return e ? std::atoi(e) : 0; }();
  if (grid < 8 || total < (min_fill > 0 ? min_fill : grid / 2)) return 0;   // the generic kernel for tiny launches
  const void* fn = tw == 32 ? reinterpret_cast<const void*>(&conv3x3_w256_kernel<32, 0, 1>)
                            : reinterpret_cast<const void*>(&conv3x3_w256_kernel<16, 0, 1>);
  const size_t lds = tw == 32 ? W2Geom<32>::LDS : W2Geom<16>::LDS;
  static DeviceOnce attr_done[2];
  if (!attr_done[tw == 32].done()) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(PRG_E_HIP, std::string("hipFuncSetAttribute(w256 downsample): ") + hipGetErrorString(e));
    attr_done[tw == 32].mark();
  }
  if (tw == 32) conv3x3_w256_kernel<32, 0, 1><<<dim3(grid), 512, lds, s>>>(L, tiles_x, tiles_y, tiles_n, 0);
  else conv3x3_w256_kernel<16, 0, 1><<<dim3(grid), 512, lds, s>>>(L, tiles_x, tiles_y, tiles_n, 0);
  PRG_LAUNCH_CHECK();
  return 1;
}

}  // namespace prg
