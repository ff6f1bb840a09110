// conv_c64.hip — weights-stationary 3x3 convolution for Cin = Cout = 64 (bf16 throughput path).
//
// The 64 -> 64 convs of the two full-resolution levels (down0/down1 blocks, the second conv of every up3 / final block,
// the last up conv) are the ones the wave-specialised kernel (conv_ws.hip) handles worst: one 64-channel chunk means a
// tile is finished after nine taps, so per 4608 MFMA cycles it re-stages the whole 72 KB weight set through LDS, meets
// nine workgroup barriers and pays one epilogue — 0.25-0.33 of the MFMA peak, while the shape itself is HBM-bound
// (256 B of activations per pixel against 73.7 kFLOP: 49 us of HBM time, 31 us of MFMA time at B = 64, 128 x 128).
//
// Here the weights never touch LDS again after the prologue: a consumer wave owns 32 output channels and keeps their
// 9 x 64 weights as 36 MFMA A-fragments in registers (144 VGPRs) for the whole launch; its B operands are the pixels of
// a 128-pixel half tile read from the halo (round 6: every fragment read once and used for the three taps of its column —
// 72 ds_read_b128 per 144 MFMAs; no weight reads, no weight ring, no per-tap barrier).  The workgroup (512 threads, 256 registers per lane) is
//
//   waves 0-3  CONSUMERS (2 pixel halves x 2 channel halves of a 8 x 32 pixel x 64 channel tile): 144 MFMAs per tile,
//              fragments prefetched three reads ahead; epilogue = bias, GroupNorm partial sums (the deterministic butterfly
//              of conv_ws.hip), bf16 and DIRECT stores: a v_permlane32_swap pairs the two lane halves' channel quads, so
//              every lane stores 16 contiguous bytes (no LDS stage, nothing for the producers to drain);
//   waves 4-7  PRODUCERS: write halo s+1 (loaded a whole step earlier; optional fused GroupNorm + (scale+1, shift) + SiLU
//              prologue of the previous Block, sd:690-696), re-issue the registers for halo s+2.  A step is ~5000 cycles,
//              so plain loads with compiler-managed waits are enough: every load has a full step to land.
//
// ONE barrier per tile, between the consumers' MFMAs and their epilogue: "halo s+1 written; buffer s & 1 no longer read".
// The epilogue of tile s therefore runs while the producers already write halo s+2 (round 2 had two barriers and an LDS
// stage: producers idle during the epilogue, consumers idle during the drain — the two were additive, 81 / 122 us).
// LDS: 2 halos x 340 rows x 144 B (padded rows: conflict-free ds_read_b128 with immediate tap offsets).
#include <atomic>
#include <cstdlib>

#include "conv.h"

namespace prg {

typedef __attribute__((ext_vector_type(8))) __bf16 c64_bf16x8;
typedef __attribute__((ext_vector_type(16))) float c64_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int c64_u32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 c64_f16x8;

namespace {

// Timing experiments only (results become garbage): -DPRG_C64_EXP=1 consumers skip fragment reads + MFMAs, 2 producers skip
// the halo loads / writes, 4 no epilogue, 8 no priority raise, 32 producers skip only the prologue arithmetic, 256 SiLU
// without transcendentals, 1024 the round-5 consumer loop (one read per MFMA) instead of the row-reuse loop; with 1024: 512 half the
// fragment reads (pairs of MFMAs share a fragment), 2048 all the reads but MFMA pairs sharing their operands (both with opaque accumulators).
#ifndef PRG_C64_EXP
#define PRG_C64_EXP 0
#endif

constexpr int TH = 8, TW = 32, HP = TW + 2, HALO = (TH + 2) * HP;   // 340 halo rows
constexpr int ROWB = 144;                                              // padded LDS row (64 bf16 = 128 B + 16)
constexpr int NPT = 256;                                               // producer threads
constexpr int RPP = NPT / 8;                                           // halo rows per pass
constexpr int KU = (HALO + RPP - 1) / RPP;                             // 11 units per producer thread
constexpr size_t AH_BYTES = (size_t)KU * RPP * ROWB;                   // 352 rows: the units past the halo end land in spare rows
constexpr size_t C64_LDS = 2 * AH_BYTES + 64 * sizeof(float);          // + the bias
constexpr bool kRowReuse = (PRG_C64_EXP & 1024) == 0;                 // -DPRG_C64_EXP=1024: the round-5 consumer loop (one read per MFMA), for A/B builds

__device__ inline float c64_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ inline float c64_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ inline uint32_t c64_pack(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ inline float c64_silu(float x) {
  if (PRG_C64_EXP & 256) return x * fmaf(fmaf(x, -1.4426950408889634f, 1.0f), x, 0.5f);   // timing experiment: no transcendentals
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
template <int CTRL>
__device__ inline float c64_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int XOR>
__device__ inline float c64_swz(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (XOR << 10) | 0x1F));
}

// tile t of this workgroup: XCD-contiguous runs (workgroup b runs on XCD b % 8; speed only), image-major
__device__ inline void c64_tile(int t, int tiles_x, int tiles_y, int& b, int& y0, int& x0) {
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y;
  b = t / tiles_y;
  y0 = ty * TH;
  x0 = tx * TW;
}

// one LDS-visibility point: this wave's LDS operations are done, then the workgroup barrier.  Never waits for VMEM: the
// producers' halo loads and drain stores stay in flight across it.
__device__ __forceinline__ void c64_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// What gn_coeff_kernel does in a launch of its own, for ONE image, by one wave (lane = channel of the 64): the image's
// (sum, sumsq) slabs summed in float64 — lanes of a group take slabs (c % cpg), + cpg, ... then a fixed xor tree —, mean /
// rstd, folded with the norm's gain / bias and the ResnetBlock conditioning into y = x * A + B.  The slabs were written by
// other CUs with agent-scope stores; they are read with agent-scope (L1-bypassing) loads.
__device__ inline void c64_fold_coefficients(const ConvLaunch<bf16_t>& L, int img, int nsplit, int lane) {
  const int G = L.gn_groups, cpg = 64 / G;
  const int c = lane, g = c / cpg, j = c % cpg;
  const GnApply& p = L.gn;
  const float* ssa = nullptr;
  const float* ssb = nullptr;
  if (p.ss_a) {
    ssa = p.ss_a + (size_t)img * p.ss_a_stride;
    if (p.ss_a_row) ssa += (size_t)(*p.ss_a_row) * p.ss_a_row_stride;
    if (p.ss_b) ssb = p.ss_b + (size_t)img * p.ss_b_stride;
  }
  const float gam = p.gamma[c], bet = p.beta[c];
  float s0 = 0.0f, s1 = 0.0f;
  if (ssa) {
    s0 = ssa[c];
    s1 = ssa[64 + c];
    if (ssb) { s0 += ssb[c]; s1 += ssb[64 + c]; }
  }
  const unsigned long long* pp =
      reinterpret_cast<const unsigned long long*>(L.gn_partials + ((size_t)img * nsplit * G + g) * 2);
  double ss = 0, qq = 0;
  for (int k0 = j; k0 < nsplit; k0 += 8 * cpg) {           // eight loads in flight per lane, summed in index order
    unsigned long long v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = k0 + i * cpg;
      v[i] = k < nsplit ? __hip_atomic_load(pp + (size_t)k * G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ss += (double)__uint_as_float((unsigned)(v[i] & 0xffffffffull));
      qq += (double)__uint_as_float((unsigned)(v[i] >> 32));
    }
  }
  for (int o = cpg >> 1; o > 0; o >>= 1) {
    ss += __shfl_xor(ss, o, 64);
    qq += __shfl_xor(qq, o, 64);
  }
  const double n = (double)L.d.Hout * L.d.Wout * cpg;
  const double mean = ss / n;
  double var = qq / n - mean * mean;
  if (var < 0) var = 0;
  const float fmean = (float)mean, frstd = (float)(1.0 / sqrt(var + 1e-5));
  float a = frstd * gam;
  float bb = bet - fmean * a;
  if (ssa) {
    a = a * (s0 + 1.0f);
    bb = fmaf(bb, s0 + 1.0f, s1);
  }
  L.gn_coef_a[(size_t)img * 64 + c] = a;
  L.gn_coef_b[(size_t)img * 64 + c] = bb;
}

// PRO 0: no prologue; 1: GroupNorm coefficient tables (pro_a / pro_b); 2: coefficients folded here from pro_fold; 3: as 2 on an
// f16 input with f16 weights (the h16 format, conv.h: packed-f16 prologue, v_mfma_f32_32x32x16_f16).  O16: f16 output.
template <int PRO, bool O16>
__global__ __launch_bounds__(512) void conv3x3_c64_kernel(const ConvLaunch<bf16_t> L, const int tiles_x, const int tiles_y,
                                                          const int flags) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int fuse_stats = flags & 1;
  const ConvDesc& d = L.d;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int total = tiles_x * tiles_y * d.B;
  // XCD-contiguous tile runs (gridDim.x is a multiple of 8): XCD x owns tiles [x per, (x+1) per); inside, every workgroup
  // takes ONE contiguous run (stride 1): neighbouring tiles share halo rows in the XCD's L2 and a workgroup leaves an
  // image at most twice per launch (the per-image coefficient ticket below is one atomic per departure).
  const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int per = (total + 7) / 8;
  const int lo_t = min(xcd * per, total), hi_t = min((xcd + 1) * per, total);
  const int nx = hi_t - lo_t, qx = nx / wpx, rx = nx % wpx;
  // fuse_stats bit 1: interleaved assignment (workgroup w of the XCD takes tiles w, w + wpx, ...) instead of contiguous runs
  const bool inter = (flags & 2) != 0;
  const int first = inter ? lo_t + widx : lo_t + widx * qx + min(widx, rx);
  const int nsteps = inter ? (widx < nx ? (nx - widx + wpx - 1) / wpx : 0) : qx + (widx < rx ? 1 : 0);
  const int stride = inter ? wpx : 1;
  if (nsteps == 0) return;

  // ---------------------------------------------------------------------------------------------------
  if (wave < 4) {
    if (!(PRG_C64_EXP & 8)) __builtin_amdgcn_s_setprio(3);
    const int wm = wave >> 1, wn = wave & 1;              // pixel half (tile rows 4 wm .. 4 wm + 3), channel half
    const int l31 = lane & 31, hi = lane >> 5;
    // weights of channel wn*32 + l31: 9 taps x 4 k-steps; lane half hi takes k = 16 c + 8 hi .. + 7
    c64_bf16x8 wf[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16_t* const wsrc = PRO == 3 ? reinterpret_cast<const bf16_t*>(L.w_f16) : L.w;
        const bf16_t* p = wsrc + ((size_t)(tap * d.kchunks + (c >> 1)) * d.CoutPad + wn * 32 + l31) * 32 + (c & 1) * 16 + hi * 8;
        wf[tap][c] = *reinterpret_cast<const c64_bf16x8*>(p);
      }
    float* const bias_lds = reinterpret_cast<float*>(smem + 2 * AH_BYTES);
    if (tid < 64) bias_lds[tid] = L.bias[tid];             // visible after the prologue barrier
    const float* const biasp = bias_lds + wn * 32 + 4 * hi;
    const int gn_per = fuse_stats ? (64 / L.gn_groups) >> 3 : 1;   // 8-channel chunks per group (1, 2, 4 or 8)
    // LDS byte offset of pixel (row 4 wm + pt, column l31), tap (0,0), k-step 0: rows are HP * ROWB apart
    const unsigned x0off = (unsigned)(((wm * 4) * HP + l31) * ROWB + hi * 16);
    auto mma = [](const c64_bf16x8& a, const c64_bf16x8& b, const c64_f32x16& c) -> c64_f32x16 {
      if constexpr (PRO == 3)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c64_f16x8, a), __builtin_bit_cast(c64_f16x8, b), c, 0, 0, 0);
      else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    };
    c64_barrier();                                         // halo 0 is in LDS (producers' prologue)
    for (int s = 0; s < nsteps; ++s) {
      const char* const xb = smem + (size_t)(s & 1) * AH_BYTES + x0off;
      // Round 5: the accumulators START at the bias (16 values per lane, re-read from LDS per tile: they are dead again before the
      // fragment sets fill up) instead of at zero — the epilogue's 64 bias additions per tile and wave are gone; the sum's rounding
      // order changes (bias first), far below the bf16 / f16 output rounding.  PRG_C64_EXP & 256: the old form (A/B builds).
      c64_f32x16 acc[4];
      if constexpr ((PRG_C64_EXP & 256) != 0) {
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[pt][e] = 0.0f;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b4 = *reinterpret_cast<const float4*>(biasp + 8 * q);
#pragma unroll
          for (int pt = 0; pt < 4; ++pt) { acc[pt][4 * q] = b4.x; acc[pt][4 * q + 1] = b4.y; acc[pt][4 * q + 2] = b4.z; acc[pt][4 * q + 3] = b4.w; }
        }
      }
      if constexpr ((PRG_C64_EXP & (512 | 2048)) != 0) {
        // the shared-operand timing experiments: make the accumulators opaque, or the compiler proves acc[1] == acc[0] and acc[3] == acc[2]
        // (identical initial values, identical MFMA chains) and DROPS half the MFMAs — which is what the first version of these experiments
        // measured (74 v_mfma instead of 144 in the ISA; profiles/r06_c64_half_reads_bound.txt)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) asm volatile("" : "+v"(acc[pt]));
      }
      if constexpr (kRowReuse) {
        // Round 6: ROW REUSE.  Tap (dy, dx) of pixel row pt reads halo row pt + dy: the twelve (pt, dy) pairs of one (dx, k-step) touch
        // only SIX halo rows.  The four accumulators are independent chains, so they are skewed by one tap row: fragment (halo row r,
        // dx, c) is read ONCE and feeds acc[r] at tap (0, dx), acc[r-1] at (1, dx) and acc[r-2] at (2, dx) back to back — 72 reads
        // per tile instead of 144 (0.5 ds_read_b128 per MFMA), no extra registers, and every accumulator still sees its MFMAs in
        // tap-major order: the outputs are bit-identical.  Measured -1.0 ... -1.5 % per launch (profiles/r06_ab_c64_row_reuse.txt): the
        // fragment reads are a small term of this kernel (profiles/r06_c64_half_reads_bound.txt).  Ring of four fragments, three ahead.
        constexpr int FD = 3, FR = 4, NF = (PRG_C64_EXP & 1) ? 0 : 72;
        c64_bf16x8 fr[FR];
        auto foff = [](int n) { return ((n / 12) * HP + (n / 4) % 3) * ROWB + (n & 3) * 32; };   // n = (r * 3 + dx) * 4 + c
#pragma unroll
        for (int j = 0; j < FD && j < NF; ++j) fr[j] = *reinterpret_cast<const c64_bf16x8*>(xb + foff(j));
#pragma unroll
        for (int n = 0; n < NF; ++n) {
          const int r = n / 12, dx = (n / 4) % 3, c = n & 3;
          if (n + FD < NF) fr[(n + FD) % FR] = *reinterpret_cast<const c64_bf16x8*>(xb + foff(n + FD));
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int pt = r - dy;
            if (pt >= 0 && pt < 4) acc[pt] = mma(wf[dy * 3 + dx][c], fr[n % FR], acc[pt]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        c64_bf16x8 fx[2][4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) fx[0][pt] = *reinterpret_cast<const c64_bf16x8*>(xb + pt * HP * ROWB);
#pragma unroll
        for (int tap = 0; tap < ((PRG_C64_EXP & 1) ? 0 : 9); ++tap) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int cur = (tap * 4 + c) & 1, nxt = cur ^ 1;
            const int ntap = c == 3 ? tap + 1 : tap, nc = c == 3 ? 0 : c + 1;   // the call after this one
            if (ntap < 9) {
              const int toff = ((ntap / 3) * HP + (ntap % 3)) * ROWB + nc * 32;
              if constexpr ((PRG_C64_EXP & 64) != 0) {         // variant: the next call's four reads first, then the four MFMAs
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) fx[nxt][pt] = *reinterpret_cast<const c64_bf16x8*>(xb + pt * HP * ROWB + toff);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
                  acc[pt] = mma(wf[tap][c], fx[cur][pt], acc[pt]);
                __builtin_amdgcn_sched_barrier(0);
              } else if constexpr ((PRG_C64_EXP & 128) != 0) { // variant: leave the interleave to the compiler
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                  fx[nxt][pt] = *reinterpret_cast<const c64_bf16x8*>(xb + pt * HP * ROWB + toff);
                  acc[pt] = mma(wf[tap][c], fx[cur][pt], acc[pt]);
                }
              } else if constexpr ((PRG_C64_EXP & 2048) != 0) { // timing experiment: ALL the reads, but MFMA pairs share their operands (garbage results)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                  fx[nxt][pt] = *reinterpret_cast<const c64_bf16x8*>(xb + pt * HP * ROWB + toff);
                  asm volatile("" ::"v"(fx[cur][pt]));      // (the fragment is waited for where its MFMA would consume it)
                  acc[pt] = mma(wf[tap][c], fx[cur][pt & ~1], acc[pt]);
                  __builtin_amdgcn_sched_barrier(0);
                }
              } else if constexpr ((PRG_C64_EXP & 512) != 0) { // timing experiment: HALF the fragment reads (0.5 per MFMA; garbage results)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                  if ((pt & 1) == 0) fx[nxt][pt] = *reinterpret_cast<const c64_bf16x8*>(xb + pt * HP * ROWB + toff);
                  acc[pt] = mma(wf[tap][c], fx[cur][pt & ~1], acc[pt]);
                  __builtin_amdgcn_sched_barrier(0);
                }
              } else {
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                  fx[nxt][pt] = *reinterpret_cast<const c64_bf16x8*>(xb + pt * HP * ROWB + toff);
                  acc[pt] = mma(wf[tap][c], fx[cur][pt], acc[pt]);
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
            } else {
#pragma unroll
              for (int pt = 0; pt < 4; ++pt)
                acc[pt] = mma(wf[tap][c], fx[cur][pt], acc[pt]);
            }
          }
        }
      }
      // ---- epilogue: lane holds pixel (row 4 wm + pt, col l31), channels wn*32 + 8 q + 4 hi + {0..3}
      int tb, ty0, tx0;
      c64_tile(first + s * stride, tiles_x, tiles_y, tb, ty0, tx0);
      // the partial sums of the previous tile have left this CU before the barrier its ticket is taken behind
      if (fuse_stats && L.gn_tickets) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      c64_barrier();                                       // halo s+1 is written; nobody reads buffer s & 1 any more
      if (PRG_C64_EXP & 4) {                               // keep the accumulators live (no dead-code elimination of the MFMAs)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) asm volatile("" ::"v"(acc[pt]));
      }
      float V[8];                                          // [sum | sumsq][q]
      if (!(PRG_C64_EXP & 4)) {
        float bv[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr ((PRG_C64_EXP & 256) != 0) {
            const float4 b4 = *reinterpret_cast<const float4*>(biasp + 8 * q);
            bv[q][0] = b4.x; bv[q][1] = b4.y; bv[q][2] = b4.z; bv[q][3] = b4.w;
          } else {
            bv[q][0] = bv[q][1] = bv[q][2] = bv[q][3] = 0.0f;        // (the bias rode in as the accumulators' initial value)
          }
          V[q] = 0.0f;
          V[4 + q] = 0.0f;
        }
        char* const obase = reinterpret_cast<char*>(L.out) +
                            ((((size_t)tb * d.Hout + ty0 + wm * 4) * d.Wout + tx0 + l31) * 64 + wn * 32 + 8 * hi) * 2;
        const size_t orow = (size_t)d.Wout * 128;          // output bytes per image row
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
          uint32_t pk[8];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = (PRG_C64_EXP & 256) ? acc[pt][4 * q + r] + bv[q][r] : acc[pt][4 * q + r];
              V[q] += v[r];
              V[4 + q] = fmaf(v[r], v[r], V[4 + q]);
            }
            pk[2 * q] = O16 ? h16_pack(v[0], v[1]) : c64_pack(v[0], v[1]);
            pk[2 * q + 1] = O16 ? h16_pack(v[2], v[3]) : c64_pack(v[2], v[3]);
          }
          // lanes l and l + 32 hold the two channel quads of the same pixel and 8-channel chunk q: swapping the upper half
          // of chunk 2m with the lower half of chunk 2m+1 leaves lane half 0 with all 8 channels of chunk 2m, half 1 with
          // those of chunk 2m+1: one 16-byte store each
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * m], pk[4 * m + 2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * m + 1], pk[4 * m + 3], false, false);
            const c64_u32x4 o = {(uint32_t)s0[0], (uint32_t)s1[0], (uint32_t)s0[1], (uint32_t)s1[1]};
            *reinterpret_cast<c64_u32x4*>(obase + pt * orow + m * 32) = o;
          }
        }
      }
      if (fuse_stats && !(PRG_C64_EXP & 4)) {
        // 8 full-wave sums with a halving butterfly (conv_ws.hip's, one level shorter): fixed order, deterministic
        const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
        float A4[4], B2[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) A4[j] = (b0 ? V[4 + j] : V[j]) + c64_dpp<0xB1>(b0 ? V[j] : V[4 + j]);        // lane ^ 1
#pragma unroll
        for (int j = 0; j < 2; ++j) B2[j] = (b1 ? A4[2 + j] : A4[j]) + c64_dpp<0x4E>(b1 ? A4[j] : A4[2 + j]);    // lane ^ 2
        float D = (b2 ? B2[1] : B2[0]) + c64_swz<4>(b2 ? B2[0] : B2[1]);
        D += c64_swz<8>(D);
        D += c64_swz<16>(D);
        D += __shfl_xor(D, 32, 64);
        // lane (< 8) holds the wave total of value i = 4 b0 + 2 b1 + b2 = [sq][q]: 8-channel chunk q of this wave's 32
        // channels.  Fold the chunks of one group (per = cpg / 8 <= 4 inside a wave; 8 = both channel halves: two slabs).
        const int per = gn_per > 4 ? 4 : gn_per;
        if (per >= 2) D += c64_swz<4>(D);                  // q ^ 1  (lane bit 2)
        if (per >= 4) D += c64_dpp<0x4E>(D);               // q ^ 2  (lane bit 1)
        const int i = (lane & 1) * 4 + (lane & 2) + ((lane >> 2) & 1);
        const int q = i & 3;
        if (lane < 8 && (q & (per - 1)) == 0) {
          const int split_n = gn_per > 4 ? 2 : 1;
          const int nsplit = tiles_x * tiles_y * 2 * split_n;
          const int slab = (((ty0 / TH) * tiles_x + tx0 / TW) * 2 + wm) * split_n + (split_n == 2 ? wn : 0);
          const int grp = (wn * 4 + q) / gn_per;
          if (L.gn_acc) {
            // fixed-point accumulators (common.h): one no-return 64-bit integer atomic per wave total
            // (`which` made opaque here: its loop-invariant scale select otherwise lives in a VGPR across the MFMA loop — with the
            //  f16 epilogue that was the 257th register, and the spill's reload put an s_waitcnt vmcnt(0) behind the tile's stores)
            int which = i >> 2;
            asm volatile("" : "+v"(which));
            gn_acc_add(L.gn_acc, L.gn_groups, tb, grp, which, D);
          } else {
            // agent-scope (write-through, sc1) store: the workgroup that completes the image reads these from another CU
            __hip_atomic_store(&L.gn_partials[(((size_t)tb * nsplit + slab) * L.gn_groups + grp) * 2 + (i >> 2)], D,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    if (fuse_stats && L.gn_tickets) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (nsteps & 1) c64_barrier();                         // the producers' padding step
    c64_barrier();                                         // matches the producers' final barrier (ticket of the last tile)
    return;
  }

  // ---------------------------------------------------------------------------------------------------
  {
    const int ptid = tid - 256;
    const int slot = ptid & 7, row = ptid >> 3;           // 16-byte unit of a 128-byte pixel row; halo rows row + 32 k
    char* const ah = smem + row * ROWB + slot * 16;
    const int Hl = d.Hout, Wl = d.Wout;
    // Tile-independent halo geometry of this thread's units (round 3: the per-tile address arithmetic was ~27 VALU
    // instructions per unit — as many per tile as the consumers' whole epilogue).  Unit k sits at a fixed byte offset from
    // the tile's halo origin, and whether it is padding depends only on which image edges the tile touches: five bit masks
    // per thread, one AND-NOT per edge and tile.  The loads are raw BUFFER loads from a descriptor whose base lies one halo
    // row + one pixel before the tensor: every in-image unit has a non-negative offset, a padding unit gets offset
    // 0xffffffff and the hardware's range check returns zeros — no clamped stand-in address, no select after the load.
    int voffk[KU];
    unsigned m_valid = 0, m_top = 0, m_bot = 0, m_left = 0, m_right = 0;
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int hp = k * RPP + row;
      const int hy = hp / HP, hx = hp - hy * HP;
      const bool valid = hp < HALO;
      // (tile origins are multiples of 8 / 32: (y0 - 1 + hy) >> ups = (y0 >> ups) + ((hy - 1) >> ups), arithmetic shift)
      const int dy = (hy - 1) >> d.ups, dx = (hx - 1) >> d.ups;
      voffk[k] = ((dy + 1) * d.Win + dx + 1) * 128 + slot * 16;
      m_valid |= (valid ? 1u : 0u) << k;
      m_top |= (valid && hy == 0 ? 1u : 0u) << k;
      m_bot |= (valid && hy == TH + 1 ? 1u : 0u) << k;
      m_left |= (valid && hx == 0 ? 1u : 0u) << k;
      m_right |= (valid && hx == HP - 1 ? 1u : 0u) << k;
    }
    const unsigned pad_bytes = (unsigned)(d.Win + 1) * 128u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(L.src0)) - pad_bytes, 0,
        (int)((size_t)d.B * d.Hin * d.Win * 128 + pad_bytes), 0x00020000);
    // Two register sets: halo h lives in set h & 1.  The loads of halo s+2 are issued BEFORE halo s+1 is transformed and
    // written, so HBM requests are in flight during the prologue arithmetic (with one set the re-issue had to wait for
    // the transform: every CU computed while HBM idled, then every CU loaded — the two times added up).
    c64_u32x4 hA[KU], hB[KU];
    const int tpi = tiles_x * tiles_y;                     // tiles per image
    int done_in_img = 0;                                   // tiles of the current image this workgroup has finished
    unsigned okmask = 0, okmask_nxt = 0;
    float4 ca[2], cb[2], ca_n[2], cb_n[2];
    h16x2 ah2[4], bh2[4];                                  // PRO == 3: the coefficients as packed f16 channel pairs
    longlong2 fs_n = make_longlong2(0, 0);                  // PRO == 2: (sum, sumsq) of the 8-channel unit's group, fixed point
    auto issue = [&](int s, c64_u32x4(&h)[KU]) {           // loads of halo s (clamped to the last tile: harmless reloads)
      int b, y0, x0;
      c64_tile(first + min(s, nsteps - 1) * stride, tiles_x, tiles_y, b, y0, x0);
      okmask_nxt = m_valid & ~((y0 == 0 ? m_top : 0u) | (y0 + TH == Hl ? m_bot : 0u) | (x0 == 0 ? m_left : 0u) |
                               (x0 + TW == Wl ? m_right : 0u));
      const int soff = __builtin_amdgcn_readfirstlane((((b * d.Hin + (y0 >> d.ups)) * d.Win) + (x0 >> d.ups)) * 128);   // the SGPR offset
#pragma unroll
      for (int k = 0; k < KU; ++k)
        h[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((okmask_nxt >> k) & 1u) ? voffk[k] : -1, soff, 0);
      if constexpr (PRO == 1) {
        const float* pa = L.pro_a + (size_t)b * 64 + slot * 8;
        const float* pb = L.pro_b + (size_t)b * 64 + slot * 8;
        ca_n[0] = *reinterpret_cast<const float4*>(pa);
        ca_n[1] = *reinterpret_cast<const float4*>(pa + 4);
        cb_n[0] = *reinterpret_cast<const float4*>(pb);
        cb_n[1] = *reinterpret_cast<const float4*>(pb + 4);
      }
      if constexpr (PRO >= 2) {
        // the raw material of the coefficients: this unit's group statistics and its folded gain / bias (P, Q); the
        // arithmetic happens in adopt(), a step later, when the loads have long landed
        const GnFold& f = L.pro_fold;
        fs_n = *reinterpret_cast<const longlong2*>(f.acc + ((size_t)b * f.G + (slot * 8) / f.cpg) * 2);
        const float* pp = f.P + (size_t)b * f.pq_stride + slot * 8;
        const float* pq = f.Q + (size_t)b * f.pq_stride + slot * 8;
        ca_n[0] = *reinterpret_cast<const float4*>(pp);
        ca_n[1] = *reinterpret_cast<const float4*>(pp + 4);
        cb_n[0] = *reinterpret_cast<const float4*>(pq);
        cb_n[1] = *reinterpret_cast<const float4*>(pq + 4);
      }
    };
    auto write = [&](int s, c64_u32x4(&h)[KU]) {           // halo s -> LDS buffer s & 1
      char* dst = ah + (size_t)(s & 1) * AH_BYTES;
      const float a8[8] = {ca[0].x, ca[0].y, ca[0].z, ca[0].w, ca[1].x, ca[1].y, ca[1].z, ca[1].w};
      const float b8[8] = {cb[0].x, cb[0].y, cb[0].z, cb[0].w, cb[1].x, cb[1].y, cb[1].z, cb[1].w};
#pragma unroll
      for (int k = 0; k < KU; ++k) {
        c64_u32x4 v = h[k];
        if constexpr (PRO == 3 && !(PRG_C64_EXP & 32)) {
          v = h16_silu8(v, ah2, bh2);
        } else if constexpr (PRO && !(PRG_C64_EXP & 32)) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float lo = c64_silu(fmaf(c64_lo(v[j]), a8[2 * j], b8[2 * j]));
            const float hh = c64_silu(fmaf(c64_hi(v[j]), a8[2 * j + 1], b8[2 * j + 1]));
            v[j] = c64_pack(lo, hh);
          }
        }
        if constexpr (PRO != 0) {                          // (PRO == 0: the padding units arrived as zeros)
          if (!((okmask >> k) & 1u)) v = c64_u32x4{0u, 0u, 0u, 0u};
        }
        *reinterpret_cast<c64_u32x4*>(dst + k * RPP * ROWB) = v;
      }
    };
    auto adopt = [&]() {                                   // the validity mask and coefficients of the halo about to be written
      okmask = okmask_nxt;
      if constexpr (PRO == 1) {
        ca[0] = ca_n[0]; ca[1] = ca_n[1]; cb[0] = cb_n[0]; cb[1] = cb_n[1];
      }
      if constexpr (PRO >= 2) {                            // A = rstd P, B = Q - mean A
        float mean, rstd;
        gn_fold_stats_raw(fs_n.x, fs_n.y, L.pro_fold.inv_n, mean, rstd);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          ca[h2] = make_float4(rstd * ca_n[h2].x, rstd * ca_n[h2].y, rstd * ca_n[h2].z, rstd * ca_n[h2].w);
          cb[h2] = make_float4(fmaf(-mean, ca[h2].x, cb_n[h2].x), fmaf(-mean, ca[h2].y, cb_n[h2].y),
                               fmaf(-mean, ca[h2].z, cb_n[h2].z), fmaf(-mean, ca[h2].w, cb_n[h2].w));
        }
        if constexpr (PRO == 3) {
          typedef __attribute__((ext_vector_type(2))) float f32x2;
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            ah2[2 * h2] = __builtin_convertvector((f32x2){ca[h2].x, ca[h2].y}, h16x2);
            ah2[2 * h2 + 1] = __builtin_convertvector((f32x2){ca[h2].z, ca[h2].w}, h16x2);
            bh2[2 * h2] = __builtin_convertvector((f32x2){cb[h2].x, cb[h2].y}, h16x2);
            bh2[2 * h2 + 1] = __builtin_convertvector((f32x2){cb[h2].z, cb[h2].w}, h16x2);
          }
        }
      }
    };
    issue(0, hA);
    adopt();
    write(0, hA);
    issue(1, hB);
    c64_barrier();                                         // halo 0 ready
    auto ticket = [&](int s) {                             // tile s is complete and its partial sums have left their CU
      if (fuse_stats && L.gn_tickets && ptid < 64) {       // first producer wave: per-image arrival ticket
        const int t = first + s * stride;
        const int img = t / tpi;
        ++done_in_img;
        const bool leave = s + 1 == nsteps || (t + stride) / tpi != img;
        if (leave) {
          int old = 0;
          if (ptid == 0) old = __hip_atomic_fetch_add(L.gn_tickets + img, done_in_img, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          old = __builtin_amdgcn_readfirstlane(old);
          if (old + done_in_img == tpi) {                  // this workgroup completed image `img`: fold its statistics
            if (ptid == 0) __hip_atomic_store(L.gn_tickets + img, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c64_fold_coefficients(L, img, tpi * 2 * (((64 / L.gn_groups) >> 3) > 4 ? 2 : 1), ptid);
          }
          done_in_img = 0;
        }
      }
    };
    // Two steps per iteration (the register set of a halo is a compile-time choice); an odd step count is padded with
    // one step of clamped reloads that nobody reads, so the loop body stays branch-free between a load and its use and
    // the compiler's s_waitcnt vmcnt(N) counts stay exact.
    const int n2 = (nsteps + 1) & ~1;
#pragma unroll 1
    for (int s = 0; s < n2; s += 2) {
      // during the consumers' MFMAs of step s and their epilogue of step s-1: the loads of halo s+2, then halo s+1 into
      // the buffer they left before barrier s-1
      adopt();
      issue(s + 2, hA);
      write(s + 1, hB);
      c64_barrier();                                       // barrier s
      if (s > 0) ticket(s - 1);                            // the consumers stored tile s-1's partial sums before this barrier
      adopt();
      issue(s + 3, hB);
      write(s + 2, hA);
      c64_barrier();                                       // barrier s+1
      if (s < nsteps) ticket(s);
    }
    c64_barrier();
    if (n2 == nsteps) ticket(nsteps - 1);
  }
}

}  // namespace

int try_launch_conv3x3_c64w(const ConvLaunch<bf16_t>& L, hipStream_t s, int* gn_nsplit_out, int* coef_done, int* acc_done);   // conv_c64w.hip

// Returns 1 when it launched, 0 when the shape is not covered (caller falls back), negative on error.
int try_launch_conv3x3_c64(const ConvLaunch<bf16_t>& L, hipStream_t s, int* gn_nsplit_out, int* coef_done, int* acc_done) {
  static const int enabled = [] {
    const char* e = std::getenv("PRG_CONV_C64");
    return e ? std::atoi(e) : 1;
  }();
  if (!enabled) return 0;
  {
    // round 6: the one-wave-per-SIMD form (conv_c64w.hip: 64 channels of weights per wave, 0.5 fragment reads per MFMA) takes the
    // launches it covers — probe calls included, so that conv_h16_pair_ok sees the dispatch that will run
    const int r = try_launch_conv3x3_c64w(L, s, gn_nsplit_out, coef_done, acc_done);
    if (r != 0) return r;
  }
  const ConvDesc& d = L.d;
  if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1)) return 0;
  if (d.C0 != 64 || d.C1 != 0 || d.Cout != 64 || d.CoutPad != 64 || d.kchunks != 2) return 0;
  if (L.residual || !L.bias) return 0;
  if (d.ups) return 0;                                       // (no 64 -> 64 Upsample conv exists in the networks: not a tested shape)
  if (d.Wout % TW || d.Hout % TH) return 0;
  // the producers address the input through ONE buffer descriptor with 32-bit byte offsets (pixel index * 128 in an SGPR,
  // num_records as an int): beyond 2 GiB of input the generic kernel (64-bit addressing) runs instead — B = 256 at 256x256
  if ((size_t)d.B * d.Hin * d.Win * 128 + 4096 >= ((size_t)1 << 31)) return 0;
  const int tiles_x = d.Wout / TW, tiles_y = d.Hout / TH;
  const int total = tiles_x * tiles_y * d.B;
  const int num_cus = device_cu_count();
  if (num_cus <= 0) return 0;
  int grid = (total < num_cus ? total : num_cus) & ~7;       // multiple of 8: XCD-contiguous tile runs
  if (grid < 8) return 0;                                    // tiny launches stay on the generic kernels
  const int cpg = L.gn_groups > 0 ? 64 / L.gn_groups : 0;
  const int split_n = cpg > 32 ? 2 : 1;
  int fuse = L.gn_partials != nullptr && cpg % 8 == 0 && cpg <= 64 && (cpg & (cpg - 1)) == 0 &&
             tiles_x * tiles_y * 2 * split_n <= kGnMaxSplit;
  if (L.gn_partials && !fuse) return 0;
  static DeviceOnce attr_done[5];     // zero-initialised; atomic: lanes launch from several host threads
  const GnFold& pf = L.pro_fold;
  const bool fold_ok = pf.acc && pf.P && pf.Q && pf.G * pf.cpg == 64 && pf.cpg % 8 == 0;
  if (pf.acc && !fold_ok) return 0;
  if (L.in_f16 && !(fold_ok && L.w_f16)) return 0;           // h16 input: only through the folded prologue, with f16 weights
  const int pro = L.in_f16 ? 3 : fold_ok ? 2 : (L.pro_a ? 1 : 0);
  if (L.out_f16 && pro != 0) return 0;                       // h16 output: conv1 of a ResnetBlock (no prologue)
  const int variant = L.out_f16 ? 4 : pro;
  const void* const fns[5] = {reinterpret_cast<const void*>(&conv3x3_c64_kernel<0, false>), reinterpret_cast<const void*>(&conv3x3_c64_kernel<1, false>),
                              reinterpret_cast<const void*>(&conv3x3_c64_kernel<2, false>), reinterpret_cast<const void*>(&conv3x3_c64_kernel<3, false>),
                              reinterpret_cast<const void*>(&conv3x3_c64_kernel<0, true>)};
  if (!attr_done[variant].done()) {
    hipError_t e = hipFuncSetAttribute(fns[variant], hipFuncAttributeMaxDynamicSharedMemorySize, (int)C64_LDS);
    if (e != hipSuccess) return fail(PRG_E_HIP, std::string("hipFuncSetAttribute(c64 conv): ") + hipGetErrorString(e));
    attr_done[variant].mark();
  }
  if (gn_nsplit_out) *gn_nsplit_out = fuse ? tiles_x * tiles_y * 2 * split_n : 0;
  // the image-completing workgroup folds the statistics into the coefficients itself (no gn_coeff launch)
  const bool use_acc = fuse && L.gn_acc != nullptr;          // fixed-point accumulators instead of slabs (+ no in-kernel ticket fold)
  const bool fold = fuse && !use_acc && L.gn_tickets && L.gn_coef_a && L.gn_coef_b && L.gn.gamma && L.gn.beta && d.B <= kMaxTicketImages;
  ConvLaunch<bf16_t> Lk = L;
  if (!fold) Lk.gn_tickets = nullptr;
  if (!use_acc) Lk.gn_acc = nullptr;
  if (coef_done) *coef_done = fold ? 1 : 0;
  if (acc_done) *acc_done = use_acc ? 1 : 0;
  // interleaved tile runs are ~2 % faster (neighbouring tiles run at the same time on neighbouring CUs of the XCD); the
  // coefficient fold wants contiguous runs (one ticket per workgroup and image).  PRG_C64_INTERLEAVE=0/1 overrides.
  static const int interleave_env = [] { const char* e = std::getenv("PRG_C64_INTERLEAVE"); return e ? std::atoi(e) : -1; }();
  const int interleave = interleave_env >= 0 ? interleave_env : (fold ? 0 : 1);
  const int flags = (fuse ? 1 : 0) | (interleave ? 2 : 0);
  if (L.probe) return 1;
  if (variant == 4) conv3x3_c64_kernel<0, true><<<dim3(grid), 512, C64_LDS, s>>>(Lk, tiles_x, tiles_y, flags);
  else if (pro == 3) conv3x3_c64_kernel<3, false><<<dim3(grid), 512, C64_LDS, s>>>(Lk, tiles_x, tiles_y, flags);
  else if (pro == 2) conv3x3_c64_kernel<2, false><<<dim3(grid), 512, C64_LDS, s>>>(Lk, tiles_x, tiles_y, flags);
  else if (pro == 1) conv3x3_c64_kernel<1, false><<<dim3(grid), 512, C64_LDS, s>>>(Lk, tiles_x, tiles_y, flags);
  else conv3x3_c64_kernel<0, false><<<dim3(grid), 512, C64_LDS, s>>>(Lk, tiles_x, tiles_y, flags);
  PRG_LAUNCH_CHECK();
  return 1;
}

}  // namespace prg
