// conv_c64w.hip — the weights-stationary 64 -> 64 3x3 convolution (bf16 throughput path) with ONE WAVE PER SIMD (round 6).
//
// conv3x3_c64_kernel (conv_c64.hip) gives a consumer wave 32 output channels: their 9 x 64 weights fill 144 of its 256 registers, and
// (until the row-reuse loop of round 6) every pixel fragment it read from LDS fed ONE MFMA, the two channel-half waves reading the same
// pixels twice.  A timing experiment that halved those reads ran the level-0 launch in 54.5 us instead of 73.0, was read as "bound by LDS
// operand traffic" and motivated this kernel.  The experiment was invalid — its paired accumulator chains were identical and hipcc dropped
// half the MFMAs; repeated correctly, half the reads are worth 3-4.5 % (profiles/r06_c64_half_reads_bound.txt).  The kernel is kept as the
// measured record of the one-wave-per-SIMD form on the bf16 path.
// Here a 256-thread workgroup owns a CU with one 512-register wave per SIMD:
//   * a wave keeps the weights of ALL 64 output channels (72 A fragments, 288 registers: hipcc places them across the VGPR and
//     AccVGPR halves of the unified file) and owns 64 pixels (two rows of the 8 x 32 tile): every pixel fragment feeds TWO MFMAs
//     (0.5 reads per MFMA), nobody reads a pixel twice;
//   * no producer waves: each thread stages its 11 halo units of the NEXT tile inside the MFMA stream (one unit per three k-steps:
//     transform + ds_write of halo s+1, then the buffer load of the same unit of halo s+2 into the freed registers), written as
//     pinned issue slots (MFMA, side operation, sched_barrier) like conv_split512.hip;
//   * one barrier per tile for four waves, between the MFMAs and the epilogue, as before.
// Per accumulator the MFMA sequence is conv3x3_c64_kernel's (bias first, taps and k-steps in order), the epilogue expression too:
// outputs are bit-identical; the GroupNorm statistics are per-wave float totals added into the fixed-point accumulators, and a
// wave now covers 64 pixels x 64 channels instead of 128 x 32, so the statistics differ in the last bits (as between any two of the
// bf16 kernels).  Only the accumulator form of the statistics (ConvLaunch::gn_acc) is implemented; other launches keep conv_c64.
// OUTCOME (profiles/r06_c64w_ablations.txt): 76-79 us against conv3x3_c64_kernel's 71-73 at the level-0 launch in every form tried — the
// side work does not hide behind the MFMAs of a single in-order stream.  Opt-in (PRG_CONV_C64W=1), off by default.
#include <atomic>
#include <cstdlib>

#include "conv.h"
#include "conv_split_ablate.h"

namespace prg {

typedef __attribute__((ext_vector_type(8))) __bf16 c64w_bf16x8;
typedef __attribute__((ext_vector_type(16))) float c64w_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int c64w_u32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 c64w_f16x8;

namespace {

// Timing experiments only (results become garbage; tools/gpu_c64w_exp.sh): -DPRG_C64W_EXP=1 no MFMAs, 2 no staging (halo loads /
// transforms / LDS writes inside the loop), 4 no epilogue (stores, statistics), 8 no fragment reads
#ifndef PRG_C64W_EXP
#define PRG_C64W_EXP 0
#endif

template <int N>
struct WI {
  static constexpr int value = N;
};
template <int N, int I = 0, typename F>
__device__ __forceinline__ void w_static_for(F&& f) {
  if constexpr (I < N) {
    f(WI<I>());
    w_static_for<N, I + 1>(f);
  }
}

constexpr int TH = 8, TW = 32, HP = TW + 2, HALO = (TH + 2) * HP;   // 340 halo rows
constexpr int ROWB = 144;                                              // padded LDS row (64 bf16 = 128 B + 16)
constexpr int NPT = 256;                                               // every thread stages
constexpr int RPP = NPT / 8;                                           // halo rows per pass
constexpr int KU = (HALO + RPP - 1) / RPP;                             // 11 units per thread
constexpr size_t AH_BYTES = (size_t)KU * RPP * ROWB;                   // 352 rows: the units past the halo end land in spare rows
constexpr int NLT = 2;                                                 // taps (the last NLT) whose weights live in LDS instead of registers
constexpr size_t WL_BYTES = (size_t)NLT * 4 * 2 * 1024;                // [tap][k-step][channel half][lane] x 16 B
constexpr size_t C64W_LDS = 2 * AH_BYTES + 64 * sizeof(float) + WL_BYTES;   // + the bias + those weights

__device__ inline uint32_t c64w_pack(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
template <int CTRL>
__device__ inline float c64w_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int XOR>
__device__ inline float c64w_swz(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (XOR << 10) | 0x1F));
}
__device__ inline void c64w_tile(int t, int tiles_x, int tiles_y, int& b, int& y0, int& x0) {
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y;
  b = t / tiles_y;
  y0 = ty * TH;
  x0 = tx * TW;
}
// one LDS-visibility point: this wave's LDS operations are done, then the workgroup barrier.  Never waits for VMEM: the halo loads
// and the epilogue's stores stay in flight across it.
__device__ __forceinline__ void c64w_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// PRO 0: no prologue; 3: coefficients folded here from pro_fold on an f16 input with f16 weights (the h16 format, conv.h).
// O16: f16 output.
template <int PRO, bool O16>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_c64w_kernel(
    const ConvLaunch<bf16_t> L, const int tiles_x, const int tiles_y, const int flags) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int fuse_stats = flags & 1;
  const ConvDesc& d = L.d;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int total = tiles_x * tiles_y * d.B;
  // XCD-contiguous tile runs (gridDim.x is a multiple of 8), workgroups of an XCD interleaved inside its run (conv_c64.hip)
  const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int per = (total + 7) / 8;
  const int lo_t = min(xcd * per, total), hi_t = min((xcd + 1) * per, total);
  const int nx = hi_t - lo_t;
  const int first = lo_t + widx;
  const int nsteps = widx < nx ? (nx - widx + wpx - 1) / wpx : 0;
  const int stride = wpx;
  if (nsteps == 0) return;

  const int l31 = lane & 31, hi = lane >> 5;
  // ---- weights of channels h2 * 32 + l31: 9 taps x 4 k-steps; lane half hi takes k = 16 c + 8 hi .. + 7.  Taps 0 .. 8 - NLT stay in
  //      registers for the whole launch (224); the fragments of the last NLT taps are identical in every wave and live in LDS (16 KB),
  //      read two k-steps ahead like the pixels (0.72 LDS reads per MFMA instead of 0.5): with all 288 resident the two-phase pipeline
  //      below spilled, and a spill reload is a scratch load behind `s_waitcnt vmcnt(0)` — it waits for every halo load in flight ----
  constexpr int NRT = 9 - NLT;
  const bf16_t* const wsrc = PRO == 3 ? reinterpret_cast<const bf16_t*>(L.w_f16) : L.w;
  auto wptr = [&](int h2, int tap, int c) {
    return wsrc + ((size_t)(tap * d.kchunks + (c >> 1)) * d.CoutPad + h2 * 32 + l31) * 32 + (c & 1) * 16 + hi * 8;
  };
  c64w_bf16x8 wf[2][NRT][4];
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
    for (int tap = 0; tap < NRT; ++tap)
#pragma unroll
      for (int c = 0; c < 4; ++c) wf[h2][tap][c] = *reinterpret_cast<const c64w_bf16x8*>(wptr(h2, tap, c));
  char* const wlds = smem + 2 * AH_BYTES + 64 * sizeof(float);
#pragma unroll
  for (int f = 0; f < NLT * 4 * 2 / 4; ++f) {              // wave w stores fragments w, w + 4, ...
    const int idx = wave + 4 * f, h2 = idx & 1, c = (idx >> 1) & 3, tl = idx >> 3;
    *reinterpret_cast<c64w_bf16x8*>(wlds + (size_t)idx * 1024 + lane * 16) = *reinterpret_cast<const c64w_bf16x8*>(wptr(h2, NRT + tl, c));
  }
  float* const bias_lds = reinterpret_cast<float*>(smem + 2 * AH_BYTES);
  if (tid < 64) bias_lds[tid] = L.bias[tid];               // visible after the prologue barrier
  const int gn_per = fuse_stats ? (64 / L.gn_groups) >> 3 : 1;   // 8-channel chunks per group (1, 2, 4 or 8)
  // LDS byte offset of pixel (row 2 wave + pt, column l31), tap (0,0), k-step 0: rows are HP * ROWB apart
  const unsigned x0off = (unsigned)(((wave * 2) * HP + l31) * ROWB + hi * 16);
  auto mma = [](const c64w_bf16x8& a, const c64w_bf16x8& b, const c64w_f32x16& c) -> c64w_f32x16 {
    if constexpr ((PRG_C64W_EXP & 1) != 0) {
      asm volatile("" ::"v"(a), "v"(b));
      return c;
    }
    if constexpr (PRO == 3)
      return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c64w_f16x8, a), __builtin_bit_cast(c64w_f16x8, b), c, 0, 0, 0);
    else
      return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  };

  // ---- halo staging: thread = (16-byte unit `slot` of a 128-byte pixel row, halo rows row + 32 k); geometry as conv_c64.hip ----
  const int slot = tid & 7, row = tid >> 3;
  char* const ah = smem + row * ROWB + slot * 16;
  const int Hl = d.Hout, Wl = d.Wout;
  int voffk[KU];
  unsigned m_valid = 0, m_top = 0, m_bot = 0, m_left = 0, m_right = 0;
#pragma unroll
  for (int k = 0; k < KU; ++k) {
    const int hp = k * RPP + row;
    const int hy = hp / HP, hx = hp - hy * HP;
    const bool valid = hp < HALO;
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
  // NSET register sets, halo x in set x % NSET: unit k of halo s+1 is transformed and written during tile s, then the same registers
  // take unit k of halo s+1+NSET.  Two sets (the plain forms) keep TWO halos = 90 KB per CU in flight like conv3x3_c64_kernel's
  // producers — with one, a CU's 45 KB could not cover the ~4500 cycles a halo takes at the ~10 B/cycle/CU HBM rate, and here a late
  // load stalls the wave's own MFMA stream; the h16 prologue form has no registers for a second set.
  constexpr int NSET = 1;   // (two sets were measured equal on the first form of the kernel: profiles/r06_c64w_ablations.txt)
  c64w_u32x4 h[NSET][KU];
  unsigned okmask = 0, okmask_nxt = 0, okq[NSET];
  int soff_nxt = 0;
  float4 ca_n[2], cb_n[2];
  h16x2 ah2[4], bh2[4];                                    // PRO == 3: the coefficients as packed f16 channel pairs
  longlong2 fs_n = make_longlong2(0, 0);
  // coordinates, padding mask and (PRO) coefficient loads of halo s (clamped to the last tile: harmless reloads)
  auto issue_head = [&](int s) __attribute__((always_inline)) {
    int b, y0, x0;
    c64w_tile(first + min(s, nsteps - 1) * stride, tiles_x, tiles_y, b, y0, x0);
    okmask_nxt = m_valid & ~((y0 == 0 ? m_top : 0u) | (y0 + TH == Hl ? m_bot : 0u) | (x0 == 0 ? m_left : 0u) |
                             (x0 + TW == Wl ? m_right : 0u));
    soff_nxt = __builtin_amdgcn_readfirstlane((((b * d.Hin + (y0 >> d.ups)) * d.Win) + (x0 >> d.ups)) * 128);
    if constexpr (PRO == 3) {
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
  auto issue_unit = [&](auto SET, auto K) __attribute__((always_inline)) {
    constexpr int k = decltype(K)::value, st = decltype(SET)::value;
    h[st][k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((okmask_nxt >> k) & 1u) ? voffk[k] : -1, soff_nxt, 0);
  };
  auto adopt = [&]() __attribute__((always_inline)) {                                     // the validity mask and coefficients of the halo about to be written
    okmask = okq[0];
    if constexpr (NSET == 2) okq[0] = okq[1];
    if constexpr (PRO == 3) {                              // A = rstd P, B = Q - mean A
      float mean, rstd;
      gn_fold_stats_raw(fs_n.x, fs_n.y, L.pro_fold.inv_n, mean, rstd);
      typedef __attribute__((ext_vector_type(2))) float f32x2;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const float4 a = make_float4(rstd * ca_n[h2].x, rstd * ca_n[h2].y, rstd * ca_n[h2].z, rstd * ca_n[h2].w);
        const float4 bq = make_float4(fmaf(-mean, a.x, cb_n[h2].x), fmaf(-mean, a.y, cb_n[h2].y), fmaf(-mean, a.z, cb_n[h2].z),
                                      fmaf(-mean, a.w, cb_n[h2].w));
        ah2[2 * h2] = __builtin_convertvector((f32x2){a.x, a.y}, h16x2);
        ah2[2 * h2 + 1] = __builtin_convertvector((f32x2){a.z, a.w}, h16x2);
        bh2[2 * h2] = __builtin_convertvector((f32x2){bq.x, bq.y}, h16x2);
        bh2[2 * h2 + 1] = __builtin_convertvector((f32x2){bq.z, bq.w}, h16x2);
      }
    }
  };
  auto write_unit = [&](int buf, auto SET, auto K) __attribute__((always_inline)) {       // unit k of the adopted halo -> LDS buffer `buf`
    constexpr int k = decltype(K)::value, st = decltype(SET)::value;
    c64w_u32x4 v = h[st][k];
    if constexpr (PRO == 3) {
      v = h16_silu8(v, ah2, bh2);
      if (!((okmask >> k) & 1u)) v = c64w_u32x4{0u, 0u, 0u, 0u};   // (PRO == 0: the padding units arrived as zeros)
    }
    *reinterpret_cast<c64w_u32x4*>(ah + (size_t)buf * AH_BYTES + k * RPP * ROWB) = v;
  };

  // ---- prologue: halo 0 written, halos 1 .. NSET in flight ----
  issue_head(0);
  w_static_for<KU>([&](auto K) { issue_unit(WI<0>(), K); });
  okq[0] = okmask_nxt;
  adopt();
  w_static_for<KU>([&](auto K) { write_unit(0, WI<0>(), K); });
  issue_head(1);
  w_static_for<KU>([&](auto K) { issue_unit(WI<(NSET == 2 ? 1 : 0)>(), K); });
  okq[0] = okmask_nxt;
  if constexpr (NSET == 2) {
    issue_head(2);
    w_static_for<KU>([&](auto K) { issue_unit(WI<0>(), K); });
    okq[1] = okmask_nxt;
  }
  c64w_barrier();                                          // halo 0 and the bias are in LDS

  // ---- the tile pipeline ----------------------------------------------------------------------------------------------------
  // A tile is TWO phases of 36 k-steps, one per pixel row of this wave (72 MFMAs each: channel halves alternate, so back-to-back
  // MFMAs never share an accumulator).  Every k-step is two pinned issue slots:
  //     MFMA (row, half 0) + the fragment read FD k-steps ahead        MFMA (row, half 1) + one SIDE piece
  // and the side pieces of a phase are the epilogue of the row the PREVIOUS phase finished (bias rode in with the accumulators:
  // pack, statistics, lane-half swap, 16-byte stores — 12 pieces; after row 1 the statistics' butterfly and the accumulator atomics
  // — 6 pieces) and this thread's halo units of the next tile (transform + ds_write, then the reload of the same registers).  Measured
  // on the first form of this kernel (whole-tile MFMA phase, then a serial epilogue; -DPRG_C64W_EXP ablations,
  // profiles/r06_c64w_ablations.txt): MFMAs alone 57 us, everything but the MFMAs 57 us, serial sum 76.7 us at the level-0 launch
  // (conv3x3_c64_kernel: 73.0; its MFMA phase and its epilogue add up the same way).
  constexpr int FD = 2, FR = FD + 1;                       // fragment ring: read FD k-steps (2 FD MFMAs) ahead
  c64w_bf16x8 fx[FR], wl[FR][2];                           // pixels; the LDS-resident taps' weights (both channel halves)
  auto frag_off = [](int ks) { const int tap = ks >> 2, c = ks & 3; return ((tap / 3) * HP + (tap % 3)) * ROWB + c * 32; };
  c64w_f32x16 acc[2][2];                                   // [pixel row][channel half]; row r is busy from its phase to the end of its epilogue
  float V[2][8];                                           // [channel half][sum | sumsq][q]: statistics of the tile being stored
  uint32_t pk[8];
  float A4[4], B2[2], Dred = 0.0f;
  char* obase_e = nullptr;                                 // output address / image of the tile whose rows are being stored
  int tb_e = 0;
  const size_t orow = (size_t)d.Wout * 128;                // output bytes per image row
  auto acc_init = [&](auto PT) __attribute__((always_inline)) {                           // the accumulators START at the bias (conv_c64.hip, round 5)
    constexpr int pt = decltype(PT)::value;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const float* const biasp = bias_lds + h2 * 32 + 4 * hi;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(biasp + 8 * q);
        acc[pt][h2][4 * q] = b4.x; acc[pt][h2][4 * q + 1] = b4.y; acc[pt][h2][4 * q + 2] = b4.z; acc[pt][h2][4 * q + 3] = b4.w;
      }
    }
  };
  // epilogue piece P (0 .. 11) of pixel row PT: per channel half four (pack + statistics) pieces and two (swap + store) pieces.
  // Lane holds pixel (row 2 wave + pt, col l31), channels h2 * 32 + 8 q + 4 hi + {0..3} (conv3x3_c64_kernel's epilogue).
  auto epi_piece = [&](auto PT, auto P) __attribute__((always_inline)) {
    constexpr int pt = decltype(PT)::value, p = decltype(P)::value, h2 = p / 6, r6 = p % 6;
    if constexpr ((PRG_C64W_EXP & 4) != 0) {
      if constexpr (p == 0) asm volatile("" ::"v"(acc[pt][0]), "v"(acc[pt][1]));
      return;
    }
    if constexpr (r6 < 4) {
      constexpr int q = r6;
      if constexpr (pt == 0 && q == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) V[h2][j] = 0.0f;
      }
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[pt][h2][4 * q + r];
        V[h2][q] += v[r];
        V[h2][4 + q] = fmaf(v[r], v[r], V[h2][4 + q]);
      }
      pk[2 * q] = O16 ? h16_pack(v[0], v[1]) : c64w_pack(v[0], v[1]);
      pk[2 * q + 1] = O16 ? h16_pack(v[2], v[3]) : c64w_pack(v[2], v[3]);
    } else {
      // lanes l and l + 32 hold the two channel quads of the same pixel and 8-channel chunk q: after the swap lane half 0 has all 8
      // channels of chunk 2m, half 1 those of chunk 2m+1: one 16-byte store each
      constexpr int m = r6 - 4;
      const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * m], pk[4 * m + 2], false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * m + 1], pk[4 * m + 3], false, false);
      const c64w_u32x4 o = {(uint32_t)s0[0], (uint32_t)s1[0], (uint32_t)s0[1], (uint32_t)s1[1]};
      *reinterpret_cast<c64w_u32x4*>(obase_e + h2 * 64 + pt * orow + m * 32) = o;
    }
  };
  // statistics piece P (0 .. 5) of the stored tile: per channel half the halving butterfly of conv3x3_c64_kernel in three pieces
  // (fixed order, deterministic), then one no-return 64-bit integer atomic per wave total
  auto stat_piece = [&](auto P) __attribute__((always_inline)) {
    constexpr int p = decltype(P)::value, h2 = p / 3, r3 = p % 3;
    if constexpr ((PRG_C64W_EXP & 4) != 0) return;
    if (!fuse_stats) return;
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    if constexpr (r3 == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) A4[j] = (b0 ? V[h2][4 + j] : V[h2][j]) + c64w_dpp<0xB1>(b0 ? V[h2][j] : V[h2][4 + j]);        // lane ^ 1
#pragma unroll
      for (int j = 0; j < 2; ++j) B2[j] = (b1 ? A4[2 + j] : A4[j]) + c64w_dpp<0x4E>(b1 ? A4[j] : A4[2 + j]);                    // lane ^ 2
    } else if constexpr (r3 == 1) {
      float D = (b2 ? B2[1] : B2[0]) + c64w_swz<4>(b2 ? B2[0] : B2[1]);
      D += c64w_swz<8>(D);
      D += c64w_swz<16>(D);
      D += __shfl_xor(D, 32, 64);
      Dred = D;
    } else {
      // lane (< 8) holds the wave total of value i = 4 b0 + 2 b1 + b2 = [sq][q]: 8-channel chunk q of this channel half.
      // Fold the chunks of one group (per = cpg / 8 <= 4 inside a half; 8 = both halves: each half adds its own total).
      float D = Dred;
      const int per = gn_per > 4 ? 4 : gn_per;
      if (per >= 2) D += c64w_swz<4>(D);                   // q ^ 1  (lane bit 2)
      if (per >= 4) D += c64w_dpp<0x4E>(D);                // q ^ 2  (lane bit 1)
      const int i = (lane & 1) * 4 + (lane & 2) + ((lane >> 2) & 1);
      const int q = i & 3;
      if (lane < 8 && (q & (per - 1)) == 0) {
        const int grp = (h2 * 4 + q) / gn_per;
        int which = i >> 2;
        asm volatile("" : "+v"(which));
        gn_acc_add(L.gn_acc, L.gn_groups, tb_e, grp, which, D);
      }
    }
  };
  // one phase: pixel row PT of tile s (halo buffer s & 1); `prev` = the previous phase left a row to store
  auto phase = [&](auto PT, auto SETW, int s, auto PREV) __attribute__((always_inline)) {
    constexpr int pt = decltype(PT)::value;
    constexpr bool prev = decltype(PREV)::value != 0;
    const char* const xb = smem + (size_t)(s & 1) * AH_BYTES + x0off + pt * HP * ROWB;
    const char* const xo = smem + (size_t)(s & 1) * AH_BYTES + x0off + (1 - pt) * HP * ROWB;   // (phase 0 prefetches row 1's first fragments)
    const int nbuf = (s + 1) & 1;
    acc_init(PT);
    __builtin_amdgcn_sched_barrier(0);
    w_static_for<36>([&](auto KS) {
      constexpr int ks = decltype(KS)::value, tap = ks >> 2, c = ks & 3, cur = ks % FR, nks = ks + FD, nxt = nks % FR;
      acc[pt][0] = mma(tap < NRT ? wf[0][tap < NRT ? tap : 0][c] : wl[cur][0], fx[cur], acc[pt][0]);
      if constexpr (!(PRG_C64W_EXP & 8)) {
        if constexpr (nks < 36) fx[nxt] = *reinterpret_cast<const c64w_bf16x8*>(xb + frag_off(nks < 36 ? nks : 0));
        else if constexpr (pt == 0) fx[nxt] = *reinterpret_cast<const c64w_bf16x8*>(xo + frag_off(nks - 36));
        if constexpr (nks < 36 && (nks >> 2) >= NRT) {
          constexpr int widx = (((nks >> 2) - NRT) * 4 + (nks & 3)) * 2;
          wl[nxt][0] = *reinterpret_cast<const c64w_bf16x8*>(wlds + (size_t)widx * 1024 + lane * 16);
          wl[nxt][1] = *reinterpret_cast<const c64w_bf16x8*>(wlds + (size_t)(widx + 1) * 1024 + lane * 16);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[pt][1] = mma(tap < NRT ? wf[1][tap < NRT ? tap : 0][c] : wl[cur][1], fx[cur], acc[pt][1]);
      // side pieces.  k-steps 0-11: the other row's epilogue; phase 0 then has the statistics (12-17) and halo units 0-5 (18-35: write
      // at 18 + 3 j, reload at 19 + 3 j); phase 1 has halo units 6-10 (12 + 3 j, 13 + 3 j)
      if constexpr (ks < 12) {
        if constexpr (prev) epi_piece(WI<1 - pt>(), KS);
      } else if constexpr (pt == 0 && ks < 18) {
        if constexpr (prev) stat_piece(WI<(ks < 18 ? ks - 12 : 0)>());
      } else if constexpr (!(PRG_C64W_EXP & 2)) {
        constexpr int base = pt == 0 ? 18 : 12, u0 = pt == 0 ? 0 : 6, j = (ks - base) / 3, r = (ks - base) % 3, u = u0 + j;
        if constexpr (u < KU && r == 0) write_unit(nbuf, SETW, WI<(u < KU ? u : 0)>());
        if constexpr (u < KU && r == 1) issue_unit(SETW, WI<(u < KU ? u : 0)>());
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  auto tile = [&](int s, auto SETW, auto FIRST) __attribute__((always_inline)) {          // halo s+1 lives in register set SETW; FIRST: no previous tile to store
    adopt();                                               // halo s+1: loaded NSET tiles ago, written during this tile
    issue_head(s + 1 + NSET);
    phase(WI<0>(), SETW, s, WI<(decltype(FIRST)::value ? 0 : 1)>());   // row 0; stores row 1 of tile s - 1 and finishes its statistics
    int tb, ty0, tx0;
    c64w_tile(first + s * stride, tiles_x, tiles_y, tb, ty0, tx0);
    tb_e = tb;
    obase_e = reinterpret_cast<char*>(L.out) + ((((size_t)tb * d.Hout + ty0 + wave * 2) * d.Wout + tx0 + l31) * 64 + 8 * hi) * 2;
    phase(WI<1>(), SETW, s, WI<1>());                      // row 1; stores row 0 of this tile
    c64w_barrier();                                        // halo s+1 is written; nobody reads buffer s & 1 any more
    if (s + 1 < nsteps) {                                  // the next tile's first fragments (row 0)
      const char* const xn = smem + (size_t)((s + 1) & 1) * AH_BYTES + x0off;
#pragma unroll
      for (int j = 0; j < FD; ++j) fx[j] = *reinterpret_cast<const c64w_bf16x8*>(xn + frag_off(j));
    }
    if constexpr (NSET == 2) okq[1] = okmask_nxt; else okq[0] = okmask_nxt;
  };
#pragma unroll
  for (int j = 0; j < FD; ++j) fx[j] = *reinterpret_cast<const c64w_bf16x8*>(smem + x0off + frag_off(j));
  static_assert(NSET == 1, "the tile loop below walks one register set");
  tile(0, WI<0>(), WI<1>());
  for (int s = 1; s < nsteps; ++s) tile(s, WI<0>(), WI<0>());
  // the last tile's second row and statistics
  w_static_for<12>([&](auto P) { epi_piece(WI<1>(), P); });
  w_static_for<6>([&](auto P) { stat_piece(P); });
}

}  // namespace

// Returns 1 when it launched, 0 when the shape / mode is not covered (the caller goes on to conv3x3_c64_kernel), negative on error.
// PRG_CONV_C64W: 0 (default) off — the same-box A/B (profiles/r06_c64w_ablations.txt) has this kernel at 76-79 us against
// conv3x3_c64_kernel's 71-73 at the level-0 launch; 1 selects it (tests/test_gpu_parity.py::test_c64w_kernel_matches_the_c64_kernel).
int try_launch_conv3x3_c64w(const ConvLaunch<bf16_t>& L, hipStream_t s, int* gn_nsplit_out, int* coef_done, int* acc_done) {
  static const int enabled = [] { const char* e = std::getenv("PRG_CONV_C64W"); return e ? std::atoi(e) : 0; }();
  if (!enabled) return 0;
  const ConvDesc& d = L.d;
  if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1)) return 0;
  if (d.C0 != 64 || d.C1 != 0 || d.Cout != 64 || d.CoutPad != 64 || d.kchunks != 2) return 0;
  if (L.residual || !L.bias || d.ups) return 0;
  if (d.Wout % TW || d.Hout % TH) return 0;
  if ((size_t)d.B * d.Hin * d.Win * 128 + 4096 >= ((size_t)1 << 31)) return 0;
  const int tiles_x = d.Wout / TW, tiles_y = d.Hout / TH;
  const int total = tiles_x * tiles_y * d.B;
  const int num_cus = device_cu_count();
  if (num_cus <= 0 || total < 2 * num_cus) return 0;         // (at least two tiles per CU: the staging pipeline needs a next tile)
  const int grid = num_cus & ~7;
  const int cpg = L.gn_groups > 0 ? 64 / L.gn_groups : 0;
  const int split_n = cpg > 32 ? 2 : 1;
  const int fuse = L.gn_partials != nullptr && cpg % 8 == 0 && cpg <= 64 && (cpg & (cpg - 1)) == 0 &&
                   tiles_x * tiles_y * 2 * split_n <= kGnMaxSplit;
  if (L.gn_partials && !(fuse && L.gn_acc)) return 0;        // statistics: the fixed-point accumulator form only
  const GnFold& pf = L.pro_fold;
  const bool fold_ok = pf.acc && pf.P && pf.Q && pf.G * pf.cpg == 64 && pf.cpg % 8 == 0;
  if (pf.acc && !fold_ok) return 0;
  if (L.pro_a && !fold_ok) return 0;                         // coefficient-table prologues stay on conv3x3_c64_kernel<1>
  if (fold_ok && !(L.in_f16 && L.w_f16)) return 0;           // the folded prologue: in its h16 form only
  if (L.in_f16 && !fold_ok) return 0;
  const int pro = L.in_f16 ? 3 : 0;
  if (L.out_f16 && pro != 0) return 0;
  static const int pro3_on = [] { const char* e = std::getenv("PRG_CONV_C64W_PRO"); return e ? std::atoi(e) : 1; }();
  if (pro == 3 && !pro3_on) return 0;
  const int variant = L.out_f16 ? 2 : (pro == 3 ? 1 : 0);
  const void* const fns[3] = {reinterpret_cast<const void*>(&conv3x3_c64w_kernel<0, false>), reinterpret_cast<const void*>(&conv3x3_c64w_kernel<3, false>),
                              reinterpret_cast<const void*>(&conv3x3_c64w_kernel<0, true>)};
  static DeviceOnce attr_done[3];
  if (!attr_done[variant].done()) {
    hipError_t e = hipFuncSetAttribute(fns[variant], hipFuncAttributeMaxDynamicSharedMemorySize, (int)C64W_LDS);
    if (e != hipSuccess) return fail(PRG_E_HIP, std::string("hipFuncSetAttribute(c64w conv): ") + hipGetErrorString(e));
    attr_done[variant].mark();
  }
  if (gn_nsplit_out) *gn_nsplit_out = fuse ? tiles_x * tiles_y * 2 * split_n : 0;   // (what conv3x3_c64_kernel reports: the caller's "statistics taken" flag)
  if (coef_done) *coef_done = 0;
  if (acc_done) *acc_done = fuse ? 1 : 0;
  if (L.probe) return 1;
  const int flags = fuse ? 1 : 0;
  if (variant == 2) conv3x3_c64w_kernel<0, true><<<dim3(grid), 256, C64W_LDS, s>>>(L, tiles_x, tiles_y, flags);
  else if (variant == 1) conv3x3_c64w_kernel<3, false><<<dim3(grid), 256, C64W_LDS, s>>>(L, tiles_x, tiles_y, flags);
  else conv3x3_c64w_kernel<0, false><<<dim3(grid), 256, C64W_LDS, s>>>(L, tiles_x, tiles_y, flags);
  PRG_LAUNCH_CHECK();
  return 1;
}

}  // namespace prg
