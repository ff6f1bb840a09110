// conv_split.hip — the convolutions of the `f16x3` precision mode (round 4): float32 storage, every contraction on the
// f16 matrix pipe with BOTH operands split into two halves on the fly.
//
//   a = a_hi + a_lo,  a_hi = f16(a),  a_lo = f16(a - a_hi)      |a - a_hi - a_lo| <= 2^-22 |a|  (f16 subnormals are kept by
//   a * w ~= a_hi * w_hi + a_hi * w_lo + a_lo * w_hi            v_mfma_f32_32x32x16_f16: probed, tools/micro/f16_split_probe.hip)
//
// Three v_mfma_f32_32x32x16_f16 per product tile instead of eight v_mfma_f32_32x32x2_f32: 5.3x fewer matrix-pipe cycles
// than the exact-f32 parity kernels at 22-bit operands; the dropped a_lo * w_lo term is 2^-22 relative.  The fp32
// accumulation chain of the matrix pipe is cut at every 32-channel chunk (9 taps x 32 channels = 288 terms) and the partials
// are added in a second float32 accumulator set: measured on the part against float64 dot products, this sits 4-5x closer to
// exact than one K-long chain (K = 4608: rms 1.3e-5 vs 4.5e-5 on outputs of rms 62) and as close as a float64 sum of 32-term
// partials to within 1.5x — i.e. at the distance from exact arithmetic of the reference's own oneDNN convolution.
//
// Weights are split once at load (pack_conv_weight_split); activations while the halo / the gather tile is written to LDS
// (v_cvt_pk_f16_f32, one subtraction: 2.5 VALU instructions per element), after the optional fused GroupNorm + SiLU
// prologue.  LDS rows are 128 bytes per pixel and 32-channel chunk — 64 B of hi halves, 64 B of lo halves — with the
// 16-byte XOR swizzle of conv.hip, so the tile shapes, the epilogue (conv_epi.h) and the GroupNorm statistics slabs are the
// parity kernels' own.  Reference: Block.proj / res_conv / to_qkv / to_out / Downsample / Upsample, sd:583-796.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "conv_split_common.h"

namespace prg {

// acc += a * w with split operands: the two cross terms first (small), then the leading term
__device__ inline void mma3(const f16x8& ah, const f16x8& al, const f16x8& wh, const f16x8& wl, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, c, 0, 0, 0);
}

// all tiles of a wave: term-major, so that back-to-back MFMAs never wait on the same accumulator
template <int TM>
__device__ inline void mma3_tiles(const f16x8 (&ah)[TM], const f16x8 (&al)[TM], const f16x8 (&wh)[2], const f16x8 (&wl)[2],
                                  f32x16 (&acc)[TM][2]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[j], acc[i][j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], acc[i][j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
}

template <int TM>
__device__ inline void flush_acc(f32x16 (&acc)[TM][2], f32x16 (&tot)[TM][2]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        tot[i][j][e] += acc[i][j][e];
        acc[i][j][e] = 0.0f;
      }
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1: halo tile in LDS (structure of conv3x3_halo_kernel, conv.hip)
// ---------------------------------------------------------------------------------------------
// Tiles of 128 pixels x 64 channels, four waves, one 32-pixel x 64-channel wave tile each; two workgroups per CU.
//   * software pipeline: body(tap) = { wait + barrier | stage one pass of the NEXT chunk's halo | prefetch the weight tile
//     NS taps ahead | issue the twelve fragment loads of the NEXT tap into the other register set | twelve MFMAs of THIS tap };
//     the loads and the staging arithmetic do not feed the MFMAs behind them, so they run in the matrix pipe's shadow;
//   * the nine taps of a chunk are unrolled with the tap a compile-time constant: halo rows are 144 bytes (128 + 16 of
//     padding: conflict-free ds_read_b128 without an XOR swizzle, as in conv_w256.hip), so every fragment address is one
//     per-lane base register + an immediate; the weight ring slot (tap % 3) and the staging pass are constants too —
//     the instruction diet that took the loop from 8 VALU + 6 SALU per MFMA to ~2;
//   * the halo is DOUBLE-BUFFERED in LDS and staged incrementally (pass k of the next chunk loaded at tap k, converted and
//     written at tap k + 1): eight staging registers instead of the whole halo's 32-48;
//   * weight tiles go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a ring
//     of NS slots, with counted vmcnt waits so that the newest prefetch stays in flight across the barrier.
// History of the round (level-0 launch, 77 GFLOP): 64 x 64 wave tiles at 256 registers spilled the weight prefetch inside
// the loop (409 us); three workgroups per CU at 168 registers with just-in-time fragment loads 284 us — and 209 us with the
// MFMAs compiled out: MFMA 105 + LDS reads 53 + L1/TA 47 + split 21 + epilogue 48 us simply added up, identical workgroups
// run in lockstep and overlap nothing; the first pipelined version 345 us at 8 VALU + 6 SALU instructions per MFMA
// (rocprofv3 SQ_INSTS_*: address arithmetic and tap bookkeeping); this structure: DESIGN.md section 4.6.

template <int TH, int TW, int NS>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_split_kernel(
    const ConvLaunch<float> L, const int tiles_x, const int tiles_y, const int tiles_n, const int fuse_stats) {
  constexpr int NW = 4, BN = 64;
  constexpr int CH = 32;                       // channels per chunk: one LDS row = 64 B of hi halves + 64 B of lo halves
  constexpr int HP = TW + 2, HALO = (TH + 2) * HP;
  constexpr int PITCH = 144;                   // halo pixel pitch in bytes (128 + 16: consecutive pixels on distinct 16-byte slots)
  // halo ROW stride: a multiple of 256 bytes (16 slots).  A 16-wide tile puts lanes 16-31 of a fragment load one halo row below
  // lanes 0-15; with the natural stride (18 x 144 B = 2 slots mod 16) two lanes of every ds_read_b128 lane group met on a bank
  // (SQ_LDS_BANK_CONFLICT above SQ_ACTIVE_INST_LDS); with a stride of 0 mod 16 slots the two half rows interleave exactly.
  constexpr int RSTRIDE = (HP * PITCH + 255) / 256 * 256;
  constexpr int NH = (HALO + 63) / 64;         // staging passes per chunk (64 halo pixels each)
  constexpr int HBYTES = (TH + 2) * RSTRIDE;
  constexpr int WPW = BN / 8 / NW;             // global_load_lds wave-instructions (8 rows each) per wave and weight tile
  constexpr int DIST = 8 - NH;                 // taps between a staging pass's load and its conversion / LDS write (last write at tap 7:
                                               // tap 8's body already issues the next chunk's first fragment loads)
  static_assert(TH * TW == 128 && NH <= 4 && (NS == 2 || NS == 3), "tile");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ah = smem;                                 // [2][HALO] rows of PITCH bytes: units 0-3 hi, 4-7 lo
  char* const Bs = smem + 2 * HBYTES;                    // [NS][64][128 B], 16-byte units XOR-swizzled by (row >> 1) & 7
  float* stage = reinterpret_cast<float*>(smem);         // epilogue scratch aliases the main-loop images

  const ConvDesc& d = L.d;
  const int nblk = tiles_x * tiles_y * tiles_n * d.B;
  // Workgroup b runs on XCD b % 8 (observed).  With 2 / 4 / 8 output-channel tiles every XCD is pinned to ONE of them, so its
  // 4 MB L2 holds that tile's weight slice (512 -> 512: 1.2 MB of the 9.4 MB set) instead of streaming the whole set through;
  // otherwise each XCD gets a contiguous run of tiles (xcd_remap).
  int tn, lin;
  if (tiles_n > 1 && 8 % tiles_n == 0 && nblk % 8 == 0) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per = 8 / tiles_n;
    tn = xcd % tiles_n;
    lin = idx * per + xcd / tiles_n;
  } else {
    lin = xcd_remap(blockIdx.x, nblk);
    tn = lin % tiles_n;
    lin /= tiles_n;
  }
  const int tx = lin % tiles_x; lin /= tiles_x;
  const int ty = lin % tiles_y;
  const int b = lin / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nchunks = (d.C0 + d.C1) / CH;
  const int Hl = d.Hout, Wl = d.Wout;
  const int q = tid & 3, prow = tid >> 2;      // this thread stages channels q * 8 .. + 7 of halo pixels prow + 64 k

  int hsrc[NH];                                // source pixel index (or -1: padding / beyond the halo)
#pragma unroll
  for (int k = 0; k < NH; ++k) {
    const int hp = prow + k * 64;
    hsrc[k] = -1;
    if (hp < HALO) {
      const int hy = hp / HP, hx = hp - hy * HP;
      int y = y0 - 1 + hy, x = x0 - 1 + hx;
      if ((unsigned)y < (unsigned)Hl && (unsigned)x < (unsigned)Wl) {
        if (d.ups) { y >>= 1; x >>= 1; }
        hsrc[k] = (b * d.Hin + y) * d.Win + x;
      }
    }
  }
  // fragment addresses: one per-lane base each, everything else immediate
  const int pix = wave * 32 + l31;
  const int a_lane = (pix / TW) * RSTRIDE + (pix % TW) * PITCH + hi * 16;         // + buffer + tap offset + step * 32 (+ 64: lo)
  int b_lane[2][2];                                                                // [step][hi / lo], + ring slot + j * 4096
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int sw = (l31 >> 1) & 7, unit = 2 * st + hi;
    b_lane[st][0] = l31 * 128 + ((unit ^ sw) << 4);
    b_lane[st][1] = l31 * 128 + (((4 + unit) ^ sw) << 4);
  }
  int w_lane[NH];                                                                  // staging write address of pass k (+ 64: lo)
#pragma unroll
  for (int k = 0; k < NH; ++k) {
    const int hp = prow + k * 64;
    w_lane[k] = (hp / HP) * RSTRIDE + (hp % HP) * PITCH + q * 16;
  }

  // one staging pass: load (global -> registers) and, one tap later, prologue + split + write (registers -> LDS)
  float4 hh0[NH], hh1[NH];                     // one register pair per pass: a pass is written DIST taps after its load
  auto halo_load = [&](int chunk, auto K) {   // always issued (padding lanes read the tensor's first bytes and are zeroed when
    constexpr int k = decltype(K)::value;      // written): the counted vmcnt waits below need a fixed number of loads per body
    const int c = chunk * CH + q * 8;
    const bool first = c < d.C0;
    const float* base = first ? L.src0 : L.src1;
    const int Cs = first ? d.C0 : d.C1, cc = first ? c : c - d.C0;
    const float4* p = reinterpret_cast<const float4*>(base + (hsrc[k] >= 0 ? (size_t)hsrc[k] * Cs + cc : (size_t)0));
    hh0[k] = p[0];
    hh1[k] = p[1];
  };
  float pa[8], pb[8];                          // fused prologue coefficients of this thread's 8 channels (chunk being staged)
  auto pro_load = [&](int chunk) {
    if (L.pro_a) {
      const float4* a4 = reinterpret_cast<const float4*>(L.pro_a + (size_t)b * d.C0 + chunk * CH + q * 8);
      const float4* b4 = reinterpret_cast<const float4*>(L.pro_b + (size_t)b * d.C0 + chunk * CH + q * 8);
      const float4 a0 = a4[0], a1 = a4[1], b0 = b4[0], b1 = b4[1];
      pa[0] = a0.x; pa[1] = a0.y; pa[2] = a0.z; pa[3] = a0.w; pa[4] = a1.x; pa[5] = a1.y; pa[6] = a1.z; pa[7] = a1.w;
      pb[0] = b0.x; pb[1] = b0.y; pb[2] = b0.z; pb[3] = b0.w; pb[4] = b1.x; pb[5] = b1.y; pb[6] = b1.z; pb[7] = b1.w;
    }
  };
  auto halo_write = [&](int buf, auto K) {
    constexpr int k = decltype(K)::value;
    if (k * 64 + 63 < HALO || prow + k * 64 < HALO) {
      const float4 h0 = hh0[k], h1 = hh1[k];
      float v[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
      if (L.pro_a) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = silu_fast(fmaf(v[u], pa[u], pb[u]));
      }
      if (hsrc[k] < 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = 0.0f;
      }
      uint4 vh, vl;
      if (L.pro_a) split8<false>(v, vh, vl);
      else split8(v, vh, vl);
      char* p = Ah + buf * HBYTES + w_lane[k];
      *reinterpret_cast<uint4*>(p) = vh;
      *reinterpret_cast<uint4*>(p + 64) = vl;
    }
  };

  // Weight tiles: the LDS destination of a global_load_lds wave-instruction is lane-linear (base + lane * 16 = eight 128-byte
  // rows), so the XOR swizzle the fragment reads apply sits on the per-lane SOURCE address: LDS unit p of row n holds source
  // unit p ^ ((n >> 1) & 7).
  const char* wtile = reinterpret_cast<const char*>(L.w_split + (size_t)tn * BN * 64);
  const size_t wstep = (size_t)d.CoutPad * 128;                                   // bytes between (tap, chunk) tiles
  int wsrc[WPW];
#pragma unroll
  for (int r = 0; r < WPW; ++r) {
    const int n = (wave * WPW + r) * 8 + (lane >> 3);
    wsrc[r] = n * 128 + (((lane & 7) ^ ((n >> 1) & 7)) << 4);
  }
  auto gload_b = [&](int chunk, int tap, int slot) {
    const char* p = wtile + (size_t)(tap * L.split_kchunks + chunk) * wstep;
#pragma unroll
    for (int r = 0; r < WPW; ++r)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + wsrc[r]),
                                       (__attribute__((address_space(3))) void*)(Bs + (slot * BN + (wave * WPW + r) * 8) * 128), 16, 0, 0);
  };

  f32x16 acc[2], tot[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[j][e] = 0.0f; tot[j][e] = 0.0f; }

  const int niter = nchunks * 9;
  // weight tiles 0 .. NS - 1 and chunk 0's halo (all passes in flight at once: one HBM latency per tile, not NH)
#pragma unroll
  for (int t = 0; t < NS; ++t)
    if (t < niter) gload_b(t / 9, t % 9, t);
  pro_load(0);
  {
    float4 g0[NH], g1[NH];
    const bool first = q * 8 < d.C0;
    const float* base = first ? L.src0 : L.src1;
    const int Cs = first ? d.C0 : d.C1, cc = first ? q * 8 : q * 8 - d.C0;
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      if (hsrc[k] >= 0) {
        const float4* p = reinterpret_cast<const float4*>(base + (size_t)hsrc[k] * Cs + cc);
        g0[k] = p[0];
        g1[k] = p[1];
      } else {
        g0[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        g1[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    auto wr = [&](auto K) {
      hh0[decltype(K)::value] = g0[decltype(K)::value];
      hh1[decltype(K)::value] = g1[decltype(K)::value];
      halo_write(0, K);
    };
    wr(IC<0>()); wr(IC<1>()); wr(IC<2>());
    if constexpr (NH > 3) wr(IC<3>());
    if constexpr (NH > 4) wr(IC<4>());
    if constexpr (NH > 5) wr(IC<5>());
    if constexpr (NH > 6) wr(IC<6>());
  }

  f16x8 fa[2][2][2], fw[2][2][4];               // [register set][k16 step][A: hi, lo | W: (hi, lo) x column tile]
  // fragment loads of (chunk buffer cb, tap T) into register set S; ring slot T % NS (9 taps per chunk and NS = 3) or runtime
  auto reads = [&](auto SET, auto TAP, int cb, int slot) {
    constexpr int S = decltype(SET)::value, T = decltype(TAP)::value;
    constexpr int toff = (T / 3) * RSTRIDE + (T % 3) * PITCH;
    const char* A = Ah + cb * HBYTES + a_lane + toff;
    const char* Bb = Bs + slot * (BN * 128);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      fa[S][st][0] = ld_frag(reinterpret_cast<const uint4*>(A + st * 32));
      fa[S][st][1] = ld_frag(reinterpret_cast<const uint4*>(A + st * 32 + 64));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fw[S][st][2 * j] = ld_frag(reinterpret_cast<const uint4*>(Bb + b_lane[st][0] + j * 4096));
        fw[S][st][2 * j + 1] = ld_frag(reinterpret_cast<const uint4*>(Bb + b_lane[st][1] + j * 4096));
      }
    }
  };
  auto mfmas = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      // term-major: back-to-back MFMAs never wait on the same accumulator; the two cross terms first (small), then hi * hi
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][st][1], fw[S][st][2 * j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][st][0], fw[S][st][2 * j + 1], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][st][0], fw[S][st][2 * j], acc[j], 0, 0, 0);
    }
  };

  // one tap of chunk c; P = parity of the chunk (register set of tap T = (T + P) & 1)
  auto body = [&](auto PAR, auto TAP, int c) {
    constexpr int P = decltype(PAR)::value, T = decltype(TAP)::value, S = (T + P) & 1;
    const int it = c * 9 + T;
    const bool more = c + 1 < nchunks;
    // (1) all but the newest weight prefetch (tile it + 2, issued by the previous body AFTER its halo pass) have landed — tile
    //     it + 1 in particular, and that halo pass; this wave's fragment loads of `it` and staging ds_writes are done.
    //     A RAW s_barrier: __syncthreads() makes hipcc drain vmcnt(0) in front of it while an LDS-DMA is in flight, i.e. a
    //     prefetch distance of zero.  The asm's memory clobber keeps LDS accesses on their side of it.  hipcc floats this
    //     tap's MFMAs (register-only) to just behind the barrier, in front of the staging code: measured the best order —
    //     pinning them in front of the wait (accumulators as asm operands) cost 20-35 % (355 / 395 us against 286 us at
    //     level 0): the wave then issues nothing else for 384 cycles and meets the barrier late.
    if (NS == 3 && (T + 2 < 9 || more)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(WPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // tile it + 1 is visible; nobody still reads tile it's ring slot or the previous chunk's halo
    if (more) {
      // (2) the next chunk's halo, one pass per tap: written one tap after its load, into the other halo buffer
      //     (loaded at taps 0 .. NH - 1, written DIST taps later: the loads come from HBM — ~2 us, two to four taps)
      if constexpr (T == 0) pro_load(c + 1);
      if constexpr (T >= DIST && T < DIST + NH) halo_write((c + 1) & 1, IC<T - DIST>());
      if constexpr (T < NH) halo_load(c + 1, IC<T>());
    }
    // (3) weight tile it + NS into the slot tile `it` just vacated
    if constexpr (T + NS < 9) gload_b(c, T + NS, NS == 3 ? T % 3 : it & 1);
    else if (more) gload_b(c + 1, T + NS - 9, NS == 3 ? T % 3 : it & 1);
    // (4) fragments of the next tap into the other register set, (5) the MFMAs of this tap
    if constexpr (T < 8) reads(IC<1 - S>(), IC<T + 1>(), c & 1, NS == 3 ? (T + 1) % 3 : (it + 1) & 1);
    else if (more) reads(IC<1 - S>(), IC<0>(), (c + 1) & 1, NS == 3 ? 0 : (it + 1) & 1);
    mfmas(IC<S>());
    if constexpr (T == 8) {                    // the 288-term partial of this channel chunk
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { tot[j][e] += acc[j][e]; acc[j][e] = 0.0f; }
    }
  };
  auto chunk_body = [&](auto PAR, int c) {
    body(PAR, IC<0>(), c); body(PAR, IC<1>(), c); body(PAR, IC<2>(), c);
    body(PAR, IC<3>(), c); body(PAR, IC<4>(), c); body(PAR, IC<5>(), c);
    body(PAR, IC<6>(), c); body(PAR, IC<7>(), c); body(PAR, IC<8>(), c);
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  reads(IC<0>(), IC<0>(), 0, 0);
  for (int c = 0; c < nchunks; c += 2) {
    chunk_body(IC<0>(), c);
    if (c + 1 < nchunks) chunk_body(IC<1>(), c + 1);
  }

  double gs, gq;
  [[maybe_unused]] auto row_to_m = [&](int r) -> int64_t {
    const int p = wave * 32 + r;
    return ((int64_t)b * d.Hout + y0 + p / TW) * d.Wout + x0 + p % TW;
  };
  if (L.split_scale) {                         // undo the packer's per-channel power-of-two weight scale (exact)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float sc = L.split_scale[tn * BN + j * 32 + l31];
#pragma unroll
      for (int e = 0; e < 16; ++e) tot[j][e] *= sc;
    }
  }
  // DIRECT epilogue: the accumulator layout already has lanes 0-31 = 32 consecutive channels of ONE pixel (register e of half hi
  // is pixel row (e & 3) + 8 (e >> 2) + 4 hi), so a dword store per register writes two full 128-byte lines per wave-instruction —
  // no LDS transpose, no barrier; the GroupNorm partial sums come from lane reductions (a group's channels are adjacent lanes).
  (void)stage; (void)gs; (void)gq;
  {
    const int cpg = L.gn_groups > 0 ? d.Cout / L.gn_groups : 8;
    float sj[2], qj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ch = tn * BN + j * 32 + l31;
      const float bv = L.bias ? L.bias[ch] : 0.0f;
      float s1 = 0.0f, q1 = 0.0f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int p = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
        const size_t m = ((size_t)b * d.Hout + y0 + p / TW) * d.Wout + x0 + p % TW;
        const float v = tot[j][e] + bv;
        L.out[m * d.Cout + ch] = v;
        s1 += v;
        q1 = fmaf(v, v, q1);
      }
      sj[j] = s1;
      qj[j] = q1;
    }
    if (fuse_stats) {
      // a lane holds (sum, sumsq) of ONE channel over its 16 pixels: fold the two pixel halves, then the cpg adjacent channels
      double* red = reinterpret_cast<double*>(smem + 2 * HBYTES + NS * BN * 128);      // [4 waves][2 j][4][2] beyond the images
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        double sd = (double)sj[j], qd = (double)qj[j];
        sd += __shfl_xor(sd, 32, 64);
        qd += __shfl_xor(qd, 32, 64);
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {                 // the 8 channels of a granule are 8 adjacent lanes
          sd += __shfl_xor(sd, o, 64);
          qd += __shfl_xor(qd, o, 64);
        }
        if (hi == 0 && (l31 & 7) == 0) {                  // (8-channel granules: cpg is a multiple of 8 on this path)
          red[((wave * 2 + j) * 4 + (l31 >> 3)) * 2] = sd;
          red[((wave * 2 + j) * 4 + (l31 >> 3)) * 2 + 1] = qd;
        }
      }
      __syncthreads();
      // groups of this tile's 64 channels: 64 / cpg of them; thread g sums the four waves (fixed order) and the granules of a group
      const int ng = 64 / cpg;
      if (tid < ng) {
        const int gran0 = tid * (cpg / 8), ngran = cpg / 8;
        double ss = 0, qq = 0;
        for (int w = 0; w < 4; ++w)
          for (int k = 0; k < ngran; ++k) {
            const int gr = gran0 + k;                       // granule 0..7 of the tile: column tile gr >> 2, lanes 8 (gr & 3) .. + 7
            ss += red[((w * 2 + (gr >> 2)) * 4 + (gr & 3)) * 2];
            qq += red[((w * 2 + (gr >> 2)) * 4 + (gr & 3)) * 2 + 1];
          }
        const int nsplit = tiles_x * tiles_y;
        float* dst = L.gn_partials + ((size_t)b * nsplit + ty * tiles_x + tx) * L.gn_groups * 2;
        const int g = (tn * BN) / cpg + tid;
        dst[g * 2] = (float)ss;
        dst[g * 2 + 1] = (float)qq;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1, Cout % 128 == 0: the WAVE-SPECIALISED form (second half of round 4)
// ---------------------------------------------------------------------------------------------
// The symmetric kernel above is LDS-read bound at one ds_read_b128 per MFMA (32 x 64 wave tiles: larger ones do not fit its 256
// registers beside the staging state) and its two identical workgroups per CU run in lockstep.  Here a 512-thread workgroup owns
// a CU and its waves have DIFFERENT jobs, so the phases really overlap:
//   waves 0-3  CONSUMERS, one per SIMD: 64 pixels x 64 channels each (2 x 2 over the 128-pixel x 128-channel tile): per tap 16
//              fragment loads for 24 MFMAs (0.67 per MFMA), double-buffered per k16 step — step 1's fragments load beside step
//              0's MFMAs, the NEXT tap's step 0 beside step 1's; they never touch global memory, never convert, and their
//              barrier carries no wait at all;
//   waves 4-5  HALO producers: all passes of the next chunk's halo loaded at tap 0 (48 registers: they have no accumulators),
//              prologue + split + ds_write at taps 2-7 into the other halo buffer;
//   waves 6-7  WEIGHT producers: global_load_lds of tile t + 3 into a ring of four, `s_waitcnt vmcnt(8)` = tile t + 2 landed.
//              Their own waves because vmcnt retires IN ORDER: behind the halo's HBM loads a counted wait on the weight
//              prefetch would wait for HBM.
// One raw s_barrier per tap for the eight waves; invariant at barrier(t): weight tiles t and t + 1 are in LDS (so a consumer may
// fetch tile t + 1's first fragments before barrier(t + 1)), tile t - 1's slot is free.  The epilogue is wave-local (each
// consumer transposes its own tile through a private LDS stage; GroupNorm partial sums per (tile, pixel half) slab): no
// workgroup barrier after the one that ends the main loop.
// UP (round 4, second half): Upsample = nearest x2 + 3 x 3 conv as four 2 x 2-tap sub-pixel convolutions of the SOURCE image
// (conv_w256.hip MODE 2 / DESIGN 4.8 for the algebra): a tile is 128 source pixels x one phase (dy, dx) = the two low bits of the
// tile index, NT = 4 taps per chunk from the phase's own split packing (L.w_up_split), the phase's window starts at halo position
// (dy, dx), the epilogue stores to (2 y + dy, 2 x + dx).  With four taps per chunk a halo pass would have ONE tap between its
// load and its conversion, so the halo producers run two chunks ahead (two register sets).
template <int NS, bool UP>
__global__ __launch_bounds__(512, 1) void conv3x3_split_ws_kernel(const ConvLaunch<float> L, const int tiles_x, const int tiles_y,
                                                                  const int tiles_n, const int fuse_stats) {
  constexpr int TH = 8, TW = 16, BN = 128, CH = 32;
  constexpr int NT = UP ? 4 : 9;                         // taps per channel chunk
  constexpr int HP = TW + 2, HALO = (TH + 2) * HP, PITCH = 144;
  constexpr int RSTRIDE = (HP * PITCH + 255) / 256 * 256, HBYTES = (TH + 2) * RSTRIDE;   // (row stride 0 mod 16 slots: see above)
  constexpr int NHP = (HALO * 4 + 127) / 128;            // halo staging passes of the 128 halo-producer threads (6)
  constexpr int WPW = BN / 8 / 2;                        // global_load_lds instructions per weight-producer wave and tile (8)
  static_assert(NS == 4 && NHP <= 6, "ring");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ah = smem;                                 // [2][HALO] rows of PITCH bytes
  char* const Bs = smem + 2 * HBYTES;                    // [NS][128][128 B], units XOR-swizzled by (row >> 1) & 7

  const ConvDesc& d = L.d;
  const int nblk = tiles_x * tiles_y * tiles_n * d.B * (UP ? 4 : 1);
  int tn, lin;
  if (tiles_n > 1 && 8 % tiles_n == 0 && nblk % 8 == 0) {  // every XCD pinned to one output-channel tile (its weight slice stays in L2)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per = 8 / tiles_n;
    tn = xcd % tiles_n;
    lin = idx * per + xcd / tiles_n;
  } else {
    lin = xcd_remap(blockIdx.x, nblk);
    tn = lin % tiles_n;
    lin /= tiles_n;
  }
  const int ph = UP ? lin & 3 : 0;                       // sub-pixel phase 2 dy + dx
  if (UP) lin >>= 2;
  const int tx = lin % tiles_x; lin /= tiles_x;
  const int ty = lin % tiles_y;
  const int b = lin / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nchunks = (d.C0 + d.C1) / CH, niter = nchunks * NT;
  const size_t wstep = (size_t)d.CoutPad * 128;
  const char* wtile = UP ? reinterpret_cast<const char*>(L.w_up_split + (size_t)ph * 4 * L.split_kchunks * d.CoutPad * 64 + (size_t)tn * BN * 64)
                         : reinterpret_cast<const char*>(L.w_split + (size_t)tn * BN * 64);
  const int Hs = UP ? d.Hin : d.Hout, Ws = UP ? d.Win : d.Wout;   // the image the tiles cover (UP: the source)

  if (wave >= 6) {
    // ------------------------------------------------ weight producers ------------------------------------------------
    const int pw = wave - 6;
    int wsrc[WPW];
#pragma unroll
    for (int r = 0; r < WPW; ++r) {
      const int n = (pw * WPW + r) * 8 + (lane >> 3);
      wsrc[r] = n * 128 + (((lane & 7) ^ ((n >> 1) & 7)) << 4);
    }
    auto gload_b = [&](int it) {
      const int c = it / NT, tap = it - NT * c;
      const char* p = wtile + (size_t)(tap * L.split_kchunks + c) * wstep;
      char* dst = Bs + ((it & (NS - 1)) * BN + pw * WPW * 8) * 128;
      if constexpr (!ablate::ws_no_wdma) {
#pragma unroll
        for (int r = 0; r < WPW; ++r)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + wsrc[r]),
                                           (__attribute__((address_space(3))) void*)(dst + r * 1024), 16, 0, 0);
      }
    };
    gload_b(0);
    if (niter > 1) gload_b(1);
    if (niter > 2) gload_b(2);
    // tiles 0 and 1 landed (tile 2 may fly on)
    if (niter > 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int it = 0; it < niter; ++it) {
      if (it + 3 < niter) {
        gload_b(it + 3);                                   // into tile it - 1's slot, free since barrier(it)
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WPW) : "memory");   // tile it + 2 landed -> barrier(it + 1)
      } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
    return;
  }
  if (wave >= 4) {
    // ------------------------------------------------- halo producers -------------------------------------------------
    const int pt = tid - 256, q = pt & 3, prow = pt >> 2;  // channels q * 8 .. + 7 of halo pixels prow + 32 k
    int hsrc[NHP];
#pragma unroll
    for (int k = 0; k < NHP; ++k) {
      const int hp = prow + k * 32;
      hsrc[k] = -1;
      if (hp < HALO) {
        const int hy = hp / HP, hx = hp - hy * HP;
        int y = y0 - 1 + hy, x = x0 - 1 + hx;
        if ((unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws) {
          if (!UP && d.ups) { y >>= 1; x >>= 1; }
          hsrc[k] = (b * d.Hin + y) * d.Win + x;
        }
      }
    }
    float4 gA0[NHP], gA1[NHP], gB0[NHP], gB1[NHP];        // (UP: two chunks in flight; otherwise only set A)
    auto load_chunk_set = [&](int chunk, float4 (&g0)[NHP], float4 (&g1)[NHP]) {
      const int c = chunk * CH + q * 8;
      const bool first = c < d.C0;
      const float* base = first ? L.src0 : L.src1;
      const int Cs = first ? d.C0 : d.C1, cc = first ? c : c - d.C0;
      if constexpr (!ablate::ws_no_halo) {
#pragma unroll
        for (int k = 0; k < NHP; ++k) {
          const float4* p = reinterpret_cast<const float4*>(base + (hsrc[k] >= 0 ? (size_t)hsrc[k] * Cs + cc : (size_t)0));
          g0[k] = p[0];
          g1[k] = p[1];
        }
      }
    };
    float4 (&g0)[NHP] = gA0;
    float4 (&g1)[NHP] = gA1;
    auto load_chunk = [&](int chunk) {
      const int c = chunk * CH + q * 8;
      const bool first = c < d.C0;
      const float* base = first ? L.src0 : L.src1;
      const int Cs = first ? d.C0 : d.C1, cc = first ? c : c - d.C0;
      if constexpr (!ablate::ws_no_halo) {
#pragma unroll
        for (int k = 0; k < NHP; ++k) {
          const float4* p = reinterpret_cast<const float4*>(base + (hsrc[k] >= 0 ? (size_t)hsrc[k] * Cs + cc : (size_t)0));
          g0[k] = p[0];
          g1[k] = p[1];
        }
      }
    };
    float pa[8], pb[8];
    auto pro_load = [&](int chunk) {
      if (L.pro_a) {
        const float4* a4 = reinterpret_cast<const float4*>(L.pro_a + (size_t)b * d.C0 + chunk * CH + q * 8);
        const float4* b4 = reinterpret_cast<const float4*>(L.pro_b + (size_t)b * d.C0 + chunk * CH + q * 8);
        const float4 a0 = a4[0], a1 = a4[1], b0 = b4[0], b1 = b4[1];
        pa[0] = a0.x; pa[1] = a0.y; pa[2] = a0.z; pa[3] = a0.w; pa[4] = a1.x; pa[5] = a1.y; pa[6] = a1.z; pa[7] = a1.w;
        pb[0] = b0.x; pb[1] = b0.y; pb[2] = b0.z; pb[3] = b0.w; pb[4] = b1.x; pb[5] = b1.y; pb[6] = b1.z; pb[7] = b1.w;
      }
    };
    auto write_pass_set = [&](int buf, auto K, const float4 (&h0)[NHP], const float4 (&h1)[NHP]) {
      constexpr int k = decltype(K)::value;
      const int hp = prow + k * 32;
      if (!ablate::ws_no_halo && hp < HALO) {
        float v[8] = {h0[k].x, h0[k].y, h0[k].z, h0[k].w, h1[k].x, h1[k].y, h1[k].z, h1[k].w};
        if (L.pro_a) {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = silu_fast(fmaf(v[u], pa[u], pb[u]));
        }
        if (hsrc[k] < 0) {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = 0.0f;
        }
        uint4 vh, vl;
        if (L.pro_a) split8<false>(v, vh, vl);
        else split8(v, vh, vl);
        char* p = Ah + buf * HBYTES + (hp / HP) * RSTRIDE + (hp % HP) * PITCH + q * 16;
        *reinterpret_cast<uint4*>(p) = vh;
        *reinterpret_cast<uint4*>(p + 64) = vl;
      }
    };
    auto write_pass = [&](int buf, auto K) { write_pass_set(buf, K, gA0, gA1); };
    if constexpr (UP) {
      // Upsample convs have no fused prologue (pro_a is null: checked by the launcher).  Chunk c + 2 is loaded at tap 0 of chunk c
      // into the register set of its parity, chunk c + 1 (loaded four taps earlier) is converted and written at taps 0-2, two
      // passes each; tap 3: nothing (the consumers fetch the next chunk's first fragments during it).
      load_chunk_set(0, gA0, gA1);
      if (nchunks > 1) load_chunk_set(1, gB0, gB1);
      write_pass_set(0, IC<0>(), gA0, gA1); write_pass_set(0, IC<1>(), gA0, gA1); write_pass_set(0, IC<2>(), gA0, gA1);
      write_pass_set(0, IC<3>(), gA0, gA1); write_pass_set(0, IC<4>(), gA0, gA1); write_pass_set(0, IC<5>(), gA0, gA1);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      auto chunk_taps = [&](int c, float4 (&n0)[NHP], float4 (&n1)[NHP], float4 (&f0)[NHP], float4 (&f1)[NHP]) {
        // n = the set holding chunk c + 1 (to be written now), f = the set chunk c held (free: reloaded with chunk c + 2)
        const bool more = c + 1 < nchunks;
        const int nb = (c + 1) & 1;
        if (c + 2 < nchunks) load_chunk_set(c + 2, f0, f1);
        if (more) { write_pass_set(nb, IC<0>(), n0, n1); write_pass_set(nb, IC<1>(), n0, n1); }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");         // -> barrier(4 c + 1)
        if (more) { write_pass_set(nb, IC<2>(), n0, n1); write_pass_set(nb, IC<3>(), n0, n1); }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");         // -> barrier(4 c + 2)
        if (more) { write_pass_set(nb, IC<4>(), n0, n1); write_pass_set(nb, IC<5>(), n0, n1); }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");         // -> barrier(4 c + 3)
        asm volatile("s_barrier" ::: "memory");                                   // -> barrier(4 c + 4)
      };
      for (int c = 0; c < nchunks; c += 2) {
        chunk_taps(c, gB0, gB1, gA0, gA1);
        if (c + 1 < nchunks) chunk_taps(c + 1, gA0, gA1, gB0, gB1);
      }
      return;
    }
    pro_load(0);
    load_chunk(0);
    write_pass(0, IC<0>()); write_pass(0, IC<1>()); write_pass(0, IC<2>());
    write_pass(0, IC<3>()); write_pass(0, IC<4>()); write_pass(0, IC<5>());
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int c = 0; c < nchunks; ++c) {
      const bool more = c + 1 < nchunks;
      const int nb = (c + 1) & 1;
      // tap 0: the next chunk's coefficients and ALL its halo loads; taps 2 .. 7: one pass each converted and written (the
      // other halo buffer was last read during the previous chunk's tap 8, i.e. before barrier(9 c)); tap 8: nothing —
      // the consumers fetch the next chunk's first fragments during it
      if (more) { pro_load(c + 1); load_chunk(c + 1); }
      asm volatile("s_barrier" ::: "memory");                                   // -> barrier(9 c + 1)
      asm volatile("s_barrier" ::: "memory");                                   // -> barrier(9 c + 2)
      if (more) write_pass(nb, IC<0>());
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (more) write_pass(nb, IC<1>());
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (more) write_pass(nb, IC<2>());
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (more) write_pass(nb, IC<3>());
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (more) write_pass(nb, IC<4>());
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (more) write_pass(nb, IC<5>());
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");           // -> barrier(9 c + 8)
      asm volatile("s_barrier" ::: "memory");                                   // -> barrier(9 c + 9)
    }
    return;
  }
  // ---------------------------------------------------- consumers -----------------------------------------------------
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  int a_lane[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pix = wm * 64 + i * 32 + l31;
    a_lane[i] = (pix / TW) * RSTRIDE + (pix % TW) * PITCH + hi * 16;
  }
  int b_lane[2][2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int sw = (l31 >> 1) & 7, unit = 2 * st + hi;
    b_lane[st][0] = wn * 8192 + l31 * 128 + ((unit ^ sw) << 4);
    b_lane[st][1] = wn * 8192 + l31 * 128 + (((4 + unit) ^ sw) << 4);
  }
  f32x16 acc[2][2], tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.0f; tot[i][j][e] = 0.0f; }
  f16x8 fa[2][4], fw[2][4];                              // [k16 step][A: (hi, lo) x row tile | W: (hi, lo) x column tile]
  const int pho = UP ? (ph >> 1) * RSTRIDE + (ph & 1) * PITCH : 0;   // UP: the phase's 2 x 2 window starts at halo position (dy, dx)
  auto reads = [&](auto ST, auto TAP, int cb, int slot) {
    constexpr int st = decltype(ST)::value, T = decltype(TAP)::value;
    constexpr int toff = UP ? (T / 2) * RSTRIDE + (T % 2) * PITCH : (T / 3) * RSTRIDE + (T % 3) * PITCH;
    const char* A = Ah + cb * HBYTES + toff + st * 32 + pho;
    const char* Bb = Bs + slot * (BN * 128);
    if constexpr (ablate::ws_no_reads) {
      (void)A; (void)Bb;
#pragma unroll
      for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(fa[st][i])); asm volatile("" : "+v"(fw[st][i])); }   // (opaque values, no LDS read)
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa[st][2 * i] = ld_frag(reinterpret_cast<const uint4*>(A + a_lane[i]));
      fa[st][2 * i + 1] = ld_frag(reinterpret_cast<const uint4*>(A + a_lane[i] + 64));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fw[st][2 * j] = ld_frag(reinterpret_cast<const uint4*>(Bb + b_lane[st][0] + j * 4096));
      fw[st][2 * j + 1] = ld_frag(reinterpret_cast<const uint4*>(Bb + b_lane[st][1] + j * 4096));
    }
  };
  auto mfmas = [&](auto ST) {
    constexpr int st = decltype(ST)::value;
    if constexpr (ablate::ws_no_mfma) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(fa[st][i])); asm volatile("" ::"v"(fw[st][i])); }   // (keep the fragment loads alive)
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st][2 * i + 1], fw[st][2 * j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st][2 * i], fw[st][2 * j + 1], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st][2 * i], fw[st][2 * j], acc[i][j], 0, 0, 0);
  };
  auto body = [&](auto TAP, int c) {
    constexpr int T = decltype(TAP)::value;
    const int it = c * NT + T;
    const bool more = c + 1 < nchunks;
    // barrier #it (tap 0 runs straight after the prologue's barrier #0): weight tiles it, it + 1 are in LDS, tile it - 1's slot
    // is free for the producers, this chunk's halo is complete.  No wait: a consumer's outstanding fragment loads read only
    // images that stay valid for another tap.
    if (T > 0 || c > 0) asm volatile("s_barrier" ::: "memory");
    if constexpr (ablate::kInterleave) {
      // one fragment read in each of the first eight MFMA shadows instead of eight reads in a row behind the twelfth MFMA: the four
      // consumers are barrier-aligned, so a burst is 32 ds_read_b128 at once on the CU's LDS port (4 cycles each) against one
      // 32-cycle MFMA of cover
      __builtin_amdgcn_sched_barrier(0);
      reads(IC<1>(), TAP, c & 1, it & (NS - 1));
      mfmas(IC<0>());
      interleave_8_reads_12_mfmas();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (T < NT - 1) reads(IC<0>(), IC<(T + 1) % NT>(), c & 1, (it + 1) & (NS - 1));
      else if (more) reads(IC<0>(), IC<0>(), (c + 1) & 1, (it + 1) & (NS - 1));
      mfmas(IC<1>());
      interleave_8_reads_12_mfmas();
      __builtin_amdgcn_sched_barrier(0);
    } else {
      reads(IC<1>(), TAP, c & 1, it & (NS - 1));
      __builtin_amdgcn_sched_barrier(0);
      mfmas(IC<0>());
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (T < NT - 1) reads(IC<0>(), IC<(T + 1) % NT>(), c & 1, (it + 1) & (NS - 1));
      else if (more) reads(IC<0>(), IC<0>(), (c + 1) & 1, (it + 1) & (NS - 1));
      __builtin_amdgcn_sched_barrier(0);
      mfmas(IC<1>());
    }
    if constexpr (T == NT - 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) { tot[i][j][e] += acc[i][j][e]; acc[i][j][e] = 0.0f; }
    }
  };
  if constexpr (ablate::kConsumerPrio > 0) __builtin_amdgcn_s_setprio(ablate::kConsumerPrio);
  asm volatile("s_barrier" ::: "memory");                // the producers' prologue: chunk 0's halo, weight tiles 0 and 1
  reads(IC<0>(), IC<0>(), 0, 0);
  for (int c = 0; c < nchunks; ++c) {
    body(IC<0>(), c); body(IC<1>(), c); body(IC<2>(), c); body(IC<3>(), c);
    if constexpr (!UP) {
      body(IC<4>(), c); body(IC<5>(), c); body(IC<6>(), c); body(IC<7>(), c); body(IC<8>(), c);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // barrier(niter): every consumer is done with the LDS images
  // DIRECT epilogue (as in the symmetric kernel): register e of lane half hi is pixel row (e & 3) + 8 (e >> 2) + 4 hi, lanes 0-31
  // are 32 consecutive channels: a dword store per register writes two full 128-byte lines per wave-instruction; GroupNorm partial
  // sums by lane reductions (a group's channels are adjacent lanes), one slab per (tile, pixel half): no LDS, no barrier.
  {
    const int cpg = L.gn_groups > 0 ? d.Cout / L.gn_groups : 64;          // 16, 32 or 64 when the statistics are fused
    double sd[2], qd[2];
    // Addresses (round 5): pixel p = 64 wm + 32 i + (e & 3) + 8 (e >> 2) + 4 hi of the 16-wide tile is tile row 4 wm + 2 i + (e >> 3),
    // column (e & 3) + 8 ((e >> 2) & 1) + 4 hi: a WAVE-UNIFORM 64-bit tile base (scalar registers) + a 32-bit per-lane offset + uniform
    // per-store terms (row step, pixel step) — one vector add per store.  The generic form (m * d.Cout + ch per element in 64 bits)
    // cost ~350 of the epilogue's ~620 VALU instructions per wave, 140 of them quarter-rate integer multiplies.
    const size_t ps = (size_t)d.Cout * (UP ? 2 : 1);                       // floats between the lane's consecutive tile columns
    const size_t rs = (size_t)d.Wout * d.Cout * (UP ? 2 : 1);              // ... and tile rows
    float* const lbase = L.out + (UP ? (((size_t)b * d.Hout + 2 * (y0 + wm * 4) + (ph >> 1)) * d.Wout + 2 * x0 + (ph & 1)) * d.Cout
                                     : (((size_t)b * d.Hout + y0 + wm * 4) * d.Wout + x0) * d.Cout) + tn * BN + wn * 64 + (size_t)(4 * hi) * ps + l31;
    float bv[2], sc[2], s1[2] = {0.0f, 0.0f}, q1[2] = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ch = tn * BN + wn * 64 + j * 32 + l31;
      bv[j] = L.bias ? L.bias[ch] : 0.0f;
      const float* scp = UP ? L.split_scale_up : L.split_scale;            // the packer's per-channel power of two, undone (exact)
      sc[j] = scp ? scp[(UP ? ph * d.CoutPad : 0) + ch] : 1.0f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float* const cp = lbase + (size_t)(2 * i + (e >> 3)) * rs + (size_t)((e & 3) + 8 * ((e >> 2) & 1)) * ps;   // (uniform offsets)
#pragma unroll
        for (int j = 0; j < 2; ++j) {                                      // (per j the values are summed in the same (i, e) order as before)
          const float v = tot[i][j][e] * sc[j] + bv[j];
          cp[j * 32] = v;
          s1[j] += v;
          q1[j] = fmaf(v, v, q1[j]);
        }
      }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      sd[j] = (double)s1[j];
      qd[j] = (double)q1[j];
    }
    if (fuse_stats) {
      const int width = cpg < 32 ? cpg : 32;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        sd[j] += __shfl_xor(sd[j], 32, 64);
        qd[j] += __shfl_xor(qd[j], 32, 64);
        for (int o = 1; o < width; o <<= 1) {
          sd[j] += __shfl_xor(sd[j], o, 64);
          qd[j] += __shfl_xor(qd[j], o, 64);
        }
      }
      const int nsplit = tiles_x * tiles_y * 2;
      float* slab = L.gn_partials + ((size_t)b * nsplit + (ty * tiles_x + tx) * 2 + wm) * L.gn_groups * 2;
      if (cpg == 64) {
        if (lane == 0) {
          const int g = (tn * BN + wn * 64) / 64;
          slab[g * 2] = (float)(sd[0] + sd[1]);
          slab[g * 2 + 1] = (float)(qd[0] + qd[1]);
        }
      } else if (hi == 0 && (l31 & (width - 1)) == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int g = (tn * BN + wn * 64 + j * 32 + l31) / cpg;
          slab[g * 2] = (float)sd[j];
          slab[g * 2 + 1] = (float)qd[j];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1, Cout = 64: the PERSISTENT wave-specialised form (round 5) — levels 0 / 1 and the narrow Upsample
// ---------------------------------------------------------------------------------------------
// The 64-channel convolutions (64 -> 64, the two-source 128 -> 64 of the up path, Upsample 128 -> 64: sixteen launches per
// evaluation, 29 % of the f16x3 mode's time) ran on the symmetric kernel at 0.33-0.41 MFMA-busy: with K = 576 a tile is 18 taps
// and a one-tile workgroup exposes its first halo's HBM latency and its epilogue, which only two co-resident (lock-step,
// LDS-read-bound) workgroups hid.  Here ONE 512-thread workgroup per CU walks a strided list of 16 x 16-pixel x 64-channel tiles
// with conv3x3_split_ws_kernel's division of labour, and the pipelines never drain between tiles:
//   waves 0-3  CONSUMERS, 64 pixels (four tile rows) x 64 channels each: per tap 16 fragment loads for 24 MFMAs, double-buffered
//              per k16 step; after a tile's last tap they run the direct epilogue (scale, bias, GroupNorm partial sums per
//              (tile, wave) slab, dword stores — no LDS, no barrier) while the producers are already a chunk / three taps ahead;
//   waves 4-5  HALO producers: chunk g + 1's 18 x 18 halo — of the NEXT tile when chunk g is a tile's last — loaded at tap 0 of
//              chunk g (eleven passes, 88 staging registers), prologue + split + ds_write two passes per tap at taps 2-7;
//   waves 6-7  WEIGHT producers: the (tap, chunk) tile sequence simply repeats per pixel tile (K = 576: 18 tiles of 8 KB, L2
//              resident), LDS-DMA three taps ahead into a ring of four, counted vmcnt.
// One raw s_barrier per tap for the eight waves, numbered through the whole tile list (invariants as in the kernel above);
// LDS = 2 x 50,688 (halos, 144-byte pixels, rows of 0 mod 16 slots) + 4 x 8,192 (weights) = 134,144 bytes.
// The arithmetic is the symmetric kernel's, term for term (same MFMA sequence per accumulator, 288-term partials, same epilogue
// expression): only the GroupNorm slab partition differs.
// (Round 5 also measured a MERGED-producer form — all four producer waves staging a quarter of the halo and a quarter of each weight
// tile — at 2-3 % slower, profiles/r05_p64_ablations.txt; removed in round 6.)
template <int NS>
__global__ __launch_bounds__(512, 1) void conv3x3_split_p64_kernel(const ConvLaunch<float> L, const int tiles_x, const int tiles_y,
                                                                   const int ntiles, const int fuse_stats) {
  constexpr int TH = 16, TW = 16, BN = 64, CH = 32, NT = 9;
  constexpr int HP = TW + 2, HALO = (TH + 2) * HP, PITCH = 144;
  constexpr int RSTRIDE = (HP * PITCH + 255) / 256 * 256, HBYTES = (TH + 2) * RSTRIDE;
  constexpr int NHP = (HALO * 4 + 127) / 128;            // halo staging passes of the 128 halo-producer threads (11)
  constexpr int WPW = BN / 8 / 2;                        // global_load_lds instructions per weight-producer wave and tile (4)
  static_assert(NS == 4 && NHP == 11, "ring / passes");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ah = smem;                                 // [2][18 rows of RSTRIDE bytes]
  char* const Bs = smem + 2 * HBYTES;                    // [NS][64][128 B], units XOR-swizzled by (row >> 1) & 7

  const ConvDesc& d = L.d;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nchunks = (d.C0 + d.C1) / CH;
  // this workgroup's tiles: t_first + k * t_stride, k < t_count.  With a grid that is a multiple of 8, XCD x (workgroups b with
  // b % 8 = x: observed) owns a contiguous eighth of the tile list and its workgroups walk it interleaved, so tiles that share halo
  // rows run at the same time on one L2.
  int t_first, t_stride, t_count;
  {
    const int G = (int)gridDim.x, bid = (int)blockIdx.x;
    if ((G & 7) == 0) {
      const int xcd = bid & 7, idx = bid >> 3, per = G >> 3;
      const int q = ntiles >> 3, r = ntiles & 7;
      const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
      const int cnt = q + (xcd < r ? 1 : 0);
      t_first = start + idx; t_stride = per; t_count = idx < cnt ? (cnt - idx + per - 1) / per : 0;
    } else {
      t_first = bid; t_stride = G; t_count = bid < ntiles ? (ntiles - bid + G - 1) / G : 0;
    }
  }
  const int nchunks_total = t_count * nchunks, niter = nchunks_total * NT;
  if (niter == 0) return;                                // (uniform over the workgroup: no barrier is ever reached)
  const size_t wstep = (size_t)d.CoutPad * 128;
  const int Hs = d.Hout, Ws = d.Wout;
  const int tiles_img = tiles_x * tiles_y;

  if (wave >= 6) {
    // ------------------------------------------------ weight producers ------------------------------------------------
    const int pw = wave - 6;
    int wsrc[WPW];
#pragma unroll
    for (int r = 0; r < WPW; ++r) {
      const int n = (pw * WPW + r) * 8 + (lane >> 3);
      wsrc[r] = n * 128 + (((lane & 7) ^ ((n >> 1) & 7)) << 4);
    }
    const char* wtile = reinterpret_cast<const char*>(L.w_split);
    int c_n = 0, tap_n = 0;                              // (chunk, tap) of the next tile to fetch: the sequence repeats per pixel tile
    auto gload_next = [&](int it) {
      const char* p = wtile + (size_t)(tap_n * L.split_kchunks + c_n) * wstep;
      char* dst = Bs + ((it & (NS - 1)) * BN + pw * WPW * 8) * 128;
      if constexpr (!ablate::p64_no_wdma) {
#pragma unroll
        for (int r = 0; r < WPW; ++r)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + wsrc[r]),
                                           (__attribute__((address_space(3))) void*)(dst + r * 1024), 16, 0, 0);
      }
      if (++tap_n == NT) { tap_n = 0; if (++c_n == nchunks) c_n = 0; }
    };
    gload_next(0);
    gload_next(1);
    gload_next(2);                                       // (niter >= 18)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WPW) : "memory");     // tiles 0 and 1 landed -> barrier(0)
    for (int it = 0; it < niter; ++it) {
      if (it + 3 < niter) {
        gload_next(it + 3);                              // into tile it - 1's slot, free since barrier(it)
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WPW) : "memory");   // tile it + 2 landed -> barrier(it + 1)
      } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
    return;
  }
  if (wave >= 4) {
    // ------------------------------------------------- halo producers -------------------------------------------------
    const int pt = tid - 256, q = pt & 3, prow = pt >> 2;  // channels q * 8 .. + 7 of halo pixels prow + 32 k
    int hyx[NHP], wl[NHP];                               // tile-independent: halo coordinates and LDS address of pass k
#pragma unroll
    for (int k = 0; k < NHP; ++k) {
      const int hp = prow + k * 32;
      const int hy = hp / HP, hx = hp - hy * HP;
      hyx[k] = hp < HALO ? ((hy << 8) | hx) : -1;
      wl[k] = hy * RSTRIDE + hx * PITCH + q * 16;
    }
    int hsrc[NHP];                                       // source pixel index (or -1: padding / beyond the halo) for the tile being staged
    int img = 0;
    auto set_tile = [&](int kt) {
      int lin = t_first + kt * t_stride;
      const int tx = lin % tiles_x; lin /= tiles_x;
      const int ty = lin % tiles_y;
      img = lin / tiles_y;
      const int y0 = ty * TH, x0 = tx * TW;
#pragma unroll
      for (int k = 0; k < NHP; ++k) {
        hsrc[k] = -1;
        if (hyx[k] >= 0) {
          int y = y0 - 1 + (hyx[k] >> 8), x = x0 - 1 + (hyx[k] & 255);
          if ((unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws) {
            if (d.ups) { y >>= 1; x >>= 1; }
            hsrc[k] = (img * d.Hin + y) * d.Win + x;
          }
        }
      }
    };
    float4 g0[NHP], g1[NHP];
    auto load_chunk = [&](int chunk) {
      const int c = chunk * CH + q * 8;
      const bool first = c < d.C0;
      const float* base = first ? L.src0 : L.src1;
      const int Cs = first ? d.C0 : d.C1, cc = first ? c : c - d.C0;
      if constexpr (!ablate::p64_no_halo) {
#pragma unroll
        for (int k = 0; k < NHP; ++k) {
          const float4* p = reinterpret_cast<const float4*>(base + (hsrc[k] >= 0 ? (size_t)hsrc[k] * Cs + cc : (size_t)0));
          g0[k] = p[0];
          g1[k] = p[1];
        }
      }
    };
    float pa[8], pb[8];
    auto pro_load = [&](int chunk) {
      if (L.pro_a) {
        const float4* a4 = reinterpret_cast<const float4*>(L.pro_a + (size_t)img * d.C0 + chunk * CH + q * 8);
        const float4* b4 = reinterpret_cast<const float4*>(L.pro_b + (size_t)img * d.C0 + chunk * CH + q * 8);
        const float4 a0 = a4[0], a1 = a4[1], b0 = b4[0], b1 = b4[1];
        pa[0] = a0.x; pa[1] = a0.y; pa[2] = a0.z; pa[3] = a0.w; pa[4] = a1.x; pa[5] = a1.y; pa[6] = a1.z; pa[7] = a1.w;
        pb[0] = b0.x; pb[1] = b0.y; pb[2] = b0.z; pb[3] = b0.w; pb[4] = b1.x; pb[5] = b1.y; pb[6] = b1.z; pb[7] = b1.w;
      }
    };
    auto write_pass = [&](int buf, auto K) {
      constexpr int k = decltype(K)::value;
      if (!ablate::p64_no_halo && hyx[k] >= 0) {
        float v[8] = {g0[k].x, g0[k].y, g0[k].z, g0[k].w, g1[k].x, g1[k].y, g1[k].z, g1[k].w};
        if (L.pro_a) {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = silu_fast(fmaf(v[u], pa[u], pb[u]));
        }
        if (hsrc[k] < 0) {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = 0.0f;
        }
        uint4 vh, vl;
        if (L.pro_a) split8<false>(v, vh, vl);
        else split8(v, vh, vl);
        char* p = Ah + buf * HBYTES + wl[k];
        *reinterpret_cast<uint4*>(p) = vh;
        *reinterpret_cast<uint4*>(p + 64) = vl;
      }
    };
    set_tile(0);
    pro_load(0);
    load_chunk(0);
    write_pass(0, IC<0>()); write_pass(0, IC<1>()); write_pass(0, IC<2>()); write_pass(0, IC<3>());
    write_pass(0, IC<4>()); write_pass(0, IC<5>()); write_pass(0, IC<6>()); write_pass(0, IC<7>());
    write_pass(0, IC<8>()); write_pass(0, IC<9>()); write_pass(0, IC<10>());
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");               // -> barrier(0)
    int kt = 0, c = 0;
    for (int g = 0; g < nchunks_total; ++g) {
      const bool more = g + 1 < nchunks_total;
      int c1 = c + 1, kt1 = kt;
      if (c1 == nchunks) { c1 = 0; kt1 = kt + 1; }
      const int nb = (g + 1) & 1;
      // tap 0: the next chunk's (the next TILE's first chunk's) coordinates, coefficients and ALL its halo loads; taps 2 .. 7: two
      // passes each converted and written (the other halo buffer was last read during the previous chunk's tap 8, i.e. before
      // barrier(9 g)); tap 8: nothing — the consumers fetch the next chunk's first fragments during it
      if (more) {
        if (c1 == 0) set_tile(kt1);
        pro_load(c1);
        load_chunk(c1);
      }
      asm volatile("s_barrier" ::: "memory");                                     // -> barrier(9 g + 1)
      asm volatile("s_barrier" ::: "memory");                                     // -> barrier(9 g + 2)
      if (more) { write_pass(nb, IC<0>()); write_pass(nb, IC<1>()); }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (more) { write_pass(nb, IC<2>()); write_pass(nb, IC<3>()); }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (more) { write_pass(nb, IC<4>()); write_pass(nb, IC<5>()); }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (more) { write_pass(nb, IC<6>()); write_pass(nb, IC<7>()); }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (more) { write_pass(nb, IC<8>()); write_pass(nb, IC<9>()); }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (more) write_pass(nb, IC<10>());
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");             // -> barrier(9 g + 8)
      asm volatile("s_barrier" ::: "memory");                                     // -> barrier(9 g + 9)
      c = c1;
      kt = kt1;
    }
    return;
  }
  // ---------------------------------------------------- consumers -----------------------------------------------------
  const int l31 = lane & 31, hi = lane >> 5;
  int a_lane[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pix = wave * 64 + i * 32 + l31;
    a_lane[i] = (pix / TW) * RSTRIDE + (pix % TW) * PITCH + hi * 16;
  }
  int b_lane[2][2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int sw = (l31 >> 1) & 7, unit = 2 * st + hi;
    b_lane[st][0] = l31 * 128 + ((unit ^ sw) << 4);
    b_lane[st][1] = l31 * 128 + (((4 + unit) ^ sw) << 4);
  }
  f32x16 acc[2][2], tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.0f; tot[i][j][e] = 0.0f; }
  f16x8 fa[2][4], fw[2][4];                              // [k16 step][A: (hi, lo) x row tile | W: (hi, lo) x column tile]
  auto reads = [&](auto ST, auto TAP, int cb, int slot) {
    constexpr int st = decltype(ST)::value, T = decltype(TAP)::value;
    if constexpr (ablate::p64_no_reads) {
      (void)cb; (void)slot; (void)T;
#pragma unroll
      for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(fa[st][i])); asm volatile("" : "+v"(fw[st][i])); }   // (opaque values, no LDS read)
      return;
    }
    constexpr int toff = (T / 3) * RSTRIDE + (T % 3) * PITCH;
    const char* A = Ah + cb * HBYTES + toff + st * 32;
    const char* Bb = Bs + slot * (BN * 128);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa[st][2 * i] = ld_frag(reinterpret_cast<const uint4*>(A + a_lane[i]));
      fa[st][2 * i + 1] = ld_frag(reinterpret_cast<const uint4*>(A + a_lane[i] + 64));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fw[st][2 * j] = ld_frag(reinterpret_cast<const uint4*>(Bb + b_lane[st][0] + j * 4096));
      fw[st][2 * j + 1] = ld_frag(reinterpret_cast<const uint4*>(Bb + b_lane[st][1] + j * 4096));
    }
  };
  auto mfmas = [&](auto ST) {
    constexpr int st = decltype(ST)::value;
    if constexpr (ablate::p64_no_mfma) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(fa[st][i])); asm volatile("" ::"v"(fw[st][i])); }   // (keep the fragment loads alive)
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st][2 * i + 1], fw[st][2 * j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st][2 * i], fw[st][2 * j + 1], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st][2 * i], fw[st][2 * j], acc[i][j], 0, 0, 0);
  };
  // one tap of global chunk g (halo buffer g & 1); `more` = another chunk (of this or of the next tile) follows
  auto body = [&](auto TAP, int g, bool more, bool first) {
    constexpr int T = decltype(TAP)::value;
    const int it = g * NT + T;
    // barrier #it (the very first tap runs straight after the prologue's barrier #0): weight tiles it, it + 1 are in LDS, tile
    // it - 1's slot is free for the producers, this chunk's halo is complete.  No wait: a consumer's outstanding fragment loads
    // read only images that stay valid for another tap.
    if (it > 0) asm volatile("s_barrier" ::: "memory");
    if constexpr (ablate::kInterleave) {
      __builtin_amdgcn_sched_barrier(0);
      reads(IC<1>(), TAP, g & 1, it & (NS - 1));
      mfmas(IC<0>());
      interleave_8_reads_12_mfmas();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (T < NT - 1) reads(IC<0>(), IC<(T + 1) % NT>(), g & 1, (it + 1) & (NS - 1));
      else if (more) reads(IC<0>(), IC<0>(), (g + 1) & 1, (it + 1) & (NS - 1));
      mfmas(IC<1>());
      interleave_8_reads_12_mfmas();
      __builtin_amdgcn_sched_barrier(0);
    } else {
      reads(IC<1>(), TAP, g & 1, it & (NS - 1));
      __builtin_amdgcn_sched_barrier(0);
      mfmas(IC<0>());
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (T < NT - 1) reads(IC<0>(), IC<(T + 1) % NT>(), g & 1, (it + 1) & (NS - 1));
      else if (more) reads(IC<0>(), IC<0>(), (g + 1) & 1, (it + 1) & (NS - 1));
      __builtin_amdgcn_sched_barrier(0);
      mfmas(IC<1>());
    }
    if constexpr (T == NT - 1) {                         // the 288-term partial of this channel chunk
      // (a tile's FIRST partial is assigned, not added to a zeroed total: 0 + x = x, and the epilogue no longer zeroes 64 registers)
      if (first) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { tot[i][j][e] = acc[i][j][e]; acc[i][j][e] = 0.0f; }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { tot[i][j][e] += acc[i][j][e]; acc[i][j][e] = 0.0f; }
      }
    }
  };
  if constexpr (ablate::kConsumerPrio > 0) __builtin_amdgcn_s_setprio(ablate::kConsumerPrio);
  asm volatile("s_barrier" ::: "memory");                // barrier(0): the producers' prologue — tile 0's first halo, weight tiles 0 and 1
  reads(IC<0>(), IC<0>(), 0, 0);
  const int cpg = L.gn_groups > 0 ? d.Cout / L.gn_groups : 64;          // 8, 16, 32 or 64 when the statistics are fused
  const int nsplit = tiles_img * 4;
  int g = 0;
  for (int kt = 0; kt < t_count; ++kt) {
    for (int c = 0; c < nchunks; ++c, ++g) {
      const bool more = g + 1 < nchunks_total;
      const bool first = c == 0;
      body(IC<0>(), g, more, first); body(IC<1>(), g, more, first); body(IC<2>(), g, more, first); body(IC<3>(), g, more, first);
      body(IC<4>(), g, more, first); body(IC<5>(), g, more, first); body(IC<6>(), g, more, first); body(IC<7>(), g, more, first);
      body(IC<8>(), g, more, first);
    }
    // DIRECT epilogue of tile kt (as in the kernels above): register e of lane half hi is pixel row (e & 3) + 8 (e >> 2) + 4 hi of
    // the 32-pixel row tile, lanes 0-31 are 32 consecutive channels: a dword store per register writes two full 128-byte lines per
    // wave-instruction; GroupNorm partial sums by lane reductions, one slab per (tile, wave).  The next tile's first fragments are
    // already in flight, the producers are a chunk / three taps ahead.
    int lin = t_first + kt * t_stride;
    const int tx = lin % tiles_x; lin /= tiles_x;
    const int ty = lin % tiles_y;
    const int b = lin / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    double sd[2], qd[2];
    // Addresses (round 5): pixel p = 64 wave + 32 i + (e & 3) + 8 (e >> 2) + 4 hi of the 16-wide tile is tile row 4 wave + 2 i + (e >> 3),
    // column (e & 3) + 8 ((e >> 2) & 1) + 4 hi, and Cout is 64 here (the launcher's condition): ONE 64-bit base per lane and tile plus the
    // image's row step — every store is base (+ row step) + a compile-time offset.  The generic form (m * d.Cout + ch per element) cost
    // ~350 of the epilogue's 665 VALU instructions per tile and wave, 140 of them quarter-rate integer multiplies.
    float* const obase = L.out + (((size_t)b * d.Hout + y0 + wave * 4) * d.Wout + x0 + 4 * hi) * BN + l31;
    const size_t rowstep = (size_t)d.Wout * BN;                          // floats between image rows
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ch = j * 32 + l31;
      const float bv = L.bias ? L.bias[ch] : 0.0f;
      const float sc = L.split_scale ? L.split_scale[ch] : 1.0f;         // the packer's per-channel power of two, undone (exact)
      float s1 = 0.0f, q1 = 0.0f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float* const r0 = obase + j * 32 + (size_t)(2 * i) * rowstep;
        float* const r1 = r0 + rowstep;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          // (sc is an exact power of two: the fused form rounds once, exactly where the product-then-sum form rounds — same bits)
          const float v = __builtin_fmaf(tot[i][j][e], sc, bv);
          if constexpr (ablate::p64_no_store) {
            if (v == 1.2345e-30f) r0[0] = v;     // (keeps the accumulators alive, stores nothing)
          } else {
            ((e >> 3) ? r1 : r0)[((e & 3) + 8 * ((e >> 2) & 1)) * BN] = v;
          }
          s1 += v;
          q1 = fmaf(v, v, q1);
        }
      }
      sd[j] = (double)s1;
      qd[j] = (double)q1;
    }
    if (fuse_stats) {
      const int width = cpg < 32 ? cpg : 32;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        sd[j] += __shfl_xor(sd[j], 32, 64);
        qd[j] += __shfl_xor(qd[j], 32, 64);
        for (int o = 1; o < width; o <<= 1) {
          sd[j] += __shfl_xor(sd[j], o, 64);
          qd[j] += __shfl_xor(qd[j], o, 64);
        }
      }
      float* slab = L.gn_partials + ((size_t)b * nsplit + (ty * tiles_x + tx) * 4 + wave) * L.gn_groups * 2;
      if (cpg == 64) {
        if (lane == 0) {
          slab[0] = (float)(sd[0] + sd[1]);
          slab[1] = (float)(qd[0] + qd[1]);
        }
      } else if (hi == 0 && (l31 & (width - 1)) == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int gr = (j * 32 + l31) / cpg;
          slab[gr * 2] = (float)sd[j];
          slab[gr * 2 + 1] = (float)qd[j];
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // barrier(niter)
}

// ---------------------------------------------------------------------------------------------
// gather form (1x1, 4x4 stride 2, small / ragged shapes): structure of conv_igemm_kernel (conv.hip)
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, bool ONE>
__global__ __launch_bounds__(256, 2) void conv_igemm_split_kernel(const ConvLaunch<float> L, const int M, const int tiles_m,
                                                                  const int tiles_n, const int fuse_stats) {
  constexpr int BK = 32;
  constexpr int AP = BM / 64, NB = BN * 8 / 256;
  constexpr int WAVES_N = BN / 64, WAVES_M = 4 / WAVES_N;
  constexpr int WM = BM / WAVES_M, TM = WM / 32;
  static_assert(TM >= 1, "wave tile");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* As = reinterpret_cast<uint4*>(smem);   // [2][BM][8]
  uint4* Bs = As + 2 * BM * 8;                  // [2][BN][8]
  float* stage = reinterpret_cast<float*>(smem);

  const ConvDesc& d = L.d;
  const int lin = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tn = lin % tiles_n, tm = lin / tiles_n;
  const int tid = threadIdx.x;
  const int lrow = tid >> 2, q = tid & 3;
  const int Cin = d.C0 + d.C1;
  const int Hl = d.ups ? 2 * d.Hin : d.Hin, Wl = d.ups ? 2 * d.Win : d.Win;
  const int HWo = d.Hout * d.Wout;

  int a_iy0[AP], a_ix0[AP], a_base[AP];
  bool a_ok[AP];
  const float* a_p0[AP];
  const float* a_p1[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = tm * BM + lrow + i * 64;
    a_ok[i] = m < M;
    const int mm = a_ok[i] ? m : 0;
    if constexpr (ONE) {
      a_p0[i] = L.src0 + (int64_t)mm * d.C0 + q * 8;
      a_p1[i] = d.C1 ? L.src1 + (int64_t)mm * d.C1 + q * 8 - d.C0 : a_p0[i];
    } else {
      const int bb = mm / HWo, rem = mm - bb * HWo;
      const int oy = rem / d.Wout, ox = rem - oy * d.Wout;
      a_iy0[i] = oy * d.stride - d.pad;
      a_ix0[i] = ox * d.stride - d.pad;
      a_base[i] = bb * d.Hin * d.Win;
    }
  }

  // Two register stages, branch-free loads, clamped re-issue past the end (conv.hip's conv_igemm_kernel, round 4): the loads of
  // iteration it + 2 are in flight while iteration it computes, and hipcc's s_waitcnt vmcnt(N) counts stay exact.
  typedef __attribute__((ext_vector_type(4))) unsigned int rb_t;   // (native vectors: hipcc kept an array of the uint4 STRUCT in scratch — every weight load waited on, stored, reloaded)
  typedef __attribute__((ext_vector_type(4))) float ra_t;
  ra_t ra0[AP][2], ra1[AP][2];
  rb_t rb0[NB], rb1[NB];
  unsigned ok0 = 0, ok1 = 0;
  const int niter = d.KH * d.KW * L.split_kchunks;
  auto gload = [&](ra_t(&ra)[AP][2], rb_t(&rb)[NB], unsigned& okm, int tap, int kc) {
    okm = 0;
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
    const int c = kc * BK + q * 8;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const float* p = L.src0;
      bool ok;
      if constexpr (ONE) {
        ok = a_ok[i] && c < Cin;
        if (ok) p = (c < d.C0 ? a_p0[i] : a_p1[i]) + kc * BK;
      } else {
        int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
        ok = a_ok[i] && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl && c < Cin;
        if (d.ups) { iy >>= 1; ix >>= 1; }
        const int64_t pix = ok ? (int64_t)a_base[i] + (int64_t)iy * d.Win + ix : 0;
        const float* pp = (c < d.C0) ? L.src0 + pix * d.C0 + c : L.src1 + pix * d.C1 + (c - d.C0);
        if (ok) p = pp;
      }
      ra[i][0] = reinterpret_cast<const ra_t*>(p)[0];
      ra[i][1] = reinterpret_cast<const ra_t*>(p)[1];
      okm |= (ok ? 1u : 0u) << i;
    }
    const rb_t* wt = reinterpret_cast<const rb_t*>(L.w_split + ((size_t)(tap * L.split_kchunks + kc) * d.CoutPad + (size_t)tn * BN) * 64);
#pragma unroll
    for (int j = 0; j < NB; ++j) rb[j] = wt[tid + j * 256];
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, hi = lane >> 5;

  f32x16 acc[TM][2], tot[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.0f; tot[i][j][e] = 0.0f; }

  int tap = 0, kc = 0, nxt = 0;                              // coordinates of the NEXT load to issue (clamped to the last chunk)
  auto advance = [&]() {
    const bool more = nxt + 1 < niter;
    nxt += more ? 1 : 0;
    const int k2 = kc + 1, wrap = k2 == L.split_kchunks;
    kc = more ? (wrap ? 0 : k2) : kc;
    tap = more ? tap + wrap : tap;
  };
  gload(ra0, rb0, ok0, tap, kc);
  advance();
  gload(ra1, rb1, ok1, tap, kc);
  advance();
  auto step = [&](ra_t(&ra)[AP][2], rb_t(&rb)[NB], unsigned& okm, int it) {
    uint4* Ab = As + (it & 1) * BM * 8;
    uint4* Bb = Bs + (it & 1) * BN * 8;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int r = lrow + i * 64, sw = (r >> 1) & 7;
      const float z = ((okm >> i) & 1u) ? 1.0f : 0.0f;      // padding, rows past M, channels past Cin (the stand-in load is finite)
      const float v[8] = {ra[i][0][0] * z, ra[i][0][1] * z, ra[i][0][2] * z, ra[i][0][3] * z,
                          ra[i][1][0] * z, ra[i][1][1] * z, ra[i][1][2] * z, ra[i][1][3] * z};
      uint4 vh, vl;
      split8(v, vh, vl);
      Ab[r * 8 + (q ^ sw)] = vh;
      Ab[r * 8 + ((4 + q) ^ sw)] = vl;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int u = tid + j * 256, n = u >> 3, slot = u & 7;
      *reinterpret_cast<rb_t*>(Bb + n * 8 + (slot ^ ((n >> 1) & 7))) = rb[j];
    }
    __syncthreads();
    gload(ra, rb, okm, tap, kc);
    advance();
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int unit = 2 * st + hi;
      f16x8 ah[TM], al[TM], wh[2], wl[2];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = wm * WM + i * 32 + l31, sw = (r >> 1) & 7;
        ah[i] = ld_frag(Ab + r * 8 + (unit ^ sw));
        al[i] = ld_frag(Ab + r * 8 + ((4 + unit) ^ sw));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = wn * 64 + j * 32 + l31, sw = (n >> 1) & 7;
        wh[j] = ld_frag(Bb + n * 8 + (unit ^ sw));
        wl[j] = ld_frag(Bb + n * 8 + ((4 + unit) ^ sw));
      }
      mma3_tiles<TM>(ah, al, wh, wl, acc);
    }
    if ((it & 7) == 7 || it + 1 == niter) flush_acc<TM>(acc, tot);   // 256-term partials
  };
  for (int it = 0; it + 1 < niter; it += 2) {
    step(ra0, rb0, ok0, it);
    step(ra1, rb1, ok1, it + 1);
  }
  if (niter & 1) step(ra0, rb0, ok0, niter - 1);

  double gs, gq;
  auto row_to_m = [&](int r) -> int64_t {
    const int m = tm * BM + wm * WM + r;
    return m < M ? (int64_t)m : (int64_t)-1;
  };
  if (L.split_scale) {                         // undo the packer's per-channel power-of-two weight scale (exact)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float sc = L.split_scale[tn * BN + wn * 64 + j * 32 + l31];      // [CoutPad]: padded columns read 1
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) tot[i][j][e] *= sc;
    }
  }
  epilogue_store<float, TM>(L, tot, stage + wave * 32 * 68, lane, tn * BN + wn * 64, row_to_m, gs, gq);
  if (fuse_stats) {
    const int nsplit = HWo / BM;
    const int bimg = (tm * BM) / HWo, slab = tm - bimg * nsplit;
    float* dst = L.gn_partials + ((size_t)bimg * nsplit + slab) * L.gn_groups * 2;
    epilogue_stats<WAVES_M, WAVES_N>(reinterpret_cast<double*>(stage + 4 * 32 * 68), gs, gq, wave, lane, tn * BN, d.Cout,
                                     L.gn_groups, dst);
  }
}

// ---------------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------------
template <int TH, int TW>
static int launch_split_halo(const ConvLaunch<float>& L, hipStream_t s, int fuse_stats, int* nsplit) {
  [[maybe_unused]] constexpr int HALO = (TH + 2) * (TW + 2);
  constexpr int RSTRIDE = ((TW + 2) * 144 + 255) / 256 * 256, HB = (TH + 2) * RSTRIDE;
  constexpr int NS = (2 * HB + 3 * 8192) * 2 <= 160 * 1024 ? 3 : 2;   // weight ring slots: two workgroups must fit a CU
  const ConvDesc& d = L.d;
  const int tiles_x = d.Wout / TW, tiles_y = d.Hout / TH, tiles_n = d.Cout / 64;
  size_t lds = (size_t)2 * HB + NS * 8192 + 512;      // (+ the statistics scratch of the direct epilogue)
  if (lds < kEpilogueLds) lds = kEpilogueLds;
  if (nsplit) *nsplit = fuse_stats ? tiles_x * tiles_y : 0;
  static DeviceOnce attr_done;   // > 64 KB of dynamic LDS needs the opt-in (idempotent: a race between lanes is benign)
  if (!attr_done.done()) {
    PRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_kernel<TH, TW, NS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    attr_done.mark();
    if (std::getenv("PRG_SPLIT_DEBUG")) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv3x3_split_kernel<TH, TW, NS>, 256, lds);
      fprintf(stderr, "conv3x3_split_kernel<%d,%d,%d>: %zu bytes of LDS, %d workgroups per CU\n", TH, TW, NS, lds, nb);
    }
  }
  conv3x3_split_kernel<TH, TW, NS><<<dim3(tiles_x * tiles_y * tiles_n * d.B), 256, lds, s>>>(L, tiles_x, tiles_y, tiles_n, fuse_stats);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

// Upsample conv as four 2 x 2-tap sub-pixel convolutions of the source image (the UP form of the wave-specialised kernel)
static int launch_split_ws_up(const ConvLaunch<float>& L, hipStream_t s) {
  constexpr int NS = 4, HB = 10 * ((18 * 144 + 255) / 256 * 256);
  const ConvDesc& d = L.d;
  const int tiles_x = d.Win / 16, tiles_y = d.Hin / 8, tiles_n = d.Cout / 128;
  const size_t lds = (size_t)2 * HB + NS * 128 * 128;
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    PRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_ws_kernel<NS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    attr_done.mark();
  }
  conv3x3_split_ws_kernel<NS, true><<<dim3(tiles_x * tiles_y * tiles_n * d.B * 4), 512, lds, s>>>(L, tiles_x, tiles_y, tiles_n, 0);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

static int launch_split_ws(const ConvLaunch<float>& L, hipStream_t s, int fuse_stats, int* nsplit) {
  constexpr int NS = 4, HB = 10 * ((18 * 144 + 255) / 256 * 256);
  const ConvDesc& d = L.d;
  const int tiles_x = d.Wout / 16, tiles_y = d.Hout / 8, tiles_n = d.Cout / 128;
  const size_t lds = (size_t)2 * HB + NS * 128 * 128;
  if (nsplit) *nsplit = fuse_stats ? tiles_x * tiles_y * 2 : 0;
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    PRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_ws_kernel<NS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    attr_done.mark();
  }
  conv3x3_split_ws_kernel<NS, false><<<dim3(tiles_x * tiles_y * tiles_n * d.B), 512, lds, s>>>(L, tiles_x, tiles_y, tiles_n, fuse_stats);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

// the persistent Cout = 64 kernel: 16 x 16-pixel tiles, one 512-thread workgroup per CU
static int launch_split_p64(const ConvLaunch<float>& L, hipStream_t s, int fuse_stats, int* nsplit) {
  constexpr int NS = 4, HB = 18 * ((18 * 144 + 255) / 256 * 256);
  const ConvDesc& d = L.d;
  const int tiles_x = d.Wout / 16, tiles_y = d.Hout / 16;
  const int ntiles = tiles_x * tiles_y * d.B;
  const size_t lds = (size_t)2 * HB + NS * 64 * 128;
  if (nsplit) *nsplit = fuse_stats ? tiles_x * tiles_y * 4 : 0;
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    PRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_p64_kernel<NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
    attr_done.mark();
  }
  const int num_cus = device_cu_count();
  PRG_CHECK(num_cus > 0, "conv3x3_split_p64: no device properties");
  int grid = ntiles < num_cus ? ntiles : num_cus;
  if (grid >= 8) grid &= ~7;                             // multiple of 8: XCD-contiguous tile runs
  conv3x3_split_p64_kernel<NS><<<dim3(grid), 512, lds, s>>>(L, tiles_x, tiles_y, ntiles, fuse_stats);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

template <int BM, int BN>
static int launch_split_igemm(const ConvLaunch<float>& L, int M, hipStream_t s, int want_stats, int* nsplit) {
  const ConvDesc& d = L.d;
  const int tiles_m = ceil_div(M, BM), tiles_n = d.CoutPad / BN;
  const int HWo = d.Hout * d.Wout;
  const int cpg = L.gn_groups > 0 ? d.Cout / L.gn_groups : 0;
  const int fuse = want_stats && cpg % 8 == 0 && cpg <= BN && HWo % BM == 0 && HWo / BM <= kGnMaxSplit;
  if (nsplit) *nsplit = fuse ? HWo / BM : 0;
  size_t lds = (size_t)2 * (BM + BN) * 8 * 16;
  if (lds < kEpilogueLds) lds = kEpilogueLds;
  const bool one = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0 && !d.ups;
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    PRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_split_kernel<BM, BN, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    PRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_split_kernel<BM, BN, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    attr_done.mark();
  }
  if (one) conv_igemm_split_kernel<BM, BN, true><<<dim3(tiles_m * tiles_n), 256, lds, s>>>(L, M, tiles_m, tiles_n, fuse);
  else conv_igemm_split_kernel<BM, BN, false><<<dim3(tiles_m * tiles_n), 256, lds, s>>>(L, M, tiles_m, tiles_n, fuse);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

int try_launch_conv3x3_split_w512(const ConvLaunch<float>& L, hipStream_t s, int want_stats, int* gn_nsplit_out);   // conv_split512.hip

// 1 = launched, 0 = shape not covered (the exact-f32 kernels of conv.hip run it), < 0 = error
int try_launch_conv_split(const ConvLaunch<float>& L, hipStream_t s, int* gn_nsplit_out) {
  if (!L.w_split) return 0;
  const ConvDesc& d = L.d;
  if (d.Cout % 8 || d.C0 % 8 || d.C1 % 8 || L.res_fold.acc || L.pro_fold.acc) return 0;
  // the activated residual (ConvLaunch::res_a: the ResnetBlock tail in a 1x1 res_conv's epilogue) is the shared transposing
  // epilogue's feature: the gather kernel below has it, the 3x3 kernels do not
  if (L.res_a && !(d.KH == 1 && d.KW == 1)) return 0;
  // precision budget (tools/gpu_r5_precision.sh): PRG_SPLIT_EXACT is a bit mask of convolution classes handed back to the exact-f32
  // kernels — 1: 3x3 / s1 (Block.proj, the last down conv, the top level's "Upsample"), 2: 1x1 (res_conv, attention projections of
  // the unfused blocks), 4: 4x4 / s2 (Downsample), 8: 3x3 on the upsampled image (Upsample).  0 (default) = every class split.
  static const int exact_mask = [] { const char* e = std::getenv("PRG_SPLIT_EXACT"); return e ? std::atoi(e) : 0; }();
  if (exact_mask && !L.probe) {
    const int cls = d.KH == 1 ? 2 : d.KH == 4 ? 4 : d.ups ? 8 : 1;
    if (exact_mask & cls) return 0;
  }
  const int64_t M64 = (int64_t)d.B * d.Hout * d.Wout;
  if ((int64_t)d.B * d.Hin * d.Win >= ((int64_t)1 << 31) || M64 >= ((int64_t)1 << 31)) return 0;
  const int M = (int)M64;
  const int want_stats = L.gn_partials != nullptr;
  int rc = PRG_OK;
  const bool halo = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.C0 % 32 == 0 && d.C1 % 32 == 0 && d.Cout % 64 == 0;
  if (halo) {
    const int H = d.Hout, W = d.Wout;
    const int cpg = L.gn_groups > 0 ? d.Cout / L.gn_groups : 0;
    auto fuse = [&](int TH, int TW, int BN) {
      return want_stats && cpg % 8 == 0 && cpg <= BN && (W / TW) * (H / TH) <= kGnMaxSplit;
    };
    // Cout % 128 == 0: the wave-specialised kernel (128-pixel x 128-channel tiles, one workgroup per CU); PRG_SPLIT_WS=0: never
    static const int ws_on = [] { const char* e = std::getenv("PRG_SPLIT_WS"); return e ? std::atoi(e) : 1; }();
    // The Upsample convs as four 2 x 2-tap sub-pixel convolutions (4 / 9 of their MFMAs: 0.3 ms = 2.5 % of an f16x3 evaluation): OFF by
    // default.  Per evaluation it is as accurate as the nine-tap form (test_split_upsample_conv_against_float64: rms 2.0e-7 against
    // torch-CPU's 2.9e-7), but every change of the rounding pattern RE-DRAWS the long chains' distance from the reference — over nine
    // equal-precision variants of the mode's arithmetic the 250-step 256 x 256 chain G21b sits anywhere in 3.7e-5 .. 8.8e-5 m at B = 1
    // (profiles/r05_chain_metric_spread_f16x3.txt; the reference's own 1-vs-8-thread spread is 4.7e-5 m) — and round 5 tried it as the
    // default: G21b drew 5.9e-5 m at B = 1 and 1.001e-4 m at B = 8 (the batch that selects the persistent 64-channel kernel).  The
    // mode's claim is the LITERAL 1e-4 m on that chain at the tested batches, so the nine-tap form stays.  PRG_SPLIT_UP2X2=1: on.
    static const int up_on = [] { const char* e = std::getenv("PRG_SPLIT_UP2X2"); return e ? std::atoi(e) : 0; }();
    if (ws_on && up_on && d.ups && L.w_up_split && d.C1 == 0 && d.Cout % 128 == 0 && d.Win % 16 == 0 && d.Hin % 8 == 0 && d.Hout == 2 * d.Hin &&
        d.Wout == 2 * d.Win && !L.residual && !L.pro_a && !want_stats) {
      rc = launch_split_ws_up(L, s);
      return rc ? rc : 1;
    }
    {
      // conv_split512.hip (round 6): one wave per SIMD, 128 x 64 wave tiles, for launches with a 256-pixel tile per CU; bit-identical to
      // the kernel below (PRG_SPLIT_W512=0: off)
      const int r = try_launch_conv3x3_split_w512(L, s, want_stats, gn_nsplit_out);
      if (r != 0) return r;
    }
    if (ws_on && d.Cout % 128 == 0 && W % 16 == 0 && H % 8 == 0 && !L.residual) {
      const int tiles = (W / 16) * (H / 8) * 2;
      const int f = want_stats && cpg % 8 == 0 && cpg <= 64 && (cpg & (cpg - 1)) == 0 && tiles <= kGnMaxSplit;
      if (f || !want_stats) { rc = launch_split_ws(L, s, f, gn_nsplit_out); return rc ? rc : 1; }
    }
    // Cout = 64 and a launch that fills the chip (>= one 16 x 16 tile per CU): the persistent kernel; PRG_SPLIT_P64=0: never,
    // PRG_SPLIT_P64=<n>: from n tiles (tests force it at small shapes)
    static const int p64_min = [] { const char* e = std::getenv("PRG_SPLIT_P64"); return e ? std::atoi(e) : 256; }();
    if (p64_min > 0 && d.Cout == 64 && W % 16 == 0 && H % 16 == 0 && !L.residual && !L.res_a && d.B * (W / 16) * (H / 16) >= p64_min &&
        (d.C0 + d.C1) >= 64) {
      const int tiles = (W / 16) * (H / 16) * 4;
      const int f = want_stats && cpg % 8 == 0 && cpg <= 64 && (cpg & (cpg - 1)) == 0 && tiles <= kGnMaxSplit;
      if (f || !want_stats) { rc = launch_split_p64(L, s, f, gn_nsplit_out); return rc ? rc : 1; }
    }
    static const int pref32 = [] { const char* e = std::getenv("PRG_SPLIT_TW32"); return e ? std::atoi(e) : 0; }();
    if (pref32 && W % 32 == 0 && H % 4 == 0) { rc = launch_split_halo<4, 32>(L, s, fuse(4, 32, 64), gn_nsplit_out); return rc ? rc : 1; }
    if (W % 16 == 0 && H % 8 == 0) { rc = launch_split_halo<8, 16>(L, s, fuse(8, 16, 64), gn_nsplit_out); return rc ? rc : 1; }
    if (W % 32 == 0 && H % 4 == 0) { rc = launch_split_halo<4, 32>(L, s, fuse(4, 32, 64), gn_nsplit_out); return rc ? rc : 1; }
  }
  if (L.pro_a) return 0;                       // (a fused prologue is only requested where pick_halo<float> = the test above holds)
  if (d.CoutPad % 128 == 0) rc = launch_split_igemm<128, 128>(L, M, s, want_stats, gn_nsplit_out);
  else rc = launch_split_igemm<128, 64>(L, M, s, want_stats, gn_nsplit_out);
  return rc ? rc : 1;
}

// ---------------------------------------------------------------------------------------------
// weight packing (host): [tap][32-channel chunk][CoutPad][32 hi | 32 lo] as f16 bit patterns, zero padded
// ---------------------------------------------------------------------------------------------
static inline uint16_t f16_bits(_Float16 h) {
  uint16_t u;
  std::memcpy(&u, &h, 2);
  return u;
}
void pack_conv_weight_split(const float* w, int Cout, int Cin, int KH, int KW, std::vector<uint16_t>& out, int* CoutPad, int* kchunks32,
                            std::vector<float>* oscale) {
  const int cp = (Cout + 63) / 64 * 64;
  const int kcn = (Cin + 31) / 32;
  out.assign((size_t)KH * KW * kcn * cp * 64, 0);
  if (oscale) oscale->assign((size_t)cp, 1.0f);
  // Round 5: |a - hi - lo| <= 2^-22 |a| only holds while the lo half is a NORMAL f16; below |a| ~ 2^-3 it is
  // subnormal with an absolute floor of 2^-25, so unstandardised weights of ~1 / sqrt(fan_in) (res_conv, Downsample, Upsample:
  // 0.02-0.1) kept 18-20 bits.  Each output channel's weights are therefore multiplied by an exact power of two that brings
  // max|w| into [2^9, 2^10) (hi exact to 2^-2, lo's 2^-25 floor = 2^-35 of the maximum; |w| < 2^10 stays far from f16's 65504
  // with O(100) activations in the other operand being a float32-accumulated product), and the kernels multiply the float32
  // total by the inverse (exact) before the bias: split_scale.
  static const int wscale_on = [] { const char* e = std::getenv("PRG_SPLIT_WSCALE"); return e ? std::atoi(e) : 1; }();
  std::vector<float> mul((size_t)Cout, 1.0f);
  if (oscale && wscale_on) {
    const size_t per = (size_t)Cin * KH * KW;
    for (int n = 0; n < Cout; ++n) {
      float m = 0.0f;
      for (size_t i = 0; i < per; ++i) m = std::fmax(m, std::fabs(w[(size_t)n * per + i]));
      // (m < 2^-100: a dead / denormal channel — 2^k would overflow to +inf and 0 * inf = NaN: left unscaled,
      //  its halves flush to zero like the products they stand for)
      if (m >= 0x1p-100f && std::isfinite(m)) {
        int e = 0;
        (void)std::frexp(m, &e);                 // m = f * 2^e, f in [0.5, 1)
        const int k = 10 - e;                    // m * 2^k in [2^9, 2^10); |k| <= 110
        mul[n] = std::ldexp(1.0f, k);
        (*oscale)[n] = std::ldexp(1.0f, -k);
      }
    }
  }
  for (int kh = 0; kh < KH; ++kh)
    for (int kw = 0; kw < KW; ++kw)
      for (int kc = 0; kc < kcn; ++kc)
        for (int n = 0; n < Cout; ++n)
          for (int k = 0; k < 32; ++k) {
            const int c = kc * 32 + k;
            if (c >= Cin) break;
            const float v = w[(((size_t)n * Cin + c) * KH + kh) * KW + kw] * mul[n];
            const _Float16 h = (_Float16)v;                       // round to nearest even (compiler-rt / F16C)
            const _Float16 l = (_Float16)(v - (float)h);
            const size_t row = ((((size_t)(kh * KW + kw)) * kcn + kc) * cp + n) * 64;
            out[row + k] = f16_bits(h);
            out[row + 32 + k] = f16_bits(l);
          }
  *CoutPad = cp;
  *kchunks32 = kcn;
}

}  // namespace prg

