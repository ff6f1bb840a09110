// conv_ws.hip — wave-specialised persistent 3x3 convolution for the bf16 throughput path (MFMA roofline).
//
// One 768-thread workgroup per CU, resident for the whole launch, walks a strided list of output tiles
// (TH x TW pixels of one image x BN output channels).  A step = one 64-channel chunk of Cin for one tile, a phase = one
// tap of a step (nine phases, one raw s_barrier each).  Its twelve waves (three per SIMD, <= 168 VGPRs) have fixed roles:
//
//   waves 0-3   CONSUMERS, one per SIMD.  Per phase 16 ds_read_b128 fragment loads and 16 v_mfma_f32_32x32x16_bf16, the
//               nine taps fully unrolled (tap offsets, ring slots and fragment sets are compile-time constants, every
//               LDS offset an immediate) with a sched_barrier-pinned one-load-one-MFMA interleave; the fragments of
//               call c+2 are loaded while call c multiplies, across the phase barrier included, and a consumer never
//               waits for its own LDS reads at a barrier.  The operands are swapped — weights are the MFMA "A" rows,
//               pixels the "B" columns — so a lane ends up with four consecutive channels of one pixel: at the end of
//               a tile it adds the bias (kept in LDS: a workgroup owns one channel tile), rounds to bf16, transposes
//               through a per-wave LDS stage, and folds the GroupNorm partial sums of the output with a 17-exchange
//               halving butterfly (DPP / ds_swizzle; fixed order: deterministic) straight to global.
//   waves 4-11  PRODUCERS, each stages an eighth of every weight tile and of every halo and drains finished tiles:
//               * weights: six register sets hold the tiles of phases ph+2..ph+7; phase ph writes tile ph+2 into LDS ring
//                 slot (ph+2) % 3 and re-issues that set for tile ph+8;
//               * halo, "write, then re-issue": during step s the units of halo s+1 are written to LDS (after the fused
//                 GroupNorm + (scale+1, shift) + SiLU of the previous Block, sd:690-696 — each halo pixel once, not
//                 nine times) from registers loaded one full step earlier, and each register is immediately re-issued
//                 for the same unit of halo s+2: nine phases of MFMAs between an HBM load and its use;
//               * drain: the tile the consumers left in the LDS stage goes to HBM in full 128-byte rows during the
//                 first four phases of the next step, under that tile's MFMAs.
//
// All producer loads are inline asm with hand-counted `s_waitcnt vmcnt(N)`: every producer wave issues loads only,
// in a fixed periodic order, so "N younger loads may stay in flight" is an exact constant.  (hipcc's own waitcnt
// bookkeeping degrades to vmcnt(0) at the merges of such unrolled, branchy streams, which drains the prefetch queue
// every phase and exposes the whole memory latency nine times per step — measured: 2x on the kernel.)
// One raw s_barrier per phase (lgkmcnt(0) only, never vmcnt) is the only synchronisation.
// LDS rows are padded to 144 bytes (WsGeom): conflict-free ds_read_b128 with purely immediate tap/unit/slot offsets.
//
// Covers every 3x3 / stride 1 / pad 1 conv of the U-Nets at dim = 64 (widths multiples of 64): Block.proj, the last
// down/up convs, Upsample's conv (x2 nearest gather folded into the halo load), skip concat as two sources.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "conv.h"

namespace prg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {

constexpr int kCH = 64;   // channels per 128-byte pixel row (bf16)

// Timing experiments only (results become garbage): -DPRG_WS_EXP=1 drops the weight waits, 2 the halo-unit waits,
// 256 the producers' whole main loop, 512 the consumers' MFMA loop, 4 the fused prologue arithmetic, 8 the producers' LDS writes, 16 the whole tile epilogue, 32 its statistics.
#ifndef PRG_WS_EXP
#define PRG_WS_EXP 0
#endif
constexpr bool kExpNoWaitW = (PRG_WS_EXP & 1) != 0, kExpNoWaitU = (PRG_WS_EXP & 2) != 0;
constexpr bool kExpNoPro = (PRG_WS_EXP & 4) != 0, kExpNoLdsWrite = (PRG_WS_EXP & 8) != 0;
constexpr bool kExpNoEpilogue = (PRG_WS_EXP & 16) != 0, kExpNoStats = (PRG_WS_EXP & 32) != 0;

__device__ inline float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ inline float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ inline uint32_t pack_bf16(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
// lane exchanges inside a wave without the LDS crossbar's address VGPR: DPP quad permute / ds_swizzle xor mask
template <int CTRL>
__device__ inline float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int XOR>   // partner = lane ^ XOR, XOR < 32
__device__ inline float swz_xor(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (XOR << 10) | 0x1F));
}
__device__ inline float fast_silu(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// Optional barrier trace (PRG_WS_TRACE=<launch index>): workgroup 0 records, per wave, the shader clock when it
// arrives at and when it leaves every phase barrier of that launch (tools/ws_trace.py shows who the others wait for).
// The record is stores only (device memory, count kept in a register): a load would put its latency into every phase
// of the traced workgroup.  Stores raise vmcnt, which only makes the producers' counted waits stricter.
// Compiled in only with -DPRG_WS_TRACE_BUILD=1 (`make trace` -> libprg_hip_trace.so): the checks alone are a dozen
// instructions per phase and wave.
#ifndef PRG_WS_TRACE_BUILD
#define PRG_WS_TRACE_BUILD 0
#endif
constexpr bool kTrace = PRG_WS_TRACE_BUILD != 0;
constexpr int kTraceStride = 4096;   // u64 slots per wave: [0] = count, then (arrive, leave) pairs

struct TraceCtx {
  unsigned long long* p;   // this wave's slots, or nullptr
  int n;
  __device__ __forceinline__ TraceCtx(unsigned long long* base)
      : p(kTrace && base != nullptr && blockIdx.x == 0 && (threadIdx.x & 63) == 0 ? base + (threadIdx.x >> 6) * kTraceStride
                                                                                   : nullptr),
        n(0) {}
  // two extra time stamps inside the phase that ends at barrier n (producers: after the weight wait, before the LDS drain)
  __device__ __forceinline__ void mark(int which) {
    if (kTrace && p != nullptr && n < 1000) p[kTraceStride / 2 + 2 * n + which] = clock64();
  }
};

template <bool LDS_DONE = true>   // wait for this wave's own LDS operations first (writers always must)
__device__ __forceinline__ void phase_barrier(TraceCtx& tr) {
  if constexpr (LDS_DONE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const bool rec = kTrace && tr.p != nullptr && tr.n < (kTraceStride - 4) / 2;
  if (rec) tr.p[1 + 2 * tr.n] = clock64();
  if (!(PRG_WS_EXP & 4096)) __builtin_amdgcn_s_barrier();
  if (rec) {
    tr.p[2 + 2 * tr.n] = clock64();
    if (tr.n == 0) tr.p[kTraceStride - 2] = wall_clock64();   // 100 MHz reference: the trace also yields the shader clock
    tr.p[kTraceStride - 1] = wall_clock64();
    ++tr.n;
    tr.p[0] = tr.n;
  }
  asm volatile("" ::: "memory");
}

template <int TH, int TW, int BN>
struct WsGeom {
  static constexpr int BM = TH * TW;
  static constexpr int HP = TW + 2;
  static constexpr int HALO = (TH + 2) * HP;
  static constexpr int WAVES_N = BN / 64, WAVES_M = 4 / WAVES_N;
  // LDS rows (one halo pixel / one weight row = 64 bf16) are padded from 128 to 144 bytes instead of XOR-swizzled:
  // 36-dword stride makes every ds_read_b128 lane group (16 rows, one 16-byte unit each) hit 64 distinct banks
  // (9 r mod 16 is a permutation), and keeps addresses linear so taps, units and ring slots are immediate offsets.
  static constexpr int ROWB = 144;
  static constexpr size_t AH_BYTES = (size_t)HALO * ROWB;
  static constexpr size_t BW_BYTES = (size_t)BN * ROWB;
  static constexpr size_t RED_BYTES = 0;
  static constexpr size_t STG_BYTES = 4 * 64 * 128;                   // per consumer wave: 64 pixels x 64 channels bf16
  static constexpr size_t BIAS_BYTES = BN * sizeof(float);            // bias of the workgroup's (fixed) channel tile
  static constexpr size_t LDS = 2 * AH_BYTES + 3 * BW_BYTES + RED_BYTES + STG_BYTES + BIAS_BYTES;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(BM / WAVES_M == 64, "consumer wave tile is 64 pixels x 64 channels");
};

// Tile schedule of one workgroup.  Round r covers a contiguous run of tiles per XCD (workgroup b runs on XCD b % 8:
// observed placement, used for speed only).  When the conv has several output-channel tiles, every XCD is pinned to
// ONE of them (tn = xcd % tiles_n) and walks pixel tiles only: its 32 CUs then stream the same weight slice at
// the same time, so the slice stays resident in the XCD's 4 MB L2.
struct TileCur { int n, x, y, b; };   // mixed-radix digits of a tile index: channel tile, tile column, tile row, image

struct TileMap {
  int tiles_x, tiles_y, tiles_n, TH, TW;
  int pinned, tn_fixed;
  int first, stride, count;   // tile (or pixel-tile) index of iteration it = first + it * stride, it < count
  TileCur c0, dc;             // digits of `first` and of `stride`: iterating is digit-wise addition, no divisions

  __device__ __forceinline__ void digits(int t, TileCur& c) const {
    if (pinned) {
      c.n = 0;
    } else {
      c.n = t % tiles_n;
      t /= tiles_n;
    }
    c.x = t % tiles_x; t /= tiles_x;
    c.y = t % tiles_y;
    c.b = t / tiles_y;
  }
  __device__ __forceinline__ void next(TileCur& c) const {
    int carry = 0;
    if (!pinned) {
      c.n += dc.n;
      if (c.n >= tiles_n) { c.n -= tiles_n; carry = 1; }
    }
    c.x += dc.x + carry; carry = 0;
    if (c.x >= tiles_x) { c.x -= tiles_x; carry = 1; }
    c.y += dc.y + carry; carry = 0;
    if (c.y >= tiles_y) { c.y -= tiles_y; carry = 1; }
    c.b += dc.b + carry;
  }
  __device__ __forceinline__ void fill(const TileCur& c, int& b, int& y0, int& x0, int& tn) const {
    b = c.b; y0 = c.y * TH; x0 = c.x * TW;
    tn = pinned ? tn_fixed : c.n;
  }

  __device__ __forceinline__ void init(int bid, int GR, int tx, int ty, int tn, int nb, int th, int tw) {
    tiles_x = tx; tiles_y = ty; tiles_n = tn; TH = th; TW = tw;
    const int npix = tx * ty * nb;
    const bool xcd_ok = (GR & 7) == 0;
    const int per_xcd = GR >> 3;
    const int xcd = bid & 7, idx = bid >> 3;
    if (xcd_ok && tn > 1 && (8 % tn) == 0) {
      pinned = 1;
      tn_fixed = xcd % tn;
      const int gx = 8 / tn, member = xcd / tn;
      first = member * per_xcd + idx;
      stride = gx * per_xcd;
      count = first < npix ? (npix - first + stride - 1) / stride : 0;
    } else {
      pinned = 0;
      tn_fixed = 0;
      first = xcd_ok ? xcd * per_xcd + idx : bid;
      stride = GR;
      const int total = npix * tn;
      count = first < total ? (total - first + stride - 1) / stride : 0;
    }
    digits(first, c0);
    digits(stride, dc);
  }
  __device__ __forceinline__ void decode(int it, int& b, int& y0, int& x0, int& tn) const {
    int t = first + it * stride;
    if (pinned) {
      tn = tn_fixed;
    } else {
      tn = t % tiles_n;
      t /= tiles_n;
    }
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    b = t / tiles_y;
    y0 = ty * TH; x0 = tx * TW;
  }
};

// =====================================================================================================
// producer waves: every one of the four stages a quarter of each weight tile and a quarter of each halo
// =====================================================================================================
template <int TH, int TW, int BN, bool PRO>
struct Producer {
  using G = WsGeom<TH, TW, BN>;
  static constexpr int HP = G::HP, HALO = G::HALO;
  // EIGHT producer waves (two per SIMD next to one consumer wave: 768 threads, <= 168 VGPRs): a producer's phase is a
  // dependent chain (wait -> LDS write -> address -> issue) that runs at 16-35 clk per instruction with one wave per
  // SIMD; two waves with half the work each overlap their chains.
  static constexpr int NPT = 512;                      // producer threads
  static constexpr int NWL = BN * 8 / NPT;             // weight units per thread per tap (BN rows x 8 units / 512)
  static constexpr int RPP = NPT / 8;                  // halo rows per pass (8 lanes per 128-byte row)
  static constexpr int DRAIN_PHASES = 2048 / NPT;      // 16-byte units of a finished tile per thread
  static constexpr int KU = (HALO + RPP - 1) / RPP;    // halo units per thread
  // units handled per phase (phases 0..7), at least two: the fused prologue of ONE unit is a latency-bound dependent
  // chain (~840 clk measured, tools/micro/coissue.hip); two units interleave to about the VALU throughput bound
  static constexpr int UPH = (KU + 7) / 8 > 2 ? (KU + 7) / 8 : 2;
  static constexpr int NCO = PRO ? 4 : 0;              // coefficient loads per step
  // The load stream of a wave is periodic with period one step:
  //   phase 0: [coefficients of halo s+2] [weight tile ph+5] [units of phase 0]; phase p: [weight tile ph+5] [units of p]
  static constexpr int PER_STEP = 9 * NWL + KU + NCO;
  static constexpr int U_YOUNGER = PER_STEP - 1;       // a unit is used exactly one period after its issue
  static_assert(U_YOUNGER <= 63, "vmcnt range");
  static constexpr int n_units(int p) {
    p = ((p % 9) + 9) % 9;
    if (p == 8) return 0;
    const int lo = p * UPH, hi = (p + 1) * UPH < KU ? (p + 1) * UPH : KU;
    return hi > lo ? hi - lo : 0;
  }
  static constexpr int n_coef(int p) { return ((p % 9) + 9) % 9 == 0 ? NCO : 0; }
  // Weight tiles are prefetched WD phases ahead in WD register sets (tile t lives in set t % WD).  Three sets made the
  // producers latency-bound: alone (consumers idle) they needed a phase of (load latency / 3) ~ 700 clk.
  static constexpr int WD = 6;
  // weight tile ph+2 was issued (after the coefficients, before the units) in phase ph-WD; everything after it is younger
  static constexpr int w_younger(int ph) {
    int n = n_units(ph - WD) + n_coef(ph);
    for (int j = 1; j < WD; ++j) n += n_coef(ph - j) + NWL + n_units(ph - j);
    return n;
  }
  static_assert(9 % 3 == 0 && 18 % WD == 0, "set index must be a compile-time function of (g & 1, phase)");

  const ConvLaunch<bf16_t>& L;
  const ConvDesc& d;
  const TileMap& tm;
  char* Ah0;
  char* Bw0;
  TraceCtx& trace;
  u32x4 wset[WD][NWL];
  u32x4 hreg[KU];
  u32x4 cf[2][4];            // [halo parity][a0..3, a4..7, b0..3, b4..7] as raw bits
  // Every global access is "wave-uniform SGPR base + per-thread 32-bit VGPR offset (+ immediate)": the bases come from
  // a few scalar instructions per step, the offsets are constants of the thread, so a phase spends almost no VALU
  // issue slots on addresses.
  unsigned hpix[KU];         // source-pixel offset of unit k relative to the halo origin (tile origin - (1,1)): >= 0
  unsigned w_voff;           // byte offset of this thread's first weight unit inside a tap's [2][CoutPad][32] slab
  unsigned cf_voff;          // byte offset of this thread's 8 coefficients
  unsigned dr_voff;          // byte offset of this thread's drain unit relative to a 32-pixel half of a wave tile
  unsigned hedge[KU];        // which tile edges (or the halo end) invalidate unit k
  unsigned hvalid, hvalid_nxt;
  const char* ld_base;       // per-step context of the halo being ISSUED: address of channel 0 of this chunk at the halo origin
  unsigned ld_cs2;           // bytes per source pixel
  unsigned ld_dummy;         // pixel offset of the tile origin itself (always inside the image): stand-in for padding taps
  const char* ld_ca;         // its coefficient rows
  const char* ld_cb;
  unsigned ld_tedge;
  int slot, row, nsteps, nchunks, Hl, Wl;
  // wave-uniform bookkeeping, advanced by counters (no divisions in the loop): steps g, g+1, g+2 and the finished tile
  struct StepInfo { int chunk, b, y0, x0, tn; TileCur cur; };
  StepInfo sA, sB, sC, dr;
  int gC;                    // step index of sC (clamped to the last step)
  bool drain_on;
  const char* wptr;          // running address of the weight tile being issued (advances one tap per phase)
  size_t w_tapb;             // bytes between consecutive taps of the packed weights
  char* dr_base;             // output address of pixel (y0, x0), channel tn*BN of the tile being drained
  size_t dr_rowb;            // output bytes per image row
  char* stage_rd;            // this thread's 16-byte unit of the consumers' output stage (row ptid >> 3 of a 32-row half)
  int dr_r0;                 // ... and its row

  __device__ __forceinline__ Producer(const ConvLaunch<bf16_t>& L_, char* smem, int ptid, const TileMap& tm_, int nsteps_,
                                      int nchunks_, TraceCtx& tr)
      : L(L_), d(L_.d), tm(tm_), trace(tr), nsteps(nsteps_), nchunks(nchunks_) {
    Ah0 = smem + (ptid >> 3) * G::ROWB + (ptid & 7) * 16;            // this thread's unit of halo row `row`
    Bw0 = smem + 2 * G::AH_BYTES + (ptid >> 3) * G::ROWB + (ptid & 7) * 16;
    slot = ptid & 7;
    row = ptid >> 3;            // 0..63
    Hl = d.Hout;
    Wl = d.Wout;
    hvalid = hvalid_nxt = 0;
    ld_ca = ld_cb = nullptr;
    drain_on = false;
    w_tapb = (size_t)d.kchunks * d.CoutPad * 64;
    dr_rowb = (size_t)d.Wout * d.Cout * 2;
    dr_base = nullptr;
    wptr = nullptr;
    dr_r0 = ptid >> 3;
    stage_rd = smem + 2 * G::AH_BYTES + 3 * G::BW_BYTES + G::RED_BYTES + dr_r0 * 128 + (((ptid & 7) ^ ((dr_r0 >> 1) & 7)) << 4);
    sA.chunk = 0;
    sA.cur = tm.c0;
    tm.fill(sA.cur, sA.b, sA.y0, sA.x0, sA.tn);
    sB = sA; gC = 0;
    advance(sB);
    sC = sB;
    advance(sC);
    dr = sA;
    w_voff = (unsigned)((((slot >> 2) * d.CoutPad + row) * 32 + (slot & 3) * 8) * 2);
    cf_voff = (unsigned)(slot * 8 * 4);
    dr_voff = (unsigned)((((dr_r0 / TW) * d.Wout + dr_r0 % TW) * d.Cout + slot * 8) * 2);
    ld_dummy = (unsigned)((d.Win + 1));
    // tile-independent part of every halo unit's address and validity, computed once
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int hp = k * RPP + row;
      const int hy = hp / HP, hx = hp - hy * HP;
      int ry = hy - 1, rx = hx - 1;                    // tile origins are even, so the x2 gather is (origin/2) + (r >> 1)
      if (d.ups) { ry >>= 1; rx >>= 1; }
      hpix[k] = (unsigned)((ry + 1) * d.Win + (rx + 1));
      hedge[k] = (hy == 0 ? 1u : 0u) | (hy == TH + 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == TW + 1 ? 8u : 0u) |
                 (hp >= HALO ? 16u : 0u);
    }
  }

  // ---- weights ----
  // next step after `si` (whose index is gC); past the last step it stays there: harmless reloads keep the load
  // stream periodic
  __device__ __forceinline__ void advance(StepInfo& si) {
    if (gC + 1 < nsteps) {
      ++gC;
      if (++si.chunk == nchunks) {
        si.chunk = 0;
        tm.next(si.cur);
        tm.fill(si.cur, si.b, si.y0, si.x0, si.tn);
      }
    }
  }
  __device__ __forceinline__ const char* w_tile(int tap, int chunk, int tn) const {
    return reinterpret_cast<const char*>(L.w) +
           ((PRG_WS_EXP & 1024) ? 0   // timing experiment: always the same (L1-resident) weight tile
                                : ((size_t)(tap * d.kchunks + 2 * chunk) * d.CoutPad + tn * BN) * 64);
  }
  template <int SET>
  __device__ __forceinline__ void w_issue(const char* base) {
    // rows row + 64 j are 64 * 32 * 2 = 4096 bytes apart (immediate offsets reach 4095: second base)
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wset[SET][0]) : "v"(w_voff), "s"(base) : "memory");
    if constexpr (NWL == 2) {
      const char* base2 = base + 4096;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wset[SET][NWL - 1]) : "v"(w_voff), "s"(base2) : "memory");
    }
  }
  template <int SET, int N>                                // N younger loads may stay in flight
  __device__ __forceinline__ void w_wait() {
    if constexpr (NWL == 2)
      asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(wset[SET][0]), "+v"(wset[SET][1]) : [n] "i"(kExpNoWaitW ? 63 : N) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(wset[SET][0]) : [n] "i"(kExpNoWaitW ? 63 : N) : "memory");
    static_assert(NWL == 2 || NWL == 1, "weight units per thread");
  }
  template <int SET>
  __device__ __forceinline__ void w_write(int ring) {
    if constexpr ((PRG_WS_EXP >> 16) != 0) {                       // timing experiment: stagger the weight writes
      if (row & 32) __builtin_amdgcn_s_sleep(PRG_WS_EXP >> 16);    // odd producer waves start N * 64 clk later
    }
#pragma unroll
    for (int j = 0; j < NWL; ++j) {    // weight row `row + 64 j`
      // 16384: timing experiment, only half of the weight tile is written to LDS (what a 256-pixel tile would stage per MFMA)
      const bool skip = (PRG_WS_EXP & 16384) && (NWL == 2 ? j == 1 : (row & 32) != 0);
      if (!kExpNoLdsWrite && !skip) *reinterpret_cast<u32x4*>(Bw0 + ring * G::BW_BYTES + j * 64 * G::ROWB) = wset[SET][j];
      if constexpr ((PRG_WS_EXP & 32768) != 0) {                 // timing experiment: gap between a wave's two stores
        if (j + 1 < NWL) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      }
    }
  }

  // ---- halo ----
  __device__ __forceinline__ void issue_setup(const StepInfo& si) {
    const int b = si.b, y0 = si.y0, x0 = si.x0, chunk = si.chunk;
    const int c = chunk * kCH;
    const bool first = c < d.C0;
    const bf16_t* src = first ? L.src0 : L.src1;
    const int cs = first ? d.C0 : d.C1;
    const int cc = first ? c : c - d.C0;
    ld_cs2 = (unsigned)(cs * 2);
    ld_tedge = (y0 == 0 ? 1u : 0u) | (y0 + TH == Hl ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + TW == Wl ? 8u : 0u) | 16u;
    // halo origin = source pixel of halo position (0,0); may lie one row/column outside the image (never dereferenced:
    // padding taps load the tile origin instead and are zeroed when written)
    const int64_t horg = ((int64_t)b * d.Hin + (y0 >> d.ups)) * d.Win + (x0 >> d.ups) - (d.Win + 1);
    ld_base = reinterpret_cast<const char*>(src) + (horg * cs + cc) * 2;
    if constexpr (PRO) {
      const size_t o = ((size_t)b * d.C0 + chunk * kCH) * 4;
      ld_ca = reinterpret_cast<const char*>(L.pro_a) + o;
      ld_cb = reinterpret_cast<const char*>(L.pro_b) + o;
    }
    hvalid_nxt = 0;
  }
  template <int CS>
  __device__ __forceinline__ void issue_coeffs() {
    if constexpr (PRO) {
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(cf[CS][0]) : "v"(cf_voff), "s"(ld_ca) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(cf[CS][1]) : "v"(cf_voff), "s"(ld_ca) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(cf[CS][2]) : "v"(cf_voff), "s"(ld_cb) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(cf[CS][3]) : "v"(cf_voff), "s"(ld_cb) : "memory");
    }
  }
  template <int K>
  __device__ __forceinline__ void issue_unit() {
    const bool ok = (hedge[K] & ld_tedge) == 0;
    // padding taps read the tile origin (always mapped) and are zeroed at write time
    const unsigned pix = ok && !(PRG_WS_EXP & 2048) ? hpix[K] : ld_dummy;
    const unsigned voff = __umul24(pix, ld_cs2) + (unsigned)(slot * 16);
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(hreg[K]) : "v"(voff), "s"(ld_base) : "memory");
    hvalid_nxt |= (ok ? 1u : 0u) << K;
  }
  template <int K, int CS, int N>
  __device__ __forceinline__ void wait_unit() {
    if constexpr (PRO) {
      asm volatile("s_waitcnt vmcnt(%[n])"
                   : "+v"(hreg[K]), "+v"(cf[CS][0]), "+v"(cf[CS][1]), "+v"(cf[CS][2]), "+v"(cf[CS][3])
                   : [n] "i"(kExpNoWaitU && N != 0 ? 63 : N)
                   : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(hreg[K]) : [n] "i"(kExpNoWaitU && N != 0 ? 63 : N) : "memory");
    }
  }
  template <int K, int CS>
  __device__ __forceinline__ void write_unit(int g_tgt, bool wr) {
    const int hp = K * RPP + row;
    if (hp < HALO && wr) {
      u32x4 v = hreg[K];
      if constexpr (PRO && !kExpNoPro) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // v[j] holds channels 2j (low half) and 2j+1 of this thread's 8: coefficients a[2j], a[2j+1] live in
          // cf[.][j / 2] elements (2j % 4, +1), b[] two vectors further
          const float a_lo = __uint_as_float(cf[CS][j >> 1][(2 * j) & 3]);
          const float a_hi = __uint_as_float(cf[CS][j >> 1][((2 * j) & 3) + 1]);
          const float b_lo = __uint_as_float(cf[CS][2 + (j >> 1)][(2 * j) & 3]);
          const float b_hi = __uint_as_float(cf[CS][2 + (j >> 1)][((2 * j) & 3) + 1]);
          const float lo = fast_silu(fmaf(bf_lo(v[j]), a_lo, b_lo));
          const float hi2 = fast_silu(fmaf(bf_hi(v[j]), a_hi, b_hi));
          v[j] = pack_bf16(lo, hi2);
        }
      }
      if (!((hvalid >> K) & 1u)) v = u32x4{0u, 0u, 0u, 0u};
      if (!kExpNoLdsWrite) *reinterpret_cast<u32x4*>(Ah0 + (g_tgt & 1) * G::AH_BYTES + K * RPP * G::ROWB) = v;
      else asm volatile("" ::"v"(v));
    }
  }
  // Output drain: the consumers leave a finished tile as bf16 [wave][64 pixels][64 channels] in the LDS stage; during
  // the next step's phases 0..7 every producer thread moves one 16-byte unit per phase to HBM (full 128-byte rows per
  // 8 lanes), so the tile's write burst overlaps the next tile's MFMAs instead of stalling the consumers.
  template <int PH>
  __device__ __forceinline__ u32x4 drain_read() {
    return *reinterpret_cast<const u32x4*>(stage_rd + PH * 8192);   // phase PH drains consumer wave PH's 64 x 64 tile
  }
  template <int PH>
  __device__ __forceinline__ void drain_store(const u32x4& v) {
    constexpr int wm = PH / G::WAVES_N, wn = PH % G::WAVES_N;
    // pixels wm*64 + [0,64) of the tile start on a tile row: that part of the address is wave-uniform
    constexpr int prow = (wm * 64) / TW;
    char* base = dr_base + prow * dr_rowb + wn * 128;
    asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(dr_voff), "v"(v), "s"(base) : "memory");
  }
  template <int PH>
  __device__ __forceinline__ void drain_unit() {
    drain_store<PH>(drain_read<PH>());
  }
  // phase PH handles units K = PH * UPH + J.  CSW = coefficient set of the halo being written
  // A phase's units are waited for together, transformed and written together (so the compiler can interleave
  // their arithmetic) and re-issued together.  Unit J of the group is waited for before J older re-issues happen,
  // hence J fewer younger loads.
  template <int PH, int J, int CSW>
  __device__ __forceinline__ void units_wait() {
    if constexpr (J < UPH && PH * UPH + J < KU) {
      if constexpr (!((PRG_WS_EXP & 8192) && (J & 1)))
        wait_unit<PH * UPH + J, CSW, U_YOUNGER - NWL - J>();   // this phase's weight tile is issued after the LDS work
      units_wait<PH, J + 1, CSW>();
    }
  }
  template <int PH, int J, int CSW>
  __device__ __forceinline__ void units_write(int g, bool wr) {
    if constexpr (J < UPH && PH * UPH + J < KU) {
      if constexpr (!((PRG_WS_EXP & 8192) && (J & 1))) write_unit<PH * UPH + J, CSW>(g + 1, wr);
      units_write<PH, J + 1, CSW>(g, wr);
    }
  }
  template <int PH, int J>
  __device__ __forceinline__ void units_issue() {
    if constexpr (J < UPH && PH * UPH + J < KU) {
      if constexpr (!((PRG_WS_EXP & 8192) && (J & 1))) issue_unit<PH * UPH + J>();
      units_issue<PH, J + 1>();
    }
  }
  // tile ph + 5 lies in this step (PH + 5 < 9) or the next one: its (chunk, tn) are per-step values
  template <int GP, int PH, bool LIVE>
  __device__ __forceinline__ void phase(int g, bool wr, int chunk0, int tn0, int chunk1, int tn1) {
    constexpr int SET = (3 * GP + PH + 2) % WD;  // (9 g + PH + 2) % WD, g & 1 == GP: register set of tile ph + 2
    constexpr int RING = (PH + 2) % 3;           // (9 g + PH + 2) % 3: its LDS ring slot
    // the prologue's issue-only step: from phase WD on the set holds an earlier (placeholder, then real) tile and the
    // WD phases behind it are regular, so the steady-state wait + write applies (the consumers have not started)
    // Order inside a phase: everything that touches LDS first (so its latency runs under the global issues and is
    // gone by the barrier), the drain's stage read first of all; then the global issues in the periodic order
    // [weight tile, halo units]; the drain store last.
    u32x4 dv;
    bool draining = false;
    if constexpr (LIVE && PH < DRAIN_PHASES) {
      draining = drain_on;
      if (draining) dv = drain_read<PH>();
    }
    if constexpr (LIVE || PH >= WD) {
      w_wait<SET, w_younger(PH)>();
      if constexpr (LIVE) trace.mark(0);
      w_write<SET>(RING);
    }
    if constexpr (LIVE && PH < 8) {
      units_wait<PH, 0, (GP + 1) & 1>();
      units_write<PH, 0, (GP + 1) & 1>(g, wr);
    }
    // tile ph + 2 + WD: the remaining taps of this step, then taps of the next one (a new tile address only at tap 0)
    constexpr int TAPN = PH + 2 + WD;
    static_assert(TAPN < 18, "weight prefetch reaches at most into the next step");
    if constexpr (!LIVE) wptr = TAPN < 9 ? w_tile(TAPN, chunk0, tn0) : w_tile(TAPN - 9, chunk1, tn1);
    else if constexpr (TAPN == 9) wptr = w_tile(0, chunk1, tn1);
    else if (!(PRG_WS_EXP & 1024)) wptr += w_tapb;
    w_issue<SET>(wptr);
    if constexpr (PH < 8) units_issue<PH, 0>();
    if constexpr (LIVE && PH < DRAIN_PHASES) {
      if (draining) drain_store<PH>(dv);
    }
    if constexpr (LIVE) trace.mark(1);
    if constexpr (LIVE) phase_barrier(trace);
  }
  // GP = g & 1.  Writes halo g+1 (coefficient set (g+1)&1) and weight tiles 9g+2..9g+10; issues halo g+2 (set g&1) and
  // weight tiles 9g+5..9g+13.  LIVE = false is the issue-only "step -1" of the prologue: it puts exactly the loads a
  // real step would leave in flight into the queue, in the same order, so step 0 can use the steady-state wait counts.
  template <int GP, bool LIVE>
  __device__ __forceinline__ void step(int g) {
    const bool wr = g + 1 < nsteps;
    if constexpr (LIVE) {
      const int c0 = sA.chunk, t0 = sA.tn, c1 = sB.chunk, t1 = sB.tn;
      drain_on = g > 0 && sA.chunk == 0;          // the previous step finished tile `dr`
      issue_coeffs<GP>();                         // halo g+2: its context was set up during the previous phase 8
      phase<GP, 0, LIVE>(g, wr, c0, t0, c1, t1); phase<GP, 1, LIVE>(g, wr, c0, t0, c1, t1);
      phase<GP, 2, LIVE>(g, wr, c0, t0, c1, t1); phase<GP, 3, LIVE>(g, wr, c0, t0, c1, t1);
      phase<GP, 4, LIVE>(g, wr, c0, t0, c1, t1); phase<GP, 5, LIVE>(g, wr, c0, t0, c1, t1);
      phase<GP, 6, LIVE>(g, wr, c0, t0, c1, t1); phase<GP, 7, LIVE>(g, wr, c0, t0, c1, t1);
      // Phase 8 has no halo units and no drain: it carries the step bookkeeping (all wave-uniform scalar work) so that
      // phase 0 is not the one everybody waits for.
      hvalid = hvalid_nxt;
      dr = sA;
      dr_base = reinterpret_cast<char*>(L.out) +
                ((((int64_t)dr.b * Hl + dr.y0) * Wl + dr.x0) * d.Cout + dr.tn * BN) * 2;
      sA = sB;
      sB = sC;
      advance(sC);
      issue_setup(sC);
      phase<GP, 8, LIVE>(g, wr, c0, t0, c1, t1);
    } else {
      // step -1: weights of step 0 (placeholders where the tile index is negative), halo of step 1
      const int c0 = sA.chunk, t0 = sA.tn;
      issue_setup(sB);
      issue_coeffs<GP>();
      phase<GP, 0, LIVE>(g, wr, c0, t0, c0, t0); phase<GP, 1, LIVE>(g, wr, c0, t0, c0, t0);
      phase<GP, 2, LIVE>(g, wr, c0, t0, c0, t0); phase<GP, 3, LIVE>(g, wr, c0, t0, c0, t0);
      phase<GP, 4, LIVE>(g, wr, c0, t0, c0, t0); phase<GP, 5, LIVE>(g, wr, c0, t0, c0, t0);
      phase<GP, 6, LIVE>(g, wr, c0, t0, c0, t0); phase<GP, 7, LIVE>(g, wr, c0, t0, c0, t0);
      phase<GP, 8, LIVE>(g, wr, c0, t0, c0, t0);
      hvalid = hvalid_nxt;
      issue_setup(sC);                            // context of halo 2, issued during step 0
    }
  }
  template <int K>
  __device__ __forceinline__ void prologue_issue() {
    if constexpr (K < KU) {
      issue_unit<K>();
      prologue_issue<K + 1>();
    }
  }
  template <int K>
  __device__ __forceinline__ void prologue_write() {
    if constexpr (K < KU) {
      wait_unit<K, 0, 0>();
      write_unit<K, 0>(0, true);
      prologue_write<K + 1>();
    }
  }
  // Every asm load's destination must stay owned by its variable until a wait has proven the load complete: a dead
  // destination could be handed to another value while the load is still in flight.  So no load is ever left unused:
  // the prologue's placeholder weight loads are waited for and written like real ones, and finish() keeps the clamped
  // tail loads alive past the final vmcnt(0).
  __device__ __forceinline__ void prologue() {
    issue_setup(sA);
    issue_coeffs<0>();
    prologue_issue<0>();
    hvalid = hvalid_nxt;
    prologue_write<0>();      // halo 0 (vmcnt(0) waits)
    // issue-only step "-1": afterwards weight tiles 0, 1 are in ring slots 0, 1 and the queue holds, in steady-state
    // order, the coefficients + units of halo 1 and weight tiles 2, 3, 4
    step<1, false>(-1);
  }
  template <int K>
  __device__ __forceinline__ void keep_units() {
    if constexpr (K < KU) {
      asm volatile("" : "+v"(hreg[K])::"memory");
      keep_units<K + 1>();
    }
  }
  __device__ __forceinline__ void finish(bool drain = true) {
    // the last step always ends a tile: `dr` is that tile (consumers passed the last phase barrier: stage complete)
    if (drain) {
      drain_unit<0>(); drain_unit<1>(); drain_unit<2>(); drain_unit<3>();
      static_assert(DRAIN_PHASES == 4, "one consumer wave's stage per drain phase");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int sidx = 0; sidx < WD; ++sidx)
#pragma unroll
      for (int j = 0; j < NWL; ++j) asm volatile("" : "+v"(wset[sidx][j])::"memory");
    keep_units<0>();
    if constexpr (PRO) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(cf[c][j])::"memory");
    }
    phase_barrier(trace);
  }
};

// =====================================================================================================
// kernel
// =====================================================================================================
template <int TH, int TW, int BN, bool PRO>
__global__ __launch_bounds__(768) void conv3x3_ws_kernel(const ConvLaunch<bf16_t> L, const int tiles_x,
                                                            const int tiles_y, const int tiles_n, const int fuse_stats,
                                                            unsigned long long* const trace_buf) {
  using G = WsGeom<TH, TW, BN>;
  constexpr int HP = G::HP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWB = G::ROWB;
  char* const stage = smem + 2 * G::AH_BYTES + 3 * G::BW_BYTES + G::RED_BYTES;
  float* const bias_lds = reinterpret_cast<float*>(stage + G::STG_BYTES);
  TraceCtx trace(trace_buf);

  const ConvDesc& d = L.d;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nchunks = (d.C0 + d.C1) / kCH;
  TileMap tmap;
  tmap.init((int)blockIdx.x, (int)gridDim.x, tiles_x, tiles_y, tiles_n, d.B, TH, TW);
  const int my_tiles = tmap.count;
  const int nsteps = my_tiles * nchunks;
  if (nsteps == 0) return;

  // ---------------------------------------------------------------------------------------------------
  if (wave < 4) {
    __builtin_amdgcn_s_setprio((PRG_WS_EXP & 64) ? 0 : ((PRG_WS_EXP & 128) ? 1 : 3));   // the MFMA waves share each SIMD's issue port with one producer wave
    // 8-channel chunks per GroupNorm group (a power of two <= 8 when the statistics are fused)
    const int gn_per = fuse_stats ? (d.Cout / L.gn_groups) >> 3 : 1;
    const int gn_per_sh = 31 - __builtin_clz(gn_per);
    const int wm = wave / G::WAVES_N, wn = wave % G::WAVES_N;
    const int l31 = lane & 31, hi = lane >> 5;
    // Pixel of the wave's 32-pixel group that lane l31 owns.  With 16-pixel tile rows a group spans two halo rows
    // (18 LDS rows apart): rotating the second row's columns by two keeps the 16 lanes of every ds_read_b128 lane
    // group ({0-3,12-15,20-27}, {4-11,16-19,28-31}) on distinct residues mod 16 of the LDS row index, i.e. bank
    // conflict free (SQ_LDS_BANK_CONFLICT was 57 % of the LDS cycles with the identity mapping).
    const int lpx = (TW == 16 && l31 >= 16) ? 16 + ((l31 - 2) & 15) : l31;
    // per-lane LDS byte addresses of the fragments of call 0 / tap 0: pixel rows of the wave's two 32-pixel groups and
    // weight rows of its two 32-channel groups; lanes 32-63 take the second 16-byte unit of each 32-byte k-slice.
    // Everything else (tap, call, ring slot) is an immediate offset.
    const char* xrow[2];
    const char* wrowp[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int p = wm * 64 + pt * 32 + lpx;
      xrow[pt] = smem + ((p / TW) * HP + (p % TW)) * ROWB + hi * 16;
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) wrowp[ct] = smem + 2 * G::AH_BYTES + (wn * 64 + ct * 32 + l31) * ROWB + hi * 16;
    f32x16 acc[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ct][pt][e] = 0.0f;

    // Consumers never wait for their own LDS reads at a barrier: every fragment of phase p is consumed by an MFMA of
    // phase p (which cannot issue before the data is back), and the fragments prefetched for phase p+1 come from a
    // ring slot / halo buffer that no producer touches before phase p+2.
    // the workgroup's channel tile never changes (one Cout tile, or pinned to one): its bias lives in LDS
    if (tid < BN) bias_lds[tid] = L.bias[(tmap.pinned ? tmap.tn_fixed : 0) * BN + tid];
    phase_barrier<true>(trace);    // prologue barrier: first halo + weight tiles 0,1 are in LDS
    // three fragment sets: the one being multiplied, the next call's, and the one being loaded two calls (256 MFMA
    // cycles) ahead; set of global call c is c % 3 (36 calls per step: the same code serves every step)
    bf16x8 fw[3][2], fx[3][2];
    const char* xa[2] = {xrow[0], xrow[1]};                                 // halo buffer of this step
    const char* xn[2] = {xrow[0] + G::AH_BYTES, xrow[1] + G::AH_BYTES};     // ... of the next one
#define PRG_LW(SET, CT, RING, CALL) fw[SET][CT] = *reinterpret_cast<const bf16x8*>(wrowp[CT] + (RING) * (int)G::BW_BYTES + (CALL) * 32)
#define PRG_LX(SET, PT, BASE, TOFF, CALL) fx[SET][PT] = *reinterpret_cast<const bf16x8*>(BASE[PT] + (TOFF) * ROWB + (CALL) * 32)
#define PRG_MM(SET, CT, PT) acc[CT][PT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[SET][CT], fx[SET][PT], acc[CT][PT], 0, 0, 0)
#define PRG_SB() __builtin_amdgcn_sched_barrier(0)
    PRG_LW(0, 0, 0, 0); PRG_LX(0, 0, xa, 0, 0); PRG_LX(0, 1, xa, 0, 0); PRG_LW(0, 1, 0, 0);
    PRG_LW(1, 0, 0, 1); PRG_LX(1, 0, xa, 0, 1); PRG_LX(1, 1, xa, 0, 1); PRG_LW(1, 1, 0, 1);
    int chunk = 0, it = 0;                 // step g = it * nchunks + chunk
    int tb = 0, ty0 = 0, tx0 = 0, ttn = 0; // tile `it`
    TileCur tcur = tmap.c0;
    tmap.fill(tcur, tb, ty0, tx0, ttn);
    for (int g = 0; g < nsteps; ++g) {
      const bool tile_end = chunk == nchunks - 1;
      // The nine taps are fully unrolled: tap offsets, the weight ring slot (9 g + p) % 3 == p % 3 and the fragment
      // set indices are compile-time constants and all LDS offsets are immediates.  The sched_barriers pin the
      // interleave (one ds_read for the call two ahead, one MFMA) that the scheduler otherwise collapses into
      // load bursts followed by waits.
#pragma unroll
      for (int p = 0; p < 9; ++p) {
        const int toff = (p / 3) * HP + (p % 3);
        const int pn = p == 8 ? 0 : p + 1;
        const int toffN = (pn / 3) * HP + (pn % 3);
#pragma unroll
        for (int call = 0; call < ((PRG_WS_EXP & 512) ? 0 : 4); ++call) {
          // set (call + 2) % 4 was consumed two calls ago: refill it for the call two ahead (this phase's calls 2,3
          // or the NEXT phase's calls 0,1 — its weight tile and halo are already visible in LDS)
          const int sl = (4 * p + call + 2) % 3, sm = (4 * p + call) % 3;   // set being loaded / multiplied
          if (call < 2) {
            PRG_LW(sl, 0, p % 3, call + 2); PRG_MM(sm, 0, 0); PRG_SB();
            PRG_LX(sl, 0, xa, toff, call + 2); PRG_MM(sm, 0, 1); PRG_SB();
            PRG_LX(sl, 1, xa, toff, call + 2); PRG_MM(sm, 1, 0); PRG_SB();
            PRG_LW(sl, 1, p % 3, call + 2); PRG_MM(sm, 1, 1); PRG_SB();
          } else if (p == 8) {
            PRG_LW(sl, 0, (p + 1) % 3, call - 2); PRG_MM(sm, 0, 0); PRG_SB();
            PRG_LX(sl, 0, xn, toffN, call - 2); PRG_MM(sm, 0, 1); PRG_SB();
            PRG_LX(sl, 1, xn, toffN, call - 2); PRG_MM(sm, 1, 0); PRG_SB();
            PRG_LW(sl, 1, (p + 1) % 3, call - 2); PRG_MM(sm, 1, 1); PRG_SB();
          } else {
            PRG_LW(sl, 0, (p + 1) % 3, call - 2); PRG_MM(sm, 0, 0); PRG_SB();
            PRG_LX(sl, 0, xa, toffN, call - 2); PRG_MM(sm, 0, 1); PRG_SB();
            PRG_LX(sl, 1, xa, toffN, call - 2); PRG_MM(sm, 1, 0); PRG_SB();
            PRG_LW(sl, 1, (p + 1) % 3, call - 2); PRG_MM(sm, 1, 1); PRG_SB();
          }
        }
        if (p == 8 && tile_end && !kExpNoEpilogue) {
          // tile finished.  Lane holds pixel (pt*32 + l31), channels ct*32 + 8q + 4hi + {0..3}: bias, round, transpose
          // through the wave's LDS stage, 16-byte stores; the 8 channels of chunk (ct, q) are shared by the whole wave.
          char* const stg = stage + wave * (64 * 128);
          float V[16];   // [sum | sum of squares][ct][q]
          const bool o16 = L.out_f16 != 0;
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + wn * 64 + ct * 32 + 8 * q + 4 * hi);
              const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
              float s = 0.0f, sq = 0.0f;
#pragma unroll
              for (int pt = 0; pt < 2; ++pt) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  v[r] = acc[ct][pt][4 * q + r] + bv[r];
                  s += v[r];
                  sq = fmaf(v[r], v[r], sq);
                }
                uint2 w;                                   // (o16: the h16 output format of conv.h, wave-uniform)
                w.x = o16 ? h16_pack(v[0], v[1]) : pack_bf16(v[0], v[1]);
                w.y = o16 ? h16_pack(v[2], v[3]) : pack_bf16(v[2], v[3]);
                // row = pixel, 16-byte unit = ct*4 + q (XOR-swizzled), half = hi
                const int px = pt * 32 + lpx;
                *reinterpret_cast<uint2*>(stg + px * 128 + (((ct * 4 + q) ^ ((px >> 1) & 7)) << 4) + hi * 8) = w;
              }
              V[ct * 4 + q] = s;
              V[8 + ct * 4 + q] = sq;
            }
          // the producer waves move the stage to HBM during the next step (Producer::drain_unit)
          if (fuse_stats && !kExpNoStats) {
            // 16 full-wave sums with 17 lane exchanges: each butterfly round halves the values a lane carries (it keeps
            // the half selected by its lane bit and sends the other half to its partner).  The rounds that move many
            // values use DPP quad permutes (VALU, no LDS crossbar); fixed order -> deterministic.
            const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
            float A8[8], B4[4], C2[2];
#pragma unroll
            for (int j = 0; j < 8; ++j) A8[j] = (b0 ? V[8 + j] : V[j]) + dpp_f32<0xB1>(b0 ? V[j] : V[8 + j]);          // lane ^ 1
#pragma unroll
            for (int j = 0; j < 4; ++j) B4[j] = (b1 ? A8[4 + j] : A8[j]) + dpp_f32<0x4E>(b1 ? A8[j] : A8[4 + j]);    // lane ^ 2
#pragma unroll
            for (int j = 0; j < 2; ++j) C2[j] = (b2 ? B4[2 + j] : B4[j]) + swz_xor<4>(b2 ? B4[j] : B4[2 + j]);
            float D = (b3 ? C2[1] : C2[0]) + swz_xor<8>(b3 ? C2[0] : C2[1]);
            D += swz_xor<16>(D);
            D += __shfl_xor(D, 32, 64);
            // lane (< 16) now holds the wave total of value i = 8 b0 + 4 b1 + 2 b2 + b3 = [sq][ct][q]: the 8-channel chunk
            // ct*4 + q of this wave's 64 channels.  Fold the chunks of one GroupNorm group (per = cpg / 8 <= 8 of them,
            // neighbours in q, then ct) and let the group's first chunk store: one partial per (wave row, group).
            const int per = gn_per;
            if (per >= 2) D += swz_xor<8>(D);
            if (per >= 4) D += swz_xor<4>(D);
            if (per >= 8) D += dpp_f32<0x4E>(D);
            const int i = (lane & 1) * 8 + (lane & 2) * 2 + ((lane >> 2) & 1) * 2 + ((lane >> 3) & 1);
            const int cc = i & 7;
            if (lane < 16 && (cc & (per - 1)) == 0) {
              const int nsplit = tiles_x * tiles_y * G::WAVES_M;
              const int slab = ((ty0 / TH) * tiles_x + tx0 / TW) * G::WAVES_M + wm;
              const int grp = (((ttn * BN + wn * 64) >> 3) + cc) >> gn_per_sh;
              if (L.gn_acc) gn_acc_add(L.gn_acc, L.gn_groups, tb, grp, i >> 3, D);   // fixed-point accumulators (common.h)
              else L.gn_partials[(((size_t)tb * nsplit + slab) * L.gn_groups + grp) * 2 + (i >> 3)] = D;
            }
          }
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
              for (int e = 0; e < 16; ++e) acc[ct][pt][e] = 0.0f;
        }
        if (p == 8 && tile_end) phase_barrier<true>(trace);   // the stage must be visible to the producer waves
        else phase_barrier<false>(trace);
      }
      // next step
      { const char* t0 = xa[0]; xa[0] = xn[0]; xn[0] = t0; const char* t1 = xa[1]; xa[1] = xn[1]; xn[1] = t1; }
      if (++chunk == nchunks) {
        chunk = 0;
        if (++it < my_tiles) {
          tmap.next(tcur);
          tmap.fill(tcur, tb, ty0, tx0, ttn);
        }
      }
    }
#undef PRG_LW
#undef PRG_LX
#undef PRG_MM
#undef PRG_SB
    phase_barrier<false>(trace);   // matches the producers' finish()
    return;
  }

  // ---------------------------------------------------------------------------------------------------
  {
    if (PRG_WS_EXP & (64 | 128)) __builtin_amdgcn_s_setprio((PRG_WS_EXP & 64) ? 3 : 1);
    Producer<TH, TW, BN, PRO> Pv(L, smem, tid - 256, tmap, nsteps, nchunks, trace);
    Pv.prologue();
    phase_barrier(trace);
    if constexpr ((PRG_WS_EXP & 256) != 0) {
      // timing experiment: consumers alone.  The loads of the prologue stay owned until they have landed.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int g = 0; g < nsteps; ++g)
        for (int p = 0; p < 9; ++p) phase_barrier(trace);
      Pv.finish(false);
      return;
    }
#pragma unroll 1
    for (int g = 0; g < nsteps; g += 2) {
      Pv.template step<0, true>(g);
      if (g + 1 < nsteps) Pv.template step<1, true>(g + 1);
    }
    Pv.finish();
  }
}

constexpr int kWsUnsupported = 100;   // launch_ws_cfg: shape not covered, caller falls back

template <int TH, int TW, int BN, bool PRO>
int launch_ws_cfg2(const ConvLaunch<bf16_t>& L, hipStream_t s, int fuse_stats, int* nsplit, int num_cus) {
  using G = WsGeom<TH, TW, BN>;
  const ConvDesc& d = L.d;
  const int tiles_x = d.Wout / TW, tiles_y = d.Hout / TH, tiles_n = d.Cout / BN;
  const int total = tiles_x * tiles_y * tiles_n * d.B;
  int grid = num_cus;
  if (grid > total) grid = total;
  if (grid >= 8) grid &= ~7;                     // multiple of 8: XCD-contiguous runs inside a round
  if (tiles_n > 1 && (grid & 7) != 0) return kWsUnsupported;   // a workgroup must own one channel tile (bias in LDS)
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ws_kernel<TH, TW, BN, PRO>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    if (e != hipSuccess) return fail(PRG_E_HIP, std::string("hipFuncSetAttribute(ws conv): ") + hipGetErrorString(e));
    attr_done.mark();
  }
  if (nsplit) *nsplit = fuse_stats ? tiles_x * tiles_y * G::WAVES_M : 0;
  if (L.probe) return PRG_OK;
  static const int trace_at = [] { const char* e = std::getenv("PRG_WS_TRACE"); return e ? std::atoi(e) : -1; }();
  static int launch_no = 0;
  unsigned long long* tbuf = nullptr;
  if (kTrace && trace_at >= 0 && launch_no++ == trace_at) {
    if (hipMalloc(reinterpret_cast<void**>(&tbuf), 12 * kTraceStride * sizeof(unsigned long long)) == hipSuccess) {
      (void)hipMemsetAsync(tbuf, 0, 12 * kTraceStride * sizeof(unsigned long long), s);
      (void)hipStreamSynchronize(s);
    }
  }
  conv3x3_ws_kernel<TH, TW, BN, PRO><<<dim3(grid), 768, G::LDS, s>>>(L, tiles_x, tiles_y, tiles_n, fuse_stats, tbuf);
  PRG_LAUNCH_CHECK();
  if (tbuf) {
    (void)hipStreamSynchronize(s);
    char path[256];
    std::snprintf(path, sizeof(path), "%s/ws_trace_%d_%d_%d_cin%d_pro%d.bin",
                  std::getenv("PRG_WS_TRACE_DIR") ? std::getenv("PRG_WS_TRACE_DIR") : "/tmp", TH, TW, BN, d.C0 + d.C1,
                  (int)PRO);
    std::vector<unsigned long long> host(12 * kTraceStride);
    (void)hipMemcpy(host.data(), tbuf, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (FILE* f = std::fopen(path, "wb")) {
      std::fwrite(host.data(), sizeof(unsigned long long), host.size(), f);
      std::fclose(f);
    }
    (void)hipFree(tbuf);
  }
  return PRG_OK;
}

template <int TH, int TW, int BN>
int launch_ws_cfg(const ConvLaunch<bf16_t>& L, hipStream_t s, int fuse_stats, int* nsplit, int num_cus) {
  return L.pro_a ? launch_ws_cfg2<TH, TW, BN, true>(L, s, fuse_stats, nsplit, num_cus)
                 : launch_ws_cfg2<TH, TW, BN, false>(L, s, fuse_stats, nsplit, num_cus);
}

}  // namespace

// Returns 1 when it launched, 0 when the shape is not covered (caller falls back), negative on error.
int try_launch_conv3x3_ws(const ConvLaunch<bf16_t>& L, hipStream_t s, int* gn_nsplit_out, int* acc_done) {
  static const int enabled = [] {
    const char* e = std::getenv("PRG_CONV_WS");
    return e ? std::atoi(e) : 1;
  }();
  if (!enabled) return 0;
  const ConvDesc& d = L.d;
  if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1)) return 0;
  if (d.C0 % kCH || d.C1 % kCH || d.Cout % 64) return 0;
  if (L.residual || !L.bias) return 0;
  if (L.in_f16 || (L.out_f16 && L.pro_a)) return 0;          // h16 (conv.h): f16 OUTPUT only, from a launch without prologue
  {
    const int tn128 = d.Cout % 128 == 0 ? d.Cout / 128 : d.Cout / 64;   // Cout tiles of the configuration chosen below
    if (tn128 != 1 && tn128 != 2 && tn128 != 4 && tn128 != 8) return 0;  // every workgroup must own ONE channel tile
  }
  const int num_cus = device_cu_count();
  if (num_cus <= 0) return 0;
  const int H = d.Hout, W = d.Wout;
  const int cpg = L.gn_groups > 0 ? d.Cout / L.gn_groups : 0;
  const bool want = L.gn_partials != nullptr;
  auto fuse_for = [&](int TH, int TW, int BN) {
    // one partial per (tile, consumer wave row): a group must lie inside one wave's 64 channels
    return want && cpg % 8 == 0 && cpg <= 64 && (cpg & (cpg - 1)) == 0 && (W / TW) * (H / TH) * (4 / (BN / 64)) <= kGnMaxSplit ? 1 : 0;
  };
  static const int cfg_mask = [] { const char* e = std::getenv("PRG_WS_CFGS"); return e ? std::atoi(e) : 7; }();   // debugging aid
  int rc = 0;
  // PRG_WS_PREFER64: route wide convs through the 256-pixel x 64-channel tile where it fits (half the weight bytes
  // staged per MFMA, the halo staged once per 64 output channels)
  static const int prefer64 = [] { const char* e = std::getenv("PRG_WS_PREFER64"); return e ? std::atoi(e) : 0; }();
  const int tn64 = d.Cout / 64;
  if (prefer64 && d.Cout % 128 == 0 && W % 32 == 0 && H % 8 == 0 && (tn64 == 2 || tn64 == 4 || tn64 == 8) &&
      (prefer64 >= 2 || tn64 == 2)) {
    rc = launch_ws_cfg<8, 32, 64>(L, s, fuse_for(8, 32, 64), gn_nsplit_out, num_cus);
    if (rc == kWsUnsupported) return 0;
    if (rc == PRG_OK && acc_done) *acc_done = (L.gn_acc && gn_nsplit_out && *gn_nsplit_out > 0) ? 1 : 0;
    return rc == PRG_OK ? 1 : rc;
  }
  if (d.Cout % 128 == 0) {
    if (!(cfg_mask & (W % 32 == 0 && H % 4 == 0 ? 1 : 2))) return 0;
    if (W % 32 == 0 && H % 4 == 0) rc = launch_ws_cfg<4, 32, 128>(L, s, fuse_for(4, 32, 128), gn_nsplit_out, num_cus);
    else if (W % 16 == 0 && H % 8 == 0) rc = launch_ws_cfg<8, 16, 128>(L, s, fuse_for(8, 16, 128), gn_nsplit_out, num_cus);
    else return 0;
  } else {
    if (!(cfg_mask & 4)) return 0;
    if (W % 32 == 0 && H % 8 == 0) rc = launch_ws_cfg<8, 32, 64>(L, s, fuse_for(8, 32, 64), gn_nsplit_out, num_cus);
    else return 0;
  }
  if (rc == kWsUnsupported) return 0;
  if (rc == PRG_OK && acc_done) *acc_done = (L.gn_acc && gn_nsplit_out && *gn_nsplit_out > 0) ? 1 : 0;
  return rc == PRG_OK ? 1 : rc;
}

}  // namespace prg
