// conv_ws.hip — wave-specialised persistent 3x3 convolution for the bf16 throughput path (MFMA roofline).
//
// One 512-thread workgroup per CU, resident for the whole launch, walks a strided list of output tiles
// (TH x TW pixels of one image x BN output channels).  Its eight waves have two roles:
//
//   waves 0-3  CONSUMERS  one per SIMD.  Nothing but LDS fragment reads and v_mfma_f32_32x32x16_bf16: per phase
//              (= one tap of one 128-byte channel chunk) 16 fragment reads + 16 MFMAs.  At the end of a tile they
//              add the bias, round to bf16 and park the 64x64 result in an LDS stage with 8-byte stores (the MFMA
//              operands are swapped — weights are the "A" rows, pixels the "B" columns — so each lane holds four
//              consecutive channels of one pixel).
//   waves 4-7  PRODUCERS  share the SIMDs with the consumers (VALU / memory pipes run beside the matrix pipe).
//              Per phase they (a) write the weight tile of the next tap into the LDS double buffer and issue the
//              loads for the one after, (b) move a slice of the NEXT step's input halo global -> registers -> LDS,
//              applying the fused GroupNorm + (scale+1, shift) + SiLU of the previous Block on the way (sd:690-696)
//              — each halo pixel once, not nine times, (c) drain a slice of the PREVIOUS tile's stage to HBM with
//              16-byte stores and accumulate the GroupNorm partial sums of the output.
//
// One s_barrier per phase is the only synchronisation.  The halo image is double-buffered across steps, so HBM
// latency is hidden behind a full phase of MFMAs instead of being exposed at the head of every tile; the epilogue
// and the elementwise transform never stall the matrix pipe.  LDS images use the same 16-byte XOR swizzle as
// conv.hip (unit ^= (row >> 1) & 7): conflict-free ds_read_b128 fragment loads.
//
// Covers every 3x3 / stride 1 / pad 1 conv of the U-Nets at dim = 64 (widths multiples of 64): Block.proj, the last
// down/up convs, Upsample's conv (x2 nearest gather folded into the halo load), skip concat as two sources.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "conv.h"

namespace prg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int kCH = 64;   // channels per 128-byte pixel row (bf16)

__device__ inline float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ inline float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ inline uint32_t pack_bf16(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
// Phase barrier: LDS traffic of this wave retired, then s_barrier.  Deliberately NOT __syncthreads(): that also
// waits vmcnt(0), which would drain the producers' in-flight global prefetches at every phase and expose the full
// memory latency nine times per step.  Global loads stay in flight across it; hipcc still inserts the counted
// vmcnt before the first use of each loaded register.
// Optional barrier trace (PRG_WS_TRACE=<launch index>): workgroup 0 records, per wave, the shader clock when it
// arrives at and when it leaves every phase barrier of that launch.  Shows which role the others wait for.
constexpr int kTraceStride = 4096;   // u64 slots per wave: [0] = count, then (arrive, leave) pairs

__device__ __forceinline__ void phase_barrier(unsigned long long* tr) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const bool rec = tr != nullptr && blockIdx.x == 0 && (threadIdx.x & 63) == 0;
  unsigned long long n = 0;
  if (rec) {
    tr += (threadIdx.x >> 6) * kTraceStride;
    n = tr[0];
    if (n < (kTraceStride - 2) / 2) tr[1 + 2 * n] = clock64();
  }
  __builtin_amdgcn_s_barrier();
  if (rec && n < (kTraceStride - 2) / 2) {
    tr[2 + 2 * n] = clock64();
    tr[0] = n + 1;
  }
  asm volatile("" ::: "memory");
}

__device__ inline float fast_silu(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

template <int TH, int TW, int BN>
struct WsGeom {
  static constexpr int BM = TH * TW;
  static constexpr int HP = TW + 2;
  static constexpr int HALO = (TH + 2) * HP;
  static constexpr int NU = HALO * 8;                      // 16-byte units of one halo image
  static constexpr int HSLICES = (NU + 511) / 512;         // 512 units (2 per producer thread) per phase
  static constexpr int NBU = BN * 8 / 256;                 // weight units per producer thread per tap
  static constexpr int WAVES_N = BN / 64, WAVES_M = 4 / WAVES_N;
  static constexpr int ST_UNITS = BM * BN / 8;             // 16-byte units of the stage (2048 for all configs)
  static constexpr size_t AH_BYTES = (size_t)HALO * 128;
  static constexpr size_t BW_BYTES = (size_t)BN * 128;
  static constexpr size_t ST_BYTES = (size_t)BM * BN * 2;
  static constexpr size_t LDS = 2 * AH_BYTES + 3 * BW_BYTES + ST_BYTES + 4 * 16 * 2 * sizeof(float);
  static_assert(HSLICES <= 8, "halo must be staged within eight phases");
  static_assert(BM / WAVES_M == 64, "consumer wave tile is 64 pixels x 64 channels");
  static_assert(ST_UNITS == 2048, "stage drains 256 units per phase over eight phases");
};

// ---------------------------------------------------------------------------------------------------
// Producer side (waves 4-7).  ONE kind of memory stream per wave, so the in-order vmcnt of a wave never makes a
// cheap operation wait behind an expensive one, and each stream can run several phases ahead of its use:
//
//   wave 4      WEIGHTS + DRAIN.  Three register sets hold the weight tiles of phases ph+1..ph+3 (9 % 3 == 0, so
//               the set of a phase is a compile-time constant of the unrolled step body); phase ph writes tile
//               ph+1 into LDS buffer (ph+1)&1 and re-issues that set for tile ph+4: three phases between an L2
//               load and its use.  It also drains the previous tile's stage: 4 x 16-byte units per lane per phase
//               (8 phases), accumulating the GroupNorm partial sums, folded with shuffles in phase 8.
//   waves 5-7   HALO.  All units of step g+1's halo are issued in phase 0 of step g and written (after the fused
//               GroupNorm/SiLU transform) in the last phases of the step: >= 3 phases of MFMAs between an HBM
//               load and its use.
// Everything indexed by phase is a compile-time constant (no dynamically indexed register arrays).
// ---------------------------------------------------------------------------------------------------
// Tile schedule of one workgroup.  Round r covers a contiguous run of tiles per XCD (workgroup b runs on XCD b % 8:
// observed placement, used for speed only).  When the conv has several output-channel tiles, every XCD is pinned to
// ONE of them (tn = xcd % tiles_n) and walks pixel tiles only: its 32 CUs then stream the same weight slice at
// the same time, so the slice (1-2 MB) stays resident in the XCD's 4 MB L2 instead of the whole 5 MB matrix
// thrashing it (weight fetch latency falls from HBM/MALL to L2 class, which is what the 3-phase prefetch covers).
struct TileMap {
  int tiles_x, tiles_y, tiles_n, TH, TW;
  int pinned;       // 1: tn fixed per XCD
  int tn_fixed;
  int first, stride, count;   // tile (or pixel-tile) index of iteration it = first + it * stride, it < count

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

template <int TH, int TW, int BN>
struct ProdBase {
  using G = WsGeom<TH, TW, BN>;
  const ConvLaunch<bf16_t>& L;
  const ConvDesc& d;
  uint4* Ah0;
  uint4* Bw0;
  char* St;
  const TileMap& tm;
  int lane, tiles_x, tiles_y, nsteps, nchunks, fuse_stats, dbg;
  unsigned long long* trace = nullptr;

  __device__ __forceinline__ ProdBase(const ConvLaunch<bf16_t>& L_, char* smem, int lane_, const TileMap& tm_,
                                      int nsteps_, int nchunks_, int fuse, int dbg_)
      : L(L_), d(L_.d), tm(tm_), lane(lane_), tiles_x(tm_.tiles_x), tiles_y(tm_.tiles_y),
        nsteps(nsteps_), nchunks(nchunks_), fuse_stats(fuse), dbg(dbg_) {
    Ah0 = reinterpret_cast<uint4*>(smem);
    Bw0 = reinterpret_cast<uint4*>(smem + 2 * G::AH_BYTES);
    St = smem + 2 * G::AH_BYTES + 3 * G::BW_BYTES;
  }
  __device__ __forceinline__ void decode(int it, int& b, int& y0, int& x0, int& tn) const { tm.decode(it, b, y0, x0, tn); }
};

// ---- weight wave(s) ----------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int TH, int TW, int BN>
struct WeightWave : ProdBase<TH, TW, BN> {
  using Base = ProdBase<TH, TW, BN>;
  using G = typename Base::G;
  using Base::L; using Base::d; using Base::Bw0; using Base::lane; using Base::nsteps; using Base::nchunks;
  using Base::dbg; using Base::trace;
  static constexpr int NW = 8;              // weight units per lane per tap: each weight wave stages 64 rows
  // The loads are inline asm on purpose: hipcc's own waitcnt bookkeeping falls back to vmcnt(0) at every merge of
  // this unrolled, branchy stream, which drains the two younger tiles on every phase and exposes the full L2
  // latency nine times per step.  Hidden from the compiler, this wave's ONLY vector-memory traffic is 8 loads per
  // phase, so a hand-counted `s_waitcnt vmcnt(16)` (two younger tiles may stay in flight) is exact.
  u32x4 wset[3][NW];
  const bf16_t* wb;
  int slot, wrow, widx;

  __device__ __forceinline__ WeightWave(const ConvLaunch<bf16_t>& L_, char* smem, int lane_, int widx_, const TileMap& tm_,
                                        int nsteps_, int nchunks_, int fuse, int dbg_)
      : Base(L_, smem, lane_, tm_, nsteps_, nchunks_, fuse, dbg_), widx(widx_) {
    slot = lane & 7;
    wrow = widx * 64 + (lane >> 3);   // this wave stages weight rows [64 widx, 64 widx + 64)
    wb = L.w + ((size_t)(slot >> 2) * d.CoutPad + wrow) * 32 + (slot & 3) * 8;
  }

  // weight tile of (tap, chunk, tn): `wb` = L.w + this lane's (row, unit) offset, hoisted out of the phase path
  template <int SET>
  __device__ __forceinline__ void w_issue_at(int tap, int chunk, int tn) {
    const bf16_t* base = wb + ((size_t)(tap * d.kchunks + 2 * chunk) * d.CoutPad + tn * BN) * 32;
    // rows wrow + 8 j are 8 * 32 * 2 = 512 bytes apart: one base address, immediate offsets
#define PRG_WLOAD(J) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(wset[SET][J]) : "v"(base), "i"((J) * 512) : "memory")
    PRG_WLOAD(0); PRG_WLOAD(1); PRG_WLOAD(2); PRG_WLOAD(3); PRG_WLOAD(4); PRG_WLOAD(5); PRG_WLOAD(6); PRG_WLOAD(7);
#undef PRG_WLOAD
  }
  __device__ __forceinline__ int tn_of_step(int step) const {
    if (step >= nsteps) step = nsteps - 1;       // past the end: harmless reload
    int b, y0, x0, tn;
    this->decode(step / nchunks, b, y0, x0, tn);
    return tn;
  }
  template <int SET>
  __device__ __forceinline__ void w_issue(int phx) {   // general form (prologue only)
    int step = phx / 9;
    const int tap = phx - step * 9;
    if (step >= nsteps) step = nsteps - 1;
    w_issue_at<SET>(tap, step % nchunks, tn_of_step(step));
  }
  // N = loads that may remain outstanding (younger tiles): 16 in steady state, 0 in the prologue
  template <int SET, int N>
  __device__ __forceinline__ void w_wait() {
    asm volatile("s_waitcnt vmcnt(%[n])"
                 : "+v"(wset[SET][0]), "+v"(wset[SET][1]), "+v"(wset[SET][2]), "+v"(wset[SET][3]), "+v"(wset[SET][4]),
                   "+v"(wset[SET][5]), "+v"(wset[SET][6]), "+v"(wset[SET][7])
                 : [n] "i"(N)
                 : "memory");
  }
  template <int SET>
  __device__ __forceinline__ void w_write(int phx) {
    u32x4* Bw = reinterpret_cast<u32x4*>(Bw0 + (phx % 3) * BN * 8);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int n = wrow + j * 8;
      Bw[n * 8 + (slot ^ ((n >> 1) & 7))] = wset[SET][j];
    }
  }

  __device__ __forceinline__ void prologue() {
    w_issue<0>(0);
    w_issue<1>(1);
    w_wait<0, 0>();
    w_wait<1, 0>();
    w_write<0>(0);          // tile 0 -> ring slot 0
    w_write<1>(1);          // tile 1 -> ring slot 1 (consumers prefetch one phase ahead)
    w_issue<2>(2);          // invariant before phase ph: set (ph+k) % 3 holds tile ph+k, k = 2..4
    w_issue<0>(3);
    w_issue<1>(4);
  }
  // tile ph + 5 lies in this step (PH + 5 < 9) or the next one: its (chunk, tn) are per-step values
  template <int PH>
  __device__ __forceinline__ void phase(int g, int chunk0, int tn0, int chunk1, int tn1) {
    constexpr int SET = (PH + 2) % 3;            // (9 g + PH + 2) % 3: register set == ring slot of tile ph + 2
    const int ph = g * 9 + PH;
    if (!(dbg & 6)) {
      w_wait<SET, 16>();
      w_write<SET>(ph + 2);
      if constexpr (PH + 5 < 9) w_issue_at<SET>(PH + 5, chunk0, tn0);
      else w_issue_at<SET>(PH + 5 - 9, chunk1, tn1);
    }
    if (!(dbg & 32) || PH % 3 == 2) phase_barrier(trace);
  }
  __device__ __forceinline__ void step(int g) {
    const int g1 = g + 1 < nsteps ? g + 1 : nsteps - 1;
    const int chunk0 = g % nchunks, chunk1 = g1 % nchunks;
    const int tn0 = tn_of_step(g), tn1 = tn_of_step(g1);
    phase<0>(g, chunk0, tn0, chunk1, tn1); phase<1>(g, chunk0, tn0, chunk1, tn1); phase<2>(g, chunk0, tn0, chunk1, tn1);
    phase<3>(g, chunk0, tn0, chunk1, tn1); phase<4>(g, chunk0, tn0, chunk1, tn1); phase<5>(g, chunk0, tn0, chunk1, tn1);
    phase<6>(g, chunk0, tn0, chunk1, tn1); phase<7>(g, chunk0, tn0, chunk1, tn1); phase<8>(g, chunk0, tn0, chunk1, tn1);
  }
  __device__ __forceinline__ void finish() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // retire the clamped tail loads before the registers die
#pragma unroll 1
    for (int p = 0; p < 9; ++p) phase_barrier(trace);
  }
};

// ---- halo waves --------------------------------------------------------------------------------------
template <int TH, int TW, int BN>
struct HaloWaves : ProdBase<TH, TW, BN> {
  using Base = ProdBase<TH, TW, BN>;
  using G = typename Base::G;
  using Base::L; using Base::d; using Base::Ah0; using Base::nsteps; using Base::nchunks; using Base::dbg;
  using Base::trace; using Base::St; using Base::lane; using Base::fuse_stats; using Base::tiles_x; using Base::tiles_y;
  static constexpr int HP = G::HP, HALO = G::HALO;
  static constexpr int UPP = BN / 8;                   // stage units per pixel (8 or 16)
  static constexpr int NHW = 4 - BN / 64;              // halo waves: 3 (BN = 64) or 2 (BN = 128: two weight waves)
  static constexpr int RPP = NHW * 8;                  // halo rows per pass (8 lanes per 128-byte row)
  static constexpr int KU = (HALO + RPP - 1) / RPP;    // units per thread
  static constexpr int UPW = 3;                        // units written per phase
  static constexpr int NWP = (KU + UPW - 1) / UPW;     // write phases (the last NWP phases of a step)
  static constexpr int WRITE0 = 8 - NWP;               // written in phases WRITE0 .. 7 (visible before phase 8 ends)
  static_assert(WRITE0 >= 2, "halo loads need at least two phases of lead");
  static constexpr int NHT = NHW * 64;                 // halo/drain threads (192 or 128)
  static constexpr int DPP = (256 + NHT - 1) / NHT;    // drain passes per phase (256 stage units per phase)
  uint4 hreg[KU];
  int hrel[KU];                                        // source-pixel offset of unit k relative to the tile origin
  unsigned hedge[KU];                                  // which tile edges (or the halo end) invalidate unit k
  unsigned hvalid;
  float pa[8], pb[8];
  float gs, gq;
  float* red;                                          // [NHW][16][2] cross-wave statistics scratch
  int htid, hwave, slot, hrow, ecc, Hl, Wl;

  __device__ __forceinline__ HaloWaves(const ConvLaunch<bf16_t>& L_, char* smem, int htid_, int lane_, const TileMap& tm_,
                                       int nsteps_, int nchunks_, int fuse, int dbg_)
      : Base(L_, smem, lane_, tm_, nsteps_, nchunks_, fuse, dbg_), htid(htid_) {
    slot = htid & 7;
    hrow = htid >> 3;           // 0..RPP-1
    hwave = htid >> 6;
    ecc = htid % UPP;           // NHT is a multiple of UPP: a thread always drains the same 8-channel chunk
    gs = gq = 0.0f;
    red = reinterpret_cast<float*>(St + G::ST_BYTES);
    Hl = d.Hout;
    Wl = d.Wout;
    hvalid = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) { pa[u] = 1.0f; pb[u] = 0.0f; }
    // tile-independent part of every halo unit's address and validity, computed once: the per-step issue is then
    // two or three instructions per load instead of a division, four compares and a 64-bit multiply
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int hp = k * RPP + hrow;
      const int hy = hp / HP, hx = hp - hy * HP;
      int ry = hy - 1, rx = hx - 1;                    // tile origins are even, so the x2 gather is (origin/2) + (r >> 1)
      if (d.ups) { ry >>= 1; rx >>= 1; }
      hrel[k] = ry * d.Win + rx;
      hedge[k] = (hy == 0 ? 1u : 0u) | (hy == TH + 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == TW + 1 ? 8u : 0u) |
                 (hp >= HALO ? 16u : 0u);
    }
  }

  __device__ __forceinline__ void issue_all(int g_next) {
    int b, y0, x0, tn;
    this->decode(g_next / nchunks, b, y0, x0, tn);
    const int chunk = g_next % nchunks;
    const int c = chunk * kCH + slot * 8;
    const bool first = c < d.C0;
    const bf16_t* base = first ? L.src0 : L.src1;
    const int Cs = first ? d.C0 : d.C1, cc = first ? c : c - d.C0;
    hvalid = 0;
    const unsigned tedge = (y0 == 0 ? 1u : 0u) | (y0 + TH == Hl ? 2u : 0u) | (x0 == 0 ? 4u : 0u) |
                           (x0 + TW == Wl ? 8u : 0u) | 16u;
    const int64_t img0 = (int64_t)b * d.Hin * d.Win;                       // pixel (0,0) of the image: always mapped
    const int64_t org = img0 + (int64_t)(y0 >> d.ups) * d.Win + (x0 >> d.ups);
    const bf16_t* p_org = base + org * Cs + cc;
    const bf16_t* p_img = base + img0 * Cs + cc;
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const bool ok = (hedge[k] & tedge) == 0;
      // out-of-image taps read pixel (0,0) of the image and are zeroed at write time
      const bf16_t* p = ok ? p_org + (int64_t)hrel[k] * Cs : p_img;
      hreg[k] = *reinterpret_cast<const uint4*>(p);
      hvalid |= (ok ? 1u : 0u) << k;
    }
    if (L.pro_a) {
      const size_t o = (size_t)b * d.C0 + chunk * kCH + slot * 8;
      const float4 a0 = *reinterpret_cast<const float4*>(L.pro_a + o), a1 = *reinterpret_cast<const float4*>(L.pro_a + o + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(L.pro_b + o), b1 = *reinterpret_cast<const float4*>(L.pro_b + o + 4);
      pa[0] = a0.x; pa[1] = a0.y; pa[2] = a0.z; pa[3] = a0.w; pa[4] = a1.x; pa[5] = a1.y; pa[6] = a1.z; pa[7] = a1.w;
      pb[0] = b0.x; pb[1] = b0.y; pb[2] = b0.z; pb[3] = b0.w; pb[4] = b1.x; pb[5] = b1.y; pb[6] = b1.z; pb[7] = b1.w;
    }
  }
  template <int K>
  __device__ __forceinline__ void write_unit(int g_next) {
    if constexpr (K < KU) {
      const int hp = K * RPP + hrow;
      if (hp < HALO) {
        uint4* Ah = Ah0 + (g_next & 1) * HALO * 8;
        uint4 v = hreg[K];
        if (L.pro_a) {
          uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float lo = fast_silu(fmaf(bf_lo(w[j]), pa[2 * j], pb[2 * j]));
            const float hi2 = fast_silu(fmaf(bf_hi(w[j]), pa[2 * j + 1], pb[2 * j + 1]));
            w[j] = pack_bf16(lo, hi2);
          }
          v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        if (!((hvalid >> K) & 1u)) v = make_uint4(0, 0, 0, 0);
        Ah[hp * 8 + (slot ^ ((hp >> 1) & 7))] = v;
      }
    }
  }
  template <int WP>
  __device__ __forceinline__ void write_phase(int g_next) {   // WP = 0..NWP-1
    write_unit<WP * UPW + 0>(g_next);
    write_unit<WP * UPW + 1>(g_next);
    write_unit<WP * UPW + 2>(g_next);
  }
  template <int WP>
  __device__ __forceinline__ void prologue_writes() {
    if constexpr (WP < NWP) {
      write_phase<WP>(0);
      prologue_writes<WP + 1>();
    }
  }
  __device__ __forceinline__ void prologue() {
    issue_all(0);
    prologue_writes<0>();
  }
  // ---- stage drain + GroupNorm partial sums (previous tile) -------------------------------------------------
  // destination of the tile being drained, decoded once per step (not per phase: runtime divisions)
  int64_t dr_m0;
  bf16_t* dr_out;
  __device__ __forceinline__ void drain_setup(int it_prev) {
    int b, y0, x0, tn;
    this->decode(it_prev, b, y0, x0, tn);
    dr_m0 = ((int64_t)b * d.Hout + y0) * d.Wout + x0;
    dr_out = L.out + tn * BN + ecc * 8;
  }
  __device__ __forceinline__ void drain_slice(int p8) {
#pragma unroll
    for (int j = 0; j < DPP; ++j) {
      const int u = j * NHT + htid;                      // unit within this phase's 256
      if (u < 256) {
        const int px = (p8 * 256 + u) / UPP;
        const int k = px & 15;
        const int slot8 = ((2 * ecc) ^ k) & ~1;
        uint4 v = *reinterpret_cast<const uint4*>(St + (size_t)px * (BN * 2) + slot8 * 8);
        if (k & 1) v = make_uint4(v.z, v.w, v.x, v.y);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = bf_lo(w[q]), hi2 = bf_hi(w[q]);
          gs += lo + hi2;
          gq = fmaf(lo, lo, gq);
          gq = fmaf(hi2, hi2, gq);
        }
        const int64_t m = dr_m0 + (int64_t)(px / TW) * d.Wout + px % TW;
        *reinterpret_cast<uint4*>(dr_out + m * d.Cout) = v;
      }
    }
  }
  __device__ __forceinline__ void stats_reduce() {       // fold the row lanes of this wave, park per-wave chunk totals
#pragma unroll
    for (int o = UPP; o < 64; o <<= 1) {
      gs += __shfl_xor(gs, o, 64);
      gq += __shfl_xor(gq, o, 64);
    }
    if (lane < UPP) {
      red[(hwave * 16 + lane) * 2 + 0] = gs;
      red[(hwave * 16 + lane) * 2 + 1] = gq;
    }
    gs = 0.0f;
    gq = 0.0f;
  }
  __device__ __forceinline__ void stats_store(int it_prev) {   // fixed-order cross-wave sum: deterministic
    int b, y0, x0, tn;
    this->decode(it_prev, b, y0, x0, tn);
    const int cpg = d.Cout / L.gn_groups;                // multiple of 8, <= BN
    const int per = cpg / 8, ngrp = BN / cpg;
    if (htid < ngrp) {
      float ss = 0.0f, qq = 0.0f;
      for (int ch = 0; ch < per; ++ch)
        for (int w = 0; w < NHW; ++w) {
          ss += red[(w * 16 + htid * per + ch) * 2 + 0];
          qq += red[(w * 16 + htid * per + ch) * 2 + 1];
        }
      const int nsplit = tiles_x * tiles_y;
      const int slab = (y0 / TH) * tiles_x + x0 / TW;
      float* dst = L.gn_partials + (((size_t)b * nsplit + slab) * L.gn_groups + tn * BN / cpg + htid) * 2;
      dst[0] = ss;
      dst[1] = qq;
    }
  }

  template <int PH>
  __device__ __forceinline__ void phase(int g, bool have_next, bool draining, int it_prev) {
    if (have_next && !(dbg & 10)) {
      if constexpr (PH == 0) issue_all(g + 1);
      if constexpr (PH >= WRITE0 && PH - WRITE0 < NWP) write_phase<PH - WRITE0>(g + 1);
    }
    if (draining && !(dbg & 18)) {
      if constexpr (PH < 8) {
        drain_slice(PH);
      } else {
        if (fuse_stats) stats_reduce();
      }
    }
    if (!(dbg & 32) || PH % 3 == 2) phase_barrier(trace);
    if constexpr (PH == 8) {
      if (draining && fuse_stats && !(dbg & 18)) stats_store(it_prev);   // red[] complete; next write is a tile away
    }
  }
  __device__ __forceinline__ void step(int g) {
    const bool have_next = g + 1 < nsteps;
    const int chunk = g % nchunks;
    const bool draining = chunk == 0 && g >= nchunks;      // a finished tile sits in the stage
    const int it_prev = g / nchunks - 1;
    if (draining) drain_setup(it_prev);
    phase<0>(g, have_next, draining, it_prev); phase<1>(g, have_next, draining, it_prev);
    phase<2>(g, have_next, draining, it_prev); phase<3>(g, have_next, draining, it_prev);
    phase<4>(g, have_next, draining, it_prev); phase<5>(g, have_next, draining, it_prev);
    phase<6>(g, have_next, draining, it_prev); phase<7>(g, have_next, draining, it_prev);
    phase<8>(g, have_next, draining, it_prev);
  }
  __device__ __forceinline__ void drain_last(int it_last) {
    drain_setup(it_last);
#pragma unroll 1
    for (int p = 0; p < 9; ++p) {
      if (p < 8) {
        drain_slice(p);
      } else if (fuse_stats) {
        stats_reduce();
      }
      phase_barrier(trace);
    }
    if (fuse_stats) stats_store(it_last);
  }
};

template <int TH, int TW, int BN>
__global__ __launch_bounds__(512, 2) void conv3x3_ws_kernel(const ConvLaunch<bf16_t> L, const int tiles_x,
                                                            const int tiles_y, const int tiles_n,
                                                            const int total_tiles, const int fuse_stats, const int dbg,
                                                            unsigned long long* const trace) {
  using G = WsGeom<TH, TW, BN>;
  constexpr int HP = G::HP, HALO = G::HALO;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* const Ah0 = reinterpret_cast<uint4*>(smem);
  uint4* const Bw0 = reinterpret_cast<uint4*>(smem + 2 * G::AH_BYTES);
  char* const St = smem + 2 * G::AH_BYTES + 3 * G::BW_BYTES;
  float* const red = reinterpret_cast<float*>(St + G::ST_BYTES);      // [4 producer waves][16 chunks][2]

  const ConvDesc& d = L.d;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool consumer = wave < 4;
  const int Cin = d.C0 + d.C1;
  const int nchunks = Cin / kCH;
  const int Hl = d.Hout, Wl = d.Wout;   // conv input extent (after the optional upsample) == output extent

  TileMap tmap;
  tmap.init((int)blockIdx.x, (int)gridDim.x, tiles_x, tiles_y, tiles_n, d.B, TH, TW);
  const int my_tiles = tmap.count;
  const int nsteps = my_tiles * nchunks;
  if (nsteps == 0) return;
  auto decode = [&](int it, int& b, int& y0, int& x0, int& tn) { tmap.decode(it, b, y0, x0, tn); };

  // ===================================================================================================
  if (consumer) {
    // The MFMA waves share each SIMD's issue port with one producer wave: static priority keeps the matrix pipe fed
    // (instruction arbitration is by priority, then age) while the producers fill the gaps.
    __builtin_amdgcn_s_setprio(3);
    const int wm = wave / G::WAVES_N, wn = wave % G::WAVES_N;
    const int l31 = lane & 31, hi = lane >> 5;
    int ahp[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int p = wm * 64 + pt * 32 + l31;
      ahp[pt] = (p / TW) * HP + (p % TW);
    }
    int wrow[2], wswz[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      wrow[ct] = wn * 64 + ct * 32 + l31;
      wswz[ct] = (wrow[ct] >> 1) & 7;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ct][pt][e] = 0.0f;

    phase_barrier(trace);   // prologue barrier: first halo + weight tiles 0,1 are in LDS
    // Fragment double buffer: the reads of call c+1 (or of the NEXT phase's call 0) are issued before the MFMAs
    // of call c, so LDS latency hides behind 128 cycles of matrix work instead of idling the pipe.
    bf16x8 fw[4][2], fx[4][2];   // one fragment set per call of a phase; loads run TWO calls (256 MFMA cycles) ahead
    auto frag_load = [&](int set, const uint4* Ahb, const uint4* Bwb, int toff, int call) {
      const int unit = call * 2 + hi;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) fw[set][ct] = *reinterpret_cast<const bf16x8*>(Bwb + wrow[ct] * 8 + (unit ^ wswz[ct]));
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        const int hp = ahp[pt] + toff;
        fx[set][pt] = *reinterpret_cast<const bf16x8*>(Ahb + hp * 8 + (unit ^ ((hp >> 1) & 7)));
      }
    };
    frag_load(0, Ah0, Bw0, 0, 0);
    frag_load(1, Ah0, Bw0, 0, 1);
    int ph = 0;
    for (int g = 0; g < nsteps; ++g) {
      const uint4* Ah = Ah0 + (g & 1) * HALO * 8;
      const uint4* AhN = Ah0 + ((g + 1) & 1) * HALO * 8;
      const int chunk = g % nchunks;
#pragma unroll 1
      for (int p = 0; p < 9; ++p, ++ph) {
        if (dbg & 1) { if (!(dbg & 32) || p % 3 == 2) phase_barrier(trace); continue; }
        const uint4* Bw = Bw0 + (ph % 3) * BN * 8;
        const uint4* BwN = Bw0 + ((ph + 1) % 3) * BN * 8;
        const int kh = (p * 11) >> 5, kw = p - kh * 3;
        const int toff = kh * HP + kw;
        const int pn = p == 8 ? 0 : p + 1;
        const int khn = (pn * 11) >> 5, kwn = pn - khn * 3;
        const int toffN = khn * HP + kwn;
#pragma unroll
        for (int call = 0; call < 4; ++call) {
          // set (call + 2) % 4 was consumed two calls ago: refill it for the call two ahead (this phase's calls 2,3
          // or the NEXT phase's calls 0,1 — its weight tile and halo are already visible in LDS)
          if (call < 2) frag_load(call + 2, Ah, Bw, toff, call + 2);
          else frag_load(call - 2, p == 8 ? AhN : Ah, BwN, toffN, call - 2);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
              acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[call][ct], fx[call][pt], acc[ct][pt], 0, 0, 0);
        }
        if (p == 8 && chunk == nchunks - 1) {
          // tile finished: bias, round, park in the stage.  Lane holds pixel (pt*32 + l31), channels
          // ct*32 + 8q + 4hi + {0..3}.  Stage row = pixel, BN*2 bytes, 8-byte slots XOR-swizzled by (pixel & 15).
          int b, y0, x0, tn;
          decode(g / nchunks, b, y0, x0, tn);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int c = wn * 64 + ct * 32 + 8 * q + 4 * hi;      // channel within the BN tile
              float bv[4] = {0, 0, 0, 0};
              if (L.bias) {
                const float4 t4 = *reinterpret_cast<const float4*>(L.bias + tn * BN + c);
                bv[0] = t4.x; bv[1] = t4.y; bv[2] = t4.z; bv[3] = t4.w;
              }
#pragma unroll
              for (int pt = 0; pt < 2; ++pt) {
                const int px = wm * 64 + pt * 32 + l31;
                uint2 w;
                w.x = pack_bf16(acc[ct][pt][4 * q + 0] + bv[0], acc[ct][pt][4 * q + 1] + bv[1]);
                w.y = pack_bf16(acc[ct][pt][4 * q + 2] + bv[2], acc[ct][pt][4 * q + 3] + bv[3]);
                const int slot8 = (c >> 2) ^ (px & 15);
                *reinterpret_cast<uint2*>(St + (size_t)px * (BN * 2) + slot8 * 8) = w;
              }
            }
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
              for (int e = 0; e < 16; ++e) acc[ct][pt][e] = 0.0f;
        }
        if (!(dbg & 32) || p % 3 == 2) phase_barrier(trace);
      }
    }
    // drain rounds: the producers flush the last tile (8 slices + statistics), consumers only keep the barrier count
#pragma unroll 1
    for (int p = 0; p < 9; ++p) phase_barrier(trace);
    return;
  }

  // ===================================================================================================
  // PRODUCERS
  constexpr int NWW = BN / 64;                       // weight waves
  if (wave < 4 + NWW) {
    WeightWave<TH, TW, BN> Wv(L, smem, lane, wave - 4, tmap, nsteps, nchunks, fuse_stats, dbg);
    Wv.trace = trace;
    Wv.prologue();
    phase_barrier(trace);
#pragma unroll 1
    for (int g = 0; g < nsteps; ++g) Wv.step(g);
    Wv.finish();
  } else {
    HaloWaves<TH, TW, BN> Hv(L, smem, tid - (4 + NWW) * 64, lane, tmap, nsteps, nchunks, fuse_stats, dbg);
    Hv.trace = trace;
    Hv.prologue();
    phase_barrier(trace);
#pragma unroll 1
    for (int g = 0; g < nsteps; ++g) Hv.step(g);
    Hv.drain_last(my_tiles - 1);
  }
}

template <int TH, int TW, int BN>
int launch_ws_cfg(const ConvLaunch<bf16_t>& L, hipStream_t s, int fuse_stats, int* nsplit, int num_cus) {
  using G = WsGeom<TH, TW, BN>;
  const ConvDesc& d = L.d;
  const int tiles_x = d.Wout / TW, tiles_y = d.Hout / TH, tiles_n = d.Cout / BN;
  const int total = tiles_x * tiles_y * tiles_n * d.B;
  int grid = num_cus;
  if (grid > total) grid = total;
  if (grid >= 8) grid &= ~7;                     // multiple of 8: XCD-contiguous runs inside a round
  static const int dbg = [] { const char* e = std::getenv("PRG_WS_DBG"); return e ? std::atoi(e) : 0; }();
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ws_kernel<TH, TW, BN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    if (e != hipSuccess) return fail(PRG_E_HIP, std::string("hipFuncSetAttribute(ws conv): ") + hipGetErrorString(e));
    attr_done = true;
  }
  if (nsplit) *nsplit = fuse_stats ? tiles_x * tiles_y : 0;
  static const int trace_at = [] { const char* e = std::getenv("PRG_WS_TRACE"); return e ? std::atoi(e) : -1; }();
  static int launch_no = 0;
  unsigned long long* tbuf = nullptr;
  if (trace_at >= 0 && launch_no++ == trace_at) {
    if (hipHostMalloc(reinterpret_cast<void**>(&tbuf), 8 * kTraceStride * sizeof(unsigned long long)) == hipSuccess) {
      std::memset(tbuf, 0, 8 * kTraceStride * sizeof(unsigned long long));
      (void)hipStreamSynchronize(s);
    }
  }
  conv3x3_ws_kernel<TH, TW, BN><<<dim3(grid), 512, G::LDS, s>>>(L, tiles_x, tiles_y, tiles_n, total, fuse_stats, dbg, tbuf);
  PRG_LAUNCH_CHECK();
  if (tbuf) {
    (void)hipStreamSynchronize(s);
    char path[256];
    std::snprintf(path, sizeof(path), "%s/ws_trace_%d_%d_%d_cin%d.bin", std::getenv("PRG_WS_TRACE_DIR") ? std::getenv("PRG_WS_TRACE_DIR") : "/tmp",
                  TH, TW, BN, d.C0 + d.C1);
    if (FILE* f = std::fopen(path, "wb")) {
      std::fwrite(tbuf, sizeof(unsigned long long), 8 * kTraceStride, f);
      std::fclose(f);
    }
    (void)hipHostFree(tbuf);
  }
  return PRG_OK;
}

}  // namespace

// Returns 1 when it launched, 0 when the shape is not covered (caller falls back), negative on error.
int try_launch_conv3x3_ws(const ConvLaunch<bf16_t>& L, hipStream_t s, int* gn_nsplit_out) {
  static const int enabled = [] {
    const char* e = std::getenv("PRG_CONV_WS");
    return e ? std::atoi(e) : 1;
  }();
  if (!enabled) return 0;
  const ConvDesc& d = L.d;
  if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1)) return 0;
  if (d.C0 % kCH || d.C1 % kCH || d.Cout % 64) return 0;
  if (L.residual) return 0;
  static int num_cus = 0;
  if (!num_cus) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    num_cus = p.multiProcessorCount;
  }
  const int H = d.Hout, W = d.Wout;
  const int cpg = L.gn_groups > 0 ? d.Cout / L.gn_groups : 0;
  const bool want = L.gn_partials != nullptr;
  auto fuse_for = [&](int TH, int TW, int BN) {
    return want && cpg % 8 == 0 && cpg <= BN && (W / TW) * (H / TH) <= kGnMaxSplit ? 1 : 0;
  };
  int rc = 0;
  if (d.Cout % 128 == 0) {
    if (W % 32 == 0 && H % 4 == 0) rc = launch_ws_cfg<4, 32, 128>(L, s, fuse_for(4, 32, 128), gn_nsplit_out, num_cus);
    else if (W % 16 == 0 && H % 8 == 0) rc = launch_ws_cfg<8, 16, 128>(L, s, fuse_for(8, 16, 128), gn_nsplit_out, num_cus);
    else return 0;
  } else {
    if (W % 32 == 0 && H % 8 == 0) rc = launch_ws_cfg<8, 32, 64>(L, s, fuse_for(8, 32, 64), gn_nsplit_out, num_cus);
    else return 0;
  }
  return rc == PRG_OK ? 1 : rc;
}

}  // namespace prg
