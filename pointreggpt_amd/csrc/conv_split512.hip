// conv_split512.hip — the f16x3 3x3 convolution with ONE WAVE PER SIMD (round 6; VERDICT round 5, item 2 "the structural one").
//
// conv3x3_split_ws_kernel (conv_split.hip) gives each consumer wave a 64-pixel x 64-channel tile: 8 fragment reads per 12 MFMAs
// (0.67), because at two waves per SIMD a wave has 256 registers and the product + flush accumulator sets of a larger tile do not
// fit.  Here a 256-thread workgroup owns a CU with one 512-register wave on every SIMD:
//   * wave tile 128 pixels x 64 channels (4 x 2 MFMA tiles): 12 fragment reads per 24 MFMAs (0.5), 256 accumulator registers
//     (product set + the 288-term flush set) in the AccVGPR half of the unified file — this file is built WITHOUT
//     -amdgpu-mfma-vgpr-form so that hipcc may place MFMA results there;
//   * workgroup tile 256 pixels (16 x 16) x 128 channels: one 16 KB weight tile per tap feeds twice the pixels of the 128-pixel
//     kernel (half the L2 -> LDS weight traffic per FLOP);
//   * no producer waves: every wave moves a quarter of each weight tile (LDS-DMA, ring of three, one tap ahead) and stages a quarter
//     of the next chunk's halo (loads at tap 0, prologue + split + ds_write one pass per tap at taps 2-7) inside its own MFMA
//     stream — one s_barrier per tap for four waves.
// The arithmetic is conv3x3_split_ws_kernel's term for term (same MFMA sequence per accumulator, 288-term partials, same epilogue
// expression, same GroupNorm slab partition: a wave's 128 pixels are one 8 x 16 tile of that kernel), so the two are bit-identical
// (tests/test_gpu_f16x3.py::test_f16x3_one_wave_per_simd_kernel).  Selected by PRG_SPLIT_W512=1; the same-box A/B is
// profiles/r06_ab_split_w512.txt.
#include <atomic>
#include <cstdlib>

#include "conv_split_common.h"

namespace prg {

template <int NS>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_split_w512_kernel(
    const ConvLaunch<float> L, const int tiles_x, const int tiles_y, const int tiles_n, const int fuse_stats) {
  constexpr int TH = 16, TW = 16, BN = 128, CH = 32, NT = 9;
  constexpr int HP = TW + 2, HALO = (TH + 2) * HP, PITCH = 144;
  constexpr int RSTRIDE = (HP * PITCH + 255) / 256 * 256, HBYTES = (TH + 2) * RSTRIDE;
  constexpr int NHP = (HALO * 4 + 255) / 256;            // halo staging passes of the 256 threads, 64 halo pixels each (6)
  constexpr int WPW = BN / 8 / 4;                        // global_load_lds instructions per wave and weight tile (4)
  static_assert(NS == 3 && NHP == 6 && NT % NS == 0, "ring / passes");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ah = smem;                                 // [2][18 rows of RSTRIDE bytes]: 144-byte pixels, units 0-3 hi, 4-7 lo
  char* const Bs = smem + 2 * HBYTES;                    // [NS][128][128 B], 16-byte units XOR-swizzled by (row >> 1) & 7

  const ConvDesc& d = L.d;
  const int nblk = tiles_x * tiles_y * tiles_n * d.B;
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
  const int tx = lin % tiles_x; lin /= tiles_x;
  const int ty = lin % tiles_y;
  const int b = lin / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;               // pixel half (eight tile rows) x channel half
  const int nchunks = (d.C0 + d.C1) / CH, niter = nchunks * NT;
  const size_t wstep = (size_t)d.CoutPad * 128;
  const char* wtile = reinterpret_cast<const char*>(L.w_split + (size_t)tn * BN * 64);
  const int Hs = d.Hout, Ws = d.Wout;

  // ---- halo staging (this thread: channels q * 8 .. + 7 of halo pixels prow + 64 k) ----
  const int q = tid & 3, prow = tid >> 2;
  int hsrc[NHP], w_lane[NHP];
#pragma unroll
  for (int k = 0; k < NHP; ++k) {
    const int hp = prow + k * 64;
    hsrc[k] = -1;
    w_lane[k] = 0;
    if (hp < HALO) {
      const int hy = hp / HP, hx = hp - hy * HP;
      int y = y0 - 1 + hy, x = x0 - 1 + hx;
      if ((unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws) {
        if (d.ups) { y >>= 1; x >>= 1; }
        hsrc[k] = (b * d.Hin + y) * d.Win + x;
      }
      w_lane[k] = hy * RSTRIDE + hx * PITCH + q * 16;
    }
  }
  float4 hh0[3], hh1[3];
  // passes [K0, K0 + 3) of a chunk: 6 loads per thread whatever the passes hold (the counted waits need a fixed number); the two
  // halves of a chunk's halo are loaded at taps 0 and 3 and share ONE set of three staging register pairs
  auto halo_load = [&](int chunk, auto K0) {
    constexpr int k0 = decltype(K0)::value;
    const int c = chunk * CH + q * 8;
    const bool first = c < d.C0;
    const float* base = first ? L.src0 : L.src1;
    const int Cs = first ? d.C0 : d.C1, cc = first ? c : c - d.C0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4* p = reinterpret_cast<const float4*>(base + (hsrc[k0 + k] >= 0 ? (size_t)hsrc[k0 + k] * Cs + cc : (size_t)0));
      hh0[k] = p[0];
      hh1[k] = p[1];
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
  auto halo_write = [&](int buf, auto K) {
    constexpr int k = decltype(K)::value;
    if (k * 64 + 63 < HALO || prow + k * 64 < HALO) {
      constexpr int r = k % 3;
      float v[8] = {hh0[r].x, hh0[r].y, hh0[r].z, hh0[r].w, hh1[r].x, hh1[r].y, hh1[r].z, hh1[r].w};
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

  // ---- weight tiles: LDS-DMA, lane-linear destination, the XOR swizzle on the per-lane source address ----
  int wsrc[WPW];
#pragma unroll
  for (int r = 0; r < WPW; ++r) {
    const int n = (wave * WPW + r) * 8 + (lane >> 3);
    wsrc[r] = n * 128 + (((lane & 7) ^ ((n >> 1) & 7)) << 4);
  }
  auto gload_b = [&](int chunk, int tap, int slot) {
    const char* p = wtile + (size_t)(tap * L.split_kchunks + chunk) * wstep;
    char* dst = Bs + (slot * BN + wave * WPW * 8) * 128;
#pragma unroll
    for (int r = 0; r < WPW; ++r)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + wsrc[r]),
                                       (__attribute__((address_space(3))) void*)(dst + r * 1024), 16, 0, 0);
  };

  // ---- fragments ----
  const int a_lane = (wm * 8 + (l31 >> 4)) * RSTRIDE + (l31 & 15) * PITCH + hi * 16;   // row tile i: + 2 i RSTRIDE
  int b_lane[2][2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int sw = (l31 >> 1) & 7, unit = 2 * st + hi;
    b_lane[st][0] = wn * 8192 + l31 * 128 + ((unit ^ sw) << 4);
    b_lane[st][1] = wn * 8192 + l31 * 128 + (((4 + unit) ^ sw) << 4);
  }
  f32x16 acc[4][2], tot[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.0f; tot[i][j][e] = 0.0f; }
  f16x8 fa[2][8], fw[2][4];                              // [k16 step][A: (hi, lo) x row tile | W: (hi, lo) x column tile]
  auto reads = [&](auto ST, auto TAP, int cb) {
    constexpr int st = decltype(ST)::value, T = decltype(TAP)::value;
    constexpr int toff = (T / 3) * RSTRIDE + (T % 3) * PITCH;
    const char* A = Ah + cb * HBYTES + toff + st * 32 + a_lane;
    const char* Bb = Bs + (T % NS) * (BN * 128);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[st][2 * i] = ld_frag(reinterpret_cast<const uint4*>(A + i * 2 * RSTRIDE));
      fa[st][2 * i + 1] = ld_frag(reinterpret_cast<const uint4*>(A + i * 2 * RSTRIDE + 64));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fw[st][2 * j] = ld_frag(reinterpret_cast<const uint4*>(Bb + b_lane[st][0] + j * 4096));
      fw[st][2 * j + 1] = ld_frag(reinterpret_cast<const uint4*>(Bb + b_lane[st][1] + j * 4096));
    }
  };
  auto mfmas = [&](auto ST) {                            // term-major: back-to-back MFMAs never wait on the same accumulator
    constexpr int st = decltype(ST)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st][2 * i + 1], fw[st][2 * j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st][2 * i], fw[st][2 * j + 1], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st][2 * i], fw[st][2 * j], acc[i][j], 0, 0, 0);
  };
  // scheduling pipelines of the two k16 steps of a tap (sched_group_barrier masks: 0x008 MFMA, 0x100 DS read, 0x200 DS write, 0x020 VMEM
  // read, 0x002 VALU): 24 MFMAs each, the 12 fragment reads of the OTHER register set one per MFMA shadow, and the tap's producer work —
  // four LDS-DMA pieces in the first half; the staging VALU, the halo's global loads and the two LDS writes in the second
  auto pipeline_a = [&]() {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
  };
  auto pipeline_b = [&]() {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
  };

  // one tap of chunk c.  Invariant at its barrier: weight tiles it and it + 1 are in LDS (tile it + 1 was issued a tap ago), tile
  // it - 1's slot is free, this chunk's halo is complete; the other halo buffer was last read before the previous barrier.
  auto body = [&](auto TAP, int c) {
    constexpr int T = decltype(TAP)::value;
    const int it = c * NT + T;
    const bool more = c + 1 < nchunks;
    // the in-order VMEM queue behind the DMA of tile it + 1: only the halo (and coefficient) loads the previous tap issued
    // (taps 0 and 3: six loads, four more at tap 0 when the launch has a fused prologue)
    if (T == 1 && more) {
      if (L.pro_a) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else if (T == 4 && more) {
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // first half: the DMA of tile it + 2 (into tile it - 1's slot), the fragments of step 1, the MFMAs of step 0.  Its own scheduling
    // region, so that the DMA stays IN FRONT of the second half's halo loads in the in-order VMEM queue (the counted wait above)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (T + 2 < NT) { gload_b(c, T + 2, (T + 2) % NS); }
    else if (more) gload_b(c + 1, T + 2 - NT, (T + 2) % NS);
    reads(IC<1>(), TAP, c & 1);
    mfmas(IC<0>());
    pipeline_a();
    __builtin_amdgcn_sched_barrier(0);
    // second half: the next chunk's staging (loads at tap 0, one converted pass at taps 2-7), the next tap's first fragments, step 1
    if (more) {
      // passes 0-2: loaded at tap 0 (a tap and a half = ~3000 cycles before their first use), written at taps 2 (two passes) and 3;
      // the set is then free for passes 3-5: loaded at tap 3 behind that write, written at taps 5, 6, 7; tap 8 hands the buffer over
      if constexpr (T == 0) { pro_load(c + 1); halo_load(c + 1, IC<0>()); }
      if constexpr (T == 2) { halo_write((c + 1) & 1, IC<0>()); halo_write((c + 1) & 1, IC<1>()); }
      if constexpr (T == 3) { halo_write((c + 1) & 1, IC<2>()); halo_load(c + 1, IC<3>()); }
      if constexpr (T == 5) halo_write((c + 1) & 1, IC<3>());
      if constexpr (T == 6) halo_write((c + 1) & 1, IC<4>());
      if constexpr (T == 7) halo_write((c + 1) & 1, IC<5>());
    }
    if constexpr (T < NT - 1) reads(IC<0>(), IC<T + 1>(), c & 1);
    else if (more) reads(IC<0>(), IC<0>(), (c + 1) & 1);
    mfmas(IC<1>());
    pipeline_b();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (T == NT - 1) {                         // the 288-term partial of this channel chunk
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) { tot[i][j][e] += acc[i][j][e]; acc[i][j][e] = 0.0f; }
    }
  };

  // ---- prologue: weight tiles 0 and 1, chunk 0's halo ----
  gload_b(0, 0, 0);
  if (niter > 1) gload_b(0, 1, 1);
  pro_load(0);
  halo_load(0, IC<0>());
  halo_write(0, IC<0>()); halo_write(0, IC<1>()); halo_write(0, IC<2>());
  halo_load(0, IC<3>());
  halo_write(0, IC<3>()); halo_write(0, IC<4>()); halo_write(0, IC<5>());
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  reads(IC<0>(), IC<0>(), 0);
  for (int c = 0; c < nchunks; ++c) {
    body(IC<0>(), c); body(IC<1>(), c); body(IC<2>(), c); body(IC<3>(), c); body(IC<4>(), c);
    body(IC<5>(), c); body(IC<6>(), c); body(IC<7>(), c); body(IC<8>(), c);
  }

  // ---- direct epilogue (conv3x3_split_ws_kernel's, per 8 x 16 pixel half): register e of lane half hi is pixel row
  //      (e & 3) + 8 (e >> 2) + 4 hi of a 32-pixel row tile, lanes 0-31 are 32 consecutive channels ----
  {
    const int cpg = L.gn_groups > 0 ? d.Cout / L.gn_groups : 64;
    const size_t ps = (size_t)d.Cout;                                       // floats between the lane's consecutive tile columns
    const size_t rs = (size_t)d.Wout * d.Cout;                              // ... and tile rows
    float* const lbase = L.out + (((size_t)b * d.Hout + y0 + wm * 8) * d.Wout + x0) * d.Cout + tn * BN + wn * 64 + (size_t)(4 * hi) * ps + l31;
    float bv[2], sc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ch = tn * BN + wn * 64 + j * 32 + l31;
      bv[j] = L.bias ? L.bias[ch] : 0.0f;
      sc[j] = L.split_scale ? L.split_scale[ch] : 1.0f;                     // the packer's per-channel power of two, undone (exact)
    }
    const int nsplit = tiles_x * tiles_y * 2 * 2;                           // the 128-pixel kernel's slabs: (8 x 16 tile, pixel half)
#pragma unroll
    for (int half = 0; half < 2; ++half) {                                  // row tiles 2 half, 2 half + 1 = one 64-pixel slab of that kernel
      float s1[2] = {0.0f, 0.0f}, q1[2] = {0.0f, 0.0f};
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * half + ii;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float* const cp = lbase + (size_t)(2 * i + (e >> 3)) * rs + (size_t)((e & 3) + 8 * ((e >> 2) & 1)) * ps;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float v = tot[i][j][e] * sc[j] + bv[j];
            cp[j * 32] = v;
            s1[j] += v;
            q1[j] = fmaf(v, v, q1[j]);
          }
        }
      }
      if (fuse_stats) {
        double sd[2], qd[2];
        const int width = cpg < 32 ? cpg : 32;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          sd[j] = (double)s1[j];
          qd[j] = (double)q1[j];
          sd[j] += __shfl_xor(sd[j], 32, 64);
          qd[j] += __shfl_xor(qd[j], 32, 64);
          for (int o = 1; o < width; o <<= 1) {
            sd[j] += __shfl_xor(sd[j], o, 64);
            qd[j] += __shfl_xor(qd[j], o, 64);
          }
        }
        // slab of the 128-pixel kernel: its tile (ty', tx) = (2 ty + wm, tx) of an (H / 8) x (W / 16) grid, pixel half `half`
        float* slab = L.gn_partials + ((size_t)b * nsplit + ((2 * ty + wm) * tiles_x + tx) * 2 + half) * L.gn_groups * 2;
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
}

// 1 = launched, 0 = not this kernel's shape / not enabled, < 0 = error.  PRG_SPLIT_W512=1 enables it (experiment: see the header).
int try_launch_conv3x3_split_w512(const ConvLaunch<float>& L, hipStream_t s, int want_stats, int* gn_nsplit_out) {
  static const int on = [] { const char* e = std::getenv("PRG_SPLIT_W512"); return e ? std::atoi(e) : 0; }();
  if (!on) return 0;
  const ConvDesc& d = L.d;
  if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.Cout % 128 == 0 && d.Wout % 16 == 0 && d.Hout % 16 == 0 && !L.residual &&
        d.C0 % 32 == 0 && d.C1 % 32 == 0))
    return 0;
  constexpr int NS = 3, HB = 18 * ((18 * 144 + 255) / 256 * 256);
  const int tiles_x = d.Wout / 16, tiles_y = d.Hout / 16, tiles_n = d.Cout / 128;
  const int cpg = L.gn_groups > 0 ? d.Cout / L.gn_groups : 0;
  const int slabs = tiles_x * tiles_y * 4;
  const int f = want_stats && cpg % 8 == 0 && cpg <= 64 && (cpg & (cpg - 1)) == 0 && slabs <= kGnMaxSplit;
  if (want_stats && !f) return 0;
  const size_t lds = (size_t)2 * HB + NS * 128 * 128;
  if (gn_nsplit_out) *gn_nsplit_out = f ? slabs : 0;
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    PRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_w512_kernel<NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    attr_done.mark();
  }
  conv3x3_split_w512_kernel<NS><<<dim3(tiles_x * tiles_y * tiles_n * d.B), 256, lds, s>>>(L, tiles_x, tiles_y, tiles_n, f);
  PRG_LAUNCH_CHECK();
  return 1;
}

}  // namespace prg
