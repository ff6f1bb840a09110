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
// (tests/test_gpu_f16x3.py::test_f16x3_one_wave_per_simd_kernel).  Dispatch: try_launch_conv3x3_split_w512 below; the same-box A/B is
// profiles/r06_ab_split_w512.txt.
#include <atomic>
#include <cstdlib>

#include "conv_split_common.h"

namespace prg {

template <int NS, bool PRO>
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
  const int nchunks = (d.C0 + d.C1) / CH;
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
  // passes [K0, K0 + 3) of a chunk are loaded together (taps 0 and 3) and share ONE set of three staging register pairs
  float4 hh0[3], hh1[3];
  const float* hbase = nullptr;                          // source tensor / row pitch / channel offset of the chunk being staged
  int hCs = 0, hcc = 0;
  auto halo_chunk = [&](int chunk) {
    const int c = chunk * CH + q * 8;
    const bool first = c < d.C0;
    hbase = first ? L.src0 : L.src1;
    hCs = first ? d.C0 : d.C1;
    hcc = first ? c : c - d.C0;
  };
  auto halo_load1 = [&](auto K) {                        // always issued (padding lanes read the tensor's first bytes and are zeroed
    constexpr int k = decltype(K)::value, r = k % 3;     // when written): the counted waits need a fixed number of loads
    const float4* p = reinterpret_cast<const float4*>(hbase + (hsrc[k] >= 0 ? (size_t)hsrc[k] * hCs + hcc : (size_t)0));
    hh0[r] = p[0];
    hh1[r] = p[1];
  };
  float pa[8], pb[8];
  auto pro_load = [&](int chunk) {
    if constexpr (PRO) {
      const float4* a4 = reinterpret_cast<const float4*>(L.pro_a + (size_t)b * d.C0 + chunk * CH + q * 8);
      const float4* b4 = reinterpret_cast<const float4*>(L.pro_b + (size_t)b * d.C0 + chunk * CH + q * 8);
      const float4 a0 = a4[0], a1 = a4[1], b0 = b4[0], b1 = b4[1];
      pa[0] = a0.x; pa[1] = a0.y; pa[2] = a0.z; pa[3] = a0.w; pa[4] = a1.x; pa[5] = a1.y; pa[6] = a1.z; pa[7] = a1.w;
      pb[0] = b0.x; pb[1] = b0.y; pb[2] = b0.z; pb[3] = b0.w; pb[4] = b1.x; pb[5] = b1.y; pb[6] = b1.z; pb[7] = b1.w;
    }
  };
  // one pass = four pair slices (prologue + split of two channels) and a write slice: five MFMA shadows
  uint32_t hp2[4], lp2[4];
  auto stage_slice = [&](int buf, auto K, auto P) {
    constexpr int k = decltype(K)::value, p = decltype(P)::value, r = k % 3;
    if constexpr (p < 4) {
      float v0 = p == 0 ? hh0[r].x : p == 1 ? hh0[r].z : p == 2 ? hh1[r].x : hh1[r].z;
      float v1 = p == 0 ? hh0[r].y : p == 1 ? hh0[r].w : p == 2 ? hh1[r].y : hh1[r].w;
      if constexpr (PRO) {
        v0 = silu_fast(fmaf(v0, pa[2 * p], pb[2 * p]));
        v1 = silu_fast(fmaf(v1, pa[2 * p + 1], pb[2 * p + 1]));
      }
      if (hsrc[k] < 0) { v0 = 0.0f; v1 = 0.0f; }
      split2<!PRO>(v0, v1, hp2[p], lp2[p]);
    } else {
      if (k * 64 + 63 < HALO || prow + k * 64 < HALO) {
        char* dst = Ah + buf * HBYTES + w_lane[k];
        *reinterpret_cast<uint4*>(dst) = make_uint4(hp2[0], hp2[1], hp2[2], hp2[3]);
        *reinterpret_cast<uint4*>(dst + 64) = make_uint4(lp2[0], lp2[1], lp2[2], lp2[3]);
      }
    }
  };
  auto stage_pass = [&](int buf, auto K) {
    stage_slice(buf, K, IC<0>()); stage_slice(buf, K, IC<1>()); stage_slice(buf, K, IC<2>()); stage_slice(buf, K, IC<3>());
    stage_slice(buf, K, IC<4>());
  };

  // ---- weight tiles: LDS-DMA, lane-linear destination, the XOR swizzle on the per-lane source address ----
  int wsrc[WPW];
#pragma unroll
  for (int r = 0; r < WPW; ++r) {
    const int n = (wave * WPW + r) * 8 + (lane >> 3);
    wsrc[r] = n * 128 + (((lane & 7) ^ ((n >> 1) & 7)) << 4);
  }
  auto dma_piece = [&](int chunk, int tap, int slot, auto R) {
    constexpr int r = decltype(R)::value;
    const char* p = wtile + (size_t)(tap * L.split_kchunks + chunk) * wstep;
    char* dst = Bs + (slot * BN + wave * WPW * 8) * 128;
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
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.0f; tot[i][j][e] = 0.0f; }   // (acc: dead store, every chunk starts from the constant)
  f16x8 fa[2][8], fw[2][4];                              // [k16 step][A: (hi, lo) x row tile | W: (hi, lo) x column tile]
  // fragment R (0 .. 7: A rows, 8 .. 11: weight columns) of k16 step ST of tap TAP
  auto read_slot = [&](auto ST, auto TAP, auto R, int cb) {
    constexpr int st = decltype(ST)::value, T = decltype(TAP)::value, r = decltype(R)::value;
    constexpr int toff = (T / 3) * RSTRIDE + (T % 3) * PITCH;
    if constexpr (r < 8) {
      fa[st][r] = ld_frag(reinterpret_cast<const uint4*>(Ah + cb * HBYTES + toff + st * 32 + a_lane + (r >> 1) * 2 * RSTRIDE + (r & 1) * 64));
    } else {
      constexpr int j = (r - 8) >> 1, h = (r - 8) & 1;
      fw[st][2 * j + h] = ld_frag(reinterpret_cast<const uint4*>(Bs + (T % NS) * (BN * 128) + b_lane[st][h] + j * 4096));
    }
  };
  auto reads = [&](auto ST, auto TAP, int cb) {
    static_for<12>([&](auto R) { read_slot(ST, TAP, R, cb); });
  };
  // MFMA M (0 .. 23) of k16 step ST: term-major over the 4 x 2 tiles (a lo x w hi, a hi x w lo, a hi x w hi), so that back-to-back
  // MFMAs never wait on the same accumulator: per accumulator the sequence is conv3x3_split_ws_kernel's
  // FIRST: the chunk's first MFMA on this accumulator takes the instruction's constant 0 as its C operand instead of a zeroed register
  // tuple (0 + x: the same bits) — the flush then only adds, it does not clear 128 registers
  auto mfma_slot = [&](auto ST, auto M, auto FIRST) {
    constexpr int st = decltype(ST)::value, m = decltype(M)::value, term = m / 8, idx = m % 8, i = idx >> 1, j = idx & 1;
    const f16x8& a = fa[st][term == 0 ? 2 * i + 1 : 2 * i];
    const f16x8& w = fw[st][term == 1 ? 2 * j + 1 : 2 * j];
    if constexpr (decltype(FIRST)::value != 0 && term == 0) {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, w, zero, 0, 0, 0);
    } else {
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, w, acc[i][j], 0, 0, 0);
    }
  };

  // One tap of chunk c, as 48 pinned issue slots: MFMA m, the slot's side work, a scheduling fence (hipcc otherwise re-orders the
  // MFMAs into dependent chains and bunches the side work: the first form of this kernel, profiles/r06_ab_split_w512.txt).
  // Invariant at the tap's barrier: weight tiles it and it + 1 are in LDS (tile it + 1 was issued a tap ago), tile it - 1's slot is
  // free, this chunk's halo is complete; the other halo buffer was last read before the previous barrier.
  //   first half  (step 0's MFMAs): slots 0-11 the fragments of step 1, slots 12-15 the four LDS-DMA pieces of tile it + 2;
  //   second half (step 1's MFMAs): slots 0-11 the next tap's step-0 fragments, slots 12-23 the next chunk's staging —
  //     tap 0: coefficients + passes 0-2 loaded;  tap 2: passes 0, 1 converted and written;  tap 3: pass 2, then passes 3-5 loaded
  //     into the freed registers;  taps 5, 6, 7: passes 3, 4, 5;  tap 8: nothing (the buffer is handed over at the next barrier).
  auto body = [&](auto TAP, auto MORE, int c) {
    constexpr int T = decltype(TAP)::value;
    constexpr bool more = decltype(MORE)::value != 0;
    constexpr bool dma = T + 2 < NT || more;
    // the in-order VMEM queue behind the DMA of tile it + 1: only the halo (and coefficient) loads the previous tap issued
    if constexpr (T == 1 && more && PRO) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr ((T == 1 || T == 4) && more) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const int nc = T + 2 < NT ? c : c + 1;               // chunk of tile it + 2
    static_for<24>([&](auto M) {
      constexpr int m = decltype(M)::value;
      mfma_slot(IC<0>(), M, IC<(T == 0)>());
      if constexpr (m < 12) read_slot(IC<1>(), TAP, M, c & 1);
      else if constexpr (m < 16 && dma) dma_piece(nc, (T + 2) % NT, (T + 2) % NS, IC<m - 12>());
      __builtin_amdgcn_sched_barrier(0);
    });
    const int nb = (c + 1) & 1;
    static_for<24>([&](auto M) {
      constexpr int m = decltype(M)::value, sl = m - 12;
      mfma_slot(IC<1>(), M, IC<0>());
      if constexpr (m < 12) {
        if constexpr (T < NT - 1) read_slot(IC<0>(), IC<(T + 1) % NT>(), M, c & 1);
        else if constexpr (more) read_slot(IC<0>(), IC<0>(), M, nb);
      } else if constexpr (more) {
        if constexpr (T == 0) {
          if constexpr (sl == 0) { halo_chunk(c + 1); pro_load(c + 1); }
          if constexpr (sl >= 1 && sl <= 3) halo_load1(IC<sl - 1>());
        } else if constexpr (T == 2) {
          if constexpr (sl < 5) stage_slice(nb, IC<0>(), IC<sl < 5 ? sl : 0>());
          else if constexpr (sl < 10) stage_slice(nb, IC<1>(), IC<sl < 10 ? sl - 5 : 0>());
        } else if constexpr (T == 3) {
          if constexpr (sl < 5) stage_slice(nb, IC<2>(), IC<sl < 5 ? sl : 0>());
          else if constexpr (sl < 8) halo_load1(IC<sl < 8 ? sl - 2 : 3>());
        } else if constexpr (T >= 5 && T <= 7) {
          if constexpr (sl < 5) stage_slice(nb, IC<T - 2>(), IC<sl < 5 ? sl : 0>());
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (T == NT - 1) {                         // the 288-term partial of this channel chunk
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) tot[i][j][e] += acc[i][j][e];
    }
  };
  auto chunk_taps = [&](auto MORE, int c) {
    body(IC<0>(), MORE, c); body(IC<1>(), MORE, c); body(IC<2>(), MORE, c); body(IC<3>(), MORE, c); body(IC<4>(), MORE, c);
    body(IC<5>(), MORE, c); body(IC<6>(), MORE, c); body(IC<7>(), MORE, c); body(IC<8>(), MORE, c);
  };

  // ---- prologue: weight tiles 0 and 1, chunk 0's halo ----
  static_for<WPW>([&](auto R) { dma_piece(0, 0, 0, R); });
  static_for<WPW>([&](auto R) { dma_piece(0, 1, 1, R); });
  halo_chunk(0);
  pro_load(0);
  halo_load1(IC<0>()); halo_load1(IC<1>()); halo_load1(IC<2>());
  stage_pass(0, IC<0>()); stage_pass(0, IC<1>()); stage_pass(0, IC<2>());
  halo_load1(IC<3>()); halo_load1(IC<4>()); halo_load1(IC<5>());
  stage_pass(0, IC<3>()); stage_pass(0, IC<4>()); stage_pass(0, IC<5>());
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  reads(IC<0>(), IC<0>(), 0);
  for (int c = 0; c + 1 < nchunks; ++c) chunk_taps(IC<1>(), c);
  chunk_taps(IC<0>(), nchunks - 1);

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

// 1 = launched, 0 = not this kernel's shape / switched off, < 0 = error.  Default ON for launches with at least one 256-pixel x
// 128-channel tile per CU (same-box A/B, profiles/r06_ab_split_w512.txt: -3 ... -5 % per launch against conv3x3_split_ws_kernel, the
// whole f16x3 pipeline +2.4 %; with half as many workgroups as the 128-pixel kernel a launch below that size leaves CUs idle:
// 256 -> 256 @16x16 at B = 64 ran 83 us against 52).  PRG_SPLIT_W512=0: never; =2: every shape it covers (tests).
int try_launch_conv3x3_split_w512(const ConvLaunch<float>& L, hipStream_t s, int want_stats, int* gn_nsplit_out) {
  static const int on = [] { const char* e = std::getenv("PRG_SPLIT_W512"); return e ? std::atoi(e) : 1; }();
  if (!on) return 0;
  const ConvDesc& d = L.d;
  if (!(d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.Cout % 128 == 0 && d.Wout % 16 == 0 && d.Hout % 16 == 0 && !L.residual &&
        d.C0 % 32 == 0 && d.C1 % 32 == 0))
    return 0;
  constexpr int NS = 3, HB = 18 * ((18 * 144 + 255) / 256 * 256);
  const int tiles_x = d.Wout / 16, tiles_y = d.Hout / 16, tiles_n = d.Cout / 128;
  if (on != 2 && tiles_x * tiles_y * tiles_n * d.B < device_cu_count()) return 0;
  const int cpg = L.gn_groups > 0 ? d.Cout / L.gn_groups : 0;
  const int slabs = tiles_x * tiles_y * 4;
  const int f = want_stats && cpg % 8 == 0 && cpg <= 64 && (cpg & (cpg - 1)) == 0 && slabs <= kGnMaxSplit;
  if (want_stats && !f) return 0;
  const size_t lds = (size_t)2 * HB + NS * 128 * 128;
  if (gn_nsplit_out) *gn_nsplit_out = f ? slabs : 0;
  static DeviceOnce attr_done;
  if (!attr_done.done()) {
    PRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_w512_kernel<NS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    PRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_w512_kernel<NS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    attr_done.mark();
  }
  const dim3 grid(tiles_x * tiles_y * tiles_n * d.B);
  if (L.pro_a) conv3x3_split_w512_kernel<NS, true><<<grid, 256, lds, s>>>(L, tiles_x, tiles_y, tiles_n, f);
  else conv3x3_split_w512_kernel<NS, false><<<grid, 256, lds, s>>>(L, tiles_x, tiles_y, tiles_n, f);
  PRG_LAUNCH_CHECK();
  return 1;
}

}  // namespace prg
