// conv_split_ablate.h — timing-ablation switches of the f16x3 convolution kernels (conv_split.hip).
//
// The product build defines nothing: every constant below is false and the `if constexpr` branches they guard compile to
// nothing.  An ablation build (tools/split_ablate.sh builds conv_split.hip with -DPRG_SPLIT_ABLATE=<n> into its own .so) turns
// exactly ONE of them on.  Its RESULTS ARE GARBAGE by design — the point is the launch time with one cost removed; the readings
// are committed under profiles/ (r05_p64_ablations.txt, r05_split_ws_ablations.txt, r06_*).
#pragma once
#ifndef PRG_SPLIT_ABLATE
#define PRG_SPLIT_ABLATE 0
#endif
namespace prg {
namespace ablate {
constexpr int kWhich = PRG_SPLIT_ABLATE;
// conv3x3_split_p64_kernel (persistent, Cout = 64)
constexpr bool p64_no_mfma = kWhich == 11;      // fragment loads kept alive, no matrix instructions
constexpr bool p64_no_store = kWhich == 12;     // epilogue arithmetic kept, stores removed
constexpr bool p64_no_halo = kWhich == 13;      // no halo loads / prologue / split / LDS writes
constexpr bool p64_no_reads = kWhich == 14;     // no fragment reads (opaque register values)
constexpr bool p64_no_wdma = kWhich == 15;      // no weight LDS-DMA
// conv3x3_split_ws_kernel (wave-specialised, Cout % 128 == 0)
constexpr bool ws_no_mfma = kWhich == 21;
constexpr bool ws_no_halo = kWhich == 23;
constexpr bool ws_no_reads = kWhich == 24;
constexpr bool ws_no_wdma = kWhich == 25;
// A/B switches (a variant build flips them; the default is what ships)
#ifndef PRG_SPLIT_ILV
#define PRG_SPLIT_ILV 1
#endif
#ifndef PRG_SPLIT_PRIO
#define PRG_SPLIT_PRIO 0
#endif
// consumer fragment reads interleaved one per MFMA shadow (round 6: -3 ... -5.6 % per launch, same bits; 0 = eight reads in a row
// behind the twelfth MFMA, the round-5 order; profiles/r06_ab_split_interleaved_reads.txt)
constexpr bool kInterleave = PRG_SPLIT_ILV != 0;
constexpr int kConsumerPrio = PRG_SPLIT_PRIO;      // s_setprio of the consumer waves before their main loop (experiment)
}  // namespace ablate
}  // namespace prg
