// Compile-time switches of the attention kernels, in one place.  The shipped library is built with the defaults
// below; the Makefile's `tune` and `diag` targets are the only builds that define anything else, and their
// libraries go to tools/, never next to the package.
//
//   TMX_V16          1  bf16 launches, band workgroups: V turned into fp16 inside LDS, ONE fp16 term of the
//                       probabilities (34 matrix instructions per q-block and wave instead of 50).  0: V stays bf16,
//                       two-term probabilities.
//   TMX_V16_LIST     0  the same for the work-list workgroups.  Off: the conversion falls to the four waves that fetch
//                       the tiles and run the scalar walk, which every tile step already waits for (32k tokens,
//                       Qwen3-8B heads, same box: 157.5 -> 165.8 us with it; parity green either way).
//   TMX_ROW_PASSES   2  band workgroups, 16-bit output rows staged through LDS in two half-row passes.  1 (whole rows)
//                       needs 16 more registers: spills, 202 us against 160.
//   TMX_TUNING       -  (tune / diag builds) the launch reads TMX_BAND_RL (q-blocks per band run), TMX_BAND_MAP and
//                       TMX_EXP from the environment; TMX_EXP bit 1: work-list workgroups return at once, bit 2: band
//                       workgroups do (each role timed alone).
//   TMX_STAMPS       -  (diag build) per-wave cycle stamps of the loops' segments, written to a buffer of their own.
//
// Switches that were measured and REMOVED (the outcome is the record; none of them is in the code any more):
//   wave priority of the work-list workgroups (s_setprio 1, 3)            no effect
//   wave priority of the band workgroups (s_setprio 1, 3)                 the work list starves: 184 us against 161
//   band masks compared inside the loop instead of kept as lane masks    44 fewer spilled scalar registers, 4 spilled
//                                                                         vector registers: 163 us against 160
//   partial results stored with the non-temporal policy                  no effect
//   last q-blocks left to the work-list workgroups (tail share)          167.9 .. 177.6 us against 162.3
//   pieces walking their tile list in 2 / 4 interleaved parts            155.1-156.0 / 155.0-155.3 us against 154.3-155.5
//   output-store placements of the band workgroups (whole cache lines per instruction, 4-lane runs, whole rows)
//                                                                         none faster than 16 rows x 64 bytes
#pragma once

#ifndef TMX_V16
#define TMX_V16 1
#endif
#ifndef TMX_V16_LIST
#define TMX_V16_LIST 0
#endif
#ifndef TMX_ROW_PASSES
#define TMX_ROW_PASSES 2
#endif
#if !defined(TMX_TUNING) && !defined(TMX_STAMPS) && (TMX_V16 != 1 || TMX_V16_LIST != 0 || TMX_ROW_PASSES != 2)
#error "the shipped attention kernels are built with the default switches (see the table above)"
#endif
